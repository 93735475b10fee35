// nsf_coop_plan.cpp -- host-side plan of the cooperative (small-batch) NSF kernels, see nsf_coop.h.
#include "nsf_coop.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>

static int round_up_i(int v, int m) { return (v + m - 1) / m * m; }

// measured cross-overs against the throughput kernels: log_prob and the sampling direction win up to
// 12 288 rows; the training pass up to 8 192 rows (256 two-tile backward workgroups = one round; beyond, the throughput
// backward kernel's persistent workgroups are faster)
// (atomics: the setter is a process-wide tuning / test hook and may race with calls on other threads; every call reads
//  the threshold ONCE, and a training pass whose two halves would disagree is refused: nsf_train.hip, ws_family_*)
static std::atomic<int64_t> g_coop_max_rows{12288}, g_coop_train_rows{8192};
int64_t coop_max_rows() { return g_coop_max_rows.load(); }
int64_t coop_train_rows() { return g_coop_train_rows.load(); }
// tuning / test hook: route calls of <= `rows` rows to the cooperative kernels (0 or negative: never); returns the
// previous log_prob / sampling threshold.  (No environment variable, no lazily initialised sentinel: both thresholds
// are plain atomics with their measured defaults.)
extern "C" int64_t sbi_amd_nsf_set_coop_max_rows(int64_t rows) {
  const int64_t v = rows < 0 ? 0 : rows;
  g_coop_train_rows.store(v);
  return g_coop_max_rows.exchange(v);
}

// SBI_AMD_COOP_LEAN=0 (-DNSF_DEBUG builds only) keeps the forward pass of > 4096-row calls on the two-tile workgroups
bool coop_lean_forward() {
#ifdef NSF_DEBUG
  static int lean_env = -1;
  if (lean_env < 0) { const char* a = getenv("SBI_AMD_COOP_LEAN"); lean_env = a ? atoi(a) : 1; }
  return lean_env != 0;
#else
  return true;
#endif
}

static void add_mat(CoMat* m, int* off, int mtiles, int quads, int kind, int lin) {
  m->off = *off;
  m->mtiles = mtiles;
  m->quads = quads;
  m->kind = kind;
  m->lin = lin;
  *off += mtiles * quads * 256;
}
static void add_bias(CoBias* b, int* off, int mtiles, int kind, int lin) {
  b->off = *off;
  b->mtiles = mtiles;
  b->kind = kind;
  b->lin = lin;
  *off += 16 * mtiles;
}

int coop_build_plan(const NsfPlan& pl, int64_t n, int nt_force, bool training, CoopPlan* cp) {
  memset(cp, 0, sizeof(*cp));
  // shapes: residual-net conditioner (theta-dim >= 2), LULinear as one 16 x 16 MFMA tile, context K-steps in registers
  const bool wide = pl.H > 16 * NSF_HT;            // two hidden m-tiles per wave, eight K-quads (nsf_coop_wide_kernel.h)
  // x-dim: the context K-steps live in registers (two quads; the wide kernels, which have no fallback, four)
  if (pl.ctx_mlp || pl.D < 2 || pl.D > 16 || pl.C > (wide ? 64 : 32) || pl.H > 16 * NSF_HT_WIDE || pl.NB < 1 ||
      pl.NB > NSF_MAX_NB)
    return SBI_AMD_E_UNSUPPORTED;
  const int HT = wide ? NSF_HT_WIDE : NSF_HT;
  const int HQ = (pl.KSH + 3) / 4;     // K-quads of a hidden-K layer (13 or 16 K-steps; wide: 32)
  if (!wide && pl.shape[0].d_tr * pl.PT > 16) return SBI_AMD_E_UNSUPPORTED;   // final-layer tiles: four per wave
  int img = 0, plp = 0;
  for (int par = 0; par < 2; ++par) {
    const ShapeDesc& S = pl.shape[par];
    CoShape& c = cp->sh[par];
    c.KCQ = (pl.C + 15) / 16;
    c.nft = S.d_tr * pl.PT;
    int o = 0;
    add_mat(&c.W0, &o, HT, c.KCQ + 1, CO_K_W0, 0);
    for (int b = 0; b < pl.NB; ++b) {
      add_mat(&c.WC[b], &o, HT, c.KCQ, CO_K_PLAIN, 1 + 3 * b);
      add_mat(&c.W1[b], &o, HT, HQ, CO_K_PLAIN, 2 + 3 * b);
      add_mat(&c.W2[b], &o, HT, HQ, CO_K_PLAIN, 3 + 3 * b);
    }
    add_mat(&c.WF, &o, c.nft, HQ, CO_K_WF, S.fin);
    add_mat(&c.U, &o, 1, 1, CO_K_U, -1);
    add_mat(&c.L, &o, 1, 1, CO_K_L, -1);
    add_mat(&c.WFT, &o, HT, S.d_tr * pl.PT, CO_K_WFT, S.fin);
    for (int b = 0; b < pl.NB; ++b) {
      add_mat(&c.W1T[b], &o, HT, HQ, CO_K_PLAIN_T, 2 + 3 * b);
      add_mat(&c.W2T[b], &o, HT, HQ, CO_K_PLAIN_T, 3 + 3 * b);
    }
    add_mat(&c.W0T, &o, 1, HQ, CO_K_W0T, 0);
    add_mat(&c.UT, &o, 1, 1, CO_K_UT, -1);
    add_mat(&c.LT, &o, 1, 1, CO_K_LT, -1);
    add_mat(&c.UI, &o, 1, 1, CO_K_UI, -1);       // explicit inverses: the sampling direction (nsf_coop_wide_kernel.h)
    add_mat(&c.LI, &o, 1, 1, CO_K_LI, -1);
    for (int b = 0; b < pl.NB; ++b) add_mat(&c.WCT[b], &o, (pl.C + 15) / 16, HQ, CO_K_CTX_T, 1 + 3 * b);
    add_mat(&c.W0CT, &o, (pl.C + 15) / 16, HQ, CO_K_CTX_T, 0);
    c.o_bias = o;        // (256-aligned: every matrix block is 256 floats) the bias blocks and the log-det slot follow
    add_bias(&c.b0, &o, HT, 0, 0);
    for (int b = 0; b < pl.NB; ++b) {
      add_bias(&c.bc[b], &o, HT, 0, 1 + 3 * b);
      add_bias(&c.b1[b], &o, HT, 0, 2 + 3 * b);
      add_bias(&c.b2[b], &o, HT, 0, 3 + 3 * b);
    }
    add_bias(&c.bf, &o, c.nft, 1, S.fin);
    add_bias(&c.blu, &o, 1, 2, -1);
    c.o_ld = o;
    o += 4;
    o = round_up_i(o, 256);
    if (o > img) img = o;
    // partial-slab tiles: every linear with ceil((in + 1) / 16) n-tiles per m-tile (the bias is input index `in`)
    int tb = 0;
    for (int k = 0; k <= S.fin; ++k) {
      const LinDesc& L = S.lin[k];
      const int mts = k == S.fin ? c.nft : HT;
      c.dw_tb[k] = tb;
      c.dw_nnt[k] = (L.in + 1 + 15) / 16;
      tb += mts * c.dw_nnt[k];
    }
    c.dw_tail = tb * 256;
    const int slab = round_up_i(c.dw_tail + pl.D * (pl.D - 1) + 2 * pl.D + 1, 64);
    if (slab > plp) plp = slab;
  }
  cp->img_floats = img;
  cp->PLP = plp;

  // rows per workgroup: one 16-row tile while that still gives every CU at most two workgroups' worth of partial
  // slabs; two tiles per workgroup beyond (the A operands are then shared by both, the slabs halve)
#ifdef NSF_DEBUG
  static int nt_env = -1;   // debug aid: SBI_AMD_COOP_NT=1|2 forces the workgroup shape
  if (nt_env < 0) { const char* a = getenv("SBI_AMD_COOP_NT"); nt_env = a ? atoi(a) : 0; }
  if (nt_force <= 0) nt_force = nt_env;
#endif
  int NT = nt_force > 0 ? nt_force : (n > 4096 ? 2 : 1);
  if (NT > CO_MAX_NT) NT = CO_MAX_NT;
  cp->MT = wide ? 2 : 1;
  const int nt_first = NT;
retry_nt:
  cp->NT = NT;
  cp->R = 16 * NT;
  cp->RS = cp->R + 4;
  cp->ZS = 17;
  cp->PSW = 16 * pl.PT + 1;
  const int d_tr_max = pl.shape[0].d_tr;
  cp->DSTR = d_tr_max * cp->PSW;
  if ((cp->DSTR & 1) == 0) cp->DSTR += 1;
  cp->s_blk = HT;
  cp->s_par = HT + 4 * HT * pl.NB;
  cp->slots = cp->s_par + d_tr_max * pl.PT;
  // conditioner-input tile [z_id ; context ; 1 ; 0 ...]^T: rows cover d W0's n-tiles and, from row d_id, d Wc's
  const int d_id_max = pl.shape[0].d_id > pl.shape[1].d_id ? pl.shape[0].d_id : pl.shape[1].d_id;
  const int nt0 = (d_id_max + pl.C + 1 + 15) / 16, ntc = (pl.C + 1 + 15) / 16;
  cp->ct_rows = 16 * nt0 > d_id_max + 16 * ntc ? 16 * nt0 : d_id_max + 16 * ntc;
  int o = 0;
  cp->o_zs = o;  o += round_up_i(cp->R * cp->ZS + 16, 4);
  cp->o_gys = o; o += round_up_i(cp->R * cp->ZS + 16, 4);
  cp->o_gzs = o; o += round_up_i(cp->R * cp->ZS + 16, 4);
  cp->o_w = o;   o += cp->R;
  cp->o_pst = o; o += round_up_i(cp->R * cp->DSTR, 4);
  cp->o_ex = o;  o += 2 * HT * NT * 256;
  cp->o_ldp = o; o += 8 * cp->R;
  if (training) {
    cp->o_gt = o;  o += 2 * 16 * HT * cp->RS;
    cp->o_at = o;  o += 2 * (16 * HT + 1) * cp->RS;
    cp->o_ct = o;  o += cp->ct_rows * cp->RS;
    cp->o_lut = o; o += 4 * 17 * cp->RS;
    cp->o_ctx = o; o += (pl.C > 32 ? 64 : 32) * cp->R;
  }
  cp->lds_floats = round_up_i(o, 4);
  if (4ll * cp->lds_floats > NSF_LDS_LIMIT_BYTES) {
    // wide nets: the 128-feature transposed tiles of two row tiles may not fit next to a large spline staging area;
    // one row tile per workgroup always does
    if (wide && NT == 2 && nt_first == 2 && nt_force <= 0) { NT = 1; goto retry_nt; }
    return SBI_AMD_E_LDS;
  }
  cp->grid = (int)((n + cp->R - 1) / cp->R);
  return 0;
}

void coop_make_consts(const NsfPlan& pl, const CoopPlan& cp, CoK* k) {
  memset(k, 0, sizeof(*k));
  k->D = pl.D; k->C = pl.C; k->H = pl.H; k->NB = pl.NB; k->T = pl.T; k->P = pl.P;
  k->KCQ = cp.sh[0].KCQ;
  k->HT = 4 * cp.MT;
  k->HQ = cp.sh[0].W1[0].quads;
  const CoShape& c0 = cp.sh[0];
  k->sA = pl.NB > 1 ? c0.W1[1].off - c0.W1[0].off : 0;
  k->sT = pl.NB > 1 ? c0.W1T[1].off - c0.W1T[0].off : 0;
  k->sC = pl.NB > 1 ? c0.WCT[1].off - c0.WCT[0].off : 0;
  k->sB = pl.NB > 1 ? c0.b1[1].off - c0.b1[0].off : 0;
  k->img_floats = cp.img_floats;
  k->ZS = cp.ZS; k->RS = cp.RS; k->PSW = cp.PSW; k->DSTR = cp.DSTR;
  k->o_zs = cp.o_zs; k->o_gys = cp.o_gys; k->o_gzs = cp.o_gzs; k->o_w = cp.o_w; k->o_pst = cp.o_pst;
  k->o_ex = cp.o_ex; k->o_ldp = cp.o_ldp; k->o_gt = cp.o_gt; k->o_at = cp.o_at; k->o_ct = cp.o_ct;
  k->o_lut = cp.o_lut; k->o_ctx = cp.o_ctx; k->ct_rows = cp.ct_rows;
  k->slots = cp.slots; k->s_blk = cp.s_blk; k->s_par = cp.s_par;
  k->PLP = cp.PLP;
  k->ntc = (pl.C + 1 + 15) / 16;
  k->nnh = (pl.H + 1 + 15) / 16;
  k->ablate = pl.ablate;
  k->B = pl.B; k->min_w = pl.min_w; k->min_h = pl.min_h; k->min_d = pl.min_d; k->inv_sqrt_h = pl.inv_sqrt_h;
  k->one_minus_kw = pl.one_minus_kw; k->one_minus_kh = pl.one_minus_kh; k->d_const = pl.d_const; k->log_z = pl.log_z;
  for (int par = 0; par < 2; ++par) {
    const ShapeDesc& S = pl.shape[par];
    const CoShape& c = cp.sh[par];
    CoKP& q = k->p[par];
    q.d_id = S.d_id; q.d_tr = S.d_tr; q.in0 = S.in0; q.nft = c.nft;
    q.nnt0 = (S.in0 + 1 + 15) / 16;
    q.dw_tail = c.dw_tail;
    q.w0 = c.W0.off; q.wc0 = c.WC[0].off; q.w10 = c.W1[0].off; q.w20 = c.W2[0].off; q.wf = c.WF.off;
    q.u = c.U.off; q.l = c.L.off; q.wft = c.WFT.off; q.w1t0 = c.W1T[0].off; q.w2t0 = c.W2T[0].off;
    q.w0t = c.W0T.off; q.ut = c.UT.off; q.lt = c.LT.off; q.wct0 = c.WCT[0].off; q.w0ct = c.W0CT.off;
    q.ui = c.UI.off; q.li = c.LI.off;
    q.b0 = c.b0.off; q.bc0 = c.bc[0].off; q.b10 = c.b1[0].off; q.b20 = c.b2[0].off; q.bf = c.bf.off;
    q.blu = c.blu.off; q.ld = c.o_ld;
  }
}
