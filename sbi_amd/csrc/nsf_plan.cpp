// nsf_plan.cpp -- host-side plan builder (see nsf_plan.h).
#include "nsf_plan.h"
#include "debug_env.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static int round_up(int v, int m) { return (v + m - 1) / m * m; }
static int two_odd_at_least(int v) {   // smallest 2*odd >= v
  int x = (v + 1) / 2;                 // ceil(v/2)
  if ((x & 1) == 0) x += 1;
  return 2 * x;
}

static int supported_bins(int K) { return K == 4 || K == 5 || K == 8 || K == 10 || K == 16; }

static int check_cfg(const sbi_amd_nsf_config* c) {
  if (!c) return SBI_AMD_E_BADARG;
  if (c->D < 1 || c->C < 1 || c->H < 1 || c->T < 1 || c->NB < 0) return SBI_AMD_E_BADARG;
  if (c->H > 16 * NSF_HT || c->T > NSF_MAX_T || c->NB > NSF_MAX_NB || !supported_bins(c->K))
    return SBI_AMD_E_UNSUPPORTED;
  if (c->D > 64 || c->C > 256) return SBI_AMD_E_UNSUPPORTED;
  if (c->min_bin_width * c->K > 1.0f || c->min_bin_height * c->K > 1.0f) return SBI_AMD_E_BADARG;
  return 0;
}

static void set_lin(LinDesc* L, int* g, int* l, int out, int in, int bias_pad, int ksteps_fixed,
                    int min_rows = 0) {
  L->out = out;
  L->in = in;
  L->ksteps = ksteps_fixed > 0 ? ksteps_fixed : round_up((in + 3) / 4, 4);
  // B-from-LDS layers: both operands are zero padded up to 4*ksteps columns.  Hidden-K
  // layers (B = activation fragments, exact zeros beyond H via the zero row): the
  // tail K-steps may run into the following row; those weights meet a zero B.
  L->ldk = two_odd_at_least(ksteps_fixed > 0 ? in : 4 * L->ksteps);
  L->g_w = *g;
  *g += out * in;
  L->g_b = *g;
  *g += out;
  L->l_w = *l;
  // rows [out, rows_alloc) are all-zero: row `out` is what out-of-range MFMA A rows read, and the
  // backward's transposed K loop walks rows 0 .. 4*KSH-1 of the hidden x hidden layers unpredicated
  int rows_alloc = out + 1;
  if (min_rows > rows_alloc) rows_alloc = min_rows;
  L->rows = rows_alloc;
  *l += rows_alloc * L->ldk;
  L->l_b = *l;
  *l += bias_pad;
}

int nsf_build_plan(const sbi_amd_nsf_config* cfg, int nw, NsfPlan* pl) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  memset(pl, 0, sizeof(*pl));
  const int D = cfg->D, C = cfg->C, H = cfg->H, K = cfg->K, T = cfg->T;
  const int ctx_mlp = (D == 1);
  const int NB = ctx_mlp ? 0 : cfg->NB;   // the context-only conditioner has no residual blocks
  pl->ctx_mlp = ctx_mlp;
  pl->D = D; pl->C = C; pl->H = H; pl->K = K; pl->T = T; pl->NB = NB;
  pl->P = 3 * K - 1;
  pl->PT = (pl->P + 15) / 16;
  pl->KSH = ((H + 3) / 4 == 13) ? 13 : 16;   // kernels are instantiated for 13 (H=49..52) and 16
  pl->B = cfg->tail_bound;
  pl->min_w = cfg->min_bin_width; pl->min_h = cfg->min_bin_height; pl->min_d = cfg->min_derivative;
  pl->lu_eps = cfg->lu_eps;
  pl->sqrt_h = (float)sqrt((double)H);
  pl->inv_sqrt_h = (float)(1.0 / sqrt((double)H));
  pl->one_minus_kw = (float)(1.0 - (double)cfg->min_bin_width * K);
  pl->one_minus_kh = (float)(1.0 - (double)cfg->min_bin_height * K);
  pl->d_const = (float)log(exp(1.0 - (double)cfg->min_derivative) - 1.0);
  pl->log_z = (float)(0.5 * D * log(2.0 * M_PI));
  pl->ablate = sbi_amd_dbg_ablate();   // debug aid, read once per process and announced on stderr

  for (int par = 0; par < 2; ++par) {
    ShapeDesc* s = &pl->shape[par];
    // create_alternating_binary_mask (torchutils.py:396-410): even transforms
    // transform the even feature indices, odd ones the odd indices.
    s->d_tr = ctx_mlp ? 1 : ((par == 0) ? (D + 1) / 2 : D / 2);   // D == 1: dummy mask [1] every transform
    s->d_id = D - s->d_tr;
    s->in0 = s->d_id + C;
    int g = 0, l = 0;
    const int hb = 16 * NSF_HT;
    const int tr_rows = 4 * pl->KSH + 1;   // transposed (backward) K loops walk 4*KSH rows
    set_lin(&s->lin[0], &g, &l, H, s->in0, hb, 0, tr_rows);
    if (ctx_mlp) {
      set_lin(&s->lin[1], &g, &l, H, H, hb, pl->KSH, tr_rows);
      s->fin = 2;
    } else {
      for (int b = 0; b < NB; ++b) {
        set_lin(&s->lin[1 + 3 * b], &g, &l, H, C, hb, 0);
        set_lin(&s->lin[2 + 3 * b], &g, &l, H, H, hb, pl->KSH, tr_rows);
        set_lin(&s->lin[3 + 3 * b], &g, &l, H, H, hb, pl->KSH, tr_rows);
      }
      s->fin = 1 + 3 * NB;
    }
    l = round_up(l, 4);   // the final layer (+ LU) is staged on its own by the backward kernel's overlay mode
    if (l > pl->hidden_img_floats) pl->hidden_img_floats = l;
    s->final_off = l;
    set_lin(&s->lin[s->fin], &g, &l, s->d_tr * pl->P, H, s->d_tr * 16 * pl->PT, pl->KSH);
    s->g_lu = g;
    if (!ctx_mlp) g += D * (D - 1) + 2 * D;   // lower, upper, unconstrained diag, bias
    s->n_params = g;
    { const int lus = D <= 16 ? 16 : D;   // dense U, L padded to 16 x 16 for D <= 16
      l = round_up(l, 4);   // 16-byte aligned: the backward kernel reads matrix rows as float4
      s->l_U = l; l += lus * lus;
      s->l_L = l; l += lus * lus; }
    s->l_lub = l; l += D + 1;   // bias, then sum_i log U_ii
    // everything above is what the training kernels stage; the explicit inverses (sampling direction only)
    // come last so that the backward kernel can leave them out of its LDS image
    l = round_up(l, 4);
    if (l > pl->lds_w_train_floats) pl->lds_w_train_floats = l;
    s->l_Ui = s->l_Li = -1;
    if (!ctx_mlp && D <= 16) { s->l_Ui = l; l += 256; s->l_Li = l; l += 256; }
    s->lds_floats = round_up(l + 8, 4);
    if (s->lds_floats > pl->lds_w_floats) pl->lds_w_floats = s->lds_floats;
  }
  pl->img_floats = pl->lds_w_floats;
  int off = 0;
  for (int t = 0; t < T; ++t) {
    pl->g_layer[t] = off;
    off += pl->shape[t & 1].n_params;
  }
  pl->n_params = off;

  // per-wave scratch; every row stride is 2*odd (bank-conflict-free b32 access
  // by (row = lane&15, k-slot = lane>>4) lane pairs)
  pl->ZW = two_odd_at_least(D);
  pl->CW = two_odd_at_least(C);
  int d_id_max = pl->shape[0].d_id > pl->shape[1].d_id ? pl->shape[0].d_id : pl->shape[1].d_id;
  int need = d_id_max + 4 * round_up((C + 3) / 4, 4);       // context-layer K-steps read past C
  int need2 = 4 * round_up((d_id_max + C + 3) / 4, 4);      // initial-layer K-steps
  pl->CINW = two_odd_at_least(need > need2 ? need : need2);
  // spline-parameter staging: P (made odd) floats per row => conflict-free per-row reads;
  // slot stride == 16 (mod 32) puts the second dim slot of a 32-lane half on the other banks
  pl->PSW = 16 * pl->PT + 1;   // odd (conflict-free per-row reads) and wide enough for all 16*PT outputs
  pl->DS = 16 * pl->PSW;
  while ((pl->DS & 31) != 16) pl->DS += 1;
  // scratch = flow state rows, context rows, and TWO spline-parameter buffers (the final-layer GEMM
  // of chunk c+1 is issued under the spline of chunk c).  The conditioner-input rows alias the
  // second buffer and the LU temporaries the first: both are dead while those are live.
  int d_tr_max = pl->shape[0].d_tr;
  pl->DCH = 2 < d_tr_max ? 2 : d_tr_max;   // a spline task occupies a lane PAIR: 16 rows x 2 dims per pass
  int pst_sz = pl->DCH * pl->DS;
  if (pst_sz < 16 * pl->CINW) pst_sz = 16 * pl->CINW;
  if (pst_sz < 16 * pl->ZW + 16) pst_sz = 16 * pl->ZW + 16;
  int o = 0;
  pl->sc_zs = o; o += 16 * pl->ZW;
  pl->sc_cs = o;
  if (C > 16) o += 16 * pl->CW;   // C <= 16: the standardized context lives in 4 registers per lane
  pl->sc_pst = o; o += pst_sz;
  // more than two waves per SIMD (nw > 8): ONE staging buffer per wave (4.9 instead of 9.1 KB at the defaults), the
  // final-layer GEMM of a chunk then runs before that chunk's spline instead of under the previous one's; latency
  // is covered by the third wave of the SIMD instead of by the second buffer
  if (nw > 8) {
    pl->sc_pst2 = pl->sc_pst;
  } else {
    pl->sc_pst2 = o; o += pst_sz;
  }
  pl->sc_us = pl->sc_pst;
  pl->sc_cin = pl->sc_pst2;
  pl->sc_total = round_up(o, 4);
  if (4ll * ((int64_t)pl->lds_w_floats + (int64_t)nw * pl->sc_total) > NSF_LDS_LIMIT_BYTES) return SBI_AMD_E_LDS;
  return 0;
}

int nsf_plan_for_rows(const sbi_amd_nsf_config* cfg, int64_t n, NsfPlan* pl, int* nw_out, bool wide) {
  // 16 rows per wave; aim for >= 256 workgroups (one per CU) before growing the workgroup: 8, 4, 2, 1 waves.
  // `wide` (the sampling direction): 12 waves (3 per SIMD, single staging buffer per wave) when there are enough rows
  // and the image leaves room; SBI_AMD_ABLATE bit 4096 switches it off (the density direction is compiled for <= 8 waves and never takes it).
  const int abl = sbi_amd_dbg_ablate();
  if (wide && !(abl & (1024 | 4096)) && (n + 16 * 12 - 1) / (16 * 12) >= 256) {
    if (nsf_build_plan(cfg, 12, pl) == 0) { *nw_out = 12; return 0; }
  }
  int nw = 8;
  while (nw > 1 && (n + 16 * nw - 1) / (16 * nw) < 256) nw >>= 1;
  if ((sbi_amd_dbg_ablate() & 1024) && nw > 4) nw = 4;   // debug aid: forward kernel with one wave per SIMD
  for (; nw >= 1; nw >>= 1) {
    int rc = nsf_build_plan(cfg, nw, pl);
    if (rc == 0) { *nw_out = nw; return 0; }
    if (rc != SBI_AMD_E_LDS) return rc;
  }
  return SBI_AMD_E_LDS;
}

extern "C" int64_t sbi_amd_nsf_param_count(const sbi_amd_nsf_config* cfg) {
  NsfPlan pl;
  int rc = nsf_build_plan(cfg, 1, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  return pl.n_params;
}
extern "C" int64_t sbi_amd_nsf_layer_offset(const sbi_amd_nsf_config* cfg, int32_t t) {
  NsfPlan pl;
  int rc = nsf_build_plan(cfg, 1, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  if (t < 0 || t >= pl.T) return SBI_AMD_E_BADARG;
  return pl.g_layer[t];
}
extern "C" int64_t sbi_amd_nsf_lu_offset(const sbi_amd_nsf_config* cfg, int32_t t) {
  NsfPlan pl;
  int rc = nsf_build_plan(cfg, 1, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  if (t < 0 || t >= pl.T) return SBI_AMD_E_BADARG;
  return pl.g_layer[t] + pl.shape[t & 1].g_lu;
}
extern "C" int sbi_amd_nsf_abi_version(void) { return SBI_AMD_NSF_ABI_VERSION; }
extern "C" const char* sbi_amd_nsf_arch(void) { return "gfx950"; }
