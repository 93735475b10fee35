// nsf_gtrain.hip -- host side of the GENERIC NSF training pass (nsf_gtrain_kernel.h): plan, workspace layout, the
// forward / backward halves nsf_train.hip falls back to when the wave-specialised backward kernel refuses a shape.
#include <hip/hip_runtime.h>
#include <string.h>
#define NSF_GTRAIN_MAIN_TU
#include "nsf_gtrain_kernel.h"

int nsf_log_prob_stash(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats, const float* theta,
                       const float* x, int64_t n, int64_t x_rows, float* logp_out, float* noise_out,
                       float* z_stash, float* astash, float* pstash, void* stream, bool fp32_bin);

static int g_round_up(int v, int m) { return (v + m - 1) / m * m; }

// plan for n rows: the forward kernel's weight image + this kernel's per-wave scratch; nw = waves per workgroup
static int g_build_plan(const sbi_amd_nsf_config* cfg, int64_t n, NsfPlan* pl, GTrainPlan* gp, int* nw_out) {
  int rc = nsf_build_plan(cfg, 1, pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  if (pl->ctx_mlp) return SBI_AMD_E_UNSUPPORTED;     // theta-dim 1 trains on the wave-specialised kernel only
  {                                                  // the stash-writing forward pass must fit as well
    NsfPlan fw;
    int fnw;
    if ((rc = nsf_plan_for_rows(cfg, n, &fw, &fnw))) return rc;
  }
  const int D = pl->D, C = pl->C, NB = pl->NB;
  const int d_id_max = pl->shape[0].d_id > pl->shape[1].d_id ? pl->shape[0].d_id : pl->shape[1].d_id;
  const int d_tr_max = pl->shape[0].d_tr;
  if (d_id_max > 32 || NB < 1) return SBI_AMD_E_UNSUPPORTED;
  memset(gp, 0, sizeof(*gp));
  gp->PTW = 16 * pl->PT;
  int o = 0;
  gp->sc_zs = o; o += 16 * pl->ZW + 16;
  gp->sc_gy = o; o += 16 * pl->ZW + 16;
  gp->sc_gz = o; o += 16 * pl->ZW + 16;
  gp->sc_us = o; o += 16 * pl->ZW + 16;
  gp->sc_cin = o; o += 16 * pl->CINW + 16;
  gp->sc_pst = o; o += 2 * pl->DS;
  gp->sc_total = g_round_up(o, 4);
  gp->gp_planes = d_tr_max * gp->PTW / 16;
  gp->g_planes = (1 + 3 * NB) * 4;
  gp->lu_planes = (D + 1 + 15) / 16 + (D + 15) / 16;
  gp->act_w = (1 + 2 * NB) * 64;
  gp->cin_w = g_round_up(d_id_max + C, 16);
  gp->lua_w = 2 * g_round_up(D, 16);
  const int pmax = pl->shape[0].n_params > pl->shape[1].n_params ? pl->shape[0].n_params : pl->shape[1].n_params;
  gp->o_dU = g_round_up(pmax, 4);
  gp->o_dUb = gp->o_dU + (D + 1) * D;
  gp->o_dL = g_round_up(gp->o_dUb + D + 1, 4);
  gp->o_dLb = gp->o_dL + D * D;
  gp->slab = g_round_up(gp->o_dLb + D, 4);
  // up to 8 waves (two per SIMD; the kernel holds a residual block's activations in registers, <= 256 VGPRs) when
  // there are enough rows; any count works (16 rows per wave): the largest that fits next to the weight image
  int nw = 8;
  while (nw > 1 && (n + 16 * nw - 1) / (16 * nw) < 256) nw >>= 1;
  for (; nw >= 1; --nw)
    if (4ll * ((int64_t)pl->lds_w_floats + (int64_t)nw * gp->sc_total) <= NSF_LDS_LIMIT_BYTES) break;
  if (nw < 1) return SBI_AMD_E_LDS;
  *nw_out = nw;
  return 0;
}

struct GWs {
  int64_t stash, noise, logp, gza, gzb, gpl, gbuf, lug, act, cin, lua, part, sums, ast, total, npad;
  int nchunks;
};
static GWs g_ws_layout(const NsfPlan& pl, const GTrainPlan& gp, int64_t n) {
  GWs w;
  int64_t o = 0;
  auto take = [&](int64_t sz) { const int64_t at = o; o += (sz + 3) / 4 * 4; return at; };
  const int D = pl.D, T = pl.T;
  w.npad = (n + MAF_DW_CHUNK - 1) / MAF_DW_CHUNK * MAF_DW_CHUNK;
  w.nchunks = (int)(w.npad / MAF_DW_CHUNK);
  w.stash = take((int64_t)T * n * D);
  w.noise = take(n * D);
  w.logp = take(n);
  w.gza = take(n * D);
  w.gzb = take(n * D);
  w.gpl = take((int64_t)gp.gp_planes * w.npad * 16);
  w.gbuf = take((int64_t)gp.g_planes * w.npad * 16);
  w.lug = take((int64_t)gp.lu_planes * w.npad * 16);
  w.act = take(w.npad * gp.act_w);
  w.cin = take(w.npad * gp.cin_w);
  w.lua = take(w.npad * gp.lua_w);
  w.part = take((int64_t)T * w.nchunks * gp.slab);
  w.sums = take((int64_t)T * gp.slab);
  w.ast = take((int64_t)T * ((n + 15) / 16) * NSF_AST_SLOTS(pl.NB) * 1024);
  w.total = o;
  return w;
}

int64_t nsf_g_workspace_floats(const sbi_amd_nsf_config* cfg, int64_t n) {
  NsfPlan pl;
  GTrainPlan gp;
  int nw;
  const int rc = g_build_plan(cfg, n > 0 ? n : 1, &pl, &gp, &nw);
  if (rc) return rc;
  return g_ws_layout(pl, gp, n > 0 ? n : 1).total;
}

int nsf_g_train_forward(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats, const float* theta,
                        const float* x, int64_t n, int64_t x_rows, float* logp_out, float* workspace, void* stream) {
  NsfPlan pl;
  GTrainPlan gp;
  int nw;
  int rc = g_build_plan(cfg, n, &pl, &gp, &nw);
  if (rc) return rc;
  const GWs w = g_ws_layout(pl, gp, n);
  rc = nsf_log_prob_stash(cfg, packed, zstats, theta, x, n, x_rows, workspace + w.logp, workspace + w.noise,
                          workspace + w.stash, workspace + w.ast, nullptr, stream, false);
  if (rc) return rc;
  if (logp_out) {
    hipError_t e = hipMemcpyAsync(logp_out, workspace + w.logp, sizeof(float) * n, hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

const float* nsf_g_logp(const sbi_amd_nsf_config* cfg, int64_t n, const float* workspace) {
  NsfPlan pl;
  GTrainPlan gp;
  int nw;
  if (g_build_plan(cfg, n, &pl, &gp, &nw)) return nullptr;
  return workspace + g_ws_layout(pl, gp, n).logp;
}

template int nsf_gbwd_launch_k<4>(const NsfPlan&, const GTrainPlan&, int, const GBwdArgs&, hipStream_t);
template int nsf_gbwd_launch_k<5>(const NsfPlan&, const GTrainPlan&, int, const GBwdArgs&, hipStream_t);
template int nsf_gbwd_launch_k<8>(const NsfPlan&, const GTrainPlan&, int, const GBwdArgs&, hipStream_t);
template int nsf_gbwd_launch_k<10>(const NsfPlan&, const GTrainPlan&, int, const GBwdArgs&, hipStream_t);
template int nsf_gbwd_launch_k<16>(const NsfPlan&, const GTrainPlan&, int, const GBwdArgs&, hipStream_t);

int nsf_g_train_backward(const sbi_amd_nsf_config* cfg, const float* params, const float* packed, const float* zstats,
                         const float* x, int64_t n, int64_t x_rows, const float* row_weight, float uniform_weight,
                         float* grad_out, float* grad_theta_out, float* workspace, void* stream) {
  NsfPlan pl;
  GTrainPlan gp;
  int nw;
  int rc = g_build_plan(cfg, n, &pl, &gp, &nw);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const GWs w = g_ws_layout(pl, gp, n);
  const int D = pl.D, C = pl.C, T = pl.T, NB = pl.NB, H = pl.H;
  float* gz[2] = {workspace + w.gza, workspace + w.gzb};
  const int64_t gts = w.npad * 16;
  for (int t = T - 1; t >= 0; --t) {
    const int par = t & 1;
    const ShapeDesc& S = pl.shape[par];
    GBwdArgs a;
    a.packed = packed; a.zstats = zstats;
    a.z_in = workspace + w.stash + (int64_t)t * n * D;
    a.x = x;
    a.gz_up = (t == T - 1) ? workspace + w.noise : gz[(t + 1) & 1];
    a.row_w = row_weight; a.uni_w = uniform_weight;
    a.n = n; a.x_rows = x_rows;
    a.gz_dn = gz[t & 1];
    a.grad_theta = grad_theta_out;
    a.astash = workspace + w.ast;
    a.GP = workspace + w.gpl; a.G = workspace + w.gbuf; a.LUG = workspace + w.lug;
    a.ACT = workspace + w.act; a.CIN = workspace + w.cin; a.LUA = workspace + w.lua;
    a.npad = w.npad; a.t = t; a.is_last = (t == T - 1); a.par = par;
    switch (cfg->K) {
      case 4: rc = nsf_gbwd_launch_k<4>(pl, gp, nw, a, st); break;
      case 5: rc = nsf_gbwd_launch_k<5>(pl, gp, nw, a, st); break;
      case 8: rc = nsf_gbwd_launch_k<8>(pl, gp, nw, a, st); break;
      case 10: rc = nsf_gbwd_launch_k<10>(pl, gp, nw, a, st); break;
      case 16: rc = nsf_gbwd_launch_k<16>(pl, gp, nw, a, st); break;
      default: rc = SBI_AMD_E_UNSUPPORTED;
    }
    if (rc) return rc;
    // ---- weight gradients of this transform: every linear (wide inputs in pieces of 64 columns) + LULinear's dU, dL
    MafDwArgs d;
    memset((void*)&d, 0, sizeof(d));
    int nl = 0;
    auto add = [&](const float* G, const float* A, int lda, int out, int in_total, int group, int gpad, int g_w,
                   int g_b, int gperm, int aperm) {
      for (int col0 = 0; col0 < in_total && nl < MAF_DW_MAX_LIN; col0 += 64) {
        MafLin& L = d.lin[nl++];
        L.G = G; L.gts = gts; L.A = A + col0; L.lda = lda;
        L.out = out; L.in = in_total - col0 < 64 ? in_total - col0 : 64; L.in_total = in_total; L.col0 = col0;
        L.group = group; L.group_pad = gpad; L.g_w = g_w; L.g_b = g_b; L.kind = 1;
        L.gperm = gperm; L.aperm = aperm;
      }
    };
    const LinDesc& LF = S.lin[S.fin];
    // (fragment order: the G planes of the hidden layers and the ACT rows; natural: GP, LUG, CIN, LUA)
    add(a.GP, a.ACT, gp.act_w, LF.out, H, pl.P, gp.PTW, LF.g_w, LF.g_b, 0, 1);                 // final layer
    for (int b = 0; b < NB; ++b) {
      const LinDesc& Lc = S.lin[1 + 3 * b];
      const LinDesc& L1 = S.lin[2 + 3 * b];
      const LinDesc& L2 = S.lin[3 + 3 * b];
      add(a.G + (int64_t)(4 * (3 + 3 * b)) * gts, a.ACT + 64 * (1 + 2 * b), gp.act_w, H, H, H, 64, L2.g_w, L2.g_b,
          1, 1);
      add(a.G + (int64_t)(4 * (2 + 3 * b)) * gts, a.ACT + 64 * (2 + 2 * b), gp.act_w, H, H, H, 64, L1.g_w, L1.g_b,
          1, 1);
      add(a.G + (int64_t)(4 * (1 + 3 * b)) * gts, a.CIN + S.d_id, gp.cin_w, H, C, H, 64, Lc.g_w, Lc.g_b, 1, 0);
    }
    add(a.G, a.CIN, gp.cin_w, H, S.in0, H, 64, S.lin[0].g_w, S.lin[0].g_b, 1, 0);            // initial layer
    const int du_pad = 16 * ((D + 1 + 15) / 16), dz_pad = 16 * ((D + 15) / 16), dwp = gp.lua_w >> 1;
    add(a.LUG, a.LUA, gp.lua_w, D + 1, D, D + 1, du_pad, gp.o_dU, gp.o_dUb, 0, 0);           // dU (+ logabsdet column)
    add(a.LUG + (int64_t)(du_pad >> 4) * gts, a.LUA + dwp, gp.lua_w, D, D, D, dz_pad, gp.o_dL, gp.o_dLb, 0, 0);   // dL, d bias
    if (nl >= MAF_DW_MAX_LIN) return SBI_AMD_E_UNSUPPORTED;
    d.n = n; d.rows_per_chunk = MAF_DW_CHUNK; d.nchunks = w.nchunks; d.n_layer = gp.slab; d.D = D; d.P = pl.P;
    d.partial = workspace + w.part + (int64_t)t * w.nchunks * gp.slab;
    rc = maf_launch_dw(d, nl, st);
    if (rc) return rc;
  }
  rc = maf_launch_reduce(workspace + w.part, workspace + w.sums, gp.slab, w.nchunks, T, st);
  if (rc) return rc;
  hipLaunchKernelGGL(nsf_gfinish_kernel, dim3((pl.n_params + 255) / 256), dim3(256), 0, st, pl, gp, params,
                     workspace + w.sums, grad_out);
  return (int)hipGetLastError();
}
