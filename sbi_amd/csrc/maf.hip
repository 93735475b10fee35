// maf.hip -- host side of the maf_rqs path (plan, C ABI of include/sbi_amd_maf.h) + the num_bins = 10 kernel
// instantiations (other bin counts: maf_k{4,5,8,16}.hip, separate translation units for a parallel build).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#define MAF_MAIN_TU
#include "maf_kernel.h"

// launch wrappers of the two non-template kernels defined in this translation unit: the generic NSF training
// path (nsf_gtrain.hip) contracts its weight gradients with the same kernels
int maf_launch_dw(const MafDwArgs& d, int nlin, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute((const void*)maf_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     MAF_DW_LDS_BYTES);
  if (e != hipSuccess) return (int)e;
  // the callers describe whole linears; a workgroup keeps at most 4 waves x MAF_DW_MTW m-tiles of accumulators, so
  // wider outputs (the final layer: D x 16 PT columns) are cut into pieces here; largest pieces first
  MafDwArgs x = d;
  int np = 0;
  auto flush = [&]() {
    if (np == 0) return 0;
    hipLaunchKernelGGL(maf_dw_kernel, dim3(d.nchunks, np), dim3(256), (size_t)MAF_DW_LDS_BYTES, st, x);
    np = 0;
    return (int)hipGetLastError();
  };
  const int cap = 4 * MAF_DW_MTW;
  for (int pass = 0; pass < 2; ++pass)          // pass 0: full-size pieces, pass 1: the remainders
    for (int i = 0; i < nlin; ++i) {
      const MafLin& L = d.lin[i];
      const int mcols = (L.out + L.group - 1) / L.group * L.group_pad;
      const int mtiles = (mcols + 15) / 16;
      for (int mt0 = 0; mt0 < mtiles; mt0 += cap) {
        const int mtn = mtiles - mt0 < cap ? mtiles - mt0 : cap;
        if ((mtn == cap) != (pass == 0)) continue;
        if (np == MAF_DW_MAX_LIN) { const int rc = flush(); if (rc) return rc; }
        x.lin[np] = L;
        x.lin[np].mt0 = mt0;
        x.lin[np].mtn = mtn;
        ++np;
      }
    }
  return flush();
}
int maf_launch_reduce(const float* partial, float* out, int n_layer, int nchunks, int T, hipStream_t st) {
  const int64_t total = (int64_t)T * n_layer;
  hipLaunchKernelGGL(maf_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, partial, out, n_layer,
                     nchunks, T);
  return (int)hipGetLastError();
}

static int m_round_up(int v, int m) { return (v + m - 1) / m * m; }
static int m_two_odd(int v) {
  int x = (v + 1) / 2;
  if ((x & 1) == 0) x += 1;
  return 2 * x;
}

// same image conventions as nsf_plan.cpp (set_lin): row-major [rows][ldk], ldk = 2 * odd, rows >= out + 1 zero rows
// (`min_rows`: the transposed K loops of the backward walk 4*KSH rows of the layers they go back through)
static void m_set_lin(LinDesc* L, int* g, int* l, int out, int in, int bias_pad, int ksteps_fixed, int min_rows) {
  L->out = out;
  L->in = in;
  L->ksteps = ksteps_fixed > 0 ? ksteps_fixed : m_round_up((in + 3) / 4, 4);
  L->ldk = m_two_odd(ksteps_fixed > 0 ? in : 4 * L->ksteps);
  L->g_w = *g; *g += out * in;
  L->g_b = *g; *g += out;
  L->l_w = *l;
  int rows = out + 1;
  if (min_rows > rows) rows = min_rows;
  L->rows = rows;
  *l += rows * L->ldk;
  L->l_b = *l;
  *l += bias_pad;
}

static int maf_build_plan(const sbi_amd_maf_config* c, int nw, MafPlan* mp) {
  if (!c) return SBI_AMD_E_BADARG;
  if (c->D < 1 || c->C < 1 || c->H < 1 || c->T < 1 || c->NB < 0) return SBI_AMD_E_BADARG;
  const int K = c->K;
  if (!(K == 4 || K == 5 || K == 8 || K == 10 || K == 16)) return SBI_AMD_E_UNSUPPORTED;
  if (c->D > 16 || c->C > 32 || c->H > 16 * NSF_HT || c->T > NSF_MAX_T || c->NB > MAF_MAX_NB)
    return SBI_AMD_E_UNSUPPORTED;
  if (c->variant != 0 && c->variant != 1) return SBI_AMD_E_UNSUPPORTED;
  if (c->min_bin_width * K > 1.0f || c->min_bin_height * K > 1.0f) return SBI_AMD_E_BADARG;
  memset(mp, 0, sizeof(*mp));
  mp->variant = c->variant;
  NsfPlan* pl = &mp->n;
  const int D = c->D, C = c->C, H = c->H, NB = c->NB;
  pl->D = D; pl->C = C; pl->H = H; pl->K = K; pl->T = c->T; pl->NB = NB;
  pl->P = 3 * K - 1;
  pl->PT = (pl->P + 15) / 16;
  pl->KSH = ((H + 3) / 4 == 13) ? 13 : 16;
  pl->B = c->tail_bound;
  pl->min_w = c->min_bin_width; pl->min_h = c->min_bin_height; pl->min_d = c->min_derivative;
  // nflows' MADE carries no `hidden_features` attribute: the autoregressive transform does not rescale the logits
  pl->sqrt_h = c->scale_by_sqrt_hidden ? (float)sqrt((double)H) : 1.f;
  pl->inv_sqrt_h = c->scale_by_sqrt_hidden ? (float)(1.0 / sqrt((double)H)) : 1.f;
  pl->one_minus_kw = (float)(1.0 - (double)c->min_bin_width * K);
  pl->one_minus_kh = (float)(1.0 - (double)c->min_bin_height * K);
  pl->d_const = (float)log(exp(1.0 - (double)c->min_derivative) - 1.0);
  if (c->variant == 1) {   // zuko's MonotonicRQSTransform: plain softmax bins, no logit rescaling
    pl->min_w = pl->min_h = pl->min_d = 0.f;
    pl->one_minus_kw = pl->one_minus_kh = 1.f;
    pl->sqrt_h = pl->inv_sqrt_h = 1.f;
  }
  pl->log_z = (float)(0.5 * D * log(2.0 * M_PI));
  ShapeDesc* s = &pl->shape[0];
  s->d_id = D; s->d_tr = D; s->in0 = D;
  int g = 0, l = 0;
  const int hb = 16 * NSF_HT, tr_rows = 4 * pl->KSH + 1;
  const int in0 = c->variant == 1 ? D + C : D;   // zuko: the hyper-net's first layer reads [z ; context]
  m_set_lin(&s->lin[0], &g, &l, H, in0, hb, 0, tr_rows);
  if (c->variant == 0) m_set_lin(&s->lin[1], &g, &l, H, C, hb, 0, 0);
  for (int b = 0; b < NB; ++b) m_set_lin(&s->lin[2 + b], &g, &l, H, H, hb, pl->KSH, tr_rows);
  s->fin = 2 + NB;
  l = m_round_up(l, 4);
  s->final_off = l;
  m_set_lin(&s->lin[s->fin], &g, &l, D * pl->P, H, D * 16 * pl->PT, pl->KSH, 0);
  s->n_params = g;
  mp->n_layer = g;
  l = m_round_up(l + 8, 4);   // slack: the transposed reads of the last dim's rows run a few rows past the layer
  mp->l_perm = l; l += 16;
  mp->l_iperm = l; l += 16;
  s->lds_floats = m_round_up(l, 4);
  pl->lds_w_floats = pl->img_floats = s->lds_floats;
  pl->n_params = g * c->T;
  for (int t = 0; t < c->T; ++t) pl->g_layer[t] = t * g;
  // per-wave scratch
  pl->ZW = m_two_odd(D);
  const int ks0 = m_round_up((in0 + 3) / 4, 4), ksc = m_round_up((C + 3) / 4, 4);
  const int need = 4 * ks0 > D + 4 * ksc ? 4 * ks0 : D + 4 * ksc;
  pl->CINW = m_two_odd(need);
  pl->PSW = 16 * pl->PT + 1;
  pl->DS = 16 * pl->PSW;
  while ((pl->DS & 31) != 16) pl->DS += 1;
  pl->DCH = D >= 2 ? 2 : 1;
  mp->PTW = 16 * pl->PT;
  mp->DP = D * mp->PTW;
  int o = 0;
  mp->sc_zs = o; o += 16 * pl->ZW + 16;    // + slack: 16-wide row reads of the helpers
  mp->sc_us = o; o += 16 * pl->ZW + 16;
  mp->sc_gy = o; o += 16 * pl->ZW + 16;
  mp->sc_cin = o; o += 16 * pl->CINW + 16;
  mp->sc_pst = o; o += 2 * pl->DS;
  mp->sc_total = m_round_up(o, 4);
  if (4ll * ((int64_t)pl->lds_w_floats + (int64_t)nw * mp->sc_total) > NSF_LDS_LIMIT_BYTES) return SBI_AMD_E_LDS;
  return 0;
}

// largest workgroup (8, 4, 2, 1 waves) that fits LDS and still yields >= 256 workgroups; `cap` bounds it
static int maf_plan_for_rows(const sbi_amd_maf_config* cfg, int64_t n, int cap, MafPlan* mp, int* nw_out) {
  int nw = cap;
  while (nw > 1 && (n + 16 * nw - 1) / (16 * nw) < 256) nw >>= 1;
  for (; nw >= 1; --nw) {       // any wave count works (16 rows per wave): take the largest that fits LDS
    const int rc = maf_build_plan(cfg, nw, mp);
    if (rc == 0) { *nw_out = nw; return 0; }
    if (rc != SBI_AMD_E_LDS) return rc;
  }
  return SBI_AMD_E_LDS;
}

template int maf_dispatch_k<10>(const MafPlan&, int, int, const float*, const float*, const float*, const float*,
                                int64_t, int64_t, float*, float*, float*, const MafBwdArgs*, hipStream_t);
#define MAF_EXTERN_K(KK) \
  extern template int maf_dispatch_k<KK>(const MafPlan&, int, int, const float*, const float*, const float*, \
                                         const float*, int64_t, int64_t, float*, float*, float*, const MafBwdArgs*, \
                                         hipStream_t);
MAF_EXTERN_K(4) MAF_EXTERN_K(5) MAF_EXTERN_K(8) MAF_EXTERN_K(16)

static int maf_dispatch(const MafPlan& mp, int nw, int mode, const float* packed, const float* zstats, const float* in,
                        const float* x, int64_t n, int64_t x_rows, float* out_main, float* out_aux, float* z_stash,
                        const MafBwdArgs* bwd, hipStream_t st) {
  switch (mp.n.K) {
#define MAF_CASE(KK) \
  case KK: return maf_dispatch_k<KK>(mp, nw, mode, packed, zstats, in, x, n, x_rows, out_main, out_aux, z_stash, bwd, st);
    MAF_CASE(4) MAF_CASE(5) MAF_CASE(8) MAF_CASE(10) MAF_CASE(16)
#undef MAF_CASE
    default: return SBI_AMD_E_UNSUPPORTED;
  }
}

extern "C" int64_t sbi_amd_maf_param_count(const sbi_amd_maf_config* cfg) {
  MafPlan mp;
  const int rc = maf_build_plan(cfg, 1, &mp);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  return mp.n.n_params;
}
extern "C" int64_t sbi_amd_maf_packed_floats(const sbi_amd_maf_config* cfg) {
  MafPlan mp;
  const int rc = maf_build_plan(cfg, 1, &mp);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  return (int64_t)mp.n.T * mp.n.img_floats;
}
extern "C" int64_t sbi_amd_maf_param_offset(const sbi_amd_maf_config* cfg, int32_t t, int32_t which, int32_t bias) {
  MafPlan mp;
  const int rc = maf_build_plan(cfg, 1, &mp);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  if (t < 0 || t >= mp.n.T || which < 0 || which > mp.n.shape[0].fin || (mp.variant == 1 && which == 1))
    return SBI_AMD_E_BADARG;
  const LinDesc& L = mp.n.shape[0].lin[which];
  return (int64_t)t * mp.n_layer + (bias ? L.g_b : L.g_w);
}

extern "C" int sbi_amd_maf_pack(const sbi_amd_maf_config* cfg, const float* params, const int32_t* perms,
                                const float* masks, float* packed, void* stream) {
  if (!cfg || !params || !perms || !packed) return SBI_AMD_E_BADARG;
  if (cfg->variant == 1 && !masks) return SBI_AMD_E_BADARG;   // zuko's adjacency masks come from the host
  MafPlan mp;
  const int rc = maf_build_plan(cfg, 1, &mp);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  hipLaunchKernelGGL(maf_pack_kernel, dim3(mp.n.T, 16), dim3(256), 0, (hipStream_t)stream, mp, params, perms, masks, packed);
  return (int)hipGetLastError();
}

extern "C" int sbi_amd_maf_log_prob(const sbi_amd_maf_config* cfg, const float* packed, const float* zstats,
                                    const float* theta, const float* x, int64_t n, int64_t x_rows, float* logp_out,
                                    float* noise_out, void* stream) {
  if (n == 0) return 0;
  if (!cfg || !packed || !zstats || !theta || !x || !logp_out || n < 0 || x_rows < 1) return SBI_AMD_E_BADARG;
  MafPlan mp;
  int nw = 0;
  const int rc = maf_plan_for_rows(cfg, n, 8, &mp, &nw);
  if (rc) return rc;
  return maf_dispatch(mp, nw, 0, packed, zstats, theta, x, n, x_rows, logp_out, noise_out, nullptr, nullptr,
                      (hipStream_t)stream);
}

extern "C" int sbi_amd_maf_sample(const sbi_amd_maf_config* cfg, const float* packed, const float* zstats,
                                  const float* noise, const float* x, int64_t n, int64_t x_rows, float* theta_out,
                                  float* logabsdet_out, void* stream) {
  if (n == 0) return 0;
  if (!cfg || !packed || !zstats || !noise || !x || !theta_out || n < 0 || x_rows < 1) return SBI_AMD_E_BADARG;
  MafPlan mp;
  int nw = 0;
  const int rc = maf_plan_for_rows(cfg, n, 8, &mp, &nw);
  if (rc) return rc;
  return maf_dispatch(mp, nw, 1, packed, zstats, noise, x, n, x_rows, theta_out, logabsdet_out, nullptr, nullptr,
                      (hipStream_t)stream);
}

// ---- training workspace layout (floats)
struct MafWs {
  int64_t stash, noise, logp, gza, gzb, gp, act, gbuf, ctx, part, total, npad;
  int nchunks, rows_per_chunk;
};
static MafWs maf_ws_layout(const MafPlan& mp, int64_t n) {
  MafWs w;
  const int D = mp.n.D, T = mp.n.T;
  int64_t o = 0;
  auto take = [&](int64_t sz) { const int64_t at = o; o += (sz + 3) / 4 * 4; return at; };
  w.stash = take((int64_t)T * n * D);
  w.noise = take(n * D);
  w.logp = take(n);
  w.gza = take(n * D);
  w.gzb = take(n * D);
  const int64_t npad = (n + MAF_DW_CHUNK - 1) / MAF_DW_CHUNK * MAF_DW_CHUNK;   // the dW kernel reads whole chunks
  w.npad = npad;
  w.gp = take(npad * mp.DP);
  w.act = take(npad * (MAF_MAX_NB + 1) * MAF_AW);
  w.gbuf = take(npad * (MAF_MAX_NB + 2) * MAF_AW);
  w.ctx = take(npad * MAF_CW);
  w.rows_per_chunk = MAF_DW_CHUNK;
  w.nchunks = (int)((n + MAF_DW_CHUNK - 1) / MAF_DW_CHUNK);
  w.part = take((int64_t)T * w.nchunks * mp.n_layer);
  w.total = o;
  return w;
}

extern "C" int64_t sbi_amd_maf_train_workspace_floats(const sbi_amd_maf_config* cfg, int64_t n) {
  MafPlan mp;
  const int rc = maf_build_plan(cfg, 1, &mp);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  return maf_ws_layout(mp, n > 0 ? n : 1).total;
}

__global__ void maf_neg_copy_kernel(const float* __restrict__ in, float* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = -in[i];
}

extern "C" int sbi_amd_maf_loss_fwd_bwd(const sbi_amd_maf_config* cfg, const float* packed, const float* zstats,
                                        const float* masks, const float* theta, const float* x, int64_t n,
                                        int64_t x_rows, const float* row_weight, float uniform_weight,
                                        float* loss_out, float* grad_out, float* grad_theta_out, float* workspace,
                                        void* stream) {
  if (!cfg || !packed || !zstats || !theta || !x || !grad_out || !workspace || n < 1 || x_rows < 1)
    return SBI_AMD_E_BADARG;
  if (cfg->variant == 1 && !masks) return SBI_AMD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  MafPlan mp;
  int nw = 0;
  int rc = maf_plan_for_rows(cfg, n, 8, &mp, &nw);
  if (rc) return rc;
  const MafWs w = maf_ws_layout(mp, n);
  const int D = mp.n.D, T = mp.n.T, NB = mp.n.NB;
  rc = maf_dispatch(mp, nw, 0, packed, zstats, theta, x, n, x_rows, workspace + w.logp, workspace + w.noise,
                    workspace + w.stash, nullptr, st);
  if (rc) return rc;
  if (loss_out)
    hipLaunchKernelGGL(maf_neg_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace + w.logp,
                       loss_out, (long long)n);
  // backward kernels: up to two waves per SIMD (<= 256 VGPRs: the activation set of the whole conditioner lives in
  // registers), as many as the LDS left over by the weight image allows
  MafPlan mpb;
  int nwb = 0;
  rc = maf_plan_for_rows(cfg, n, 8, &mpb, &nwb);
  if (rc) return rc;
  float* gz[2] = {workspace + w.gza, workspace + w.gzb};
  for (int t = T - 1; t >= 0; --t) {
    MafBwdArgs a;
    a.packed = packed; a.zstats = zstats;
    a.z_in = workspace + w.stash + (int64_t)t * n * D;
    a.x = x;
    a.gz_up = (t == T - 1) ? workspace + w.noise : gz[(t + 1) & 1];
    a.row_w = row_weight; a.uni_w = uniform_weight;
    a.n = n; a.x_rows = x_rows;
    a.gz_dn = gz[t & 1];
    a.grad_theta = grad_theta_out;
    a.GP = workspace + w.gp; a.ACT = workspace + w.act; a.G = workspace + w.gbuf; a.CTX = workspace + w.ctx;
    a.npad = w.npad;
    a.t = t; a.is_last = (t == T - 1);
    rc = maf_dispatch(mpb, nwb, 2, packed, zstats, nullptr, x, n, x_rows, nullptr, nullptr, nullptr, &a, st);
    if (rc) return rc;
    MafDwArgs d;
    memset(&d, 0, sizeof(d));
    const ShapeDesc& S = mp.n.shape[0];
    const int AWS = (MAF_MAX_NB + 1) * MAF_AW;
    const int64_t gts = w.npad * 16;     // floats per m-tile plane
    auto set = [&](int i, const float* G, const float* A, int lda, const LinDesc& L, int group, int gpad, int kind,
                   int gperm, int aperm) {
      d.lin[i].G = G; d.lin[i].gts = gts; d.lin[i].A = A; d.lin[i].lda = lda;
      d.lin[i].out = L.out; d.lin[i].in = L.in; d.lin[i].in_total = L.in; d.lin[i].col0 = 0;
      d.lin[i].group = group; d.lin[i].group_pad = gpad;
      d.lin[i].g_w = L.g_w; d.lin[i].g_b = L.g_b; d.lin[i].kind = kind;
      d.lin[i].gperm = gperm; d.lin[i].aperm = aperm;
    };
    // largest first (the final layer has D*(3K-1) outputs)
    // (G planes of the hidden layers and the ACT rows are written in fragment order by the backward kernel; the
    // spline-parameter planes GP and the CTX rows in natural order)
    set(0, a.GP, a.ACT + 64 * NB, AWS, S.lin[S.fin], mp.n.P, mp.PTW, 3, 0, 1);
    for (int b = 0; b < NB; ++b)
      set(1 + b, a.G + 4 * (2 + b) * gts, a.ACT + 64 * b, AWS, S.lin[2 + b], mp.n.H, 64, 2, 1, 1);
    set(1 + NB, a.G, a.CTX, MAF_CW, S.lin[0], mp.n.H, 64, 0, 1, 0);     // inputs: CTX rows = [z ; context]
    int nlin = 2 + NB;
    if (mp.variant == 0) { set(2 + NB, a.G + 4 * gts, a.CTX + D, MAF_CW, S.lin[1], mp.n.H, 64, 1, 1, 0); nlin = 3 + NB; }
    d.mask = masks ? masks + (int64_t)t * mp.n_layer : nullptr;
    d.n = n; d.rows_per_chunk = w.rows_per_chunk; d.nchunks = w.nchunks; d.n_layer = mp.n_layer;
    d.D = D; d.P = mp.n.P;
    d.partial = workspace + w.part + (int64_t)t * w.nchunks * mp.n_layer;
    rc = maf_launch_dw(d, nlin, st);
    if (rc) return rc;
  }
  return maf_launch_reduce(workspace + w.part, grad_out, mp.n_layer, w.nchunks, T, st);
}
