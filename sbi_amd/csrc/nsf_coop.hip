// nsf_coop.hip -- host side of the cooperative (small-batch) NSF kernels (nsf_coop.h): image packing, the forward /
// training-forward / training-backward entry points the C ABI functions of nsf_flow.hip / nsf_train.hip route small
// batches to, and the num_bins = 10 instantiations.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#define NSF_COOP_MAIN_TU
#include "nsf_train_kernel.h"
#include "nsf_coop_wide_kernel.h"
#include "debug_env.h"

template int co_fwd_k<10>(const NsfPlan&, const CoopPlan&, const CoFwdArgs&, hipStream_t);
template int co_bwd_k<10>(const NsfPlan&, const CoopPlan&, const CoBwdArgs&, hipStream_t);
template int co_inv_k<10>(const NsfPlan&, const CoopPlan&, const float*, const float*, const float*, const float*,
                          long long, long long, float*, float*, hipStream_t);
#define CO_EXTERN(KK)                                                                                       \
  extern template int co_fwd_k<KK>(const NsfPlan&, const CoopPlan&, const CoFwdArgs&, hipStream_t);        \
  extern template int co_bwd_k<KK>(const NsfPlan&, const CoopPlan&, const CoBwdArgs&, hipStream_t);        \
  extern template int co_inv_k<KK>(const NsfPlan&, const CoopPlan&, const float*, const float*, const float*,  \
                                   const float*, long long, long long, float*, float*, hipStream_t);
CO_EXTERN(4) CO_EXTERN(5) CO_EXTERN(8) CO_EXTERN(16)

// Does an n-row call take the cooperative kernels?  `training`: the stash-writing forward + backward pair.
// SBI_AMD_ABLATE bit 16384 switches the path off (A/B measurements, tests of the throughput kernels at small n).
bool coop_applies(const sbi_amd_nsf_config* cfg, int64_t n, bool training, NsfPlan* pl, CoopPlan* cp) {
  if (n < 1) return false;
  int rc = nsf_build_plan(cfg, 1, pl);
  if (rc && rc != SBI_AMD_E_LDS) return false;   // (E_LDS speaks about the throughput kernels' weight image)
  // hidden > 64: the wide cooperative kernels are the ONLY path, at every batch size and whatever the switches say
  const bool wide = pl->H > 16 * NSF_HT;
  if (!wide && (n > (training ? coop_train_rows() : coop_max_rows()) || (sbi_amd_dbg_ablate() & 16384))) return false;
  return coop_build_plan(*pl, n, 0, training, cp) == 0;
}
// Is there a cooperative image for this configuration at all (any n)?  Deliberately independent of the row
// threshold / ablation switches: the size of the packed buffer must not change when those are toggled at run time.
bool coop_shape_ok(const sbi_amd_nsf_config* cfg, NsfPlan* pl, CoopPlan* cp) {
  int rc = nsf_build_plan(cfg, 1, pl);
  if (rc && rc != SBI_AMD_E_LDS) return false;
  return coop_build_plan(*pl, 1, 0, true, cp) == 0;
}

int64_t coop_packed_floats(const sbi_amd_nsf_config* cfg) {
  NsfPlan pl;
  CoopPlan cp;
  return coop_shape_ok(cfg, &pl, &cp) ? coop_image_floats(pl, cp) : 0;
}

int coop_pack(const sbi_amd_nsf_config* cfg, const float* params, float* cimg, int which, void* stream) {
  NsfPlan pl;
  CoopPlan cp;
  if (!coop_shape_ok(cfg, &pl, &cp)) return 0;
  if (which & 1)      // the image log_prob and the training pass read
    hipLaunchKernelGGL(nsf_coop_pack_kernel, dim3(pl.T, cp.img_floats >> 8), dim3(256), 0, (hipStream_t)stream, pl, cp,
                       params, cimg, 0);
  if ((which & 2) && cp.sh[0].UI.mtiles > 0)      // + the explicit LU inverses (sampling direction, wide nets)
    hipLaunchKernelGGL(nsf_coop_pack_kernel, dim3(pl.T, cp.img_floats >> 8), dim3(256), 0, (hipStream_t)stream, pl, cp,
                       params, cimg, 1);
  return (int)hipGetLastError();
}

static int co_dispatch_fwd(const sbi_amd_nsf_config* cfg, const NsfPlan& pl, const CoopPlan& cp, const CoFwdArgs& a,
                           hipStream_t st) {
  switch (cfg->K) {
    case 4: return co_fwd_k<4>(pl, cp, a, st);
    case 5: return co_fwd_k<5>(pl, cp, a, st);
    case 8: return co_fwd_k<8>(pl, cp, a, st);
    case 10: return co_fwd_k<10>(pl, cp, a, st);
    case 16: return co_fwd_k<16>(pl, cp, a, st);
  }
  return SBI_AMD_E_UNSUPPORTED;
}
int coop_sample(const sbi_amd_nsf_config* cfg, const NsfPlan& pl, const CoopPlan& cp, const float* cimg,
                const float* zstats, const float* noise, const float* x, int64_t n, int64_t x_rows, float* theta_out,
                float* logabsdet_out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (cfg->K) {
    case 4: return co_inv_k<4>(pl, cp, cimg, zstats, noise, x, n, x_rows, theta_out, logabsdet_out, st);
    case 5: return co_inv_k<5>(pl, cp, cimg, zstats, noise, x, n, x_rows, theta_out, logabsdet_out, st);
    case 8: return co_inv_k<8>(pl, cp, cimg, zstats, noise, x, n, x_rows, theta_out, logabsdet_out, st);
    case 10: return co_inv_k<10>(pl, cp, cimg, zstats, noise, x, n, x_rows, theta_out, logabsdet_out, st);
    case 16: return co_inv_k<16>(pl, cp, cimg, zstats, noise, x, n, x_rows, theta_out, logabsdet_out, st);
  }
  return SBI_AMD_E_UNSUPPORTED;
}

static int co_dispatch_bwd(const sbi_amd_nsf_config* cfg, const NsfPlan& pl, const CoopPlan& cp, const CoBwdArgs& a,
                           hipStream_t st) {
  switch (cfg->K) {
    case 4: return co_bwd_k<4>(pl, cp, a, st);
    case 5: return co_bwd_k<5>(pl, cp, a, st);
    case 8: return co_bwd_k<8>(pl, cp, a, st);
    case 10: return co_bwd_k<10>(pl, cp, a, st);
    case 16: return co_bwd_k<16>(pl, cp, a, st);
  }
  return SBI_AMD_E_UNSUPPORTED;
}

// workgroups of nsf_coop_reduce_kernel = partial sums of squares it leaves behind the stash
static inline int64_t co_sq_parts(const NsfPlan& pl, const CoopPlan& cp) { return (int64_t)((cp.PLP / 4 + 63) / 64) * pl.T; }
// training workspace of the cooperative path: per-transform input state, z_T, log p, partial slabs, stash
static int64_t co_ws_layout(const NsfPlan& pl, const CoopPlan& cp, int64_t n, int64_t* o_zst, int64_t* o_noise,
                            int64_t* o_logp, int64_t* o_part, int64_t* o_ast) {
  int64_t o = 0;
  *o_zst = o; o += (int64_t)pl.T * n * pl.D;
  *o_noise = o; o += n * pl.D;
  *o_logp = o; o += (n + 3) / 4 * 4;
  o = (o + 3) / 4 * 4;
  *o_part = o; o += (int64_t)pl.T * cp.grid * cp.PLP;
  o = (o + 63) / 64 * 64;
  *o_ast = o; o += (int64_t)pl.T * ((n + 15) / 16) * cp.slots * 256;
  o += (co_sq_parts(pl, cp) + 3) / 4 * 4;   // partial sums of squares of the reduced gradient (the clip's norm)
  o += 1024;   // debug timelines (SBI_AMD_TIMELINE): the last 512 int64 of the workspace
  return o;
}
int64_t coop_workspace_floats(const NsfPlan& pl, const CoopPlan& cp, int64_t n) {
  int64_t a, b, c, d, e;
  return co_ws_layout(pl, cp, n, &a, &b, &c, &d, &e);
}
const float* coop_sqnorm_parts(const NsfPlan& pl, const CoopPlan& cp, int64_t n, const float* workspace, int64_t* n_parts) {
  if (n_parts) *n_parts = co_sq_parts(pl, cp);
  return workspace + coop_workspace_floats(pl, cp, n) - 1024 - (co_sq_parts(pl, cp) + 3) / 4 * 4;
}

int coop_log_prob(const sbi_amd_nsf_config* cfg, const NsfPlan& pl, const CoopPlan& cp, const float* cimg,
                  const float* zstats, const float* theta, const float* x, int64_t n, int64_t x_rows, float* logp,
                  float* noise, void* stream) {
  CoFwdArgs a = {cimg, zstats, theta, x, (long long)n, (long long)x_rows, logp, noise, nullptr, nullptr, nullptr};
  return co_dispatch_fwd(cfg, pl, cp, a, (hipStream_t)stream);
}

// `nticks` ticks of the vectorised slice sampler for all chains in ONE launch (nsf_coop_kernel.h, MC = true).
// theta_next must hold the constrained image of next_param on entry (sbi_amd_mcmc_to_constrained) and does on exit.
extern "C" int sbi_amd_mcmc_slice_run(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats,
                                      const float* x_o, int32_t num_chains, int32_t num_samples, int32_t tuning,
                                      float max_width, float* x, float* next_param, float* width, int32_t* order,
                                      int32_t* istate, float* fstate, float* samples, int32_t* done_count, uint64_t seed,
                                      uint64_t tick0, int32_t nticks, int32_t kind, const float* p0, const float* p1,
                                      float* theta_next, float* logabsdet_next, float* logp_scratch, void* stream) {
  if (!cfg || !packed || !zstats || !x_o || num_chains < 1 || num_samples < 0 || tuning < 0 || nticks < 0 || !x ||
      !next_param || !width || !order || !istate || !fstate || !samples || !done_count || !theta_next ||
      !logabsdet_next || !logp_scratch || kind < 0 || kind > 2 || (kind && (!p0 || !p1)))
    return SBI_AMD_E_BADARG;
  if (nticks == 0) return 0;
  NsfPlan pl;
  CoopPlan cp;
  int rc = nsf_build_plan(cfg, 1, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  if (pl.H > 16 * NSF_HT) return SBI_AMD_E_UNSUPPORTED;            // (the wide kernels have no sampler mode)
  rc = coop_build_plan(pl, num_chains, 1, false, &cp);
  if (rc) return rc;
  McArgs mc{};
  mc.num_samples = num_samples; mc.tuning = tuning; mc.nticks = nticks; mc.kind = kind; mc.max_width = max_width;
  mc.seed = seed; mc.tick0 = tick0; mc.p0 = p0; mc.p1 = p1;
  mc.x = x; mc.next_param = next_param; mc.width = width; mc.fstate = fstate; mc.samples = samples;
  mc.theta_next = theta_next; mc.lad_next = logabsdet_next; mc.logp_buf = logp_scratch;
  mc.order = order; mc.istate = istate; mc.done_count = done_count;
  CoFwdArgs a = {packed + nsf_packed_floats(pl), zstats, theta_next, x_o, (long long)num_chains, 1, logp_scratch, nullptr,
                 nullptr, nullptr, nullptr, &mc};
  return co_dispatch_fwd(cfg, pl, cp, a, (hipStream_t)stream);
}

int coop_train_forward(const sbi_amd_nsf_config* cfg, const NsfPlan& pl, const CoopPlan& cp, const float* cimg,
                       const float* zstats, const float* theta, const float* x, int64_t n, int64_t x_rows,
                       float* logp_out, float* workspace, void* stream) {
  int64_t o_zst, o_noise, o_logp, o_part, o_ast;
  co_ws_layout(pl, cp, n, &o_zst, &o_noise, &o_logp, &o_part, &o_ast);
  // debug timeline (SBI_AMD_TIMELINE): the stamps land in the (not yet used) partial-slab region and are printed here
  long long* dbg = sbi_amd_dbg_timeline() ? (long long*)(workspace + o_part) : nullptr;
  if (dbg) hipMemsetAsync(dbg, 0, 256 * sizeof(long long), (hipStream_t)stream);
  CoFwdArgs a = {cimg, zstats, theta, x, (long long)n, (long long)x_rows, workspace + o_logp, workspace + o_noise,
                 workspace + o_zst, workspace + o_ast, dbg};
  int rc = co_dispatch_fwd(cfg, pl, cp, a, (hipStream_t)stream);
  if (rc) return rc;
  if (dbg) {
    static int printed = 0;
    long long h[256];
    hipStreamSynchronize((hipStream_t)stream);
    hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    if (printed++ == 20) {      // a warm call
      fprintf(stderr, "coop fwd timeline (cycles since stamp 0 of wave 0), n = %lld, NT = %d\n", (long long)n, cp.NT);
      for (int i = 0; i < 64; ++i) {
        if (!h[i] && !h[64 + i]) continue;
        fprintf(stderr, "  stamp %2d:", i);
        for (int w = 0; w < 4; ++w) fprintf(stderr, " %8lld", h[64 * w + i] ? h[64 * w + i] - h[0] : -1);
        fprintf(stderr, "\n");
      }
    }
  }
  if (logp_out) {
    hipError_t e = hipMemcpyAsync(logp_out, workspace + o_logp, sizeof(float) * n, hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

// backward launch + the fixed-order reduction of its partial slabs (grid = cp.grid slabs of cp.PLP floats per transform)
int coop_train_backward(const sbi_amd_nsf_config* cfg, const NsfPlan& pl, const CoopPlan& cp, const float* params,
                        const float* cimg, const float* zstats, const float* x, int64_t n, int64_t x_rows,
                        const float* row_weight, float uniform_weight, float* grad_out, float* grad_theta_out,
                        float* grad_x_out, float* loss_out, float* workspace, void* stream) {
  int64_t o_zst, o_noise, o_logp, o_part, o_ast;
  co_ws_layout(pl, cp, n, &o_zst, &o_noise, &o_logp, &o_part, &o_ast);
  const int64_t ws_total = coop_workspace_floats(pl, cp, n);
  long long* dbg = sbi_amd_dbg_timeline() ? (long long*)(workspace + ws_total - 1024) : nullptr;
  if (dbg) hipMemsetAsync(dbg, 0, 256 * sizeof(long long), (hipStream_t)stream);
  CoBwdArgs a = {cimg, zstats, x, (long long)n, (long long)x_rows, row_weight, uniform_weight, workspace + o_noise,
                 workspace + o_zst, workspace + o_ast, workspace + o_part, grad_theta_out, grad_x_out, dbg};
  int rc = co_dispatch_bwd(cfg, pl, cp, a, (hipStream_t)stream);
  if (rc) return rc;
  if (dbg) {
    static int printed = 0;
    long long h[256];
    hipStreamSynchronize((hipStream_t)stream);
    hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    if (printed++ == 20) {
      fprintf(stderr, "coop bwd timeline (cycles since stamp 0 of wave 0), n = %lld, NT = %d\n", (long long)n, cp.NT);
      for (int i = 0; i < 64; ++i) {
        if (!h[i] && !h[64 + i]) continue;
        fprintf(stderr, "  stamp %2d:", i);
        for (int w = 0; w < 4; ++w) fprintf(stderr, " %8lld", h[64 * w + i] ? h[64 * w + i] - h[0] : -1);
        fprintf(stderr, "\n");
      }
    }
  }
  hipLaunchKernelGGL(nsf_coop_reduce_kernel, dim3((cp.PLP / 4 + 63) / 64, pl.T), dim3(64 * CO_RED_GROUPS), 0,
                     (hipStream_t)stream, pl, cp, params, (const float*)(workspace + o_part), grad_out,
                     (const float*)(workspace + o_logp), loss_out, (long long)n,
                     workspace + ws_total - 1024 - (co_sq_parts(pl, cp) + 3) / 4 * 4);
  return (int)hipGetLastError();
}

// Host-side self-check (CPU tests, no device work): the arithmetic the kernels use in place of table look-ups must
// reproduce the plan's tables -- slab tile bases (nsf_coop_bwd_kernel: tb_blk0 / blk_tiles / tb_wf) against
// CoShape::dw_tb, the per-block image strides of CoK against every CoMat offset -- and, for the benchmark
// configuration, the compile-time layouts (kStaticPl, kStaticTp) must equal the run-time plans bit for bit.
// Returns 0 or the number of the first failing check.
extern "C" int sbi_amd_nsf_coop_selfcheck(const sbi_amd_nsf_config* cfg) {
  NsfPlan pl;
  CoopPlan cp;
  if (!coop_shape_ok(cfg, &pl, &cp)) return -1;
  CoK k;
  coop_make_consts(pl, cp, &k);
  for (int par = 0; par < 2; ++par) {
    const CoShape& c = cp.sh[par];
    const ShapeDesc& S = pl.shape[par];
    const CoKP& q = k.p[par];
    const int HT = k.HT, HQ = k.HQ;         // 4 / 4 (hidden <= 64) or 8 / 8 (the wide kernels' COW_HT / COW_KQ)
    if (HT != 4 * cp.MT || HQ != 4 * cp.MT || (cp.MT == 2) != (pl.H > 16 * NSF_HT)) return 13;
    const int blk_tiles = HT * k.ntc + 2 * HT * k.nnh, tb_blk0 = HT * q.nnt0;
    if (c.dw_tb[0] != 0 || c.dw_nnt[0] != q.nnt0) return 1;
    for (int b = 0; b < pl.NB; ++b) {
      if (c.dw_tb[1 + 3 * b] != tb_blk0 + b * blk_tiles || c.dw_nnt[1 + 3 * b] != k.ntc) return 2;
      if (c.dw_tb[2 + 3 * b] != tb_blk0 + b * blk_tiles + HT * k.ntc || c.dw_nnt[2 + 3 * b] != k.nnh) return 3;
      if (c.dw_tb[3 + 3 * b] != tb_blk0 + b * blk_tiles + HT * k.ntc + HT * k.nnh || c.dw_nnt[3 + 3 * b] != k.nnh) return 4;
      if (c.WC[b].off != q.wc0 + b * k.sA || c.W1[b].off != q.w10 + b * k.sA || c.W2[b].off != q.w20 + b * k.sA) return 5;
      if (c.W1T[b].off != q.w1t0 + b * k.sT || c.W2T[b].off != q.w2t0 + b * k.sT || c.WCT[b].off != q.wct0 + b * k.sC) return 6;
      if (c.bc[b].off != q.bc0 + b * k.sB || c.b1[b].off != q.b10 + b * k.sB || c.b2[b].off != q.b20 + b * k.sB) return 7;
      if (c.W1[b].quads != HQ || c.W2[b].quads != HQ || c.W1T[b].quads != HQ || c.W2T[b].quads != HQ || c.WC[b].quads != k.KCQ) return 8;
      if (c.W1[b].mtiles != HT || c.W2T[b].mtiles != HT || c.WC[b].mtiles != HT) return 14;
    }
    if (c.dw_tb[S.fin] != tb_blk0 + pl.NB * blk_tiles || c.dw_nnt[S.fin] != k.nnh) return 9;
    if (c.dw_tail != (c.dw_tb[S.fin] + c.nft * k.nnh) * 256 || c.dw_tail + pl.D * (pl.D - 1) + 2 * pl.D + 1 > cp.PLP) return 10;
    if (c.W0.quads != k.KCQ + 1 || c.WF.quads != HQ || c.WFT.quads != S.d_tr * pl.PT || c.W0T.quads != HQ) return 11;
    if (c.o_bias % 256 != 0 || c.o_ld >= cp.img_floats) return 12;
    // the wide kernels' stash slots (h0 | per block t1 t2 sigmoid(gate) h | parameter tiles) and their packed LU inverses
    if (cp.s_blk != HT || cp.s_par != HT + 4 * HT * pl.NB || cp.slots != cp.s_par + pl.shape[0].d_tr * pl.PT) return 15;
    if (c.UI.mtiles != 1 || c.LI.mtiles != 1 || q.ui != c.UI.off || q.li != c.LI.off) return 16;
    if (k.KCQ > (cp.MT == 2 ? 4 : 2) || (cp.MT == 1 && c.nft > 16)) return 17;
  }
  if (cfg->D == 10 && cfg->C == 10 && cfg->H == 50 && cfg->K == 10 && cfg->T == 5 && cfg->NB == 2) {
    NsfPlan p4;
    TrainPlan tp;
    if (nsf_build_plan(cfg, TR_NW, &p4) != 0 || build_train_plan(p4, 65536, &tp) != 0) return 20;
    if (!plan_is_static_default(p4, tp) && !NSF_DBG_ABL(p4.ablate, 0x40000)) return 21;
  }
  return 0;
}
