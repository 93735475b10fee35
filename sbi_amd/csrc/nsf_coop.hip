// nsf_coop.hip -- host side of the cooperative (small-batch) NSF kernels (nsf_coop.h): image packing, the forward /
// training-forward / training-backward entry points the C ABI functions of nsf_flow.hip / nsf_train.hip route small
// batches to, and the num_bins = 10 instantiations.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#define NSF_COOP_MAIN_TU
#include "nsf_train_kernel.h"
#include "nsf_coop_kernel.h"
#include "debug_env.h"

template int co_fwd_k<10>(const NsfPlan&, const CoopPlan&, const CoFwdArgs&, hipStream_t);
template int co_bwd_k<10>(const NsfPlan&, const CoopPlan&, const CoBwdArgs&, hipStream_t);
#define CO_EXTERN(KK)                                                                                       \
  extern template int co_fwd_k<KK>(const NsfPlan&, const CoopPlan&, const CoFwdArgs&, hipStream_t);        \
  extern template int co_bwd_k<KK>(const NsfPlan&, const CoopPlan&, const CoBwdArgs&, hipStream_t);
CO_EXTERN(4) CO_EXTERN(5) CO_EXTERN(8) CO_EXTERN(16)

// Does an n-row call take the cooperative kernels?  `training`: the stash-writing forward + backward pair.
// SBI_AMD_ABLATE bit 16384 switches the path off (A/B measurements, tests of the throughput kernels at small n).
bool coop_applies(const sbi_amd_nsf_config* cfg, int64_t n, bool training, NsfPlan* pl, CoopPlan* cp) {
  if (n < 1 || n > coop_max_rows() || (sbi_amd_dbg_ablate() & 16384)) return false;
  int rc = nsf_build_plan(cfg, 1, pl);
  if (rc && rc != SBI_AMD_E_LDS) return false;   // (E_LDS speaks about the throughput kernels' weight image)
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

int coop_pack(const sbi_amd_nsf_config* cfg, const float* params, float* cimg, void* stream) {
  NsfPlan pl;
  CoopPlan cp;
  if (!coop_shape_ok(cfg, &pl, &cp)) return 0;
  hipLaunchKernelGGL(nsf_coop_pack_kernel, dim3(pl.T, cp.img_floats >> 8), dim3(256), 0, (hipStream_t)stream, pl, cp,
                     params, cimg);
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
  o += 1024;   // debug timelines (SBI_AMD_TIMELINE): the last 512 int64 of the workspace
  return o;
}
int64_t coop_workspace_floats(const NsfPlan& pl, const CoopPlan& cp, int64_t n) {
  int64_t a, b, c, d, e;
  return co_ws_layout(pl, cp, n, &a, &b, &c, &d, &e);
}

int coop_log_prob(const sbi_amd_nsf_config* cfg, const NsfPlan& pl, const CoopPlan& cp, const float* cimg,
                  const float* zstats, const float* theta, const float* x, int64_t n, int64_t x_rows, float* logp,
                  float* noise, void* stream) {
  CoFwdArgs a = {cimg, zstats, theta, x, (long long)n, (long long)x_rows, logp, noise, nullptr, nullptr, nullptr};
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
                     (const float*)(workspace + o_logp), loss_out, (long long)n);
  return (int)hipGetLastError();
}
