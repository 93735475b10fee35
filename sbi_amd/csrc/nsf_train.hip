// nsf_train.hip -- training pass C ABI, gradient reduction, num_bins = 10 instantiations of the
// backward kernel (other bin counts: nsf_train_k5.hip / nsf_train_k8.hip, separate TUs for a parallel build).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "nsf_train_kernel.h"
#include "debug_env.h"
#include "nsf_coop_host.h"

// grad[p] = sum over workgroups of the partial slabs (fixed association => deterministic);
// finishes LULinear's diagonal: d/d(unconstrained_upper_diag_i) =
//   (dL/dU_ii + (sum_n dL/dlogabsdet_n) / U_ii) * sigmoid(unconstrained_i)
// block = 64 params x 8 slab groups: enough loads in flight to stream the ~100 MB of partials.
#ifndef RED_GROUPS
#define RED_GROUPS 8
#endif
// grid (ceil(PLP / 256), T): one thread sums FOUR consecutive slab words (16-byte loads, 4 x 8-way in flight) over the
// workgroups' slabs; a slab is indexed relative to its transform's parameter block, so the words are 16-byte aligned
__global__ void __launch_bounds__(64 * RED_GROUPS)
nsf_grad_reduce_kernel(const NsfPlan pl, const TrainPlan tp, const float* __restrict__ params,
                       const float* __restrict__ partial, float* __restrict__ grad,
                       const float* __restrict__ logp, float* __restrict__ loss_out, long long n_rows,
                       float* __restrict__ sq_out) {
  // rider (saves a launch per step): the per-row loss of the one-call form, loss = -log p
  if (loss_out)
    for (long long i = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < n_rows;
         i += (long long)gridDim.x * gridDim.y * blockDim.x)
      loss_out[i] = -logp[i];
  typedef float f4r __attribute__((ext_vector_type(4)));
  __shared__ f4r red[RED_GROUPS][64];
  __shared__ float red_sgl[RED_GROUPS];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int t = blockIdx.y;
  const ShapeDesc& S = pl.shape[pl.ctx_mlp ? 0 : (t & 1)];
  const int li = 4 * (blockIdx.x * 64 + lane);
  const bool live = li < S.n_params;
  const float* base = partial + (long long)t * tp.grid * tp.PLP;
  const int ntri = pl.D * (pl.D - 1) / 2;
  const int d0 = S.g_lu + 2 * ntri;
  f4r a = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    f4r acc4[4] = {a, a, a, a};
    int w = grp;
    for (; w + 3 * RED_GROUPS < tp.grid; w += 4 * RED_GROUPS) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc4[u] += *reinterpret_cast<const f4r*>(base + (long long)(w + u * RED_GROUPS) * tp.PLP + li);
    }
    for (; w < tp.grid; w += RED_GROUPS) acc4[0] += *reinterpret_cast<const f4r*>(base + (long long)w * tp.PLP + li);
    a = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
  }
  red[grp][lane] = a;
  {   // sum_n d loss / d logabsdet_n: one word per slab (slot n_params)
    float sgl = 0.f;
    if (!pl.ctx_mlp)
      for (int w2 = grp * 64 + lane; w2 < tp.grid; w2 += 64 * RED_GROUPS) sgl += base[(long long)w2 * tp.PLP + S.n_params];
    for (int off = 32; off > 0; off >>= 1) sgl += __shfl_xor(sgl, off);
    if (lane == 0) red_sgl[grp] = sgl;
  }
  __syncthreads();
  if (grp == 0) {
    float sq = 0.f;     // second rider: this workgroup's share of |grad|^2 (the clip's norm: no separate pass over grad)
    if (live) {
      f4r tot = {0.f, 0.f, 0.f, 0.f};
      float tsg = 0.f;
#pragma unroll
      for (int g = 0; g < RED_GROUPS; ++g) { tot += red[g][lane]; tsg += red_sgl[g]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (li + r >= S.n_params) continue;
        const int idx = pl.g_layer[t] + li + r;
        float v = tot[r];
        if (!pl.ctx_mlp && li + r >= d0 && li + r < d0 + pl.D) {
          const float ud = params[idx];
          const float uii = softplus_f(ud) + pl.lu_eps;
          v = (v + tsg / uii) * (1.f / (1.f + expf(-ud)));
        }
        grad[idx] = v;
        sq += v * v;
      }
    }
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);     // fixed butterfly: deterministic
    if (lane == 0 && sq_out) sq_out[blockIdx.y * gridDim.x + blockIdx.x] = sq;
  }
}

__global__ void neg_copy_kernel(const float* __restrict__ in, float* __restrict__ out, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = -in[i];
}

// ------------------------------------------------------------------ host side
// generic training pass (nsf_gtrain.hip) for the shapes the wave-specialised backward kernel refuses
int64_t nsf_g_workspace_floats(const sbi_amd_nsf_config* cfg, int64_t n);
int nsf_g_train_forward(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats, const float* theta,
                        const float* x, int64_t n, int64_t x_rows, float* logp_out, float* workspace, void* stream);
int nsf_g_train_backward(const sbi_amd_nsf_config* cfg, const float* params, const float* packed, const float* zstats,
                         const float* x, int64_t n, int64_t x_rows, const float* row_weight, float uniform_weight,
                         float* grad_out, float* grad_theta_out, float* workspace, void* stream);
const float* nsf_g_logp(const sbi_amd_nsf_config* cfg, int64_t n, const float* workspace);
// SBI_AMD_ABLATE bit 2048 (diagnostic): route every shape through the generic pass
static bool fast_path_refuses(int rc) {
  return rc == SBI_AMD_E_UNSUPPORTED || rc == SBI_AMD_E_LDS || (rc == 0 && (sbi_amd_dbg_ablate() & 2048));
}

int nsf_log_prob_stash(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats, const float* theta,
                       const float* x, int64_t n, int64_t x_rows, float* logp_out, float* noise_out,
                       float* z_stash, float* astash, float* pstash, void* stream, bool fp32_bin);

// workgroups of nsf_grad_reduce_kernel = partial sums of squares it leaves behind the activation stash
static inline int64_t thr_sq_parts(const NsfPlan& pl, const TrainPlan& tp) { return (int64_t)((tp.PLP / 4 + 63) / 64) * pl.T; }
static int64_t ws_layout(const NsfPlan& pl, const TrainPlan& tp, int64_t n, int64_t* o_stash, int64_t* o_noise,
                         int64_t* o_logp, int64_t* o_gza, int64_t* o_gzb, int64_t* o_part, int64_t* o_ast,
                         int64_t* o_pst = nullptr) {
  int64_t o = 0;
  *o_stash = o; o += (int64_t)pl.T * n * pl.D;
  *o_noise = o; o += n * pl.D;
  *o_logp = o; o += (n + 3) / 4 * 4;
  *o_gza = o; o += n * pl.D;
  *o_gzb = o; o += n * pl.D;
  o = (o + 3) / 4 * 4;
  *o_part = o; o += (int64_t)pl.T * tp.grid * tp.PLP;
  o = (o + 3) / 4 * 4;
  *o_ast = o;   // activation stash: T x ceil(n/16) wave-tiles x slots x 1024 floats
  o += (int64_t)pl.T * ((n + 15) / 16) * nsf_ast_slots(pl) * 1024;
  if (o_pst) *o_pst = o;   // spline-parameter stash: T x ceil(n/16) wave-tiles x d_tr x PT x 256 floats
  if (TR_PSTASH) o += (int64_t)pl.T * ((n + 15) / 16) * nsf_pst_tile_floats(pl);
  o += (thr_sq_parts(pl, tp) + 3) / 4 * 4;   // partial sums of squares of the reduced gradient (the clip's norm)
  o += 2048;   // debug timeline (SBI_AMD_TIMELINE): last 1024 int64 of the workspace
  return o;
}

extern "C" int64_t sbi_amd_nsf_train_workspace_floats(const sbi_amd_nsf_config* cfg, int64_t n) {
  {
    NsfPlan cpl;
    CoopPlan cp;
    if (coop_applies(cfg, n > 0 ? n : 1, true, &cpl, &cp)) return coop_workspace_floats(cpl, cp, n > 0 ? n : 1);
  }
  NsfPlan pl;
  // (an E_LDS here speaks about the FORWARD kernel's 4-wave layout; the backward kernel has its own budget, checked
  // by build_train_plan, and the training forward picks its workgroup size in nsf_plan_for_rows)
  int rc = nsf_build_plan(cfg, TR_NW, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  TrainPlan tp;
  rc = build_train_plan(pl, n > 0 ? n : 1, &tp);
  if (fast_path_refuses(rc)) {
    const int64_t g = nsf_g_workspace_floats(cfg, n);
    return (g >= 0 || g == SBI_AMD_E_LDS) ? g : rc;
  }
  if (rc) return rc;
  int64_t a, b, c, d, e, f, g;
  return ws_layout(pl, tp, n > 0 ? n : 1, &a, &b, &c, &d, &e, &f, &g);
}

template int launch_bwd_k<10>(const NsfPlan&, const TrainPlan&, const BwdIo&, hipStream_t);
extern template int launch_bwd_k<5>(const NsfPlan&, const TrainPlan&, const BwdIo&, hipStream_t);
extern template int launch_bwd_k<4>(const NsfPlan&, const TrainPlan&, const BwdIo&, hipStream_t);
extern template int launch_bwd_k<16>(const NsfPlan&, const TrainPlan&, const BwdIo&, hipStream_t);
extern template int launch_bwd_k<8>(const NsfPlan&, const TrainPlan&, const BwdIo&, hipStream_t);

// Which kernel family laid out a workspace: the two halves of a training pass take the decision independently from
// (cfg, n) and a process-wide threshold (sbi_amd_nsf_set_coop_max_rows); a backward pass that would read a stash the
// OTHER family wrote (the threshold moved in between) is refused instead of walking a foreign layout.
#include <mutex>
#include <unordered_map>
enum { WS_FAM_THROUGHPUT = 0, WS_FAM_COOP = 1, WS_FAM_GENERIC = 2 };
static std::mutex g_ws_mu;
static std::unordered_map<const void*, std::pair<int, int64_t>> g_ws_family;
static void ws_family_record(const void* ws, int fam, int64_t n) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  if (g_ws_family.size() > 256) g_ws_family.clear();     // (workspaces are few and long-lived; bound it anyway)
  g_ws_family[ws] = {fam, n};
}
// true: this workspace was last written by a forward pass of a different family / row count
static bool ws_family_mismatch(const void* ws, int fam, int64_t n) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  auto it = g_ws_family.find(ws);
  return it != g_ws_family.end() && (it->second.first != fam || it->second.second != n);
}

// forward half of the training pass: log p of every row + the per-transform state / activation stash
static int train_forward_impl(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats,
                              const float* theta, const float* x, int64_t n, int64_t x_rows, float* logp_out,
                              float* workspace, void* stream, bool fp32_bin) {
  if (!cfg || !packed || !zstats || !theta || !x || !workspace || n < 1 || x_rows < 1) return SBI_AMD_E_BADARG;
  NsfPlan pl;
  {
    CoopPlan cp;
    if (coop_applies(cfg, n, true, &pl, &cp)) {
      ws_family_record(workspace, WS_FAM_COOP, n);
      return coop_train_forward(cfg, pl, cp, packed + nsf_packed_floats(pl), zstats, theta, x, n, x_rows, logp_out,
                                workspace, stream);
    }
  }
  // (an E_LDS here speaks about the FORWARD kernel's 4-wave layout; the backward kernel has its own budget, checked
  // by build_train_plan, and the training forward picks its workgroup size in nsf_plan_for_rows)
  int rc = nsf_build_plan(cfg, TR_NW, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  TrainPlan tp;
  rc = build_train_plan(pl, n, &tp);
  if (fast_path_refuses(rc)) {
    ws_family_record(workspace, WS_FAM_GENERIC, n);
    const int rg = nsf_g_train_forward(cfg, packed, zstats, theta, x, n, x_rows, logp_out, workspace, stream);
    return fast_path_refuses(rg) ? rc : rg;
  }
  if (rc) return rc;
  ws_family_record(workspace, WS_FAM_THROUGHPUT, n);
  int64_t o_stash, o_noise, o_logp, o_gza, o_gzb, o_part, o_ast, o_pst;
  ws_layout(pl, tp, n, &o_stash, &o_noise, &o_logp, &o_gza, &o_gzb, &o_part, &o_ast, &o_pst);
  rc = nsf_log_prob_stash(cfg, packed, zstats, theta, x, n, x_rows, workspace + o_logp, workspace + o_noise,
                          workspace + o_stash, workspace + o_ast, TR_PSTASH ? workspace + o_pst : nullptr, stream, fp32_bin);
  if (rc) return rc;
  if (logp_out) {
    hipError_t e = hipMemcpyAsync(logp_out, workspace + o_logp, sizeof(float) * n, hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

extern "C" int sbi_amd_nsf_train_forward(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats,
                                         const float* theta, const float* x, int64_t n, int64_t x_rows,
                                         float* logp_out, float* workspace, void* stream) {
  return train_forward_impl(cfg, packed, zstats, theta, x, n, x_rows, logp_out, workspace, stream, false);
}

// backward half; loss_out (optional) = -log p of the stash's forward pass, written by the reduction kernel
static int train_backward_impl(const sbi_amd_nsf_config* cfg, const float* params, const float* packed,
                               const float* zstats, const float* x, int64_t n, int64_t x_rows,
                               const float* row_weight, float uniform_weight, float* grad_out,
                               float* grad_theta_out, float* grad_x_out, float* workspace, float* loss_out,
                               void* stream) {
  if (!cfg || !params || !packed || !zstats || !x || !grad_out || !workspace || n < 1 || x_rows < 1)
    return SBI_AMD_E_BADARG;
  if (grad_x_out && x_rows != n) return SBI_AMD_E_BADARG;   // one context row per theta row (no reduction here)
  NsfPlan pl;
  {
    // small batches: ONE cooperative backward launch over all transforms, then the same fixed-order slab reduction.
    // (The workspace was laid out by the cooperative forward: both halves take the same decision from (cfg, n).)
    CoopPlan cp;
    if (coop_applies(cfg, n, true, &pl, &cp)) {
      if (ws_family_mismatch(workspace, WS_FAM_COOP, n)) return SBI_AMD_E_BADARG;
      return coop_train_backward(cfg, pl, cp, params, packed + nsf_packed_floats(pl), zstats, x, n, x_rows, row_weight,
                                 uniform_weight, grad_out, grad_theta_out, grad_x_out, loss_out, workspace, stream);
    }
  }
  // (an E_LDS here speaks about the FORWARD kernel's 4-wave layout; the backward kernel has its own budget, checked
  // by build_train_plan, and the training forward picks its workgroup size in nsf_plan_for_rows)
  int rc = nsf_build_plan(cfg, TR_NW, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  TrainPlan tp;
  rc = build_train_plan(pl, n, &tp);
  if (ws_family_mismatch(workspace, fast_path_refuses(rc) ? WS_FAM_GENERIC : WS_FAM_THROUGHPUT, n)) return SBI_AMD_E_BADARG;
  if (fast_path_refuses(rc)) {
    if (grad_x_out) return rc;   // d loss / d embedded x comes from the wave-specialised kernel only
    if (loss_out) {
      const float* logp = nsf_g_logp(cfg, n, workspace);
      if (!logp) return rc;
      hipLaunchKernelGGL(neg_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logp,
                         loss_out, (long long)n);
    }
    const int rg = nsf_g_train_backward(cfg, params, packed, zstats, x, n, x_rows, row_weight, uniform_weight,
                                        grad_out, grad_theta_out, workspace, stream);
    return fast_path_refuses(rg) ? rc : rg;
  }
  if (rc) return rc;
  tp.grad_x = grad_x_out;
  hipStream_t st = (hipStream_t)stream;
  int64_t o_stash, o_noise, o_logp, o_gza, o_gzb, o_part, o_ast, o_pst;
  const int64_t ws_total = ws_layout(pl, tp, n, &o_stash, &o_noise, &o_logp, &o_gza, &o_gzb, &o_part, &o_ast, &o_pst);
  float* astash = workspace + o_ast;
  float* stash = workspace + o_stash;
  float* noise = workspace + o_noise;
  float* gz[2] = {workspace + o_gza, workspace + o_gzb};
  float* partial = workspace + o_part;
  long long* dbg = (long long*)(workspace + ws_total - 2048);
  BwdIo io{};
  io.t_hi = pl.T - 1;
  io.t_lo = 0;
  io.packed = packed;
  io.zstats = zstats;
  io.stash = stash;
  io.x = x;
  io.noise = noise;
  io.gz[0] = gz[0];
  io.gz[1] = gz[1];
  io.row_w = row_weight;
  io.uni_w = uniform_weight;
  io.n = n;
  io.x_rows = x_rows;
  io.partial = partial;
  io.grad_theta = grad_theta_out;
  io.astash = astash;
  io.pstash = TR_PSTASH ? workspace + o_pst : nullptr;
  io.dbg = sbi_amd_dbg_timeline() ? dbg : nullptr;
  switch (cfg->K) {
    case 4: rc = launch_bwd_k<4>(pl, tp, io, st); break;
    case 5: rc = launch_bwd_k<5>(pl, tp, io, st); break;
    case 8: rc = launch_bwd_k<8>(pl, tp, io, st); break;
    case 10: rc = launch_bwd_k<10>(pl, tp, io, st); break;
    case 16: rc = launch_bwd_k<16>(pl, tp, io, st); break;
    default: rc = SBI_AMD_E_UNSUPPORTED;
  }
  if (rc) return rc;
  float* sq_out = workspace + ws_total - 2048 - (thr_sq_parts(pl, tp) + 3) / 4 * 4;
  hipLaunchKernelGGL(nsf_grad_reduce_kernel, dim3((tp.PLP / 4 + 63) / 64, pl.T), dim3(64 * RED_GROUPS), 0, st, pl, tp,
                     params, partial, grad_out, (const float*)(workspace + o_logp), loss_out, (long long)n, sq_out);
  return (int)hipGetLastError();
}

// backward half: consumes the stash the preceding sbi_amd_nsf_train_forward left in `workspace`
// (same cfg, n, x, x_rows, packed image and stream order); grad_out = d( sum_n w_n * (-log p_n) ) / d params
extern "C" int sbi_amd_nsf_train_backward(const sbi_amd_nsf_config* cfg, const float* params, const float* packed,
                                          const float* zstats, const float* x, int64_t n, int64_t x_rows,
                                          const float* row_weight, float uniform_weight, float* grad_out,
                                          float* grad_theta_out, float* grad_x_out, float* workspace, void* stream) {
  return train_backward_impl(cfg, params, packed, zstats, x, n, x_rows, row_weight, uniform_weight, grad_out,
                             grad_theta_out, grad_x_out, workspace, nullptr, stream);
}

// one-call form: forward + backward with weights known up front (plain NPE loss)
extern "C" int sbi_amd_nsf_loss_fwd_bwd(const sbi_amd_nsf_config* cfg, const float* params, const float* packed,
                                        const float* zstats, const float* theta, const float* x, int64_t n,
                                        int64_t x_rows, const float* row_weight, float uniform_weight,
                                        float* loss_out, float* grad_out, float* grad_theta_out, float* grad_x_out,
                                        float* workspace, void* stream) {
  if (!cfg || !params || !packed || !zstats || !theta || !x || !grad_out || !workspace || n < 1 || x_rows < 1)
    return SBI_AMD_E_BADARG;
  // (the fused step's forward: its log p is the reported training loss only -- plain fp32 bin, nsf_device.h)
  int rc = train_forward_impl(cfg, packed, zstats, theta, x, n, x_rows, nullptr, workspace, stream, true);
  if (rc) return rc;
  return train_backward_impl(cfg, params, packed, zstats, x, n, x_rows, row_weight, uniform_weight, grad_out,
                             grad_theta_out, grad_x_out, workspace, loss_out, stream);
}

// Where the gradient reduction of the last training pass of an n-row batch left the partial sums of squares of grad_out
// (their sum = |grad_out|^2, what clip_grad_norm_ needs: sbi_amd_adam_clip_step_parts), and how many there are.
// NULL / 0: that pass does not leave any (the generic training pass), the caller runs sbi_amd_adam_clip_step.
extern "C" const float* sbi_amd_nsf_train_sqnorm_parts(const sbi_amd_nsf_config* cfg, int64_t n, const float* workspace,
                                                       int64_t* n_parts) {
  if (n_parts) *n_parts = 0;
  if (!cfg || !workspace || n < 1) return nullptr;
  NsfPlan pl;
  {
    CoopPlan cp;
    if (coop_applies(cfg, n, true, &pl, &cp)) return coop_sqnorm_parts(pl, cp, n, workspace, n_parts);
  }
  int rc = nsf_build_plan(cfg, TR_NW, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return nullptr;
  TrainPlan tp;
  rc = build_train_plan(pl, n, &tp);
  if (rc || fast_path_refuses(rc)) return nullptr;
  int64_t a, b, c, d, e, f, g;
  const int64_t ws_total = ws_layout(pl, tp, n, &a, &b, &c, &d, &e, &f, &g);
  if (n_parts) *n_parts = thr_sq_parts(pl, tp);
  return workspace + ws_total - 2048 - (thr_sq_parts(pl, tp) + 3) / 4 * 4;
}
