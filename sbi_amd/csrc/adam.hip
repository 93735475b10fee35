// adam.hip -- fused global-norm gradient clip + Adam step on the flat fp32
// parameter buffer (gfx950).  Replaces, for the NSF hot path,
//   torch.nn.utils.clip_grad_norm_(params, max_norm)   trainers/base.py:1182-1186
//   torch.optim.Adam.step()                            trainers/base.py:1097, 1187
// (Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).
// Two launches, no atomics, deterministic: (1) per-workgroup partial sums of
// g^2 -> scratch[1..nwg]; (2) every workgroup re-reduces the <=ADAM_NWG partials
// in a fixed order, derives the clip coefficient and updates its slice.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/sbi_amd_nsf.h"
#include "adam_math.h"

#define ADAM_NWG 128
#define ADAM_THREADS 256

__global__ void __launch_bounds__(ADAM_THREADS)
grad_sqnorm_partials(const float* __restrict__ g, long long count, float* __restrict__ scratch) {
  __shared__ float red[ADAM_THREADS / 64];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * ADAM_THREADS + threadIdx.x; i < count;
       i += (long long)gridDim.x * ADAM_THREADS) {
    float v = g[i];
    acc += v * v;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < ADAM_THREADS / 64; ++w) s += red[w];
    scratch[1 + blockIdx.x] = s;
  }
}

__global__ void __launch_bounds__(ADAM_THREADS)
adam_update(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            long long count, float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
            float max_norm, const float* __restrict__ parts, int npart, float* __restrict__ scratch) {
  float norm;
  const float coef = adam_clip_coef_block(parts, npart, max_norm, &norm);
  if (blockIdx.x == 0 && threadIdx.x == 0) scratch[0] = norm;
  const AdamK k = {coef, beta1, beta2, eps, lr / bc1, bc2_sqrt};
  for (long long i = (long long)blockIdx.x * ADAM_THREADS + threadIdx.x; i < count;
       i += (long long)gridDim.x * ADAM_THREADS) {
    float mi = m[i], vi = v[i];
    p[i] = adam_apply_one(p[i], g[i], mi, vi, k);     // (roundings pinned: adam_math.h)
    m[i] = mi;
    v[i] = vi;
  }
}

static int adam_launch(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, int64_t step,
                       float lr, float beta1, float beta2, float eps, float max_norm, const float* parts, int64_t n_parts,
                       float* scratch, hipStream_t st) {
  static_assert(ADAM_THREADS == ADAM_BLOCK, "adam_clip_coef_block is written for the update kernel's workgroup size");
  int nwg = (int)((count + ADAM_THREADS * 4 - 1) / (ADAM_THREADS * 4));     // norm kernel: <= ADAM_NWG partial sums (scratch)
  if (nwg > ADAM_NWG) nwg = ADAM_NWG;
  if (nwg < 1) nwg = 1;
  // update kernel: one element per thread up to 512 workgroups (the kernel is a latency chain -- load, a dozen dependent
  // flops, store -- per element a thread owns: 98 025 parameters over 96 workgroups were four such chains in a row)
  int uwg = (int)((count + ADAM_THREADS - 1) / ADAM_THREADS);
  if (uwg > 512) uwg = 512;
  if (uwg < 1) uwg = 1;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  if (!parts) {     // the caller has no partial sums of squares of `grad`: one more launch makes them
    hipLaunchKernelGGL(grad_sqnorm_partials, dim3(nwg), dim3(ADAM_THREADS), 0, st, grad, (long long)count, scratch);
    parts = scratch + 1;
    n_parts = nwg;
  }
  hipLaunchKernelGGL(adam_update, dim3(uwg), dim3(ADAM_THREADS), 0, st, params, grad, exp_avg, exp_avg_sq,
                     (long long)count, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), max_norm, parts, (int)n_parts,
                     scratch);
  return (int)hipGetLastError();
}

extern "C" int sbi_amd_adam_clip_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq,
                                      int64_t count, int64_t step, float lr, float beta1, float beta2, float eps,
                                      float max_norm, float* scratch, void* stream) {
  if (!params || !grad || !exp_avg || !exp_avg_sq || !scratch || count < 0 || step < 1) return SBI_AMD_E_BADARG;
  if (count == 0) return 0;
  return adam_launch(params, grad, exp_avg, exp_avg_sq, count, step, lr, beta1, beta2, eps, max_norm, nullptr, 0, scratch,
                     (hipStream_t)stream);
}

// The same step when the caller already holds partial sums of squares of `grad` (their sum = |grad|^2): the training
// pass's gradient reduction leaves them in its workspace (sbi_amd_nsf_train_sqnorm_parts), which saves the norm launch.
extern "C" int sbi_amd_adam_clip_step_parts(float* params, const float* grad, float* exp_avg, float* exp_avg_sq,
                                            int64_t count, int64_t step, float lr, float beta1, float beta2, float eps,
                                            float max_norm, const float* sqnorm_parts, int64_t n_parts, float* scratch,
                                            void* stream) {
  if (!params || !grad || !exp_avg || !exp_avg_sq || !scratch || !sqnorm_parts || n_parts < 1 || n_parts > (1 << 24) ||
      count < 0 || step < 1)
    return SBI_AMD_E_BADARG;
  if (count == 0) return 0;
  return adam_launch(params, grad, exp_avg, exp_avg_sq, count, step, lr, beta1, beta2, eps, max_norm, sqnorm_parts,
                     n_parts, scratch, (hipStream_t)stream);
}
