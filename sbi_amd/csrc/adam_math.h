#pragma once
// adam_math.h -- ONE definition of the clip + Adam arithmetic for every kernel that applies it (adam.hip's
// adam_update; anything that fuses the step into another kernel later), with the roundings pinned: hipcc contracts a * b + c into an
// FMA wherever it likes, and `beta2 * v + (1 - beta2) * g * g` has two legal contractions -- two kernels that spell
// the same expression then differ in the last bit of ~0.2 % of the parameters after the second step (measured).
// Data-parallel replicas (and any two kernels that apply the same step) are held to bit-identity, so the choice is
// made here:
//   m <- m + (1 - beta1) (g - m)                 exp_avg.lerp_(grad, 1 - beta1)       one FMA
//   v <- beta2 v + ((1 - beta2) g) g             mul_(beta2).addcmul_(g, g, 1-beta2)  products rounded, one FMA
//   p <- p - (lr / bc1) * (m / (sqrt(v) / sqrt(bc2) + eps))                           one FMA
// (torch.optim.Adam, trainers/base.py:1097, :1187; defaults betas (0.9, 0.999), eps 1e-8, no weight decay / amsgrad).
#include <hip/hip_runtime.h>

struct AdamK {
  float coef;        // gradient scale of clip_grad_norm_ (1 when the norm is below max_norm)
  float beta1, beta2, eps, step_size, bc2_sqrt;
};

__device__ __forceinline__ float adam_apply_one(float p, float g, float& m, float& v, const AdamK& k) {
#pragma clang fp contract(off)
  const float gi = g * k.coef;
  const float mi = __builtin_fmaf(1.f - k.beta1, gi - m, m);
  const float gg = ((1.f - k.beta2) * gi) * gi;
  const float vi = __builtin_fmaf(k.beta2, v, gg);
  m = mi;
  v = vi;
  const float denom = __builtin_sqrtf(vi) / k.bc2_sqrt + k.eps;
  return __builtin_fmaf(-k.step_size, mi / denom, p);
}

// clip coefficient from the partial sums of squares, computed by EVERY workgroup of the update kernel the same way
// (fixed order: thread t sums parts t, t + 256, ...; then a fixed tree): deterministic, identical in all workgroups.
// blockDim.x must be ADAM_BLOCK.  Also returns the norm.
#define ADAM_BLOCK 256
__device__ __forceinline__ float adam_clip_coef_block(const float* __restrict__ partials, int npart, float max_norm,
                                                      float* norm_out) {
  __shared__ float red[ADAM_BLOCK];
  float s = 0.f;
  for (int i = threadIdx.x; i < npart; i += ADAM_BLOCK) s += partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = ADAM_BLOCK / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  const float norm = sqrtf(red[0]);
  *norm_out = norm;
  return max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.f) : 1.f;   // clip_grad_norm_
}
