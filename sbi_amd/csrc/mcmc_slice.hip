// mcmc_slice.hip -- one tick of the vectorised slice sampler, all chains in lockstep on the device.
// Restates the per-chain state machine of SliceSamplerVectorized.run (sbi/samplers/mcmc/slice_numpy.py:
// 353-587: BEGIN -> LOWER -> UPPER -> SAMPLE_SLICE -> BEGIN ..., bracket-width tuning during the first
// `tuning` sweeps, a fresh random dimension order per sweep).  The reference advances every chain in a
// Python loop between two batched log-prob evaluations; here the evaluation is the fused NSF log_prob
// kernel on `next_param` and this kernel is the loop body: one thread per chain, no host involvement.
// Uniform random numbers come from the caller (torch's generator, DESIGN.md RNG contract).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sbi_amd_nsf.h"

#include "mcmc_tick.h"

__global__ void slice_tick_kernel(int C, int D, int num_samples, int tuning, float max_width,
                                  const float* __restrict__ logp, const float* __restrict__ logp_offset,
                                  const float* __restrict__ rnd, float* __restrict__ x, float* __restrict__ next_param,
                                  float* __restrict__ width, int* __restrict__ order, int* __restrict__ istate,
                                  float* __restrict__ fstate, float* __restrict__ samples, int* __restrict__ done_count,
                                  unsigned long long seed, unsigned long long tick_no, int kind,
                                  const float* __restrict__ tp0, const float* __restrict__ tp1,
                                  float* __restrict__ theta_next, float* __restrict__ lad_next) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  slice_tick_one(c, D, num_samples, tuning, max_width, logp, logp_offset, rnd, x, next_param, width, order, istate, fstate,
                 samples, done_count, seed, tick_no, kind, tp0, tp1, theta_next, lad_next);
}

// theta = T^-1(u) and log|det dT/dtheta| for the two parameter transforms `mcmc_transform` builds
// (sbi/utils/sbiutils.py:867-980): kind 1 = z-scoring with the prior's mean / std (unbounded support),
// kind 2 = logit map of a box [low, high] (biject_to(Independent(Uniform))); kind 0 = identity.
__global__ void mcmc_to_constrained_kernel(int kind, int C, int D, const float* __restrict__ p0,
                                           const float* __restrict__ p1, const float* __restrict__ u,
                                           float* __restrict__ theta, float* __restrict__ lad) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float acc = 0.f;
  for (int d = 0; d < D; ++d) {
    const float v = u[(size_t)c * D + d];
    float th = v;
    if (kind == 1) {          // u = (theta - loc) / scale
      th = p0[d] + p1[d] * v;
      acc -= logf(fabsf(p1[d]));
    } else if (kind == 2) {   // theta = low + (high - low) * sigmoid(u)
      const float sg = 1.f / (1.f + expf(-v));
      th = p0[d] + p1[d] * sg;
      // -log|d theta/d u| = -(log(high-low) + log sigmoid(u) + log sigmoid(-u)), softplus form for the tails
      const float sp_pos = fmaxf(v, 0.f) + log1pf(expf(-fabsf(v)));    // softplus(u)  = -log sigmoid(-u)
      const float sp_neg = sp_pos - v;                                   // softplus(-u) = -log sigmoid(u)
      acc -= logf(p1[d]) - sp_pos - sp_neg;
    }
    theta[(size_t)c * D + d] = th;
  }
  lad[c] = acc;
}

extern "C" int sbi_amd_mcmc_to_constrained(int32_t kind, int32_t num_chains, int32_t dim, const float* p0,
                                           const float* p1, const float* u, float* theta_out, float* logabsdet_out,
                                           void* stream) {
  if (kind < 0 || kind > 2 || num_chains < 1 || dim < 1 || !u || !theta_out || !logabsdet_out || (kind && (!p0 || !p1)))
    return SBI_AMD_E_BADARG;
  hipLaunchKernelGGL(mcmc_to_constrained_kernel, dim3((num_chains + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     kind, num_chains, dim, p0, p1, u, theta_out, logabsdet_out);
  return (int)hipGetLastError();
}

extern "C" int sbi_amd_mcmc_slice_tick(int32_t num_chains, int32_t dim, int32_t num_samples, int32_t tuning,
                                       float max_width, const float* logp, const float* logp_offset,
                                       const float* uniforms, float* x,
                                       float* next_param, float* width, int32_t* order, int32_t* istate,
                                       float* fstate, float* samples, int32_t* done_count, uint64_t seed,
                                       uint64_t tick_no, int32_t kind, const float* p0, const float* p1,
                                       float* theta_next, float* logabsdet_next, void* stream) {
  if (num_chains < 1 || dim < 1 || num_samples < 0 || tuning < 0 || !logp || !x || !next_param ||
      !width || !order || !istate || !fstate || !samples || !done_count)
    return SBI_AMD_E_BADARG;
  if ((theta_next != nullptr) != (logabsdet_next != nullptr) || kind < 0 || kind > 2 ||
      (theta_next && kind && (!p0 || !p1)))
    return SBI_AMD_E_BADARG;
  hipLaunchKernelGGL(slice_tick_kernel, dim3((num_chains + 255) / 256), dim3(256), 0, (hipStream_t)stream, num_chains,
                     dim, num_samples, tuning, max_width, logp, logp_offset, uniforms, x, next_param, width, order, istate,
                     fstate, samples, done_count, (unsigned long long)seed, (unsigned long long)tick_no, kind, p0, p1,
                     theta_next, logabsdet_next);
  return (int)hipGetLastError();
}
