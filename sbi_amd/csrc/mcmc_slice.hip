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

// every product and sum rounds on its own (as in the reference's numpy arithmetic and in the test's tensor
// restatement): no fused multiply-add contraction in the bracket arithmetic
#pragma clang fp contract(off)

enum { ST_BEGIN = 0, ST_LOWER = 1, ST_UPPER = 2, ST_SAMPLE = 3, ST_DONE = 4 };

__global__ void slice_tick_kernel(int C, int D, int num_samples, int tuning, float max_width,
                                  const float* __restrict__ logp, const float* __restrict__ rnd,   // (C), (C, 4 + D)
                                  float* __restrict__ x, float* __restrict__ next_param,           // (C, D) each
                                  float* __restrict__ width,                                       // (C, D)
                                  int* __restrict__ order, int* __restrict__ istate,                // (C, D), (C, 4): state, i, t, -
                                  float* __restrict__ fstate,                                       // (C, 8): cxi wi lx ux xi logu
                                  float* __restrict__ samples,                                      // (C, num_samples, D)
                                  int* __restrict__ done_count) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  int st = istate[4 * c + 0];
  if (st == ST_DONE) return;
  int i = istate[4 * c + 1], t = istate[4 * c + 2];
  float* fs = fstate + 8 * c;
  float cxi = fs[0], wi = fs[1], lx = fs[2], ux = fs[3], xi = fs[4], logu = fs[5];
  const float lp = logp[c];
  const float* u = rnd + (size_t)c * (4 + D);
  const int dim = order[(size_t)c * D + i];
  float* xp = x + (size_t)c * D;
  float* np_ = next_param + (size_t)c * D;
  if (st == ST_BEGIN) {
    // position the bracket randomly around the current sample
    cxi = xp[dim];
    wi = width[(size_t)c * D + dim];
    logu = lp + logf(1.0f - u[0]);
    lx = cxi - wi * u[1];
    ux = lx + wi;
    np_[dim] = lx;
    st = ST_LOWER;
  } else if (st == ST_LOWER) {
    const bool outside_lower = (lp >= logu) && (cxi - lx < max_width);
    if (outside_lower) {
      lx -= wi;
      np_[dim] = lx;
    } else {
      np_[dim] = ux;
      st = ST_UPPER;
    }
  } else if (st == ST_UPPER) {
    const bool outside_upper = (lp >= logu) && (ux - cxi < max_width);
    if (outside_upper) {
      ux += wi;
      np_[dim] = ux;
    } else {
      xi = (ux - lx) * u[2] + lx;
      np_[dim] = xi;
      st = ST_SAMPLE;
    }
  } else {   // ST_SAMPLE
    const bool rejected = lp < logu;
    if (rejected) {   // shrink the bracket towards the current point
      if (xi < cxi) lx = xi; else ux = xi;
      xi = (ux - lx) * u[2] + lx;
      np_[dim] = xi;
    } else {
      xp[dim] = xi;   // accept: x = next_param
      if (t < tuning) {
        float* w = width + (size_t)c * D + dim;
        *w += ((ux - lx) - *w) / (float)(t + 1);
      }
      st = ST_BEGIN;
      if (i < D - 1) {
        ++i;
      } else {
        if (t >= tuning) {
          float* out = samples + ((size_t)c * num_samples + (t - tuning)) * D;
          for (int d = 0; d < D; ++d) out[d] = xp[d];
        }
        ++t;
        i = 0;
        // fresh dimension order: Fisher-Yates on the caller's uniforms
        int* ord = order + (size_t)c * D;
        for (int d = 0; d < D; ++d) ord[d] = d;
        for (int d = D - 1; d > 0; --d) {
          int k = (int)(u[4 + d] * (float)(d + 1));
          k = k > d ? d : k;
          const int tmp = ord[d]; ord[d] = ord[k]; ord[k] = tmp;
        }
        if (t >= num_samples + tuning) {
          st = ST_DONE;
          atomicAdd(done_count, 1);
        }
      }
    }
  }
  istate[4 * c + 0] = st; istate[4 * c + 1] = i; istate[4 * c + 2] = t;
  fs[0] = cxi; fs[1] = wi; fs[2] = lx; fs[3] = ux; fs[4] = xi; fs[5] = logu;
}

extern "C" int sbi_amd_mcmc_slice_tick(int32_t num_chains, int32_t dim, int32_t num_samples, int32_t tuning,
                                       float max_width, const float* logp, const float* uniforms, float* x,
                                       float* next_param, float* width, int32_t* order, int32_t* istate,
                                       float* fstate, float* samples, int32_t* done_count, void* stream) {
  if (num_chains < 1 || dim < 1 || num_samples < 0 || tuning < 0 || !logp || !uniforms || !x || !next_param ||
      !width || !order || !istate || !fstate || !samples || !done_count)
    return SBI_AMD_E_BADARG;
  hipLaunchKernelGGL(slice_tick_kernel, dim3((num_chains + 255) / 256), dim3(256), 0, (hipStream_t)stream, num_chains,
                     dim, num_samples, tuning, max_width, logp, uniforms, x, next_param, width, order, istate, fstate,
                     samples, done_count);
  return (int)hipGetLastError();
}
