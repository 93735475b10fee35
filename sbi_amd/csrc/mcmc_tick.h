// mcmc_tick.h -- one tick of one chain of the vectorised slice sampler (sbi/samplers/mcmc/slice_numpy.py:353-587),
// shared by the stand-alone tick kernel (mcmc_slice.hip) and the persistent sampler kernel (nsf_coop_kernel.h, MC = true).
#pragma once
#include <hip/hip_runtime.h>

// every product and sum rounds on its own (as in the reference's numpy arithmetic and in the test's tensor
// restatement): no fused multiply-add contraction in the bracket arithmetic
#pragma clang fp contract(off)

enum { ST_BEGIN = 0, ST_LOWER = 1, ST_UPPER = 2, ST_SAMPLE = 3, ST_DONE = 4 };

// Philox4x32-10 (Salmon et al., SC'11): counter = (tick, tick >> 32, chain, block), key = the run's seed.  The
// reference's slice sampler draws from NumPy's global generator (slice_numpy.py:353-587) -- there is no stream to
// reproduce, only a distribution; the seed comes from torch's generator, so `torch.manual_seed` fixes a run.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(unsigned r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }   // [0, 1), as torch.rand

// theta = T^-1(u) of ONE coordinate and its contribution to log|det dT/dtheta| (see mcmc_to_constrained_kernel)
__device__ __forceinline__ float to_constrained_1(int kind, float v, float p0, float p1, float& acc) {
  if (kind == 1) {
    acc -= logf(fabsf(p1));
    return p0 + p1 * v;
  }
  if (kind == 2) {
    const float sg = 1.f / (1.f + expf(-v));
    const float sp_pos = fmaxf(v, 0.f) + log1pf(expf(-fabsf(v)));
    const float sp_neg = sp_pos - v;
    acc -= logf(p1) - sp_pos - sp_neg;
    return p0 + p1 * sg;
  }
  return v;
}

__device__ __forceinline__ void slice_tick_one(const int c, int D, int num_samples, int tuning, float max_width,
                                  const float* __restrict__ logp, const float* logp_offset,   // (C), (C)|null (may alias lad_next)
                                  const float* __restrict__ rnd,                                  // (C, 4 + D)
                                  float* __restrict__ x, float* __restrict__ next_param,           // (C, D) each
                                  float* __restrict__ width,                                       // (C, D)
                                  int* __restrict__ order, int* __restrict__ istate,                // (C, D), (C, 4): state, i, t, -
                                  float* __restrict__ fstate,                                       // (C, 8): cxi wi lx ux xi logu
                                  float* __restrict__ samples,                                      // (C, num_samples, D)
                                  int* __restrict__ done_count,
                                  // fused extras (all optional): in-kernel uniforms when rnd == null; the NEXT evaluation
                                  // point already mapped to constrained space (theta_next, lad_next = what the batched
                                  // log_prob kernel and this kernel's `logp_offset` read in the next tick)
                                  unsigned long long seed, unsigned long long tick_no, int kind,
                                  const float* __restrict__ tp0, const float* __restrict__ tp1,
                                  float* __restrict__ theta_next, float* lad_next) {
  int st = istate[4 * c + 0];
  if (st == ST_DONE) return;
  // the four uniforms of a tick (u[0]: slice height, u[1]: bracket position, u[2]: proposal; u[3] unused) and, at
  // the end of a sweep, D more for the dimension order
  float u4[4];
  if (rnd) {
#pragma unroll
    for (int q = 0; q < 4; ++q) u4[q] = rnd[(size_t)c * (4 + D) + q];
  } else {
    unsigned r[4];
    philox4x32_10((unsigned)tick_no, (unsigned)(tick_no >> 32), (unsigned)c, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
#pragma unroll
    for (int q = 0; q < 4; ++q) u4[q] = u01(r[q]);
  }
  int i = istate[4 * c + 1], t = istate[4 * c + 2];
  float* fs = fstate + 8 * c;
  float cxi = fs[0], wi = fs[1], lx = fs[2], ux = fs[3], xi = fs[4], logu = fs[5];
  const float lp = logp_offset ? logp[c] - logp_offset[c] : logp[c];
  const float* u = u4;
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
        unsigned rr[4] = {0u, 0u, 0u, 0u};
        for (int d = D - 1; d > 0; --d) {
          float ud;
          if (rnd) {
            ud = rnd[(size_t)c * (4 + D) + 4 + d];
          } else {
            if ((d & 3) == 3 || d == D - 1)      // one Philox block serves four consecutive dims
              philox4x32_10((unsigned)tick_no, (unsigned)(tick_no >> 32), (unsigned)c, 1u + (unsigned)(d >> 2),
                            (unsigned)seed, (unsigned)(seed >> 32), rr);
            ud = u01(rr[d & 3]);
          }
          int k = (int)(ud * (float)(d + 1));
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
  if (theta_next) {      // the point the next tick evaluates, in constrained space, and its log|det|
    float acc = 0.f;
    for (int d = 0; d < D; ++d)
      theta_next[(size_t)c * D + d] = to_constrained_1(kind, np_[d], kind ? tp0[d] : 0.f, kind ? tp1[d] : 1.f, acc);
    lad_next[c] = acc;
  }
}


#pragma clang fp contract(fast)
