// atomic.hip -- the host-side arithmetic of multi-round NPE-C's atomic loss as two launches
// (sbi/inference/trainers/npe/npe_c.py:356-440, `_log_prob_proposal_posterior_atomic`):
//   1. atoms: for every row b of the batch, num_atoms - 1 contrasting rows != b, uniform without replacement
//      (npe_c.py:387-392: multinomial over a (B, B) matrix of ones - eye), and the (A, B, D) atom tensor
//      (atom 0 = the row's own theta, atoms-major so the kernels read x[r % B] and the context is never repeated;
//      the reference materialises repeat_rows(x, num_atoms));
//   2. after the batched log_prob: u = log q - log prior, log q~_b = u[0, b] - logsumexp_a u[a, b] and the weights
//      d log q~_b / d log q[a, b] = delta_{a0} - softmax_a(u[., b]) the backward pass runs with.
// The reference builds these from ~80 eager tensor operations per step; the contrasting set here is Floyd's algorithm
// plus a Fisher-Yates shuffle per row on Philox4x32-10 uniforms keyed by a seed from torch's generator (the reference's
// multinomial stream cannot be reproduced on a device anyway: the distribution is what is kept -- every ordered
// (A-1)-tuple of distinct rows != b is equally likely).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sbi_amd_nsf.h"

#define AT_MAX_K 63      // contrasting atoms per row this kernel takes (num_atoms <= 64); the host falls back beyond

__device__ __forceinline__ void at_philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
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
// uniform integer in [0, n) from 32 random bits (multiply-shift; bias < n / 2^32)
__device__ __forceinline__ unsigned at_below(unsigned r, unsigned n) { return (unsigned)(((unsigned long long)r * n) >> 32); }

__global__ void __launch_bounds__(256)
atomic_atoms_kernel(const float* __restrict__ theta, int B, int A, int D, unsigned long long seed,
                    const long long* __restrict__ choices_in, long long* __restrict__ choices_out,
                    float* __restrict__ atoms) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int k = A - 1;
  int pick[AT_MAX_K];
  if (choices_in) {
    for (int i = 0; i < k; ++i) pick[i] = (int)choices_in[(long long)b * k + i];
  } else {
    // Floyd: k distinct values of range(m), m = B - 1
    const int m = B - 1;
    unsigned r[4];
    int used = 4, blk = 0;
    auto next = [&]() -> unsigned {
      if (used == 4) { at_philox((unsigned)b, (unsigned)blk++, 0x61746f6du, 0u, (unsigned)seed, (unsigned)(seed >> 32), r); used = 0; }
      return r[used++];
    };
    for (int i = 0, j = m - k; j < m; ++i, ++j) {
      int t = (int)at_below(next(), (unsigned)(j + 1));
      bool taken = false;
      for (int q = 0; q < i; ++q) taken = taken || (pick[q] == t);
      pick[i] = taken ? j : t;
    }
    // Floyd yields a uniformly random SET: shuffle so that the positions are exchangeable too
    for (int i = k - 1; i > 0; --i) {
      const int q = (int)at_below(next(), (unsigned)(i + 1));
      const int tmp = pick[i]; pick[i] = pick[q]; pick[q] = tmp;
    }
    for (int i = 0; i < k; ++i) pick[i] += (pick[i] >= b) ? 1 : 0;     // skip the own row
  }
  if (choices_out)
    for (int i = 0; i < k; ++i) choices_out[(long long)b * k + i] = pick[i];
  // atoms-major (A, B, D): atom 0 the own row
  for (int d = 0; d < D; ++d) atoms[(long long)b * D + d] = theta[(long long)b * D + d];
  for (int i = 0; i < k; ++i) {
    const float* src = theta + (long long)pick[i] * D;
    float* dst = atoms + ((long long)(i + 1) * B + b) * D;
    for (int d = 0; d < D; ++d) dst[d] = src[d];
  }
}

// log q, log prior: (A, B) atoms-major.  lpp (B) = log q~; w (A, B) = scale * (delta_{a0} - softmax_a [+ mask_b at a = 0])
__global__ void __launch_bounds__(256)
atomic_weights_kernel(const float* __restrict__ logq, const float* __restrict__ logprior, const float* __restrict__ masks,
                      int B, int A, float scale, float* __restrict__ lpp, float* __restrict__ w) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float mx = -INFINITY;
  for (int a = 0; a < A; ++a) mx = fmaxf(mx, logq[(long long)a * B + b] - logprior[(long long)a * B + b]);
  float s = 0.f;
  for (int a = 0; a < A; ++a) s += expf(logq[(long long)a * B + b] - logprior[(long long)a * B + b] - mx);
  const float lse = mx + logf(s);
  const float u0 = logq[b] - logprior[b];
  float out = u0 - lse;
  const float m = masks ? masks[b] : 0.f;
  if (masks) out = m * logq[b] + out;          // combined loss (npe_c.py:425-436): + the MLE term on prior samples
  lpp[b] = out;
  for (int a = 0; a < A; ++a) {
    const float u = logq[(long long)a * B + b] - logprior[(long long)a * B + b];
    float g = -expf(u - lse);
    if (a == 0) g += 1.f + m;
    w[(long long)a * B + b] = scale * g;
  }
}

extern "C" int sbi_amd_atomic_atoms(const float* theta, int32_t batch, int32_t num_atoms, int32_t dim, uint64_t seed,
                                    const int64_t* choices_in, int64_t* choices_out, float* atoms_out, void* stream) {
  if (!theta || !atoms_out || batch < 2 || dim < 1 || num_atoms < 2 || num_atoms > batch) return SBI_AMD_E_BADARG;
  if (num_atoms - 1 > AT_MAX_K) return SBI_AMD_E_UNSUPPORTED;
  hipLaunchKernelGGL(atomic_atoms_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, theta, batch,
                     num_atoms, dim, (unsigned long long)seed, (const long long*)choices_in, (long long*)choices_out,
                     atoms_out);
  return (int)hipGetLastError();
}

extern "C" int sbi_amd_atomic_weights(const float* log_q, const float* log_prior, const float* masks, int32_t batch,
                                      int32_t num_atoms, float scale, float* log_prob_out, float* weights_out,
                                      void* stream) {
  if (!log_q || !log_prior || !log_prob_out || !weights_out || batch < 1 || num_atoms < 2) return SBI_AMD_E_BADARG;
  hipLaunchKernelGGL(atomic_weights_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, log_q,
                     log_prior, masks, batch, num_atoms, scale, log_prob_out, weights_out);
  return (int)hipGetLastError();
}
