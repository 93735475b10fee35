// shuffle.hip -- the training loop's minibatch sampler on the device: one launch per minibatch gathers the batch's
// (theta, x) rows of the resident simulations in a fresh pseudo-random order.
//
// Replaces, for the NPE inner loop, torch's SubsetRandomSampler + DataLoader collation
// (sbi/inference/trainers/base.py:541-560: `SubsetRandomSampler(self.train_indices.tolist())`, drop_last, batch_size
// = min(training_batch_size, n_train)): per epoch a random permutation pi of the training split, batch b = rows
// train_indices[pi[b B : (b + 1) B]].  torch draws pi with randperm (a sort of random keys: a dozen launches, and the
// gather of theta and x two more); here pi is never materialised: pi_e(i) is a keyed pseudo-random permutation of
// [0, n) evaluated per row -- a 6-round balanced Feistel network on the smallest even-width binary domain >= n with
// cycle walking (Black & Rogaway 2002: walking a permutation of the larger domain until it lands in [0, n) is a
// permutation of [0, n)).  Every epoch uses a fresh key, so every epoch visits every training row exactly once, in
// an order that is a different member of the family; the family is not ALL n! orders (neither is a 64-bit-seeded
// randperm), which is immaterial for minibatch SGD.  tests/test_shuffle_*.py hold a Python restatement to the kernel
// index for index and check bijectivity and the statistics of the orders.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/sbi_amd_nsf.h"

__host__ __device__ __forceinline__ uint32_t shf_mix(uint32_t v) {     // murmur3's 32-bit finaliser
  v ^= v >> 16; v *= 0x85ebca6bu; v ^= v >> 13; v *= 0xc2b2ae35u; v ^= v >> 16;
  return v;
}
// position i of the order -> element pi(i) of [0, n); hb = bits per Feistel half (2 hb >= ceil(log2 n))
__host__ __device__ __forceinline__ uint32_t shf_prp(uint32_t i, uint32_t n, int hb, uint64_t key) {
  const uint32_t mask = (1u << hb) - 1u;
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  uint32_t v = i;
  do {
    uint32_t l = v >> hb, r = v & mask;
#pragma unroll
    for (int round = 0; round < 6; ++round) {
      const uint32_t f = shf_mix(r + 0x9e3779b9u * (uint32_t)(round + 1) + ((round & 1) ? k1 : k0)) & mask;
      const uint32_t t = l ^ f;
      l = r;
      r = t;
    }
    v = (l << hb) | r;
  } while (v >= n);
  return v;
}

__global__ void __launch_bounds__(256)
shuffled_gather_kernel(const float* __restrict__ a, int da, const float* __restrict__ b, int db,
                       const long long* __restrict__ base_idx, unsigned n_perm, int hb, unsigned long long key,
                       long long offset, long long count, float* __restrict__ a_out, float* __restrict__ b_out,
                       long long* __restrict__ idx_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const unsigned p = shf_prp((unsigned)(offset + i), n_perm, hb, key);
  const long long src = base_idx ? base_idx[p] : (long long)p;
  if (idx_out) idx_out[i] = src;
  if (a_out) {
    const float* s = a + src * da;
    float* d = a_out + i * da;
    for (int k = 0; k < da; ++k) d[k] = s[k];
  }
  if (b_out) {
    const float* s = b + src * db;
    float* d = b_out + i * db;
    for (int k = 0; k < db; ++k) d[k] = s[k];
  }
}

static int shuffled_gather_launch(const float* a, int32_t da, const float* b, int32_t db, const int64_t* base_idx,
                                  int64_t n_perm, uint64_t key, int64_t offset, int64_t count,
                                  float* a_out, float* b_out, int64_t* idx_out, void* stream) {
  if (n_perm < 1 || n_perm > 0x7fffffffll || offset < 0 || count < 0 || offset + count > n_perm)
    return SBI_AMD_E_BADARG;
  if ((a_out && (!a || da < 1)) || (b_out && (!b || db < 1))) return SBI_AMD_E_BADARG;
  if (count == 0) return 0;
  int bits = 2;
  while (bits < 32 && (1ll << bits) < n_perm) ++bits;
  const int hb = (bits + 1) / 2;
  hipLaunchKernelGGL(shuffled_gather_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     a, da, b, db, (const long long*)base_idx, (unsigned)n_perm, hb, (unsigned long long)key,
                     (long long)offset, (long long)count, a_out, b_out, (long long*)idx_out);
  return (int)hipGetLastError();
}

extern "C" int sbi_amd_shuffled_gather(const float* a, int32_t da, const float* b, int32_t db, const int64_t* base_idx,
                                       int64_t n_perm, uint64_t key, int64_t offset, int64_t count, float* a_out,
                                       float* b_out, int64_t* idx_out, void* stream) {
  return shuffled_gather_launch(a, da, b, db, base_idx, n_perm, key, offset, count, a_out, b_out, idx_out, stream);
}
