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

// splitmix64 of (seed, epoch): the 64-bit Feistel key of one epoch's order (sbi_amd/utils/shuffle.py::epoch_key)
__host__ __device__ __forceinline__ unsigned long long shf_epoch_key(unsigned long long seed, unsigned long long epoch) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (epoch + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// epoch_dev != NULL: `key` is the SEED and the epoch number is read from device memory (the launch can then sit in a
// captured HIP graph that is replayed every epoch: sbi_amd_shuffled_gather_clock)
__global__ void __launch_bounds__(256)
shuffled_gather_kernel(const float* __restrict__ a, int da, const float* __restrict__ b, int db,
                       const long long* __restrict__ base_idx, unsigned n_perm, int hb, unsigned long long key,
                       const long long* __restrict__ epoch_dev,
                       long long offset, long long count, float* __restrict__ a_out, float* __restrict__ b_out,
                       long long* __restrict__ idx_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  if (epoch_dev) key = shf_epoch_key(key, (unsigned long long)*epoch_dev);
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
                                  int64_t n_perm, uint64_t key, const int64_t* epoch_dev, int64_t offset, int64_t count,
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
                     (const long long*)epoch_dev, (long long)offset, (long long)count, a_out, b_out, (long long*)idx_out);
  return (int)hipGetLastError();
}

extern "C" int sbi_amd_shuffled_gather(const float* a, int32_t da, const float* b, int32_t db, const int64_t* base_idx,
                                       int64_t n_perm, uint64_t key, int64_t offset, int64_t count, float* a_out,
                                       float* b_out, int64_t* idx_out, void* stream) {
  return shuffled_gather_launch(a, da, b, db, base_idx, n_perm, key, nullptr, offset, count, a_out, b_out, idx_out, stream);
}

// The same gather with the epoch number in DEVICE memory: key = splitmix64(seed, *epoch_dev), evaluated by the kernel.
extern "C" int sbi_amd_shuffled_gather_clock(const float* a, int32_t da, const float* b, int32_t db,
                                             const int64_t* base_idx, int64_t n_perm, uint64_t seed,
                                             const int64_t* epoch_dev, int64_t offset, int64_t count, float* a_out,
                                             float* b_out, int64_t* idx_out, void* stream) {
  if (!epoch_dev) return SBI_AMD_E_BADARG;
  return shuffled_gather_launch(a, da, b, db, base_idx, n_perm, seed, epoch_dev, offset, count, a_out, b_out, idx_out,
                                stream);
}

// clock[0] = epoch number, clock[1] = optimizer step count (int64, device); bias_corr[0] = 1 - beta1^step,
// bias_corr[1] = sqrt(1 - beta2^step) (fp32, device): what a captured epoch advances by itself.
// which = 0: the epoch; which = 1: the optimizer step and its bias corrections.
__global__ void train_clock_tick_kernel(long long* __restrict__ clock, float* __restrict__ bias_corr, int which,
                                        double beta1, double beta2) {
  if (which == 0) {
    clock[0] += 1;
    return;
  }
  const long long step = clock[1] + 1;
  clock[1] = step;
  bias_corr[0] = (float)(1.0 - pow(beta1, (double)step));
  bias_corr[1] = (float)sqrt(1.0 - pow(beta2, (double)step));
}

extern "C" int sbi_amd_train_clock_tick(int64_t* clock, float* bias_corr, int32_t which, float beta1, float beta2,
                                        void* stream) {
  if (!clock || (which != 0 && which != 1) || (which == 1 && !bias_corr)) return SBI_AMD_E_BADARG;
  hipLaunchKernelGGL(train_clock_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)clock, bias_corr,
                     (int)which, (double)beta1, (double)beta2);
  return (int)hipGetLastError();
}
