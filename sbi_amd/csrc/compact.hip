// compact.hip -- stable stream compaction of accepted proposal draws: the loop body of accept_reject_sample
// (sbi/samplers/rejection/rejection.py:368-409: `candidates[are_accepted]` appended per condition, running counts for
// the batch-size rule) as ONE launch and one small read-back per iteration.
//   * acceptance: either a caller-provided mask (any prior: torch evaluates `within_support`) or, for a box prior,
//     lo <= theta <= hi evaluated here on the candidates as they are read (NaN fails, as torch's interval check does);
//   * order-preserving: row r of condition x goes to out[filled[x] + #accepted rows before r]  (single-pass scan with
//     decoupled look-back: one 8-byte {generation, flag, count} word per tile, agent-scope relaxed accesses -- the word
//     is its own flag, nothing else is published between workgroups);
//   * rows past num_samples are dropped; the last workgroup to finish folds the totals into `state`.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sbi_amd_nsf.h"

#define CP_THREADS 256
#define CP_RPT 4                          // consecutive rows per thread (keeps the order: thread-major)
#define CP_TILE (CP_THREADS * CP_RPT)

__device__ __forceinline__ unsigned long long cp_word(unsigned gen, unsigned flag, unsigned val) {
  return ((unsigned long long)(gen & 0x3fffffffu) << 34) | ((unsigned long long)flag << 32) | val;
}

// Decoupled look-back by the WHOLE WORKGROUP (every thread calls it): publish this tile's aggregate, then read
// CP_THREADS predecessors at a time -- thread t the word of tile (i - t) -- until a window holds an inclusive prefix;
// the nearest one and the aggregates in front of it are the tile's exclusive prefix.  (A single thread walking back one
// word per L2 round trip made the launch latency-bound: with ~770 tiles resident at once a late tile walked hundreds
// of words -- 186 us per 10^6 rows however few bytes moved; a wave-wide window still chains ~15 round trips.)
// Returns the exclusive prefix in every thread.
__device__ __forceinline__ long long cp_lookback(unsigned long long* __restrict__ st, int tile, unsigned gen,
                                                 int tile_total) {
  __shared__ long long s_sum[CP_THREADS / 64];
  __shared__ int s_hit[CP_THREADS / 64];
  __shared__ long long s_res;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long long excl = 0;
  if (tile > 0) {
    if (tid == 0)
      __hip_atomic_store(&st[tile], cp_word(gen, 1, (unsigned)tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = tile - 1;; i -= CP_THREADS) {      // (workgroup-uniform loop: the exit test reads shared memory)
      const int idx = i - tid;
      unsigned flag = 2, val = 0;               // (before tile 0: an inclusive prefix of nothing)
      if (idx >= 0) {
        unsigned long long w;
        do {
          w = __hip_atomic_load(&st[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((w >> 34) == (gen & 0x3fffffffu) && ((w >> 32) & 3u) != 0) break;
          __builtin_amdgcn_s_sleep(1);
        } while (true);
        flag = (unsigned)(w >> 32) & 3u;
        val = (unsigned)(w & 0xffffffffu);
      }
      const unsigned long long incl = __ballot(flag == 2);
      const int first = incl ? (int)__builtin_ctzll(incl) : 64;      // nearest inclusive prefix of this wave's window
      long long v = lane <= first ? (long long)val : 0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) { s_sum[wave] = v; s_hit[wave] = incl ? 1 : 0; }
      __syncthreads();
      bool hit = false;
#pragma unroll
      for (int w = 0; w < CP_THREADS / 64; ++w)
        if (!hit) { excl += s_sum[w]; hit = s_hit[w] != 0; }
      __syncthreads();
      if (hit) break;
    }
  }
  if (tid == 0)
    __hip_atomic_store(&st[tile], cp_word(gen, 2, (unsigned)(excl + tile_total)), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

// state (int64, per condition x): [0, X) filled, [X, 2X) total accepted so far, [2X, 3X) accepted by this call
// ctl (int32, per condition): [0, X) tile tickets, [X, 2X) finished tiles -- both left at 0 by the last workgroup
__global__ void __launch_bounds__(CP_THREADS)
accept_compact_kernel(const float* __restrict__ cand, const unsigned char* __restrict__ mask,
                      const float* __restrict__ lo, const float* __restrict__ hi, long long bs, int num_xos, int ev,
                      float* __restrict__ out, long long num_samples, long long* __restrict__ state,
                      int* __restrict__ ctl, unsigned long long* __restrict__ scan, int ntiles, unsigned gen) {
  const int xo = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int s_tile;
  __shared__ int s_wave[CP_THREADS / 64];
  __shared__ long long s_excl;
  if (tid == 0) s_tile = __hip_atomic_fetch_add(&ctl[xo], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int tile = s_tile;                 // tiles are taken in dispatch order: a predecessor is always running or done
  const long long row0 = (long long)tile * CP_TILE + (long long)tid * CP_RPT;
  // ---- acceptance of this thread's rows
  bool acc[CP_RPT];
  int cnt = 0;
#pragma unroll
  for (int u = 0; u < CP_RPT; ++u) {
    const long long r = row0 + u;
    bool a = false;
    if (r < bs) {
      if (mask) {
        a = mask[r * num_xos + xo] != 0;
      } else {
        const float* c = cand + (r * num_xos + xo) * ev;
        a = true;
        for (int d = 0; d < ev; ++d) a = a && (c[d] >= lo[d]) && (c[d] <= hi[d]);
      }
    }
    acc[u] = a;
    cnt += a ? 1 : 0;
  }
  // ---- exclusive scan of the per-thread counts inside the tile
  int incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int wave_base = 0, tile_total = 0;
#pragma unroll
  for (int w = 0; w < CP_THREADS / 64; ++w) {
    if (w < wave) wave_base += s_wave[w];
    tile_total += s_wave[w];
  }
  const int thread_excl = wave_base + incl - cnt;
  // ---- decoupled look-back over the tiles of this condition
  unsigned long long* st = scan + (long long)xo * ntiles;
  {
    const long long excl = cp_lookback(st, tile, gen, tile_total);
    if (tid == 0) s_excl = excl;
  }
  __syncthreads();
  // ---- scatter (stable), dropping rows past the request
  long long dest = state[xo] + s_excl + thread_excl;
#pragma unroll
  for (int u = 0; u < CP_RPT; ++u) {
    if (acc[u]) {
      if (dest < num_samples) {
        const float* c = cand + ((row0 + u) * num_xos + xo) * ev;
        float* o = out + (dest * num_xos + xo) * ev;
        for (int d = 0; d < ev; ++d) o[d] = c[d];
      }
      ++dest;
    }
  }
  // ---- the last workgroup of this condition folds the totals in (everybody has read state[xo] by then)
  __syncthreads();
  if (tid == 0) {
    // (relaxed: an agent-scope release would write this workgroup's 40 KB of output back from L2 before the counter moves,
    //  one workgroup after the other.  What the last workgroup needs is only that every other one has READ state[xo] --
    //  each has consumed the value in its scatter above -- and the prefix word it reads is itself an agent-scope atomic.)
    const int done = __hip_atomic_fetch_add(&ctl[num_xos + xo], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == ntiles - 1) {
      const unsigned long long w = __hip_atomic_load(&st[ntiles - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long total = (long long)(w & 0xffffffffu);
      const long long f = state[xo] + total;
      state[2 * num_xos + xo] = total;
      state[num_xos + xo] += total;
      state[xo] = f < num_samples ? f : num_samples;
      ctl[xo] = 0;
      ctl[num_xos + xo] = 0;
    }
  }
}

// ---- one condition (num_xos == 1), event_floats <= CP_EV_MAX: the same compaction with every global access coalesced.
// A tile's candidates are ONE contiguous span (staged in LDS with 16-byte loads), and so are its accepted rows in the
// output (the compaction is stable): the workgroup writes them as a span, lane = (row of the step, float of the row).
// Acceptance and ranks per 256-row sub-tile from wave ballots.  (The kernel above reads and writes rows per thread:
// 40-byte pieces at a 160-byte lane stride -- 186 us per 10^6 ten-float rows against 25 here.)
#define CP_EV_MAX 32
__global__ void __launch_bounds__(CP_THREADS)
accept_compact_rows_kernel(const float* __restrict__ cand, const unsigned char* __restrict__ mask,
                           const float* __restrict__ lo, const float* __restrict__ hi, long long bs, int ev,
                           float* __restrict__ out, long long num_samples, long long* __restrict__ state,
                           int* __restrict__ ctl, unsigned long long* __restrict__ scan, int ntiles, unsigned gen) {
  extern __shared__ float4 cp_lds4[];
  float* A = reinterpret_cast<float*>(cp_lds4);                  // CP_TILE x ev candidates
  int* src = reinterpret_cast<int*>(A + (long long)CP_TILE * ev);   // rank inside the tile -> row inside the tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int s_tile;
  __shared__ int s_cnt[CP_RPT][CP_THREADS / 64];
  __shared__ long long s_excl;
  if (tid == 0) s_tile = __hip_atomic_fetch_add(&ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int tile = s_tile;
  const long long row0 = (long long)tile * CP_TILE;
  const int rows = (int)((bs - row0) < CP_TILE ? (bs - row0) : CP_TILE);
  // ---- stage the tile
  {
    const float* g = cand + row0 * ev;
    const int nfl = rows * ev;
    if (rows == CP_TILE) {            // (CP_TILE * ev floats from a 16-byte aligned base: whole float4s)
      const float4* g4 = reinterpret_cast<const float4*>(g);
      for (int i = tid; i < nfl / 4; i += CP_THREADS) cp_lds4[i] = g4[i];
    } else {
      for (int i = tid; i < nfl; i += CP_THREADS) A[i] = g[i];
    }
  }
  __syncthreads();
  // ---- acceptance: sub-tile u = rows [256 u, 256 u + 256), thread = row
  bool acc[CP_RPT];
  int below[CP_RPT];            // accepted rows of the same wave before this lane
#pragma unroll
  for (int u = 0; u < CP_RPT; ++u) {
    const int r = u * CP_THREADS + tid;
    bool a = false;
    if (r < rows) {
      if (mask) {
        a = mask[row0 + r] != 0;
      } else {
        const float* c = A + r * ev;
        a = true;
        for (int d = 0; d < ev; ++d) a = a && (c[d] >= lo[d]) && (c[d] <= hi[d]);
      }
    }
    acc[u] = a;
    const unsigned long long b = __ballot(a);
    below[u] = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[u][wave] = __popcll(b);
  }
  __syncthreads();
  int tile_total = 0;
  int base[CP_RPT];
#pragma unroll
  for (int u = 0; u < CP_RPT; ++u) {
    base[u] = tile_total;
#pragma unroll
    for (int w = 0; w < CP_THREADS / 64; ++w) {
      if (w < wave) base[u] += s_cnt[u][w];
      tile_total += s_cnt[u][w];
    }
  }
#pragma unroll
  for (int u = 0; u < CP_RPT; ++u)
    if (acc[u]) src[base[u] + below[u]] = u * CP_THREADS + tid;
  // ---- decoupled look-back (as above)
  {
    const long long excl = cp_lookback(scan, tile, gen, tile_total);
    if (tid == 0) s_excl = excl;
  }
  __syncthreads();
  // ---- write the tile's accepted rows as one span, dropping rows past the request
  {
    const long long dest0 = state[0] + s_excl;
    long long room = num_samples - dest0;
    room = room < 0 ? 0 : room;
    const int nout = (long long)tile_total < room ? tile_total : (int)room;
    const int rps = 64 / ev;                       // rows per wave step
    const int lr = lane / ev, d = lane - lr * ev;
    if (lr < rps) {
      float* o = out + dest0 * ev;
      for (int r = wave * rps + lr; r < nout; r += (CP_THREADS / 64) * rps) o[(long long)r * ev + d] = A[src[r] * ev + d];
    }
  }
  // ---- the last workgroup folds the totals in
  __syncthreads();
  if (tid == 0) {
    const int done = __hip_atomic_fetch_add(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (as above)
    if (done == ntiles - 1) {
      const unsigned long long w = __hip_atomic_load(&scan[ntiles - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long total = (long long)(w & 0xffffffffu);
      const long long f = state[0] + total;
      state[2] = total;
      state[1] += total;
      state[0] = f < num_samples ? f : num_samples;
      ctl[0] = 0;
      ctl[1] = 0;
    }
  }
}

extern "C" int64_t sbi_amd_accept_compact_scan_words(int64_t batch_rows, int32_t num_xos) {
  if (batch_rows < 0 || num_xos < 1) return SBI_AMD_E_BADARG;
  return ((batch_rows + CP_TILE - 1) / CP_TILE) * (int64_t)num_xos;
}

extern "C" int sbi_amd_accept_compact(const float* candidates, const uint8_t* accepted, const float* box_low,
                                      const float* box_high, int64_t batch_rows, int32_t num_xos, int32_t event_floats,
                                      float* out, int64_t num_samples, int64_t* state, int32_t* control,
                                      uint64_t* scan, uint32_t generation, void* stream) {
  if (!candidates || !out || !state || !control || !scan || num_xos < 1 || event_floats < 1 || batch_rows < 0 ||
      num_samples < 0 || batch_rows >= (1ll << 31) || (!accepted && (!box_low || !box_high)) || generation == 0)
    return SBI_AMD_E_BADARG;
  if (batch_rows == 0) return 0;
  const int ntiles = (int)((batch_rows + CP_TILE - 1) / CP_TILE);
  if (num_xos == 1 && event_floats <= CP_EV_MAX && ((uintptr_t)candidates & 15) == 0) {
    const size_t lds = ((size_t)CP_TILE * event_floats + CP_TILE) * sizeof(float);
    // (more than 64 KiB of dynamic LDS has to be asked for once: a function-local static is initialised exactly once,
    //  whichever thread gets there first)
    static const hipError_t attr_rc =
        hipFuncSetAttribute((const void*)accept_compact_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(((size_t)CP_TILE * CP_EV_MAX + CP_TILE) * sizeof(float)));
    if (attr_rc != hipSuccess) return SBI_AMD_E_UNSUPPORTED;
    hipLaunchKernelGGL(accept_compact_rows_kernel, dim3(ntiles), dim3(CP_THREADS), lds, (hipStream_t)stream, candidates,
                       accepted, box_low, box_high, (long long)batch_rows, event_floats, out, (long long)num_samples,
                       (long long*)state, control, (unsigned long long*)scan, ntiles, generation);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(accept_compact_kernel, dim3(ntiles, num_xos), dim3(CP_THREADS), 0, (hipStream_t)stream, candidates,
                     accepted, box_low, box_high, (long long)batch_rows, num_xos, event_floats, out,
                     (long long)num_samples, (long long*)state, control, (unsigned long long*)scan, ntiles, generation);
  return (int)hipGetLastError();
}
