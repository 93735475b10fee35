// Micro-benchmark: do VALU instructions overlap with v_mfma_f32_16x16x4_f32 on gfx950?
//   mode 0: MFMAs only                 (4 independent accumulators, back to back)
//   mode 1: VALU only                  (V independent v_fma_f32 chains per MFMA slot)
//   mode 2: one wave, interleaved      (each MFMA followed by V FMAs, order pinned with sched_barrier)
//   mode 3: two waves per SIMD         (waves 0..3 run the MFMA stream, waves 4..7 the VALU stream)
// Prints cycles per slot (1 MFMA and/or V FMAs).  Overlap => mode 2/3 ~ max(mode 0, mode 1); none => sum.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip ; run: ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
constexpr int V = 6;

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane + i;
  const float a = lane * 1e-3f, b = 1.0f + lane * 1e-4f;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (do_m && do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc[s & 3] = MFMA16(a, b, acc[s & 3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < V; ++u) v[u] = fmaf(v[u], 1.0001f, b);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (do_m) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) acc[s & 3] = MFMA16(a, b, acc[s & 3]);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
#pragma unroll
        for (int u = 0; u < V; ++u) v[u] = fmaf(v[u], 1.0001f, b);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += v[i];
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][3];
  out[blockIdx.x * 512 + tid] = r;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 10);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s", name);
  for (int w = 0; w < threads / 64; ++w) printf(" w%d %.1f", w, (double)h[w] / (iters * 16.0));
  printf("   cycles per slot\n");
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("0 MFMA only, 1 wave/SIMD", 256);
  run<1>("1 VALU only (6 fma/slot), 1 wave/SIMD", 256);
  run<2>("2 interleaved in one wave, 1 wave/SIMD", 256);
  run<3>("3 MFMA waves 0-3 + VALU waves 4-7", 512);
  run<0>("0b MFMA only, 2 waves/SIMD", 512);
  run<1>("1b VALU only, 2 waves/SIMD", 512);
  return 0;
}
