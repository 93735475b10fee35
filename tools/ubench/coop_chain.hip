// Micro-benchmark for the small-batch floor ("Small batches"): a chain of L hidden layers
// 64 -> 64 on ONE 16-row tile, fp32 MFMA 16x16x4, weights in LDS (stride 68, as the flow kernels hold them).
//   mode 0  one wave walks the chain alone: 4 m-tiles x 16 K-steps per layer, the D fragments of a layer are the B
//           fragments of the next (what nsf_flow_kernel does today: ~23 us per transform at batch 200);
//   mode 1  four waves share the tile: wave w owns m-tile w (16 K-steps per layer), publishes its D fragment in LDS
//           (one ds_write_b128 per lane), workgroup barrier, reads the other three (3 x ds_read_b128) as B operands.
// Prints cycles per layer of the slowest wave; one workgroup per CU-sized grid so that nothing else interferes.
// build: hipcc --offload-arch=gfx950 -O3 -o coop_chain coop_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define SW 68

__global__ void __launch_bounds__(256) chain(float* out, long long* cyc, int layers, int iters, int mode) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* W = lds;                      // 64 x SW
  float* X = lds + 64 * SW;            // exchange tile: 4 blocks x 64 lanes x 4 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  for (int i = tid; i < 64 * SW; i += blockDim.x) W[i] = 1e-2f * ((i * 7) % 13 - 6);
  __syncthreads();
  f4 h[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) h[b] = f4{0.01f * lane, 0.02f, 0.03f * b, 0.04f};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    for (int l = 0; l < layers; ++l) {
      if (mode == 0) {
        if (wave == 0) {
          f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = 16 * kb + 4 * r + g;             // feature held by register r of k-slot g in block kb
#pragma unroll
              for (int mt = 0; mt < 4; ++mt) acc[mt] = MFMA16(W[(16 * mt + j) * SW + k], h[kb][r], acc[mt]);
            }
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[mt][r] = fmaxf(acc[mt][r], 0.f) * 0.05f;
        }
      } else {
        f4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = 16 * kb + 4 * r + g;
            acc = MFMA16(W[(16 * wave + j) * SW + k], h[kb][r], acc);
          }
        f4 mine;
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[r] = fmaxf(acc[r], 0.f) * 0.05f;
        *reinterpret_cast<f4*>(X + (wave * 64 + lane) * 4) = mine;
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 4; ++b) h[b] = *reinterpret_cast<const f4*>(X + (b * 64 + lane) * 4);
        __syncthreads();          // the tile is reused by the next layer
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < 4; ++b) s += h[b][0] + h[b][1] + h[b][2] + h[b][3];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

int main() {
  float* out;
  long long* cyc;
  const int grid = 256, layers = 8, iters = 200;
  hipMalloc(&out, grid * 256 * 4);
  hipMalloc(&cyc, grid * 4 * 8);
  const int lds = (64 * SW + 4 * 64 * 4) * 4;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(cyc, 0, grid * 4 * 8);
      hipLaunchKernelGGL(chain, dim3(grid), dim3(256), lds, 0, out, cyc, layers, iters, mode);
      hipDeviceSynchronize();
    }
    long long h[grid * 4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    long long worst = 0;
    for (int i = 0; i < grid * 4; ++i) worst = h[i] > worst ? h[i] : worst;
    printf("mode %d (%s): %.0f cycles (readcyclecounter ticks) per layer, %d MFMAs per wave and layer\n", mode,
           mode == 0 ? "one wave, D->B chaining" : "four waves, LDS exchange per layer",
           (double)worst / (layers * iters), mode == 0 ? 64 : 16);
  }
  return 0;
}
