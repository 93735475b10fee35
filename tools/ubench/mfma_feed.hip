// Micro-benchmark: how fast can a wave feed v_mfma_f32_16x16x4_f32 from LDS on gfx950?
// 8 waves per workgroup (2 per SIMD), 256 workgroups.  Variants (cycles per MFMA per wave are printed;
// with two MFMA waves per SIMD the matrix pipe is saturated at 64 cycles / MFMA / wave, with one at 32):
//   0: operands in registers                         1: one ds_read_b32 (A) per MFMA, B in registers
//   2: 1 A + 4 B ds_read_b32 per 4 MFMAs             3: 1 A b32 + 1 B ds_read_b128 per 4 MFMAs
// MODE bit 4 (0x10): a workgroup barrier after every 64-MFMA call (short GEMM calls as in the backward kernel)
// MODE bit 8: only waves 4..7 run MFMAs, waves 0..3 run a VALU loop; bit 9: ... an LDS-heavy VALU loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int V>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters, int mode) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  for (int i = tid; i < 32768; i += 512) lds[i] = 1e-3f * (i & 255);
  __syncthreads();
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float r = 0.f;
  long long t0 = __builtin_readcyclecounter();
  const bool split = mode & 0x300;
  if (split && wave < 4) {
    if (mode & 0x200) {   // LDS-heavy VALU (spline-like): read, fma, write
      float* p = lds + 16384 + wave * 2048 + lane * 17;
      float v = lane;
      for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { v = fmaf(v, 1.0001f, p[u]); p[(u + 3) & 7] = v; }
        if ((mode & 0x10) && (it & 3) == 3) __syncthreads();
      }
      r = v;
    } else {
      float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
      for (int it = 0; it < iters * 16; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { v0 = fmaf(v0, 1.0001f, v1); v1 = fmaf(v1, 0.9999f, v2); v2 = fmaf(v2, 1.0002f, v3); v3 = fmaf(v3, 0.9998f, v0); }
        if ((mode & 0x10) && (it & 15) == 15) __syncthreads();
      }
      r = v0 + v1 + v2 + v3;
    }
  } else {
    const float* ap = lds + g * 68 + 16 * (wave & 3) + j;          // A tile stride 68 (as the backward kernel)
    const float* bp = lds + 8192 + g * 80 + j;                     // B tile stride 80
    const float* bq = lds + 8192 + g * 68 + 4 * j;                 // interleaved B tile, stride 68
    for (int it = 0; it < iters; ++it) {
      if (mode & 0x10) __syncthreads();
      if (V == 0) {
        float a = lane * 1e-3f + it, b = j;
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA16(a, b + nt, acc[nt]);
      } else if (V == 1) {
        float b = j;
        float a_cur[4], a_nxt[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a_cur[m] = ap[m * 16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          if (s + 1 < 16)
#pragma unroll
            for (int m = 0; m < 4; ++m) a_nxt[m] = ap[(s + 1) * 200 + m * 16];
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[m] = MFMA16(a_cur[m], b, acc[m]);
#pragma unroll
          for (int m = 0; m < 4; ++m) a_cur[m] = a_nxt[m];
        }
      } else if (V == 2) {
        float a[3], b[3][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          a[u] = ap[4 * u * 68];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) b[u][nt] = bp[4 * u * 80 + 16 * nt];
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          if (s + 2 < 16) {
            a[(s + 2) % 3] = ap[4 * (s + 2) * 68];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) b[(s + 2) % 3][nt] = bp[4 * (s + 2) * 80 + 16 * nt];
          }
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA16(a[s % 3], b[s % 3][nt], acc[nt]);
        }
      } else {
        float a[3];
        f4 b[3];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          a[u] = ap[4 * u * 68];
          b[u] = *(const f4*)(bq + 4 * u * 68);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          if (s + 2 < 16) {
            a[(s + 2) % 3] = ap[4 * (s + 2) * 68];
            b[(s + 2) % 3] = *(const f4*)(bq + 4 * (s + 2) * 68);
          }
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA16(a[s % 3], b[s % 3][nt], acc[nt]);
        }
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float sum = r;
  for (int nt = 0; nt < 4; ++nt) sum += acc[nt][0] + acc[nt][1] + acc[nt][2] + acc[nt][3];
  out[blockIdx.x * 512 + tid] = sum;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int V>
void run(int mode, int iters) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<V>, dim3(256), dim3(512), 131072, 0, out, cyc, iters, mode);
  hipDeviceSynchronize();
  long long h[2048];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mf = 0, va = 0; int nm = 0, nv = 0;
  for (int i = 0; i < 2048; ++i) { if ((mode & 0x300) && (i & 7) < 4) { va += h[i]; ++nv; } else { mf += h[i]; ++nm; } }
  printf("variant %d mode 0x%x: %.1f cycles/MFMA/wave (MFMA waves)", V, mode, mf / nm / (iters * 64.0));
  if (nv) printf(", other waves %.0f cycles total vs MFMA waves %.0f", va / nv, mf / nm);
  printf("\n");
  hipFree(out); hipFree(cyc);
}

int main() {
  const int iters = 200;
  for (int mode : {0, 0x100, 0x200, 0x10, 0x110, 0x210}) { run<0>(mode, iters); run<1>(mode, iters); run<2>(mode, iters); run<3>(mode, iters); }
  return 0;
}
