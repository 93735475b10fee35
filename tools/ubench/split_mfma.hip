// Micro-benchmark: the hidden-layer GEMM chain of the conditioner (H = 64, 16 rows per wave,
// weights in LDS, D fragment of a layer = B fragment of the next) on three matrix paths:
//   mode 0: v_mfma_f32_16x16x4_f32          64 MFMAs / layer (what the kernels use today)
//   mode 1: f16 two-way split, 3 products   24 x v_mfma_f32_16x16x32_f16 / layer   (x = xh + xl/2048, drop xl*wl)
//   mode 2: bf16 three-way split, 6 products 48 x v_mfma_f32_16x16x32_bf16 / layer  (drop terms below 2^-24)
// Prints cycles per layer per wave (8 waves per workgroup = 2 per SIMD, as the flow kernels run) and the error of
// the L-layer chain  h <- relu(W h)  against an fp64 host evaluation of the same weights / inputs.
// build: hipcc --offload-arch=gfx950 -O3 -o split_mfma split_mfma.hip ; run: ./split_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int H = 64, L = 4, ROWS = 16;

// ---- fp32 path ---------------------------------------------------------------------------------------------
// image: [layer][mt][s = 4 mt' + r][lane] = W[16 mt + (lane & 15)][16 mt' + 4 (lane >> 4) + r]
__device__ __forceinline__ void layer_f32(const float* __restrict__ img, int lane, f4 (&h)[4]) {
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float bv = h[s >> 2][s & 3];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(img[(mt * 16 + s) * 64 + lane], bv, acc[mt], 0, 0, 0);
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[mt][r] = fmaxf(acc[mt][r], 0.f);
}

// ---- f16 split path ----------------------------------------------------------------------------------------
// image: [layer][part (hi, lo)][mt][p][lane] = 8 halves: W[16 mt + (lane & 15)][16 (2p + (e >> 2)) + 4 (lane >> 4) + (e & 3)]
__device__ __forceinline__ unsigned pk_f16_rtz(float a, float b) {
  typedef __fp16 hh2 __attribute__((ext_vector_type(2)));
  const hh2 v = __builtin_amdgcn_cvt_pkrtz(a, b);
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float f16_lo_to_f32(unsigned pk) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(pk & 0xffffu));
}
__device__ __forceinline__ float f16_hi_to_f32(unsigned pk) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(pk >> 16));
}
__device__ __forceinline__ void layer_f16(const u4* __restrict__ img, int lane, f4 (&h)[4]) {
  f4 am[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // xh * wh
  f4 ac[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // (xh * wl' + xl' * wh), lo parts scaled by 2048
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    u4 xh, xl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {   // halves 2q, 2q+1 of the K-step
      const float a = h[2 * p + (q >> 1)][2 * (q & 1)], b = h[2 * p + (q >> 1)][2 * (q & 1) + 1];
      const unsigned hi = pk_f16_rtz(a, b);
      const float la = (a - f16_lo_to_f32(hi)) * 2048.f, lb = (b - f16_hi_to_f32(hi)) * 2048.f;   // exact
      xh[q] = hi;
      xl[q] = pk_f16_rtz(la, lb);
    }
    const h8 bh = __builtin_bit_cast(h8, xh), bl = __builtin_bit_cast(h8, xl);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const h8 wh = __builtin_bit_cast(h8, img[((0 * 4 + mt) * 2 + p) * 64 + lane]);
      const h8 wl = __builtin_bit_cast(h8, img[((1 * 4 + mt) * 2 + p) * 64 + lane]);
      am[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh, am[mt], 0, 0, 0);
      ac[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh, ac[mt], 0, 0, 0);
      ac[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl, ac[mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[mt][r] = fmaxf(fmaf(ac[mt][r], 1.f / 2048.f, am[mt][r]), 0.f);
}

// ---- bf16 three-way split --------------------------------------------------------------------------------
// image: [layer][part (hi, mid, lo)][mt][p][lane] = 8 bf16, same element order as the f16 image
__device__ __forceinline__ unsigned pk_bf16_trunc(float a, float b) {   // truncation: the remainder is exact
  return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
}
__device__ __forceinline__ float trunc_bf16(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }
__device__ __forceinline__ void layer_bf16(const u4* __restrict__ img, int lane, f4 (&h)[4]) {
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f4 acs[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // small terms, added last
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    u4 x0, x1, x2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float a = h[2 * p + (q >> 1)][2 * (q & 1)], b = h[2 * p + (q >> 1)][2 * (q & 1) + 1];
      const float a0 = trunc_bf16(a), b0 = trunc_bf16(b);
      const float ra = a - a0, rb = b - b0;
      const float a1 = trunc_bf16(ra), b1 = trunc_bf16(rb);
      x0[q] = pk_bf16_trunc(a0, b0);
      x1[q] = pk_bf16_trunc(a1, b1);
      x2[q] = pk_bf16_trunc(ra - a1, rb - b1);
    }
    const b8 bx0 = __builtin_bit_cast(b8, x0), bx1 = __builtin_bit_cast(b8, x1), bx2 = __builtin_bit_cast(b8, x2);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const b8 w0 = __builtin_bit_cast(b8, img[((0 * 4 + mt) * 2 + p) * 64 + lane]);
      const b8 w1 = __builtin_bit_cast(b8, img[((1 * 4 + mt) * 2 + p) * 64 + lane]);
      const b8 w2 = __builtin_bit_cast(b8, img[((2 * 4 + mt) * 2 + p) * 64 + lane]);
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, bx0, acc[mt], 0, 0, 0);
      acs[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, bx1, acs[mt], 0, 0, 0);
      acs[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, bx0, acs[mt], 0, 0, 0);
      acs[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, bx1, acs[mt], 0, 0, 0);
      acs[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, bx2, acs[mt], 0, 0, 0);
      acs[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, bx0, acs[mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[mt][r] = fmaxf(acc[mt][r] + acs[mt][r], 0.f);
}

template <int MODE>
__global__ void __launch_bounds__(512) chain(const float* __restrict__ x, const void* __restrict__ wimg, int img_bytes,
                                             float* __restrict__ out, long long* __restrict__ cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < img_bytes / 16; i += blockDim.x) ((u4*)smem)[i] = ((const u4*)wimg)[i];
  __syncthreads();
  const int j = lane & 15, g = lane >> 4;
  const long long row = ((long long)blockIdx.x * 8 + wave) * ROWS + j;
  f4 h0[4], h[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) h0[mt][r] = x[row * H + 16 * mt + 4 * g + r];
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) h[mt] = h0[mt];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      if (MODE == 0) layer_f32((const float*)smem + l * H * H, lane, h);
      else if (MODE == 1) layer_f16((const u4*)smem + l * 2 * 4 * 2 * 64, lane, h);
      else layer_bf16((const u4*)smem + l * 3 * 4 * 2 * 64, lane, h);
    }
    // keep the iterations dependent so that nothing is hoisted
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) h0[mt][r] += 0.f * h[mt][r];
  }
  const long long t1 = __builtin_readcyclecounter();
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[row * H + 16 * mt + 4 * g + r] = h[mt][r];
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

static unsigned short f2h_rtz(float f) {   // float -> f16 bits, round toward zero (normal range only)
  _Float16 h = (_Float16)f;
  if (fabsf((float)h) > fabsf(f)) {        // rounded away from zero: step back one ulp
    unsigned short b = __builtin_bit_cast(unsigned short, h);
    b -= 1;
    return b;
  }
  return __builtin_bit_cast(unsigned short, h);
}
static unsigned short f2h_rne(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
static unsigned short f2bf_trunc(float f) { return (unsigned short)(__builtin_bit_cast(unsigned, f) >> 16); }
static float bf2f(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

int main() {
  const int nblocks = 256, n = nblocks * 8 * ROWS, iters = 200;
  std::vector<float> W(L * H * H), X((size_t)n * H);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& w : W) w = rnd() * 0.35f;   // ~ kaiming-uniform scale for fan-in 64 x relu gain
  for (auto& v : X) v = rnd() * 2.f;
  // fp64 reference
  std::vector<double> ref((size_t)n * H);
  for (int r = 0; r < n; ++r) {
    double a[H], b[H];
    for (int k = 0; k < H; ++k) a[k] = X[(size_t)r * H + k];
    for (int l = 0; l < L; ++l) {
      for (int o = 0; o < H; ++o) {
        double s = 0;
        for (int k = 0; k < H; ++k) s += (double)W[(l * H + o) * H + k] * a[k];
        b[o] = s > 0 ? s : 0;
      }
      for (int k = 0; k < H; ++k) a[k] = b[k];
    }
    for (int k = 0; k < H; ++k) ref[(size_t)r * H + k] = a[k];
  }
  // weight images
  std::vector<float> img32(L * H * H);
  std::vector<unsigned short> img16((size_t)L * 2 * 4 * 2 * 64 * 8), imgbf((size_t)L * 3 * 4 * 2 * 64 * 8);
  for (int l = 0; l < L; ++l)
    for (int mt = 0; mt < 4; ++mt)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, g = lane >> 4;
        for (int s = 0; s < 16; ++s)
          img32[((l * 4 + mt) * 16 + s) * 64 + lane] = W[(l * H + 16 * mt + i) * H + 16 * (s >> 2) + 4 * g + (s & 3)];
        for (int p = 0; p < 2; ++p)
          for (int e = 0; e < 8; ++e) {
            const float w = W[(l * H + 16 * mt + i) * H + 16 * (2 * p + (e >> 2)) + 4 * g + (e & 3)];
            const unsigned short hi = f2h_rtz(w);
            const float lo = (w - (float)__builtin_bit_cast(_Float16, hi)) * 2048.f;
            img16[((((size_t)l * 2 + 0) * 4 + mt) * 2 + p) * 64 * 8 + lane * 8 + e] = hi;
            img16[((((size_t)l * 2 + 1) * 4 + mt) * 2 + p) * 64 * 8 + lane * 8 + e] = f2h_rne(lo);
            const unsigned short b0 = f2bf_trunc(w);
            const float r1 = w - bf2f(b0);
            const unsigned short b1 = f2bf_trunc(r1);
            const unsigned short b2 = f2bf_trunc(r1 - bf2f(b1));
            imgbf[((((size_t)l * 3 + 0) * 4 + mt) * 2 + p) * 64 * 8 + lane * 8 + e] = b0;
            imgbf[((((size_t)l * 3 + 1) * 4 + mt) * 2 + p) * 64 * 8 + lane * 8 + e] = b1;
            imgbf[((((size_t)l * 3 + 2) * 4 + mt) * 2 + p) * 64 * 8 + lane * 8 + e] = b2;
          }
      }
  float *dX, *dOut;
  void* dImg;
  long long* dCyc;
  hipMalloc(&dX, X.size() * 4);
  hipMalloc(&dOut, X.size() * 4);
  hipMalloc(&dImg, 1 << 20);
  hipMalloc(&dCyc, 64);
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> out(X.size());
  const char* names[3] = {"fp32 16x16x4 (64 MFMA/layer)", "f16 2-split, 3 products (24 MFMA/layer)",
                          "bf16 3-split, 6 products (48 MFMA/layer)"};
  for (int mode = 0; mode < 3; ++mode) {
    const void* src = mode == 0 ? (const void*)img32.data() : mode == 1 ? (const void*)img16.data() : (const void*)imgbf.data();
    const int bytes = mode == 0 ? (int)img32.size() * 4 : mode == 1 ? (int)img16.size() * 2 : (int)imgbf.size() * 2;
    hipMemcpy(dImg, src, bytes, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(nblocks), dim3(512), bytes, 0, dX, dImg, bytes, dOut, dCyc, iters);
      if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(nblocks), dim3(512), bytes, 0, dX, dImg, bytes, dOut, dCyc, iters);
      if (mode == 2) hipLaunchKernelGGL(chain<2>, dim3(nblocks), dim3(512), bytes, 0, dX, dImg, bytes, dOut, dCyc, iters);
      hipDeviceSynchronize();
    }
    long long cyc[8];
    hipMemcpy(cyc, dCyc, 64, hipMemcpyDeviceToHost);
    hipMemcpy(out.data(), dOut, out.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0, sumsq = 0;
    for (size_t i = 0; i < out.size(); ++i) {
      maxerr = fmax(maxerr, fabs((double)out[i] - ref[i]));
      maxref = fmax(maxref, fabs(ref[i]));
      sumsq += ((double)out[i] - ref[i]) * ((double)out[i] - ref[i]);
    }
    printf("%-44s %8.0f cycles/layer/wave (2 waves per SIMD)   max|err| %.3e  rms %.3e  (max|ref| %.2f, rel %.2e)\n",
           names[mode], (double)cyc[0] / iters / L, maxerr, sqrt(sumsq / out.size()), maxref, maxerr / maxref);
  }
  return 0;
}
