// Micro-benchmark: does a different fp32 MFMA tiling change the floor of the conditioner's
// hidden-layer chain?  h <- relu(W h), H = 64, L = 4 layers, weights in LDS, D fragment of a layer = B fragment of the
// next (no LDS round trip for activations), 256 workgroups, 1 or 2 waves per SIMD.  Variants:
//   0  v_mfma_f32_16x16x4_f32, 16 rows per wave, ONE ds_read_b32 per MFMA  (fragment-ordered image [mt][s][lane]:
//      the best case of what the throughput kernels do today -- their row-major image costs the same instruction count)
//   1  the same MFMA, image ordered [mt][s / 4][lane][4]: ONE ds_read_b128 per FOUR MFMAs
//   2  v_mfma_f32_32x32x2_f32, 32 rows per wave, one ds_read_b32 per MFMA (half the LDS reads per row and FLOP)
//   3  the same, image [mt][kk / 4][lane][4]: one ds_read_b128 per four MFMAs
// Prints ns per layer per 16 rows (device events around the launch, all CUs busy), the implied TFLOP/s against the
// 157.3 TF fp32 matrix peak, and the max error against an fp64 host evaluation (all four are exact-fp32 FMA chains).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shapes mfma_shapes.hip ; run: ./mfma_shapes
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int H = 64, L = 4;

// ---- 16x16x4: lane (j = lane & 15, g = lane >> 4); D reg r of m-tile mt = feature 16 mt + 4 r + g of row j;
// K-step s = 4 mt' + r consumes B = h[mt'][r] (feature 16 mt' + 4 r + g), A[lane] = W[16 mt + perm(j)][16 mt' + 4 r + g]
template <bool B128>
__device__ __forceinline__ void layer16(const float* __restrict__ img, int lane, f4 (&h)[4]) {
  f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  if (B128) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f4 a[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[mt] = *(const f4*)(img + ((mt * 4 + q) * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][r], h[q][r], acc[mt], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(img[(mt * 16 + s) * 64 + lane], h[s >> 2][s & 3], acc[mt], 0, 0, 0);
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[mt][r] = fmaxf(acc[mt][r], 0.f);
}

// ---- 32x32x2: lane (n = lane & 31, hf = lane >> 5); D reg i of m-tile T = feature 32 T + 8 (i >> 2) + 4 hf + (i & 3)
// of row n; K-step kk = 16 T + i consumes B = h[T][i] (that feature for this lane's hf), so
// A[lane] = W[32 mt + (lane & 31)][32 T + 8 (i >> 2) + 4 hf + (i & 3)]
template <bool B128>
__device__ __forceinline__ void layer32(const float* __restrict__ img, int lane, f16v (&h)[2]) {
  f16v acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
  if (B128) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {   // four K-steps kk = 4 q .. 4 q + 3
      f4 a[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) a[mt] = *(const f4*)(img + ((mt * 8 + q) * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][r], h[q >> 2][4 * (q & 3) + r], acc[mt], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int kk = 0; kk < 32; ++kk)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(img[(mt * 32 + kk) * 64 + lane], h[kk >> 4][kk & 15], acc[mt], 0, 0, 0);
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) h[mt][i] = fmaxf(acc[mt][i], 0.f);
}

template <int MODE>
__global__ void __launch_bounds__(512) chain(const float* __restrict__ x, const float* __restrict__ wimg,
                                             float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  for (int i = tid; i < L * H * H / 4; i += blockDim.x) ((f4*)smem)[i] = ((const f4*)wimg)[i];
  __syncthreads();
  if (MODE < 2) {
    const int j = lane & 15, g = lane >> 4;
    const long long row = ((long long)blockIdx.x * nw + wave) * 16 + j;
    f4 h0[4], h[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) h0[mt][r] = x[row * H + 16 * mt + 4 * r + g];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) h[mt] = h0[mt];
#pragma unroll
      for (int l = 0; l < L; ++l) layer16<MODE == 1>(smem + l * H * H, lane, h);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) h0[mt][r] += 0.f * h[mt][r];   // keeps the iterations dependent
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[row * H + 16 * mt + 4 * r + g] = h[mt][r];
  } else {
    const int n = lane & 31, hf = lane >> 5;
    const long long row = ((long long)blockIdx.x * nw + wave) * 32 + n;
    f16v h0[2], h[2];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int i = 0; i < 16; ++i) h0[T][i] = x[row * H + 32 * T + 8 * (i >> 2) + 4 * hf + (i & 3)];
    for (int it = 0; it < iters; ++it) {
      h[0] = h0[0];
      h[1] = h0[1];
#pragma unroll
      for (int l = 0; l < L; ++l) layer32<MODE == 3>(smem + l * H * H, lane, h);
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int i = 0; i < 16; ++i) h0[T][i] += 0.f * h[T][i];
    }
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int i = 0; i < 16; ++i) out[row * H + 32 * T + 8 * (i >> 2) + 4 * hf + (i & 3)] = h[T][i];
  }
}

int main() {
  const int nblocks = 256, iters = 400;
  const int max_rows = nblocks * 8 * 32;
  std::vector<float> W(L * H * H), X((size_t)max_rows * H);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& w : W) w = rnd() * 0.35f;
  for (auto& v : X) v = rnd() * 2.f;
  std::vector<double> ref((size_t)max_rows * H);
  for (int r = 0; r < max_rows; ++r) {
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
  // images.  16x16: output row of lane i in m-tile mt is feature 16 mt + 4 (i & 3) + (i >> 2) (so that D reg r of lane
  // (j, g) -- MFMA row 4 r'... -- lands on feature 16 mt + 4 r + g: the within-tile transpose of nsf_device.h)
  std::vector<float> img[4];
  for (auto& v : img) v.assign(L * H * H, 0.f);
  for (int l = 0; l < L; ++l) {
    for (int mt = 0; mt < 4; ++mt)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, g = lane >> 4;
        const int orow = 16 * mt + 4 * (i & 3) + (i >> 2);
        for (int s = 0; s < 16; ++s) {
          const float w = W[(l * H + orow) * H + 16 * (s >> 2) + 4 * (s & 3) + g];
          img[0][l * H * H + (mt * 16 + s) * 64 + lane] = w;
          img[1][l * H * H + ((mt * 4 + (s >> 2)) * 64 + lane) * 4 + (s & 3)] = w;
        }
      }
    // 32x32: MFMA output row m of the tile is produced by A-lane n = m; D reg i of lane (n, hf) holds MFMA row
    // 8 (i >> 2) + 4 hf + (i & 3): take that as the feature index directly (identity row permutation)
    for (int mt = 0; mt < 2; ++mt)
      for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 31, hf = lane >> 5;
        for (int kk = 0; kk < 32; ++kk) {
          const int T = kk >> 4, i = kk & 15;
          const float w = W[(l * H + 32 * mt + m) * H + 32 * T + 8 * (i >> 2) + 4 * hf + (i & 3)];
          img[2][l * H * H + (mt * 32 + kk) * 64 + lane] = w;
          img[3][l * H * H + ((mt * 8 + (kk >> 2)) * 64 + lane) * 4 + (kk & 3)] = w;
        }
      }
  }
  float *dX, *dOut, *dImg;
  hipMalloc(&dX, X.size() * 4);
  hipMalloc(&dOut, X.size() * 4);
  hipMalloc(&dImg, L * H * H * 4);
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::vector<float> out(X.size());
  const char* names[4] = {"16x16x4, ds_read_b32 per MFMA", "16x16x4, ds_read_b128 per 4 MFMAs", "32x32x2, ds_read_b32 per MFMA",
                          "32x32x2, ds_read_b128 per 4 MFMAs"};
  printf("# hidden-layer chain h <- relu(W h), H = 64, %d layers, weights in LDS, 256 workgroups, %d iterations\n", L, iters);
  for (int nw : {4, 8}) {
    for (int mode = 0; mode < 4; ++mode) {
      hipMemcpy(dImg, img[mode].data(), L * H * H * 4, hipMemcpyHostToDevice);
      const int rows_per_wave = mode < 2 ? 16 : 32;
      const long long rows = (long long)nblocks * nw * rows_per_wave;
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        const size_t sm = L * H * H * 4;
        if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(nblocks), dim3(64 * nw), sm, 0, dX, dImg, dOut, iters);
        if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(nblocks), dim3(64 * nw), sm, 0, dX, dImg, dOut, iters);
        if (mode == 2) hipLaunchKernelGGL(chain<2>, dim3(nblocks), dim3(64 * nw), sm, 0, dX, dImg, dOut, iters);
        if (mode == 3) hipLaunchKernelGGL(chain<3>, dim3(nblocks), dim3(64 * nw), sm, 0, dX, dImg, dOut, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      hipMemcpy(out.data(), dOut, (size_t)rows * H * 4, hipMemcpyDeviceToHost);
      double maxerr = 0;
      for (size_t i = 0; i < (size_t)rows * H; ++i) maxerr = fmax(maxerr, fabs((double)out[i] - ref[i]));
      const double flop = 2.0 * H * H * L * (double)rows * iters;
      const double ns_per_layer_16rows = ms * 1e6 / ((double)iters * L) / ((double)nw * rows_per_wave / 16.0 / 4.0);
      printf("%d wave(s)/SIMD  %-36s %7.3f ms  %6.1f TFLOP/s (%.2f of 157.3)  %7.1f ns per layer and 16 rows per SIMD  max|err| %.2e\n",
             nw / 4, names[mode], ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3, ns_per_layer_16rows, maxerr);
    }
  }
  return 0;
}
