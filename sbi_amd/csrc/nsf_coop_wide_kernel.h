#pragma once
// nsf_coop_wide_kernel.h -- the cooperative NSF kernels for hidden_features 65 ... 128 (nflows' ResidualNet accepts any
// width, sbi/neural_nets/net_builders/flow.py:333-349; 100 ... 128 is common practice).
//
// Every throughput kernel stages ONE transform's weight image in LDS; at hidden 128 that image is ~400 KB.  The
// cooperative kernels (nsf_coop.h) read their weights from L2 instead, so the only thing that changes with the width is
// how many hidden m-tiles a wave owns: here TWO (m-tiles wave and wave + 4 of the eight), with K = eight quads (whole
// quads: the padding features are zero in both operands).  Same image format, same stash format (HT = 8 slots per entry),
// same partial slabs and reduction as the narrow kernels; one 16-row tile per workgroup at any batch size.  The schedule is
// the plain one (weights requested where they are used, the stash written where it is produced): these kernels exist for
// coverage of the shape, the tuned pipeline of nsf_coop_kernel.h is not repeated here.
//   nsf_coopw_fwd_kernel<K, INV, MT>   log p (+ noise, + training stash)  |  INV: the sampling direction (theta from noise,
//                                      nflows Flow._sample -> CompositeTransform.inverse behind NFlowsFlow.sample,
//                                      nflows_flow.py:111-128), with U^-1 / L^-1 packed into the image.  MT = 2: the wide nets;
//                                      MT = 1 (hidden <= 64, the narrow image) is instantiated for INV only: small sampling
//                                      calls of ordinary nets, which the tuned narrow family has no kernel for
//   nsf_coopw_bwd_kernel<K, NT>        all T transforms backward in one launch, partial slabs as the narrow kernel writes
//                                      them; NT = 16-row tiles per workgroup (two above 4 096 rows when they fit LDS)
#include "nsf_coop_kernel.h"

#define COW_HT 8      // hidden m-tiles
#define COW_KQ 8      // K-quads of a hidden-K matrix
#define COW_CQ 4      // K-quads of the context part of the initial / gate layers: x-dim <= 64 (no throughput kernel to
                      // fall back to at this width, so the wide kernels take twice the narrow kernels' x-dim)

// all-gather of the 4 MT hidden D fragments (wave w holds m-tiles w, w + 4, ...); one barrier
template <int MT>
__device__ __forceinline__ void cow_gather(float* __restrict__ ex, int& buf, int wave, int lane, const f4 (&mine)[MT],
                                           f4 (&out)[4 * MT]) {
  f4* e = reinterpret_cast<f4*>(ex) + buf * (4 * MT * 64);
#pragma unroll
  for (int q = 0; q < MT; ++q) e[(wave + CO_WAVES * q) * 64 + lane] = mine[q];
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < 4 * MT; ++mt) out[mt] = e[mt * 64 + lane];
  buf ^= 1;
}
// acc += A (one m-tile, KQ quads) * B (the KQ gathered fragments): 4 KQ K-steps
template <int KQ>
__device__ __forceinline__ void cow_gemm(const f4 (&a)[KQ], const f4 (&b)[KQ], f4& acc) {
#pragma unroll
  for (int s = 0; s < 4 * KQ; ++s) acc = MFMA16(a[s >> 2][s & 3], b[s >> 2][s & 3], acc);
}
// acc += A (context quads) * standardized context (K-steps of the context live in registers)
__device__ __forceinline__ void cow_gemm_ctx(const f4 (&a)[COW_CQ], int kcq, const float (&cb)[4 * COW_CQ], f4& acc) {
#pragma unroll
  for (int q = 0; q < COW_CQ; ++q)
    if (q < kcq) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = MFMA16(a[q][r], cb[4 * q + r], acc);
    }
}

// ------------------------------------------------------------------------------------------------ forward / inverse
// MT = hidden m-tiles per wave: 2 (hidden 65 ... 128: eight m-tiles, eight K-quads) or 1 (hidden <= 64: four / four, the
// narrow image) -- the latter is instantiated for the SAMPLING direction only, which the tuned narrow family lacks.
template <int K, bool INV, int MT>
__global__ void __launch_bounds__(64 * CO_WAVES, 1)
nsf_coopw_fwd_kernel(const CoK k, const float* __restrict__ cimg, const float* __restrict__ zstats,
                     const float* __restrict__ in, const float* __restrict__ x, long long n, long long x_rows,
                     float* __restrict__ out_main, float* __restrict__ out_aux, float* __restrict__ zst,
                     float* __restrict__ ast) {
  constexpr int PT = (3 * K - 1 + 15) / 16;
  constexpr int R = 16;
  constexpr int HT = 4 * MT, KQ = 4 * MT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const LaneId id = make_lane();
  const int D = k.D, C = k.C, ZS = k.ZS, NB = k.NB, kcq = k.KCQ;
  float* zs = lds + k.o_zs;
  float* pst = lds + k.o_pst;
  float* ex = lds + k.o_ex;
  float* ldp = lds + k.o_ldp;
  const long long row0 = (long long)blockIdx.x * R;
  const long long nt16 = (n + 15) / 16;
  const float* th_shift = zstats;
  const float* th_scale = zstats + D;
  const float* x_mean = zstats + 2 * D;
  const float* x_std = x_mean + C;
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- prologue: state rows -> LDS (density direction: z-scored theta; sampling direction: the noise as it is);
  //      standardized context as B fragments (K-step s <-> c = 4 s + g)
  for (int i = tid; i < R * ZS + 16; i += 64 * CO_WAVES) zs[i] = 0.f;
  float cb[4 * COW_CQ];
  {
    const long long row = row0 + id.j;
    const long long rs = row < n ? row : 0;
    const long long xr = (x_rows == n) ? rs : (x_rows == 1 ? 0 : rs % x_rows);
#pragma unroll
    for (int s = 0; s < 4 * COW_CQ; ++s) {
      const int c = 4 * s + id.g;
      const int cc = c < C ? c : 0;
      cb[s] = (c < C && row < n) ? (x[xr * C + cc] - x_mean[cc]) / x_std[cc] : 0.f;
    }
  }
  __syncthreads();
  for (int i = tid; i < R * D; i += 64 * CO_WAVES) {
    const int r = i / D, d = i - r * D;
    const long long row = row0 + r;
    const float v = row < n ? in[row * D + d] : 0.f;
    zs[r * ZS + d] = INV ? v : (row < n ? v * th_scale[d] + th_shift[d] : 0.f);
  }
  float ld_acc = 0.f, ld_const = 0.f;
  for (int d = 0; d < D; ++d) ld_const += logf(fabsf(th_scale[d]));
  if (INV) ld_const = -ld_const;
  int buf = 0;
  float* abase = (!INV && ast && (row0 >> 4) < nt16) ? ast + (row0 >> 4) * k.slots * 256 + 4 * id.lane : nullptr;
  const long long astride = nt16 * k.slots * 256;
  __syncthreads();

  for (int li = 0; li < k.T; ++li) {
    const int t = INV ? k.T - 1 - li : li;
    const int par = t & 1;
    const CoKP& kp = k.p[par];
    const float* img = cimg + (long long)t * k.img_floats;
    float* ab = abase ? abase + t * astride : nullptr;
    if (!INV && zst)
      for (int i = tid; i < R * D; i += 64 * CO_WAVES) {
        const int r = i / D, d = i - r * D;
        if (row0 + r < n) zst[((long long)t * n + row0 + r) * D + d] = zs[r * ZS + d];
      }
    if (INV) {     // z = U^-1 (L^-1 (y - b)): two chained 16 x 16 MFMA mat-vecs with the packed inverses
      if (wave == 0) {
        const f4 ali = *(reinterpret_cast<const f4*>(img + kp.li) + id.lane);
        const f4 aui = *(reinterpret_cast<const f4*>(img + kp.ui) + id.lane);
        const f4 blu = co_load_bias(img + kp.blu, 0, id.g);
        f4 w = zero4, zn = zero4;
#pragma unroll
        for (int s = 0; s < 4; ++s) w = MFMA16(ali[s], zs[id.j * ZS + 4 * s + id.g] - blu[s], w);
#pragma unroll
        for (int s = 0; s < 4; ++s) zn = MFMA16(aui[s], w[s], zn);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * r + id.g < D) zs[id.j * ZS + 4 * r + id.g] = zn[r];
      }
      ld_const -= img[kp.ld];
      __syncthreads();
    }
    // ---- initial layer: h = W0 [context ; z_id] + b0   (m-tiles wave, wave + 4)
    f4 h[MT], gate[MT], tt[MT];
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      const int mt = wave + CO_WAVES * q;
      f4 a[COW_CQ + 1];
      co_load_a<COW_CQ + 1>(img + kp.w0 + mt * (kcq + 1) * 256, id.lane, a);
      h[q] = co_load_bias(img + kp.b0, mt, id.g);
      const f4 wc[COW_CQ] = {a[0], a[1], a[2], a[3]};
      cow_gemm_ctx(wc, kcq, cb, h[q]);
      // the identity features' quad sits behind the context quads
      const f4 az = kcq == 1 ? a[1] : (kcq == 2 ? a[2] : (kcq == 3 ? a[3] : a[4]));
#pragma unroll
      for (int sz = 0; sz < 2; ++sz) {
        const int kz = 4 * sz + id.g;
        const int kzc = kz < kp.d_id ? kz : 0;
        const float zv = zs[id.j * ZS + 2 * kzc + (1 - par)];
        h[q] = MFMA16(az[sz], kz < kp.d_id ? zv : 0.f, h[q]);
      }
      if (ab) *reinterpret_cast<f4*>(ab + mt * 256) = h[q];
    }
    // ---- residual blocks: h += W2 relu(W1 relu(h) + b1) * sigmoid(Wc c + bc)
    // (weights are requested where they are used: the kernel fits 2 waves per SIMD that way, which hides the L2 round
    //  trips better than requesting a block's matrices up front at one wave per SIMD -- measured: 0.95 vs 1.65 ms at
    //  65 536 rows, 0.072 vs 0.077 ms at 200)
    for (int b = 0; b < NB; ++b) {
      f4 bg[HT];
#pragma unroll
      for (int q = 0; q < MT; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) tt[q][r] = fmaxf(h[q][r], 0.f);
      cow_gather<MT>(ex, buf, wave, id.lane, tt, bg);
      const int sb = k.s_blk + 4 * HT * b;      // stash slots of the block: t1 | t2 | sigmoid(gate) | h_{b+1}
#pragma unroll
      for (int q = 0; q < MT; ++q) {
        const int mt = wave + CO_WAVES * q;
        f4 a[KQ], ac[COW_CQ];
        co_load_a<KQ>(img + kp.w10 + b * k.sA + mt * KQ * 256, id.lane, a);
        co_load_a<COW_CQ>(img + kp.wc0 + b * k.sA + mt * kcq * 256, id.lane, ac);
        f4 u1 = co_load_bias(img + kp.b10 + b * k.sB, mt, id.g);
        gate[q] = co_load_bias(img + kp.bc0 + b * k.sB, mt, id.g);
        cow_gemm_ctx(ac, kcq, cb, gate[q]);
        cow_gemm<KQ>(a, bg, u1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gate[q][r] = sigmoid_gate(gate[q][r]);
          tt[q][r] = fmaxf(u1[r], 0.f);
        }
        if (ab) {
          *reinterpret_cast<f4*>(ab + (sb + mt) * 256) = u1;                         // t1 (pre-relu)
          *reinterpret_cast<f4*>(ab + (sb + 2 * HT + mt) * 256) = gate[q];       // sigmoid(gate)
        }
      }
      cow_gather<MT>(ex, buf, wave, id.lane, tt, bg);
#pragma unroll
      for (int q = 0; q < MT; ++q) {
        const int mt = wave + CO_WAVES * q;
        f4 a[KQ];
        co_load_a<KQ>(img + kp.w20 + b * k.sA + mt * KQ * 256, id.lane, a);
        f4 u2 = co_load_bias(img + kp.b20 + b * k.sB, mt, id.g);
        cow_gemm<KQ>(a, bg, u2);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[q][r] += u2[r] * gate[q][r];
        if (ab) {
          *reinterpret_cast<f4*>(ab + (sb + HT + mt) * 256) = u2;                // t2
          *reinterpret_cast<f4*>(ab + (sb + 3 * HT + mt) * 256) = h[q];          // h_{b+1}
        }
      }
    }
    // ---- final layer: parameter tiles wave, wave + 4, ... -> staging rows pst[row][dim][3K-1 raw outputs]
    {
      f4 hb[HT];
      cow_gather<MT>(ex, buf, wave, id.lane, h, hb);
      for (int mt = wave; mt < kp.nft; mt += CO_WAVES) {
        f4 a[KQ];
        co_load_a<KQ>(img + kp.wf + mt * KQ * 256, id.lane, a);
        f4 acc = co_load_bias(img + kp.bf, mt, id.g);
        cow_gemm<KQ>(a, hb, acc);
        const int dd = mt / PT, pt = mt - dd * PT;
#pragma unroll
        for (int r = 0; r < 4; ++r) pst[id.j * k.DSTR + dd * k.PSW + 16 * pt + 4 * r + id.g] = acc[r];
        if (ab) *reinterpret_cast<f4*>(ab + (k.s_par + mt) * 256) = acc;
      }
    }
    __syncthreads();
    // ---- spline: task (row j, dim 2 wave + slot) on the lane pair (lane, lane ^ 32)
    {
      const int slot = id.g & 1, part = id.g >> 1;
      const int dd_raw = 2 * wave + slot;
      const bool live = dd_raw < kp.d_tr;
      const int dd = live ? dd_raw : 0;
      const int zi = id.j * ZS + 2 * dd + par;
      float y, ld;
      rq_spline_pair<K, INV>(pst + id.j * k.DSTR + dd * k.PSW, zs[zi], k, part, y, ld);
      if (live && part == 0) zs[zi] = y;
      ld_acc += (live && part == 0) ? ld : 0.f;
    }
    __syncthreads();
    if (!INV) {    // LULinear: y = L (U z) + b as two chained 16 x 16 MFMA mat-vecs
      if (wave == 0) {
        const f4 au = *(reinterpret_cast<const f4*>(img + kp.u) + id.lane);
        const f4 al = *(reinterpret_cast<const f4*>(img + kp.l) + id.lane);
        f4 acc = zero4;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = MFMA16(au[s], zs[id.j * ZS + 4 * s + id.g], acc);
        f4 yv = co_load_bias(img + kp.blu, 0, id.g);
#pragma unroll
        for (int s = 0; s < 4; ++s) yv = MFMA16(al[s], acc[s], yv);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * r + id.g < D) zs[id.j * ZS + 4 * r + id.g] = yv[r];
      }
      ld_const += img[kp.ld];
      __syncthreads();
    }
  }

  // ---- epilogue: per-row sums in a fixed order (deterministic)
  {
    const int slot = id.g & 1, part = id.g >> 1;
    if (part == 0) ldp[(2 * wave + slot) * R + id.j] = ld_acc;
  }
  __syncthreads();
  if (tid < R) {
    const long long row = row0 + tid;
    if (row < n) {
      float ld = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) ld += ldp[q * R + tid];
      if (!INV) {
        float ss = 0.f;
        for (int d = 0; d < D; ++d) {
          const float z = zs[tid * ZS + d];
          ss += z * z;
          if (out_aux) out_aux[row * D + d] = z;
        }
        out_main[row] = -0.5f * ss + ld + ld_const - k.log_z;
      } else {
        for (int d = 0; d < D; ++d) out_main[row * D + d] = (zs[tid * ZS + d] - th_shift[d]) / th_scale[d];
        if (out_aux) out_aux[row] = ld + ld_const;
      }
    }
  }
}

template <int K, bool INV, int MT>
static int cow_launch_fwd(const CoK& k, const CoopPlan& cp, const float* cimg, const float* zstats, const float* in,
                          const float* x, long long n, long long x_rows, float* out_main, float* out_aux, float* zst,
                          float* ast, hipStream_t st) {
  auto kern = nsf_coopw_fwd_kernel<K, INV, MT>;
  const int lds_bytes = 4 * cp.lds_floats;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(cp.grid), dim3(64 * CO_WAVES), (size_t)lds_bytes, st, k, cimg, zstats, in, x, n, x_rows,
                     out_main, out_aux, zst, ast);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ backward
// Same phases, LDS tiles, stash and slab formats as nsf_coop_bwd_kernel; wave w owns hidden m-tiles w and w + 4
// everywhere: the d-activation chain (transposed matrices, K = eight quads), the transposed tiles it publishes and the
// weight-gradient tiles of those output features (up to nine n-tiles per m-tile).  NT = 16-row tiles per workgroup: two
// above 4 096 rows when the tiles fit LDS (the weight gradients then contract over 32 rows: half the partial slabs).
// d loss / d embedded x (trainable embedding nets) is not offered at this width.
template <int NT>
__device__ __forceinline__ void cow_gather_nt(float* __restrict__ ex, int& buf, int wave, int lane,
                                              const f4 (&mine)[2][NT], f4 (&out)[NT][COW_HT]) {
  f4* e = reinterpret_cast<f4*>(ex) + buf * (COW_HT * NT * 64);
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int u = 0; u < NT; ++u) e[((wave + CO_WAVES * q) * NT + u) * 64 + lane] = mine[q][u];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int mt = 0; mt < COW_HT; ++mt) out[u][mt] = e[(mt * NT + u) * 64 + lane];
  buf ^= 1;
}

template <int K, int NT>
__global__ void __launch_bounds__(64 * CO_WAVES, 1)
nsf_coopw_bwd_kernel(const CoK k, const float* __restrict__ cimg, const float* __restrict__ zstats,
                     const float* __restrict__ x, long long n, long long x_rows, const float* __restrict__ row_w,
                     const float uni_w, const float* __restrict__ z_last, const float* __restrict__ zst,
                     const float* __restrict__ ast, float* __restrict__ partial, float* __restrict__ grad_theta) {
  constexpr int PT = (3 * K - 1 + 15) / 16;
  constexpr int R = 16 * NT;
  constexpr int NNH = COW_HT + 1;          // n-tiles of a hidden-input layer: 8 x 16 inputs + the bias column
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const LaneId id = make_lane();
  const int D = k.D, C = k.C, H = k.H, ZS = k.ZS, RS = k.RS, NB = k.NB;
  float* zs = lds + k.o_zs;
  float* gys = lds + k.o_gys;
  float* gzs = lds + k.o_gzs;
  float* wrow = lds + k.o_w;
  float* pst = lds + k.o_pst;
  float* ex = lds + k.o_ex;
  float* GT0 = lds + k.o_gt;
  float* GT1 = GT0 + 16 * COW_HT * RS;
  float* AT0 = lds + k.o_at;
  float* AT1 = AT0 + (16 * COW_HT + 1) * RS;
  float* CT = lds + k.o_ct;
  float* GUT = lds + k.o_lut;
  float* GZT = GUT + 17 * RS;
  float* YT = GZT + 17 * RS;
  float* UTt = YT + 17 * RS;
  float* ctx = lds + k.o_ctx;
  const long long row0 = (long long)blockIdx.x * R;
  const long long nt16 = (n + 15) / 16;
  const float* x_mean = zstats + 2 * D;
  const float* x_std = x_mean + C;
  const bool hbf = H == 16 * COW_HT;       // no spare activation-tile column: bias gradients from an extra MFMA
  const int ones_h = hbf ? -1 : H;
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int ntc = k.ntc, nnh = k.nnh;
  const int blk_tiles = COW_HT * ntc + 2 * COW_HT * nnh;     // slab tiles of one residual block: d Wc | d W1 | d W2
  const long long astride = nt16 * k.slots * 256;
  const float* abase[NT];       // stash of this workgroup's tiles (clamped: tiles past the last row were never written)
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const long long t16 = (row0 >> 4) + u < nt16 ? (row0 >> 4) + u : nt16 - 1;
    abase[u] = ast + t16 * k.slots * 256 + 4 * id.lane;
  }

  for (int i = tid; i < k.o_w - k.o_zs; i += 64 * CO_WAVES) lds[k.o_zs + i] = 0.f;   // state rows incl. padding
  if (tid < R) {
    const long long row = row0 + tid;
    wrow[tid] = row < n ? (row_w ? row_w[row] : uni_w) : 0.f;
  }
  for (int i = tid; i < C * R; i += 64 * CO_WAVES) {
    const int cc = i / R, r = i - cc * R;
    const long long row = row0 + r;
    const long long rs = row < n ? row : 0;
    const long long xr = (x_rows == n) ? rs : (x_rows == 1 ? 0 : rs % x_rows);
    ctx[cc * R + r] = row < n ? (x[xr * C + cc] - x_mean[cc]) / x_std[cc] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < R * D; i += 64 * CO_WAVES) {   // d/dz_T of w * 0.5 |z_T|^2
    const int r = i / D, d = i - r * D;
    const long long row = row0 + r;
    gzs[r * ZS + d] = row < n ? wrow[r] * z_last[row * D + d] : 0.f;
  }
  int buf = 0;
  __syncthreads();

  // weight-gradient tiles of one output m-tile: acc[nt] of lane (g, j) = d W[out0 + j][16 nt + 4 g + r], contracted
  // over all rows of the workgroup
  auto dw_tiles = [&](const float* Gt, const float* At, int out0, int in_off, int nnt, f4 (&acc)[NNH], f4* accb) {
    f4 b[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) b[u] = *reinterpret_cast<const f4*>(Gt + (out0 + id.j) * RS + 16 * u + 4 * id.g);
#pragma unroll
    for (int nt = 0; nt < NNH; ++nt) {
      acc[nt] = zero4;
      if (nt < nnt) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const f4 a = *reinterpret_cast<const f4*>(At + (in_off + 16 * nt + id.j) * RS + 16 * u + 4 * id.g);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[nt] = MFMA16(a[s], b[u][s], acc[nt]);
        }
      }
    }
    if (accb) {
      *accb = zero4;
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) *accb = MFMA16(1.f, b[u][s], *accb);
    }
  };
  auto store_T = [&](float* T, int f0, int u, const f4& v, bool relu, int ones_row) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = f0 + 4 * r + id.g;
      float a = relu ? fmaxf(v[r], 0.f) : v[r];
      a = (f == ones_row) ? 1.f : a;
      T[f * RS + 16 * u + id.j] = a;
    }
  };

  for (int t = k.T - 1; t >= 0; --t) {
    const int par = t & 1;
    const CoKP& kp = k.p[par];
    const float* img = cimg + (long long)t * k.img_floats;
    float* part = NSF_DBG_ABL(k.ablate, 32768) ? nullptr : partial + ((long long)t * gridDim.x + blockIdx.x) * k.PLP;
    const float* at[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) at[u] = abase[u] + t * astride;
    const int tb_blk0 = COW_HT * kp.nnt0;             // slab tile bases: d W0 | blocks | d Wf | LULinear tail
    const int tb_wf = tb_blk0 + NB * blk_tiles;
    // ---- P0: state rows + conditioner-input tile, spline parameters -> LDS; LULinear backward
    for (int i = tid; i < R * D; i += 64 * CO_WAVES) {
      const int r = i / D, d = i - r * D;
      const long long row = row0 + r;
      const float v = row < n ? zst[((long long)t * n + row) * D + d] : 0.f;
      zs[r * ZS + d] = v;
      if ((d & 1) == (1 - par)) CT[(d >> 1) * RS + r] = v;     // identity feature k = (d - (1 - par)) / 2
    }
    for (int i = tid; i < (k.ct_rows - kp.d_id) * R; i += 64 * CO_WAVES) {
      const int kk = kp.d_id + i / R, r = i % R;
      CT[kk * RS + r] = kk < kp.in0 ? ctx[(kk - kp.d_id) * R + r] : (kk == kp.in0 ? 1.f : 0.f);
    }
    for (int mt = wave; mt < kp.nft; mt += CO_WAVES) {
      const int dd = mt / PT, pt = mt - dd * PT;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const f4 pt4 = *reinterpret_cast<const f4*>(at[u] + (k.s_par + mt) * 256);
#pragma unroll
        for (int r = 0; r < 4; ++r) pst[(16 * u + id.j) * k.DSTR + dd * k.PSW + 16 * pt + 4 * r + id.g] = pt4[r];
      }
    }
    if (wave < NT) {   // g_u = L^T g_z, g_y = U^T g_u for row tile `wave`
      const int u = wave;
      const f4 a_lt = *(reinterpret_cast<const f4*>(img + kp.lt) + id.lane);
      const f4 a_ut = *(reinterpret_cast<const f4*>(img + kp.ut) + id.lane);
      f4 gu = zero4, gy = zero4;
      float gz[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        gz[s] = gzs[(16 * u + id.j) * ZS + 4 * s + id.g];
        gu = MFMA16(a_lt[s], gz[s], gu);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) gy = MFMA16(a_ut[s], gu[s], gy);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 4 * r + id.g;
        if (d < D) gys[(16 * u + id.j) * ZS + d] = gy[r];
        GUT[d * RS + 16 * u + id.j] = gu[r];
        GZT[d * RS + 16 * u + id.j] = gz[r];
      }
    }
    __syncthreads();
    // ---- P1: spline forward + reverse mode; the parameter rows become d loss / d(raw conditioner outputs)
    {
      const int slt = id.g & 1, sp = id.g >> 1;
      const int dd = 2 * wave + slt;
      if (dd < kp.d_tr) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const int r = 16 * u + id.j;
          const int zi = r * ZS + 2 * dd + par;
          float yv, gxv;
          rq_spline_pair_bwd<K>(pst + r * k.DSTR + dd * k.PSW, 16 * PT, zs[zi], gys[zi], -wrow[r], k, sp, yv, gxv);
          if (sp == 0) {
            zs[zi] = yv;
            gys[zi] = gxv;
          }
        }
      }
    }
    __syncthreads();
    // ---- P2: u = U y (LU parameter gradients), h_last -> activation tile, g_h = Wf^T g_p
    if (wave < NT) {
      const int u = wave;
      const f4 a_u = *(reinterpret_cast<const f4*>(img + kp.u) + id.lane);
      f4 uv = zero4;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float yv = zs[(16 * u + id.j) * ZS + 4 * s + id.g];
        uv = MFMA16(a_u[s], yv, uv);
        YT[(4 * s + id.g) * RS + 16 * u + id.j] = yv;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) UTt[(4 * r + id.g) * RS + 16 * u + id.j] = uv[r];
    }
    f4 gh[2][NT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int mt = wave + CO_WAVES * q;
      f4 acc0[NT], acc1[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const f4 hl = *reinterpret_cast<const f4*>(at[u] + (k.s_blk + 4 * COW_HT * (NB - 1) + 3 * COW_HT + mt) * 256);
        store_T(AT0, 16 * mt, u, hl, false, ones_h);
        acc0[u] = zero4;
        acc1[u] = zero4;
      }
      const f4* ap = reinterpret_cast<const f4*>(img + kp.wft + mt * kp.d_tr * PT * 256) + id.lane;
      for (int dd = 0; dd < kp.d_tr; ++dd) {
#pragma unroll
        for (int qq = 0; qq < PT; ++qq) {
          const f4 a = ap[(dd * PT + qq) * 64];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
              const float bv = pst[(16 * u + id.j) * k.DSTR + dd * k.PSW + 16 * qq + 4 * r + id.g];
              if (qq & 1) acc1[u] = MFMA16(a[r], bv, acc1[u]);
              else acc0[u] = MFMA16(a[r], bv, acc0[u]);
            }
        }
      }
#pragma unroll
      for (int u = 0; u < NT; ++u) gh[q][u] = acc0[u] + acc1[u];
    }
    __syncthreads();
    // ---- d Wf (parameter tiles wave, wave + 4, ...), LULinear parameter gradients
    for (int mt = wave; mt < kp.nft; mt += CO_WAVES) {
      const int dd = mt / PT, pt = mt - dd * PT;
      f4 acc[NNH], accb = zero4;
      float bv[NT][4];    // B side = g_p: parameter 16 pt + j of rows 16 u + 4 g + s
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) bv[u][s] = pst[(16 * u + 4 * id.g + s) * k.DSTR + dd * k.PSW + 16 * pt + id.j];
#pragma unroll
      for (int nt = 0; nt < NNH; ++nt) {
        acc[nt] = zero4;
        if (nt < nnh) {
#pragma unroll
          for (int u = 0; u < NT; ++u) {
            const f4 a = *reinterpret_cast<const f4*>(AT0 + (16 * nt + id.j) * RS + 16 * u + 4 * id.g);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[nt] = MFMA16(a[s], bv[u][s], acc[nt]);
          }
        }
      }
      if (hbf) {
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
          for (int s = 0; s < 4; ++s) accb = MFMA16(1.f, bv[u][s], accb);
      }
      const bool p_ok = 16 * pt + id.j < k.P;
#pragma unroll
      for (int nt = 0; nt < NNH; ++nt)
        if (nt < nnh && !(hbf && nt == COW_HT)) co_write_tile(part, tb_wf + mt * nnh + nt, p_ok, nt, H, id, acc[nt]);
      if (hbf) co_write_tile(part, tb_wf + mt * nnh + COW_HT, p_ok, COW_HT, H, id, accb);
    }
    if (part) {
      const int ntri = D * (D - 1) / 2;
      float* plow = part + kp.dw_tail;
      float* pup = plow + ntri;
      float* pdiag = pup + ntri;
      float* pbias = pdiag + D;
      if (wave == 3 || wave == 2) {   // d U = g_u (x) y (wave 3), d L = g_z (x) u (wave 2): one 16 x 16 tile each
        f4 acc[1];
        co_dw<NT, 1>(wave == 3 ? GUT : GZT, wave == 3 ? YT : UTt, RS, 0, 0, 1, id, acc, nullptr);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = id.j, kk = 4 * id.g + r;
          if (i < D && kk < D) {
            if (wave == 3) {
              if (kk > i) pup[i * D - i * (i + 1) / 2 + (kk - i - 1)] = acc[0][r];
              else if (kk == i) pdiag[i] = acc[0][r];     // dL/dU_ii; chain rule finished in the reduction
            } else if (kk < i) {
              plow[i * (i - 1) / 2 + kk] = acc[0][r];
            }
          }
        }
      } else if (wave == 1) {
        if (id.lane < D) {            // d bias = sum_n g_z
          float a = 0.f;
          for (int r = 0; r < R; ++r) a += gzs[r * ZS + id.lane];
          pbias[id.lane] = a;
        } else if (id.lane == 63) {   // sum_n d loss / d logabsdet_n = - sum_n w_n
          float a = 0.f;
          for (int r = 0; r < R; ++r) a -= wrow[r];
          plow[D * (D - 1) + 2 * D] = a;
        }
      }
    }
    // ---- residual blocks, last -> first
    for (int b = NB - 1; b >= 0; --b) {
      const int sb = k.s_blk + 4 * COW_HT * b;
      const int tb_c = tb_blk0 + b * blk_tiles, tb_1 = tb_c + COW_HT * ntc, tb_2 = tb_1 + COW_HT * nnh;
      f4 t1[2][NT], hin[2][NT], ga[2][NT], bg[NT][COW_HT];
      f4 a2t[2][COW_KQ], a1t[2][COW_KQ];     // both transposed matrices of the block: one L2 round trip with the stash
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int mt = wave + CO_WAVES * q;
        co_load_a<COW_KQ>(img + kp.w2t0 + b * k.sT + mt * COW_KQ * 256, id.lane, a2t[q]);
        co_load_a<COW_KQ>(img + kp.w1t0 + b * k.sT + mt * COW_KQ * 256, id.lane, a1t[q]);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int mt = wave + CO_WAVES * q;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          t1[q][u] = *reinterpret_cast<const f4*>(at[u] + (sb + mt) * 256);
          const f4 t2 = *reinterpret_cast<const f4*>(at[u] + (sb + COW_HT + mt) * 256);
          const f4 sg = *reinterpret_cast<const f4*>(at[u] + (sb + 2 * COW_HT + mt) * 256);
          hin[q][u] = *reinterpret_cast<const f4*>(at[u] + (b == 0 ? mt : sb - COW_HT + mt) * 256);   // h_0 or h_b
          f4 gc;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            ga[q][u][r] = gh[q][u][r] * sg[r];                                  // d t2
            gc[r] = gh[q][u][r] * t2[r] * sg[r] * (1.f - sg[r]);                // d (Wc c + bc)
          }
          store_T(GT0, 16 * mt, u, ga[q][u], false, -1);
          store_T(GT1, 16 * mt, u, gc, false, -1);
          store_T(AT1, 16 * mt, u, t1[q][u], true, ones_h);
        }
      }
      cow_gather_nt<NT>(ex, buf, wave, id.lane, ga, bg);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int mt = wave + CO_WAVES * q;
        const bool out_ok = 16 * mt + id.j < H;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          f4 gr = zero4;
          cow_gemm<COW_KQ>(a2t[q], bg[u], gr);
#pragma unroll
          for (int r = 0; r < 4; ++r) ga[q][u][r] = t1[q][u][r] > 0.f ? gr[r] : 0.f;     // d t1
        }
        f4 acc[NNH], accb;
        dw_tiles(GT0, AT1, 16 * mt, 0, nnh, acc, hbf ? &accb : nullptr);
#pragma unroll
        for (int nt = 0; nt < NNH; ++nt)
          if (nt < nnh && !(hbf && nt == COW_HT)) co_write_tile(part, tb_2 + mt * nnh + nt, out_ok, nt, H, id, acc[nt]);
        if (hbf) co_write_tile(part, tb_2 + mt * nnh + COW_HT, out_ok, COW_HT, H, id, accb);
        dw_tiles(GT1, CT, 16 * mt, kp.d_id, ntc, acc, nullptr);
#pragma unroll
        for (int nt = 0; nt < COW_CQ + 1; ++nt)
          if (nt < ntc) co_write_tile(part, tb_c + mt * ntc + nt, out_ok, nt, C, id, acc[nt]);
      }
      wave_lds_fence();
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int mt = wave + CO_WAVES * q;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          store_T(GT0, 16 * mt, u, ga[q][u], false, -1);
          store_T(AT0, 16 * mt, u, hin[q][u], true, ones_h);
        }
      }
      cow_gather_nt<NT>(ex, buf, wave, id.lane, ga, bg);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int mt = wave + CO_WAVES * q;
        const bool out_ok = 16 * mt + id.j < H;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          f4 gr = zero4;
          cow_gemm<COW_KQ>(a1t[q], bg[u], gr);
#pragma unroll
          for (int r = 0; r < 4; ++r) gh[q][u][r] += hin[q][u][r] > 0.f ? gr[r] : 0.f;
        }
        f4 acc[NNH], accb;
        dw_tiles(GT0, AT0, 16 * mt, 0, nnh, acc, hbf ? &accb : nullptr);
#pragma unroll
        for (int nt = 0; nt < NNH; ++nt)
          if (nt < nnh && !(hbf && nt == COW_HT)) co_write_tile(part, tb_1 + mt * nnh + nt, out_ok, nt, H, id, acc[nt]);
        if (hbf) co_write_tile(part, tb_1 + mt * nnh + COW_HT, out_ok, COW_HT, H, id, accb);
      }
      wave_lds_fence();
    }
    // ---- initial layer
    {
      f4 bg[NT][COW_HT];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int u = 0; u < NT; ++u) store_T(GT0, 16 * (wave + CO_WAVES * q), u, gh[q][u], false, -1);
      cow_gather_nt<NT>(ex, buf, wave, id.lane, gh, bg);
      if (wave == 0) {      // identity features receive W0[:, :d_id]^T g_h0
        f4 a0t[COW_KQ];
        co_load_a<COW_KQ>(img + kp.w0t, id.lane, a0t);
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          f4 gin = zero4;
          cow_gemm<COW_KQ>(a0t, bg[u], gin);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kk = 4 * r + id.g;
            if (kk < kp.d_id) gys[(16 * u + id.j) * ZS + 2 * kk + (1 - par)] += gin[r];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int mt = wave + CO_WAVES * q;
        const bool out_ok = 16 * mt + id.j < H;
        f4 acc[NNH];
        dw_tiles(GT0, CT, 16 * mt, 0, kp.nnt0, acc, nullptr);
#pragma unroll
        for (int nt = 0; nt < COW_CQ + 2; ++nt)
          if (nt < kp.nnt0) co_write_tile(part, mt * kp.nnt0 + nt, out_ok, nt, kp.in0, id, acc[nt]);
      }
    }
    __syncthreads();
    // gradient wrt this transform's input becomes the upstream gradient of the transform below
    if (t > 0) {
      float* tmp = gzs;
      gzs = gys;
      gys = tmp;
    } else if (grad_theta) {
      for (int i = tid; i < R * D; i += 64 * CO_WAVES) {
        const int r = i / D, d = i - r * D;
        if (row0 + r < n) grad_theta[(row0 + r) * D + d] = gys[r * ZS + d] * zstats[D + d];
      }
    }
  }
}

template <int K, int NT>
static int cow_launch_bwd(const CoK& k, const CoopPlan& cp, const CoBwdArgs& a, hipStream_t st) {
  if (a.grad_x) return SBI_AMD_E_UNSUPPORTED;      // d loss / d embedded x: narrow nets only
  auto kern = nsf_coopw_bwd_kernel<K, NT>;
  const int lds_bytes = 4 * cp.lds_floats;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(cp.grid), dim3(64 * CO_WAVES), (size_t)lds_bytes, st, k, a.cimg, a.zstats, a.x, a.n,
                     a.x_rows, a.row_w, a.uni_w, a.z_last, a.zst, a.ast, a.partial, a.grad_theta);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ dispatch
// one translation unit per bin count (parallel build): nsf_coop.hip holds K = 10, nsf_coop_k{4,5,8,16}.hip the rest
template <int K>
int co_fwd_k(const NsfPlan& pl, const CoopPlan& cp, const CoFwdArgs& a, hipStream_t st) {
  CoK k;
  if (a.mc) {      // persistent slice sampler: one-tile workgroups (16 chains each), whatever the chain count
    if (cp.MT == 2) return SBI_AMD_E_UNSUPPORTED;
    CoopPlan cf = cp;
    if (cp.NT != 1) {
      int rc = coop_build_plan(pl, a.n, 1, false, &cf);
      if (rc) return rc;
    }
    coop_make_consts(pl, cf, &k);
    return pl.KSH == 13 ? co_launch_fwd<K, 13, 1, false, true>(k, cf, a, st) : co_launch_fwd<K, 16, 1, false, true>(k, cf, a, st);
  }
  if (cp.MT == 2) {       // hidden > 64: the wide kernel, always one tile per workgroup (the backward pass may take two)
    CoopPlan cf = cp;
    if (cp.NT != 1) {
      int rc = coop_build_plan(pl, a.n, 1, false, &cf);
      if (rc) return rc;
    }
    coop_make_consts(pl, cf, &k);
    return cow_launch_fwd<K, false, 2>(k, cf, a.cimg, a.zstats, a.theta, a.x, a.n, a.x_rows, a.logp, a.noise, a.zst, a.ast,
                                    st);
  }
  if (cp.NT == 2 && coop_lean_forward()) {
    // more than 4096 rows: the forward pass runs as one-tile workgroups, two to a CU (the LEAN instantiation, <= 256
    // registers) -- 17 % faster than the two-tile workgroups the backward pass keeps for its halved partial slabs.
    // The stash and the per-transform states are laid out per 16-row tile / per row: independent of the workgroup shape.
    CoopPlan cf;
    int rc = coop_build_plan(pl, a.n, 1, false, &cf);
    if (rc) return rc;
    coop_make_consts(pl, cf, &k);
    return pl.KSH == 13 ? co_launch_fwd<K, 13, 1, true>(k, cf, a, st) : co_launch_fwd<K, 16, 1, true>(k, cf, a, st);
  }
  coop_make_consts(pl, cp, &k);
  if (pl.KSH == 13) return cp.NT == 2 ? co_launch_fwd<K, 13, 2, false>(k, cp, a, st) : co_launch_fwd<K, 13, 1, false>(k, cp, a, st);
  return cp.NT == 2 ? co_launch_fwd<K, 16, 2, false>(k, cp, a, st) : co_launch_fwd<K, 16, 1, false>(k, cp, a, st);
}
// sampling direction: wide nets at every batch size; narrow nets (MT = 1 instantiation of the same kernel) for the small
// calls the cooperative family takes -- above, they keep the throughput kernel nsf_flow_kernel<..., INV = true>
template <int K>
int co_inv_k(const NsfPlan& pl, const CoopPlan& cp, const float* cimg, const float* zstats, const float* noise,
             const float* x, long long n, long long x_rows, float* theta_out, float* logabsdet_out, hipStream_t st) {
  CoK k;
  CoopPlan cf = cp;
  if (cp.NT != 1) {
    int rc = coop_build_plan(pl, n, 1, false, &cf);
    if (rc) return rc;
  }
  coop_make_consts(pl, cf, &k);
  if (cp.MT == 2)
    return cow_launch_fwd<K, true, 2>(k, cf, cimg, zstats, noise, x, n, x_rows, theta_out, logabsdet_out, nullptr, nullptr, st);
  return cow_launch_fwd<K, true, 1>(k, cf, cimg, zstats, noise, x, n, x_rows, theta_out, logabsdet_out, nullptr, nullptr, st);
}
template <int K>
int co_bwd_k(const NsfPlan& pl, const CoopPlan& cp, const CoBwdArgs& a, hipStream_t st) {
  CoK k;
  coop_make_consts(pl, cp, &k);
  if (cp.MT == 2) return cp.NT == 2 ? cow_launch_bwd<K, 2>(k, cp, a, st) : cow_launch_bwd<K, 1>(k, cp, a, st);
  if (pl.KSH == 13) return cp.NT == 2 ? co_launch_bwd<K, 13, 2>(k, cp, a, st) : co_launch_bwd<K, 13, 1>(k, cp, a, st);
  return cp.NT == 2 ? co_launch_bwd<K, 16, 2>(k, cp, a, st) : co_launch_bwd<K, 16, 1>(k, cp, a, st);
}
