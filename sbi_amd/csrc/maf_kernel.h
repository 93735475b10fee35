#pragma once
// maf_kernel.h -- `maf_rqs` on gfx950: masked autoregressive flow whose transforms are rational-quadratic splines
// (sbi build_maf_rqs, /root/reference/sbi/neural_nets/net_builders/flow.py:212-330; arithmetic = nflows 0.14
// transforms/made.py + transforms/autoregressive.py + transforms/permutations.py).
//
// Execution model = the NSF kernels' (nsf_device.h): one wavefront owns 16 rows, lane = (row j, k-slot g); every
// linear of the MADE conditioner runs on v_mfma_f32_16x16x4_f32 with M = output feature, N = row, K = input
// feature, activations chained through registers (D fragment of a layer = B fragment of the next), weights read
// from the LDS image a workgroup stages once per transform.  The degree masks of MaskedLinear are static: the
// pack kernel multiplies them into the image ("MADE masked linear" = a dense GEMM on pre-masked weights), the
// weight-gradient kernel multiplies them into the gradients.
//   maf_flow_kernel<K,KSH,false>: theta, x -> log p [+ noise]          one conditioner pass per transform
//   maf_flow_kernel<K,KSH,true >: noise, x -> theta [+ logabsdet]      D passes per transform (pass i fixes dim i)
//   maf_bwd_kernel<K,KSH>       : per transform, row-parallel backward; leaves the per-row layer gradients and
//                                 activations in HBM for
//   maf_dw_kernel               : dW = G^T A as split-K (K = rows) MFMA GEMMs, masks applied, per-chunk partials
//   maf_reduce_kernel           : fixed-order sum of the partials (deterministic)
#include <hip/hip_runtime.h>
#include <math.h>
#include "nsf_train_kernel.h"   // nsf_device.h + rq_spline_pair_bwd, gemm_T_breg
#include "../../include/sbi_amd_maf.h"

#define MAF_MAX_NB 4
#define MAF_AW 64          // activation / gradient rows in HBM are 64 floats per layer
#define MAF_CW 48          // conditioner-input rows [z ; standardized context] in HBM (D <= 16, C <= 32)

struct MafPlan {
  NsfPlan n;               // dims, spline constants, shape[0] = the MADE's linears:
                           //   lin[0] initial (H x D, masked) | lin[1] context (H x C) | lin[2+b] block b (H x H, masked)
                           //   | lin[fin = 2+NB] final (D*P x H, masked)
  int l_perm, l_iperm;     // image offsets of the transform's permutation / inverse permutation (int32 bits)
  int n_layer;             // parameters per transform
  int sc_zs, sc_us, sc_gy, sc_cin, sc_pst, sc_total;   // per-wave scratch (floats)
  int PTW, DP;             // 16*PT; D*PTW: width of a padded spline-parameter-gradient row
  int variant;             // 0: nflows maf_rqs (MADE: separate context layer, tanh, degree masks, RandomPermutation)
                           // 1: zuko NSF (hyper-net on [z ; context], ReLU, adjacency masks from a buffer, fixed
                           //    autoregressive order instead of a permutation, MonotonicRQSTransform parametrisation)
};

// hidden / output degrees of nflows' MADE (made.py, random_mask=False)
__host__ __device__ __forceinline__ int maf_hidden_degree(int o, int D) {
  const int mx = D - 1 > 1 ? D - 1 : 1, mn = D - 1 < 1 ? D - 1 : 1;
  return o % mx + mn;
}
// kind: 0 initial (inputs -> hidden), 1 context (dense), 2 block (hidden -> hidden), 3 final (hidden -> outputs)
__host__ __device__ __forceinline__ bool maf_mask(int kind, int o, int c, int D, int P) {
  switch (kind) {
    case 0: return maf_hidden_degree(o, D) >= c + 1;
    case 2: return maf_hidden_degree(o, D) >= maf_hidden_degree(c, D);
    case 3: return o / P + 1 > maf_hidden_degree(c, D);
    default: return true;
  }
}

__device__ __forceinline__ float tanh_f(float x) {
  // 1 - 2 / (e^{2|x|} + 1) on the hardware exp / rcp: absolute error ~1e-7 (the activations feed dense sums)
  const float t = exp_f(-2.f * fabsf(x));
  const float r = (1.f - t) * rcp_f(1.f + t);
  return copysignf(r, x);
}

__device__ __forceinline__ void maf_pack_linear(float* __restrict__ img, const float* __restrict__ gl,
                                                const float* __restrict__ ml, const LinDesc& L,
                                                int kind, int D, int P, int bias_pad, int bias_group,
                                                int bias_group_pad, int tid, int nthreads) {
  const int total = L.rows * L.ldk;
  for (int idx = tid; idx < total; idx += nthreads) {
    const int r = idx / L.ldk, c = idx - r * L.ldk;
    float v = 0.f;
    if (r < L.out && c < L.in) {
      const bool keep = ml ? (ml[L.g_w + r * L.in + c] != 0.f) : maf_mask(kind, r, c, D, P);
      if (keep) v = gl[L.g_w + r * L.in + c];
    }
    img[L.l_w + idx] = v;
  }
  for (int idx = tid; idx < bias_pad; idx += nthreads) {
    const int grp = idx / bias_group_pad, p = idx - grp * bias_group_pad;
    const int src = grp * bias_group + p;
    img[L.l_b + idx] = (p < bias_group && src < L.out) ? gl[L.g_b + src] : 0.f;
  }
}

#ifdef MAF_MAIN_TU   // non-template kernels: defined by maf.hip only
__global__ void __launch_bounds__(256)
maf_pack_kernel(const MafPlan mp, const float* __restrict__ params, const int* __restrict__ perms,
                const float* __restrict__ masks, float* __restrict__ packed) {
  const NsfPlan& pl = mp.n;
  const ShapeDesc& S = pl.shape[0];
  const int t = blockIdx.x;
  float* img = packed + (long long)t * pl.img_floats;
  const float* gl = params + (long long)t * mp.n_layer;
  const float* ml = masks ? masks + (long long)t * mp.n_layer : nullptr;
  const int tid = blockIdx.y * blockDim.x + threadIdx.x, nthreads = gridDim.y * blockDim.x;
  const int hb = 16 * NSF_HT;
  maf_pack_linear(img, gl, ml, S.lin[0], 0, pl.D, pl.P, hb, hb, hb, tid, nthreads);
  if (mp.variant == 0) maf_pack_linear(img, gl, ml, S.lin[1], 1, pl.D, pl.P, hb, hb, hb, tid, nthreads);
  for (int b = 0; b < pl.NB; ++b) maf_pack_linear(img, gl, ml, S.lin[2 + b], 2, pl.D, pl.P, hb, hb, hb, tid, nthreads);
  maf_pack_linear(img, gl, ml, S.lin[S.fin], 3, pl.D, pl.P, pl.D * 16 * pl.PT, pl.P, 16 * pl.PT, tid, nthreads);
  for (int d = tid; d < pl.D; d += nthreads) {
    const int p = perms[t * pl.D + d];
    img[mp.l_perm + d] = __int_as_float(p);
    img[mp.l_iperm + p] = __int_as_float(d);
  }
  for (int idx = mp.l_iperm + 16 + tid; idx < pl.img_floats; idx += nthreads) img[idx] = 0.f;
}

#endif

// MADE hidden stack: h = tanh(W0m z + b0 + gate), then NB x h = tanh(Wb h + bb); gate = tanh(Wc c + bc) is
// computed once per transform by the caller.  `hs` (optional) receives the activation after every layer.
template <int VAR>
__device__ __forceinline__ float maf_act(float x) { return VAR == 0 ? tanh_f(x) : fmaxf(x, 0.f); }
// derivative of the activation expressed through its OUTPUT h
template <int VAR>
__device__ __forceinline__ float maf_act_grad(float h) { return VAR == 0 ? 1.f - h * h : (h > 0.f ? 1.f : 0.f); }

template <int KSH, int VAR>
__device__ __forceinline__ void made_hidden(const float* __restrict__ lds, const NsfPlan& pl, const ShapeDesc& S,
                                            const LaneId& id, const float* __restrict__ cin_row,
                                            const f4 (&gate)[NSF_HT], f4 (&h)[NSF_HT],
                                            f4 (*hs)[NSF_HT] = nullptr) {
  acc_init_bias(lds, S.lin[0], id, h);
  gemm_blds(lds, S.lin[0], id, cin_row, h);   // VAR 1: the layer's input is [z ; context] (one masked linear)
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[mt][r] = maf_act<VAR>(VAR == 0 ? h[mt][r] + gate[mt][r] : h[mt][r]);
  if (hs) {
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt) hs[0][mt] = h[mt];
  }
  // compile-time block index (the activation array must stay in registers): guarded unroll over the maximum
#pragma unroll
  for (int b = 0; b < MAF_MAX_NB; ++b) {
    if (b < pl.NB) {
      f4 u[NSF_HT];
      acc_init_bias(lds, S.lin[2 + b], id, u);
      gemm_breg<KSH>(lds, S.lin[2 + b], id, h, u);
#pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[mt][r] = maf_act<VAR>(u[mt][r]);
      if (hs) {
#pragma unroll
        for (int mt = 0; mt < NSF_HT; ++mt) hs[b + 1][mt] = h[mt];
      }
    }
  }
}

template <int KSH>
__device__ __forceinline__ void made_gate(const float* __restrict__ lds, const ShapeDesc& S, const LaneId& id,
                                          const float* __restrict__ ctx_row, f4 (&gate)[NSF_HT]) {
  acc_init_bias(lds, S.lin[1], id, gate);
  gemm_blds(lds, S.lin[1], id, ctx_row, gate);
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) gate[mt][r] = tanh_f(gate[mt][r]);
}

// RandomPermutation on the wave's state rows: zs[j][d] <- zs[j][idx[d]]
__device__ __forceinline__ void permute_rows(float* __restrict__ zs, int ZW, int D, const float* __restrict__ idx_f,
                                             const LaneId& id) {
  float v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int d = id.g + 4 * u;
    v[u] = d < D ? zs[id.j * ZW + __float_as_int(idx_f[d])] : 0.f;
  }
  wave_lds_fence();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int d = id.g + 4 * u;
    if (d < D) zs[id.j * ZW + d] = v[u];
  }
  wave_lds_fence();
}

template <int K, int KSH, bool INV, int VAR>
__global__ void __launch_bounds__(512)
maf_flow_kernel(const MafPlan mp, const float* __restrict__ packed, const float* __restrict__ zstats,
                const float* __restrict__ in, const float* __restrict__ x, long long n, long long x_rows,
                float* __restrict__ out_main, float* __restrict__ out_aux, float* __restrict__ z_stash) {
  constexpr int PT = (3 * K - 1 + 15) / 16;
  const NsfPlan& pl = mp.n;
  const ShapeDesc& S = pl.shape[0];
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int wave = tid >> 6, nw = nthreads >> 6;
  const LaneId id = make_lane();
  float* sc = lds + pl.lds_w_floats + wave * mp.sc_total;
  float* zs = sc + mp.sc_zs;
  float* us = sc + mp.sc_us;
  float* cin = sc + mp.sc_cin;
  float* pst = sc + mp.sc_pst;
  const long long row = (long long)blockIdx.x * (16 * nw) + 16 * wave + id.j;
  const bool valid = row < n;
  const int D = pl.D, C = pl.C;
  const float* th_shift = zstats;
  const float* th_scale = zstats + D;
  const float* x_mean = zstats + 2 * D;
  const float* x_std = x_mean + C;
  float ld_acc = 0.f;
  for (int i = id.lane; i < mp.sc_total; i += 64) sc[i] = 0.f;
  {
    const long long xr = (x_rows == n) ? row : (x_rows == 1 ? 0 : row % x_rows);
    for (int d = id.g; d < D; d += 4) {
      float v = valid ? in[row * D + d] : 0.f;
      if (!INV) {
        v = v * th_scale[d] + th_shift[d];
        ld_acc += logf(fabsf(th_scale[d]));
      }
      zs[id.j * pl.ZW + d] = v;
    }
    // standardized context: constant over the transforms, lives at columns [D, D + C) of the conditioner input
    for (int c = id.g; c < C; c += 4)
      cin[id.j * pl.CINW + D + c] = ((valid ? x[xr * C + c] : 0.f) - x_mean[c]) / x_std[c];
  }
  wave_lds_fence();
  const float* cin_row = cin + id.j * pl.CINW + id.g;

  for (int li = 0; li < pl.T; ++li) {
    const int t = INV ? (pl.T - 1 - li) : li;
    __syncthreads();
    stage_layer(lds, packed + (long long)t * pl.img_floats, pl.img_floats, tid, nthreads);
    __syncthreads();
    if (!INV && z_stash) {
      for (int d = id.g; d < D; d += 4)
        if (valid) z_stash[((long long)t * n + row) * D + d] = zs[id.j * pl.ZW + d];
    }
    f4 gate[NSF_HT], h[NSF_HT];
    if (VAR == 0) made_gate<KSH>(lds, S, id, cin_row + D, gate);
    if (!INV) {
      for (int k = id.g; k < D; k += 4) cin[id.j * pl.CINW + k] = zs[id.j * pl.ZW + k];
      wave_lds_fence();
      made_hidden<KSH, VAR>(lds, pl, S, id, cin_row, gate, h);
      // all D dims are transformed, parameters from the ORIGINAL inputs (already folded into h)
      const int nchunks = (D + 1) / 2;
      for (int c = 0; c < nchunks; ++c) {
        if (D - 2 * c >= 2) final_layer_chunk_n<PT, KSH, 2>(lds, pst, pl, S, id, h, 2 * c);
        else final_layer_chunk_n<PT, KSH, 1>(lds, pst, pl, S, id, h, 2 * c);
        wave_lds_fence();
        const int slot = id.g & 1, part = id.g >> 1;
        const bool live = 2 * c + slot < D;
        const int sl = live ? slot : 0;
        const int dd = 2 * c + sl;
        const int zi = id.j * pl.ZW + dd;
        float y, ld;
        rq_spline_pair<K, false, NoYield, VAR>(pst + sl * pl.DS + id.j * pl.PSW, zs[zi], pl, part, y, ld);
        zs[zi] = y;   // idle lanes recompute slot 0's task and store the same value
        ld_acc += (live && part == 0) ? ld : 0.f;
        wave_lds_fence();
      }
      if (VAR == 0) permute_rows(zs, pl.ZW, D, lds + mp.l_perm, id);
    } else {
      // inverse of the permutation that FOLLOWS the transform (zuko: none, the order lives in the masks)
      if (VAR == 0) permute_rows(zs, pl.ZW, D, lds + mp.l_iperm, id);
      for (int k = id.g; k < D; k += 4) cin[id.j * pl.CINW + k] = 0.f;
      wave_lds_fence();
      // autoregressive inverse (autoregressive.py: D passes from zeros): pass i sees the exact outputs of the
      // dims < i, which is all dim i depends on, so only dim i's parameters and spline are evaluated
      for (int pass = 0; pass < D; ++pass) {
        // maf_rqs: natural order (degrees 1..D); zuko: the feature whose autoregressive order is `pass`
        const int i = VAR == 0 ? pass : __float_as_int(lds[mp.l_iperm + pass]);
        made_hidden<KSH, VAR>(lds, pl, S, id, cin_row, gate, h);
        final_layer_chunk_n<PT, KSH, 1>(lds, pst, pl, S, id, h, i);
        wave_lds_fence();
        const int part = id.g >> 1;
        float y, ld;
        rq_spline_pair<K, true, NoYield, VAR>(pst + id.j * pl.PSW, zs[id.j * pl.ZW + i], pl, part, y, ld);
        cin[id.j * pl.CINW + i] = y;
        us[id.j * pl.ZW + i] = y;
        ld_acc += (id.g == 0) ? ld : 0.f;
        wave_lds_fence();
      }
      for (int k = id.g; k < D; k += 4) zs[id.j * pl.ZW + k] = us[id.j * pl.ZW + k];
      wave_lds_fence();
    }
  }

  if (!INV) {
    float part = 0.f;
    for (int d = id.g; d < D; d += 4) {
      const float z = zs[id.j * pl.ZW + d];
      part += z * z;
      if (out_aux && valid) out_aux[row * D + d] = z;
    }
    float v = -0.5f * part + ld_acc;
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (id.g == 0 && valid) out_main[row] = v - pl.log_z;
  } else {
    for (int d = id.g; d < D; d += 4) {
      const float z = zs[id.j * pl.ZW + d];
      ld_acc -= logf(fabsf(th_scale[d]));
      if (valid) out_main[row * D + d] = (z - th_shift[d]) / th_scale[d];
    }
    float v = ld_acc;
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (out_aux && id.g == 0 && valid) out_aux[row] = v;
  }
}

// ------------------------------------------------------------------ training: row-parallel backward
struct MafBwdArgs {
  const float* packed;
  const float* zstats;
  const float* z_in;       // (n, D) input state of this transform (forward stash)
  const float* x;
  const float* gz_up;      // (n, D) gradient wrt this transform's output (after the permutation); the LAST
                           // transform receives the flow output z_T instead (d/dz of w 0.5|z|^2 = w z)
  const float* row_w;
  float uni_w;
  long long n, x_rows;
  float* gz_dn;            // (n, D) gradient wrt this transform's input (t > 0)
  float* grad_theta;       // optional (n, D), written by t == 0
  // gradient operands of the weight-gradient GEMMs are stored M-TILE MAJOR: column c of row r lives at
  // plane (c >> 4) * npad * 16 + r * 16 + (c & 15), so that maf_dw_kernel streams a (128 rows x 16 columns) tile
  // as one contiguous 8 KB block (within a 16-block the hidden layers' planes are in fragment order, MafLin.gperm)
  float* GP;               // DP / 16 planes: gradient wrt the raw spline parameters, 16*PT columns per dim
  float* ACT;              // (n, (NB+1)*64) h_0 .. h_NB, row major (inputs of the GEMMs)
  float* G;                // (NB+2) x 4 planes; slot 0: d/d(a1), slot 1: d/d(context pre-activation), 2+b: block b
  long long npad;          // rows padded to whole dW chunks
  float* CTX;              // (n, MAF_CW) conditioner input rows [z ; standardized context]
  int t, is_last;
};

// g_h += Wf[rows of dims d0 .. d0+1]^T g_p, g_p read from the wave's spline-parameter staging buffer
template <int PT>
__device__ __forceinline__ void maf_wft_chunk(const float* __restrict__ lds, const LinDesc& LF, const NsfPlan& pl,
                                              const LaneId& id, const float* __restrict__ pst, int d0, int nact,
                                              f4 (&gh)[NSF_HT]) {
  constexpr int KS = 4 * PT;
  int col[NSF_HT];
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt) {
    const int f = 16 * mt + id.iperm;
    col[mt] = f < LF.in ? f : 0;
  }
  const int kstride = 4 * LF.ldk;
  for (int sl = 0; sl < nact; ++sl) {
    const float* wrow = lds + LF.l_w + ((d0 + sl) * pl.P + id.g) * LF.ldk;
    const float* brow = pst + sl * pl.DS + id.j * pl.PSW + id.g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float bv = brow[4 * s];
#pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt) gh[mt] = MFMA16(wrow[col[mt] + s * kstride], bv, gh[mt]);
    }
  }
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) gh[mt][r] = (16 * mt + 4 * r + id.g < LF.in) ? gh[mt][r] : 0.f;
}

// D fragments -> one 64-float row segment per batch row in HBM, in FRAGMENT ORDER: lane (row j, slot g) holds
// features 16 mt + 4 r + g in register r and stores them as ONE 16-byte word at position 16 mt + 4 g, i.e. position
// 4 g + r of a 16-block holds feature 4 r + g (a natural-order store would be 16 scattered 4-byte stores per lane).
// The weight-gradient kernel undoes the permutation in its index arithmetic (MafLin.aperm / gperm); ld % 4 == 0.
__device__ __forceinline__ void store_frag_rows(float* __restrict__ dst, int ld, long long row, bool valid,
                                                const LaneId& id, const f4 (&v)[NSF_HT]) {
  if (!valid) return;
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt)
    *reinterpret_cast<float4*>(dst + row * ld + 16 * mt + 4 * id.g) = float4{v[mt][0], v[mt][1], v[mt][2], v[mt][3]};
}

// D fragments -> four m-tile planes (tile-major gradient operand, see MafBwdArgs), fragment order inside a tile
__device__ __forceinline__ void store_frag_planes(float* __restrict__ plane0, long long npad, long long row, bool valid,
                                                  const LaneId& id, const f4 (&v)[NSF_HT]) {
  if (!valid) return;
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt)
    *reinterpret_cast<float4*>(plane0 + (mt * npad + row) * 16 + 4 * id.g) =
        float4{v[mt][0], v[mt][1], v[mt][2], v[mt][3]};
}

// spline-parameter gradients of `nact` dims (LDS rows of 16 PT floats) -> planes of GP in natural column order:
// lane (row j, slot g) moves columns 4 g .. 4 g + 3 of every 16-block as one 16-byte store
template <int PT>
__device__ __forceinline__ void store_param_planes(float* __restrict__ GP, long long npad, long long row, bool valid,
                                                   const LaneId& id, const float* __restrict__ pst, int DS, int PSW,
                                                   int dim0, int nact) {
  if (!valid) return;
  for (int sl = 0; sl < nact; ++sl) {
    const float* src = pst + sl * DS + id.j * PSW + 4 * id.g;
#pragma unroll
    for (int h = 0; h < PT; ++h)
      *reinterpret_cast<float4*>(GP + (((long long)(dim0 + sl) * PT + h) * npad + row) * 16 + 4 * id.g) =
          float4{src[16 * h], src[16 * h + 1], src[16 * h + 2], src[16 * h + 3]};
  }
}

template <int K, int KSH, int VAR>
__global__ void __launch_bounds__(512)
maf_bwd_kernel(const MafPlan mp, const MafBwdArgs a) {
  constexpr int PT = (3 * K - 1 + 15) / 16;
  const NsfPlan& pl = mp.n;
  const ShapeDesc& S = pl.shape[0];
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int wave = tid >> 6, nw = nthreads >> 6;
  const LaneId id = make_lane();
  float* sc = lds + pl.lds_w_floats + wave * mp.sc_total;
  float* zs = sc + mp.sc_zs;
  float* gxs = sc + mp.sc_us;      // direct-path gradient wrt the transform input (through the spline argument)
  float* gys = sc + mp.sc_gy;      // gradient wrt the spline outputs
  float* cin = sc + mp.sc_cin;
  float* pst = sc + mp.sc_pst;
  const int D = pl.D, C = pl.C, NB = pl.NB;
  const long long n = a.n;
  const long long row = (long long)blockIdx.x * (16 * nw) + 16 * wave + id.j;
  const bool valid = row < n;
  const long long rs = valid ? row : 0;
  const float* x_mean = a.zstats + 2 * D;
  const float* x_std = x_mean + C;
  const float wn = valid ? (a.row_w ? a.row_w[rs] : a.uni_w) : 0.f;
  const float gld = -wn;   // d(sum w loss) / d(any log|det| term)
  stage_layer(lds, a.packed + (long long)a.t * pl.img_floats, pl.img_floats, tid, nthreads);
  for (int i = id.lane; i < mp.sc_total; i += 64) sc[i] = 0.f;
  __syncthreads();
  {
    const long long xr = (a.x_rows == n) ? rs : (a.x_rows == 1 ? 0 : rs % a.x_rows);
    for (int d = id.g; d < D; d += 4) {
      const float z = valid ? a.z_in[rs * D + d] : 0.f;
      zs[id.j * pl.ZW + d] = z;
      cin[id.j * pl.CINW + d] = z;
      if (valid) a.CTX[row * MAF_CW + d] = z;
      // undo the permutation on the way back: out[d] = in[perm[d]]  =>  g_in[k] = g_out[iperm[k]]
      const int src = VAR == 0 ? __float_as_int(lds[mp.l_iperm + d]) : d;
      const float g = valid ? a.gz_up[rs * D + src] : 0.f;
      gys[id.j * pl.ZW + d] = a.is_last ? wn * g : g;
    }
    for (int c = id.g; c < C; c += 4) {
      const float v = ((valid ? a.x[xr * C + c] : 0.f) - x_mean[c]) / x_std[c];
      cin[id.j * pl.CINW + D + c] = v;
      if (valid) a.CTX[row * MAF_CW + D + c] = v;
    }
  }
  wave_lds_fence();
  const float* cin_row = cin + id.j * pl.CINW + id.g;
  // ---- recompute the conditioner (activations stay in registers)
  f4 gate[NSF_HT], h[NSF_HT], hs[MAF_MAX_NB + 1][NSF_HT];
  if (VAR == 0) made_gate<KSH>(lds, S, id, cin_row + D, gate);
  made_hidden<KSH, VAR>(lds, pl, S, id, cin_row, gate, h, hs);
#pragma unroll
  for (int b = 0; b <= MAF_MAX_NB; ++b)
    if (b <= NB) store_frag_rows(a.ACT + 64 * b, (MAF_MAX_NB + 1) * MAF_AW, row, valid, id, hs[b]);
  // ---- spline forward + reverse mode per chunk of two dims; g_h = Wf^T g_p accumulated on the fly
  f4 gh[NSF_HT];
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt) gh[mt] = {0.f, 0.f, 0.f, 0.f};
  const LinDesc& LF = S.lin[S.fin];
  const int nchunks = (D + 1) / 2;
  for (int c = 0; c < nchunks; ++c) {
    const int nact = D - 2 * c >= 2 ? 2 : 1;
    if (nact == 2) final_layer_chunk_n<PT, KSH, 2>(lds, pst, pl, S, id, h, 2 * c);
    else final_layer_chunk_n<PT, KSH, 1>(lds, pst, pl, S, id, h, 2 * c);
    wave_lds_fence();
    const int slot = id.g & 1, part = id.g >> 1;
    const int dd = 2 * c + slot;
    if (dd < D) {     // both lanes of a pair (lane, lane ^ 32) share the slot: the exchange inside stays convergent
      float* pp = pst + slot * pl.DS + id.j * pl.PSW;
      const int zi = id.j * pl.ZW + dd;
      float yv, gxv;
      rq_spline_pair_bwd<K, VAR>(pp, mp.PTW, zs[zi], gys[zi], gld, pl, part, yv, gxv);
      if (part == 0) gxs[zi] = gxv;
    }
    wave_lds_fence();
    // g_p rows -> HBM (operand of d Wf), 16*PT floats per (row, dim)
    store_param_planes<PT>(a.GP, a.npad, row, valid, id, pst, pl.DS, pl.PSW, 2 * c, nact);
    maf_wft_chunk<PT>(lds, LF, pl, id, pst, 2 * c, nact, gh);
    wave_lds_fence();
  }
  // ---- back through the feed-forward blocks: G_b = g (1 - h_{b+1}^2), g <- W_b^T G_b
#pragma unroll
  for (int b = MAF_MAX_NB - 1; b >= 0; --b) {
    if (b < NB) {
      f4 gb[NSF_HT];
#pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float hv = hs[b + 1][mt][r];
          gb[mt][r] = gh[mt][r] * maf_act_grad<VAR>(hv);
          gh[mt][r] = 0.f;
        }
      store_frag_planes(a.G + (long long)(4 * (2 + b)) * a.npad * 16, a.npad, row, valid, id, gb);
      gemm_T_breg<KSH, NSF_HT>(lds, S.lin[2 + b], id, gb, gh);
    }
  }
  {
    f4 g0[NSF_HT], gc[NSF_HT], gin[1];
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = hs[0][mt][r];
        g0[mt][r] = gh[mt][r] * maf_act_grad<VAR>(hv);    // d / d(W0 z + b0 + gate)
        if (VAR == 0) {
          const float gt = gate[mt][r];
          gc[mt][r] = g0[mt][r] * (1.f - gt * gt);        // d / d(Wc c + bc)
        }
      }
    store_frag_planes(a.G, a.npad, row, valid, id, g0);
    if (VAR == 0) store_frag_planes(a.G + 4 * a.npad * 16, a.npad, row, valid, id, gc);
    gin[0] = {0.f, 0.f, 0.f, 0.f};
    gemm_T_breg<KSH, 1>(lds, S.lin[0], id, g0, gin);     // through the (masked) initial layer: dims < their own
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 4 * r + id.g;
      if (k < D && valid) {
        const float g = gxs[id.j * pl.ZW + k] + gin[0][r];
        if (a.t > 0) a.gz_dn[row * D + k] = g;
        else if (a.grad_theta) a.grad_theta[row * D + k] = g * a.zstats[D + k];
      }
    }
  }
}

// ------------------------------------------------------------------ training: weight gradients
struct MafLin {
  const float* G;   // m-tile-major gradient wrt the layer's outputs: plane t = columns [16 t, 16 t + 16) of all rows
  const float* A;   // (n, lda): per-row inputs of the layer
  long long gts;    // floats between consecutive planes (= padded rows x 16)
  int lda;
  int out, in;      // natural dims (`in` <= 64: a wider input is handed over as several column pieces)
  int in_total, col0;   // row length of the weight in the parameter block and first column of this piece
  int group, group_pad;   // output o lives at column (o / group) * group_pad + o % group  (final layer: P -> 16*PT)
  int g_w, g_b;     // offsets inside the transform's parameter block
  int kind;         // mask kind (maf_mask) when no mask buffer is given
  int gperm, aperm; // G tiles / A rows are in fragment order (store_frag_planes / store_frag_rows) instead of natural
  int mt0, mtn;     // m-tile range of this piece (filled in by maf_launch_dw: at most 4 * MAF_DW_MTW tiles each)
};
#define MAF_DW_MAX_LIN 40
struct MafDwArgs {
  MafLin lin[MAF_DW_MAX_LIN];
  long long n;
  int rows_per_chunk, nchunks, n_layer, D, P;
  float* partial;   // (nchunks, n_layer) for this transform
  const float* mask;   // optional (n_layer) 0/1 mask of this transform's parameter block (zuko adjacency masks)
  int abl;             // timing experiments only (tools/ubench/maf_dw_bench.hip): 1 no MFMA sweep, 2 no G tile load,
                       // 4 no input staging, 8 no write-out; 0 in the library
};

#ifndef MAF_DW_ROWS
#define MAF_DW_ROWS 128    // rows per sub-chunk staged in LDS (multiple of 64)
#endif
#ifndef MAF_DW_SUB
#define MAF_DW_SUB 4       // sub-chunks per workgroup: the accumulators persist, ONE partial slab per MAF_DW_CHUNK rows
#endif
#define MAF_DW_CHUNK (MAF_DW_ROWS * MAF_DW_SUB)
#define MAF_DW_MTW 4       // m-tiles per wave and piece (their accumulators live in registers across the sub-chunks)
#define MAF_DW_TV (MAF_DW_ROWS / 16)   // float4 per lane of one 16-column G tile
#define MAF_DW_AV (MAF_DW_ROWS / 4)    // staged input values per thread
#define MAF_DW_SA 68       // LDS row stride of the staged input tile: rows 4 apart sit 16 banks apart (ds_read_b32: 32 banks)
#define MAF_DW_GS 20       // row stride of a wave's 16-column G tile (4 * 20 = 16 mod 32 as well)
#define MAF_DW_LDS_BYTES ((MAF_DW_ROWS * MAF_DW_SA + 4 * MAF_DW_ROWS * MAF_DW_GS) * 4)
template <int NT>
__device__ __forceinline__ void maf_dw_sweep(const float* __restrict__ ap, const float* __restrict__ bp,
                                             f4 (&acc)[4]) {
  constexpr int KS = MAF_DW_ROWS / 4;
#pragma unroll 16
  for (int s = 0; s < KS; ++s) {
    const int kr = 16 * (s >> 2) + (s & 3);
    const float a_ = ap[kr * MAF_DW_GS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = MFMA16(a_, bp[kr * MAF_DW_SA + 16 * nt], acc[nt]);
  }
}

#ifdef MAF_MAIN_TU
// dW = G^T A of one linear (one piece of <= 16 m-tiles of it) over MAF_DW_SUB sub-chunks of MAF_DW_ROWS rows.  Per
// sub-chunk the input rows A (<= 64 columns) are staged in LDS by the whole workgroup, then every wave contracts its
// m-tiles: a tile of G (ROWS x 16) goes through the wave's private LDS region with coalesced 16-byte loads.  Nothing
// waits on HBM in steady state: the NEXT G tile (next m-tile, or the first one of the next sub-chunk) and the NEXT
// sub-chunk's input rows are requested before the current tile's MFMA sweep and land under it; two workgroups per CU
// (76 KB of LDS each) cover what is left.  The accumulators of all the wave's m-tiles stay in registers across the
// sub-chunks, so one partial slab is written per 512 rows.  K-step s covers rows krow(s) + 4 g (the conflict-free
// assignment of the NSF backward kernel's dw_gemm).  Bias gradients (column sums of G): VALU adds on the tile's
// registers on their way into LDS, one cross-lane reduction at the end.
// Masks are applied on the way out; partial slabs are summed by maf_reduce_kernel.
__device__ __forceinline__ void maf_dw_load_tile(const MafLin& L, long long r0, int mt, int lane, float4 (&v)[MAF_DW_TV]) {
  // four lanes per row (lane l of load `it` holds row (64 it + l) / 4, columns 4 (l & 3) ..); the planes are padded
  // to whole chunks, so every load is in bounds
  const float4* gsrc = reinterpret_cast<const float4*>(L.G + (long long)(L.mt0 + mt) * L.gts + r0 * 16);
#pragma unroll
  for (int it = 0; it < MAF_DW_TV; ++it) v[it] = gsrc[it * 64 + lane];
}
// thread = (column c, row phase): rows phase + 4 u (clamped addresses + selects at the LDS store: a predicated load
// would sit in its own basic block with its own s_waitcnt)
__device__ __forceinline__ void maf_dw_load_rows(const MafLin& L, long long r0, int tid, float (&t)[MAF_DW_AV]) {
  // (fragment-ordered rows: every position of the 16-blocks up to round_up(in, 16) exists and holds a finite value)
  const int c = tid & 63, cc = c < (L.aperm ? ((L.in + 15) & ~15) : L.in) ? c : 0;
  const float* ap0 = L.A + (r0 + (tid >> 6)) * L.lda + cc;
#pragma unroll
  for (int u = 0; u < MAF_DW_AV; ++u) t[u] = ap0[(long long)(4 * u) * L.lda];
}
// orders this wave's LDS writes before its LDS reads WITHOUT draining the vector-memory counter (the prefetches in
// flight must stay in flight): DS ops of one wave execute in issue order, the asm only pins the compiler
__device__ __forceinline__ void maf_dw_lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

__global__ void __launch_bounds__(256, 2)
maf_dw_kernel(const MafDwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float As[];
  const MafLin& L = a.lin[blockIdx.y];
  const int chunk = blockIdx.x;
  const long long R0 = (long long)chunk * MAF_DW_CHUNK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int mtn = L.mtn;
  const int ntiles = (L.in + 15) / 16;
  long long left = a.n - R0;
  int nsub = left <= 0 ? 0 : (int)((left + MAF_DW_ROWS - 1) / MAF_DW_ROWS);
  if (nsub > MAF_DW_SUB) nsub = MAF_DW_SUB;
  float4 v[MAF_DW_TV];
  float ta[MAF_DW_AV];
  const bool g_on = !(a.abl & 2), a_on = !(a.abl & 4);
  if (nsub > 0) {
    if (wave < mtn && g_on) maf_dw_load_tile(L, R0, wave, lane, v);
    if (a_on) maf_dw_load_rows(L, R0, tid, ta);
  }
  f4 acc[MAF_DW_MTW][4];
  float4 accb[MAF_DW_MTW];      // per lane: sums of columns 4 (lane & 3) .. + 3 over the rows this lane loaded
#pragma unroll
  for (int q = 0; q < MAF_DW_MTW; ++q) {
    accb[q] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[q][nt] = {0.f, 0.f, 0.f, 0.f};
  }
  float* Gs = As + MAF_DW_ROWS * MAF_DW_SA + wave * (MAF_DW_ROWS * MAF_DW_GS);
  const float* ap = Gs + 4 * g * MAF_DW_GS + j;
  const float* bp = As + 4 * g * MAF_DW_SA + j;
  for (int sc = 0; sc < nsub; ++sc) {
    const long long r0 = R0 + (long long)sc * MAF_DW_ROWS;
    const long long rem = a.n - r0;
    const int nrows = rem < MAF_DW_ROWS ? (int)rem : MAF_DW_ROWS;
    if (sc > 0) __syncthreads();          // every wave is done reading the previous sub-chunk's input tile
    if (a_on) {
      const int c = tid & 63;
      const int cfeat = L.aperm ? (c & 48) + 4 * (c & 3) + ((c >> 2) & 3) : c;   // feature held by column position c
#pragma unroll
      for (int u = 0; u < MAF_DW_AV; ++u) {
        const int r = (tid >> 6) + 4 * u;
        As[r * MAF_DW_SA + c] = (r < nrows && cfeat < L.in) ? ta[u] : 0.f;
      }
    }
    __syncthreads();
    const bool more = sc + 1 < nsub;
#pragma unroll
    for (int q = 0; q < MAF_DW_MTW; ++q) {
      const int mt = wave + 4 * q;
      if (mt < mtn) {
        if (g_on) {
#pragma unroll
          for (int it = 0; it < MAF_DW_TV; ++it) {
            const int idx = it * 64 + lane, row = idx >> 2;
            float4 val = v[it];       // (component selects: a ?: between two float4 lvalues keeps v[] in scratch)
            const bool in = row < nrows;
            val.x = in ? val.x : 0.f; val.y = in ? val.y : 0.f; val.z = in ? val.z : 0.f; val.w = in ? val.w : 0.f;
            *reinterpret_cast<float4*>(Gs + row * MAF_DW_GS + 4 * (idx & 3)) = val;
            accb[q].x += val.x; accb[q].y += val.y; accb[q].z += val.z; accb[q].w += val.w;
          }
        }
        maf_dw_lds_fence();
        // requests that fly under this sweep: the next G tile and (once per sub-chunk) the next input rows
        if (g_on) {
          if (mt + 4 < mtn) maf_dw_load_tile(L, r0, mt + 4, lane, v);
          else if (more) maf_dw_load_tile(L, r0 + MAF_DW_ROWS, wave, lane, v);
        }
        if (q == 0 && more && a_on) maf_dw_load_rows(L, r0 + MAF_DW_ROWS, tid, ta);
        // one guard-free MFMA stream per input width (a guard per MFMA costs a basic block and an s_waitcnt each)
        if (!(a.abl & 1))
        switch (ntiles) {
          case 1: maf_dw_sweep<1>(ap, bp, acc[q]); break;
          case 2: maf_dw_sweep<2>(ap, bp, acc[q]); break;
          case 3: maf_dw_sweep<3>(ap, bp, acc[q]); break;
          default: maf_dw_sweep<4>(ap, bp, acc[q]); break;
        }
        maf_dw_lds_fence();
      }
    }
    // (a wave without m-tiles in this piece still stages input rows; its row prefetch rides on q == 0 above only when
    // it owns a tile, so give it the same request here)
    if (wave >= mtn && more && a_on) maf_dw_load_rows(L, r0 + MAF_DW_ROWS, tid, ta);
  }
  if (a.abl & 8) return;
  float* part = a.partial + (long long)chunk * a.n_layer;
  // mask inputs that do not depend on the m-tile: the degree of this lane's input column in every n-tile
  int deg_in[4];
  const int ji = L.aperm ? 4 * (j & 3) + (j >> 2) : j;     // input feature (inside its 16-block) of B column j
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) deg_in[nt] = L.kind == 0 ? 16 * nt + ji + 1 : maf_hidden_degree(16 * nt + ji, a.D);
#pragma unroll
  for (int q = 0; q < MAF_DW_MTW; ++q) {
    const int mt = wave + 4 * q;
    if (mt >= mtn) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // padded output column of this accumulator row (tile position 4 g + r)
      const int m = 16 * (L.mt0 + mt) + (L.gperm ? 4 * r + g : 4 * g + r);
      const int grp = m / L.group_pad, p = m - grp * L.group_pad;
      const int o = grp * L.group + p;
      if (p < L.group && o < L.out) {
        const int deg_out = L.kind == 3 ? grp + 1 : maf_hidden_degree(o, a.D);   // kind 3: o / P + 1 with group = P
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int i = 16 * nt + ji;
          if (nt < ntiles && i < L.in) {
            const int widx = L.g_w + o * L.in_total + L.col0 + i;
            bool keep;
            if (a.mask) keep = a.mask[widx] != 0.f;
            else keep = L.kind == 1 ? true : (L.kind == 3 ? deg_out > deg_in[nt] : deg_out >= deg_in[nt]);
            part[widx] = keep ? acc[q][nt][r] : 0.f;
          }
        }
      }
    }
    if (L.col0 == 0) {
      // column sums: lanes with equal (lane & 3) hold the same four columns for different rows
      float bs[4] = {accb[q].x, accb[q].y, accb[q].z, accb[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) bs[e] += __shfl_xor(bs[e], off, 64);
      }
      if (lane < 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = 16 * (L.mt0 + mt) + (L.gperm ? 4 * e + lane : 4 * lane + e);
          const int grp = m / L.group_pad, p = m - grp * L.group_pad;
          const int o = grp * L.group + p;
          if (p < L.group && o < L.out) part[L.g_b + o] = bs[e];
        }
      }
    }
  }
}

// grad[t][idx] = sum over chunks (fixed order, 4-way ILP)
__global__ void __launch_bounds__(256)
maf_reduce_kernel(const float* __restrict__ partial, float* __restrict__ grad, int n_layer, int nchunks, int T) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)T * n_layer) return;
  const int t = (int)(idx / n_layer);
  const int li = (int)(idx - (long long)t * n_layer);
  const float* base = partial + (long long)t * nchunks * n_layer + li;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  int c = 0;
  for (; c + 3 < nchunks; c += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += base[(long long)(c + u) * n_layer];
  }
  for (; c < nchunks; ++c) s[0] += base[(long long)c * n_layer];
  grad[idx] = (s[0] + s[1]) + (s[2] + s[3]);
}

#endif

// defined in maf.hip (the translation unit that owns maf_dw_kernel / maf_reduce_kernel)
int maf_launch_dw(const MafDwArgs& d, int nlin, hipStream_t st);
int maf_launch_reduce(const float* partial, float* out, int n_layer, int nchunks, int T, hipStream_t st);

// ------------------------------------------------------------------ per-K launchers (instantiated per TU)
template <int K, int KSH, bool INV, int VAR>
static int maf_launch_flow(const MafPlan& mp, int nw, const float* packed, const float* zstats, const float* in,
                           const float* x, int64_t n, int64_t x_rows, float* out_main, float* out_aux,
                           float* z_stash, hipStream_t st) {
  const int lds_bytes = 4 * (mp.n.lds_w_floats + nw * mp.sc_total);
  auto kern = maf_flow_kernel<K, KSH, INV, VAR>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return (int)e;
  const int64_t grid = (n + 16 * nw - 1) / (16 * nw);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * nw), (size_t)lds_bytes, st, mp, packed, zstats, in, x,
                     (long long)n, (long long)x_rows, out_main, out_aux, z_stash);
  return (int)hipGetLastError();
}

template <int K, int KSH, int VAR>
static int maf_launch_bwd(const MafPlan& mp, int nw, const MafBwdArgs& a, hipStream_t st) {
  const int lds_bytes = 4 * (mp.n.lds_w_floats + nw * mp.sc_total);
  auto kern = maf_bwd_kernel<K, KSH, VAR>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return (int)e;
  const int64_t grid = (a.n + 16 * nw - 1) / (16 * nw);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * nw), (size_t)lds_bytes, st, mp, a);
  return (int)hipGetLastError();
}

// mode 0: log_prob (+ stash), 1: inverse, 2: backward of one transform
template <int K, int KSH, int VAR>
static int maf_dispatch_kv(const MafPlan& mp, int nw, int mode, const float* packed, const float* zstats,
                           const float* in, const float* x, int64_t n, int64_t x_rows, float* out_main, float* out_aux,
                           float* z_stash, const MafBwdArgs* bwd, hipStream_t st) {
  if (mode == 0)
    return maf_launch_flow<K, KSH, false, VAR>(mp, nw, packed, zstats, in, x, n, x_rows, out_main, out_aux, z_stash, st);
  if (mode == 1)
    return maf_launch_flow<K, KSH, true, VAR>(mp, nw, packed, zstats, in, x, n, x_rows, out_main, out_aux, z_stash, st);
  return maf_launch_bwd<K, KSH, VAR>(mp, nw, *bwd, st);
}

template <int K>
int maf_dispatch_k(const MafPlan& mp, int nw, int mode, const float* packed, const float* zstats, const float* in,
                   const float* x, int64_t n, int64_t x_rows, float* out_main, float* out_aux, float* z_stash,
                   const MafBwdArgs* bwd, hipStream_t st) {
#define MAF_KV(KS, V) maf_dispatch_kv<K, KS, V>(mp, nw, mode, packed, zstats, in, x, n, x_rows, out_main, out_aux, z_stash, bwd, st)
  if (mp.n.KSH == 13) return mp.variant ? MAF_KV(13, 1) : MAF_KV(13, 0);
  return mp.variant ? MAF_KV(16, 1) : MAF_KV(16, 0);
#undef MAF_KV
}
