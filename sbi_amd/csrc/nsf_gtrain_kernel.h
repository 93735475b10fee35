#pragma once
// nsf_gtrain_kernel.h -- GENERIC backward pass of the NSF training step for the shapes the wave-specialised
// kernel of nsf_train_kernel.h refuses (its weight image + exchange tiles must fit 160 KiB of LDS and its
// weight-gradient accumulators the register file: theta-dim <= 15, <= 2 residual blocks, conditioner input <= 32):
// every configuration the FORWARD kernel runs (hidden <= 64, <= 4 blocks, any theta-dim / x-dim whose weight image
// fits LDS) trains through this path.  Same decomposition as the maf kernels (maf_kernel.h):
//   nsf_gbwd_kernel<K,KSH>  per transform, row-parallel (one wave = 16 rows, no inter-wave traffic): LULinear backward,
//                           final layer from the stashed h_last, spline forward + reverse mode per two-dim chunk,
//                           Wf^T g_p on the fly, back through the residual blocks with the transposed weight image
//                           (activations from the forward's stash); leaves the per-row layer gradients (m-tile-major
//                           planes) and layer inputs (row-major) in HBM;
//   maf_dw_kernel           (shared) every linear's dW = G^T A as split-K MFMA GEMMs, LULinear's dense dU / dL included;
//   maf_reduce_kernel       fixed-order sum of the per-chunk slabs;
//   nsf_gfinish_kernel      slab -> flat gradient (LULinear: triangular entries, softplus chain rule of the diagonal).
// About 2x the time of the fast path at the shapes both run; deterministic; no atomics.
#include "maf_kernel.h"

#define GT_MAX_LIN 28      // linears (incl. column pieces of wide inputs) per transform handed to maf_dw_kernel

struct GTrainPlan {
  int sc_zs, sc_gy, sc_gz, sc_us, sc_cin, sc_pst, sc_total;   // per-wave scratch (floats)
  int PTW;                 // 16 * PT
  int gp_planes;           // d_tr_max * PTW / 16
  int g_planes;            // (1 + 3 NB) * 4
  int lu_planes;           // planes of g_u (+ the logabsdet column) and of g_z
  int act_w;               // (1 + 2 NB) * 64: h_last | per block relu(t1), relu(h_b)
  int cin_w;               // conditioner-input row [z_id ; context] padded to a multiple of 16
  int lua_w;               // 2 * round_up(D, 16): y | u
  int slab;                // floats per (chunk, transform) partial slab: layer parameters + dense LU tails
  int o_dU, o_dUb, o_dL, o_dLb;   // slab offsets of the LU pieces
};

struct GBwdArgs {
  const float* packed;
  const float* zstats;
  const float* z_in;       // (n, D) input state of this transform
  const float* x;
  const float* gz_up;      // (n, D) gradient wrt this transform's output; the LAST transform receives z_T
  const float* row_w;
  float uni_w;
  long long n, x_rows;
  float* gz_dn;
  float* grad_theta;
  const float* astash;     // the forward's activation stash
  float *GP, *G, *LUG;     // gradient planes (m-tile major)
  float *ACT, *CIN, *LUA;  // layer inputs (row major)
  long long npad;
  int t, is_last, par;
};

// D fragments -> 64 floats of a row-major row segment (activation side of the weight-gradient GEMMs)
__device__ __forceinline__ void gt_store_rows(float* __restrict__ dst, int ld, long long row, bool valid,
                                              const LaneId& id, const f4 (&v)[NSF_HT], bool relu) {
  if (!valid) return;
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt) {     // fragment order, one 16-byte store per m-tile (see store_frag_rows)
    float4 o = {v[mt][0], v[mt][1], v[mt][2], v[mt][3]};
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *reinterpret_cast<float4*>(dst + row * ld + 16 * mt + 4 * id.g) = o;
  }
}

template <int K, int KSH>
__global__ void __launch_bounds__(512)
nsf_gbwd_kernel(const NsfPlan pl, const GTrainPlan gp, const GBwdArgs a) {
  constexpr int PT = (3 * K - 1 + 15) / 16;
  const ShapeDesc& S = pl.shape[a.par];
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int wave = tid >> 6, nw = nthreads >> 6;
  const LaneId id = make_lane();
  float* sc = lds + pl.lds_w_floats + wave * gp.sc_total;
  float* zs = sc + gp.sc_zs;     // z; the transformed dims become the spline output y in place
  float* gys = sc + gp.sc_gy;    // gradient wrt the coupling output -> wrt this transform's input, in place
  float* gzs = sc + gp.sc_gz;    // upstream gradient wrt the LULinear output
  float* us = sc + gp.sc_us;     // g_u, later u = U y
  float* cin = sc + gp.sc_cin;
  float* pst = sc + gp.sc_pst;
  const int D = pl.D, C = pl.C, NB = pl.NB, par = a.par;
  const int LUS = D <= 16 ? 16 : D;
  const long long n = a.n;
  const long long row = (long long)blockIdx.x * (16 * nw) + 16 * wave + id.j;
  const bool valid = row < n;
  const long long rs = valid ? row : 0;
  const float* x_mean = a.zstats + 2 * D;
  const float* x_std = x_mean + C;
  const float wn = valid ? (a.row_w ? a.row_w[rs] : a.uni_w) : 0.f;
  const float gld = -wn;
  stage_layer(lds, a.packed + (long long)a.t * pl.img_floats, pl.img_floats, tid, nthreads);
  for (int i = id.lane; i < gp.sc_total; i += 64) sc[i] = 0.f;
  __syncthreads();
  // ---- inputs: state, upstream gradient, conditioner input row [z_id ; standardized context]
  {
    const long long xr = (a.x_rows == n) ? rs : (a.x_rows == 1 ? 0 : rs % a.x_rows);
    for (int d = id.g; d < D; d += 4) {
      const float z = valid ? a.z_in[rs * D + d] : 0.f;
      const float g = valid ? a.gz_up[rs * D + d] : 0.f;
      zs[id.j * pl.ZW + d] = z;
      gzs[id.j * pl.ZW + d] = a.is_last ? wn * g : g;
    }
    wave_lds_fence();
    for (int k = id.g; k < S.d_id; k += 4) cin[id.j * pl.CINW + k] = zs[id.j * pl.ZW + 2 * k + (1 - par)];
    for (int c = id.g; c < C; c += 4)
      cin[id.j * pl.CINW + S.d_id + c] = ((valid ? a.x[xr * C + c] : 0.f) - x_mean[c]) / x_std[c];
    wave_lds_fence();
    if (valid)
      for (int k = id.g; k < gp.cin_w; k += 4) a.CIN[row * gp.cin_w + k] = k < S.in0 ? cin[id.j * pl.CINW + k] : 0.f;
  }
  // ---- LULinear backward wrt its input: g_u = L^T g_z, g_y = U^T g_u (no forward values needed)
  for (int i = id.g; i < D; i += 4) {
    float acc = 0.f;
    for (int k = i; k < D; ++k) acc += lds[S.l_L + k * LUS + i] * gzs[id.j * pl.ZW + k];
    us[id.j * pl.ZW + i] = acc;
  }
  wave_lds_fence();
  for (int i = id.g; i < D; i += 4) {
    float acc = 0.f;
    for (int k = 0; k <= i; ++k) acc += lds[S.l_U + k * LUS + i] * us[id.j * pl.ZW + k];
    gys[id.j * pl.ZW + i] = acc;
  }
  // g_u and g_z rows -> LU gradient planes; column D of the g_u operand carries d/d(logabsdet) so that the bias MFMA of
  // the dU GEMM delivers sum_n d loss / d logabsdet_n
  if (valid) {
    const int du_cols = 16 * ((D + 1 + 15) / 16), dz_cols = 16 * ((D + 15) / 16);
    for (int k = id.g; k < du_cols; k += 4)
      a.LUG[((k >> 4) * a.npad + row) * 16 + (k & 15)] = k < D ? us[id.j * pl.ZW + k] : (k == D ? gld : 0.f);
    float* gzp = a.LUG + (long long)(du_cols >> 4) * a.npad * 16;
    for (int k = id.g; k < dz_cols; k += 4)
      gzp[((k >> 4) * a.npad + row) * 16 + (k & 15)] = k < D ? gzs[id.j * pl.ZW + k] : 0.f;
  }
  wave_lds_fence();
  // ---- final layer from the stashed h_last, spline forward + reverse mode, g_h = Wf^T g_p
  const long long nt16 = (n + 15) / 16;
  const long long t16 = (long long)blockIdx.x * nw + wave;
  const long long wt16 = t16 < nt16 ? t16 : nt16 - 1;   // wave-tiles past the last row were never stashed
  const float* ast = a.astash + (((long long)a.t * nt16 + wt16) * NSF_AST_SLOTS(NB)) * 1024 + 4 * id.lane;
  f4 gh[NSF_HT];
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt) gh[mt] = {0.f, 0.f, 0.f, 0.f};
  {
    f4 hl[NSF_HT];
    ast_load<KSH>(ast, 4 * NB, hl);
    gt_store_rows(a.ACT, gp.act_w, row, valid, id, hl, false);
    const LinDesc& LF = S.lin[S.fin];
    const int nchunks = (S.d_tr + 1) / 2;
    for (int c = 0; c < nchunks; ++c) {
      const int nact = S.d_tr - 2 * c >= 2 ? 2 : 1;
      if (nact == 2) final_layer_chunk_n<PT, KSH, 2>(lds, pst, pl, S, id, hl, 2 * c);
      else final_layer_chunk_n<PT, KSH, 1>(lds, pst, pl, S, id, hl, 2 * c);
      wave_lds_fence();
      const int slot = id.g & 1, part = id.g >> 1;
      const int dd = 2 * c + slot;
      if (dd < S.d_tr) {
        float* pp = pst + slot * pl.DS + id.j * pl.PSW;
        const int zi = id.j * pl.ZW + 2 * dd + par;
        float yv, gxv;
        rq_spline_pair_bwd<K>(pp, gp.PTW, zs[zi], gys[zi], gld, pl, part, yv, gxv);
        if (part == 0) {
          zs[zi] = yv;
          gys[zi] = gxv;
        }
      }
      wave_lds_fence();
      store_param_planes<PT>(a.GP, a.npad, row, valid, id, pst, pl.DS, pl.PSW, 2 * c, nact);
      maf_wft_chunk<PT>(lds, LF, pl, id, pst, 2 * c, nact, gh);
      wave_lds_fence();
    }
  }
  // ---- LULinear forward piece its parameter gradients need: u = U y  (y = zs after the spline); rows -> HBM
  for (int i = id.g; i < D; i += 4) {
    float acc = 0.f;
    for (int k = i; k < D; ++k) acc += lds[S.l_U + i * LUS + k] * zs[id.j * pl.ZW + k];
    us[id.j * pl.ZW + i] = acc;
  }
  wave_lds_fence();
  if (valid) {
    const int dwp = gp.lua_w >> 1;
    for (int k = id.g; k < dwp; k += 4) {
      a.LUA[row * gp.lua_w + k] = k < D ? zs[id.j * pl.ZW + k] : 0.f;
      a.LUA[row * gp.lua_w + dwp + k] = k < D ? us[id.j * pl.ZW + k] : 0.f;
    }
  }
  // ---- residual blocks, last -> first (activations from the forward's stash, one block in registers at a time)
  for (int b = NB - 1; b >= 0; --b) {
    f4 bt1[NSF_HT], bt2[NSF_HT], bsg[NSF_HT], hb[NSF_HT], ga[NSF_HT], gb[NSF_HT];
    ast_load<KSH>(ast, 2 + 4 * b, bt2);
    ast_load<KSH>(ast, 3 + 4 * b, bsg);
    ast_load<KSH>(ast, 1 + 4 * b, bt1);
    ast_load<KSH>(ast, 4 * b, hb);
    {
      f4 gc[NSF_HT];
#pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sgm = bsg[mt][r];
          ga[mt][r] = gh[mt][r] * sgm;                                 // d t2
          gc[mt][r] = gh[mt][r] * bt2[mt][r] * sgm * (1.f - sgm);      // d (Wc c + bc)
        }
      store_frag_planes(a.G + (long long)(4 * (1 + 3 * b)) * a.npad * 16, a.npad, row, valid, id, gc);
      store_frag_planes(a.G + (long long)(4 * (3 + 3 * b)) * a.npad * 16, a.npad, row, valid, id, ga);
    }
    gt_store_rows(a.ACT + 64 * (1 + 2 * b), gp.act_w, row, valid, id, bt1, true);   // relu(t1): input of W2
    gt_store_rows(a.ACT + 64 * (2 + 2 * b), gp.act_w, row, valid, id, hb, true);    // relu(h_b): input of W1
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt) gb[mt] = {0.f, 0.f, 0.f, 0.f};
    gemm_T_breg<KSH, NSF_HT>(lds, S.lin[3 + 3 * b], id, ga, gb);
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ga[mt][r] = bt1[mt][r] > 0.f ? gb[mt][r] : 0.f;                // d t1
        gb[mt][r] = 0.f;
      }
    store_frag_planes(a.G + (long long)(4 * (2 + 3 * b)) * a.npad * 16, a.npad, row, valid, id, ga);
    gemm_T_breg<KSH, NSF_HT>(lds, S.lin[2 + 3 * b], id, ga, gb);
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) gh[mt][r] += hb[mt][r] > 0.f ? gb[mt][r] : 0.f;
  }
  // ---- initial layer: G0 = g_h0; identity features receive W0[:, :d_id]^T g_h0
  store_frag_planes(a.G, a.npad, row, valid, id, gh);
  {
    f4 gin[2];
    gin[0] = {0.f, 0.f, 0.f, 0.f};
    gin[1] = {0.f, 0.f, 0.f, 0.f};
    gemm_T_breg<KSH, 2>(lds, S.lin[0], id, gh, gin);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * mt + 4 * r + id.g;
        if (k < S.d_id) gys[id.j * pl.ZW + 2 * k + (1 - par)] += gin[mt][r];
      }
  }
  wave_lds_fence();
  for (int d = id.g; d < D; d += 4) {
    if (valid) {
      const float g = gys[id.j * pl.ZW + d];
      if (a.t > 0) a.gz_dn[row * D + d] = g;
      else if (a.grad_theta) a.grad_theta[row * D + d] = g * a.zstats[D + d];
    }
  }
}

#ifdef NSF_GTRAIN_MAIN_TU
// slab sums (T, slab) -> flat gradient: conditioner parameters copied, LULinear from its dense pieces:
//   lower_entries[i(i-1)/2 + k] = dL[i][k] (k < i); upper_entries = dU[i][k] (k > i); bias = sum_n g_z;
//   d/d(unconstrained_upper_diag_i) = (dU[i][i] + (sum_n dloss/dlogabsdet_n) / U_ii) * sigmoid(unconstrained_i)
__global__ void __launch_bounds__(256)
nsf_gfinish_kernel(const NsfPlan pl, const GTrainPlan gp, const float* __restrict__ params,
                   const float* __restrict__ sums, float* __restrict__ grad) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pl.n_params) return;
  int t = 0;
  while (t + 1 < pl.T && idx >= pl.g_layer[t + 1]) ++t;
  const ShapeDesc& S = pl.shape[t & 1];
  const int li = idx - pl.g_layer[t];
  const float* s = sums + (long long)t * gp.slab;
  const int D = pl.D, ntri = D * (D - 1) / 2;
  float v;
  if (li < S.g_lu) {
    v = s[li];
  } else {
    const int q = li - S.g_lu;
    if (q < ntri) {                       // lower_entries, np.tril_indices(D, -1) order
      int i = 1;
      while ((i + 1) * i / 2 <= q) ++i;
      const int k = q - i * (i - 1) / 2;
      v = s[gp.o_dL + i * D + k];
    } else if (q < 2 * ntri) {            // upper_entries, np.triu_indices(D, 1) order
      const int qq = q - ntri;
      int i = 0, base = 0;
      while (base + (D - 1 - i) <= qq) { base += D - 1 - i; ++i; }
      const int k = i + 1 + (qq - base);
      v = s[gp.o_dU + i * D + k];
    } else if (q < 2 * ntri + D) {        // unconstrained_upper_diag
      const int i = q - 2 * ntri;
      const float ud = params[idx];
      const float uii = softplus_f(ud) + pl.lu_eps;
      v = (s[gp.o_dU + i * D + i] + s[gp.o_dUb + D] / uii) * (1.f / (1.f + expf(-ud)));
    } else {                              // bias
      v = s[gp.o_dLb + (q - 2 * ntri - D)];
    }
  }
  grad[idx] = v;
}
#endif

template <int K>
int nsf_gbwd_launch_k(const NsfPlan& pl, const GTrainPlan& gp, int nw, const GBwdArgs& a, hipStream_t st) {
  const int lds_bytes = 4 * (pl.lds_w_floats + nw * gp.sc_total);
  const int64_t grid = (a.n + 16 * nw - 1) / (16 * nw);
  if (pl.KSH == 13) {
    auto kern = nsf_gbwd_kernel<K, 13>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * nw), (size_t)lds_bytes, st, pl, gp, a);
  } else {
    auto kern = nsf_gbwd_kernel<K, 16>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * nw), (size_t)lds_bytes, st, pl, gp, a);
  }
  return (int)hipGetLastError();
}
