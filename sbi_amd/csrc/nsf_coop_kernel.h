#pragma once
// nsf_coop_kernel.h -- the cooperative (latency-oriented) NSF kernels for small batches on gfx950; design and
// reference path in nsf_coop.h.  Three kernels:
//   nsf_coop_pack_kernel      flat parameters -> fragment-ordered image (forward AND transposed matrices)
//   nsf_coop_fwd_kernel       log p (+ noise, + the stash the backward pass needs) for all T transforms
//   nsf_coop_bwd_kernel       all T transforms backward in one launch: d loss / d theta, per-workgroup partial
//                             weight-gradient slabs in the throughput path's layout (nsf_grad_reduce_kernel sums them)
#include <hip/hip_runtime.h>
#include "nsf_coop.h"
#include "nsf_device.h"

// ------------------------------------------------------------------------------------------------ pack
__device__ __forceinline__ float co_lu_entry(const float* __restrict__ lu, int D, float eps, bool upper, int i, int k) {
  // LULinear._create_lower_upper (nflows transforms/lu.py): np.tril_indices(D,-1) / np.triu_indices(D,1) order
  const int ntri = D * (D - 1) / 2;
  if (i >= D || k >= D) return 0.f;
  if (upper) {
    if (k > i) return lu[ntri + i * D - i * (i + 1) / 2 + (k - i - 1)];
    if (k == i) return softplus_f(lu[2 * ntri + i]) + eps;
    return 0.f;
  }
  if (k < i) return lu[i * (i - 1) / 2 + k];
  return k == i ? 1.f : 0.f;
}

__device__ __forceinline__ float co_mat_value(const CoMat& m, const NsfPlan& pl, const ShapeDesc& S, const CoShape& c,
                                              const float* __restrict__ gl, int rel) {
  const int blk = rel >> 8, lane = (rel >> 2) & 63, r = rel & 3;
  const int mt = blk / m.quads, q = blk - mt * m.quads;
  const int i = lane & 15, g = lane >> 4;
  const int mi = 4 * (i & 3) + (i >> 2);          // iperm
  const int k = 4 * (4 * q + r) + g;
  const int P = pl.P;
  switch (m.kind) {
    case CO_K_W0: {
      const LinDesc& L = S.lin[0];
      const int f = 16 * mt + mi;
      if (f >= L.out) return 0.f;
      if (k < 4 * c.KC) return k < pl.C ? gl[L.g_w + f * L.in + S.d_id + k] : 0.f;
      const int kz = k - 4 * c.KC;
      return kz < S.d_id ? gl[L.g_w + f * L.in + kz] : 0.f;
    }
    case CO_K_PLAIN: {
      const LinDesc& L = S.lin[m.lin];
      const int f = 16 * mt + mi;
      return (f < L.out && k < L.in) ? gl[L.g_w + f * L.in + k] : 0.f;
    }
    case CO_K_WF: {
      const LinDesc& L = S.lin[m.lin];
      const int dd = mt / pl.PT, p = 16 * (mt % pl.PT) + mi;
      return (p < P && k < L.in) ? gl[L.g_w + (dd * P + p) * L.in + k] : 0.f;
    }
    case CO_K_WFT: {
      const LinDesc& L = S.lin[m.lin];
      const int f = 16 * mt + mi;
      const int dd = k / (16 * pl.PT), p = k - dd * 16 * pl.PT;
      return (f < L.in && p < P && dd < S.d_tr) ? gl[L.g_w + (dd * P + p) * L.in + f] : 0.f;
    }
    case CO_K_PLAIN_T: {
      const LinDesc& L = S.lin[m.lin];
      const int f = 16 * mt + mi;
      return (f < L.in && k < L.out) ? gl[L.g_w + k * L.in + f] : 0.f;
    }
    case CO_K_W0T: {
      const LinDesc& L = S.lin[0];
      const int f = 16 * mt + mi;     // identity slot
      return (f < S.d_id && k < L.out) ? gl[L.g_w + k * L.in + f] : 0.f;
    }
    case CO_K_CTX_T: {
      const LinDesc& L = S.lin[m.lin];
      const int cc = 16 * mt + mi;    // context feature; the initial layer keeps its context columns behind z_id
      const int col0 = m.lin == 0 ? S.d_id : 0;
      return (cc < pl.C && k < L.out) ? gl[L.g_w + k * L.in + col0 + cc] : 0.f;
    }
    case CO_K_U: return co_lu_entry(gl + S.g_lu, pl.D, pl.lu_eps, true, mi, k);
    case CO_K_L: return co_lu_entry(gl + S.g_lu, pl.D, pl.lu_eps, false, mi, k);
    case CO_K_UT: return co_lu_entry(gl + S.g_lu, pl.D, pl.lu_eps, true, k, mi);
    case CO_K_LT: return co_lu_entry(gl + S.g_lu, pl.D, pl.lu_eps, false, k, mi);
  }
  return 0.f;
}

__device__ __forceinline__ float co_bias_value(const CoBias& b, const NsfPlan& pl, const ShapeDesc& S,
                                               const float* __restrict__ gl, int rel) {
  const int mt = rel >> 4, e = rel & 15;
  const int g = e >> 2, r = e & 3;
  const int row = 4 * r + g;
  if (b.kind == 0) {
    const LinDesc& L = S.lin[b.lin];
    const int f = 16 * mt + row;
    return f < L.out ? gl[L.g_b + f] : 0.f;
  }
  if (b.kind == 1) {
    const LinDesc& L = S.lin[b.lin];
    const int dd = mt / pl.PT, p = 16 * (mt % pl.PT) + row;
    return p < pl.P ? gl[L.g_b + dd * pl.P + p] : 0.f;
  }
  const int ntri = pl.D * (pl.D - 1) / 2;
  return row < pl.D ? gl[S.g_lu + 2 * ntri + pl.D + row] : 0.f;
}

#ifdef NSF_COOP_MAIN_TU   // (translation-unit guard: the non-template kernel is defined once, in nsf_coop.hip)
// grid (T, slices): every destination float of the transform's image is computed from its index
__global__ void __launch_bounds__(256)
nsf_coop_pack_kernel(const NsfPlan pl, const CoopPlan cp, const float* __restrict__ params,
                     float* __restrict__ cimg) {
  const int t = blockIdx.x;
  const int par = t & 1;
  const ShapeDesc& S = pl.shape[par];
  const CoShape& c = cp.sh[par];
  const float* gl = params + pl.g_layer[t];
  float* img = cimg + (long long)t * cp.img_floats;
  const CoMat* mats = &c.W0;                      // the CoMat members are laid out contiguously ...
  constexpr int NMAT = 9 + 6 * NSF_MAX_NB;      // W0 WC[] W1[] W2[] WF U L | WFT W1T[] W2T[] W0T UT LT | WCT[] W0CT
  const CoBias* bias = &c.b0;                     // ... and so are the CoBias members
  constexpr int NBIAS = 3 + 3 * NSF_MAX_NB;
  for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < cp.img_floats; idx += gridDim.y * blockDim.x) {
    float v = 0.f;
    bool found = false;
    for (int m = 0; m < NMAT && !found; ++m) {
      const CoMat& M = mats[m];
      const int sz = M.mtiles * M.quads * 256;
      if (sz > 0 && idx >= M.off && idx < M.off + sz) {
        v = co_mat_value(M, pl, S, c, gl, idx - M.off);
        found = true;
      }
    }
    for (int b = 0; b < NBIAS && !found; ++b) {
      const CoBias& B = bias[b];
      const int sz = 16 * B.mtiles;
      if (sz > 0 && idx >= B.off && idx < B.off + sz) {
        v = co_bias_value(B, pl, S, gl, idx - B.off);
        found = true;
      }
    }
    if (!found && idx == c.o_ld) {                // logabsdet of the LULinear = sum_i log(softplus(u_i) + eps)
      const int ntri = pl.D * (pl.D - 1) / 2;
      float a = 0.f;
      for (int i = 0; i < pl.D; ++i) a += logf(softplus_f(gl[S.g_lu + 2 * ntri + i]) + pl.lu_eps);
      v = a;
    }
    img[idx] = v;
  }
}

#endif

// ------------------------------------------------------------------------------------------------ device helpers
// A fragments of one m-tile of a matrix: NQ 16-byte words per lane (block stride 256 floats)
template <int NQ>
__device__ __forceinline__ void co_load_a(const float* __restrict__ img, const CoMat& m, int mt, int lane, f4 (&a)[NQ]) {
  const f4* p = reinterpret_cast<const f4*>(img + m.off + mt * m.quads * 256) + lane;
#pragma unroll
  for (int q = 0; q < NQ; ++q) a[q] = q < m.quads ? p[q * 64] : f4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ f4 co_load_bias(const float* __restrict__ img, const CoBias& b, int mt, int g) {
  return *reinterpret_cast<const f4*>(img + b.off + 16 * mt + 4 * g);
}

// all-gather of the waves' D fragments: out[u][mt] = fragment of wave mt (lane for lane); one barrier
template <int NT>
__device__ __forceinline__ void co_gather(float* __restrict__ ex, int& buf, int wave, int lane, const f4 (&mine)[NT],
                                          f4 (&out)[NT][CO_WAVES]) {
  f4* e = reinterpret_cast<f4*>(ex) + buf * (CO_WAVES * NT * 64);
#pragma unroll
  for (int u = 0; u < NT; ++u) e[(wave * NT + u) * 64 + lane] = mine[u];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int mt = 0; mt < CO_WAVES; ++mt) out[u][mt] = e[(mt * NT + u) * 64 + lane];
  buf ^= 1;
}

// acc[u] += A(m-tile) * B, B = gathered fragments (K-step s <-> b[u][s >> 2][s & 3]); KS K-steps
template <int NT, int KS>
__device__ __forceinline__ void co_gemm_h(const f4 (&a)[4], const f4 (&b)[NT][CO_WAVES], f4 (&acc)[NT]) {
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int u = 0; u < NT; ++u) acc[u] = MFMA16(a[s >> 2][s & 3], b[u][s >> 2][s & 3], acc[u]);
}

// stash slot address of (transform t, 16-row tile, slot): 256 floats, lane-major 16-byte words
__device__ __forceinline__ f4* co_slot(float* __restrict__ ast, const CoopPlan& cp, long long nt16, int t,
                                       long long tile16, int slot, int lane) {
  return reinterpret_cast<f4*>(ast + (((long long)t * nt16 + tile16) * cp.slots + slot) * 256) + lane;
}

// ------------------------------------------------------------------------------------------------ forward
template <int K, int KSH, int NT>
__global__ void __launch_bounds__(64 * CO_WAVES)
nsf_coop_fwd_kernel(const NsfPlan pl, const CoopPlan cp, const float* __restrict__ cimg,
                    const float* __restrict__ zstats, const float* __restrict__ theta, const float* __restrict__ x,
                    long long n, long long x_rows, float* __restrict__ logp, float* __restrict__ noise_out,
                    float* __restrict__ zst, float* __restrict__ ast) {
  constexpr int PT = (3 * K - 1 + 15) / 16;
  constexpr int R = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const LaneId id = make_lane();
  const int D = pl.D, C = pl.C, ZS = cp.ZS;
  float* zs = lds + cp.o_zs;
  float* pst = lds + cp.o_pst;
  float* ex = lds + cp.o_ex;
  float* ldp = lds + cp.o_ldp;
  const long long row0 = (long long)blockIdx.x * R;
  const long long nt16 = (n + 15) / 16;
  const float* th_shift = zstats;
  const float* th_scale = zstats + D;
  const float* x_mean = zstats + 2 * D;
  const float* x_std = x_mean + C;

  // ---- prologue: z-scored theta rows -> LDS; standardized context as B fragments (K-step s <-> c = 4 s + g)
  for (int i = tid; i < R * ZS + 16; i += 64 * CO_WAVES) zs[i] = 0.f;
  __syncthreads();
  for (int i = tid; i < R * D; i += 64 * CO_WAVES) {
    const int r = i / D, d = i - r * D;
    const long long row = row0 + r;
    zs[r * ZS + d] = row < n ? theta[row * D + d] * th_scale[d] + th_shift[d] : 0.f;
  }
  float cb[NT][8];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const long long row = row0 + 16 * u + id.j;
    const long long rs = row < n ? row : 0;
    const long long xr = (x_rows == n) ? rs : (x_rows == 1 ? 0 : rs % x_rows);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int c = 4 * s + id.g;
      cb[u][s] = (c < C && row < n) ? (x[xr * C + c] - x_mean[c]) / x_std[c] : 0.f;
    }
  }
  float ld_acc[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) ld_acc[u] = 0.f;
  float ld_const = 0.f;
  for (int d = 0; d < D; ++d) ld_const += logf(fabsf(th_scale[d]));
  int buf = 0;
  __syncthreads();

  for (int t = 0; t < pl.T; ++t) {
    const int par = t & 1;
    const ShapeDesc& S = pl.shape[par];
    const CoShape& c = cp.sh[par];
    const float* img = cimg + (long long)t * cp.img_floats;
    if (zst)
      for (int i = tid; i < R * D; i += 64 * CO_WAVES) {
        const int r = i / D, d = i - r * D;
        if (row0 + r < n) zst[((long long)t * n + row0 + r) * D + d] = zs[r * ZS + d];
      }
    // ---- initial layer: h = W0 [context ; z_id] + b0   (m-tile = wave)
    f4 h[NT];
    {
      f4 a0[4];
      co_load_a<4>(img, c.W0, wave, id.lane, a0);
      const f4 bias = co_load_bias(img, c.b0, wave, id.g);
#pragma unroll
      for (int u = 0; u < NT; ++u) h[u] = bias;
#pragma unroll
      for (int s = 0; s < 8; ++s)
        if (s < c.KC) {
#pragma unroll
          for (int u = 0; u < NT; ++u) h[u] = MFMA16(a0[s >> 2][s & 3], cb[u][s], h[u]);
        }
#pragma unroll
      for (int sz = 0; sz < 4; ++sz)
        if (sz < c.KZ) {
          const int s = c.KC + sz;
          const int kz = 4 * sz + id.g;
          // a0 is indexed with a run-time K-step: select among the (<= 16) candidates without scratch
          float av = 0.f;
#pragma unroll
          for (int ss = 0; ss < 12; ++ss) av = (ss == s) ? a0[ss >> 2][ss & 3] : av;
#pragma unroll
          for (int u = 0; u < NT; ++u) {
            const float bv = kz < S.d_id ? zs[(16 * u + id.j) * ZS + 2 * kz + (1 - par)] : 0.f;
            h[u] = MFMA16(av, bv, h[u]);
          }
        }
    }
    if (ast) {
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const long long t16 = (row0 >> 4) + u;
        if (t16 < nt16) __builtin_nontemporal_store(h[u], co_slot(ast, cp, nt16, t, t16, wave, id.lane));
      }
    }
    // ---- residual blocks
    for (int b = 0; b < pl.NB; ++b) {
      f4 a1[4], a2[4], ac[2];
      co_load_a<4>(img, c.W1[b], wave, id.lane, a1);
      co_load_a<2>(img, c.WC[b], wave, id.lane, ac);
      f4 gate[NT], tt[NT], u1[NT], bg[NT][CO_WAVES];
      {
        const f4 bias = co_load_bias(img, c.bc[b], wave, id.g);
#pragma unroll
        for (int u = 0; u < NT; ++u) gate[u] = bias;
#pragma unroll
        for (int s = 0; s < 8; ++s)
          if (s < c.KC) {
#pragma unroll
            for (int u = 0; u < NT; ++u) gate[u] = MFMA16(ac[s >> 2][s & 3], cb[u][s], gate[u]);
          }
      }
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) tt[u][r] = fmaxf(h[u][r], 0.f);
      co_gather<NT>(ex, buf, wave, id.lane, tt, bg);
      co_load_a<4>(img, c.W2[b], wave, id.lane, a2);
      {
        const f4 bias = co_load_bias(img, c.b1[b], wave, id.g);
#pragma unroll
        for (int u = 0; u < NT; ++u) u1[u] = bias;
      }
      co_gemm_h<NT, KSH>(a1, bg, u1);
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gate[u][r] = sigmoid_f(gate[u][r]);
          tt[u][r] = fmaxf(u1[u][r], 0.f);
        }
      if (ast) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const long long t16 = (row0 >> 4) + u;
          if (t16 < nt16) {
            __builtin_nontemporal_store(u1[u], co_slot(ast, cp, nt16, t, t16, cp.s_blk + 16 * b + wave, id.lane));
            __builtin_nontemporal_store(gate[u], co_slot(ast, cp, nt16, t, t16, cp.s_blk + 16 * b + 8 + wave, id.lane));
          }
        }
      }
      co_gather<NT>(ex, buf, wave, id.lane, tt, bg);
      {
        const f4 bias = co_load_bias(img, c.b2[b], wave, id.g);
#pragma unroll
        for (int u = 0; u < NT; ++u) u1[u] = bias;
      }
      co_gemm_h<NT, KSH>(a2, bg, u1);
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[u][r] += u1[u][r] * gate[u][r];
      if (ast) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const long long t16 = (row0 >> 4) + u;
          if (t16 < nt16) {
            __builtin_nontemporal_store(u1[u], co_slot(ast, cp, nt16, t, t16, cp.s_blk + 16 * b + 4 + wave, id.lane));
            __builtin_nontemporal_store(h[u], co_slot(ast, cp, nt16, t, t16, cp.s_blk + 16 * b + 12 + wave, id.lane));
          }
        }
      }
    }
    // ---- final layer: parameter tiles wave, wave + 4, ... -> staging rows pst[row][dim][3K-1 raw outputs]
    {
      f4 hb[NT][CO_WAVES];
      co_gather<NT>(ex, buf, wave, id.lane, h, hb);
      for (int mt = wave; mt < c.nft; mt += CO_WAVES) {
        f4 af[4];
        co_load_a<4>(img, c.WF, mt, id.lane, af);
        const f4 bias = co_load_bias(img, c.bf, mt, id.g);
        f4 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[u] = bias;
        co_gemm_h<NT, KSH>(af, hb, acc);
        const int dd = mt / PT, pt = mt - dd * PT;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pst[(16 * u + id.j) * cp.DSTR + dd * cp.PSW + 16 * pt + 4 * r + id.g] = acc[u][r];
          if (ast) {
            const long long t16 = (row0 >> 4) + u;
            if (t16 < nt16) __builtin_nontemporal_store(acc[u], co_slot(ast, cp, nt16, t, t16, cp.s_par + mt, id.lane));
          }
        }
      }
    }
    __syncthreads();
    // ---- spline: task (row 16 u + j, dim 2 wave + slot) on the lane pair (lane, lane ^ 32)
    {
      const int slot = id.g & 1, part = id.g >> 1;
      const int dd_raw = 2 * wave + slot;
      const bool live = dd_raw < S.d_tr;
      const int dd = live ? dd_raw : 0;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int r = 16 * u + id.j;
        const int zi = r * ZS + 2 * dd + par;
        float y, ld;
        rq_spline_pair<K, false>(pst + r * cp.DSTR + dd * cp.PSW, zs[zi], pl, part, y, ld);
        if (live && part == 0) zs[zi] = y;
        ld_acc[u] += (live && part == 0) ? ld : 0.f;
      }
    }
    __syncthreads();
    // ---- LULinear: y = L (U z) + b as two chained 16 x 16 MFMA mat-vecs; row tile u on wave u
    if (wave < NT) {
      const int u = wave;
      const f4 au = *(reinterpret_cast<const f4*>(img + c.U.off) + id.lane);
      const f4 al = *(reinterpret_cast<const f4*>(img + c.L.off) + id.lane);
      f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = MFMA16(au[s], zs[(16 * u + id.j) * ZS + 4 * s + id.g], acc);
      f4 yv = co_load_bias(img, c.blu, 0, id.g);
#pragma unroll
      for (int s = 0; s < 4; ++s) yv = MFMA16(al[s], acc[s], yv);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * r + id.g < D) zs[(16 * u + id.j) * ZS + 4 * r + id.g] = yv[r];
    }
    ld_const += img[c.o_ld];
    __syncthreads();
  }

  // ---- epilogue: per-row sums in a fixed order (deterministic)
  {
    const int slot = id.g & 1, part = id.g >> 1;
    if (part == 0) {
#pragma unroll
      for (int u = 0; u < NT; ++u) ldp[(2 * wave + slot) * R + 16 * u + id.j] = ld_acc[u];
    }
  }
  __syncthreads();
  if (tid < R) {
    const long long row = row0 + tid;
    if (row < n) {
      float ld = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) ld += ldp[q * R + tid];
      float ss = 0.f;
      for (int d = 0; d < D; ++d) {
        const float z = zs[tid * ZS + d];
        ss += z * z;
        if (noise_out) noise_out[row * D + d] = z;
      }
      logp[row] = -0.5f * ss + ld + ld_const - pl.log_z;
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// D fragment (lane (g, j), register r <-> feature f0 + 4 r + g, row 16 u + j) -> transposed tile T[feature][row]
template <int NT>
__device__ __forceinline__ void co_store_T(float* __restrict__ T, int RS, int f0, const LaneId& id, const f4 (&v)[NT],
                                           bool relu, int ones_row) {
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = f0 + 4 * r + id.g;
      float a = relu ? fmaxf(v[u][r], 0.f) : v[u][r];
      a = (f == ones_row) ? 1.f : a;
      T[f * RS + 16 * u + id.j] = a;
    }
}

// weight-gradient tile block of one m-tile: acc[nt] = sum over the workgroup's rows of A^T[out0 + i][row] B^T[in][row],
// both operands from transposed tiles (one ds_read_b128 per operand, row tile and K-step group)
template <int NT, int NNT>
__device__ __forceinline__ void co_dw(const float* __restrict__ At, const float* __restrict__ Bt, int RS, int out0,
                                      int in0, int nnt, const LaneId& id, f4 (&acc)[NNT], f4* accb) {
  f4 a[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) a[u] = *reinterpret_cast<const f4*>(At + (out0 + id.j) * RS + 16 * u + 4 * id.g);
#pragma unroll
  for (int nt = 0; nt < NNT; ++nt) {
    acc[nt] = f4{0.f, 0.f, 0.f, 0.f};
    if (nt < nnt) {
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const f4 b = *reinterpret_cast<const f4*>(Bt + (in0 + 16 * nt + id.j) * RS + 16 * u + 4 * id.g);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[nt] = MFMA16(a[u][s], b[s], acc[nt]);
      }
    }
  }
  if (accb) {   // bias gradients when the layer input has no spare column for the ones row (hidden_features = 64)
    *accb = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) *accb = MFMA16(a[u][s], 1.f, *accb);
  }
}

// partial-gradient write-out of one weight tile (lane (g, j), register r: out = out0 + 4 g + r, in = 16 nt + j);
// column `in == L.in` of the activation tile is the ones row => bias gradient
__device__ __forceinline__ void co_write_tile(float* __restrict__ part, const LinDesc& L, int out0, int nt,
                                              const LaneId& id, const f4& acc) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int out = out0 + 4 * id.g + r;
    const int in = 16 * nt + id.j;
    if (out < L.out) {
      if (in < L.in) part[L.g_w + out * L.in + in] = acc[r];
      else if (in == L.in) part[L.g_b + out] = acc[r];
    }
  }
}
__device__ __forceinline__ void co_write_bias(float* __restrict__ part, const LinDesc& L, int out0, const LaneId& id,
                                              const f4& accb) {
  if (id.j != 0) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int out = out0 + 4 * id.g + r;
    if (out < L.out) part[L.g_b + out] = accb[r];
  }
}

template <int K, int KSH, int NT>
__global__ void __launch_bounds__(64 * CO_WAVES)
nsf_coop_bwd_kernel(const NsfPlan pl, const CoopPlan cp, const float* __restrict__ cimg,
                    const float* __restrict__ zstats, const float* __restrict__ x, long long n, long long x_rows,
                    const float* __restrict__ row_w, const float uni_w, const float* __restrict__ z_last,
                    const float* __restrict__ zst, const float* __restrict__ ast, float* __restrict__ partial,
                    float* __restrict__ grad_theta, float* __restrict__ grad_x) {
  constexpr int PT = (3 * K - 1 + 15) / 16;
  constexpr int R = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const LaneId id = make_lane();
  const int D = pl.D, C = pl.C, ZS = cp.ZS, RS = cp.RS;
  float* zs = lds + cp.o_zs;
  float* gys = lds + cp.o_gys;
  float* gzs = lds + cp.o_gzs;
  float* wrow = lds + cp.o_w;
  float* pst = lds + cp.o_pst;
  float* ex = lds + cp.o_ex;
  float* GT0 = lds + cp.o_gt;
  float* GT1 = GT0 + 64 * RS;
  float* ATb[2] = {lds + cp.o_at, lds + cp.o_at + 65 * RS};
  float* CT = lds + cp.o_ct;
  float* GUT = lds + cp.o_lut;
  float* GZT = GUT + 17 * RS;
  float* YT = GZT + 17 * RS;
  float* UTt = YT + 17 * RS;
  float* ctx = lds + cp.o_ctx;
  const long long row0 = (long long)blockIdx.x * R;
  const long long nt16 = (n + 15) / 16;
  const float* x_mean = zstats + 2 * D;
  const float* x_std = x_mean + C;
  const bool hb64 = pl.H == 64;
  const int ones_h = hb64 ? -1 : pl.H;
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};

  for (int i = tid; i < cp.lds_floats; i += 64 * CO_WAVES) lds[i] = 0.f;
  __syncthreads();
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
  constexpr int atc = 0;
  // d loss / d context (a trainable embedding net in front of the flow): waves 0 and 1 own context features
  // [16 wave, 16 wave + 16) and accumulate W0[:, ctx]^T g_h0 + sum_b Wc_b^T g_c over all transforms
  const bool want_gx = grad_x != nullptr;
  f4 gxacc[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) gxacc[u] = zero4;
  // clamped stash tiles: wave-tiles past the last row were never written by the forward pass
  long long t16c[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) t16c[u] = (row0 >> 4) + u < nt16 ? (row0 >> 4) + u : nt16 - 1;
  __syncthreads();

  for (int t = pl.T - 1; t >= 0; --t) {
    const int par = t & 1;
    const ShapeDesc& S = pl.shape[par];
    const CoShape& c = cp.sh[par];
    const float* img = cimg + (long long)t * cp.img_floats;
    float* part = partial + ((long long)t * gridDim.x + blockIdx.x) * cp.PLP;
    const LinDesc& LF = S.lin[S.fin];
    auto slot = [&](int u, int sl) -> f4 {
      return __builtin_nontemporal_load(
          reinterpret_cast<const f4*>(ast + (((long long)t * nt16 + t16c[u]) * cp.slots + sl) * 256) + id.lane);
    };
    // ---- P0: state rows, conditioner-input tile, spline parameters (stash) -> LDS; LULinear backward
    for (int i = tid; i < R * D; i += 64 * CO_WAVES) {
      const int r = i / D, d = i - r * D;
      const long long row = row0 + r;
      zs[r * ZS + d] = row < n ? zst[((long long)t * n + row) * D + d] : 0.f;
    }
    for (int i = tid; i < cp.ct_rows * R; i += 64 * CO_WAVES) {
      const int k = i / R, r = i - k * R;
      const long long row = row0 + r;
      float v = (k == S.in0) ? 1.f : 0.f;
      if (k < S.d_id) v = row < n ? zst[((long long)t * n + row) * D + 2 * k + (1 - par)] : 0.f;
      else if (k < S.in0) v = ctx[(k - S.d_id) * R + r];
      CT[k * RS + r] = v;
    }
    for (int mt = wave; mt < c.nft; mt += CO_WAVES) {
      const int dd = mt / PT, pt = mt - dd * PT;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const f4 v = slot(u, cp.s_par + mt);
#pragma unroll
        for (int r = 0; r < 4; ++r) pst[(16 * u + id.j) * cp.DSTR + dd * cp.PSW + 16 * pt + 4 * r + id.g] = v[r];
      }
    }
    f4 hl[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) hl[u] = slot(u, cp.s_blk + 16 * (pl.NB - 1) + 12 + wave);
    if (wave < NT) {   // g_u = L^T g_z, g_y = U^T g_u for row tile `wave`
      const int u = wave;
      const f4 alt = *(reinterpret_cast<const f4*>(img + c.LT.off) + id.lane);
      const f4 aut = *(reinterpret_cast<const f4*>(img + c.UT.off) + id.lane);
      f4 gu = zero4, gy = zero4;
      float gz[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        gz[s] = gzs[(16 * u + id.j) * ZS + 4 * s + id.g];
        gu = MFMA16(alt[s], gz[s], gu);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) gy = MFMA16(aut[s], gu[s], gy);
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
      if (dd < S.d_tr) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const int r = 16 * u + id.j;
          const int zi = r * ZS + 2 * dd + par;
          float yv, gxv;
          rq_spline_pair_bwd<K>(pst + r * cp.DSTR + dd * cp.PSW, 16 * PT, zs[zi], gys[zi], -wrow[r], pl, sp, yv, gxv);
          if (sp == 0) {
            zs[zi] = yv;
            gys[zi] = gxv;
          }
        }
      }
    }
    __syncthreads();
    // ---- P2: u = U y (LU parameter gradients), h_last -> activation tile, g_h = Wf^T g_p (m-tile = wave)
    if (wave < NT) {
      const int u = wave;
      const f4 au = *(reinterpret_cast<const f4*>(img + c.U.off) + id.lane);
      f4 uv = zero4;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float yv = zs[(16 * u + id.j) * ZS + 4 * s + id.g];
        uv = MFMA16(au[s], yv, uv);
        YT[(4 * s + id.g) * RS + 16 * u + id.j] = yv;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) UTt[(4 * r + id.g) * RS + 16 * u + id.j] = uv[r];
    }
    float* AT = ATb[atc];
    co_store_T<NT>(AT, RS, 16 * wave, id, hl, false, ones_h);
    f4 gh[NT];
    {
      f4 acc0[NT], acc1[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) { acc0[u] = zero4; acc1[u] = zero4; }
      const f4* ap = reinterpret_cast<const f4*>(img + c.WFT.off + wave * c.WFT.quads * 256) + id.lane;
      const int nq = c.WFT.quads;
      for (int q = 0; q < nq; q += 2) {
        const f4 a0 = ap[q * 64];
        const f4 a1 = (q + 1 < nq) ? ap[(q + 1) * 64] : zero4;
        const int dd0 = q / PT, p0 = 16 * (q - dd0 * PT);
        const int dd1 = (q + 1) / PT, p1 = 16 * ((q + 1) - dd1 * PT);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int u = 0; u < NT; ++u) {
            const float b0 = pst[(16 * u + id.j) * cp.DSTR + dd0 * cp.PSW + p0 + 4 * r + id.g];
            acc0[u] = MFMA16(a0[r], b0, acc0[u]);
            if (q + 1 < nq) {
              const float b1 = pst[(16 * u + id.j) * cp.DSTR + dd1 * cp.PSW + p1 + 4 * r + id.g];
              acc1[u] = MFMA16(a1[r], b1, acc1[u]);
            }
          }
      }
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) gh[u][r] = acc0[u][r] + acc1[u][r];
    }
    __syncthreads();
    // ---- d Wf (parameter tiles wave, wave + 4, ...), LULinear parameter gradients
    for (int mt = wave; mt < c.nft; mt += CO_WAVES) {
      const int dd = mt / PT, pt = mt - dd * PT;
      f4 acc[4], accb = zero4;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = zero4;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        float a[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = pst[(16 * u + 4 * id.g + s) * cp.DSTR + dd * cp.PSW + 16 * pt + id.j];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const f4 b = *reinterpret_cast<const f4*>(AT + (16 * nt + id.j) * RS + 16 * u + 4 * id.g);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[nt] = MFMA16(a[s], b[s], acc[nt]);
        }
        if (hb64) {
#pragma unroll
          for (int s = 0; s < 4; ++s) accb = MFMA16(a[s], 1.f, accb);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = 16 * pt + 4 * id.g + r;
          const int in = 16 * nt + id.j;
          if (p < pl.P) {
            const int out = dd * pl.P + p;
            if (in < LF.in) part[LF.g_w + out * LF.in + in] = acc[nt][r];
            else if (in == LF.in) part[LF.g_b + out] = acc[nt][r];
            if (hb64 && nt == 0 && id.j == 0) part[LF.g_b + out] = accb[r];
          }
        }
    }
    {
      const int ntri = D * (D - 1) / 2;
      float* plow = part + S.g_lu;
      float* pup = plow + ntri;
      float* pdiag = pup + ntri;
      float* pbias = pdiag + D;
      if (wave == 0 || wave == 1) {   // d U = g_u (x) y (wave 0), d L = g_z (x) u (wave 1): one 16 x 16 tile each
        f4 acc[1];
        co_dw<NT, 1>(wave == 0 ? GUT : GZT, wave == 0 ? YT : UTt, RS, 0, 0, 1, id, acc, nullptr);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 4 * id.g + r, k = id.j;
          if (i < D && k < D) {
            if (wave == 0) {
              if (k > i) pup[i * D - i * (i + 1) / 2 + (k - i - 1)] = acc[0][r];
              else if (k == i) pdiag[i] = acc[0][r];     // dL/dU_ii; chain rule finished in the reduction
            } else if (k < i) {
              plow[i * (i - 1) / 2 + k] = acc[0][r];
            }
          }
        }
      } else if (wave == 2) {
        if (id.lane < D) {            // d bias = sum_n g_z
          float a = 0.f;
          for (int r = 0; r < R; ++r) a += gzs[r * ZS + id.lane];
          pbias[id.lane] = a;
        } else if (id.lane == 63) {   // sum_n d loss / d logabsdet_n = - sum_n w_n
          float a = 0.f;
          for (int r = 0; r < R; ++r) a -= wrow[r];
          part[S.n_params] = a;
        }
      }
    }
    // ---- residual blocks, last -> first
    for (int b = pl.NB - 1; b >= 0; --b) {
      f4 t1[NT], hbk[NT], ga[NT], gc[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const f4 t2 = slot(u, cp.s_blk + 16 * b + 4 + wave);
        const f4 sg = slot(u, cp.s_blk + 16 * b + 8 + wave);
        t1[u] = slot(u, cp.s_blk + 16 * b + wave);
        hbk[u] = b == 0 ? slot(u, wave) : slot(u, cp.s_blk + 16 * (b - 1) + 12 + wave);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ga[u][r] = gh[u][r] * sg[r];                                   // d t2
          gc[u][r] = gh[u][r] * t2[r] * sg[r] * (1.f - sg[r]);           // d (Wc c + bc)
        }
      }
      f4 a2[4], a1[4];
      co_load_a<4>(img, c.W2T[b], wave, id.lane, a2);
      co_load_a<4>(img, c.W1T[b], wave, id.lane, a1);
      float* ATn = ATb[atc ^ 1];
      co_store_T<NT>(GT0, RS, 16 * wave, id, ga, false, -1);
      co_store_T<NT>(GT1, RS, 16 * wave, id, gc, false, -1);
      co_store_T<NT>(ATn, RS, 16 * wave, id, t1, true, ones_h);
      f4 bg[NT][CO_WAVES], gr[NT];
      if (want_gx) {      // (uniform) context gradient of this block's gate: Wc^T g_c
        co_gather<NT>(ex, buf, wave, id.lane, gc, bg);
        if (wave * 16 < C) {
          f4 act[4];
          co_load_a<4>(img, c.WCT[b], wave, id.lane, act);
          co_gemm_h<NT, KSH>(act, bg, gxacc);
        }
      }
      co_gather<NT>(ex, buf, wave, id.lane, ga, bg);
#pragma unroll
      for (int u = 0; u < NT; ++u) gr[u] = zero4;
      co_gemm_h<NT, KSH>(a2, bg, gr);
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) ga[u][r] = t1[u][r] > 0.f ? gr[u][r] : 0.f;     // d t1
      {
        f4 acc[4], accb;
        co_dw<NT, 4>(GT0, ATn, RS, 16 * wave, 0, 4, id, acc, hb64 ? &accb : nullptr);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) co_write_tile(part, S.lin[3 + 3 * b], 16 * wave, nt, id, acc[nt]);
        if (hb64) co_write_bias(part, S.lin[3 + 3 * b], 16 * wave, id, accb);
        f4 accc[3];       // x-dim <= 32 plus the bias column: up to three n-tiles
        const int ntc = (C + 1 + 15) / 16;
        co_dw<NT, 3>(GT1, CT, RS, 16 * wave, S.d_id, ntc, id, accc, nullptr);
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
          if (nt < ntc) co_write_tile(part, S.lin[1 + 3 * b], 16 * wave, nt, id, accc[nt]);
      }
      wave_lds_fence();
      co_store_T<NT>(GT0, RS, 16 * wave, id, ga, false, -1);
      co_store_T<NT>(ATb[atc], RS, 16 * wave, id, hbk, true, ones_h);
      co_gather<NT>(ex, buf, wave, id.lane, ga, bg);
#pragma unroll
      for (int u = 0; u < NT; ++u) gr[u] = zero4;
      co_gemm_h<NT, KSH>(a1, bg, gr);
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) gh[u][r] += hbk[u][r] > 0.f ? gr[u][r] : 0.f;
      {
        f4 acc[4], accb;
        co_dw<NT, 4>(GT0, ATb[atc], RS, 16 * wave, 0, 4, id, acc, hb64 ? &accb : nullptr);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) co_write_tile(part, S.lin[2 + 3 * b], 16 * wave, nt, id, acc[nt]);
        if (hb64) co_write_bias(part, S.lin[2 + 3 * b], 16 * wave, id, accb);
      }
      wave_lds_fence();
      // ATb[atc] is being read by slower waves until the next barrier; the next block writes ATb[atc ^ 1] first
    }
    // ---- initial layer
    {
      co_store_T<NT>(GT0, RS, 16 * wave, id, gh, false, -1);
      f4 bg[NT][CO_WAVES];
      co_gather<NT>(ex, buf, wave, id.lane, gh, bg);
      if (wave == 0) {      // identity features receive W0[:, :d_id]^T g_h0
        f4 a0[4], gin[NT];
        co_load_a<4>(img, c.W0T, 0, id.lane, a0);
#pragma unroll
        for (int u = 0; u < NT; ++u) gin[u] = zero4;
        co_gemm_h<NT, KSH>(a0, bg, gin);
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = 4 * r + id.g;
            if (k < S.d_id) gys[(16 * u + id.j) * ZS + 2 * k + (1 - par)] += gin[u][r];
          }
      }
      if (want_gx && wave * 16 < C) {
        f4 act[4];
        co_load_a<4>(img, c.W0CT, wave, id.lane, act);
        co_gemm_h<NT, KSH>(act, bg, gxacc);
      }
      f4 acc[3];
      const int nt0 = (S.in0 + 1 + 15) / 16;
      co_dw<NT, 3>(GT0, CT, RS, 16 * wave, 0, nt0, id, acc, nullptr);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
        if (nt < nt0) co_write_tile(part, S.lin[0], 16 * wave, nt, id, acc[nt]);
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
  if (want_gx && wave * 16 < C) {      // through the kernel's own z-scoring: d c / d x = 1 / std
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cc = 16 * wave + 4 * r + id.g;
        const long long row = row0 + 16 * u + id.j;
        if (cc < C && row < n) grad_x[row * C + cc] = gxacc[u][r] / x_std[cc];
      }
  }
}

// ------------------------------------------------------------------------------------------------ launch helpers
struct CoFwdArgs {
  const float *cimg, *zstats, *theta, *x;
  long long n, x_rows;
  float *logp, *noise, *zst, *ast;
};
struct CoBwdArgs {
  const float *cimg, *zstats, *x;
  long long n, x_rows;
  const float* row_w;
  float uni_w;
  const float *z_last, *zst, *ast;
  float *partial, *grad_theta, *grad_x;
};

template <int K, int KSH, int NT>
static int co_launch_fwd(const NsfPlan& pl, const CoopPlan& cp, const CoFwdArgs& a, hipStream_t st) {
  auto kern = nsf_coop_fwd_kernel<K, KSH, NT>;
  const int lds_bytes = 4 * cp.lds_floats;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(cp.grid), dim3(64 * CO_WAVES), (size_t)lds_bytes, st, pl, cp, a.cimg, a.zstats, a.theta,
                     a.x, a.n, a.x_rows, a.logp, a.noise, a.zst, a.ast);
  return (int)hipGetLastError();
}
template <int K, int KSH, int NT>
static int co_launch_bwd(const NsfPlan& pl, const CoopPlan& cp, const CoBwdArgs& a, hipStream_t st) {
  auto kern = nsf_coop_bwd_kernel<K, KSH, NT>;
  const int lds_bytes = 4 * cp.lds_floats;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(cp.grid), dim3(64 * CO_WAVES), (size_t)lds_bytes, st, pl, cp, a.cimg, a.zstats, a.x, a.n,
                     a.x_rows, a.row_w, a.uni_w, a.z_last, a.zst, a.ast, a.partial, a.grad_theta, a.grad_x);
  return (int)hipGetLastError();
}

// one translation unit per bin count (parallel build): nsf_coop.hip holds K = 10, nsf_coop_k{4,5,8,16}.hip the rest
template <int K>
int co_fwd_k(const NsfPlan& pl, const CoopPlan& cp, const CoFwdArgs& a, hipStream_t st) {
  if (pl.KSH == 13) return cp.NT == 2 ? co_launch_fwd<K, 13, 2>(pl, cp, a, st) : co_launch_fwd<K, 13, 1>(pl, cp, a, st);
  return cp.NT == 2 ? co_launch_fwd<K, 16, 2>(pl, cp, a, st) : co_launch_fwd<K, 16, 1>(pl, cp, a, st);
}
template <int K>
int co_bwd_k(const NsfPlan& pl, const CoopPlan& cp, const CoBwdArgs& a, hipStream_t st) {
  if (pl.KSH == 13) return cp.NT == 2 ? co_launch_bwd<K, 13, 2>(pl, cp, a, st) : co_launch_bwd<K, 13, 1>(pl, cp, a, st);
  return cp.NT == 2 ? co_launch_bwd<K, 16, 2>(pl, cp, a, st) : co_launch_bwd<K, 16, 1>(pl, cp, a, st);
}
