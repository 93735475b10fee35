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
      if (k < 16 * c.KCQ) return k < pl.C ? gl[L.g_w + f * L.in + S.d_id + k] : 0.f;
      const int kz = k - 16 * c.KCQ;
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
// grid (T, blocks): one 256-float block of the transform's image per workgroup iteration.  A block belongs to ONE
// matrix (every matrix block is 256 floats), so the descriptor lookup is wave-uniform scalar code; only the bias region
// behind the matrices is searched per element.
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
  const int nblk = cp.img_floats >> 8;
  for (int blk = blockIdx.y; blk < nblk; blk += gridDim.y) {
    const int idx = (blk << 8) + threadIdx.x;
    float v = 0.f;
    if ((blk << 8) < c.o_bias) {
      int mi = -1;
      for (int m = 0; m < NMAT; ++m) {            // uniform: scalar compares
        const int sz = mats[m].mtiles * mats[m].quads * 256;
        if (sz > 0 && (blk << 8) >= mats[m].off && (blk << 8) < mats[m].off + sz) mi = m;
      }
      if (mi >= 0) v = co_mat_value(mats[mi], pl, S, c, gl, idx - mats[mi].off);
    } else {
      for (int b = 0; b < NBIAS; ++b) {
        const CoBias& B = bias[b];
        const int sz = 16 * B.mtiles;
        if (sz > 0 && idx >= B.off && idx < B.off + sz) v = co_bias_value(B, pl, S, gl, idx - B.off);
      }
      if (idx == c.o_ld) {                        // logabsdet of the LULinear = sum_i log(softplus(u_i) + eps)
        const int ntri = pl.D * (pl.D - 1) / 2;
        float a = 0.f;
        for (int i = 0; i < pl.D; ++i) a += logf(softplus_f(gl[S.g_lu + 2 * ntri + i]) + pl.lu_eps);
        v = a;
      }
    }
    img[idx] = v;
  }
}
#endif

// ------------------------------------------------------------------------------------------------ device helpers
// A fragments of one m-tile of a matrix: NQ 16-byte words per lane (block stride 256 floats)
template <int NQ>
__device__ __forceinline__ void co_load_a(const float* __restrict__ img, const CoMat& m, int mt, int lane, f4 (&a)[NQ]) {
  // NQ words are read unconditionally (branch-free): a matrix with fewer quads is followed by other readable image
  // data, and the K-steps of those words are never issued
  const f4* p = reinterpret_cast<const f4*>(img + m.off + mt * m.quads * 256) + lane;
#pragma unroll
  for (int q = 0; q < NQ; ++q) a[q] = p[q * 64];
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
// Every weight a wave needs is requested from L2 well before its use (the image is written by another XCD's pack
// workgroups: an L2 miss, ~900 cycles).  The hidden stages (k = 2 b: W1_b + the gate's Wc_b, k = 2 b + 1: W2_b) take
// their A fragments from three register sets requested two stages ahead; the final-layer tiles two tiles ahead;
// LULinear's factors and the NEXT transform's first sets under the spline.  The stage sequence is unrolled at compile
// time (generic lambda over the stage index, guarded by the run-time stage count) so that every register index is
// static and the compiler's s_waitcnt accounting is exact: no stage waits for more than it needs.
struct CoSet {
  f4 a[4];        // A fragments of the stage's hidden-K matrix (m-tile = wave)
  f4 ac[2];       // W1 stages: A fragments of the block's context layer (gate)
  f4 bias, biasc;
};
template <int K_>
struct CoIdx { static constexpr int value = K_; };

template <int KS>   // stage index: even = W1_b (+ gate), odd = W2_b
__device__ __forceinline__ void co_load_set(const float* __restrict__ img, const CoShape& c, int wave,
                                            const LaneId& id, CoSet& s) {
  constexpr int b = KS >> 1;
  if ((KS & 1) == 0) {
    co_load_a<4>(img, c.W1[b], wave, id.lane, s.a);
    co_load_a<2>(img, c.WC[b], wave, id.lane, s.ac);
    s.bias = co_load_bias(img, c.b1[b], wave, id.g);
    s.biasc = co_load_bias(img, c.bc[b], wave, id.g);
  } else {
    co_load_a<4>(img, c.W2[b], wave, id.lane, s.a);
    s.bias = co_load_bias(img, c.b2[b], wave, id.g);
  }
}
struct CoW0 {
  f4 a[3];
  f4 bias;
};
__device__ __forceinline__ void co_load_w0(const float* __restrict__ img, const CoShape& c, int wave, const LaneId& id,
                                           CoW0& w) {
  co_load_a<3>(img, c.W0, wave, id.lane, w.a);
  w.bias = co_load_bias(img, c.b0, wave, id.g);
}
struct CoWf {
  f4 a[4];
  f4 bias;
};
__device__ __forceinline__ void co_load_wf(const float* __restrict__ img, const CoShape& c, int mt, const LaneId& id,
                                           CoWf& w) {
  const int m = mt < c.nft ? mt : c.nft - 1;     // clamped instead of predicated: straight-line code
  co_load_a<4>(img, c.WF, m, id.lane, w.a);
  w.bias = co_load_bias(img, c.bf, m, id.g);
}
// acc[u] += A(context quads) * standardized context (K-steps of the context live in registers)
template <int NT>
__device__ __forceinline__ void co_gemm_ctx(const f4 (&a)[2], int kcq, const float (&cb)[NT][8], f4 (&acc)[NT]) {
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (q < kcq) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[u] = MFMA16(a[q][r], cb[u][4 * q + r], acc[u]);
    }
}

template <int K, int KSH, int NT>
__global__ void __launch_bounds__(64 * CO_WAVES, 1)
nsf_coop_fwd_kernel(const NsfPlan pl, const CoopPlan cp, const float* __restrict__ cimg,
                    const float* __restrict__ zstats, const float* __restrict__ theta, const float* __restrict__ x,
                    long long n, long long x_rows, float* __restrict__ logp, float* __restrict__ noise_out,
                    float* __restrict__ zst, float* __restrict__ ast, long long* __restrict__ dbg) {
  // debug timeline (SBI_AMD_TIMELINE): cycle stamps of workgroup 0's waves while they walk transform 1
#define TSC(i) do { if (dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && t == 1) \
    dbg[(threadIdx.x >> 6) * 64 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
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
  const int nstages = 2 * pl.NB;
  const int kcq = cp.sh[0].KCQ;

  // ---- weights of the first transform: requested before anything else
  CoW0 w0;
  CoSet S[3];
  co_load_w0(cimg, cp.sh[0], wave, id, w0);
  co_load_set<0>(cimg, cp.sh[0], wave, id, S[0]);
  co_load_set<1>(cimg, cp.sh[0], wave, id, S[1]);

  // ---- prologue: z-scored theta rows -> LDS; standardized context as B fragments (K-step s <-> c = 4 s + g)
  for (int i = tid; i < R * ZS + 16; i += 64 * CO_WAVES) zs[i] = 0.f;
  float cb[NT][8];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const long long row = row0 + 16 * u + id.j;
    const long long rs = row < n ? row : 0;
    const long long xr = (x_rows == n) ? rs : (x_rows == 1 ? 0 : rs % x_rows);
    float xv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int c = 4 * s + id.g;
      xv[s] = x[xr * C + (c < C ? c : 0)];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int c = 4 * s + id.g;
      const int cc = c < C ? c : 0;
      cb[u][s] = (c < C && row < n) ? (xv[s] - x_mean[cc]) / x_std[cc] : 0.f;
    }
  }
  __syncthreads();
  for (int i = tid; i < R * D; i += 64 * CO_WAVES) {
    const int r = i / D, d = i - r * D;
    const long long row = row0 + r;
    zs[r * ZS + d] = row < n ? theta[row * D + d] * th_scale[d] + th_shift[d] : 0.f;
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
    const ShapeDesc& S_ = pl.shape[par];
    const CoShape& c = cp.sh[par];
    const float* img = cimg + (long long)t * cp.img_floats;
    if (zst)
      for (int i = tid; i < R * D; i += 64 * CO_WAVES) {
        const int r = i / D, d = i - r * D;
        if (row0 + r < n) zst[((long long)t * n + row0 + r) * D + d] = zs[r * ZS + d];
      }
    TSC(0);
    if (2 < nstages) co_load_set<2>(img, c, wave, id, S[2]);
    // ---- initial layer: h = W0 [context ; z_id] + b0   (m-tile = wave)
    f4 h[NT];
    {
#pragma unroll
      for (int u = 0; u < NT; ++u) h[u] = w0.bias;
      const f4 wc[2] = {w0.a[0], w0.a[1]};
      co_gemm_ctx<NT>(wc, kcq, cb, h);
      const f4 az = kcq == 1 ? w0.a[1] : w0.a[2];     // the identity features' quad sits behind the context quads
#pragma unroll
      for (int sz = 0; sz < 2; ++sz) {
        const int kz = 4 * sz + id.g;
        const int kzc = kz < S_.d_id ? kz : 0;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const float zv = zs[(16 * u + id.j) * ZS + 2 * kzc + (1 - par)];
          h[u] = MFMA16(az[sz], kz < S_.d_id ? zv : 0.f, h[u]);
        }
      }
    }
    // The stash (what the backward pass reloads) is collected in registers and written in ONE burst right before the
    // spline: a store issued in the middle of the stage sequence would sit in the in-order memory counter in front of
    // every later weight load, and each wait for a prefetched set would also wait for the store's acknowledgement.
    f4 sv0[NT], sv[2 * NSF_MAX_NB][2][NT], pv[4][NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) sv0[u] = h[u];
    TSC(1);
    // ---- hidden stages
    CoWf wf0, wf1;
    f4 gate[NT], tt[NT];
    auto stage = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr int b = k >> 1;
      CoSet& cur = S[k % 3];
      f4 u1[NT], bg[NT][CO_WAVES];
      if (k + 2 < nstages) co_load_set<(k + 2 < 2 * NSF_MAX_NB ? k + 2 : 0)>(img, c, wave, id, S[(k + 2) % 3]);
      if (k == nstages - 2) {   // the wave's first two final-layer tiles, two stages ahead
        co_load_wf(img, c, wave, id, wf0);
        co_load_wf(img, c, wave + CO_WAVES, id, wf1);
      }
      if ((k & 1) == 0) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          gate[u] = cur.biasc;
#pragma unroll
          for (int r = 0; r < 4; ++r) tt[u][r] = fmaxf(h[u][r], 0.f);
        }
        co_gemm_ctx<NT>(cur.ac, kcq, cb, gate);
      }
      co_gather<NT>(ex, buf, wave, id.lane, tt, bg);
      TSC(2 + 2 * k);
#pragma unroll
      for (int u = 0; u < NT; ++u) u1[u] = cur.bias;
      co_gemm_h<NT, KSH>(cur.a, bg, u1);
      if ((k & 1) == 0) {
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gate[u][r] = sigmoid_f(gate[u][r]);
            tt[u][r] = fmaxf(u1[u][r], 0.f);
          }
#pragma unroll
        for (int u = 0; u < NT; ++u) { sv[k][0][u] = u1[u]; sv[k][1][u] = gate[u]; }    // t1 (pre-relu), sigmoid(gate)
      } else {
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) h[u][r] += u1[u][r] * gate[u][r];
#pragma unroll
        for (int u = 0; u < NT; ++u) { sv[k][0][u] = u1[u]; sv[k][1][u] = h[u]; }       // t2, h_{b+1}
      }
      TSC(3 + 2 * k);
    };
    stage(CoIdx<0>{});
    stage(CoIdx<1>{});
    if (2 < nstages) { stage(CoIdx<2>{}); stage(CoIdx<3>{}); }
    if (4 < nstages) { stage(CoIdx<4>{}); stage(CoIdx<5>{}); }
    if (6 < nstages) { stage(CoIdx<6>{}); stage(CoIdx<7>{}); }
    // ---- final layer: parameter tiles wave, wave + 4, ... -> staging rows pst[row][dim][3K-1 raw outputs]
    {
      f4 hb[NT][CO_WAVES];
      co_gather<NT>(ex, buf, wave, id.lane, h, hb);
      TSC(20);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mt = wave + CO_WAVES * i;
        if (mt < c.nft) {
          CoWf& w = (i & 1) ? wf1 : wf0;
          f4 acc[NT];
#pragma unroll
          for (int u = 0; u < NT; ++u) acc[u] = w.bias;
          co_gemm_h<NT, KSH>(w.a, hb, acc);
          if (i < 2) co_load_wf(img, c, mt + 2 * CO_WAVES, id, w);     // two tiles ahead
          const int dd = mt / PT, pt = mt - dd * PT;
#pragma unroll
          for (int u = 0; u < NT; ++u) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              pst[(16 * u + id.j) * cp.DSTR + dd * cp.PSW + 16 * pt + 4 * r + id.g] = acc[u][r];
            pv[i][u] = acc[u];
          }
        }
      }
    }
    TSC(21);
    if (ast) {      // the stash burst (plain stores: the backward workgroup of the same index runs on the same XCD)
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const long long t16 = (row0 >> 4) + u;
        if (t16 < nt16) {
          *co_slot(ast, cp, nt16, t, t16, wave, id.lane) = sv0[u];
#pragma unroll
          for (int k = 0; k < 2 * NSF_MAX_NB; ++k)
            if (k < nstages) {
              const int b = k >> 1, o = (k & 1) ? 4 : 0;      // even stage: t1 | gate, odd stage: t2 | h_{b+1}
              *co_slot(ast, cp, nt16, t, t16, cp.s_blk + 16 * b + o + wave, id.lane) = sv[k][0][u];
              *co_slot(ast, cp, nt16, t, t16, cp.s_blk + 16 * b + o + 8 + wave, id.lane) = sv[k][1][u];
            }
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (wave + CO_WAVES * i < c.nft) *co_slot(ast, cp, nt16, t, t16, cp.s_par + wave + CO_WAVES * i, id.lane) = pv[i][u];
        }
      }
    }
    // LULinear's factors and the next transform's first weights: requested before the spline, landed after it
    const f4 au = *(reinterpret_cast<const f4*>(img + c.U.off) + id.lane);
    const f4 al = *(reinterpret_cast<const f4*>(img + c.L.off) + id.lane);
    const f4 blu = co_load_bias(img, c.blu, 0, id.g);
    const float ld_lu = img[c.o_ld];
    {
      const int tn = t + 1 < pl.T ? t + 1 : t;           // (the last transform re-requests itself: harmless)
      const float* imgn = cimg + (long long)tn * cp.img_floats;
      const CoShape& cn = cp.sh[tn & 1];
      co_load_w0(imgn, cn, wave, id, w0);
      co_load_set<0>(imgn, cn, wave, id, S[0]);
      co_load_set<1>(imgn, cn, wave, id, S[1]);
    }
    __syncthreads();
    TSC(22);
    // ---- spline: task (row 16 u + j, dim 2 wave + slot) on the lane pair (lane, lane ^ 32)
    {
      const int slot = id.g & 1, part = id.g >> 1;
      const int dd_raw = 2 * wave + slot;
      const bool live = dd_raw < S_.d_tr;
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
    TSC(23);
    __syncthreads();
    TSC(24);
    // ---- LULinear: y = L (U z) + b as two chained 16 x 16 MFMA mat-vecs; row tile u on wave u
    if (wave < NT) {
      const int u = wave;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = MFMA16(au[s], zs[(16 * u + id.j) * ZS + 4 * s + id.g], acc);
      f4 yv = blu;
#pragma unroll
      for (int s = 0; s < 4; ++s) yv = MFMA16(al[s], acc[s], yv);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * r + id.g < D) zs[(16 * u + id.j) * ZS + 4 * r + id.g] = yv[r];
    }
    ld_const += ld_lu;
    TSC(25);
    __syncthreads();
    TSC(26);
  }
#undef TSC

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

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte store to a 4-byte-aligned address

// Weight-gradient tiles of one 16-wide slice of OUTPUT features, all from transposed LDS tiles (one ds_read_b128 per
// operand, row tile and n-tile):  acc[nt][r] of lane (g, j) = d W[out0 + j][16 nt + 4 g + r]
//   = sum over the workgroup's rows of Gt[out0 + j][row] * At[in_off + 16 nt + 4 g + r][row].
// The INPUT index runs along the registers, so a lane owns four consecutive entries of one row of the (row-major)
// weight gradient: one 16-byte store per tile instead of four scattered 4-byte ones (the partial slabs are the
// kernel's largest output: 78 KB per 16 * NT rows and transform).
template <int NT, int NNT>
__device__ __forceinline__ void co_dw(const float* __restrict__ Gt, const float* __restrict__ At, int RS, int out0,
                                      int in_off, int nnt, const LaneId& id, f4 (&acc)[NNT], f4* accb) {
  f4 b[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) b[u] = *reinterpret_cast<const f4*>(Gt + (out0 + id.j) * RS + 16 * u + 4 * id.g);
#pragma unroll
  for (int nt = 0; nt < NNT; ++nt) {
    acc[nt] = f4{0.f, 0.f, 0.f, 0.f};
    if (nt < nnt) {
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const f4 a = *reinterpret_cast<const f4*>(At + (in_off + 16 * nt + id.j) * RS + 16 * u + 4 * id.g);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[nt] = MFMA16(a[s], b[u][s], acc[nt]);
      }
    }
  }
  if (accb) {   // bias gradients when the layer input has no spare column for the ones row (hidden_features = 64):
    *accb = f4{0.f, 0.f, 0.f, 0.f};   // every row of the result is sum over rows of Gt[out0 + j][row]
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) *accb = MFMA16(1.f, b[u][s], *accb);
  }
}

// partial-gradient write-out of one tile of co_dw; column `in == L.in` of the activation tile is the ones row
__device__ __forceinline__ void co_write_tile(float* __restrict__ part, const LinDesc& L, int out0, int nt,
                                              const LaneId& id, const f4& acc) {
  const int out = out0 + id.j;
  const int in = 16 * nt + 4 * id.g;
  if (out >= L.out) return;
  float* w = part + L.g_w + out * L.in + in;
  if (in + 3 < L.in) {
    *reinterpret_cast<f4u*>(w) = f4u{acc[0], acc[1], acc[2], acc[3]};
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (in + r < L.in) w[r] = acc[r];
      else if (in + r == L.in) part[L.g_b + out] = acc[r];
    }
  }
}
__device__ __forceinline__ void co_write_bias(float* __restrict__ part, const LinDesc& L, int out0, const LaneId& id,
                                              const f4& accb) {
  const int out = out0 + id.j;
  if (id.g == 0 && out < L.out) part[L.g_b + out] = accb[0];
}

// transposed hidden-K matrix of backward stage k (execution order: k even W2^T of block NB-1-k/2, k odd its W1^T)
template <int KS>
__device__ __forceinline__ void co_load_tset(const float* __restrict__ img, const CoShape& c, int NB, int wave,
                                             int lane, f4 (&a)[4]) {
  const int b = NB - 1 - (KS >> 1) > 0 ? NB - 1 - (KS >> 1) : 0;     // (clamped: a stage past the last re-reads block 0)
  co_load_a<4>(img, (KS & 1) ? c.W1T[b] : c.W2T[b], wave, lane, a);
}

template <int K, int KSH, int NT>
__global__ void __launch_bounds__(64 * CO_WAVES, 1)
nsf_coop_bwd_kernel(const NsfPlan pl, const CoopPlan cp, const float* __restrict__ cimg,
                    const float* __restrict__ zstats, const float* __restrict__ x, long long n, long long x_rows,
                    const float* __restrict__ row_w, const float uni_w, const float* __restrict__ z_last,
                    const float* __restrict__ zst, const float* __restrict__ ast, float* __restrict__ partial,
                    float* __restrict__ grad_theta, float* __restrict__ grad_x) {
  constexpr int PT = (3 * K - 1 + 15) / 16;
  constexpr int R = 16 * NT;
  constexpr int NZ = (R * 16 + 64 * CO_WAVES - 1) / (64 * CO_WAVES);   // state values per thread (theta-dim <= 16)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const LaneId id = make_lane();
  const int D = pl.D, C = pl.C, ZS = cp.ZS, RS = cp.RS, NB = pl.NB;
  float* zs = lds + cp.o_zs;
  float* gys = lds + cp.o_gys;
  float* gzs = lds + cp.o_gzs;
  float* wrow = lds + cp.o_w;
  float* pst = lds + cp.o_pst;
  float* ex = lds + cp.o_ex;
  float* GT0 = lds + cp.o_gt;
  float* GT1 = GT0 + 64 * RS;
  float* AT0 = lds + cp.o_at;
  float* AT1 = AT0 + 65 * RS;
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
  const bool want_gx = grad_x != nullptr;
  // clamped stash tiles: wave-tiles past the last row were never written by the forward pass
  long long t16c[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) t16c[u] = (row0 >> 4) + u < nt16 ? (row0 >> 4) + u : nt16 - 1;
  auto slot_at = [&](int t, int u, int sl) -> f4 {
    return *(reinterpret_cast<const f4*>(ast + (((long long)t * nt16 + t16c[u]) * cp.slots + sl) * 256) + id.lane);
  };
  // What a transform needs FIRST is requested a transform ahead (for t = T - 1: before the prologue touches LDS):
  // its input state rows, its spline-parameter tiles, LULinear's factors.
  float zin[NZ];
  f4 ptile[4][NT];
  f4 a_lt, a_ut, a_u;
  auto request_entry = [&](int t) {
    const CoShape& c = cp.sh[t & 1];
    const float* img = cimg + (long long)t * cp.img_floats;
#pragma unroll
    for (int q = 0; q < NZ; ++q) {
      const int i = tid + 64 * CO_WAVES * q;
      const int r = i / D, d = i - r * D;
      const long long row = row0 + r;
      zin[q] = (i < R * D && row < n) ? zst[((long long)t * n + row) * D + d] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mt = wave + CO_WAVES * i;
      if (mt < c.nft) {
#pragma unroll
        for (int u = 0; u < NT; ++u) ptile[i][u] = slot_at(t, u, cp.s_par + mt);
      }
    }
    a_lt = *(reinterpret_cast<const f4*>(img + c.LT.off) + id.lane);
    a_ut = *(reinterpret_cast<const f4*>(img + c.UT.off) + id.lane);
    a_u = *(reinterpret_cast<const f4*>(img + c.U.off) + id.lane);
  };
  request_entry(pl.T - 1);

  for (int i = tid; i < cp.o_w - cp.o_zs; i += 64 * CO_WAVES) lds[cp.o_zs + i] = 0.f;   // state rows incl. padding
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
  // d loss / d context (a trainable embedding net in front of the flow): waves 0 and 1 own context features
  // [16 wave, 16 wave + 16) and accumulate W0[:, ctx]^T g_h0 + sum_b Wc_b^T g_c over all transforms
  f4 gxacc[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) gxacc[u] = zero4;
  __syncthreads();

  for (int t = pl.T - 1; t >= 0; --t) {
    const int par = t & 1;
    const ShapeDesc& S = pl.shape[par];
    const CoShape& c = cp.sh[par];
    const float* img = cimg + (long long)t * cp.img_floats;
    float* part = partial + ((long long)t * gridDim.x + blockIdx.x) * cp.PLP;
    const LinDesc& LF = S.lin[S.fin];
    // ---- requests whose results are needed after the spline: h_last, Wf^T, the last block's stash, the first two
    //      transposed hidden matrices
    struct BSt { f4 t1[NT], t2[NT], sg[NT], hb[NT]; };   // a block's stash: t1 (pre-relu), t2, sigmoid(gate), input
    f4 hl[NT], wft[8][PT], T[3][4];
    BSt B[2];
#pragma unroll
    for (int u = 0; u < NT; ++u) hl[u] = slot_at(t, u, cp.s_blk + 16 * (NB - 1) + 12 + wave);
    {
      const f4* ap = reinterpret_cast<const f4*>(img + c.WFT.off + wave * c.WFT.quads * 256) + id.lane;
#pragma unroll
      for (int dd = 0; dd < 8; ++dd)
        if (dd < S.d_tr) {
#pragma unroll
          for (int q = 0; q < PT; ++q) wft[dd][q] = ap[(dd * PT + q) * 64];
        }
    }
    auto request_block = [&](int b, BSt& st) {
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        st.t1[u] = slot_at(t, u, cp.s_blk + 16 * b + wave);
        st.t2[u] = slot_at(t, u, cp.s_blk + 16 * b + 4 + wave);
        st.sg[u] = slot_at(t, u, cp.s_blk + 16 * b + 8 + wave);
        st.hb[u] = slot_at(t, u, b == 0 ? wave : cp.s_blk + 16 * (b - 1) + 12 + wave);
      }
    };
    request_block(NB - 1, B[0]);
    co_load_tset<0>(img, c, NB, wave, id.lane, T[0]);
    co_load_tset<1>(img, c, NB, wave, id.lane, T[1]);
    // ---- P0: state rows + conditioner-input tile (from the registers requested a transform ago), spline
    //      parameters -> LDS; LULinear backward
#pragma unroll
    for (int q = 0; q < NZ; ++q) {
      const int i = tid + 64 * CO_WAVES * q;
      if (i < R * D) {
        const int r = i / D, d = i - r * D;
        zs[r * ZS + d] = zin[q];
        if ((d & 1) == (1 - par)) CT[(d >> 1) * RS + r] = zin[q];     // identity feature k = (d - (1 - par)) / 2
      }
    }
    for (int i = tid; i < (cp.ct_rows - S.d_id) * R; i += 64 * CO_WAVES) {
      const int k = S.d_id + i / R, r = i % R;
      CT[k * RS + r] = k < S.in0 ? ctx[(k - S.d_id) * R + r] : (k == S.in0 ? 1.f : 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mt = wave + CO_WAVES * i;
      if (mt < c.nft) {
        const int dd = mt / PT, pt = mt - dd * PT;
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pst[(16 * u + id.j) * cp.DSTR + dd * cp.PSW + 16 * pt + 4 * r + id.g] = ptile[i][u][r];
      }
    }
    if (wave < NT) {   // g_u = L^T g_z, g_y = U^T g_u for row tile `wave`
      const int u = wave;
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
    co_store_T<NT>(AT0, RS, 16 * wave, id, hl, false, ones_h);
    f4 gh[NT];
    {
      f4 acc0[NT], acc1[NT];     // two accumulators per row tile: the chain is not bound by the MFMA latency
#pragma unroll
      for (int u = 0; u < NT; ++u) { acc0[u] = zero4; acc1[u] = zero4; }
#pragma unroll
      for (int dd = 0; dd < 8; ++dd) {
        if (dd < S.d_tr) {
#pragma unroll
          for (int q = 0; q < PT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int u = 0; u < NT; ++u) {
                const float bv = pst[(16 * u + id.j) * cp.DSTR + dd * cp.PSW + 16 * q + 4 * r + id.g];
                if ((dd * PT + q) & 1) acc1[u] = MFMA16(wft[dd][q][r], bv, acc1[u]);
                else acc0[u] = MFMA16(wft[dd][q][r], bv, acc0[u]);
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
#pragma unroll 1
    for (int mt = wave; mt < c.nft; mt += CO_WAVES) {
      const int dd = mt / PT, pt = mt - dd * PT;
      f4 acc[4], accb = zero4;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = zero4;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        float bv[4];    // B side = g_p: parameter 16 pt + j of rows 16 u + 4 g + s
#pragma unroll
        for (int s = 0; s < 4; ++s) bv[s] = pst[(16 * u + 4 * id.g + s) * cp.DSTR + dd * cp.PSW + 16 * pt + id.j];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const f4 a = *reinterpret_cast<const f4*>(AT0 + (16 * nt + id.j) * RS + 16 * u + 4 * id.g);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[nt] = MFMA16(a[s], bv[s], acc[nt]);
        }
        if (hb64) {
#pragma unroll
          for (int s = 0; s < 4; ++s) accb = MFMA16(1.f, bv[s], accb);
        }
      }
      const int p = 16 * pt + id.j;
      if (p < pl.P) {
        const int out = dd * pl.P + p;
        float* w = part + LF.g_w + out * LF.in;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int in = 16 * nt + 4 * id.g;
          if (in + 3 < LF.in) {
            *reinterpret_cast<f4u*>(w + in) = f4u{acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]};
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (in + r < LF.in) w[in + r] = acc[nt][r];
              else if (in + r == LF.in) part[LF.g_b + out] = acc[nt][r];
            }
          }
        }
        if (hb64 && id.g == 0) part[LF.g_b + out] = accb[0];
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
          const int i = id.j, k = 4 * id.g + r;
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
    // the transform below: its state rows, parameter tiles and LU factors land under this transform's block phase
    if (t > 0) request_entry(t - 1);
    // ---- residual blocks, last -> first, unrolled at compile time over the execution ordinal i (block b = NB-1-i):
    //      stage 2 i is W2_b^T, stage 2 i + 1 is W1_b^T; transposed matrices two stages ahead, the stash one block ahead
    f4 a0t[4];     // W0^T (identity columns), wave 0
    auto block = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int b = NB - 1 - i;
      BSt& cur = B[i & 1];
      f4 ga[NT], gc[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sg = cur.sg[u][r];
          ga[u][r] = gh[u][r] * sg;                                     // d t2
          gc[u][r] = gh[u][r] * cur.t2[u][r] * sg * (1.f - sg);         // d (Wc c + bc)
        }
      co_load_tset<2 * i + 2>(img, c, NB, wave, id.lane, T[(2 * i + 2) % 3]);
      request_block(b > 0 ? b - 1 : 0, B[(i + 1) & 1]);
      if (b == 0) co_load_a<4>(img, c.W0T, 0, id.lane, a0t);
      co_store_T<NT>(GT0, RS, 16 * wave, id, ga, false, -1);
      co_store_T<NT>(GT1, RS, 16 * wave, id, gc, false, -1);
      co_store_T<NT>(AT1, RS, 16 * wave, id, cur.t1, true, ones_h);
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
      co_gemm_h<NT, KSH>(T[(2 * i) % 3], bg, gr);
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) ga[u][r] = cur.t1[u][r] > 0.f ? gr[u][r] : 0.f;     // d t1
      co_load_tset<2 * i + 3>(img, c, NB, wave, id.lane, T[(2 * i + 3) % 3]);
      {
        f4 acc[4], accb;
        co_dw<NT, 4>(GT0, AT1, RS, 16 * wave, 0, 4, id, acc, hb64 ? &accb : nullptr);
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
      co_store_T<NT>(AT0, RS, 16 * wave, id, cur.hb, true, ones_h);
      co_gather<NT>(ex, buf, wave, id.lane, ga, bg);
#pragma unroll
      for (int u = 0; u < NT; ++u) gr[u] = zero4;
      co_gemm_h<NT, KSH>(T[(2 * i + 1) % 3], bg, gr);
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) gh[u][r] += cur.hb[u][r] > 0.f ? gr[u][r] : 0.f;
      {
        f4 acc[4], accb;
        co_dw<NT, 4>(GT0, AT0, RS, 16 * wave, 0, 4, id, acc, hb64 ? &accb : nullptr);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) co_write_tile(part, S.lin[2 + 3 * b], 16 * wave, nt, id, acc[nt]);
        if (hb64) co_write_bias(part, S.lin[2 + 3 * b], 16 * wave, id, accb);
      }
      wave_lds_fence();
      // (AT0 / AT1 are re-written only after the next barrier: no reader of this block is still on them)
    };
    block(CoIdx<0>{});
    if (1 < NB) block(CoIdx<1>{});
    if (2 < NB) block(CoIdx<2>{});
    if (3 < NB) block(CoIdx<3>{});
    // ---- initial layer
    {
      co_store_T<NT>(GT0, RS, 16 * wave, id, gh, false, -1);
      f4 bg[NT][CO_WAVES];
      co_gather<NT>(ex, buf, wave, id.lane, gh, bg);
      if (wave == 0) {      // identity features receive W0[:, :d_id]^T g_h0
        f4 gin[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) gin[u] = zero4;
        co_gemm_h<NT, KSH>(a0t, bg, gin);
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
  long long* dbg;
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
                     a.x, a.n, a.x_rows, a.logp, a.noise, a.zst, a.ast, a.dbg);
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
