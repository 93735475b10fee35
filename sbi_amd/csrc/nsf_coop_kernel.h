#pragma once
// nsf_coop_kernel.h -- the cooperative (latency-oriented) NSF kernels for small batches on gfx950; design and
// reference path in nsf_coop.h.  Three kernels:
//   nsf_coop_pack_kernel      flat parameters -> fragment-ordered image (forward AND transposed matrices)
//   nsf_coop_fwd_kernel       log p (+ noise, + the stash the backward pass needs) for all T transforms
//   nsf_coop_bwd_kernel       all T transforms backward in one launch: d loss / d theta, per-workgroup partial
//                             weight-gradient slabs (tile order, nsf_coop_reduce_kernel sums them)
#include <hip/hip_runtime.h>
#include "nsf_coop.h"
#include "mcmc_tick.h"
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

// entry (i, k) of U^-1 (upper) or L^-1 by substitution down / up column k, in double (D <= 16)
__device__ __forceinline__ float co_lu_inv_entry(const float* __restrict__ lu, int D, float eps, bool upper, int i,
                                                 int k) {
  if (i >= D || k >= D) return 0.f;
  double x[16];
  if (!upper) {                 // L unit lower triangular: x_k = 1, x_i = - sum_{k <= j < i} L_ij x_j
    if (i < k) return 0.f;
    x[k] = 1.0;
    for (int r = k + 1; r <= i; ++r) {
      double a = 0.0;
      for (int j = k; j < r; ++j) a -= (double)co_lu_entry(lu, D, eps, false, r, j) * x[j];
      x[r] = a;
    }
    return (float)x[i];
  }
  if (i > k) return 0.f;        // U upper triangular: x_k = 1 / U_kk, x_i = - (sum_{i < j <= k} U_ij x_j) / U_ii
  x[k] = 1.0 / (double)co_lu_entry(lu, D, eps, true, k, k);
  for (int r = k - 1; r >= i; --r) {
    double a = 0.0;
    for (int j = r + 1; j <= k; ++j) a -= (double)co_lu_entry(lu, D, eps, true, r, j) * x[j];
    x[r] = a / (double)co_lu_entry(lu, D, eps, true, r, r);
  }
  return (float)x[i];
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
    case CO_K_UI: return co_lu_inv_entry(gl + S.g_lu, pl.D, pl.lu_eps, true, mi, k);
    case CO_K_LI: return co_lu_inv_entry(gl + S.g_lu, pl.D, pl.lu_eps, false, mi, k);
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
                     float* __restrict__ cimg, const int inverses) {
  // inverses = 0: everything but the explicit LU inverses (what log_prob and the training pass read; re-packed every
  // optimizer step); 1: only U^-1 / L^-1 (sampling direction of the wide nets: a column substitution in fp64 per entry,
  // four times the cost of the rest of the image, so it is not paid per training step)
  const int t = blockIdx.x;
  const int par = t & 1;
  const ShapeDesc& S = pl.shape[par];
  const CoShape& c = cp.sh[par];
  const float* gl = params + pl.g_layer[t];
  float* img = cimg + (long long)t * cp.img_floats;
  const CoMat* mats = &c.W0;                      // the CoMat members are laid out contiguously ...
  constexpr int NMAT = 11 + 6 * NSF_MAX_NB;     // W0 WC[] W1[] W2[] WF U L | WFT W1T[] W2T[] W0T UT LT | WCT[] W0CT | UI LI
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
      const bool is_inv = mi >= 0 && (mats[mi].kind == CO_K_UI || mats[mi].kind == CO_K_LI);
      if (is_inv != (inverses != 0)) continue;
      if (mi >= 0) v = co_mat_value(mats[mi], pl, S, c, gl, idx - mats[mi].off);
    } else {
      if (inverses) continue;
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

// grad[p] = sum over workgroups of the partial slabs in a fixed association (deterministic, no atomics).  The kernel
// walks the slab in ITS order (one thread per slab word: the loads of a wave are contiguous) and scatters each sum to
// the parameter the word belongs to (CoShape::dw_tb); padding words are skipped.  Finishes LULinear's diagonal as the
// throughput path's nsf_grad_reduce_kernel does:  d/d(unconstrained_upper_diag_i) =
//   (dL/dU_ii + (sum_n dL/dlogabsdet_n) / U_ii) * sigmoid(unconstrained_i).   Rider: loss_out = -log p.
#ifndef CO_RED_GROUPS
#define CO_RED_GROUPS 8
#endif
__global__ void __launch_bounds__(64 * CO_RED_GROUPS)
nsf_coop_reduce_kernel(const NsfPlan pl, const CoopPlan cp, const float* __restrict__ params,
                       const float* __restrict__ partial, float* __restrict__ grad, const float* __restrict__ logp,
                       float* __restrict__ loss_out, long long n_rows, float* __restrict__ sq_out) {
  if (loss_out)
    for (long long i = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < n_rows;
         i += (long long)gridDim.x * gridDim.y * blockDim.x)
      loss_out[i] = -logp[i];
  __shared__ f4 red[CO_RED_GROUPS][64];
  __shared__ float red_sgl[CO_RED_GROUPS];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int t = blockIdx.y;
  const ShapeDesc& S = pl.shape[t & 1];
  const CoShape& c = cp.sh[t & 1];
  const int pos = 4 * (blockIdx.x * 64 + lane);       // four consecutive words of the slab: one lane's 16-byte store
  const int D = pl.D, ntri = D * (D - 1) / 2;
  int li[4] = {-1, -1, -1, -1};                       // parameter (relative to the transform's block) of each word
  if (pos < c.dw_tail) {
    const int tile = pos >> 8, l = (pos >> 2) & 63;
    int k = 0;
    for (int kk = 1; kk <= S.fin; ++kk) k = tile >= c.dw_tb[kk] ? kk : k;
    const LinDesc& L = S.lin[k];
    const int rel = tile - c.dw_tb[k];
    const int mt = rel / c.dw_nnt[k], nt = rel - mt * c.dw_nnt[k];
    const int j = l & 15, g = l >> 4;
    int out = 16 * mt + j;
    if (k == S.fin) {
      const int dd = mt / pl.PT, p = 16 * (mt - dd * pl.PT) + j;
      out = p < pl.P ? dd * pl.P + p : L.out;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int in = 16 * nt + 4 * g + r;
      if (out < L.out && in <= L.in) li[r] = in < L.in ? L.g_w + out * L.in + in : L.g_b + out;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (pos + r - c.dw_tail < 2 * ntri + 2 * D) li[r] = S.g_lu + (pos + r - c.dw_tail);
  }
  const bool live = (li[0] & li[1] & li[2] & li[3]) >= 0 || li[0] >= 0 || li[1] >= 0 || li[2] >= 0 || li[3] >= 0;
  const float* base = partial + (long long)t * cp.grid * cp.PLP;
  f4 a = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    f4 acc4[4] = {a, a, a, a};
    int w = grp;
    for (; w + 3 * CO_RED_GROUPS < cp.grid; w += 4 * CO_RED_GROUPS) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc4[u] += *reinterpret_cast<const f4*>(base + (long long)(w + u * CO_RED_GROUPS) * cp.PLP + pos);
    }
    for (; w < cp.grid; w += CO_RED_GROUPS) acc4[0] += *reinterpret_cast<const f4*>(base + (long long)w * cp.PLP + pos);
    a = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
  }
  red[grp][lane] = a;
  // sum_n d loss / d logabsdet_n (one word per slab): every block needs it only if it holds LU-diagonal words; cheap
  {
    float sgl = 0.f;
    for (int w2 = grp * 64 + lane; w2 < cp.grid; w2 += 64 * CO_RED_GROUPS) sgl += base[(long long)w2 * cp.PLP + c.dw_tail + 2 * ntri + 2 * D];
    for (int off = 32; off > 0; off >>= 1) sgl += __shfl_xor(sgl, off);
    if (lane == 0) red_sgl[grp] = sgl;
  }
  __syncthreads();
  if (grp == 0) {
    float sq = 0.f;     // rider: this workgroup's share of |grad|^2 (the clip's norm: no separate pass over grad)
    if (live) {
      f4 tot = {0.f, 0.f, 0.f, 0.f};
      float tsg = 0.f;
#pragma unroll
      for (int g = 0; g < CO_RED_GROUPS; ++g) { tot += red[g][lane]; tsg += red_sgl[g]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (li[r] < 0) continue;
        const int idx = pl.g_layer[t] + li[r];
        float v = tot[r];
        if (li[r] >= S.g_lu + 2 * ntri && li[r] < S.g_lu + 2 * ntri + D) {
          const float ud = params[idx];
          const float uii = softplus_f(ud) + pl.lu_eps;
          v = (v + tsg / uii) * (1.f / (1.f + expf(-ud)));
        }
        grad[idx] = v;
        sq += v * v;
      }
    }
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off);     // fixed butterfly: deterministic
    if (lane == 0 && sq_out) sq_out[blockIdx.y * gridDim.x + blockIdx.x] = sq;
  }
}
#endif

// ------------------------------------------------------------------------------------------------ device helpers
// The forward / backward kernels read nothing but the compact constant block CoK (nsf_coop.h): image offsets are
// plain integers (block b of a per-block matrix: offset + b * stride), every matrix block is 256 floats (64 lanes x 4
// K-steps), an m-tile of a matrix with Q K-quads is Q consecutive blocks.
template <int NQ>
__device__ __forceinline__ void co_load_a(const float* __restrict__ p, int lane, f4 (&a)[NQ]) {
  // NQ 16-byte words per lane, read unconditionally (branch-free): a matrix with fewer quads is followed by other
  // readable image data, and the K-steps of those words are never issued
  const f4* q = reinterpret_cast<const f4*>(p) + lane;
#pragma unroll
  for (int i = 0; i < NQ; ++i) a[i] = q[i * 64];
}
__device__ __forceinline__ f4 co_load_bias(const float* __restrict__ p, int mt, int g) {
  return *reinterpret_cast<const f4*>(p + 16 * mt + 4 * g);
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

template <int K_>
struct CoIdx { static constexpr int value = K_; };

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
template <int KS>   // stage index: even = W1_b (+ gate), odd = W2_b
__device__ __forceinline__ void co_load_set(const float* __restrict__ img, const CoK& k, const CoKP& kp, int wave,
                                            const LaneId& id, CoSet& s) {
  constexpr int b = KS >> 1;
  if ((KS & 1) == 0) {
    co_load_a<4>(img + kp.w10 + b * k.sA + wave * 1024, id.lane, s.a);
    co_load_a<2>(img + kp.wc0 + b * k.sA + wave * k.KCQ * 256, id.lane, s.ac);
    s.bias = co_load_bias(img + kp.b10 + b * k.sB, wave, id.g);
    s.biasc = co_load_bias(img + kp.bc0 + b * k.sB, wave, id.g);
  } else {
    co_load_a<4>(img + kp.w20 + b * k.sA + wave * 1024, id.lane, s.a);
    s.bias = co_load_bias(img + kp.b20 + b * k.sB, wave, id.g);
  }
}
struct CoW0 {
  f4 a[3];
  f4 bias;
};
__device__ __forceinline__ void co_load_w0(const float* __restrict__ img, const CoK& k, const CoKP& kp, int wave,
                                           const LaneId& id, CoW0& w) {
  co_load_a<3>(img + kp.w0 + wave * (k.KCQ + 1) * 256, id.lane, w.a);
  w.bias = co_load_bias(img + kp.b0, wave, id.g);
}
struct CoWf {
  f4 a[4];
  f4 bias;
};
__device__ __forceinline__ void co_load_wf(const float* __restrict__ img, const CoKP& kp, int mt, const LaneId& id,
                                           CoWf& w) {
  const int m = mt < kp.nft ? mt : kp.nft - 1;     // clamped instead of predicated: straight-line code
  co_load_a<4>(img + kp.wf + m * 1024, id.lane, w.a);
  w.bias = co_load_bias(img + kp.bf, m, id.g);
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

// MC = true: the PERSISTENT slice sampler (sbi_amd_mcmc_slice_run).  A workgroup owns 16 chains for the whole launch
// and alternates, `nticks` times, the log-density of the chains' next evaluation points (this forward pass, one x_o)
// with one tick of their state machines (mcmc_tick.h, run by the 16 threads that hold the rows' log-densities) --
// chains never interact, so there is no rendezvous between workgroups and no launch per tick.  The reference runs
// the same loop in Python around batched potential calls (sbi/samplers/mcmc/slice_numpy.py:353-587).
struct McArgs {
  int num_samples, tuning, nticks, kind;
  float max_width;
  unsigned long long seed, tick0;
  const float *p0, *p1;                       // the constrained map's parameters (sbi_amd_mcmc_to_constrained)
  float *x, *next_param, *width, *fstate, *samples, *theta_next, *lad_next, *logp_buf;
  int *order, *istate, *done_count;
};
template <int K, int KSH, int NT, bool LEAN, bool MC = false>
__global__ void __launch_bounds__(64 * CO_WAVES, LEAN ? 2 : 1)
nsf_coop_fwd_kernel(const CoK k, const float* __restrict__ cimg, const float* __restrict__ zstats,
                    const float* __restrict__ theta, const float* __restrict__ x, long long n, long long x_rows,
                    float* __restrict__ logp, float* __restrict__ noise_out, float* __restrict__ zst,
                    float* __restrict__ ast, long long* __restrict__ dbg, const McArgs mc) {
  static_assert(!MC || (NT == 1 && !LEAN), "the persistent sampler runs one-tile workgroups");
  // debug timeline (SBI_AMD_TIMELINE): cycle stamps of workgroup 0's waves while they walk transform 1
#ifdef NSF_DEBUG
#define TSC(i) do { if (dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && t == 1) \
    dbg[(threadIdx.x >> 6) * 64 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TSC(i) do { } while (0)
#endif
  constexpr int PT = (3 * K - 1 + 15) / 16;
  constexpr int R = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const LaneId id = make_lane();
  const int D = k.D, C = k.C, ZS = k.ZS;
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
  const int nstages = 2 * k.NB;
  const int kcq = k.KCQ;

  // ---- weights of the first transform: requested before anything else
  // LEAN (two workgroups per CU, <= 256 registers): two sets requested ONE stage ahead, the stash written where it is
  // produced -- the co-resident workgroup's waves cover the latencies the deeper pipeline of the default variant hides
  constexpr int NS = LEAN ? 2 : 3, PD = NS - 1;
  CoW0 w0;
  CoSet S[NS];
  co_load_w0(cimg, k, k.p[0], wave, id, w0);
  co_load_set<0>(cimg, k, k.p[0], wave, id, S[0]);
  if (!LEAN) co_load_set<1>(cimg, k, k.p[0], wave, id, S[1 % NS]);

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
  __shared__ float mc_theta[MC ? 16 * 16 : 1];      // MC: the chains' next evaluation points (constrained space)
  float ld_acc[NT];
  float ld_const0 = 0.f;
  for (int d = 0; d < D; ++d) ld_const0 += logf(fabsf(th_scale[d]));
  float ld_const = ld_const0;
  int buf = 0;
  const int nticks = MC ? mc.nticks : 1;
  for (int tick = 0; tick < nticks; ++tick) {
  for (int i = tid; i < R * D; i += 64 * CO_WAVES) {
    const int r = i / D, d = i - r * D;
    const long long row = row0 + r;
    const float th = (MC && tick > 0) ? mc_theta[r * 16 + d] : (row < n ? theta[row * D + d] : 0.f);
    zs[r * ZS + d] = row < n ? th * th_scale[d] + th_shift[d] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < NT; ++u) ld_acc[u] = 0.f;
  ld_const = ld_const0;
  // stash addresses of this wave's fragments: slot s of transform t, row tile u is (abase[u] + t * astride + s * 256)
  float* abase[NT];
  const long long astride = nt16 * k.slots * 256;
#pragma unroll
  for (int u = 0; u < NT; ++u)
    abase[u] = (ast && (row0 >> 4) + u < nt16) ? ast + ((row0 >> 4) + u) * k.slots * 256 + 4 * id.lane : nullptr;
  __syncthreads();

  for (int t = 0; t < k.T; ++t) {
    const int par = t & 1;
    const CoKP& kp = k.p[par];
    const float* img = cimg + (long long)t * k.img_floats;
    if (zst)
      for (int i = tid; i < R * D; i += 64 * CO_WAVES) {
        const int r = i / D, d = i - r * D;
        if (row0 + r < n) zst[((long long)t * n + row0 + r) * D + d] = zs[r * ZS + d];
      }
    TSC(0);
    if (!LEAN && 2 < nstages) co_load_set<2>(img, k, kp, wave, id, S[2 % NS]);
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
        const int kzc = kz < kp.d_id ? kz : 0;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const float zv = zs[(16 * u + id.j) * ZS + 2 * kzc + (1 - par)];
          h[u] = MFMA16(az[sz], kz < kp.d_id ? zv : 0.f, h[u]);
        }
      }
    }
    // The stash (what the backward pass reloads) is collected in registers and written in ONE burst right before the
    // spline: a store issued in the middle of the stage sequence would sit in the in-order memory counter in front of
    // every later weight load, and each wait for a prefetched set would also wait for the store's acknowledgement.
    f4 sv0[NT], sv[LEAN ? 1 : 2 * NSF_MAX_NB][2][NT], pv[LEAN ? 1 : 4][NT];
    float* ab_t[NT];     // (LEAN only)
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      sv0[u] = h[u];
      if constexpr (LEAN) {
        ab_t[u] = (ast && abase[u]) ? abase[u] + t * astride : nullptr;
        if (ab_t[u]) *reinterpret_cast<f4*>(ab_t[u] + wave * 256) = h[u];
      }
    }
    TSC(1);
    // ---- hidden stages
    CoWf wf0, wf1;
    f4 gate[NT], tt[NT];
    auto stage = [&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      CoSet& cur = S[ks % NS];
      f4 u1[NT], bg[NT][CO_WAVES];
      if (ks + PD < nstages) co_load_set<(ks + PD < 2 * NSF_MAX_NB ? ks + PD : 0)>(img, k, kp, wave, id, S[(ks + PD) % NS]);
      if (ks == nstages - PD) {   // the wave's first two final-layer tiles, two stages (LEAN: one) ahead
        co_load_wf(img, kp, wave, id, wf0);
        co_load_wf(img, kp, wave + CO_WAVES, id, wf1);
      }
      if ((ks & 1) == 0) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          gate[u] = cur.biasc;
#pragma unroll
          for (int r = 0; r < 4; ++r) tt[u][r] = fmaxf(h[u][r], 0.f);
        }
        co_gemm_ctx<NT>(cur.ac, kcq, cb, gate);
      }
      co_gather<NT>(ex, buf, wave, id.lane, tt, bg);
      TSC(2 + 2 * ks);
#pragma unroll
      for (int u = 0; u < NT; ++u) u1[u] = cur.bias;
      co_gemm_h<NT, KSH>(cur.a, bg, u1);
      if ((ks & 1) == 0) {
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gate[u][r] = sigmoid_gate(gate[u][r]);
            tt[u][r] = fmaxf(u1[u][r], 0.f);
          }
#pragma unroll
        for (int u = 0; u < NT; ++u) {     // t1 (pre-relu), sigmoid(gate)
          if constexpr (LEAN) {
            if (ab_t[u]) {
              *reinterpret_cast<f4*>(ab_t[u] + (k.s_blk + 16 * (ks >> 1) + wave) * 256) = u1[u];
              *reinterpret_cast<f4*>(ab_t[u] + (k.s_blk + 16 * (ks >> 1) + 8 + wave) * 256) = gate[u];
            }
          } else { sv[ks][0][u] = u1[u]; sv[ks][1][u] = gate[u]; }
        }
      } else {
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) h[u][r] += u1[u][r] * gate[u][r];
#pragma unroll
        for (int u = 0; u < NT; ++u) {     // t2, h_{b+1}
          if constexpr (LEAN) {
            if (ab_t[u]) {
              *reinterpret_cast<f4*>(ab_t[u] + (k.s_blk + 16 * (ks >> 1) + 4 + wave) * 256) = u1[u];
              *reinterpret_cast<f4*>(ab_t[u] + (k.s_blk + 16 * (ks >> 1) + 12 + wave) * 256) = h[u];
            }
          } else { sv[ks][0][u] = u1[u]; sv[ks][1][u] = h[u]; }
        }
      }
      TSC(3 + 2 * ks);
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
        if (mt < kp.nft) {
          CoWf& w = (i & 1) ? wf1 : wf0;
          f4 acc[NT];
#pragma unroll
          for (int u = 0; u < NT; ++u) acc[u] = w.bias;
          co_gemm_h<NT, KSH>(w.a, hb, acc);
          if (i < 2) co_load_wf(img, kp, mt + 2 * CO_WAVES, id, w);     // two tiles ahead
          const int dd = mt / PT, pt = mt - dd * PT;
#pragma unroll
          for (int u = 0; u < NT; ++u) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              pst[(16 * u + id.j) * k.DSTR + dd * k.PSW + 16 * pt + 4 * r + id.g] = acc[u][r];
            if constexpr (LEAN) {
              if (ab_t[u]) *reinterpret_cast<f4*>(ab_t[u] + (k.s_par + mt) * 256) = acc[u];
            } else pv[i][u] = acc[u];
          }
        }
      }
    }
    TSC(21);
    // LULinear's factors and the next transform's first weights: requested before the spline (and before the stash
    // burst enters the in-order memory queue), landed after it
    const f4 au = *(reinterpret_cast<const f4*>(img + kp.u) + id.lane);
    const f4 al = *(reinterpret_cast<const f4*>(img + kp.l) + id.lane);
    const f4 blu = co_load_bias(img + kp.blu, 0, id.g);
    const float ld_lu = img[kp.ld];
    {
      const int tn = t + 1 < k.T ? t + 1 : (MC ? 0 : t);   // (last transform: re-requests itself, harmless; MC: the next tick's first)
      const float* imgn = cimg + (long long)tn * k.img_floats;
      const CoKP& kn = k.p[tn & 1];
      co_load_w0(imgn, k, kn, wave, id, w0);
      co_load_set<0>(imgn, k, kn, wave, id, S[0]);
      if (!LEAN) co_load_set<1>(imgn, k, kn, wave, id, S[1 % NS]);
    }
    if (!LEAN && ast) {      // the stash burst (plain stores: the backward workgroup of the same index runs on the same XCD)
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        float* ab = abase[u] ? abase[u] + t * astride : nullptr;
        if (ab) {
          *reinterpret_cast<f4*>(ab + wave * 256) = sv0[u];
#pragma unroll
          for (int ks = 0; ks < 2 * NSF_MAX_NB; ++ks)
            if (ks < nstages) {
              const int b = ks >> 1, o = (ks & 1) ? 4 : 0;      // even stage: t1 | gate, odd stage: t2 | h_{b+1}
              *reinterpret_cast<f4*>(ab + (k.s_blk + 16 * b + o + wave) * 256) = sv[LEAN ? 0 : ks][0][u];
              *reinterpret_cast<f4*>(ab + (k.s_blk + 16 * b + o + 8 + wave) * 256) = sv[LEAN ? 0 : ks][1][u];
            }
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (wave + CO_WAVES * i < kp.nft)
              *reinterpret_cast<f4*>(ab + (k.s_par + wave + CO_WAVES * i) * 256) = pv[LEAN ? 0 : i][u];
        }
      }
    }
    __syncthreads();
    TSC(22);
    // ---- spline: task (row 16 u + j, dim 2 wave + slot) on the lane pair (lane, lane ^ 32)
    {
      const int slot = id.g & 1, part = id.g >> 1;
      const int dd_raw = 2 * wave + slot;
      const bool live = dd_raw < kp.d_tr;
      const int dd = live ? dd_raw : 0;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int r = 16 * u + id.j;
        const int zi = r * ZS + 2 * dd + par;
        float y, ld;
        rq_spline_pair<K, false>(pst + r * k.DSTR + dd * k.PSW, zs[zi], k, part, y, ld);
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
      const float lp = -0.5f * ss + ld + ld_const - k.log_z;
      if constexpr (MC) {
        // this thread holds chain `row`'s log-density: advance its state machine and publish the next evaluation point
        mc.logp_buf[row] = lp;
        slice_tick_one((int)row, D, mc.num_samples, mc.tuning, mc.max_width, mc.logp_buf, mc.lad_next, nullptr, mc.x,
                       mc.next_param, mc.width, mc.order, mc.istate, mc.fstate, mc.samples, mc.done_count, mc.seed,
                       mc.tick0 + (unsigned long long)tick, mc.kind, mc.p0, mc.p1, mc.theta_next, mc.lad_next);
        for (int d = 0; d < D; ++d) mc_theta[tid * 16 + d] = mc.theta_next[row * D + d];
      } else {
        logp[row] = lp;
      }
    }
  }
  if constexpr (MC) __syncthreads();
  }   // ticks
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

// Weight-gradient tiles of one 16-wide slice of OUTPUT features, all from transposed LDS tiles (one ds_read_b128 per
// operand, row tile and n-tile):  acc[nt][r] of lane (g, j) = d W[out0 + j][16 nt + 4 g + r]
//   = sum over the workgroup's rows of Gt[out0 + j][row] * At[in_off + 16 nt + 4 g + r][row].
// The INPUT index runs along the registers, so a lane owns four consecutive entries of one row of the weight gradient.
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

// partial-gradient write-out of one tile of co_dw (slab layout: CoShape::dw_tb): ONE aligned 16-byte store per lane,
// 1 KiB contiguous per wave.  `tile` = the linear's tile base + mt * nnt + nt; lanes whose four inputs all lie behind
// the bias column (16 nt + 4 g > in_dim), or whose output does not exist, store nothing (the reduction skips them).
__device__ __forceinline__ void co_write_tile(float* __restrict__ part, int tile, bool out_ok, int nt, int in_dim,
                                              const LaneId& id, const f4& acc) {
  if (part && out_ok && 16 * nt + 4 * id.g <= in_dim)      // (part == null: SBI_AMD_ABLATE bit 32768, timing only)
    *reinterpret_cast<f4*>(part + tile * 256 + 4 * id.lane) = acc;
}

template <int K, int KSH, int NT>
__global__ void __launch_bounds__(64 * CO_WAVES, 1)
nsf_coop_bwd_kernel(const CoK k, const float* __restrict__ cimg, const float* __restrict__ zstats,
                    const float* __restrict__ x, long long n, long long x_rows, const float* __restrict__ row_w,
                    const float uni_w, const float* __restrict__ z_last, const float* __restrict__ zst,
                    const float* __restrict__ ast, float* __restrict__ partial, float* __restrict__ grad_theta,
                    float* __restrict__ grad_x, long long* __restrict__ dbg) {
  // debug timeline (SBI_AMD_TIMELINE): cycle stamps of workgroup 0's waves while they walk transform T - 2
#ifdef NSF_DEBUG
#define TSB(i) do { if (dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && t == k.T - 2) \
    dbg[(threadIdx.x >> 6) * 64 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TSB(i) do { } while (0)
#endif
  constexpr int PT = (3 * K - 1 + 15) / 16;
  constexpr int R = 16 * NT;
  constexpr int NZ = (R * 16 + 64 * CO_WAVES - 1) / (64 * CO_WAVES);   // state values per thread (theta-dim <= 16)
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
  float* GT1 = GT0 + 64 * RS;
  float* AT0 = lds + k.o_at;
  float* AT1 = AT0 + 65 * RS;
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
  const bool hb64 = H == 64;
  const int ones_h = hb64 ? -1 : H;
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bool want_gx = grad_x != nullptr;
  const int ntc = k.ntc, nnh = k.nnh;
  const int blk_tiles = 4 * ntc + 8 * nnh;          // slab tiles of one residual block: d Wc | d W1 | d W2
  const bool out_ok = 16 * wave + id.j < H;         // this lane's output feature of a hidden-width layer exists
  // stash addresses of this wave's fragments (clamped tiles: wave-tiles past the last row were never written by the
  // forward pass): slot s of transform t, row tile u is abase[u] + t * astride + s * 256
  const float* abase[NT];
  const long long astride = nt16 * k.slots * 256;
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const long long t16 = (row0 >> 4) + u < nt16 ? (row0 >> 4) + u : nt16 - 1;
    abase[u] = ast + t16 * k.slots * 256 + 4 * id.lane;
  }
  // What a transform needs FIRST is requested a transform ahead (for t = T - 1: before the prologue touches LDS):
  // its input state rows, its spline-parameter tiles, LULinear's factors.
  float zin[NZ];
  f4 ptile[4][NT];
  f4 a_lt, a_ut, a_u;
  auto request_entry = [&](int t) {
    const CoKP& kq = k.p[t & 1];
    const float* img = cimg + (long long)t * k.img_floats;
#pragma unroll
    for (int q = 0; q < NZ; ++q) {
      const int i = tid + 64 * CO_WAVES * q;
      const int r = i / D, d = i - r * D;
      const long long row = row0 + r;
      zin[q] = (i < R * D && row < n) ? zst[((long long)t * n + row) * D + d] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mt = wave + CO_WAVES * i < kq.nft ? wave + CO_WAVES * i : kq.nft - 1;    // clamped: straight-line
#pragma unroll
      for (int u = 0; u < NT; ++u)
        ptile[i][u] = *reinterpret_cast<const f4*>(abase[u] + t * astride + (k.s_par + mt) * 256);
    }
    a_lt = *(reinterpret_cast<const f4*>(img + kq.lt) + id.lane);
    a_ut = *(reinterpret_cast<const f4*>(img + kq.ut) + id.lane);
    a_u = *(reinterpret_cast<const f4*>(img + kq.u) + id.lane);
  };
  request_entry(k.T - 1);

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
  // d loss / d context (a trainable embedding net in front of the flow): waves 0 and 1 own context features
  // [16 wave, 16 wave + 16) and accumulate W0[:, ctx]^T g_h0 + sum_b Wc_b^T g_c over all transforms
  f4 gxacc[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) gxacc[u] = zero4;
  __syncthreads();

  for (int t = k.T - 1; t >= 0; --t) {
    const int par = t & 1;
    const CoKP& kp = k.p[par];
    const float* img = cimg + (long long)t * k.img_floats;
    float* part = NSF_DBG_ABL(k.ablate, 32768) ? nullptr : partial + ((long long)t * gridDim.x + blockIdx.x) * k.PLP;
    const int tb_blk0 = 4 * kp.nnt0;                 // slab tile bases: d W0 | blocks | d Wf | LULinear tail
    const int tb_wf = tb_blk0 + NB * blk_tiles;
    TSB(0);
    // ---- requests whose results are needed after the spline: h_last, Wf^T, the last block's stash, the first two
    //      transposed hidden matrices
    struct BSt { f4 t1[NT], t2[NT], sg[NT], hb[NT]; };   // a block's stash: t1 (pre-relu), t2, sigmoid(gate), input
    f4 hl[NT], wft[8][PT], T[3][4];
    BSt B[2];
    const float* at[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      at[u] = abase[u] + t * astride + wave * 256;
      hl[u] = *reinterpret_cast<const f4*>(at[u] + (k.s_blk + 16 * (NB - 1) + 12) * 256);
    }
    {
      const f4* ap = reinterpret_cast<const f4*>(img + kp.wft + wave * kp.d_tr * PT * 256) + id.lane;
#pragma unroll
      for (int dd = 0; dd < 8; ++dd)
        if (dd < kp.d_tr) {
#pragma unroll
          for (int q = 0; q < PT; ++q) wft[dd][q] = ap[(dd * PT + q) * 64];
        }
    }
    auto request_block = [&](int b, BSt& st) {
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const float* a = at[u] + (k.s_blk + 16 * b) * 256;
        st.t1[u] = *reinterpret_cast<const f4*>(a);
        st.t2[u] = *reinterpret_cast<const f4*>(a + 4 * 256);
        st.sg[u] = *reinterpret_cast<const f4*>(a + 8 * 256);
        st.hb[u] = *reinterpret_cast<const f4*>(b == 0 ? at[u] : a - 4 * 256);     // h_0 (slot 0) or h_b of block b - 1
      }
    };
    auto load_tset = [&](auto kc, f4 (&a)[4]) {      // transposed hidden matrix of backward stage ks (clamped)
      constexpr int ks = decltype(kc)::value;
      const int b = NB - 1 - (ks >> 1) > 0 ? NB - 1 - (ks >> 1) : 0;
      co_load_a<4>(img + ((ks & 1) ? kp.w1t0 : kp.w2t0) + b * k.sT + wave * 1024, id.lane, a);
    };
    request_block(NB - 1, B[0]);
    load_tset(CoIdx<0>{}, T[0]);
    load_tset(CoIdx<1>{}, T[1]);
    TSB(1);
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
    for (int i = tid; i < (k.ct_rows - kp.d_id) * R; i += 64 * CO_WAVES) {
      const int kk = kp.d_id + i / R, r = i % R;
      CT[kk * RS + r] = kk < kp.in0 ? ctx[(kk - kp.d_id) * R + r] : (kk == kp.in0 ? 1.f : 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mt = wave + CO_WAVES * i;
      if (mt < kp.nft) {
        const int dd = mt / PT, pt = mt - dd * PT;
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pst[(16 * u + id.j) * k.DSTR + dd * k.PSW + 16 * pt + 4 * r + id.g] = ptile[i][u][r];
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
    TSB(2);
    __syncthreads();
    TSB(3);
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
    TSB(4);
    __syncthreads();
    TSB(5);
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
        if (dd < kp.d_tr) {
#pragma unroll
          for (int q = 0; q < PT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int u = 0; u < NT; ++u) {
                const float bv = pst[(16 * u + id.j) * k.DSTR + dd * k.PSW + 16 * q + 4 * r + id.g];
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
    TSB(6);
    // the transform below: its state rows, parameter tiles and LU factors are requested BEFORE this transform's
    // partial-slab stores enter the (in-order) memory queue; they land under the block phase
    if (t > 0) request_entry(t - 1);
    __syncthreads();
    TSB(7);
    // ---- d Wf (parameter tiles wave, wave + 4, ...), LULinear parameter gradients
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mt = wave + CO_WAVES * i;
      if (mt < kp.nft) {
        const int dd = mt / PT, pt = mt - dd * PT;
        f4 acc[4], accb = zero4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = zero4;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          float bv[4];    // B side = g_p: parameter 16 pt + j of rows 16 u + 4 g + s
#pragma unroll
          for (int s = 0; s < 4; ++s) bv[s] = pst[(16 * u + 4 * id.g + s) * k.DSTR + dd * k.PSW + 16 * pt + id.j];
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
        const bool p_ok = 16 * pt + id.j < k.P;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) co_write_tile(part, tb_wf + mt * nnh + nt, p_ok, nt, H, id, acc[nt]);
        if (hb64) co_write_tile(part, tb_wf + mt * nnh + 4, p_ok, 4, H, id, accb);
      }
    }
    if (part) {
      const int ntri = D * (D - 1) / 2;
      float* plow = part + kp.dw_tail;
      float* pup = plow + ntri;
      float* pdiag = pup + ntri;
      float* pbias = pdiag + D;
      // (on the waves with the fewest final-layer tiles: 10 tiles over four waves is 3, 3, 2, 2)
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
    TSB(8);
    // ---- residual blocks, last -> first, unrolled at compile time over the execution ordinal i (block b = NB-1-i):
    //      stage 2 i is W2_b^T, stage 2 i + 1 is W1_b^T; transposed matrices two stages ahead, the stash one block ahead
    f4 a0t[4];     // W0^T (identity columns), wave 0
    auto block = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int b = NB - 1 - i;
      const int tb_c = tb_blk0 + b * blk_tiles, tb_1 = tb_c + 4 * ntc, tb_2 = tb_1 + 4 * nnh;
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
      load_tset(CoIdx<2 * i + 2>{}, T[(2 * i + 2) % 3]);
      request_block(b > 0 ? b - 1 : 0, B[(i + 1) & 1]);
      if (b == 0) co_load_a<4>(img + kp.w0t, id.lane, a0t);
      co_store_T<NT>(GT0, RS, 16 * wave, id, ga, false, -1);
      co_store_T<NT>(GT1, RS, 16 * wave, id, gc, false, -1);
      co_store_T<NT>(AT1, RS, 16 * wave, id, cur.t1, true, ones_h);
      f4 bg[NT][CO_WAVES], gr[NT];
      if (want_gx) {      // (uniform) context gradient of this block's gate: Wc^T g_c
        co_gather<NT>(ex, buf, wave, id.lane, gc, bg);
        if (wave * 16 < C) {
          f4 act[4];
          co_load_a<4>(img + kp.wct0 + b * k.sC + wave * 1024, id.lane, act);
          co_gemm_h<NT, KSH>(act, bg, gxacc);
        }
      }
      TSB(10 + 8 * i);
      co_gather<NT>(ex, buf, wave, id.lane, ga, bg);
      TSB(11 + 8 * i);
#pragma unroll
      for (int u = 0; u < NT; ++u) gr[u] = zero4;
      co_gemm_h<NT, KSH>(T[(2 * i) % 3], bg, gr);
      TSB(12 + 8 * i);
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) ga[u][r] = cur.t1[u][r] > 0.f ? gr[u][r] : 0.f;     // d t1
      load_tset(CoIdx<2 * i + 3>{}, T[(2 * i + 3) % 3]);
      {
        f4 acc[4], accb;
        co_dw<NT, 4>(GT0, AT1, RS, 16 * wave, 0, 4, id, acc, hb64 ? &accb : nullptr);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) co_write_tile(part, tb_2 + wave * nnh + nt, out_ok, nt, H, id, acc[nt]);
        if (hb64) co_write_tile(part, tb_2 + wave * nnh + 4, out_ok, 4, H, id, accb);
        f4 accc[3];       // x-dim <= 32 plus the bias column: up to three n-tiles
        co_dw<NT, 3>(GT1, CT, RS, 16 * wave, kp.d_id, ntc, id, accc, nullptr);
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
          if (nt < ntc) co_write_tile(part, tb_c + wave * ntc + nt, out_ok, nt, C, id, accc[nt]);
      }
      wave_lds_fence();
      TSB(13 + 8 * i);
      co_store_T<NT>(GT0, RS, 16 * wave, id, ga, false, -1);
      co_store_T<NT>(AT0, RS, 16 * wave, id, cur.hb, true, ones_h);
      co_gather<NT>(ex, buf, wave, id.lane, ga, bg);
      TSB(14 + 8 * i);
#pragma unroll
      for (int u = 0; u < NT; ++u) gr[u] = zero4;
      co_gemm_h<NT, KSH>(T[(2 * i + 1) % 3], bg, gr);
      TSB(15 + 8 * i);
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) gh[u][r] += cur.hb[u][r] > 0.f ? gr[u][r] : 0.f;
      {
        f4 acc[4], accb;
        co_dw<NT, 4>(GT0, AT0, RS, 16 * wave, 0, 4, id, acc, hb64 ? &accb : nullptr);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) co_write_tile(part, tb_1 + wave * nnh + nt, out_ok, nt, H, id, acc[nt]);
        if (hb64) co_write_tile(part, tb_1 + wave * nnh + 4, out_ok, 4, H, id, accb);
      }
      wave_lds_fence();
      TSB(16 + 8 * i);
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
      TSB(50);
      if (wave == 0) {      // identity features receive W0[:, :d_id]^T g_h0
        f4 gin[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) gin[u] = zero4;
        co_gemm_h<NT, KSH>(a0t, bg, gin);
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kk = 4 * r + id.g;
            if (kk < kp.d_id) gys[(16 * u + id.j) * ZS + 2 * kk + (1 - par)] += gin[u][r];
          }
      }
      if (want_gx && wave * 16 < C) {
        f4 act[4];
        co_load_a<4>(img + kp.w0ct + wave * 1024, id.lane, act);
        co_gemm_h<NT, KSH>(act, bg, gxacc);
      }
      f4 acc[3];
      co_dw<NT, 3>(GT0, CT, RS, 16 * wave, 0, kp.nnt0, id, acc, nullptr);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
        if (nt < kp.nnt0) co_write_tile(part, wave * kp.nnt0 + nt, out_ok, nt, kp.in0, id, acc[nt]);
    }
    TSB(51);
    __syncthreads();
    TSB(52);
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
#undef TSB
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
  const McArgs* mc;      // non-null: the persistent slice sampler (theta = the chains' next evaluation points)
};
struct CoBwdArgs {
  const float *cimg, *zstats, *x;
  long long n, x_rows;
  const float* row_w;
  float uni_w;
  const float *z_last, *zst, *ast;
  float *partial, *grad_theta, *grad_x;
  long long* dbg;
};

template <int K, int KSH, int NT, bool LEAN, bool MC = false>
static int co_launch_fwd(const CoK& k, const CoopPlan& cp, const CoFwdArgs& a, hipStream_t st) {
  auto kern = nsf_coop_fwd_kernel<K, KSH, NT, LEAN, MC>;
  const int lds_bytes = 4 * cp.lds_floats;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(cp.grid), dim3(64 * CO_WAVES), (size_t)lds_bytes, st, k, a.cimg, a.zstats, a.theta, a.x,
                     a.n, a.x_rows, a.logp, a.noise, a.zst, a.ast, a.dbg, (MC && a.mc) ? *a.mc : McArgs{});
  return (int)hipGetLastError();
}
template <int K, int KSH, int NT>
static int co_launch_bwd(const CoK& k, const CoopPlan& cp, const CoBwdArgs& a, hipStream_t st) {
  auto kern = nsf_coop_bwd_kernel<K, KSH, NT>;
  const int lds_bytes = 4 * cp.lds_floats;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(cp.grid), dim3(64 * CO_WAVES), (size_t)lds_bytes, st, k, a.cimg, a.zstats, a.x, a.n,
                     a.x_rows, a.row_w, a.uni_w, a.z_last, a.zst, a.ast, a.partial, a.grad_theta, a.grad_x, a.dbg);
  return (int)hipGetLastError();
}

