// nsf_device.h -- device building blocks shared by the NSF kernels (gfx950).
//
// Execution model (see DESIGN.md "kernel anatomy"):
//   * one wavefront (64 lanes) owns 16 batch rows; lane = (j = lane&15 : row,
//     g = lane>>4 : k-slot).  All dense layers of the ResidualNet conditioner
//     run on v_mfma_f32_16x16x4_f32 with   M = output feature, N = batch row,
//     K = input feature, so the D fragment of one layer (lane (g,j), reg r of
//     tile mt holds feature 16*mt + 4*r + g of row j) IS the B fragment of the
//     next layer's K-step s = 4*mt + r: activations never leave registers.
//   * weights (A operand) are read from the LDS image staged once per coupling
//     layer per workgroup: lane (i = lane&15, g) reads W[feat(i)][4*s + g],
//     feat(i) = 16*mt + 4*(i&3) + (i>>2)   (the within-tile transpose that makes
//     the D->B hand-off line up); row stride 2*odd => conflict-free ds_read_b32.
#pragma once
#include <hip/hip_runtime.h>
#include "nsf_plan.h"

typedef float f4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Orders this wave's LDS traffic (DS ops of one wave execute in issue order; the
// fence only stops the compiler from moving them across phase boundaries).
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct LaneId {
  int lane, j, g, iperm;
};
__device__ __forceinline__ LaneId make_lane() {
  LaneId L;
  L.lane = threadIdx.x & 63;
  L.j = L.lane & 15;
  L.g = L.lane >> 4;
  L.iperm = 4 * (L.j & 3) + (L.j >> 2);
  return L;
}

// ---- weight packing: flat natural layout -> per-layer MFMA A-operand image ----
// Run by nsf_pack_kernel (one workgroup per transform) whenever the parameters
// change; the compute kernels then stage a layer with a plain float4 copy.
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__device__ __forceinline__ void pack_linear(float* __restrict__ img, const float* __restrict__ gl,
                                            const LinDesc& L, int bias_pad, int bias_group, int bias_group_pad,
                                            int tid, int nthreads) {
  const int total = L.rows * L.ldk;
  for (int idx = tid; idx < total; idx += nthreads) {
    int r = idx / L.ldk;
    int c = idx - r * L.ldk;
    float v = 0.f;
    if (r < L.out && c < L.in) v = gl[L.g_w + r * L.in + c];
    img[L.l_w + idx] = v;
  }
  // bias: groups of `bias_group` real entries padded to `bias_group_pad`
  for (int idx = tid; idx < bias_pad; idx += nthreads) {
    int grp = idx / bias_group_pad;
    int p = idx - grp * bias_group_pad;
    int src = grp * bias_group + p;
    float v = 0.f;
    if (p < bias_group && src < L.out) v = gl[L.g_b + src];
    img[L.l_b + idx] = v;
  }
}

__device__ __forceinline__ void pack_lu(float* __restrict__ img, const float* __restrict__ gl,
                                        const ShapeDesc& S, int D, float eps, int tid, int nthreads) {
  // LULinear._create_lower_upper: np.tril_indices(D,-1) / np.triu_indices(D,1) order.
  const int ntri = D * (D - 1) / 2;
  const float* lower = gl + S.g_lu;
  const float* upper = lower + ntri;
  const float* udiag = upper + ntri;
  const float* bias = udiag + D;
  // dense U, L stored [LUS][LUS] (LUS = 16 for D <= 16: zero padded so the mat-vec helpers need
  // no bounds checks; LUS = D otherwise)
  const int LUS = D <= 16 ? 16 : D;
  for (int idx = tid; idx < LUS * LUS; idx += nthreads) {
    int i = idx / LUS, k = idx - i * LUS;
    float u = 0.f, l = 0.f;
    if (i < D && k < D) {
      if (k > i) u = upper[i * D - i * (i + 1) / 2 + (k - i - 1)];
      else if (k == i) { u = softplus_f(udiag[i]) + eps; l = 1.f; }
      else l = lower[i * (i - 1) / 2 + k];
    }
    img[S.l_U + idx] = u;
    img[S.l_L + idx] = l;
  }
  for (int idx = tid; idx < D; idx += nthreads) img[S.l_lub + idx] = bias[idx];
  if (tid == 0) {   // logabsdet of the LULinear = sum_i log(softplus(u_i) + eps)
    float a = 0.f;
    for (int i = 0; i < D; ++i) a += logf(softplus_f(udiag[i]) + eps);
    img[S.l_lub + D] = a;
  }
}

// Explicit inverses for the sampling direction (D <= 16), by ONE workgroup: the 16 x 16 factors go through
// LDS once (softplus on the diagonal evaluated D times, not D^3), then one thread per column substitutes in
// double -- the inverse pass becomes two dense mat-vecs instead of a serial per-row triangular solve.
__device__ __forceinline__ void pack_lu_inverse(float* __restrict__ img, const float* __restrict__ gl,
                                                const ShapeDesc& S, int D, float eps, int ltid, int lthreads) {
  __shared__ float sU[256], sL[256];
  const int ntri = D * (D - 1) / 2;
  const float* lower = gl + S.g_lu;
  const float* upper = lower + ntri;
  const float* udiag = upper + ntri;
  for (int idx = ltid; idx < 256; idx += lthreads) {
    const int i = idx >> 4, k = idx & 15;
    float u = 0.f, l = 0.f;
    if (i < D && k < D) {
      if (k > i) u = upper[i * D - i * (i + 1) / 2 + (k - i - 1)];
      else if (k == i) { u = softplus_f(udiag[i]) + eps; l = 1.f; }
      else l = lower[i * (i - 1) / 2 + k];
    }
    sU[idx] = u;
    sL[idx] = l;
  }
  __syncthreads();
  __shared__ double xl[16][17], xu[16][17];   // [row][column]: dynamically indexed, so LDS rather than scratch
  if (ltid < 16) {
    const int c = ltid;
    for (int i = 0; i < 16; ++i) { xl[i][c] = 0.0; xu[i][c] = 0.0; }
    if (c < D) {
      xl[c][c] = 1.0;
      for (int i = c + 1; i < D; ++i) {
        double a = 0.0;
        for (int k = c; k < i; ++k) a -= (double)sL[i * 16 + k] * xl[k][c];
        xl[i][c] = a;
      }
      xu[c][c] = 1.0 / (double)sU[c * 16 + c];
      for (int i = c - 1; i >= 0; --i) {
        double a = 0.0;
        for (int k = i + 1; k <= c; ++k) a -= (double)sU[i * 16 + k] * xu[k][c];
        xu[i][c] = a / (double)sU[i * 16 + i];
      }
    }
    for (int i = 0; i < 16; ++i) {
      img[S.l_Li + i * 16 + c] = (float)xl[i][c];
      img[S.l_Ui + i * 16 + c] = (float)xu[i][c];
    }
  }
}

// (tid, nthreads) may span several workgroups: tid = blockIdx.y*blockDim.x + threadIdx.x
__device__ __forceinline__ void pack_layer(float* __restrict__ img, const float* __restrict__ gl,
                                           const NsfPlan& pl, const ShapeDesc& S, int tid, int nthreads) {
  const int hb = 16 * NSF_HT;
  pack_linear(img, gl, S.lin[0], hb, hb, hb, tid, nthreads);
  for (int b = 0; b < pl.NB; ++b) {
    pack_linear(img, gl, S.lin[1 + 3 * b], hb, hb, hb, tid, nthreads);
    pack_linear(img, gl, S.lin[2 + 3 * b], hb, hb, hb, tid, nthreads);
    pack_linear(img, gl, S.lin[3 + 3 * b], hb, hb, hb, tid, nthreads);
  }
  if (pl.ctx_mlp && pl.ctx_reps > 0) pack_linear(img, gl, S.lin[1], hb, hb, hb, tid, nthreads);
  pack_linear(img, gl, S.lin[S.fin], S.d_tr * 16 * pl.PT, pl.P, 16 * pl.PT, tid, nthreads);
  if (!pl.ctx_mlp) pack_lu(img, gl, S, pl.D, pl.lu_eps, tid, nthreads);
  else for (int idx = S.l_U + tid; idx < S.l_lub + pl.D + 1; idx += nthreads) img[idx] = 0.f;
  // zero the padding behind the LU bias -- but not the explicit inverses a different workgroup writes
  const int inv_lo = S.l_Ui >= 0 ? S.l_Ui : pl.img_floats, inv_hi = S.l_Ui >= 0 ? S.l_Li + 256 : pl.img_floats;
  for (int idx = S.l_lub + pl.D + 1 + tid; idx < pl.img_floats; idx += nthreads)
    if (idx < inv_lo || idx >= inv_hi) img[idx] = 0.f;
}

// stage one layer's image into LDS: coalesced 16-byte copies, all loads issued first
__device__ __forceinline__ void stage_layer(float* __restrict__ lds, const float* __restrict__ img,
                                            int img_floats, int tid, int nthreads) {
  const float4* __restrict__ src = reinterpret_cast<const float4*>(img);
  float4* dst = reinterpret_cast<float4*>(lds);
  const int n4 = img_floats >> 2;
  int idx = tid;
  for (; idx + 3 * nthreads < n4; idx += 4 * nthreads) {
    const float4 a = src[idx], b = src[idx + nthreads], c = src[idx + 2 * nthreads], d = src[idx + 3 * nthreads];
    dst[idx] = a;
    dst[idx + nthreads] = b;
    dst[idx + 2 * nthreads] = c;
    dst[idx + 3 * nthreads] = d;
  }
  for (; idx < n4; idx += nthreads) dst[idx] = src[idx];
}

// The same copy in two halves, for kernels that know the image size at compile time: the loads are issued EARLY (under
// the LULinear phase of the transform before) into NPRE float4 registers per thread, the LDS stores happen between the
// two barriers -- the L2 round trip no longer sits on the barrier-to-barrier critical path of every transform.
template <int NPRE>
__device__ __forceinline__ void stage_issue(const float* __restrict__ img, int img_floats, int tid, int nthreads,
                                            float4 (&pre)[NPRE]) {
  const float4* __restrict__ src = reinterpret_cast<const float4*>(img);
  const int n4 = img_floats >> 2;
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    const int idx = tid + i * nthreads;
    pre[i] = idx < n4 ? src[idx] : float4{0.f, 0.f, 0.f, 0.f};
  }
}
template <int NPRE>
__device__ __forceinline__ void stage_commit(float* __restrict__ lds, int img_floats, int tid, int nthreads,
                                             const float4 (&pre)[NPRE]) {
  float4* dst = reinterpret_cast<float4*>(lds);
  const int n4 = img_floats >> 2;
#pragma unroll
  for (int i = 0; i < NPRE; ++i) {
    const int idx = tid + i * nthreads;
    if (idx < n4) dst[idx] = pre[i];
  }
}

// ---- MFMA GEMM pieces ------------------------------------------------------
__device__ __forceinline__ void acc_init_bias(const float* __restrict__ lds, const LinDesc& L, const LaneId& id,
                                              f4 (&acc)[NSF_HT]) {
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[mt][r] = lds[L.l_b + 16 * mt + 4 * r + id.g];
  }
}

__device__ __forceinline__ void a_row_offsets(const LinDesc& L, const LaneId& id, int (&ro)[NSF_HT]) {
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt) {
    int f = 16 * mt + id.iperm;
    ro[mt] = L.l_w + (f < L.out ? f : L.out) * L.ldk + id.g;
  }
}

// acc += W * B, B operand read from a per-wave LDS row buffer (conditioner input).
// L.ksteps is a multiple of 4 (zero padded on both operands).
__device__ __forceinline__ void gemm_blds(const float* __restrict__ lds, const LinDesc& L, const LaneId& id,
                                          const float* __restrict__ brow /* &buf[j*stride + coff + g] */,
                                          f4 (&acc)[NSF_HT]) {
  int ro[NSF_HT];
  a_row_offsets(L, id, ro);
  for (int s4 = 0; s4 < L.ksteps; s4 += 4) {
    float bv[4], av[4][NSF_HT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bv[u] = brow[4 * (s4 + u)];
#pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt) av[u][mt] = lds[ro[mt] + 4 * (s4 + u)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt) acc[mt] = MFMA16(av[u][mt], bv[u], acc[mt]);
  }
}

// the same with an explicit K-step count (broadcast-x kernels: the initial layer over the identity features only;
// the B row is zero from column d_id on, so the context columns of W meet zeros)
__device__ __forceinline__ void gemm_blds_steps(const float* __restrict__ lds, const LinDesc& L, const LaneId& id,
                                                const float* __restrict__ brow, int nsteps, f4 (&acc)[NSF_HT]) {
  int ro[NSF_HT];
  a_row_offsets(L, id, ro);
  for (int s = 0; s < nsteps; ++s) {
    const float bv = brow[4 * s];
    float av[NSF_HT];
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt) av[mt] = lds[ro[mt] + 4 * s];
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt) acc[mt] = MFMA16(av[mt], bv, acc[mt]);
  }
}

// acc += W * B, B operand = the previous layer's D fragments (registers); KSH K-steps,
// fully unrolled so the LDS reads of step s+1 are in flight under the MFMAs of step s.
template <int KSH>
__device__ __forceinline__ void gemm_breg(const float* __restrict__ lds, const LinDesc& L, const LaneId& id,
                                          const f4 (&b)[NSF_HT], f4 (&acc)[NSF_HT]) {
  int ro[NSF_HT];
  a_row_offsets(L, id, ro);
  float a_cur[NSF_HT], a_nxt[NSF_HT];
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt) a_cur[mt] = lds[ro[mt]];
#pragma unroll
  for (int s = 0; s < KSH; ++s) {
    if (s + 1 < KSH) {
#pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt) a_nxt[mt] = lds[ro[mt] + 4 * (s + 1)];
    }
    const float bv = b[s >> 2][s & 3];
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt) acc[mt] = MFMA16(a_cur[mt], bv, acc[mt]);
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt) a_cur[mt] = a_nxt[mt];
  }
}

// ---- VALU math: the spline is issue-bound, so transcendental-heavy pieces use the
// hardware v_exp_f32 / v_rcp_f32 (1 ulp) with the argument product carried in two
// floats -- same accuracy class as libm's expf / a correctly rounded divide, a
// fraction of the instructions.
__device__ __forceinline__ float rcp_f(float x) { return __builtin_amdgcn_rcpf(x); }
// v_rcp_f32 + one Newton step (two FMAs): within half an ulp of the correctly rounded quotient the eager reference
// computes.  Used where a reciprocal feeds a CANCELLING expression -- the bin's slope h / w inside the inverse's
// quadratic, the softmax normaliser that places every knot -- so that a saturated bin (slope 1e3) does not amplify the
// hardware reciprocal's last ulp into several spacings of the spline input (tests/test_spline_adversarial_gpu.py).
__device__ __forceinline__ float rcp_nr(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(r, fmaf(-x, r, 1.f), r);
}
// exp(x) = 2^t * exp(x - t ln 2) with t = fl(x log2 e): the residual x - t ln 2 (|.| <= |t| 2^-24) is what the rounding
// of t loses; ln 2 = hi + lo keeps it to a few 1e-9 relative.  exp2, three FMAs and a multiply.
__device__ __forceinline__ float exp_f(float x) {
  const float t = x * 1.4426950408889634f;
  float r = fmaf(-t, 0.693147182464599609375f, x);      // ln 2 = hi + lo
  r = fmaf(-t, -1.90465429995776804525e-9f, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r, e);
}
__device__ __forceinline__ float sigmoid_f(float x) { return rcp_f(1.0f + exp_f(-x)); }
// the residual blocks' GLU gate: 32 of these per lane and transform.  Without the two-float argument product the
// exponent t = -x log2(e) is off by <= half an ulp of |t| (5e-7 at |x| = 10), i.e. sigma is off by
// <= sigma (1 - sigma) 3.3e-7 <= 8e-8 absolute -- one ulp of the value it is multiplied into; 4 instructions instead of 8.
#ifndef NSF_GATE_FAST
#define NSF_GATE_FAST 1      // A/B: -DNSF_GATE_FAST=0 in SBI_AMD_EXTRA_HIPCC_FLAGS
#endif
__device__ __forceinline__ float sigmoid_gate(float x) {
  return NSF_GATE_FAST ? rcp_f(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)) : sigmoid_f(x);
}
// natural log on the hardware v_log_f32 (1 ulp log2): branch-free, ~3 instructions
__device__ __forceinline__ float log_f(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
// softplus(x) = max(x,0) + log1p(exp(-|x|)), branch-free; log1p via the (1+t) compensation trick
__device__ __forceinline__ float softplus_bf(float x) {
  const float t = exp_f(-fabsf(x));
  const float u = 1.0f + t;
  const float l1p = log_f(u) - ((u - 1.0f) - t) * rcp_f(u);
  return fmaxf(x, 0.f) + l1p;
}

// ResidualNet hidden stack (nflows nn/nets/resnet.py, configuration flow.py:411-419):
// h = W0 [z_id; c] + b0;  per block: t = W2 relu(W1 relu(h) + b1) + b2; h += t * sigmoid(Wc c + bc)
// activation stash (training): one slot = this wave's 16x64 D-fragment array stored as [mt][lane][r]
// (one 16-byte word per lane and m-tile: a wave store / load moves 1 KB contiguous; the base pointer handed to
// ast_store / ast_load already includes 4 * lane); slots: 0 h_0 | per block b: 1+4b t1 (pre-relu),
// 2+4b t2, 3+4b sigmoid(gate), 4+4b h_{b+1}.  The backward kernel reloads them instead of
// recomputing the conditioner (trading ~0.75 GB/step of HBM traffic for 480 MFMAs per 16 rows).
#define NSF_AST_SLOTS(NB) (1 + 4 * (NB))   // residual-net conditioner with NB blocks (ctx_mlp: nsf_ast_slots(pl), nsf_plan.h)
// hidden <= 52 (KSH == 13, sbi's default 50): of the fourth m-tile only register 0 is real (features 48 + g; features
// 52 ... 63 are padding: zero, or sigmoid(0) for the gate, and nothing downstream depends on them) -- it is stored as
// ONE float per lane (256 B per wave) behind three full tiles: a slot is 832 floats instead of 1 024, -19 % of the
// stash's write (forward) and read (backward) traffic.  The tile stride (slots x 1 024 floats) is unchanged: the saved
// bytes are gaps at the end of every tile's block.
template <int KSH>
__device__ __forceinline__ constexpr int ast_slot_floats() { return KSH == 13 ? 3 * 256 + 64 : 4 * 256; }
template <int KSH>
__device__ __forceinline__ void ast_store(float* __restrict__ ast, int slot, const f4 (&v)[NSF_HT]) {
  float* base = ast + slot * ast_slot_floats<KSH>();
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt) {
    // streaming (written once by the forward, read once by the backward): keep it out of L2, which the
    // packed weight image lives in (measured -5 % step time)
    if (KSH == 13 && mt == NSF_HT - 1)
      __builtin_nontemporal_store(v[mt][0], base + mt * 256 - 3 * (int)(threadIdx.x & 63));   // (ast includes 4 * lane)
    else
      __builtin_nontemporal_store(v[mt], reinterpret_cast<f4*>(base + mt * 256));
  }
}
template <int KSH>
__device__ __forceinline__ void ast_load(const float* __restrict__ ast, int slot, f4 (&v)[NSF_HT]) {
  const float* base = ast + slot * ast_slot_floats<KSH>();
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt) {
    if (KSH == 13 && mt == NSF_HT - 1) {
      const float r0 = __builtin_nontemporal_load(base + mt * 256 - 3 * (int)(threadIdx.x & 63));
      v[mt] = f4{r0, 0.f, 0.f, 0.f};
    } else {
      v[mt] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(base + mt * 256));
    }
  }
}

// BX (one condition row for the whole launch: every sampler / potential call): everything that depends on the context
// alone is row-invariant -- W0[:, d_id:] c + b0 and, per block, sigmoid(Wc c + bc) -- and was put into the LDS table
// `bx` ([0, 64): initial-layer offset, [64 (1 + b), 64 (2 + b)): gate of block b; indexed by feature) once per
// transform and workgroup (bx_fold_context): the initial layer runs over the identity features only, the gates
// cost an LDS read instead of 16 MFMAs + 16 sigmoids per lane and block.
template <int KSH, bool BX = false>
__device__ __forceinline__ void conditioner_hidden(const float* __restrict__ lds, const NsfPlan& pl,
                                                   const ShapeDesc& S, const LaneId& id,
                                                   const float* __restrict__ cin_row, f4 (&h)[NSF_HT],
                                                   float* __restrict__ ast = nullptr,
                                                   const float* __restrict__ bx = nullptr) {
  if (BX) {
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[mt][r] = bx[16 * mt + 4 * r + id.g];
    gemm_blds_steps(lds, S.lin[0], id, cin_row, (S.d_id + 3) >> 2, h);
  } else {
    acc_init_bias(lds, S.lin[0], id, h);
    gemm_blds(lds, S.lin[0], id, cin_row, h);
  }
  if (pl.ctx_mlp) {
    // ContextSplineMap (flow.py:1419-1478): h_1 = relu(W_in c + b_in), h_{i+1} = relu(W_h h_i + b_h) for i = 1 ... reps
    // (hidden_layers_spline_context applications of ONE Linear: the reference repeats the same module object)
    f4 u[NSF_HT];
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[mt][r] = fmaxf(h[mt][r], 0.f);
    if (ast) ast_store<KSH>(ast, 0, h);
    for (int i = 0; i < pl.ctx_reps; ++i) {
      acc_init_bias(lds, S.lin[1], id, u);
      gemm_breg<KSH>(lds, S.lin[1], id, h, u);
#pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[mt][r] = fmaxf(u[mt][r], 0.f);
      if (ast) ast_store<KSH>(ast, 1 + i, h);
    }
    return;
  }
  if (ast) ast_store<KSH>(ast, 0, h);
  for (int b = 0; b < pl.NB; ++b) {
    f4 gate[NSF_HT], t[NSF_HT], u[NSF_HT];
    if (BX) {
#pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t[mt][r] = fmaxf(h[mt][r], 0.f);
          gate[mt][r] = bx[64 * (1 + b) + 16 * mt + 4 * r + id.g];
        }
    } else {
      acc_init_bias(lds, S.lin[1 + 3 * b], id, gate);
      gemm_blds(lds, S.lin[1 + 3 * b], id, cin_row + S.d_id, gate);
#pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t[mt][r] = fmaxf(h[mt][r], 0.f);
          gate[mt][r] = sigmoid_gate(gate[mt][r]);
        }
    }
    acc_init_bias(lds, S.lin[2 + 3 * b], id, u);
    gemm_breg<KSH>(lds, S.lin[2 + 3 * b], id, t, u);
    if (ast) { ast_store<KSH>(ast, 1 + 4 * b, u); ast_store<KSH>(ast, 3 + 4 * b, gate); }
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) u[mt][r] = fmaxf(u[mt][r], 0.f);
    acc_init_bias(lds, S.lin[3 + 3 * b], id, t);
    gemm_breg<KSH>(lds, S.lin[3 + 3 * b], id, u, t);
    if (ast) ast_store<KSH>(ast, 2 + 4 * b, t);
#pragma unroll
    for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[mt][r] += t[mt][r] * gate[mt][r];
    if (ast) ast_store<KSH>(ast, 4 + 4 * b, h);
  }
}

// final_layer for the spline dims [d0, d0+NACT) -> per-wave LDS staging pst[slot][row][param].
// NACT (active dim slots of this chunk) is a template parameter so the K loop is one
// branch-free stream of MFMAs the scheduler can software-pipeline.
// spline-parameter stash (training forward): `pstw` = this wave-tile's block + 4 * lane (nsf_plan.h, nsf_pst_tile_floats)
template <int PT>
__device__ __forceinline__ void pst_store(float* __restrict__ pstw, int dd, int pt, const f4& v) {
  __builtin_nontemporal_store(v, reinterpret_cast<f4*>(pstw + (dd * PT + pt) * 256));
}
template <int PT>
__device__ __forceinline__ f4 pst_load(const float* __restrict__ pstw, int dd, int pt) {
  return __builtin_nontemporal_load(reinterpret_cast<const f4*>(pstw + (dd * PT + pt) * 256));
}
template <int PT, int KSH, int NACT>
__device__ __forceinline__ void final_layer_chunk_n(const float* __restrict__ lds, float* __restrict__ pst,
                                                    const NsfPlan& pl, const ShapeDesc& S, const LaneId& id,
                                                    const f4 (&h)[NSF_HT], int d0, float* __restrict__ pstw = nullptr) {
  const LinDesc& L = S.lin[S.fin];
  f4 acc[NACT][PT];
  int ro[NACT][PT];
#pragma unroll
  for (int sl = 0; sl < NACT; ++sl) {
    const int dd = d0 + sl;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int p = 16 * pt + id.iperm;
      ro[sl][pt] = L.l_w + ((p < pl.P) ? dd * pl.P + p : L.out) * L.ldk + id.g;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[sl][pt][r] = lds[L.l_b + dd * 16 * PT + 16 * pt + 4 * r + id.g];
    }
  }
#pragma unroll
  for (int s = 0; s < KSH; ++s) {
    const float bv = h[s >> 2][s & 3];
#pragma unroll
    for (int sl = 0; sl < NACT; ++sl)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[sl][pt] = MFMA16(lds[ro[sl][pt] + 4 * s], bv, acc[sl][pt]);
  }
#pragma unroll
  for (int sl = 0; sl < NACT; ++sl)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int r = 0; r < 4; ++r)   // rows are 16*PT+1 wide: padding outputs (zeros) are stored too, no branch
        pst[sl * pl.DS + id.j * pl.PSW + 16 * pt + 4 * r + id.g] = acc[sl][pt][r];
  if (pstw) {
#pragma unroll
    for (int sl = 0; sl < NACT; ++sl)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) pst_store<PT>(pstw, d0 + sl, pt, acc[sl][pt]);
  }
}

// The same final-layer GEMM as a resumable stream: operator() issues ONE MFMA (K-step major, then
// dim slot, then row tile) and pins it in program order; finish() drains the rest and stores.
template <int PT, int KSH, int NACT>
struct FinalLayerStream {
  f4 acc[NACT][PT];
  int ro[NACT][PT];
  const float* lds;
  const f4* h;
  float abuf[4];
  __device__ __forceinline__ void init(const float* __restrict__ lds_, const NsfPlan& pl, const ShapeDesc& S,
                                       const LaneId& id, const f4 (&h_)[NSF_HT], int d0) {
    const LinDesc& L = S.lin[S.fin];
    lds = lds_;
    h = h_;
#pragma unroll
    for (int sl = 0; sl < NACT; ++sl) {
      const int dd = d0 + sl;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int p = 16 * pt + id.iperm;
        ro[sl][pt] = L.l_w + ((p < pl.P) ? dd * pl.P + p : L.out) * L.ldk + id.g;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[sl][pt][r] = lds[L.l_b + dd * 16 * PT + 16 * pt + 4 * r + id.g];
      }
    }
    abuf[0] = a_load(0);
    abuf[1] = a_load(1);
  }
  __device__ __forceinline__ float a_load(int n) const {
    const int s = n / (NACT * PT), rem = n % (NACT * PT);
    return lds[ro[rem / PT][rem % PT] + 4 * s];
  }
  // n is a compile-time constant at every call site.  The A operand of step n was requested two
  // yield points earlier (the scheduling barriers would otherwise pin each LDS read right in front
  // of its MFMA and expose the full LDS latency 52 times).
  __device__ __forceinline__ void step(int n) {
    const int s = n / (NACT * PT), rem = n % (NACT * PT);
    const int sl = rem / PT, pt = rem % PT;
    const float a = abuf[n & 3];
    if (n + 2 < NACT * PT * KSH) abuf[(n + 2) & 3] = a_load(n + 2);
    acc[sl][pt] = MFMA16(a, h[s >> 2][s & 3], acc[sl][pt]);
  }
  __device__ __forceinline__ void operator()(int n) {
    if (n < NACT * PT * KSH) step(n);
    __builtin_amdgcn_sched_barrier(0);
  }
  // nyield = number of yield points the host routine went through (steps 0 .. nyield-1 are done)
  template <int NYIELD>
  __device__ __forceinline__ void finish(float* __restrict__ pst, const NsfPlan& pl, const LaneId& id,
                                         float* __restrict__ pstw = nullptr, int d0 = 0) {
#pragma unroll
    for (int n = NYIELD; n < NACT * PT * KSH; ++n) step(n);
    if (pstw) {
#pragma unroll
      for (int sl = 0; sl < NACT; ++sl)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) pst_store<PT>(pstw, d0 + sl, pt, acc[sl][pt]);
    }
#pragma unroll
    for (int sl = 0; sl < NACT; ++sl)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pst[sl * pl.DS + id.j * pl.PSW + 16 * pt + 4 * r + id.g] = acc[sl][pt][r];
  }
};

// ---- rational-quadratic spline: one (row, dim) task per LANE PAIR (lane, lane^32) -------
// Restates nflows 0.14 transforms/splines/rational_quadratic.py
// (unconstrained_rational_quadratic_spline, tails="linear") with the constants sbi passes
// (flow.py:425-432; estimator_configs.py:49-51).  `p` points at the 3K-1 raw conditioner
// outputs of the task.  The two lanes of a pair split the work: part 0 (lanes 0-31) owns the
// bin WIDTHS (softmax, knot cumsum, the forward bin search, derivative d_i), part 1 (lanes
// 32-63) the bin HEIGHTS (softmax, knots, the inverse bin search, d_{i+1}); the handful of
// selected scalars is exchanged with v_permlane32_swap and both lanes evaluate the rational-
// quadratic core.  This halves the longest dependency chains of a VALU-latency-bound phase.
__device__ __forceinline__ float xchg32(float x) {   // value held by lane ^ 32
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  const unsigned u = __float_as_uint(x);
  const u2_t r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 32) ? r.x : r.y);
}
__device__ __forceinline__ int xchg32i(int x) { return __float_as_int(xchg32(__int_as_float(x))); }
// the values the pair's part-0 lane (lanes 0-31) and part-1 lane (32-63) hold, in BOTH lanes: one swap, no select
__device__ __forceinline__ void pair_both(float x, float& from0, float& from1) {
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  const unsigned u = __float_as_uint(x);
  const u2_t r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  from0 = __uint_as_float(r.x);
  from1 = __uint_as_float(r.y);
}
__device__ __forceinline__ void pair_both_i(int x, int& from0, int& from1) {
  float a, b;
  pair_both(__int_as_float(x), a, b);
  from0 = __float_as_int(a);
  from1 = __float_as_int(b);
}

template <int K>
struct SplineSide {
  float e[K];       // exp(logit - max): the softmax numerators
  float inv_s;      // 1 / sum(e)
  float c[K + 1];   // knots of this side in [-B, B]
  float n2, a;      // size of bin k on [-B, B]: e[k] * n2 + a   (n2 = 2B (1 - K min) / sum(e), a = 2B min)
};

// "Yield points": a VALU-heavy routine calls y() between small groups of instructions; the functor
// issues one MFMA of an independent matrix stream followed by a scheduling barrier, so an in-order
// wave alternates matrix and vector instructions in program order (hipcc otherwise emits all MFMAs
// first and the VALU never overlaps them).  NoYield compiles to nothing.
// Yield points are numbered 0 .. 5K (compile-time constants after unrolling): the functor maps the
// number straight to one MFMA of its stream, so every register index stays static.
struct NoYield {
  __device__ __forceinline__ void operator()(int) {}
};

// Spline parametrisation variants (VAR): 0 = nflows (rational_quadratic.py: softmax with minimum bin size,
// softplus + min_derivative knot slopes, logits optionally / sqrt(hidden), bins closed on the left, last knot
// bumped by 1e-6); 1 = zuko's MonotonicRQSTransform (transforms.py: logits soft-clipped by w / (1 + |2 w / ln slope|),
// slopes exp(d / (1 + |d / ln slope|)), plain softmax, bins closed on the right, identity outside (-B, B]).
// The rational-quadratic map, its inverse and its log-derivative are the same function of (knots, slopes).
#define ZUKO_CW 0.28952965460216784f   /* 2 / |ln 1e-3| */
#define ZUKO_CD 0.14476482730108392f   /* 1 / |ln 1e-3| */
// (PL: any struct with the spline constants B, min_w, min_h, min_d, inv_sqrt_h, one_minus_kw, one_minus_kh, d_const --
//  NsfPlan, or the compact kernel-constant block of the cooperative kernels)
template <int VAR, class PL>
__device__ __forceinline__ float spline_logit(float q, const PL& pl) {
  return VAR == 0 ? q * pl.inv_sqrt_h : q * rcp_f(1.f + fabsf(q) * ZUKO_CW);
}
// d logit' / d logit
template <int VAR, class PL>
__device__ __forceinline__ float spline_logit_grad(float q, const PL& pl) {
  if (VAR == 0) return pl.inv_sqrt_h;
  const float r = rcp_f(1.f + fabsf(q) * ZUKO_CW);
  return r * r;
}

// NSF_SOFTMAX_EXP: how the K softmax numerators of a side are taken (nflows parametrisation).  1 (default): exp2 of ONE
// fma -- the logit scale 1/sqrt(hidden), the max subtraction and log2(e) folded into it: two instructions per logit
// instead of six.  The exponent then carries its own rounding, <= |t| 1.2e-7 relative on a numerator, which a bin of
// softmax mass p turns into <= p |ln p| / (p + min_bin) of that on the bin's size (<= 2.6e-7 at min_bin = 1e-3).
// 0: the compensated exp_f (1 ulp).  Same-box A/B (profiles/r6h_spline_ab.txt, 16 384 rows against the oracle in fp64):
// log_prob rms 2.595e-6 / max 1.65e-5 with 1, 2.603e-6 / 1.62e-5 with 0 (eager fp32 reference: 6.2e-6 / 4.5e-5);
// log_prob -1.9 %, 10^6 draws -2.6 %.  (A/B: SBI_AMD_EXTRA_HIPCC_FLAGS=-DNSF_SOFTMAX_EXP=0 rebuilds the library.)
#ifndef NSF_SOFTMAX_EXP
#define NSF_SOFTMAX_EXP 1
#endif
template <int K, class Y = NoYield, int VAR = 0, class PL = NsfPlan>
__device__ __forceinline__ void spline_side(const float* __restrict__ q, const PL& pl, int part,
                                            SplineSide<K>& S, Y&& y = Y()) {
  const float B = pl.B;
  if (VAR == 0) {
    // nflows: `unnormalized_widths /= sqrt(hidden_features)` (coupling.py), then softmax: the scale is positive, so the
    // max is taken on the raw logits and scale + shift are one fma per logit (softmax is shift-invariant: the shift
    // needs no particular rounding)
    float m = q[0];
#pragma unroll
    for (int k = 1; k < K; ++k) {
      m = fmaxf(m, q[k]);
      y(k - 1);
    }
    y(K - 1);
    if (NSF_SOFTMAX_EXP == 1) {
      const float sc = pl.inv_sqrt_h * 1.4426950408889634f;
      const float nms = -m * sc;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        S.e[k] = __builtin_amdgcn_exp2f(fmaf(q[k], sc, nms));
        y(K + k);
      }
    } else {
      const float sc = pl.inv_sqrt_h;
      const float nms = -m * sc;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        S.e[k] = exp_f(fmaf(q[k], sc, nms));
        y(K + k);
      }
    }
  } else {
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      S.e[k] = spline_logit<VAR>(q[k], pl);
      m = fmaxf(m, S.e[k]);
      y(k);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      S.e[k] = exp_f(S.e[k] - m);
      y(K + k);
    }
  }
  // knots: knot_{k+1} = -B + 2B ((k+1) min + (1 - K min) P_k / s), P_k the running sum of the numerators (its last
  // element is the softmax denominator): one add and two FMAs per knot.  nflows reaches the same numbers as cumsum of
  // the normalised, padded sizes followed by the affine map (one more rounding per knot); the end knots are overwritten.
  float a = (2.f * B) * (part ? pl.min_h : pl.min_w);
  asm volatile("" : "+v"(a));      // (the K knot offsets below are loop invariants: hoisted, they would pin K registers)
  float P[K];
  P[0] = S.e[0];
#pragma unroll
  for (int k = 1; k < K; ++k) P[k] = P[k - 1] + S.e[k];
  S.inv_s = rcp_nr(P[K - 1]);
  const float n2 = ((2.f * B) * (part ? pl.one_minus_kh : pl.one_minus_kw)) * S.inv_s;
  S.n2 = n2;
  S.a = a;
  S.c[0] = -B;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    S.c[k + 1] = fmaf(P[k], n2, fmaf((float)(k + 1), a, -B));
    y(2 * K + k);
  }
  if (VAR == 0) S.c[K] = B;   // nflows overwrites the end knots; zuko keeps B (2 cumsum - 1)
}

struct SplineSel {   // per-task scalars both lanes hold after the exchange
  int idx;
  bool inside;
  float cw_i, ch_i;     // the bin's left knots
  float w_i, h_i;       // the bin's width and height
  float d_i, d_n, ud_mine;
};

template <int K, bool INV, class Y = NoYield, int VAR = 0, class PL = NsfPlan>
__device__ __forceinline__ void spline_select(const float* __restrict__ p, float x, const PL& pl, int part,
                                              const SplineSide<K>& S, SplineSel& o, Y&& y = Y()) {
  const float B = pl.B;
  // searchsorted (torchutils.py:449-463): sum(x >= knots) - 1, last knot + 1e-6; done by the
  // side that owns the searched knots (widths forward, heights inverse).
  // zuko: torch.searchsorted(knots, x) - 1 = #(knots < x) - 1, transformed iff 0 <= bin < K.
  int idx;
  if (VAR == 0) {
    // the clamp to [0, K - 1] that follows the count makes the end knots' comparisons redundant: x < -B and x > B + 1e-6
    // land in bins 0 and K - 1 either way (and are mapped by the identity, below)
    idx = 0;
    y(3 * K);
#pragma unroll
    for (int k = 1; k < K; ++k) {
      idx += (x >= S.c[k]) ? 1 : 0;
      y(3 * K + k);
    }
    o.inside = (x >= -B) && (x <= B);
  } else {
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      cnt += (x > S.c[k]) ? 1 : 0;
      y(3 * K + k);
    }
    cnt += (x > S.c[K]) ? 1 : 0;
    idx = cnt - 1;
    // the side that searched decides (both sides see the same x but own different knots)
    const int in_mine = (idx >= 0 && idx <= K - 1) ? 1 : 0;
    int in0, in1;
    pair_both_i(in_mine, in0, in1);
    o.inside = (INV ? in1 : in0) != 0;
    idx = idx < 0 ? 0 : (idx > K - 1 ? K - 1 : idx);
  }
  {
    int i0, i1;
    pair_both_i(idx, i0, i1);
    idx = INV ? i1 : i0;             // the searching side's bin, in both lanes
  }
  // the bin's left knot and its size.  The size is taken from the bin's own softmax numerator, e n2 + a (relative
  // error ~1e-7 however narrow the bin), not as the difference of two rounded knots as the eager reference does
  // (relative error ulp(knot) / size: 4e-5 for a bin of the minimum size) -- except in the last bin, whose right knot
  // nflows overwrites with B
  float c_i = S.c[0], e_i = S.e[0];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    const bool hit = (idx == k);
    c_i = hit ? S.c[k] : c_i;
    e_i = hit ? S.e[k] : e_i;
    y(4 * K + k - 1);
  }
  float sz = fmaf(e_i, S.n2, S.a);
  if (VAR == 0) sz = (idx == K - 1) ? B - c_i : sz;
  o.idx = idx;
  pair_both(c_i, o.cw_i, o.ch_i);    // part 0 owns the width knots, part 1 the height knots
  pair_both(sz, o.w_i, o.h_i);
  // derivatives: part 0 evaluates knot idx, part 1 knot idx+1; boundary knots use the constant
  const int kd = idx + part;
  const int kc = kd - 1 < 0 ? 0 : (kd - 1 > K - 2 ? K - 2 : kd - 1);   // always a valid slot
  float ud_ld = p[2 * K + kc];
  asm volatile("" : "+v"(ud_ld));    // (keeps the read unconditional: the compiler otherwise branches around it)
  o.ud_mine = ((unsigned)(kd - 1) >= (unsigned)(K - 1)) ? (VAR == 0 ? pl.d_const : 0.f) : ud_ld;
  y(5 * K - 1);
  const float d_mine = VAR == 0 ? pl.min_d + softplus_bf(o.ud_mine)
                                : exp_f(o.ud_mine * rcp_f(1.f + fabsf(o.ud_mine) * ZUKO_CD));   // boundary: exp(0) = 1
  y(5 * K);
  pair_both(d_mine, o.d_i, o.d_n);
}

// ---- the selected bin re-derived beyond fp32 (density direction, nflows parametrisation) ----------------------------
// Where log p loses its digits (tools/diag/spline_precision.py): not in the conditioner GEMMs
// (2e-7 of the 6e-6 rms error against an fp64 evaluation) and not in the order of the final sums (fp64 summation of
// the fp32 terms: 9.3 % -> 8.9 % of rows beyond 1e-5), but in the spline's own normalisation: the softmax denominator,
// the (1 - K min) p + min affine map, the knot cumsum and the bin width / height taken back as a DIFFERENCE of two
// rounded knots each carry ~1e-7 relative error into w and h, the log-derivative holds log((h / w)^2 ...), and 25
// spline evaluations add up.  The reference's eager fp32 arithmetic has the same error (and is what the oracle
// restates); an implementation is free to do better.  So: the bin is still FOUND on the fp32 knots (same bin as the
// backward pass, which recomputes them the same way), then its width, its left knot and x - knot are re-derived with
// the denominator and the prefix sum held in fp64 -- ~6 fp64-rate instructions per logit and side, a dozen per task.
// NSF_PRECISE_SPLINE=0 compiles it out (A/B: SBI_AMD_EXTRA_HIPCC_FLAGS=-DNSF_PRECISE_SPLINE=0 rebuilds the library).
#ifndef NSF_PRECISE_SPLINE
#define NSF_PRECISE_SPLINE 1
#endif
__device__ __forceinline__ double rcp_d(double x) {
  const double r = __builtin_amdgcn_rcp(x);     // v_rcp_f64: >= 2^-20; one Newton step squares the error (1e-12: plenty)
  return fma(r, fma(-x, r, 1.0), r);
}
// this side's selected bin: extent (width or height) and left knot, both in fp64, from the side's K exps
template <int K, class PL>
__device__ __forceinline__ void precise_bin(const SplineSide<K>& S, int idx, const PL& pl, int part, double& extent,
                                            double& knot) {
  double sd = 0.0, pd = 0.0;
  float e_i = S.e[0];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const bool hit = (k == idx);
    pd = hit ? sd : pd;             // prefix sum of the bins left of the selected one
    e_i = hit ? S.e[k] : e_i;
    sd += (double)S.e[k];
  }
  const double mn = (double)(part ? pl.min_h : pl.min_w);
  const double n = (1.0 - mn * (double)K) * rcp_d(sd);
  const double twoB = 2.0 * (double)pl.B;
  extent = twoB * fma((double)e_i, n, mn);
  knot = fma(fma(pd, n, mn * (double)idx), twoB, -(double)pl.B);
  if (idx == K - 1) extent = (double)pl.B - knot;     // nflows overwrites the last knot with B
}

// forward returns logabsdet, inverse returns -logabsdet (as nflows does); both lanes of the pair
// receive y and ld.
// PREC = false: the plain fp32 bin (what the eager reference computes).  The forward half of the fused training step
// (sbi_amd_nsf_loss_fwd_bwd, static default layout) runs that way: its log p only feeds the reported training loss,
// the backward pass recomputes the spline in fp32 anyway, and the fp64 re-derivation costs ~4 % of the forward kernel.
// Every log_prob a caller can see -- sbi_amd_nsf_log_prob, and sbi_amd_nsf_train_forward behind `estimator.log_prob`
// under autograd and the atomic loss -- keeps PREC = true.
template <int K, bool INV, int VAR, bool PREC, class Y, class PL>
__device__ __forceinline__ void rq_spline_pair_impl(const float* __restrict__ p, float x, const PL& pl, int part,
                                                    float& y, float& ld, Y&& yield) {
  SplineSide<K> S;
  spline_side<K, Y, VAR>(p + part * K, pl, part, S, static_cast<Y&&>(yield));
  SplineSel o;
  spline_select<K, INV, Y, VAR>(p, x, pl, part, S, o, static_cast<Y&&>(yield));
  float w_i = o.w_i;
  float h_i = o.h_i;
  float xm = x - o.cw_i;          // distance of the input from the bin's left knot
  float ch_lo = 0.f;              // low word of the bin's bottom knot
  if (PREC && NSF_PRECISE_SPLINE && !INV && VAR == 0) {   // (compile time: the fp32 knot selects of spline_select then fold away)
    double ext, knot;
    precise_bin<K>(S, o.idx, pl, part, ext, knot);
    // part 0 holds the width side, part 1 the height side: each rounds its own three scalars, then they swap
    const float ext_f = (float)ext;
    const float a_f = part ? (float)knot : (float)((double)x - knot);          // heights: knot (hi) | widths: x - knot
    const float b_f = part ? (float)(knot - (double)(float)knot) : 0.f;        // heights: knot (lo)
    float unused;
    pair_both(ext_f, w_i, h_i);
    pair_both(a_f, xm, o.ch_i);
    pair_both(b_f, unused, ch_lo);
  }
  const float rw_i = rcp_nr(w_i);
  const float delta = h_i * rw_i;
  float yo, lo;
  if (!INV) {
    // (the bin was found on the fp32 knots: against the re-derived knot the input can sit an ulp outside [0, 1], and
    //  in a saturated bin -- slope 1e3 between slopes 1e-3 -- that turns the derivative's numerator negative)
    const float th = (PREC && NSF_PRECISE_SPLINE) ? __builtin_amdgcn_fmed3f(xm * rw_i, 0.f, 1.f) : xm * rw_i;
    const float tt = th * (1.f - th);
    const float num = h_i * (delta * (th * th) + o.d_i * tt);
    const float den = delta + ((o.d_i + o.d_n - 2.f * delta) * tt);
    yo = o.ch_i + (num * rcp_nr(den) + ch_lo);
    const float omt = 1.f - th;
    const float dnum = (delta * delta) * (o.d_n * (th * th) + 2.f * delta * tt + o.d_i * (omt * omt));
    lo = log_f(dnum) - 2.f * log_f(den);
  } else {
    const float s = o.d_i + o.d_n - 2.f * delta;
    const float xc = x - o.ch_i;
    const float a = xc * s + h_i * (delta - o.d_i);
    const float b = h_i * o.d_i - xc * s;
    const float c = -delta * xc;
    // In exact arithmetic disc >= 0 and root in [0, 1].  In a saturated bin (slope ~1e3 against ~1e-3 next door) the
    // fp32 discriminant cancels to a few ulps of b^2 either side of zero and the eager reference itself returns NaN
    // there (nflows asserts disc >= 0) -- on WHICH rows depends on the last-bit rounding of the knots.  The kernel
    // clamps both instead: well-conditioned rows are untouched bit for bit, the others stay finite (root in [0, 1]
    // makes den >= delta / 2 + (d_i + d_n) / 4 > 0 and dnum > 0), so a sampler never hands NaN downstream
    // (tests/test_spline_adversarial_gpu.py).
    const float disc = fmaxf(b * b - 4.f * a * c, 0.f);
    const float root = fminf(fmaxf((2.f * c) * rcp_nr(-b - sqrtf(disc)), 0.f), 1.f);
    yo = root * w_i + o.cw_i;
    const float tt = root * (1.f - root);
    const float den = delta + s * tt;
    const float omr = 1.f - root;
    const float dnum = (delta * delta) * (o.d_n * (root * root) + 2.f * delta * tt + o.d_i * (omr * omr));
    lo = -(log_f(dnum) - 2.f * log_f(den));
  }
  y = o.inside ? yo : x;
  ld = o.inside ? lo : 0.f;
}
template <int K, bool INV, class Y = NoYield, int VAR = 0, class PL = NsfPlan>
__device__ __forceinline__ void rq_spline_pair(const float* __restrict__ p, float x, const PL& pl, int part,
                                               float& y, float& ld, Y&& yield = Y()) {
  rq_spline_pair_impl<K, INV, VAR, true>(p, x, pl, part, y, ld, static_cast<Y&&>(yield));
}

// ---- LULinear on the per-wave state rows (nflows transforms/lu.py) ----------
// dense D x D mat-vec on a per-wave row buffer for D <= 16: the row is pulled into registers
// first and every LDS read is independent, so the loads pipeline instead of forming one
// latency-bound chain per element (mat-vec itself: dense_mv16c below).
// The per-wave row buffers are followed by further finite scratch, and M is zero padded to
// 16 x 16, so neither needs a bounds check: entries past D meet a zero.
__device__ __forceinline__ void row_to_regs16(const float* __restrict__ row, int D, float (&v)[16]) {
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = row[k];
}
// 16 x 16 mat-vec with CONTIGUOUS outputs per lane group: out[ii] = sum_k M(4g + ii, k) v[k]
// (TRANSPOSED: M(i,k) = m[k][i], a float4 per k; else M(i,k) = m[i][k], four float4 per output).
// m is zero padded to 16 x 16 and 16-byte aligned (nsf_plan.cpp).
template <bool TRANSPOSED>
__device__ __forceinline__ void dense_mv16c(const float* __restrict__ m, const float (&v)[16], int g, float (&out)[4]) {
#pragma unroll
  for (int ii = 0; ii < 4; ++ii) out[ii] = 0.f;
  if (TRANSPOSED) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const f4 w = *(const f4*)(m + k * 16 + 4 * g);
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) out[ii] = fmaf(w[ii], v[k], out[ii]);
    }
  } else {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const f4 w = *(const f4*)(m + (4 * g + ii) * 16 + 4 * k4);
#pragma unroll
        for (int u = 0; u < 4; ++u) out[ii] = fmaf(w[u], v[4 * k4 + u], out[ii]);
      }
  }
}

// forward: y = L (U z) + b           (F.linear(F.linear(x, U), L, bias))
__device__ __forceinline__ void lu_forward(const float* __restrict__ lds, const NsfPlan& pl, const ShapeDesc& S,
                                           const LaneId& id, float* __restrict__ zs, float* __restrict__ us) {
  const int D = pl.D;
  const int LUS = D <= 16 ? 16 : D;
  if (D <= 16) {
    // two chained 16 x 16 mat-vecs on the matrix pipe (was: 128 VALU FMAs per lane group, 32 ds_read_b128 and an LDS
    // round trip through `us`).  K-step s covers k = 4 g + s, so a lane's four A values are ONE 16-byte read of row j,
    // and the D fragment of U z (reg r of lane (j, g) = dim 4 g + r of row j) is the B operand of L u unchanged.
    // (U, L zero padded to 16 x 16; the state rows are followed by finite scratch: no bounds checks.)
    const f4 au = *(const f4*)(lds + S.l_U + id.j * 16 + 4 * id.g);
    const f4 al = *(const f4*)(lds + S.l_L + id.j * 16 + 4 * id.g);
    float bz[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) bz[s4] = zs[id.j * pl.ZW + 4 * id.g + s4];
    f4 u = {0.f, 0.f, 0.f, 0.f}, y;
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = 4 * id.g + r < D ? lds[S.l_lub + 4 * id.g + r] : 0.f;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) u = MFMA16(au[s4], bz[s4], u);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) y = MFMA16(al[s4], u[s4], y);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * id.g + r < D) zs[id.j * pl.ZW + 4 * id.g + r] = y[r];
    wave_lds_fence();
    return;
  }
  for (int i = id.g; i < D; i += 4) {
    float a = 0.f;
    for (int k = i; k < D; ++k) a += lds[S.l_U + i * LUS + k] * zs[id.j * pl.ZW + k];
    us[id.j * pl.ZW + i] = a;
  }
  wave_lds_fence();
  for (int i = id.g; i < D; i += 4) {
    float a = lds[S.l_lub + i];
    for (int k = 0; k <= i; ++k) a += lds[S.l_L + i * LUS + k] * us[id.j * pl.ZW + k];
    zs[id.j * pl.ZW + i] = a;
  }
  wave_lds_fence();
}
// inverse: z = U^{-1} L^{-1} (y - b) by forward/back substitution; one lane per row.
__device__ __forceinline__ void lu_inverse(const float* __restrict__ lds, const NsfPlan& pl, const ShapeDesc& S,
                                           const LaneId& id, float* __restrict__ zs, float* __restrict__ us) {
  const int D = pl.D;
  const int LUS = D <= 16 ? 16 : D;
  if (S.l_Ui >= 0) {   // D <= 16: z = U^-1 (L^-1 (y - b)) with the inverses the pack kernel prepared, as in lu_forward
    const f4 ali = *(const f4*)(lds + S.l_Li + id.j * 16 + 4 * id.g);
    const f4 aui = *(const f4*)(lds + S.l_Ui + id.j * 16 + 4 * id.g);
    float bv[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int k = 4 * id.g + s4;
      bv[s4] = zs[id.j * pl.ZW + k] - (k < D ? lds[S.l_lub + k] : 0.f);
    }
    f4 t = {0.f, 0.f, 0.f, 0.f}, z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) t = MFMA16(ali[s4], bv[s4], t);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) z = MFMA16(aui[s4], t[s4], z);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * id.g + r < D) zs[id.j * pl.ZW + 4 * id.g + r] = z[r];
    wave_lds_fence();
    return;
  }
  if (id.g == 0) {
    float* z = zs + id.j * pl.ZW;
    float* u = us + id.j * pl.ZW;
    for (int i = 0; i < D; ++i) {
      float a = z[i] - lds[S.l_lub + i];
      for (int k = 0; k < i; ++k) a -= lds[S.l_L + i * LUS + k] * u[k];
      u[i] = a;
    }
    for (int i = D - 1; i >= 0; --i) {
      float a = u[i];
      for (int k = i + 1; k < D; ++k) a -= lds[S.l_U + i * LUS + k] * z[k];
      z[i] = a / lds[S.l_U + i * LUS + i];
    }
  }
  wave_lds_fence();
}
// sum_i log U_ii, precomputed by the pack kernel (slot right behind the LU bias)
__device__ __forceinline__ float lu_logabsdet(const float* __restrict__ lds, const NsfPlan& pl, const ShapeDesc& S) {
  return lds[S.l_lub + pl.D];
}

// conditioner input rows: cin[j] = [ z[identity dims] ; standardized context ; 0 pad ].
// For C <= 16 the context of lane (j,g) is held in registers: cr[u] = c[g + 4u].
// broadcast-x kernels: the conditioner input row holds the identity features only, zero up to the K-steps the initial
// layer runs ((d_id + 3) / 4 of them)
__device__ __forceinline__ void build_cin_bx(const NsfPlan& pl, const ShapeDesc& S, int parity, const LaneId& id,
                                             const float* __restrict__ zs, float* __restrict__ cin) {
  const int kmax = ((S.d_id + 3) >> 2) << 2;
  for (int k = id.g; k < kmax; k += 4)
    cin[id.j * pl.CINW + k] = k < S.d_id ? zs[id.j * pl.ZW + 2 * k + (1 - parity)] : 0.f;
  wave_lds_fence();
}

// broadcast-x kernels, once per transform and workgroup, between the two barriers of the weight staging: the
// context-only terms of the conditioner from the transform's packed image in global memory (`img`: L2 resident, read
// by every workgroup) and the standardized condition row `cstd` (LDS) into the LDS table `bx` (conditioner_hidden).
__device__ __forceinline__ void bx_fold_context(const float* __restrict__ img, const NsfPlan& pl, const ShapeDesc& S,
                                                const float* __restrict__ cstd, float* __restrict__ bx,
                                                int tid, int nthreads) {
  const int C = pl.C;
  for (int idx = tid; idx < 64 * (1 + pl.NB); idx += nthreads) {
    const int which = idx >> 6, f = idx & 63;
    const LinDesc& L = S.lin[which == 0 ? 0 : 1 + 3 * (which - 1)];
    float v = which == 0 ? 0.f : 0.5f;          // padding features: what the per-row path computes there
    if (f < L.out) {
      const float* __restrict__ w = img + L.l_w + f * L.ldk + (which == 0 ? S.d_id : 0);
      float a = img[L.l_b + f];
      for (int c = 0; c < C; ++c) a = fmaf(w[c], cstd[c], a);
      v = which == 0 ? a : sigmoid_gate(a);
    }
    bx[idx] = v;
  }
}

__device__ __forceinline__ void build_cin(const NsfPlan& pl, const ShapeDesc& S, int parity, const LaneId& id,
                                          const float* __restrict__ zs, const float* __restrict__ cs,
                                          const float (&cr)[4], float* __restrict__ cin) {
  if (pl.C <= 16) {
    for (int k = id.g; k < pl.CINW; k += 4)
      if (k < S.d_id || k >= S.in0) cin[id.j * pl.CINW + k] = k < S.d_id ? zs[id.j * pl.ZW + 2 * k + (1 - parity)] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = id.g + 4 * u;
      if (c < pl.C) cin[id.j * pl.CINW + S.d_id + c] = cr[u];
    }
  } else {
    for (int k = id.g; k < pl.CINW; k += 4) {
      float v = 0.f;
      if (k < S.d_id) v = zs[id.j * pl.ZW + 2 * k + (1 - parity)];
      else if (k < S.in0) v = cs[id.j * pl.CW + (k - S.d_id)];
      cin[id.j * pl.CINW + k] = v;
    }
  }
  wave_lds_fence();
}
