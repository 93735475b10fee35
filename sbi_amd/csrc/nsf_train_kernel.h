#pragma once
// nsf_train_kernel.h -- NPE training pass for the NSF estimator on gfx950:
//   loss_n = -log p(theta_n | x_n)                       (NFlowsFlow.loss, nflows_flow.py:99-109)
//   grad   = d( sum_n w_n loss_n ) / d params            (what loss.mean().backward() produces,
//                                                         trainers/base.py:1178-1181)
// Structure (DESIGN.md "training pass"):
//   1. forward flow kernel with the per-transform input state stashed (T*N*D floats);
//   2. one backward launch per transform, last -> first.  Persistent workgroups of 4
//      waves walk 64-row tiles: each wave recomputes its 16 rows' conditioner
//      activations on MFMA (registers), back-propagates spline -> conditioner -> input
//      on MFMA with the transposed weight image, and the four waves share their
//      (activation, gradient) tiles through LDS so that every wave accumulates a fixed
//      quarter of the layer's weight-gradient tiles in registers across all of the
//      workgroup's rows;
//   3. a deterministic reduction of the per-workgroup partial gradients (no atomics).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "nsf_device.h"

#define TR_NW 4            // waves per workgroup
#define TR_ROWS 64         // rows per tile
#define TR_MAXCH 4         // spline chunks per transform the accumulators are sized for
#define TR_GRID_MAX 256    // persistent workgroups (one per CU)

struct TrainPlan {
  int SA;                  // row stride of the shared staging tiles
  int o_Ast, o_Bst;        // LDS float offsets of the staging tiles [64][SA]
  int o_wave, w_total;     // per-wave scratch base / size
  int w_zs, w_ys, w_gys, w_gxs, w_gzs, w_cs, w_cin, w_us, w_gus;
  int DCHB, PTW;           // spline dims per chunk, floats per dim slot (16*PT)
  int nch[2];              // chunks per mask parity
  int PLP;                 // floats per (workgroup, transform) partial-gradient slab
  int grid, ntiles;
  int lds_floats;
};

static int build_train_plan(const NsfPlan& pl, int64_t n, TrainPlan* tp) {
  if (pl.D > 15 || pl.H > 63 || pl.NB > 2 || (pl.NB < 1 && !pl.ctx_mlp)) return SBI_AMD_E_UNSUPPORTED;
  const int d_id_max = pl.shape[0].d_id > pl.shape[1].d_id ? pl.shape[0].d_id : pl.shape[1].d_id;
  if (d_id_max + pl.C + 1 > 32 || pl.C + 1 > 32) return SBI_AMD_E_UNSUPPORTED;
  tp->DCHB = 4 / pl.PT;
  if (tp->DCHB < 1) return SBI_AMD_E_UNSUPPORTED;
  if (tp->DCHB > 2) tp->DCHB = 2;   // a spline task occupies a lane pair
  tp->PTW = 16 * pl.PT;
  for (int par = 0; par < 2; ++par) {
    tp->nch[par] = (pl.shape[par].d_tr + tp->DCHB - 1) / tp->DCHB;
    if (tp->nch[par] > TR_MAXCH) return SBI_AMD_E_UNSUPPORTED;
  }
  tp->SA = 68;
  int o = pl.lds_w_floats;
  tp->o_Ast = o; o += TR_ROWS * tp->SA;
  tp->o_Bst = o; o += TR_ROWS * tp->SA;
  tp->o_wave = o;
  int w = 0;
  tp->w_zs = w; w += 16 * pl.ZW;
  tp->w_ys = w; w += 16 * pl.ZW;
  tp->w_gys = w; w += 16 * pl.ZW;
  tp->w_gxs = w; w += 16 * pl.ZW;
  tp->w_gzs = w; w += 16 * pl.ZW;
  tp->w_us = w; w += 16 * pl.ZW;
  tp->w_gus = w; w += 16 * pl.ZW;
  tp->w_cs = w; w += 16 * pl.CW;
  tp->w_cin = w; w += 16 * pl.CINW;
  tp->w_total = (w + 3) / 4 * 4;
  tp->lds_floats = tp->o_wave + TR_NW * tp->w_total;
  if (4ll * tp->lds_floats > NSF_LDS_LIMIT_BYTES) return SBI_AMD_E_LDS;
  int pmax = pl.shape[0].n_params > pl.shape[1].n_params ? pl.shape[0].n_params : pl.shape[1].n_params;
  tp->PLP = (pmax + 1 + 3) / 4 * 4;
  tp->ntiles = (int)((n + TR_ROWS - 1) / TR_ROWS);
  tp->grid = tp->ntiles < TR_GRID_MAX ? tp->ntiles : TR_GRID_MAX;
  return 0;
}

// ------------------------------------------------------------------ device helpers
// D-fragment (lane (g,j), tile mt, reg r = feature 16mt+4r+g of row j) -> row-major tile
__device__ __forceinline__ void stage_D(float* __restrict__ st, int SA, int row, const LaneId& id,
                                        const f4 (&v)[NSF_HT], bool relu) {
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = v[mt][r];
      st[row * SA + 16 * mt + 4 * r + id.g] = relu ? fmaxf(a, 0.f) : a;
    }
}

// weight-gradient tile(s): acc[nt] += sum_{rows of the 64-row tile} A[row][acol0+i] * B[row][bcol0+16nt+j]
template <int NT>
__device__ __forceinline__ void dw_gemm(const float* __restrict__ Ast, const float* __restrict__ Bst, int SA,
                                        int acol0, int bcol0, const LaneId& id, f4 (&acc)[NT], int nt_on = NT,
                                        int abl = 0) {
  if (abl & 1) return;
#pragma unroll 4
  for (int s = 0; s < TR_ROWS / 4; ++s) {
    const int row = 4 * s + id.g;
    const float a = Ast[row * SA + acol0 + id.j];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      if (nt < nt_on) acc[nt] = MFMA16(a, Bst[row * SA + bcol0 + 16 * nt + id.j], acc[nt]);
  }
}

// row-wise backward through a linear layer: acc[mt] += sum_k W[k][feat(mt)] * g[k], g = D fragments.
// The image keeps rows [out, 4*KS) zero (nsf_plan.cpp: rows_alloc), so K-steps past `out` need no
// predicate; lanes that supply an A row for an in-feature slot >= in load a neighbouring (finite)
// weight and replace it by zero with one v_cndmask.
template <int KS, int MT>
__device__ __forceinline__ void gemm_T_breg(const float* __restrict__ lds, const LinDesc& L, const LaneId& id,
                                            const f4 (&gb)[NSF_HT], f4 (&acc)[MT], int abl = 0) {
  if (abl & 2) return;
  const float* base[MT];
  bool ok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int f = 16 * mt + id.iperm;
    ok[mt] = f < L.in;
    base[mt] = lds + L.l_w + id.g * L.ldk + (ok[mt] ? f : 0);
  }
  const int kstride = 4 * L.ldk;
  float a_cur[MT], a_nxt[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) a_cur[mt] = ok[mt] ? base[mt][0] : 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + 1 < KS) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = ok[mt] ? base[mt][(s + 1) * kstride] : 0.f;
    }
    const float bv = gb[s >> 2][s & 3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = MFMA16(a_cur[mt], bv, acc[mt]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a_cur[mt] = a_nxt[mt];
  }
}

// final_layer output for the backward chunk -> this wave's rows of the shared A tile
template <int PT, int KSH>
__device__ __forceinline__ void final_layer_chunk_T(const float* __restrict__ lds, float* __restrict__ arow,
                                                    const NsfPlan& pl, const TrainPlan& tp, const ShapeDesc& S,
                                                    const LaneId& id, const f4 (&h)[NSF_HT], int d0) {
  constexpr int DCHB = (4 / PT) > 2 ? 2 : (4 / PT);
  const LinDesc& L = S.lin[S.fin];
  f4 acc[DCHB][PT];
  int ro[DCHB][PT];
#pragma unroll
  for (int sl = 0; sl < DCHB; ++sl) {
    const int dd = d0 + sl;
    const bool on = dd < S.d_tr;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int p = 16 * pt + id.iperm;
      ro[sl][pt] = L.l_w + ((on && p < pl.P) ? dd * pl.P + p : L.out) * L.ldk + id.g;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[sl][pt][r] = on ? lds[L.l_b + dd * 16 * PT + 16 * pt + 4 * r + id.g] : 0.f;
    }
  }
#pragma unroll
  for (int s = 0; s < KSH; ++s) {
    const float bv = h[s >> 2][s & 3];
#pragma unroll
    for (int sl = 0; sl < DCHB; ++sl)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[sl][pt] = MFMA16(lds[ro[sl][pt] + 4 * s], bv, acc[sl][pt]);
  }
#pragma unroll
  for (int sl = 0; sl < DCHB; ++sl)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int r = 0; r < 4; ++r) arow[id.j * tp.SA + sl * 16 * PT + 16 * pt + 4 * r + id.g] = acc[sl][pt][r];
}

// RQ spline forward + reverse-mode gradient for one (row, dim) task on a lane pair (see
// rq_spline_pair).  `p` holds the 3K-1 raw conditioner outputs on entry and
// d(gy*y + gl*logabsdet)/d(raw outputs) on exit (entries [3K-1, plen) zeroed): part 0 writes the
// width logits and all derivative slots but one, part 1 the height logits, its own derivative
// slot and the padding.  Formulas: DESIGN.md "spline backward".
template <int K>
__device__ __forceinline__ void rq_spline_pair_bwd(float* __restrict__ p, int plen, float x, float gy, float gl,
                                                   const NsfPlan& pl, int part, float& y, float& gx) {
  const float B = pl.B;
  SplineSide<K> S;
  spline_side<K>(p + part * K, pl, part, S);
  SplineSel o;
  spline_select<K, false>(p, x, pl, part, S, o);
  const bool inside = o.inside;
  const int idx = o.idx;
  const float d_i = o.d_i, d_n = o.d_n;
  // ---- forward
  const float w = o.cw_n - o.cw_i, h = o.ch_n - o.ch_i;
  const float rw = rcp_f(w);
  const float delta = h * rw;
  const float th = (x - o.cw_i) * rw;
  const float omt = 1.f - th;
  const float tt = th * omt;
  const float q = delta * (th * th) + d_i * tt;
  const float num = h * q;
  const float s = d_i + d_n - 2.f * delta;
  const float den = delta + s * tt;
  const float rden = rcp_f(den);
  const float r = d_n * (th * th) + 2.f * delta * tt + d_i * (omt * omt);
  y = inside ? (o.ch_i + num * rden) : x;
  // ---- reverse
  float gc = gy;
  const float gnum = gy * rden;
  const float gden = -gy * num * rden * rden - 2.f * gl * rden;
  const float gr = gl * rcp_f(r);
  float gdelta = 2.f * gl * rcp_f(delta);
  float gdn = gr * (th * th);
  float gth = gr * 2.f * d_n * th;
  gdelta += gr * 2.f * tt;
  float gt = gr * 2.f * delta;
  float gdi = gr * (omt * omt);
  float gomt = gr * 2.f * d_i * omt;
  gdelta += gden;
  const float gs = gden * tt;
  gt += gden * s;
  gdi += gs;
  gdn += gs;
  gdelta -= 2.f * gs;
  float gh = gnum * q;
  const float gq = gnum * h;
  gdelta += gq * (th * th);
  gth += gq * 2.f * delta * th;
  gdi += gq * tt;
  gt += gq * d_i;
  gth += gt * omt;
  gomt += gt * th;
  gth -= gomt;
  const float gxi = gth * rw;
  float ga = -gth * rw;
  float gw = -gth * th * rw;
  gh += gdelta * rw;
  gw -= gdelta * delta * rw;
  const float gf = gh;
  gc -= gh;
  const float ge = gw;
  ga -= gw;
  gx = inside ? gxi : gy;
  // ---- my side: knots -> softmax logits
  const float g_lo = part ? gc : ga, g_hi = part ? gf : ge;
  const float Gi = (inside && idx >= 1) ? (2.f * B) * g_lo : 0.f;
  const float Gn = (inside && idx <= K - 2) ? (2.f * B) * g_hi : 0.f;
  const float omk = part ? pl.one_minus_kh : pl.one_minus_kw;
  float dot = 0.f;
  float gsm[K];
#pragma unroll
  for (int m = 0; m < K; ++m) {
    const float gwm = (m < idx) ? (Gi + Gn) : ((m == idx) ? Gn : 0.f);
    S.e[m] *= S.inv_s;   // softmax probabilities
    gsm[m] = omk * gwm;
    dot += gsm[m] * S.e[m];
  }
  float* q_out = p + part * K;
#pragma unroll
  for (int m = 0; m < K; ++m) q_out[m] = S.e[m] * (gsm[m] - dot) * pl.inv_sqrt_h;
  // ---- derivative slots: knot kd = idx + part is mine (interior knots only)
  const int kd = idx + part;
  const float gud = (inside && kd >= 1 && kd <= K - 1) ? (part ? gdn : gdi) * sigmoid_f(o.ud_mine) : 0.f;
  if (part == 0) {
#pragma unroll
    for (int k = 0; k < K - 1; ++k)
      if (k != idx) p[2 * K + k] = (k + 1 == idx) ? gud : 0.f;
  } else {
    if (idx <= K - 2) p[2 * K + idx] = gud;
    for (int k = 3 * K - 1; k < plen; ++k) p[k] = 0.f;
  }
}

// g_h += Wf[rows of the chunk's dims]^T g_p for this wave's 16 rows (B operand = the g_p this wave
// just wrote into the shared A tile).  Select-free like gemm_T_breg: the padding K slots (p >= P)
// carry exact-zero g_p, so whatever finite weight they meet is harmless; lanes supplying an A row
// for an in-feature >= H substitute 0.
template <int PT>
__device__ __forceinline__ void wft_chunk(const float* __restrict__ lds, const LinDesc& LF, const NsfPlan& pl,
                                          const ShapeDesc& S, const LaneId& id, const float* __restrict__ Arow,
                                          int SA, int d0, f4 (&gh)[NSF_HT]) {
  constexpr int DCHB = (4 / PT) > 2 ? 2 : (4 / PT);
  bool ok[NSF_HT];
  int col[NSF_HT];
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt) {
    const int f = 16 * mt + id.iperm;
    ok[mt] = f < LF.in;
    col[mt] = ok[mt] ? f : 0;
  }
#pragma unroll
  for (int sl = 0; sl < DCHB; ++sl) {
    const int dd = d0 + sl;
    if (dd < S.d_tr) {
      const float* wrow = lds + LF.l_w + (dd * pl.P + id.g) * LF.ldk;
      const float* brow = Arow + id.j * SA + sl * 16 * PT + id.g;
#pragma unroll
      for (int s = 0; s < 4 * PT; ++s) {
        const float bv = brow[4 * s];
#pragma unroll
        for (int mt = 0; mt < NSF_HT; ++mt) {
          const float a = wrow[4 * s * LF.ldk + col[mt]];
          gh[mt] = MFMA16(ok[mt] ? a : 0.f, bv, gh[mt]);
        }
      }
    }
  }
}

// partial-gradient write-out of one weight tile (lane (g,j), reg r: out = out0+4g+r, in = 16nt+j)
__device__ __forceinline__ void write_tile(float* __restrict__ part, const LinDesc& L, int out0, int nt,
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

// ------------------------------------------------------------------ backward kernel
// Wave specialisation: waves 0-3 ("row" waves) own 16 rows each and run recompute +
// row-wise backward; waves 4-7 ("grad" waves) own the weight-gradient accumulators
// (m-tile = wave-4 of every linear layer) and consume the (gradient, activation) tiles
// the row waves publish in LDS.  One wave of each kind shares a SIMD, so a row wave's
// transposed GEMM overlaps its partner's weight-gradient GEMM; both stay under 256 VGPRs.
// NBT = residual blocks; NBT == 0 selects the theta-dim-1 ContextSplineMap conditioner (compile time, so the
// residual-net instantiations carry none of its code or registers).
template <int K, int KSH, int NBT, int NCH>
__global__ void __launch_bounds__(128 * TR_NW, 2)
nsf_bwd_layer_kernel(const NsfPlan pl, const TrainPlan tp, const int t, const float* __restrict__ packed,
                     const float* __restrict__ zstats, const float* __restrict__ z_in,
                     const float* __restrict__ x, const float* __restrict__ gz_up,
                     const float* __restrict__ row_w, const float uni_w, long long n, long long x_rows,
                     float* __restrict__ gz_dn, float* __restrict__ partial, float* __restrict__ grad_theta,
                     const float* __restrict__ astash, long long* __restrict__ dbg) {
  const int dbg_tile_sel = pl.ablate & 128;   // timeline of the 2nd tile (warm caches) instead of the 1st
#define TS(i) do { if (dbg && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && tile == (int)(blockIdx.x + (dbg_tile_sel ? gridDim.x : 0))) \
    dbg[(threadIdx.x >> 6) * 64 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
  constexpr int PT = (3 * K - 1 + 15) / 16;
  constexpr int DCHB = (4 / PT) > 2 ? 2 : (4 / PT);   // dim slots per chunk (lane pairs: <= 2)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const LaneId id = make_lane();
  constexpr bool cm = (NBT == 0);             // theta-dim 1: context-only MLP conditioner, no LULinear
  constexpr int NB = cm ? 1 : NBT;            // ctx_mlp: one hidden H x H gradient tile set
  const int par = cm ? 0 : (t & 1);
  const ShapeDesc& S = pl.shape[par];
  const int D = pl.D, C = pl.C, SA = tp.SA;
  const bool is_last = (t == pl.T - 1);
  float* Ast = lds + tp.o_Ast;
  float* Bst = lds + tp.o_Bst;
  float* sc = lds + tp.o_wave + wave * tp.w_total;
  float* zs = sc + tp.w_zs;
  float* ys = sc + tp.w_ys;
  float* gys = sc + tp.w_gys;
  float* gxs = sc + tp.w_gxs;
  float* gzs = sc + tp.w_gzs;
  float* us = sc + tp.w_us;
  float* gus = sc + tp.w_gus;
  float* cs = sc + tp.w_cs;
  float* cin = sc + tp.w_cin;
  const int arow0 = 16 * wave;                 // this wave's rows inside the shared tiles
  float* Arow = Ast + arow0 * SA;
  float* Brow = Bst + arow0 * SA;
  const float* x_mean = zstats + 2 * D;
  const float* x_std = x_mean + C;

  stage_layer(lds, packed + (long long)t * pl.img_floats, pl.img_floats, tid, blockDim.x);

  const LinDesc& L0 = S.lin[0];
  const LinDesc& LF = S.lin[S.fin];
  const int nch = tp.nch[par];
  const int nt0 = (S.in0 + 1 + 15) / 16;        // n-tiles of d W0 (incl. the bias column)
  const int ntc = (C + 1 + 15) / 16;
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};

  if (wave < TR_NW) {
    // =========================== row waves ===========================
    // the guard-free mat-vec helpers read up to 6 floats past a 10-float row: make sure that
    // never is uninitialised LDS (NaN x 0 = NaN)
    for (int i = id.lane; i < tp.w_total; i += 64) sc[i] = 0.f;
    const LaneId id0 = id;
    for (int tile = blockIdx.x; tile < tp.ntiles; tile += gridDim.x) {
      // Re-materialise the lane coordinates per tile: otherwise LICM hoists every
      // lane-dependent LDS address of the body out of the persistent loop and the
      // kernel drowns in live registers (hundreds of spills).
      LaneId id = id0;
      asm volatile("" : "+v"(id.j), "+v"(id.g), "+v"(id.iperm));
      const long long row = (long long)tile * TR_ROWS + arow0 + id.j;
      const bool valid = row < n;
      const float wn = valid ? (row_w ? row_w[row] : uni_w) : 0.f;
      const float gld = -wn;                       // d(sum w loss)/d(any logabsdet term)
      float cr[4];
      __syncthreads();                             // weights staged / previous tile's shared reads done
      TS(0);
      // ---- P0: load state, context, upstream gradient.  All loads are issued before the first
      // use (clamped addresses instead of predicated loads), so one HBM round trip covers them.
      {
        const long long xr = (x_rows == n) ? row : (x_rows == 1 ? 0 : row % x_rows);
        const long long rs = valid ? row : 0;
        float zv[4], gv[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int d = id.g + 4 * u;
          const int dc = d < D ? d : 0;
          zv[u] = z_in[rs * D + dc];
          gv[u] = gz_up[rs * D + dc];
          const int c = d < C ? d : 0;
          xv[u] = x[(valid ? xr : 0) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int d = id.g + 4 * u;
          if (d < D) {
            zs[id.j * pl.ZW + d] = valid ? zv[u] : 0.f;
            const float gz = valid ? gv[u] : 0.f;
            gzs[id.j * pl.ZW + d] = is_last ? wn * gz : gz;   // last transform: d/dz_T of w*(0.5|z|^2) = w z
          }
          if (d < C) {
            const float v = ((valid ? xv[u] : 0.f) - x_mean[d]) / x_std[d];
            cs[id.j * pl.CW + d] = v;
          }
          cr[u] = (d < C && C <= 16) ? ((valid ? xv[u] : 0.f) - x_mean[d < C ? d : 0]) / x_std[d < C ? d : 0] : 0.f;
        }
        for (int c = id.g + 16; c < C; c += 4) {   // C > 16: remaining context columns
          float v = valid ? x[xr * C + c] : 0.f;
          cs[id.j * pl.CW + c] = (v - x_mean[c]) / x_std[c];
        }
      }
      wave_lds_fence();
      // ---- P1: reload the block inputs h_0..h_NB the forward pass stashed (register-order slabs,
      // 256-byte coalesced loads) instead of recomputing the hidden stack; issued right after
      // P0's own loads (vmcnt retires in order) so the HBM latency hides under the LULinear backward
      f4 hpre[NB + 1][NSF_HT];
      const float* ast = astash + (((long long)t * ((n + 15) / 16) + (long long)tile * TR_NW + wave) *
                                   NSF_AST_SLOTS(cm ? 0 : NB)) * 1024 + id.lane;
      ast_load(ast, cm ? 1 : 4 * NB, hpre[NB]);   // ctx_mlp: slot 1 = h2, slot 0 = h1 (both post-relu)
#pragma unroll
      for (int b = NB - 1; b >= 0; --b) ast_load(ast, cm ? 0 : 4 * b, hpre[b]);
      // ---- LULinear backward wrt its input (needs no forward values): g_u = L^T gz, g_y = U^T g_u
      if (cm) {   // no LULinear for theta-dim 1: the transform output IS the layer output
        for (int k = id.g; k < D; k += 4) {
          const float gzv = gzs[id.j * pl.ZW + k];
          gys[id.j * pl.ZW + k] = gzv;
          gxs[id.j * pl.ZW + k] = gzv;
          ys[id.j * pl.ZW + k] = zs[id.j * pl.ZW + k];
        }
      } else if (!(pl.ablate & 64)) {
        float v[16], o[4];
        row_to_regs16(gzs + id.j * pl.ZW, D, v);
        dense_mv16<true>(lds + S.l_L, D, v, id.g, o);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
          if (id.g + 4 * ii < D) gus[id.j * pl.ZW + id.g + 4 * ii] = o[ii];
        wave_lds_fence();
        row_to_regs16(gus + id.j * pl.ZW, D, v);
        dense_mv16<true>(lds + S.l_U, D, v, id.g, o);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int k = id.g + 4 * ii;
          if (k < D) {
            gys[id.j * pl.ZW + k] = o[ii];
            gxs[id.j * pl.ZW + k] = o[ii];             // identity dims pass through (transformed dims overwritten)
            ys[id.j * pl.ZW + k] = zs[id.j * pl.ZW + k];
          }
        }
      }
      build_cin(pl, S, par, id, zs, cs, cr, cin);
      const float* cin_row = cin + id.j * pl.CINW + id.g;
      TS(1);

      TS(2);
      // ---- P2: final layer + spline, chunk by chunk; d Wf; g_h = Wf^T g_p
      stage_D(Bst, SA, arow0 + id.j, id, hpre[NB], false);
      if (id.g == 0) Bst[(arow0 + id.j) * SA + pl.H] = 1.f;      // bias column
      f4 gh[NSF_HT];
#pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt) gh[mt] = zero4;
      f4 bt1[NSF_HT], bt2[NSF_HT], bsg[NSF_HT];   // block temporaries, loaded one phase ahead of their use
      // chunk loop with the LAST iteration peeled (it also starts the prefetch of the last block's
      // temporaries; peeling keeps those 48 registers dead during the earlier chunks' splines)
#define CHUNK_BODY(LAST)                                                                                              \
      {                                                                                                               \
          const int d0 = c * DCHB;                                                                                    \
          TS(3 + 4 * c);                                                                                              \
          if (!(pl.ablate & 32)) final_layer_chunk_T<PT, KSH>(lds, Arow, pl, tp, S, id, hpre[NB], d0);                \
          wave_lds_fence();                                                                                           \
          TS(4 + 4 * c);                                                                                              \
          {                                                                                                           \
            const int slot = id.g & 1, part = id.g >> 1;                                                              \
            const int dd = d0 + slot;                                                                                 \
            float* pp = Arow + id.j * SA + slot * tp.PTW;                                                             \
            if (slot < DCHB) {                                                                                        \
              if (dd < S.d_tr && !(pl.ablate & 4)) {                                                                  \
                const int zi = id.j * pl.ZW + 2 * dd + par;                                                           \
                float yv, gxv;                                                                                        \
                rq_spline_pair_bwd<K>(pp, tp.PTW, zs[zi], gys[zi], gld, pl, part, yv, gxv);                           \
                if (part == 0) {                                                                                      \
                  ys[zi] = yv;                                                                                        \
                  gxs[zi] = gxv;                                                                                      \
                }                                                                                                     \
              } else if (part == 0) {                                                                                 \
                for (int k = 0; k < tp.PTW; ++k) pp[k] = 0.f;                                                         \
              }                                                                                                       \
            }                                                                                                         \
          }                                                                                                           \
          TS(5 + 4 * c);                                                                                              \
          __syncthreads();                                                                                            \
          TS(6 + 4 * c);                                                                                              \
          if (LAST && !cm) {                                                                                          \
            ast_load(ast, 1 + 4 * (NB - 1), bt1);                                                                     \
            ast_load(ast, 2 + 4 * (NB - 1), bt2);                                                                     \
            ast_load(ast, 3 + 4 * (NB - 1), bsg);                                                                     \
          }                                                                                                           \
          if (!(pl.ablate & 2)) wft_chunk<PT>(lds, LF, pl, S, id, Arow, SA, d0, gh);                                  \
          __syncthreads();                                                                                            \
      }
      for (int c = 0; c < nch - 1; ++c) CHUNK_BODY(false)
      { const int c = nch - 1; CHUNK_BODY(true) }
#undef CHUNK_BODY

      TS(19);
      // ---- P3 (ctx_mlp): h2 = relu(W_h h1 + b_h): one hidden layer to walk back through
      if (cm) {
        f4 ga[NSF_HT], gb[NSF_HT];
#pragma unroll
        for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) ga[mt][r] = hpre[NB][mt][r] > 0.f ? gh[mt][r] : 0.f;
        stage_D(Ast, SA, arow0 + id.j, id, ga, false);
        stage_D(Bst, SA, arow0 + id.j, id, hpre[0], true);
        if (id.g == 0) Bst[(arow0 + id.j) * SA + pl.H] = 1.f;
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < NSF_HT; ++mt) gb[mt] = zero4;
        gemm_T_breg<KSH, NSF_HT>(lds, S.lin[1], id, ga, gb, pl.ablate);
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) gh[mt][r] = hpre[0][mt][r] > 0.f ? gb[mt][r] : 0.f;
      }
      // ---- P3: residual blocks, last -> first
      if (!cm)
#pragma unroll
      for (int b = NB - 1; b >= 0; --b) {
        f4 ga[NSF_HT], gb[NSF_HT];
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
          // d Wc first (A = g_c, B = standardized context): g_c dies right away
          stage_D(Ast, SA, arow0 + id.j, id, gc, false);
          for (int k = id.g; k < 16 * ntc; k += 4)
            Brow[id.j * SA + k] = k < C ? cs[id.j * pl.CW + k] : (k == C ? 1.f : 0.f);
        }
        if (b > 0) {   // next (earlier) block's t2 / gate: fetch under this block's GEMM phases
          ast_load(ast, 2 + 4 * (b - 1), bt2);
          ast_load(ast, 3 + 4 * (b - 1), bsg);
        }
        __syncthreads();
        __syncthreads();
        TS(20 + 8 * b);
        // d W2 : A = g_t2, B = relu(t1)
        stage_D(Ast, SA, arow0 + id.j, id, ga, false);
        stage_D(Bst, SA, arow0 + id.j, id, bt1, true);
        if (id.g == 0) Bst[(arow0 + id.j) * SA + pl.H] = 1.f;
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < NSF_HT; ++mt) gb[mt] = zero4;
        TS(21 + 8 * b);
        gemm_T_breg<KSH, NSF_HT>(lds, S.lin[3 + 3 * b], id, ga, gb, pl.ablate);       // d relu(t1)
        TS(22 + 8 * b);
        __syncthreads();
        TS(23 + 8 * b);
#pragma unroll
        for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) ga[mt][r] = bt1[mt][r] > 0.f ? gb[mt][r] : 0.f;   // d t1
        if (b > 0) ast_load(ast, 1 + 4 * (b - 1), bt1);
        // d W1 : A = g_t1, B = relu(h_b)
        stage_D(Ast, SA, arow0 + id.j, id, ga, false);
        stage_D(Bst, SA, arow0 + id.j, id, hpre[b], true);
        if (id.g == 0) Bst[(arow0 + id.j) * SA + pl.H] = 1.f;
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < NSF_HT; ++mt) gb[mt] = zero4;
        TS(24 + 8 * b);
        gemm_T_breg<KSH, NSF_HT>(lds, S.lin[2 + 3 * b], id, ga, gb, pl.ablate);       // d relu(h_b)
        TS(25 + 8 * b);
        __syncthreads();
        TS(26 + 8 * b);
#pragma unroll
        for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) gh[mt][r] += hpre[b][mt][r] > 0.f ? gb[mt][r] : 0.f;
      }

      TS(40);
      // ---- P4: initial layer
      stage_D(Ast, SA, arow0 + id.j, id, gh, false);
      for (int k = id.g; k < 16 * nt0; k += 4)
        Brow[id.j * SA + k] = k < S.in0 ? cin[id.j * pl.CINW + k] : (k == S.in0 ? 1.f : 0.f);
      __syncthreads();
      {
        f4 gin[1];
        gin[0] = zero4;
        gemm_T_breg<KSH, 1>(lds, L0, id, gh, gin, pl.ablate);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 4 * r + id.g;     // identity feature slot
          if (k < S.d_id) gxs[id.j * pl.ZW + 2 * k + (1 - par)] += gin[0][r];
        }
      }
      __syncthreads();

      TS(41);
      // ---- P5: LULinear parameter gradients as two more 16x16 tiles
      if (!cm && !(pl.ablate & 64)) {
        float v[16], o[4];
        row_to_regs16(ys + id.j * pl.ZW, D, v);
        dense_mv16<false>(lds + S.l_U, D, v, id.g, o);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
          if (id.g + 4 * ii < D) us[id.j * pl.ZW + id.g + 4 * ii] = o[ii];
      }
      wave_lds_fence();
      for (int k = id.g; k < 16; k += 4) {
        const int o = id.j * pl.ZW + k;
        Arow[id.j * SA + k] = k < D ? gus[o] : (k == D ? gld : 0.f);
        Arow[id.j * SA + 16 + k] = k < D ? gzs[o] : 0.f;
        Brow[id.j * SA + k] = k < D ? ys[o] : (k == D ? 1.f : 0.f);
        Brow[id.j * SA + 16 + k] = k < D ? us[o] : (k == D ? 1.f : 0.f);
      }
      __syncthreads();

      TS(42);
      // ---- P6: gradient wrt this transform's input
      for (int d = id.g; d < D; d += 4) {
        if (valid) {
          const float g = gxs[id.j * pl.ZW + d];
          if (t > 0) gz_dn[row * D + d] = g;
          else if (grad_theta) grad_theta[row * D + d] = g * zstats[D + d];
        }
      }
      TS(43);
    }

  } else {
    // =========================== grad waves ==========================
    const int gw = wave - TR_NW;
    // weight-gradient accumulators owned by this wave (m-tile = wave) for the whole launch
    f4 acc0[2], accC[NB][2], acc1[NB][4], acc2[NB][4], accF[NCH][4], accLU[1];
  #pragma unroll
    for (int i = 0; i < 2; ++i) acc0[i] = zero4;
  #pragma unroll
    for (int b = 0; b < NB; ++b) {
  #pragma unroll
      for (int i = 0; i < 2; ++i) accC[b][i] = zero4;
  #pragma unroll
      for (int i = 0; i < 4; ++i) { acc1[b][i] = zero4; acc2[b][i] = zero4; }
    }
  #pragma unroll
    for (int c = 0; c < NCH; ++c)
  #pragma unroll
      for (int i = 0; i < 4; ++i) accF[c][i] = zero4;
    accLU[0] = zero4;


    const LaneId id0 = id;
    for (int tile = blockIdx.x; tile < tp.ntiles; tile += gridDim.x) {
      LaneId id = id0;
      asm volatile("" : "+v"(id.j), "+v"(id.g));
      // mirrors the row waves' barrier sequence exactly
      __syncthreads();
      TS(0);
  #pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c < nch) {
          __syncthreads();
          TS(6 + 4 * c);
          dw_gemm<4>(Ast, Bst, SA, 16 * gw, 0, id, accF[c], 4, pl.ablate);
          TS(7 + 4 * c);
          __syncthreads();
        }
      }
      if (cm) {
        __syncthreads();
        dw_gemm<4>(Ast, Bst, SA, 16 * gw, 0, id, acc1[0], 4, pl.ablate);
        __syncthreads();
      }
      if (!cm)
  #pragma unroll
      for (int b = NB - 1; b >= 0; --b) {
        __syncthreads();
        dw_gemm<2>(Ast, Bst, SA, 16 * gw, 0, id, accC[b], ntc, pl.ablate);
        __syncthreads();
        __syncthreads();
        TS(21 + 8 * b);
        dw_gemm<4>(Ast, Bst, SA, 16 * gw, 0, id, acc2[b], 4, pl.ablate);
        TS(22 + 8 * b);
        __syncthreads();
        __syncthreads();
        TS(24 + 8 * b);
        dw_gemm<4>(Ast, Bst, SA, 16 * gw, 0, id, acc1[b], 4, pl.ablate);
        TS(25 + 8 * b);
        __syncthreads();
      }
      __syncthreads();
      dw_gemm<2>(Ast, Bst, SA, 16 * gw, 0, id, acc0, nt0, pl.ablate);
      __syncthreads();
      __syncthreads();
      if (gw < 2 && !cm) dw_gemm<1>(Ast, Bst, SA, 16 * gw, 16 * gw, id, accLU);
    }

    // ---- write this workgroup's partial gradients (natural parameter order)
    float* part = partial + ((long long)t * gridDim.x + blockIdx.x) * tp.PLP;
    const int out0 = 16 * gw;
  #pragma unroll
    for (int nt = 0; nt < 2; ++nt) write_tile(part, L0, out0, nt, id, acc0[nt]);
    if (cm) {
  #pragma unroll
      for (int nt = 0; nt < 4; ++nt) write_tile(part, S.lin[1], out0, nt, id, acc1[0][nt]);   // hidden H->H layer
    } else {
  #pragma unroll
      for (int b = 0; b < NB; ++b) {
  #pragma unroll
        for (int nt = 0; nt < 2; ++nt) write_tile(part, S.lin[1 + 3 * b], out0, nt, id, accC[b][nt]);
  #pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          write_tile(part, S.lin[2 + 3 * b], out0, nt, id, acc1[b][nt]);
          write_tile(part, S.lin[3 + 3 * b], out0, nt, id, acc2[b][nt]);
        }
      }
    }
  #pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c < nch) {
        const int dd = c * DCHB + gw / PT;
        const int pt = gw % PT;
        if (gw < DCHB * PT && dd < S.d_tr) {
  #pragma unroll
          for (int nt = 0; nt < 4; ++nt)
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int p = 16 * pt + 4 * id.g + r;
              const int in = 16 * nt + id.j;
              if (p < pl.P) {
                const int out = dd * pl.P + p;
                if (in < LF.in) part[LF.g_w + out * LF.in + in] = accF[c][nt][r];
                else if (in == LF.in) part[LF.g_b + out] = accF[c][nt][r];
              }
            }
        }
      }
    }
    if (!cm) {
      const int ntri = D * (D - 1) / 2;
      float* plow = part + S.g_lu;
      float* pup = plow + ntri;
      float* pdiag = pup + ntri;
      float* pbias = pdiag + D;
  #pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * id.g + r, k = id.j;
        const float v = accLU[0][r];
        if (gw == 0) {
          if (i < D && k < D) {
            if (k > i) pup[i * D - i * (i + 1) / 2 + (k - i - 1)] = v;
            else if (k == i) pdiag[i] = v;          // dL/dU_ii; chain rule finished in the reduction
          } else if (i == D && k == D) part[S.n_params] = v;   // sum_n d/d(logabsdet)
        } else if (gw == 1) {
          if (i < D) {
            if (k < i) plow[i * (i - 1) / 2 + k] = v;
            else if (k == D) pbias[i] = v;
          }
        }
      }
    }

  }
}


// ---- launch helpers
template <int K, int KSH, int NB, int NCH>
static int launch_bwd(const NsfPlan& pl, const TrainPlan& tp, int t, const float* packed, const float* zstats,
                      const float* z_in, const float* x, const float* gz_up, const float* row_w, float uni_w,
                      int64_t n, int64_t x_rows, float* gz_dn, float* partial, float* grad_theta,
                      const float* astash, long long* dbg, hipStream_t st) {
  auto kern = nsf_bwd_layer_kernel<K, KSH, NB, NCH>;
  const int lds_bytes = 4 * tp.lds_floats;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(tp.grid), dim3(128 * TR_NW), (size_t)lds_bytes, st, pl, tp, t, packed, zstats, z_in, x,
                     gz_up, row_w, uni_w, (long long)n, (long long)x_rows, gz_dn, partial, grad_theta, astash, dbg);
  return (int)hipGetLastError();
}

template <int K>
int launch_bwd_k(const NsfPlan& pl, const TrainPlan& tp, int t, const float* packed, const float* zstats,
                        const float* z_in, const float* x, const float* gz_up, const float* row_w, float uni_w,
                        int64_t n, int64_t x_rows, float* gz_dn, float* partial, float* grad_theta,
                        const float* astash, long long* dbg, hipStream_t st) {
#define BWD_ARGS pl, tp, t, packed, zstats, z_in, x, gz_up, row_w, uni_w, n, x_rows, gz_dn, partial, grad_theta, \
                 astash, dbg, st
  const int nchmax = tp.nch[0] > tp.nch[1] ? tp.nch[0] : tp.nch[1];
#define BWD_NCH(KS, NBV) \
  switch (nchmax) { \
    case 1: return launch_bwd<K, KS, NBV, 1>(BWD_ARGS); \
    case 2: case 3: return launch_bwd<K, KS, NBV, 3>(BWD_ARGS); \
    default: return launch_bwd<K, KS, NBV, 4>(BWD_ARGS); \
  }
  if (pl.ctx_mlp) {   // theta-dim 1: one transformed dim => one chunk
    if (pl.KSH == 13) return launch_bwd<K, 13, 0, 1>(BWD_ARGS);
    return launch_bwd<K, 16, 0, 1>(BWD_ARGS);
  }
  if (pl.KSH == 13) { if (pl.NB <= 1) { BWD_NCH(13, 1) } else { BWD_NCH(13, 2) } }
  if (pl.NB <= 1) { BWD_NCH(16, 1) } else { BWD_NCH(16, 2) }
#undef BWD_NCH
#undef BWD_ARGS
}

