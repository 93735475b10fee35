#pragma once
// nsf_train_kernel.h -- NPE training pass for the NSF estimator on gfx950:
//   loss_n = -log p(theta_n | x_n)                       (NFlowsFlow.loss, nflows_flow.py:99-109)
//   grad   = d( sum_n w_n loss_n ) / d params            (what loss.mean().backward() produces,
//                                                         trainers/base.py:1178-1181)
// Structure (DESIGN.md "training pass"):
//   1. forward flow kernel with the per-transform input state stashed (T*N*D floats);
//   2. one backward launch per transform, last -> first.  Persistent workgroups walk
//      64-row tiles with 4 "row" waves (spline forward + reverse mode on the VALU, the
//      row-wise backward through the residual blocks on MFMA with the transposed weight
//      image) and 4 "grad" waves (final-layer recompute, Wf^T g, and a fixed quarter of
//      every weight-gradient tile, all on MFMA); the two kinds share a SIMD pairwise and
//      exchange (activation, gradient) tiles through LDS, one barrier per phase;
//   3. a deterministic reduction of the per-workgroup partial gradients (no atomics).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "nsf_device.h"
#include "nsf_plan_layout.h"

#define TR_NW 4            // row waves per workgroup (+ as many grad waves)
#define TR_ROWS 64         // rows per tile
#define TR_MAXCH 4         // spline chunks per transform the accumulators are sized for
#define TR_GRID_MAX 256    // persistent workgroups (one per CU)
#define TR_SA 68           // row stride of the gradient tiles A0/A1
#define TR_SB 68           // row stride of the activation tile B (column-interleaved, see stage_DB)
// Column where dim slot 1 of a spline-parameter tile row starts: one float past the first slot's 16*PT, so that the
// four (slot, widths | heights) lane classes of the spline phase fall into four different bank residues mod 4 for the
// default 10 bins (rows are 4 banks apart: stride 68) -- it was a 2-way conflict on every parameter access.
#define TR_SLOT(PT) (16 * (PT) + 1)
#define TR_LDK_FAST 50     // image row stride of the default hidden_features = 50: compiled-in fast path
#ifndef TR_LA
#define TR_LA 2            // K-steps of operand lookahead in the LDS-fed MFMA loops
#endif
// A/B switches of round 6 (defaults = what measured best; -D... in SBI_AMD_EXTRA_HIPCC_FLAGS rebuilds a variant)
#ifndef TR_SPLINE_PRIO
#define TR_SPLINE_PRIO 0   // s_setprio of the row waves while they run the (VALU-only, latency-bound) spline chunks
#endif
#ifndef TR_H_SPLIT
#define TR_H_SPLIT 0       // hand g_h to the row waves BEFORE the last chunk's d Wf (one more barrier, ~2.5 k cycles less wait)
#endif
#ifndef TR_PSTASH
#define TR_PSTASH 1        // the forward pass stashes the spline parameters, the backward pass does not recompute the final layer
#endif
#ifndef TR_NO_S0
#define TR_NO_S0 1         // no workgroup barrier between tiles: the grad waves rendezvous among themselves instead
#endif

struct TrainPlan {
  int SA, SB, SS;          // row strides: gradient tiles A0/A1, activation tile B, static input tile Bs
  int o_A0, o_A1, o_B, o_Bs, o_cnt;   // LDS float offsets ([64][stride] tiles; grad-wave sync counter)
  int o_xs;                // [32] x mean, [32] 1 / x std (context z-scoring constants)
  int o_wave, w_total;     // per-row-wave scratch base / size
  int w_zs, w_gys, w_gzs;  // 16 x ZW each: state (z -> y in place), gradient (g_y -> g_x in place), upstream g_z
  int DCHB, PTW;           // spline dims per chunk, floats per dim slot (16*PT)
  int nch[2];              // chunks per mask parity
  int PLP;                 // floats per (workgroup, transform) partial-gradient slab
  int grid, ntiles;
  int lds_floats;
  int overlay;             // 1: the weight region holds [final layer + LU] during the chunk steps and the hidden
                           //    layers during the block phase (re-staged per tile) because both do not fit at once
  float* grad_x;           // optional output (n, C): d(sum w loss)/d x, accumulated over the transforms' launches
};

// Tile layouts.  The grad waves read K = row down a tile (row = 4s + g, column = c0 + j); the row
// waves write/read D fragments (row = j, column = 16mt + 4r + g).
//   A0/A1 (gradient side): row major, stride 68: conflict-free for the row waves (they also run the
//     spline on these rows); the grad waves' one A read per K-step takes the 4-way conflict.
//   B (activation side): column-interleaved, physical column = 4*(c & 15) + (c >> 4), stride 68: the four
//     n-tiles' operands of a lane are adjacent => ONE ds_read_b128 per K-step (and one ds_write_b128 per
//     register row when staging); LDS-fed MFMA loops are limited by the number of LDS instructions.
//   Bs (static conditioner input): row major, stride 48 (= 48 mod 64: the g groups hit disjoint banks).
// n-independent part (tile offsets, strides): constexpr, see nsf_static_plan.h
constexpr int build_train_layout(const NsfPlan& pl, TrainPlan* tp) {
  // hidden_features == 64 has no spare activation-tile column for the bias trick: the kernel then takes the bias
  // gradients from one extra MFMA per K-step against a ones vector (template flag HB)
  if (pl.D > 15 || pl.H > 64 || pl.NB > 2 || (pl.NB < 1 && !pl.ctx_mlp)) return SBI_AMD_E_UNSUPPORTED;
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
  tp->SA = TR_SA;
  tp->SB = TR_SB;
  {   // Bs row = [z_id ; context ; 1 ; 0...]: d W0 reads columns [0, 16 nt0), d Wc columns [d_id, d_id + 16 ntc).
    const int in0max = d_id_max + pl.C;
    const int nt0 = (in0max + 1 + 15) / 16, ntc = (pl.C + 1 + 15) / 16;
    int need = 16 * nt0 > d_id_max + 16 * ntc ? 16 * nt0 : d_id_max + 16 * ntc;
    // 4 * odd: the row waves' row-major writes (row j, column g) and the grad waves' reads of rows krow + 4 g
    // (4 * SS = 16 * odd mod 64) are both bank-conflict free
    tp->SS = (need + 3) / 4 * 4;
    if ((tp->SS / 4) % 2 == 0) tp->SS += 4;
  }
  int w = 0;
  tp->w_zs = w; w += 16 * pl.ZW;
  tp->w_gys = w; w += 16 * pl.ZW;
  tp->w_gzs = w; w += 16 * pl.ZW;
  tp->w_total = (w + 16 + 3) / 4 * 4;  // + slack: a 16-wide read of the LAST row of the LAST array runs 16 - ZW
                                       // floats past it; unwritten LDS there could hold NaN (NaN x 0 = NaN)
  for (int ov = 0; ov < 2; ++ov) {
    // ov = 0: the whole training image (explicit LU inverses at its tail excluded) stays resident;
    // ov = 1: one region, sized for the larger of [hidden layers] and [final layer + LU], re-staged per tile
    const int fin_min = pl.shape[0].final_off < pl.shape[1].final_off ? pl.shape[0].final_off : pl.shape[1].final_off;
    int o = ov ? (pl.hidden_img_floats > pl.lds_w_train_floats - fin_min ? pl.hidden_img_floats
                                                                         : pl.lds_w_train_floats - fin_min)
               : pl.lds_w_train_floats;
    o = (o + 3) / 4 * 4;
    tp->overlay = ov;
    tp->o_A0 = o; o += TR_ROWS * tp->SA;
    tp->o_A1 = o; o += TR_ROWS * tp->SA;
    tp->o_B = o; o += TR_ROWS * tp->SB;
    tp->o_Bs = o; o += TR_ROWS * tp->SS;
    tp->o_cnt = o; o += 4;
    tp->o_xs = o; o += 64;
    tp->o_wave = o;
    tp->lds_floats = tp->o_wave + TR_NW * tp->w_total;
    if (4ll * tp->lds_floats <= NSF_LDS_LIMIT_BYTES) break;
  }
  if (4ll * tp->lds_floats > NSF_LDS_LIMIT_BYTES) return SBI_AMD_E_LDS;
  int pmax = pl.shape[0].n_params > pl.shape[1].n_params ? pl.shape[0].n_params : pl.shape[1].n_params;
  tp->PLP = (pmax + 1 + 3) / 4 * 4;
  tp->grad_x = nullptr;
  tp->ntiles = 0;
  tp->grid = 0;
  return 0;
}
static int build_train_plan(const NsfPlan& pl, int64_t n, TrainPlan* tp) {
  *tp = TrainPlan{};
  const int rc = build_train_layout(pl, tp);
  if (rc) return rc;
  tp->ntiles = (int)((n + TR_ROWS - 1) / TR_ROWS);
  tp->grid = tp->ntiles < TR_GRID_MAX ? tp->ntiles : TR_GRID_MAX;
  if (NSF_DBG_ABL(pl.ablate, 512) && tp->grid > 4) tp->grid = 4;   // debug aid: many tiles per persistent workgroup at small n
  return 0;
}

// ---- the benchmark configuration (BASELINE configs[1]: sbi's NSF defaults, theta-dim = x-dim = 10) as compile-time
// constants.  A kernel that receives its plan as a 1.3 KB kernel argument re-reads the fields it needs from the scalar
// cache all through its tile loop (98 s_load per tile and wave in nsf_bwd_layer_kernel, each followed by an
// lgkmcnt wait that also drains the LDS reads feeding the MFMA loops: profiles/r3_pmc_bwd_smem.txt); the SP
// instantiations of the backward kernel take the integer layout from these constants instead -- every offset an
// immediate -- and only the floating-point constants and the batch-dependent fields from the argument.
constexpr sbi_amd_nsf_config kNsfDefaultCfg = {10, 10, 50, 10, 5, 2, 3.0f, 1e-3f, 1e-3f, 1e-3f, 1e-3f};
constexpr NsfPlan nsf_make_static_plan() {
  NsfPlan p{};
  nsf_build_layout(&kNsfDefaultCfg, TR_NW, &p);
  return p;
}
constexpr NsfPlan kStaticPl = nsf_make_static_plan();
constexpr TrainPlan nsf_make_static_train() {
  TrainPlan t{};
  build_train_layout(kStaticPl, &t);
  return t;
}
constexpr TrainPlan kStaticTp = nsf_make_static_train();
static_assert(kStaticPl.n_params == 98025 && kStaticTp.overlay == 0 && kStaticTp.nch[0] == 3, "default NSF layout");

// does a run-time plan have exactly the static layout?  (floats and debug switches are taken from the argument anyway)
static bool plan_is_static_default(const NsfPlan& pl, const TrainPlan& tp) {
  NsfPlan a = pl;
  a.B = a.min_w = a.min_h = a.min_d = a.lu_eps = a.sqrt_h = a.inv_sqrt_h = 0.f;
  a.one_minus_kw = a.one_minus_kh = a.d_const = a.log_z = 0.f;
  a.ablate = 0;
  const NsfPlan b = kStaticPl;
  if (memcmp(&a, &b, sizeof(NsfPlan)) != 0) return false;
  TrainPlan c = tp;
  c.grid = c.ntiles = 0;
  c.grad_x = nullptr;
  const TrainPlan d = kStaticTp;
  return memcmp(&c, &d, sizeof(TrainPlan)) == 0 && NSF_DBG_ABL(pl.ablate, 0x40000) == 0;   // bit 0x40000: force the dynamic plan
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

// column-interleaved activation tile: logical column c lives at 4*(c & 15) + (c >> 4)
__device__ __forceinline__ int il_col(int c) { return 4 * (c & 15) + (c >> 4); }
__device__ __forceinline__ void stage_DB(float* __restrict__ st, int SB, int row, const LaneId& id,
                                         const f4 (&v)[NSF_HT], bool relu) {
  static_assert(NSF_HT == 4, "one float4 per register row");
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f4 w = {v[0][r], v[1][r], v[2][r], v[3][r]};
    if (relu) w = {fmaxf(w[0], 0.f), fmaxf(w[1], 0.f), fmaxf(w[2], 0.f), fmaxf(w[3], 0.f)};
    *(f4*)(st + row * SB + 4 * (4 * r + id.g)) = w;
  }
}

// tile row of k-slot 0 in K-step s of the weight-gradient GEMMs (k-slot g adds 4 g): 0,1,2,3,16,17,18,19,32,...
__device__ __forceinline__ constexpr int dw_krow(int s) { return 16 * (s >> 2) + (s & 3); }

// weight-gradient tile(s): acc[nt] += sum_{rows of the 64-row tile} A[row][acol0+i] * B[row][bcol0+16nt+j]
// Strides are compile time (every LDS read is base + immediate) and the operands of K-step s+2 are
// requested before the MFMAs of step s issue, so the LDS latency hides under the matrix pipe.
template <int NT, int SA, int SB, bool IL = false>
__device__ __forceinline__ void dw_gemm(const float* __restrict__ Ast, const float* __restrict__ Bst,
                                        int acol0, int bcol0, const LaneId& id, f4 (&acc)[NT], int nt_on = NT,
                                        int abl = 0, f4* accb = nullptr) {
  if (abl & 1) return;
  const float ones_b = id.j == 0 ? 1.f : 0.f;   // B operand of the bias-gradient MFMA: column 0 = sum over the rows
  // few MFMAs per K-step (NT < 4) cover less LDS latency per step: look further ahead
  constexpr int KS = TR_ROWS / 4, LA = NT >= 4 ? TR_LA : 2 * TR_LA;
  // K-step s covers tile rows krow(s) + 4 g (not 4 s + g): with the gradient tiles' row stride 68 (= 4 mod 64,
  // what keeps the ROW waves' accesses conflict-free) the four k-slots of an A read then sit 16 banks apart
  // (4 * 68 = 16 mod 64) instead of 4, so the read is conflict-free as well (it was a 4-way conflict).
  const float* ap = Ast + 4 * id.g * SA + acol0 + id.j;
  // interleaved B (bcol0 a multiple of 16): the NT operands of this lane are adjacent floats
  const float* bp = IL ? Bst + 4 * id.g * SB + 4 * id.j + (bcol0 >> 4) : Bst + 4 * id.g * SB + bcol0 + id.j;
  float a[LA + 1], b[LA + 1][NT];
  auto load_b = [&](int u, int s) {
    if (IL && NT == 4) {
      const f4 w = *(const f4*)(bp + dw_krow(s) * SB);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[u][nt] = w[nt];
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[u][nt] = bp[dw_krow(s) * SB + (IL ? nt : 16 * nt)];
    }
  };
#pragma unroll
  for (int u = 0; u < LA; ++u) {
    a[u] = ap[dw_krow(u) * SA];
    load_b(u, u);
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + LA < KS) {
      a[(s + LA) % (LA + 1)] = ap[dw_krow(s + LA) * SA];
      load_b((s + LA) % (LA + 1), s + LA);
    }
    // pin the order: hipcc otherwise sinks every load next to its MFMA (load, wait, 2 MFMAs, load, ...)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)   // NT == 1: no guard (a guard per MFMA costs a basic block and an s_waitcnt each)
      if (NT == 1 || nt < nt_on) acc[nt] = MFMA16(a[s % (LA + 1)], b[s % (LA + 1)][nt], acc[nt]);
    if (accb) *accb = MFMA16(a[s % (LA + 1)], ones_b, *accb);   // compile-time null in the H < 64 instantiations
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the same with a run-time activation-side stride (the small static tile Bs: d W0, d Wc)
template <int NT, int SA>
__device__ __forceinline__ void dw_gemm_rs(const float* __restrict__ Ast, const float* __restrict__ Bst, int SBr,
                                           int acol0, int bcol0, const LaneId& id, f4 (&acc)[NT], int nt_on = NT,
                                           int abl = 0) {
  if (abl & 1) return;
  constexpr int KS = TR_ROWS / 4, LA = NT >= 4 ? TR_LA : 2 * TR_LA;
  const float* ap = Ast + 4 * id.g * SA + acol0 + id.j;     // rows krow(s) + 4 g, see dw_gemm
  const float* bp = Bst + 4 * id.g * SBr + bcol0 + id.j;
  float a[LA + 1], b[LA + 1][NT];
#pragma unroll
  for (int u = 0; u < LA; ++u) {
    a[u] = ap[dw_krow(u) * SA];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[u][nt] = bp[dw_krow(u) * SBr + 16 * nt];
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + LA < KS) {
      a[(s + LA) % (LA + 1)] = ap[dw_krow(s + LA) * SA];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[(s + LA) % (LA + 1)][nt] = bp[dw_krow(s + LA) * SBr + 16 * nt];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)   // NT == 1: no guard (a guard per MFMA costs a basic block and an s_waitcnt each)
      if (NT == 1 || nt < nt_on) acc[nt] = MFMA16(a[s % (LA + 1)], b[s % (LA + 1)][nt], acc[nt]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// row-major tile -> D fragments (inverse of stage_D)
__device__ __forceinline__ void load_D(const float* __restrict__ st, int SA, int row, const LaneId& id,
                                       f4 (&v)[NSF_HT]) {
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[mt][r] = st[row * SA + 16 * mt + 4 * r + id.g];
}

// Rendezvous of the four grad waves only (the row waves are busy in their own phase): LDS counter,
// monotonically increasing; `target` = 4 x (number of rendezvous so far).  DS ops of a wave execute
// in order, so the tile reads issued before the increment have completed when it lands.
__device__ __forceinline__ void grad_wave_sync(int* cnt, int target, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// row-wise backward through a linear layer: acc[mt] += sum_k W[k][feat(mt)] * g[k], g = D fragments.
// The image keeps rows [out, 4*KS) zero (nsf_plan.cpp: rows_alloc), so K-steps past `out` need no
// predicate; lanes that supply an A row for an in-feature slot >= in load column 0 instead (finite) and
// the corresponding output slots are cleared after the loop.
template <int KS, int MT, int LDK>
__device__ __forceinline__ void gemm_T_breg_ldk(const float* __restrict__ lds, const LinDesc& L, const LaneId& id,
                                                const f4 (&gb)[NSF_HT], f4 (&acc)[MT]) {
  const int ldk = LDK ? LDK : L.ldk;   // compile-time row stride: immediate offsets, two K-steps per ds_read2_b32
  const float* base[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int f = 16 * mt + id.iperm;
    base[mt] = lds + L.l_w + id.g * ldk + (f < L.in ? f : 0);
  }
  const int kstride = 4 * ldk;
  // K-steps go in pairs: the loads of pair p+1 are requested before the MFMAs of pair p, and with a
  // compile-time stride the two steps of a pair share one ds_read2_b32 per m-tile
  constexpr int NP = (KS + 1) / 2;
  float a[2][2][MT];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[0][u][mt] = u < KS ? base[mt][u * kstride] : 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (p + 1 < NP) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          a[(p + 1) & 1][u][mt] = (2 * p + 2 + u < KS) ? base[mt][(2 * p + 2 + u) * kstride] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int s = 2 * p + u;
      if (s < KS) {
        const float bv = gb[s >> 2][s & 3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = MFMA16(a[p & 1][u][mt], bv, acc[mt]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // in-feature slots >= in accumulated a (finite) neighbouring weight column: clear them on the way out
  // (16 selects per call instead of one select + hazard nop in front of every MFMA)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[mt][r] = (16 * mt + 4 * r + id.g < L.in) ? acc[mt][r] : 0.f;
}
template <int KS, int MT>
__device__ __forceinline__ void gemm_T_breg(const float* __restrict__ lds, const LinDesc& L, const LaneId& id,
                                            const f4 (&gb)[NSF_HT], f4 (&acc)[MT], int abl = 0) {
  if (abl & 2) return;
  if (L.ldk == TR_LDK_FAST) gemm_T_breg_ldk<KS, MT, TR_LDK_FAST>(lds, L, id, gb, acc);   // wave-uniform
  else gemm_T_breg_ldk<KS, MT, 0>(lds, L, id, gb, acc);
}

// d loss / d context contribution of one linear layer whose input contains the context at columns
// [col0, col0 + C) (Wc: col0 = 0; W0: col0 = d_id): part = W[:, col0 + c]^T g for this wave's 16 rows, produced in
// CONTEXT coordinates (slot = c) so that every contribution to grad_x[row][c] is made by the same lane and the
// read-modify-writes of one launch are ordered by program order.  Only runs when the caller asked for grad_x.
template <int KS>
__device__ __forceinline__ void ctx_grad_update(const float* __restrict__ lds, const LinDesc& L, const LaneId& id,
                                                const f4 (&gb)[NSF_HT], int col0, int C, const float* __restrict__ xs,
                                                float* __restrict__ grad_x_row, bool valid, bool overwrite) {
  const float* base[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int c = 16 * mt + id.iperm;
    base[mt] = lds + L.l_w + id.g * L.ldk + (c < C ? col0 + c : 0);
  }
  const int kstride = 4 * L.ldk;
  f4 part[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float bv = gb[s >> 2][s & 3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) part[mt] = MFMA16(base[mt][s * kstride], bv, part[mt]);
  }
  if (!valid) return;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * mt + 4 * r + id.g;
      if (c < C) {
        const float v = part[mt][r] * xs[32 + c];     // through the kernel's own z-scoring: d c / d x = 1 / std
        grad_x_row[c] = overwrite ? v : grad_x_row[c] + v;
      }
    }
}

// final_layer output for the backward chunk -> this wave's rows of the shared A tile.  NACT = live dim slots of the
// chunk (an odd number of transformed dims leaves the last chunk half empty: its dead slot is neither computed --
// 26 of the 52 MFMAs at the defaults -- nor stored; the row wave zero-fills that slot's rows itself).
template <int PT, int KSH, int NACT>
__device__ __forceinline__ void final_layer_chunk_T_n(const float* __restrict__ lds, float* __restrict__ arow,
                                                      const NsfPlan& pl, const TrainPlan& tp, const ShapeDesc& S,
                                                      const LaneId& id, const f4 (&h)[NSF_HT], int d0) {
  const LinDesc& L = S.lin[S.fin];
  f4 acc[NACT][PT];
  int ro[NACT][PT];
#pragma unroll
  for (int sl = 0; sl < NACT; ++sl) {
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
  constexpr int LA = TR_LA;
  float a[LA + 1][NACT][PT];
#pragma unroll
  for (int u = 0; u < LA; ++u)
#pragma unroll
    for (int sl = 0; sl < NACT; ++sl)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) a[u][sl][pt] = lds[ro[sl][pt] + 4 * u];
#pragma unroll
  for (int s = 0; s < KSH; ++s) {
    if (s + LA < KSH) {
#pragma unroll
      for (int sl = 0; sl < NACT; ++sl)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) a[(s + LA) % (LA + 1)][sl][pt] = lds[ro[sl][pt] + 4 * (s + LA)];
    }
    const float bv = h[s >> 2][s & 3];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sl = 0; sl < NACT; ++sl)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[sl][pt] = MFMA16(a[s % (LA + 1)][sl][pt], bv, acc[sl][pt]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int sl = 0; sl < NACT; ++sl)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int r = 0; r < 4; ++r) arow[id.j * tp.SA + sl * TR_SLOT(PT) + 16 * pt + 4 * r + id.g] = acc[sl][pt][r];
}
template <int PT, int KSH>
__device__ __forceinline__ void final_layer_chunk_T(const float* __restrict__ lds, float* __restrict__ arow,
                                                    const NsfPlan& pl, const TrainPlan& tp, const ShapeDesc& S,
                                                    const LaneId& id, const f4 (&h)[NSF_HT], int d0) {
  constexpr int DCHB = (4 / PT) > 2 ? 2 : (4 / PT);
  if (DCHB == 2 && d0 + 1 >= S.d_tr)      // wave-uniform (a compile-time constant in the static-plan instantiations)
    final_layer_chunk_T_n<PT, KSH, 1>(lds, arow, pl, tp, S, id, h, d0);
  else
    final_layer_chunk_T_n<PT, KSH, DCHB>(lds, arow, pl, tp, S, id, h, d0);
}

// RQ spline forward + reverse-mode gradient for one (row, dim) task on a lane pair (see
// rq_spline_pair).  `p` holds the 3K-1 raw conditioner outputs on entry and
// d(gy*y + gl*logabsdet)/d(raw outputs) on exit (entries [3K-1, plen) zeroed): part 0 writes the
// width logits and all derivative slots but one, part 1 the height logits, its own derivative
// slot and the padding.  Formulas: DESIGN.md "spline backward".
template <int K, int VAR = 0, class PL = NsfPlan>
__device__ __forceinline__ void rq_spline_pair_bwd(float* __restrict__ p, int plen, float x, float gy, float gl,
                                                   const PL& pl, int part, float& y, float& gx) {
  const float B = pl.B;
  SplineSide<K> S;
  spline_side<K, NoYield, VAR>(p + part * K, pl, part, S);
  SplineSel o;
  spline_select<K, false, NoYield, VAR>(p, x, pl, part, S, o);
  const bool inside = o.inside;
  const int idx = o.idx;
  const float d_i = o.d_i, d_n = o.d_n;
  // ---- forward
  const float w = o.w_i, h = o.h_i;
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
  for (int m = 0; m < K; ++m) q_out[m] = S.e[m] * (gsm[m] - dot) * spline_logit_grad<VAR>(q_out[m], pl);
  // ---- derivative slots: knot kd = idx + part is mine (interior knots only)
  const int kd = idx + part;
  float dact;   // d slope / d raw parameter
  if (VAR == 0) {
    dact = sigmoid_f(o.ud_mine);
  } else {
    const float r = rcp_f(1.f + fabsf(o.ud_mine) * ZUKO_CD);
    dact = (part ? d_n : d_i) * r * r;       // slope = exp(clip(u)): slope * clip'(u)
  }
  const float gud = (inside && kd >= 1 && kd <= K - 1) ? (part ? gdn : gdi) * dact : 0.f;
  // Branch-free write-out (it was a ladder of K exec-masked single stores): both lanes hold both knots' gradients after
  // one swap -- slot idx - 1 (knot idx) <- g_lo, slot idx (knot idx + 1) <- g_hi, every other slot 0 -- and split the
  // K - 1 slots plus the padding [3K - 1, plen) between them as unconditional selects.
  const float gud_o = xchg32(gud);
  const float gk_lo = part ? gud_o : gud, gk_hi = part ? gud : gud_o;
  constexpr int KH = K / 2;                   // part 0: slots [0, KH), part 1: slots [KH, K - 1) and the padding
  float* ds_out = p + 2 * K;
  if (part == 0) {
#pragma unroll
    for (int k = 0; k < KH; ++k) ds_out[k] = (k + 1 == idx) ? gk_lo : ((k == idx) ? gk_hi : 0.f);
  } else {
#pragma unroll
    for (int k = KH; k < K - 1; ++k) ds_out[k] = (k + 1 == idx) ? gk_lo : ((k == idx) ? gk_hi : 0.f);
    for (int k = 3 * K - 1; k < plen; ++k) p[k] = 0.f;
  }
}

// g_h += Wf[rows of the chunk's dims]^T g_p for this wave's 16 rows (B operand = the g_p this wave
// just wrote into the shared A tile).  Select-free like gemm_T_breg: the padding K slots (p >= P)
// carry exact-zero g_p, so whatever finite weight they meet is harmless; output slots of in-features
// >= H are cleared after the loop.
template <int PT, int LDK>
__device__ __forceinline__ void wft_chunk_ldk(const float* __restrict__ lds, const LinDesc& LF, const NsfPlan& pl,
                                              const ShapeDesc& S, const LaneId& id, const float* __restrict__ Arow,
                                              int SA, int d0, f4 (&gh)[NSF_HT]) {
  constexpr int DCHB = (4 / PT) > 2 ? 2 : (4 / PT);
  constexpr int KS = 4 * PT;
  int col[NSF_HT];
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt) {
    const int f = 16 * mt + id.iperm;
    col[mt] = f < LF.in ? f : 0;
  }
  const int ldk = LDK ? LDK : LF.ldk;
  const int kstride = 4 * ldk;
#pragma unroll
  for (int sl = 0; sl < DCHB; ++sl) {
    const int dd = d0 + sl;
    if (dd < S.d_tr) {
      const float* wrow = lds + LF.l_w + (dd * pl.P + id.g) * ldk;
      const float* brow = Arow + id.j * SA + sl * TR_SLOT(PT) + id.g;
      const float* wcol[NSF_HT];
#pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt) wcol[mt] = wrow + col[mt];
      constexpr int NP = KS / 2;   // KS = 4 PT is even
      float a[2][2][NSF_HT], bb[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bb[0][u] = brow[4 * u];
#pragma unroll
        for (int mt = 0; mt < NSF_HT; ++mt) a[0][u][mt] = wcol[mt][u * kstride];
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (p + 1 < NP) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            bb[(p + 1) & 1][u] = brow[4 * (2 * p + 2 + u)];
#pragma unroll
            for (int mt = 0; mt < NSF_HT; ++mt) a[(p + 1) & 1][u][mt] = wcol[mt][(2 * p + 2 + u) * kstride];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int mt = 0; mt < NSF_HT; ++mt) gh[mt] = MFMA16(a[p & 1][u][mt], bb[p & 1][u], gh[mt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) gh[mt][r] = (16 * mt + 4 * r + id.g < LF.in) ? gh[mt][r] : 0.f;
}
template <int PT>
__device__ __forceinline__ void wft_chunk(const float* __restrict__ lds, const LinDesc& LF, const NsfPlan& pl,
                                          const ShapeDesc& S, const LaneId& id, const float* __restrict__ Arow,
                                          int SA, int d0, f4 (&gh)[NSF_HT]) {
  if (LF.ldk == TR_LDK_FAST) wft_chunk_ldk<PT, TR_LDK_FAST>(lds, LF, pl, S, id, Arow, SA, d0, gh);
  else wft_chunk_ldk<PT, 0>(lds, LF, pl, S, id, Arow, SA, d0, gh);
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
// bias gradients from the ones-vector accumulator (HB instantiations): lane column 0 holds the row sums
__device__ __forceinline__ void write_bias(float* __restrict__ part, const LinDesc& L, int out0, const LaneId& id,
                                           const f4& accb) {
  if (id.j != 0) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int out = out0 + 4 * id.g + r;
    if (out < L.out) part[L.g_b + out] = accb[r];
  }
}

// ------------------------------------------------------------------ backward kernel
__device__ __forceinline__ constexpr int cm_reps(const NsfPlan& pl) { return pl.ctx_reps; }   // (0: no hidden layer)
// Wave specialisation, one workgroup = 64-row tile:
//   waves 0-3 ("row" waves): 16 rows each.  Loads, LULinear backward, the spline forward + reverse
//     mode (VALU), the row-wise backward through the residual blocks (transposed-weight MFMA GEMMs);
//   waves 4-7 ("grad" waves): wave 4+w recomputes the final layer for row wave w's rows, multiplies the
//     spline-parameter gradients back through Wf (both MFMA, B operand = stashed h in registers), and
//     owns m-tile w of every weight-gradient for the whole launch (accumulators in registers).
// Wave w and wave 4+w share a SIMD, so the spline's VALU work runs under its partner's MFMAs.
// Tiles in LDS: A0/A1 (gradient side, double buffered: spline parameters of chunk k+1 are produced
// while chunk k is in the spline and chunk k-1 is being consumed), B (activation side), Bs (conditioner
// input [z_id ; context ; 1], static per tile: B operand of d W0 and d Wc).  One workgroup barrier per
// phase; both wave kinds execute the same barrier sequence:
//   S0 | prologue | K0 | chunk steps 0..nch (K1..K_nch) | H | per block: X1 X2 X3 (X4) | Y1 | Y2
// NBT = residual blocks; NBT == 0 selects the theta-dim-1 ContextSplineMap conditioner (compile time, so the
// residual-net instantiations carry none of its code or registers).
// NTW = n-tiles of the narrow input-side weight gradients (d W0, d Wc): 1 when their inputs (+ bias column) fit
// 16 columns, which frees 12 accumulator registers in the grad waves.
// SP = 0: layout from the kernel arguments; SP = 1 + parity: the static default layout (kStaticPl / kStaticTp)
// Everything a backward launch reads and writes.  One launch walks the transforms t_hi ... t_lo (last -> first): the
// dependency between consecutive transforms is tile-local (the gradient wrt a transform's input rows is produced and
// consumed by the same lanes of the same persistent workgroup), so no grid-wide rendezvous is needed between them.
struct BwdIo {
  int t_hi, t_lo;
  const float* packed;       // T weight images
  const float* zstats;
  const float* stash;        // (T, n, D) per-transform input states of the forward pass
  const float* x;
  const float* noise;        // (n, D) base-space point: upstream "gradient" of the last transform
  float* gz[2];              // ping-pong (n, D): gradient wrt transform t's input lives in gz[t & 1]
  const float* row_w;
  float uni_w;
  long long n, x_rows;
  float* partial;            // (T, grid, PLP) per-workgroup partial gradients
  float* grad_theta;
  const float* astash;
  const float* pstash;       // spline-parameter stash of the forward pass (nullptr: the final layer is recomputed)
  long long* dbg;
};
template <int K, int KSH, int NBT, int NCH, int NTW, bool HB, int SP = 0>
__global__ void __launch_bounds__(128 * TR_NW, HB ? 1 : 2)
nsf_bwd_layer_kernel(const NsfPlan pl_, const TrainPlan tp_, const BwdIo io) {
  const float* __restrict__ packed = io.packed;
  const float* __restrict__ zstats = io.zstats;
  const float* __restrict__ x = io.x;
  const float* __restrict__ row_w = io.row_w;
  const float uni_w = io.uni_w;
  const long long n = io.n, x_rows = io.x_rows;
  float* __restrict__ partial = io.partial;
  float* __restrict__ grad_theta = io.grad_theta;
  const float* __restrict__ astash = io.astash;
  long long* __restrict__ dbg = io.dbg;
  // LAYOUT fields (offsets, strides, counts) come from `pl` / `tp`: compile-time constants in the SP instantiations;
  // the floating-point spline constants, the debug switches and the batch-dependent fields from the arguments
  const NsfPlan& pl = SP != 0 ? kStaticPl : pl_;
  const TrainPlan& tp = SP != 0 ? kStaticTp : tp_;
  // Debug aids exist in -DNSF_DEBUG builds only (tools/timeline.py, tools/ablate.sh): the shipped kernels carry neither
  // the cycle-counter stores nor the run-time ablation switches.
#ifdef NSF_DEBUG
  const int dbg_tile_sel = pl_.ablate & 128;   // timeline of the 2nd tile (warm caches) instead of the 1st
#define NSF_ABL(bit) (pl_.ablate & (bit))
#define NSF_ABLV (pl_.ablate)
#define TS(i) do { if (dbg && t == 0 && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && tile == (int)(blockIdx.x + (dbg_tile_sel ? gridDim.x : 0))) \
    dbg[(threadIdx.x >> 6) * 64 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define NSF_ABL(bit) 0
#define NSF_ABLV 0
#define TS(i) do { } while (0)
#endif
  constexpr int PT = (3 * K - 1 + 15) / 16;
  constexpr int DCHB = (4 / PT) > 2 ? 2 : (4 / PT);   // dim slots per chunk (lane pairs: <= 2)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const LaneId id = make_lane();
  constexpr bool cm = (NBT == 0);             // theta-dim 1: context-only MLP conditioner, no LULinear
  constexpr int NB = cm ? 1 : NBT;            // ctx_mlp: one hidden H x H gradient tile set
  const int SLOTS = nsf_ast_slots(pl);           // (a compile-time constant in the static-plan instantiations)
  const int reps = cm_reps(pl);                  // ctx_mlp: applications of the shared hidden layer (>= 1)
  const int D = pl.D, C = pl.C;
  constexpr int SA = TR_SA, SB = TR_SB;
  const int SS = tp.SS;
  float* Bt = lds + tp.o_B;
  float* Bs = lds + tp.o_Bs;
  const float* x_mean = zstats + 2 * D;
  const float* x_std = x_mean + C;

  if (NSF_ABL(256)) {   // test aid (SBI_AMD_ABLATE=256): NaN-filled LDS exposes reads of unwritten locations
    for (int i = tid; i < tp.lds_floats; i += blockDim.x) lds[i] = __builtin_nanf("");
    __syncthreads();
  }
  // Overlay mode (shapes whose whole image does not fit next to the tiles): the weight region holds
  // [final layer, U, L] during the chunk steps and the hidden layers during the block phase, re-staged every
  // tile; `ldsF` makes the final-layer / LU offsets of the plan valid in either mode.
  const bool ov = tp.overlay != 0;
  if (!ov) stage_layer(lds, packed + (long long)io.t_hi * pl.img_floats, pl.lds_w_train_floats, tid, blockDim.x);
  if (tid == 0) *(int*)(lds + tp.o_cnt) = 0;
  if (tid < 32) {
    lds[tp.o_xs + tid] = tid < C ? x_mean[tid] : 0.f;
    lds[tp.o_xs + 32 + tid] = tid < C ? 1.f / x_std[tid] : 1.f;
  }
  const float* xs = lds + tp.o_xs;
  const int ntc = (C + 1 + 15) / 16;
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // does this workgroup walk more than one tile per transform?  (then the first tile of the NEXT transform can be
  // prefetched during the last tile of this one: its upstream gradient was written several tiles ago)
  const bool multi = (int)(blockIdx.x + gridDim.x) < tp_.ntiles;
  // Per-transform view of the layout.  SP != 0: the two mask parities of the static default layout differ only in WHICH
  // dims are transformed (2 d + par) -- every offset, stride and count is the same (static_assert below), so ONE body
  // serves both and takes `par` as a run-time scalar.
#define BWD_T_VARS                                                                                                  \
  const int par = cm ? 0 : (t & 1);                                                                                 \
  const ShapeDesc& S = pl.shape[(SP != 0 || cm) ? 0 : par];                                                         \
  const bool is_last = (t == pl.T - 1);                                                                             \
  const float* img = packed + (long long)t * pl.img_floats;                                                         \
  const int F0 = ov ? S.final_off : 0;                                                                              \
  const float* ldsF = lds - F0;                                                                                     \
  const LinDesc& L0 = S.lin[0];                                                                                     \
  const LinDesc& LF = S.lin[S.fin];                                                                                 \
  const int nch = tp.nch[(SP != 0 || cm) ? 0 : par];                                                                \
  const int nt0 = (S.in0 + 1 + 15) / 16;        /* n-tiles of d W0 (incl. the bias column) */                       \
  /* after the chunk steps: AX receives g_h from the grad waves, AY is the other gradient tile */                   \
  const int o_AX = (nch & 1) ? tp.o_A1 : tp.o_A0;                                                                   \
  const int o_AY = (nch & 1) ? tp.o_A0 : tp.o_A1;                                                                   \
  (void)img; (void)L0; (void)LF; (void)nt0; (void)o_AX; (void)o_AY; (void)ldsF; (void)is_last;

  if (wave < TR_NW) {
    // =========================== row waves ===========================
    float* sc = lds + tp.o_wave + wave * tp.w_total;
    float* zs = sc + tp.w_zs;      // z; transformed dims become the spline output y in place
    float* gys = sc + tp.w_gys;    // g_y; becomes g_x (gradient wrt this transform's input) in place
    float* gzs = sc + tp.w_gzs;    // upstream gradient wrt the LULinear output
    const int arow0 = 16 * wave;   // this wave's rows inside the shared tiles
    // the guard-free mat-vec helpers read 16 floats from a ZW-float row: make sure that never is
    // uninitialised LDS (NaN x 0 = NaN)
    for (int i = id.lane; i < tp.w_total; i += 64) sc[i] = 0.f;
    const LaneId id0 = id;
    // state, context, upstream gradient and row weight of the NEXT tile are requested a whole block phase
    // ahead (clamped addresses instead of predicated loads), so the tile prologue never waits on HBM
    float zv[4], gv[4], xv[8], wv;
    auto fetch_inputs = [&](int t_, int tile_, const LaneId& id) {
      const long long row_ = (long long)tile_ * TR_ROWS + arow0 + id.j;
      const long long rs = row_ < n ? row_ : 0;
      const long long xr = (x_rows == n) ? rs : (x_rows == 1 ? 0 : rs % x_rows);
      const float* z_in = io.stash + (long long)t_ * n * D;
      const float* gz_up = (t_ == pl.T - 1) ? io.noise : io.gz[(t_ + 1) & 1];
      const int d_id_ = pl.shape[(SP != 0 || cm) ? 0 : (t_ & 1)].d_id;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int d = id.g + 4 * u;
        const int dc = d < D ? d : 0;
        zv[u] = z_in[rs * D + dc];
        // (written by these very lanes while they walked the transform above: agent-scope load, not an L1 hit)
        gv[u] = __hip_atomic_load(gz_up + rs * D + dc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = id.g + 4 * u - d_id_;
        xv[u] = x[xr * C + ((c >= 0 && c < C) ? c : 0)];
      }
      wv = row_w ? row_w[rs] : uni_w;
    };
    for (int t = io.t_hi; t >= io.t_lo; --t) {
    BWD_T_VARS
    // (lane coordinates re-materialised per transform as well: LICM would otherwise carry every lane-dependent address
    //  of the loop body across the whole launch)
    LaneId id = id0;
    asm volatile("" : "+v"(id.lane), "+v"(id.j), "+v"(id.g), "+v"(id.iperm));
    float* __restrict__ gz_dn = io.gz[t & 1];
    if (t == io.t_hi || !multi) fetch_inputs(t, blockIdx.x, id);
    for (int tile = blockIdx.x; tile < tp_.ntiles; tile += gridDim.x) {
      // Re-materialise the lane coordinates per tile: otherwise LICM hoists every
      // lane-dependent LDS address of the body out of the persistent loop and the
      // kernel drowns in live registers (hundreds of spills).
      LaneId id = id0;
      asm volatile("" : "+v"(id.j), "+v"(id.g), "+v"(id.iperm));
      const long long row = (long long)tile * TR_ROWS + arow0 + id.j;
      const bool valid = row < n;
      const float wn = valid ? wv : 0.f;
      const float gld = -wn;                       // d(sum w loss)/d(any logabsdet term)
      const int trow = arow0 + id.j;
      // S0 (weights staged / previous tile fully consumed).  Between two tiles of one transform a row wave needs no
      // barrier at all: after Y2 it touches only its own scratch rows and Bs (whose readers finished before Y2), and the
      // shared tiles are next written by the grad waves, who rendezvous among themselves (TR_NO_S0).
      const bool need_s0 = !TR_NO_S0 || ov || tile == (int)blockIdx.x;
      if (need_s0) __syncthreads();
      if (ov) {
        stage_layer(lds, img + F0, pl.lds_w_train_floats - F0, tid, blockDim.x);
        __syncthreads();
      }
      TS(0);
      // ---- P0: state, context, upstream gradient (prefetched) -> LDS
      float gzr[4];       // upstream gradient of dims g + 4 u (zero past D / past the last row)
      {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int d = id.g + 4 * u;
          const float gz = (valid && d < D) ? gv[u] : 0.f;
          gzr[u] = is_last ? wn * gz : gz;                    // last transform: d/dz_T of w*(0.5|z|^2) = w z
          if (d < D) {
            zs[id.j * pl.ZW + d] = valid ? zv[u] : 0.f;
            gzs[id.j * pl.ZW + d] = gzr[u];
          }
        }
        wave_lds_fence();
          // conditioner-input row [z_id ; standardized context ; 1 ; 0 ...] -> static tile Bs (branch-free:
        // clamped reads + selects).  The context is x * (1/std): within 1 ulp of the forward kernel's
        // (x - mean) / std; it only feeds the B operand of d W0 / d Wc.
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = id.g + 4 * u;
          const int c = k - S.d_id;
          const int cc = (c >= 0 && c < C) ? c : 0;
          const int zd = 2 * k + (1 - par);
          const float zid = zs[id.j * pl.ZW + (zd < D ? zd : 0)];
          const float ctx = ((valid ? xv[u] : 0.f) - xs[cc]) * xs[32 + cc];
          float v = (k == S.in0) ? 1.f : 0.f;
          v = (c >= 0 && c < C) ? ctx : v;
          v = (k < S.d_id) ? zid : v;
          if (k < SS) Bs[trow * SS + k] = v;
        }
        for (int k = 32 + id.g; k < SS; k += 4) Bs[trow * SS + k] = 0.f;
      }
      // ---- LULinear backward wrt its input (needs no forward values): g_u = L^T g_z, g_y = U^T g_u
      float gus_r[4] = {0.f, 0.f, 0.f, 0.f};     // g_u of dims 4 g + ii, kept for the LU parameter gradients
      if (cm) {   // no LULinear for theta-dim 1: the transform output IS the layer output
        for (int k = id.g; k < D; k += 4) gys[id.j * pl.ZW + k] = gzs[id.j * pl.ZW + k];
      } else if (!(NSF_ABL(64))) {
        // two chained 16 x 16 mat-vecs on the matrix pipe (was: two VALU mat-vecs with an LDS round trip between them,
        // ~3 k cycles of the prologue while the partner grad wave waits at K0): D fragment reg r of lane (j, g) is dim
        // 4 g + r of row j, and a K-step that covers k = 4 g + s takes that register as its B operand unchanged.
        //   g_u[i] = sum_k L[k][i] g_z[k]     K-step s: k = 4 s + g,  A = L[4 s + g][i],  B = g_z[row j][4 s + g]
        //   g_y[i] = sum_k U[k][i] g_u[k]     K-step s: k = 4 g + s,  A = U[4 g + s][i],  B = g_u fragment reg s
        // (L, U are zero padded to 16 x 16 and the state rows are followed by finite scratch: no bounds checks)
        const float* Lm = ldsF + S.l_L;
        const float* Um = ldsF + S.l_U;
        f4 gu = zero4, gy = zero4;
        float al[4], bz[4], au[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          al[s4] = Lm[(4 * s4 + id.g) * 16 + id.j];
          bz[s4] = gzr[s4];       // = gzs[row j][4 s4 + g]: the prefetched registers already have this layout
          au[s4] = Um[(4 * id.g + s4) * 16 + id.j];
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) gu = MFMA16(al[s4], bz[s4], gu);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) gy = MFMA16(au[s4], gu[s4], gy);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          gus_r[ii] = gu[ii];
          if (4 * id.g + ii < D) gys[id.j * pl.ZW + 4 * id.g + ii] = gy[ii];   // identity dims pass through
        }
      }
      wave_lds_fence();
      // wave-tiles past the last row were never stashed by the forward pass: read the last real one instead
      // (their rows carry zero weight, but 0 x stale NaN would still poison the weight gradients)
      const long long nt16 = (n + 15) / 16;
      const long long wt16 = (long long)tile * TR_NW + wave < nt16 ? (long long)tile * TR_NW + wave : nt16 - 1;
      const float* ast = astash + (((long long)t * nt16 + wt16) * SLOTS) * 1024 +
                         4 * id.lane;
      TS(1);
      __syncthreads();                             // K0: spline parameters of chunk 0 are in A0
      TS(2);

      // ---- chunk steps: spline forward + reverse mode in place on the grad waves' parameter rows
      f4 bt1[NSF_HT], bt2[NSF_HT], bsg[NSF_HT];   // block temporaries, loaded one phase ahead of their use
      f4 hpre[cm ? 2 : 1][NSF_HT];                // block input h_b (ctx_mlp: h1, h2)
      if (TR_SPLINE_PRIO) __builtin_amdgcn_s_setprio(TR_SPLINE_PRIO);
      for (int c = 0; c < nch; ++c) {
        const int d0 = c * DCHB;
        const int slot = id.g & 1, part = id.g >> 1;
        const int dd = d0 + slot;
        float* pp = lds + ((c & 1) ? tp.o_A1 : tp.o_A0) + trow * SA + slot * TR_SLOT(PT);
        if (slot < DCHB) {
          if (dd < S.d_tr && !(NSF_ABL(4))) {
            const int zi = id.j * pl.ZW + 2 * dd + par;
            float yv, gxv;
            rq_spline_pair_bwd<K>(pp, tp.PTW, zs[zi], gys[zi], gld, pl_, part, yv, gxv);
            if (part == 0) {
              zs[zi] = yv;
              gys[zi] = gxv;
            }
          } else if (part == 0) {
            for (int k = 0; k < tp.PTW; ++k) pp[k] = 0.f;
          }
        }
        TS(3 + c);
        __syncthreads();                           // K_{c+1}
      }
      if (TR_SPLINE_PRIO) __builtin_amdgcn_s_setprio(0);
      // ---- step nch (the grad waves finish d Wf / Wf^T g of the last chunk): fetch the last block's
      // temporaries and do the LULinear forward piece its parameter gradients need, u = U y
      if (cm) {     // the last application's input and output: h_reps, h_{reps+1} (stash slots reps - 1, reps)
        ast_load<KSH>(ast, reps > 0 ? reps - 1 : 0, hpre[0]);
        if (reps > 0) ast_load<KSH>(ast, reps, hpre[1]);
      } else {
        ast_load<KSH>(ast, 2 + 4 * (NB - 1), bt2);
        ast_load<KSH>(ast, 3 + 4 * (NB - 1), bsg);
        ast_load<KSH>(ast, 1 + 4 * (NB - 1), bt1);
      }
      float us_r[4] = {0.f, 0.f, 0.f, 0.f};
      if (!cm && !(NSF_ABL(64))) {     // u[i] = sum_k U[i][k] y[k] on the matrix pipe: K-step s covers k = 4 s + g
        const float* Um = ldsF + S.l_U;
        float au[4], by[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          au[s4] = Um[id.j * 16 + 4 * s4 + id.g];
          by[s4] = zs[id.j * pl.ZW + 4 * s4 + id.g];
        }
        f4 uv = zero4;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) uv = MFMA16(au[s4], by[s4], uv);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) us_r[ii] = uv[ii];
      }
      {   // next tile's inputs (requested while this wave waits for g_h); after the last tile of a transform: the first
          // tile of the next one (its upstream gradient left this workgroup several tiles ago -- unless this IS the only
          // tile: then the fetch waits for the top of the next transform; re-reading the own tile is harmless)
        const int nxt = tile + (int)gridDim.x;
        const bool more = nxt < tp_.ntiles, hop = !more && multi && t > io.t_lo;
        fetch_inputs(hop ? t - 1 : t, more ? nxt : (hop ? (int)blockIdx.x : tile), id);
      }
      TS(8);
      __syncthreads();                             // H: g_h = Wf^T g_p of this wave's rows is in AX
      if (ov) {   // nobody needs the final layer / LU any more this tile: the hidden layers take the region
        stage_layer(lds, img, S.final_off, tid, blockDim.x);
        __syncthreads();
      }
      TS(9);
      // TR_H_SPLIT: the grad waves published g_h BEFORE the last chunk's d Wf, which still reads AY and B: everything up
      // to this wave's writes into those two tiles runs under that GEMM, then barrier H2
      const bool hsplit = TR_H_SPLIT && !cm && !ov;
      f4 gh[NSF_HT];
      load_D(lds + o_AX, SA, trow, id, gh);
      wave_lds_fence();
      // the caller trains an embedding net in front of the flow: also produce d loss / d context (launches run
      // last -> first transform on one stream: the first contribution of the first launch overwrites)
      const bool want_gx = tp_.grad_x != nullptr;
      float* gx_row = want_gx ? tp_.grad_x + (valid ? row : 0) * C : nullptr;

      if (cm) {
        // ---- ctx_mlp: h_{i+1} = relu(W_h h_i + b_h), i = reps ... 1: the SAME hidden layer walked back `reps` times
        // (hidden_layers_spline_context; its weight gradient accumulates over the applications in the grad waves)
        f4 ga[NSF_HT], gb[NSF_HT];
        for (int i = reps; i >= 1; --i) {
          if (i < reps) {           // (the first iteration's operands were requested before the H barrier)
            ast_load<KSH>(ast, i - 1, hpre[0]);
            ast_load<KSH>(ast, i, hpre[1]);
            __syncthreads();        // X0: the previous application's d W_h has consumed the tiles
          }
#pragma unroll
          for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ga[mt][r] = hpre[1][mt][r] > 0.f ? gh[mt][r] : 0.f;
          stage_D(lds + o_AY, SA, trow, id, ga, false);
          stage_DB(Bt, SB, trow, id, hpre[0], true);
          if (!HB && id.g == 0) Bt[trow * SB + il_col(pl.H)] = 1.f;
          __syncthreads();                           // X1
#pragma unroll
          for (int mt = 0; mt < NSF_HT; ++mt) gb[mt] = zero4;
          gemm_T_breg<KSH, NSF_HT>(lds, S.lin[1], id, ga, gb, NSF_ABLV);
#pragma unroll
          for (int mt = 0; mt < NSF_HT; ++mt) gh[mt] = gb[mt];     // gradient wrt h_i (post-relu)
        }
#pragma unroll
        for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) gh[mt][r] = hpre[0][mt][r] > 0.f ? gh[mt][r] : 0.f;   // through h_1 = relu(...)
      } else {
        // ---- residual blocks, last -> first
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
            stage_D(lds + o_AX, SA, trow, id, gc, false);     // (own rows of AX: nobody else reads them before X1)
            if (want_gx)
              ctx_grad_update<KSH>(lds, S.lin[1 + 3 * b], id, gc, 0, C, xs, gx_row, valid, is_last && b == NB - 1);
            if (hsplit && b == NB - 1) {
              TS(50);
              __syncthreads();                     // H2: the last chunk's d Wf has consumed AY and B
              TS(51);
            }
            stage_D(lds + o_AY, SA, trow, id, ga, false);
          }
          stage_DB(Bt, SB, trow, id, bt1, true);
          if (!HB && id.g == 0) Bt[trow * SB + il_col(pl.H)] = 1.f;      // bias column
          ast_load<KSH>(ast, 4 * b, hpre[0]);                  // h_b: needed two phases from now
          if (b > 0) {   // next (earlier) block's t2 / gate: fetch under this block's GEMM phases
            ast_load<KSH>(ast, 2 + 4 * (b - 1), bt2);
            ast_load<KSH>(ast, 3 + 4 * (b - 1), bsg);
          }
          TS(20 + 8 * b);
          __syncthreads();                         // X1: (g_t2, relu t1) and (g_c, Bs) published
          TS(21 + 8 * b);
#pragma unroll
          for (int mt = 0; mt < NSF_HT; ++mt) gb[mt] = zero4;
          gemm_T_breg<KSH, NSF_HT>(lds, S.lin[3 + 3 * b], id, ga, gb, NSF_ABLV);       // d relu(t1)
#pragma unroll
          for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ga[mt][r] = bt1[mt][r] > 0.f ? gb[mt][r] : 0.f;   // d t1
          if (b > 0) ast_load<KSH>(ast, 1 + 4 * (b - 1), bt1);
          TS(22 + 8 * b);
          __syncthreads();                         // X2: d W2 / d Wc done, tiles free
          TS(23 + 8 * b);
          stage_D(lds + o_AY, SA, trow, id, ga, false);
          stage_DB(Bt, SB, trow, id, hpre[0], true);
          if (!HB && id.g == 0) Bt[trow * SB + il_col(pl.H)] = 1.f;
          __syncthreads();                         // X3: (g_t1, relu h_b) published
          TS(24 + 8 * b);
#pragma unroll
          for (int mt = 0; mt < NSF_HT; ++mt) gb[mt] = zero4;
          gemm_T_breg<KSH, NSF_HT>(lds, S.lin[2 + 3 * b], id, ga, gb, NSF_ABLV);       // d relu(h_b)
#pragma unroll
          for (int mt = 0; mt < NSF_HT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) gh[mt][r] += hpre[0][mt][r] > 0.f ? gb[mt][r] : 0.f;
          TS(25 + 8 * b);
          if (b > 0) __syncthreads();              // X4: d W1 done, tiles free for the next block
        }
      }

      // ---- initial layer: publish g_h0 (AX is free: d Wc finished before X2)
      stage_D(lds + o_AX, SA, trow, id, gh, false);
      TS(40);
      __syncthreads();                             // Y1
      TS(41);
      {
        f4 gin[1];
        gin[0] = zero4;
        gemm_T_breg<KSH, 1>(lds, L0, id, gh, gin, NSF_ABLV);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 4 * r + id.g;     // identity feature slot
          if (k < S.d_id) gys[id.j * pl.ZW + 2 * k + (1 - par)] += gin[0][r];
        }
      }
      if (want_gx) ctx_grad_update<KSH>(lds, L0, id, gh, S.d_id, C, xs, gx_row, valid, is_last && cm);
      // ---- LULinear parameter gradients as two more 16x16 tiles (AY / B are free after Y1):
      //   d U = g_u (x) y (+ the logabsdet row),  d L = g_z (x) u,  d bias = sum g_z
      if (!cm) {
        float* Ay = lds + o_AY + trow * SA;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int k = 4 * id.g + ii;
          const int o = id.j * pl.ZW + k;
          Ay[k] = k < D ? gus_r[ii] : (k == D ? gld : 0.f);
          Ay[16 + k] = k < D ? gzs[o] : 0.f;
          Bt[trow * SB + il_col(k)] = k < D ? zs[o] : (k == D ? 1.f : 0.f);
          Bt[trow * SB + il_col(16 + k)] = k < D ? us_r[ii] : (k == D ? 1.f : 0.f);
        }
      }
      TS(42);
      __syncthreads();                             // Y2
      // ---- gradient wrt this transform's input
      for (int d = id.g; d < D; d += 4) {
        if (valid) {
          const float g = gys[id.j * pl.ZW + d];
          if (t > 0) gz_dn[row * D + d] = g;
          else if (grad_theta) grad_theta[row * D + d] = g * zstats[D + d];
        }
      }
      TS(43);
    }
    // the next transform's weight image (nobody reads the old one after the last tile's Y2; the grad waves are busy
    // writing their partial gradients meanwhile; the S0 barrier of the next tile publishes it)
    if (t > io.t_lo && !ov) stage_layer(lds, packed + (long long)(t - 1) * pl.img_floats, pl.lds_w_train_floats, tid, 64 * TR_NW);
    }   // transforms

  } else {
    // =========================== grad waves ==========================
    const int gw = wave - TR_NW;
    int* cnt = (int*)(lds + tp.o_cnt);
    int sync_target = 0;
    const LaneId id0 = id;
    f4 hl[NSF_HT];
    auto fetch_hl = [&](int t_, int tile_) {
      const long long nt16 = (n + 15) / 16;   // clamp as the row waves do: unstashed wave-tiles hold stale memory
      const long long wt16 = (long long)tile_ * TR_NW + gw < nt16 ? (long long)tile_ * TR_NW + gw : nt16 - 1;
      const float* ast = astash + (((long long)t_ * nt16 + wt16) * SLOTS) * 1024 +
                         4 * id0.lane;
      ast_load<KSH>(ast, cm ? reps : 4 * NB, hl);
    };
    // spline parameters of the NEXT chunk, requested from the forward pass's stash one phase ahead (they replace the
    // final-layer recompute: 52 MFMAs + 60 LDS reads per chunk and wave)
    constexpr bool use_pst = TR_PSTASH != 0;      // (the host passes the stash whenever the kernels were built for it)
    f4 pq[DCHB][PT];
    auto pq_load = [&](int t_, int tile_, int c_) {
      const long long nt16 = (n + 15) / 16;
      const long long wt16 = (long long)tile_ * TR_NW + gw < nt16 ? (long long)tile_ * TR_NW + gw : nt16 - 1;
      const float* pstw = io.pstash + ((long long)t_ * nt16 + wt16) * nsf_pst_tile_floats(pl) + 4 * id0.lane;
      const int d_tr_ = pl.shape[(SP != 0 || cm) ? 0 : (t_ & 1)].d_tr;
#pragma unroll
      for (int sl = 0; sl < DCHB; ++sl) {
        const int dd = c_ * DCHB + sl;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) pq[sl][pt] = pst_load<PT>(pstw, dd < d_tr_ ? dd : d_tr_ - 1, pt);   // (dead slot: any valid word)
      }
    };
    // parameters -> this wave's rows of a gradient tile, exactly where final_layer_chunk_T puts them
    auto pq_store = [&](float* __restrict__ arow, const LaneId& id, int c_, int d_tr_) {
#pragma unroll
      for (int sl = 0; sl < DCHB; ++sl)
        if (c_ * DCHB + sl < d_tr_) {        // (wave-uniform; a dead slot's rows are zero-filled by the row wave)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int r = 0; r < 4; ++r) arow[id.j * SA + sl * TR_SLOT(PT) + 16 * pt + 4 * r + id.g] = pq[sl][pt][r];
        }
    };
    fetch_hl(io.t_hi, blockIdx.x);
    if (use_pst) pq_load(io.t_hi, blockIdx.x, 0);
    for (int t = io.t_hi; t >= io.t_lo; --t) {
    BWD_T_VARS
    LaneId id = id0;      // (re-materialised per transform: keeps the write-out's ~130 lane-dependent offsets out of registers)
    asm volatile("" : "+v"(id.lane), "+v"(id.j), "+v"(id.g), "+v"(id.iperm));
    // weight-gradient accumulators owned by this wave (m-tile = wave) for the whole transform
    f4 acc0[NTW], accC[NB][NTW], acc1[NB][4], acc2[NB][4], accF[NCH][4], accLU[1];
    f4 acc1b[HB ? NB : 1], acc2b[HB ? NB : 1], accFb[HB ? NCH : 1];   // bias gradients when hidden_features == 64
  #pragma unroll
    for (int i = 0; i < (HB ? NB : 1); ++i) { acc1b[i] = zero4; acc2b[i] = zero4; }
  #pragma unroll
    for (int i = 0; i < (HB ? NCH : 1); ++i) accFb[i] = zero4;
  #pragma unroll
    for (int i = 0; i < NTW; ++i) acc0[i] = zero4;
  #pragma unroll
    for (int b = 0; b < NB; ++b) {
  #pragma unroll
      for (int i = 0; i < NTW; ++i) accC[b][i] = zero4;
  #pragma unroll
      for (int i = 0; i < 4; ++i) { acc1[b][i] = zero4; acc2[b][i] = zero4; }
    }
  #pragma unroll
    for (int c = 0; c < NCH; ++c)
  #pragma unroll
      for (int i = 0; i < 4; ++i) accF[c][i] = zero4;
    accLU[0] = zero4;

    for (int tile = blockIdx.x; tile < tp_.ntiles; tile += gridDim.x) {
      LaneId id = id0;
      asm volatile("" : "+v"(id.j), "+v"(id.g), "+v"(id.iperm));
      const int trow = 16 * gw + id.j;             // rows of the partner row wave
      // next h_last: the next tile's, or after the last tile the first tile of the next transform (the stash is
      // read-only here)
      const bool more_tiles = tile + (int)gridDim.x < tp_.ntiles;
      const int t_nxt = more_tiles ? t : (t > io.t_lo ? t - 1 : t);
      const int tile_nxt = more_tiles ? tile + (int)gridDim.x : (t > io.t_lo ? (int)blockIdx.x : tile);
      const bool need_s0 = !TR_NO_S0 || ov || tile == (int)blockIdx.x;
      if (need_s0) __syncthreads();                // S0
      else { sync_target += 4; grad_wave_sync(cnt, sync_target, id.lane); }   // every d LU read of the previous tile is done
      if (ov) {
        stage_layer(lds, img + F0, pl.lds_w_train_floats - F0, tid, blockDim.x);
        __syncthreads();
      }
      TS(0);
      // ---- prologue: h_last of the partner's rows (stash, D-fragment order = MFMA B operand; requested
      // during the previous tile), its activation-tile rows, and the spline parameters of chunk 0
      stage_DB(Bt, SB, trow, id, hl, false);
      if (!HB && id.g == 0) Bt[trow * SB + il_col(pl.H)] = 1.f;   // bias column
      if (use_pst) {
        pq_store(lds + tp.o_A0 + 16 * gw * SA, id, 0, S.d_tr);
        if (1 < nch) pq_load(t, tile, 1);
      } else if (!(NSF_ABL(32))) final_layer_chunk_T<PT, KSH>(ldsF, lds + tp.o_A0 + 16 * gw * SA, pl, tp, S, id, hl, 0);
      f4 gh[NSF_HT];
  #pragma unroll
      for (int mt = 0; mt < NSF_HT; ++mt) gh[mt] = zero4;
      TS(1);
      __syncthreads();                             // K0
      TS(2);
      // ---- chunk steps: d Wf and Wf^T g of chunk k-1, spline parameters of chunk k+1
  #pragma unroll
      for (int k = 0; k <= NCH; ++k) {
        if (k <= nch) {
          if (k >= 1) {
            const int oa = ((k - 1) & 1) ? tp.o_A1 : tp.o_A0;
            // m-tile gw = 16 spline-parameter columns of dim slot gw / PT (the second slot starts one float late)
            const bool hsplit = TR_H_SPLIT && !cm && !ov;
            if (hsplit && k == nch) {     // critical path first: g_h to the partner row wave, THEN this chunk's d Wf
              if (!(NSF_ABL(2))) wft_chunk<PT>(ldsF, LF, pl, S, id, lds + oa + 16 * gw * SA, SA, (k - 1) * DCHB, gh);
              stage_D(lds + o_AX, SA, trow, id, gh, false);
              TS(9);
              __syncthreads();            // H
            }
            if (!HB && DCHB == 2 && k == nch && (S.d_tr & 1)) {
              // last chunk of an odd number of dims: only dim slot 0 is live (PT parameter tiles).  Instead of two
              // waves multiplying the dead slot's zeros, the four waves split the live tiles' n-tiles: wave gw takes
              // parameter tile gw % PT, n-tiles [PT (gw / PT), PT (gw / PT) + PT)  (accF[.][0 .. PT), see the write-out)
              f4 (&af)[4] = accF[k - 1 < NCH ? k - 1 : 0];
              f4 part[PT < 3 ? PT : 1];
#pragma unroll
              for (int i = 0; i < (PT < 3 ? PT : 1); ++i) part[i] = af[i];
              dw_gemm<(PT < 3 ? PT : 1), TR_SA, TR_SB, true>(lds + oa, Bt, 16 * (gw % PT), 16 * PT * (gw / PT), id, part,
                                                             PT < 3 ? PT : 1, NSF_ABLV, nullptr);
#pragma unroll
              for (int i = 0; i < (PT < 3 ? PT : 1); ++i) af[i] = part[i];
            } else
            dw_gemm<4, TR_SA, TR_SB, true>(lds + oa, Bt, 16 * gw + (gw / PT < DCHB ? gw / PT : 0), 0, id,
                                           accF[k - 1 < NCH ? k - 1 : 0], 4, NSF_ABLV,
                                           HB ? &accFb[k - 1 < NCH ? k - 1 : 0] : nullptr);
            TS(13 + k);
            if (!(hsplit && k == nch) && !(NSF_ABL(2)))
              wft_chunk<PT>(ldsF, LF, pl, S, id, lds + oa + 16 * gw * SA, SA, (k - 1) * DCHB, gh);
          }
          TS(3 + 2 * k);
          if (k + 1 < nch) {
            if (k >= 1) { sync_target += 4; grad_wave_sync(cnt, sync_target, id.lane); }   // all d Wf reads of that tile done
            TS(17 + k);
            if (use_pst) {
              pq_store(lds + (((k + 1) & 1) ? tp.o_A1 : tp.o_A0) + 16 * gw * SA, id, k + 1, S.d_tr);
              if (k + 2 < nch) pq_load(t, tile, k + 2);
            } else if (!(NSF_ABL(32)))
              final_layer_chunk_T<PT, KSH>(ldsF, lds + (((k + 1) & 1) ? tp.o_A1 : tp.o_A0) + 16 * gw * SA, pl, tp, S, id,
                                           hl, (k + 1) * DCHB);
          }
          if (k == nch && !(TR_H_SPLIT && !cm && !ov)) stage_D(lds + o_AX, SA, trow, id, gh, false);   // hand g_h to the partner row wave
          TS(4 + 2 * k);
          __syncthreads();                         // K_{k+1} / H (TR_H_SPLIT: H2)
          if (ov && k == nch) {                    // mirrors the row waves: hidden layers take the weight region
            stage_layer(lds, img, S.final_off, tid, blockDim.x);
            __syncthreads();
          }
        }
      }
      if (cm) {
        for (int i = reps; i >= 1; --i) {
          if (i < reps) __syncthreads();           // X0
          __syncthreads();                         // X1
          dw_gemm<4, TR_SA, TR_SB, true>(lds + o_AY, Bt, 16 * gw, 0, id, acc1[0], 4, NSF_ABLV, HB ? &acc1b[0] : nullptr);
        }
      } else {
  #pragma unroll
        for (int b = NB - 1; b >= 0; --b) {
          __syncthreads();                         // X1
          TS(21 + 8 * b);
          dw_gemm<4, TR_SA, TR_SB, true>(lds + o_AY, Bt, 16 * gw, 0, id, acc2[b], 4, NSF_ABLV, HB ? &acc2b[b] : nullptr);
          TS(22 + 8 * b);
          __syncthreads();                         // X2
          // d Wc under the row waves' re-staging of AY / B (the matrix pipe used to idle between X2 and X3): its
          // operands -- g_c in AX, the static input tile Bs -- are not rewritten before X4 / the initial layer's Y1
          dw_gemm_rs<NTW, TR_SA>(lds + o_AX, Bs, SS, 16 * gw, S.d_id, id, accC[b], ntc, NSF_ABLV);
          __syncthreads();                         // X3
          TS(24 + 8 * b);
          dw_gemm<4, TR_SA, TR_SB, true>(lds + o_AY, Bt, 16 * gw, 0, id, acc1[b], 4, NSF_ABLV, HB ? &acc1b[b] : nullptr);
          TS(25 + 8 * b);
          if (b > 0) __syncthreads();              // X4
        }
      }
      __syncthreads();                             // Y1
      TS(41);
      fetch_hl(t_nxt, tile_nxt);   // next tile's h_last: lands under this tile's last two phases
      if (use_pst) pq_load(t_nxt, tile_nxt, 0);
      dw_gemm_rs<NTW, TR_SA>(lds + o_AX, Bs, SS, 16 * gw, 0, id, acc0, nt0, NSF_ABLV);
      TS(42);
      __syncthreads();                             // Y2
      if (gw < 2 && !cm) dw_gemm<1, TR_SA, TR_SB, true>(lds + o_AY, Bt, 16 * gw, 16 * gw, id, accLU);
      TS(43);
    }

    // ---- write this workgroup's partial gradients (natural parameter order)
    float* part = partial + ((long long)t * gridDim.x + blockIdx.x) * tp.PLP;
    const int out0 = 16 * gw;
  #pragma unroll
    for (int nt = 0; nt < NTW; ++nt) write_tile(part, L0, out0, nt, id, acc0[nt]);
    if (cm) {
  #pragma unroll
      for (int nt = 0; nt < 4; ++nt) write_tile(part, S.lin[1], out0, nt, id, acc1[0][nt]);   // hidden H->H layer
      if (HB) write_bias(part, S.lin[1], out0, id, acc1b[0]);
    } else {
  #pragma unroll
      for (int b = 0; b < NB; ++b) {
  #pragma unroll
        for (int nt = 0; nt < NTW; ++nt) write_tile(part, S.lin[1 + 3 * b], out0, nt, id, accC[b][nt]);
  #pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          write_tile(part, S.lin[2 + 3 * b], out0, nt, id, acc1[b][nt]);
          write_tile(part, S.lin[3 + 3 * b], out0, nt, id, acc2[b][nt]);
        }
        if (HB) {
          write_bias(part, S.lin[2 + 3 * b], out0, id, acc1b[b]);
          write_bias(part, S.lin[3 + 3 * b], out0, id, acc2b[b]);
        }
      }
    }
  #pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c < nch) {
        if (!HB && DCHB == 2 && c == nch - 1 && (S.d_tr & 1)) {     // the split last chunk (see the chunk steps)
          const int dd = c * DCHB, pt = gw % PT;
  #pragma unroll
          for (int i = 0; i < (PT < 3 ? PT : 1); ++i)
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int p = 16 * pt + 4 * id.g + r;
              const int in = 16 * (PT * (gw / PT) + i) + id.j;
              if (p < pl.P) {
                const int out = dd * pl.P + p;
                if (in < LF.in) part[LF.g_w + out * LF.in + in] = accF[c][i][r];
                else if (in == LF.in) part[LF.g_b + out] = accF[c][i][r];
              }
            }
          continue;
        }
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
                if (HB && nt == 0 && id.j == 0) part[LF.g_b + out] = accFb[c][r];
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
    }   // transforms

  }
#undef BWD_T_VARS
#undef NSF_ABL
#undef NSF_ABLV
#undef TS
}


// ---- launch helpers
template <int K, int KSH, int NB, int NCH, int NTW, bool HB = false, int SP = 0>
static int launch_bwd(const NsfPlan& pl, const TrainPlan& tp, const BwdIo& io, hipStream_t st) {
  auto kern = nsf_bwd_layer_kernel<K, KSH, NB, NCH, NTW, HB, SP>;
  const int lds_bytes = 4 * tp.lds_floats;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(kern, dim3(tp.grid), dim3(128 * TR_NW), (size_t)lds_bytes, st, pl, tp, io);
  return (int)hipGetLastError();
}

template <int K>
static int launch_bwd_one(const NsfPlan& pl, const TrainPlan& tp, const BwdIo& io, int nchmax, bool wide, hipStream_t st) {
#define BWD_ARGS pl, tp, io, st
#define BWD_NCH(KS, NBV) \
  switch (nchmax) { \
    case 1: return wide ? launch_bwd<K, KS, NBV, 1, 2>(BWD_ARGS) : launch_bwd<K, KS, NBV, 1, 1>(BWD_ARGS); \
    case 2: case 3: return wide ? launch_bwd<K, KS, NBV, 3, 2>(BWD_ARGS) : launch_bwd<K, KS, NBV, 3, 1>(BWD_ARGS); \
    default: return wide ? launch_bwd<K, KS, NBV, 4, 2>(BWD_ARGS) : launch_bwd<K, KS, NBV, 4, 1>(BWD_ARGS); \
  }
  if (pl.ctx_mlp) {   // theta-dim 1: one transformed dim => one chunk
    if (pl.KSH == 13) return wide ? launch_bwd<K, 13, 0, 1, 2>(BWD_ARGS) : launch_bwd<K, 13, 0, 1, 1>(BWD_ARGS);
    if (pl.H == 64) return wide ? launch_bwd<K, 16, 0, 1, 2, true>(BWD_ARGS) : launch_bwd<K, 16, 0, 1, 1, true>(BWD_ARGS);
    return wide ? launch_bwd<K, 16, 0, 1, 2>(BWD_ARGS) : launch_bwd<K, 16, 0, 1, 1>(BWD_ARGS);
  }
  if (pl.KSH == 13) { if (pl.NB <= 1) { BWD_NCH(13, 1) } else { BWD_NCH(13, 2) } }
  if (pl.H == 64) {   // bias gradients through the ones-vector MFMA
#define BWD_NCH_HB(NBV) \
  switch (nchmax) { \
    case 1: return wide ? launch_bwd<K, 16, NBV, 1, 2, true>(BWD_ARGS) : launch_bwd<K, 16, NBV, 1, 1, true>(BWD_ARGS); \
    case 2: case 3: return wide ? launch_bwd<K, 16, NBV, 3, 2, true>(BWD_ARGS) : launch_bwd<K, 16, NBV, 3, 1, true>(BWD_ARGS); \
    default: return wide ? launch_bwd<K, 16, NBV, 4, 2, true>(BWD_ARGS) : launch_bwd<K, 16, NBV, 4, 1, true>(BWD_ARGS); \
  }
    if (pl.NB <= 1) { BWD_NCH_HB(1) } else { BWD_NCH_HB(2) }
#undef BWD_NCH_HB
  }
  if (pl.NB <= 1) { BWD_NCH(16, 1) } else { BWD_NCH(16, 2) }
#undef BWD_NCH
#undef BWD_ARGS
}

// the two mask parities of the static default layout are the same layout (ONE kernel body, run-time parity)
constexpr bool static_parities_match() {
  const ShapeDesc& a = kStaticPl.shape[0];
  const ShapeDesc& b = kStaticPl.shape[1];
  if (a.d_id != b.d_id || a.d_tr != b.d_tr || a.in0 != b.in0 || a.fin != b.fin || a.final_off != b.final_off ||
      a.l_U != b.l_U || a.l_L != b.l_L || a.l_lub != b.l_lub || a.g_lu != b.g_lu || a.n_params != b.n_params)
    return false;
  for (int i = 0; i <= a.fin; ++i) {
    const LinDesc& x = a.lin[i];
    const LinDesc& y = b.lin[i];
    if (x.in != y.in || x.out != y.out || x.ldk != y.ldk || x.rows != y.rows || x.l_w != y.l_w || x.l_b != y.l_b ||
        x.g_w != y.g_w || x.g_b != y.g_b || x.ksteps != y.ksteps)
      return false;
  }
  return kStaticTp.nch[0] == kStaticTp.nch[1];
}
static_assert(static_parities_match(), "static default layout: both mask parities must share one layout");

// Runs the transforms io.t_hi ... io.t_lo.  The static default layout takes them all in ONE launch (no kernel boundaries,
// the next image staged under the partial-gradient write-out); every other shape one launch per transform.
template <int K>
int launch_bwd_k(const NsfPlan& pl, const TrainPlan& tp, const BwdIo& io_all, hipStream_t st) {
  if constexpr (K == 10) {     // the benchmark configuration: layout folded into the kernel (see kStaticPl)
    if (plan_is_static_default(pl, tp)) return launch_bwd<10, 13, 2, 3, 1, false, 1>(pl, tp, io_all, st);
  }
  const int nchmax = tp.nch[0] > tp.nch[1] ? tp.nch[0] : tp.nch[1];
  // narrow input side (d W0 / d Wc fit one 16-column n-tile incl. the bias column): the common case
  int in0max = pl.shape[0].in0 > pl.shape[1].in0 ? pl.shape[0].in0 : pl.shape[1].in0;
  const bool wide = (in0max + 1 > 16) || (pl.C + 1 > 16);
  for (int t = io_all.t_hi; t >= io_all.t_lo; --t) {
    BwdIo io = io_all;
    io.t_hi = io.t_lo = t;
    const int rc = launch_bwd_one<K>(pl, tp, io, nchmax, wide, st);
    if (rc) return rc;
  }
  return 0;
}
