// fmpe.hip -- FMPE (flow matching) vector-field MLP on gfx950: velocity, CFM loss, loss + gradients.
//
// Reference behaviour (restated in oracle/fmpe_oracle.py, pinned to the real classes by tests/golden):
//   VectorFieldMLP.forward            sbi/neural_nets/net_builders/vector_field_nets.py:683-719
//   FlowMatchingEstimator.forward/loss sbi/neural_nets/estimators/flowmatching_estimator.py:206-347
//
// Execution model
//   * one wavefront owns 16 batch rows for the whole network.  Every dense layer runs on
//     v_mfma_f32_16x16x4_f32 in the transposed form  Y^T = W X^T  (M = output feature, N = batch row,
//     K = input feature): lane (c = lane&15, g = lane>>4) holds, per 16-feature block, the four features
//     16*blk + 4*g + {0..3} of row c.  That is at once the D fragment a layer produces and the B fragment the
//     next one consumes (K-step r of block kb uses k = 16*kb + 4*g + r on both operands), so activations
//     never leave registers; bias, GELU, the time embedding, the skip connection and LayerNorm are applied to
//     the accumulators in place.
//   * weights are the A operand: one ds_read_b128 per lane (row 16*ob + c of the zero-padded image, columns
//     16*kb + 4*g ..+3) feeds four MFMAs; row stride = 16*KB + 4 floats (stride/4 odd: conflict free).
//     The images do not fit LDS together (sbi's default net: 352 KB), so a workgroup (4 waves, 64 rows) stages
//     them group by group from L2; two workgroups per CU overlap one's staging with the other's MFMAs.
//   * training: the forward kernel stashes what the backward needs (pre-activations, normalised LayerNorm
//     outputs, network inputs) in 1 KB blocks [16 rows][16 features] -- one coalesced 16-byte store per lane.
//     The backward kernel walks the layers in reverse with W^T images (dX chain, LayerNorm / GELU backward in
//     registers) and stores each linear's output gradient G transposed ([feature][row]); the weight-gradient
//     kernel then contracts G^T X over rows with both operands straight from L2 (A by 16-byte loads), one
//     workgroup per (row chunk, linear), partial sums per chunk, deterministic reduce.  LayerNorm outputs
//     are stashed normalised (s_hat); the reduce turns  M = G^T s_hat, db  into  dW = M*gamma + db (x) beta.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/sbi_amd_fmpe.h"
#include "../../include/sbi_amd_nsf.h"
#include "debug_env.h"

typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define FM_MAX_L 8
#define FM_MAX_LIN (6 + FM_MAX_L)
#define FM_THREADS 512              // forward / backward kernels: 8 waves x 16 rows
#define FM_WAVES 8
#define FM_ROWS 128
#define FM_DW_THREADS 256
#define FM_FWD_GROUP_FLOATS 17920   // 70 KB of weight image per staging group (2 workgroups per CU)
#define FM_BWD_GROUP_FLOATS 17920
#define FM_DW_TILES 64              // wave-tiles (16 rows) per weight-gradient chunk

enum { J_IN = 0, J_CT = 1, J_TM = 2, J_MA = 3, J_MB = 4, J_L0 = 5 };

struct FmLin {
  int out, in, OB, KB;          // natural dims, 16-blocks
  int g_w, g_ld, g_b;           // flat buffer: W[o][i] at g_w + o*g_ld + i; bias at g_b (-1: none)
  int w_off, ldk, lw, lb;       // forward image: packed offset, row stride; LDS offsets of W / bias inside its group
  int t_off, ldt, ltw, ltg;     // backward image (W^T [in][out]): packed offset, stride; LDS offsets of W^T / gamma
  int fg_first, bg_first;       // this linear opens a new forward / backward staging group
  int s_x, s_g;                 // stash slots (in blocks): X natural, G transposed
  int x_gelu;                   // X = GELU(stashed pre-activation)
  int ln_fix;                   // >= 0: X was s_hat of that layer; dW = M*gamma + db (x) beta
  int pf_w, pf_b;               // offsets in a chunk's partial (fragment layout): OB*KB blocks of 256, then OB*16 bias
};

struct FmPlan {
  int D, C, H, L, E, HB, DB, CB, EB, NL, P;
  float noise_scale, ln_eps, log_max_freq_over_E;
  FmLin lin[FM_MAX_LIN];
  int g_ln;                      // flat offset of layers_norm.0.weight (then bias, then layer 1 ...)
  int packed_floats;
  int lds_fwd_floats, lds_bwd_floats;   // size of ONE staging buffer (largest group, multiple of 256 floats)
  int nfg, nbg;                         // number of forward / backward staging groups
  int fgrp_off[FM_MAX_LIN], fgrp_floats[FM_MAX_LIN], bgrp_off[FM_MAX_LIN], bgrp_floats[FM_MAX_LIN];
  // stash slots in blocks of 256 floats, per wave-tile
  int s_in, s_c, s_te, s_ie, s_ce, s_h0, s_u, s_sh, s_rstd, s_diff;     // s_u + l*HB, s_sh + l*HB
  int g_v, g_u, g_te, g_h0, g_ie, g_ce;
  int SB;                        // blocks per wave-tile
  int PF;                        // floats per weight-gradient partial (fragment layout)
  int dw_order[FM_MAX_LIN];      // linears sorted by weight-gradient work (blocks), largest first
  int ablate;                    // timing experiments only (env SBI_AMD_FM_ABLATE): 1 no stash traffic,
                                 // 2 no weight staging, 4 no GELU, 8 no hidden-layer MFMAs; results invalid
};

static int round_up(int a, int m) { return (a + m - 1) / m * m; }

static int fm_build_plan(const sbi_amd_fmpe_config* cfg, FmPlan* pl) {
  memset(pl, 0, sizeof(*pl));
  const int D = cfg->D, C = cfg->C, H = cfg->H, L = cfg->L, E = cfg->E;
  if (D < 1 || D > 128 || C < 1 || C > 128 || H < 16 || H > 128 || L < 1 || L > FM_MAX_L || E < 2 || E > 64 ||
      (E & 1))
    return SBI_AMD_E_UNSUPPORTED;
  pl->D = D; pl->C = C; pl->H = H; pl->L = L; pl->E = E;
  pl->HB = H <= 64 ? 4 : (H <= 112 ? 7 : 8);
  pl->DB = (D + 15) / 16; pl->CB = (C + 15) / 16; pl->EB = (E + 15) / 16;
  pl->NL = 6 + L;
  pl->noise_scale = cfg->noise_scale; pl->ln_eps = cfg->ln_eps;
  pl->log_max_freq_over_E = logf(cfg->max_freq) / (float)E;
  const int HB = pl->HB;
  // ---- flat offsets
  int o = 0;
  auto lin = [&](int j, int out, int in, int ld, int col0, bool bias) {
    FmLin& l = pl->lin[j];
    l.out = out; l.in = in; l.OB = j == pl->NL - 1 ? pl->DB : HB;
    l.KB = j == J_IN ? pl->DB : j == J_CT ? pl->CB : j == J_TM ? pl->EB : HB;
    l.g_ld = ld; l.g_w = o + col0; l.g_b = -1; l.ln_fix = -1;
    if (bias) { l.g_b = o + out * ld; }
  };
  lin(J_IN, H, D, D, 0, true); o += H * D + H;
  lin(J_CT, H, C, C, 0, true); o += H * C + H;
  lin(J_MA, H, H, 2 * H, 0, true); lin(J_MB, H, H, 2 * H, H, false); o += 2 * H * H + H;
  lin(J_TM, H, E, E, 0, true); o += H * E + H;
  for (int l = 0; l < L; ++l) { lin(J_L0 + l, H, H, H, 0, true); o += H * H + H; }
  pl->g_ln = o; o += 2 * H * L;
  lin(J_L0 + L, D, H, H, 0, true); o += D * H + D;
  pl->P = o;
  // ---- stash slots
  int s = 0;
  pl->s_in = s; s += pl->DB; pl->s_c = s; s += pl->CB; pl->s_te = s; s += pl->EB;
  pl->s_ie = s; s += HB; pl->s_ce = s; s += HB; pl->s_h0 = s; s += HB;
  pl->s_u = s; s += L * HB; pl->s_sh = s; s += L * HB; pl->s_rstd = s; s += 1; pl->s_diff = s; s += pl->DB;
  pl->g_v = s; s += pl->DB; pl->g_u = s; s += L * HB; pl->g_te = s; s += HB; pl->g_h0 = s; s += HB;
  pl->g_ie = s; s += HB; pl->g_ce = s; s += HB;
  pl->SB = s;
  pl->lin[J_IN].s_x = pl->s_in; pl->lin[J_IN].s_g = pl->g_ie;
  pl->lin[J_CT].s_x = pl->s_c; pl->lin[J_CT].s_g = pl->g_ce;
  pl->lin[J_TM].s_x = pl->s_te; pl->lin[J_TM].s_g = pl->g_te;
  pl->lin[J_MA].s_x = pl->s_ie; pl->lin[J_MA].s_g = pl->g_h0; pl->lin[J_MA].x_gelu = 1;
  pl->lin[J_MB].s_x = pl->s_ce; pl->lin[J_MB].s_g = pl->g_h0; pl->lin[J_MB].x_gelu = 1;
  for (int l = 0; l < L; ++l) {
    FmLin& q = pl->lin[J_L0 + l];
    q.s_g = pl->g_u + l * HB;
    if (l == 0) { q.s_x = pl->s_h0; q.x_gelu = 1; } else { q.s_x = pl->s_sh + (l - 1) * HB; q.ln_fix = l - 1; }
  }
  pl->lin[J_L0 + L].s_x = pl->s_sh + (L - 1) * HB; pl->lin[J_L0 + L].s_g = pl->g_v; pl->lin[J_L0 + L].ln_fix = L - 1;
  // ---- packed images.  forward order: IN MA CT MB TM L0.. OUT; image = W[16*OB][ldk] + bias, gamma, beta [16*OB]
  const int order_n = pl->NL;
  int fo[FM_MAX_LIN];
  for (int j = 0; j < order_n; ++j) fo[j] = j;
  fo[0] = J_IN; fo[1] = J_MA; fo[2] = J_CT; fo[3] = J_MB; fo[4] = J_TM;   // execution order of the forward kernel
  int p = 0;
  {
    int goff = 0;
    for (int k = 0; k < order_n; ++k) {
      FmLin& l = pl->lin[fo[k]];
      l.ldk = 16 * l.KB + 4;
      const int sz = round_up(16 * l.OB * l.ldk + 3 * 16 * l.OB, 4);
      if (sz > FM_FWD_GROUP_FLOATS) return SBI_AMD_E_LDS;
      if (k == 0 || p + sz - goff > FM_FWD_GROUP_FLOATS) {
        if (k) {   // close the previous group: 1 KB granules (one global_load_lds per wave and granule)
          p = goff + round_up(p - goff, 256);
          pl->fgrp_floats[pl->nfg - 1] = p - goff;
        }
        goff = p; l.fg_first = 1; pl->fgrp_off[pl->nfg++] = p;
      }
      l.w_off = p; l.lw = p - goff; l.lb = l.lw + 16 * l.OB * l.ldk;
      p += sz;
    }
    p = goff + round_up(p - goff, 256);
    pl->fgrp_floats[pl->nfg - 1] = p - goff;
    for (int k = 0; k < pl->nfg; ++k)
      if (pl->fgrp_floats[k] > pl->lds_fwd_floats) pl->lds_fwd_floats = pl->fgrp_floats[k];
  }
  // backward order: OUT L(L-1) .. L0 MA MB; image = W^T[16*KB][ldt] + gamma [16*OB]
  {
    int bo[FM_MAX_LIN], nb = 0;
    bo[nb++] = J_L0 + L;
    for (int l = L - 1; l >= 0; --l) bo[nb++] = J_L0 + l;
    bo[nb++] = J_MA; bo[nb++] = J_MB;
    int goff = p;
    for (int k = 0; k < nb; ++k) {
      FmLin& l = pl->lin[bo[k]];
      l.ldt = 16 * l.OB + 4;
      const int sz = round_up(16 * l.KB * l.ldt + 16 * l.OB, 4);
      if (sz > FM_BWD_GROUP_FLOATS) return SBI_AMD_E_LDS;
      if (k == 0 || p + sz - goff > FM_BWD_GROUP_FLOATS) {
        if (k) {
          p = goff + round_up(p - goff, 256);
          pl->bgrp_floats[pl->nbg - 1] = p - goff;
        }
        goff = p; l.bg_first = 1; pl->bgrp_off[pl->nbg++] = p;
      }
      l.t_off = p; l.ltw = p - goff; l.ltg = l.ltw + 16 * l.KB * l.ldt;
      p += sz;
    }
    p = goff + round_up(p - goff, 256);
    pl->bgrp_floats[pl->nbg - 1] = p - goff;
    for (int k = 0; k < pl->nbg; ++k)
      if (pl->bgrp_floats[k] > pl->lds_bwd_floats) pl->lds_bwd_floats = pl->bgrp_floats[k];
  }
  pl->packed_floats = p;
  for (int j = 0; j < pl->NL; ++j) pl->dw_order[j] = j;
  for (int a_ = 0; a_ < pl->NL; ++a_)
    for (int b_ = a_ + 1; b_ < pl->NL; ++b_) {
      const FmLin& la = pl->lin[pl->dw_order[a_]];
      const FmLin& lb = pl->lin[pl->dw_order[b_]];
      if (lb.OB * lb.KB > la.OB * la.KB) {
        const int t_ = pl->dw_order[a_]; pl->dw_order[a_] = pl->dw_order[b_]; pl->dw_order[b_] = t_;
      }
    }
  for (int j = 0; j < pl->NL; ++j) {
    FmLin& l = pl->lin[j];
    l.pf_w = pl->PF; pl->PF += l.OB * l.KB * 256;
    l.pf_b = pl->PF; pl->PF += l.OB * 16;
  }
  pl->ablate = sbi_amd_dbg_fm_ablate();
  return 0;
}

// ---------------------------------------------------------------- device helpers
// GELU(v) = v Phi(v) and its derivative Phi(v) + v phi(v), with Phi from the Abramowitz-Stegun 7.1.26 erfc
// (|error| < 1.5e-7 on erf, i.e. < 1e-7 |v| on GELU): one v_exp_f32, one v_rcp_f32 and a degree-5 Horner chain
// instead of libm's erff.  The tail side is computed without cancellation (Phi(v) = erfc(|x|)/2 for v < 0).
__device__ __forceinline__ void gelu_core(float v, float& cdf, float& ex) {
  const float x = fabsf(v) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * x);
  ex = __expf(-x * x);                           // = exp(-v^2 / 2)
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
  const float half_erfc = 0.5f * p * t * ex;     // erfc(|x|) / 2
  cdf = v < 0.f ? half_erfc : 1.0f - half_erfc;
}
__device__ __forceinline__ float gelu_f(float v) {
  float cdf, ex;
  gelu_core(v, cdf, ex);
  return v * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float v) {
  float cdf, ex;
  gelu_core(v, cdf, ex);
  return cdf + v * 0.3989422804014327f * ex;
}
__device__ __forceinline__ f4 gelu4(f4 v) { return f4{gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3])}; }
__device__ __forceinline__ f4 gelu_grad4(f4 v) {
  return f4{gelu_grad_f(v[0]), gelu_grad_f(v[1]), gelu_grad_f(v[2]), gelu_grad_f(v[3])};
}

// Asynchronous weight staging: global_load_lds_dwordx4 writes 1 KB per wave instruction straight into LDS
// (destination = wave-uniform base + lane * 16 B), no registers involved.  Group images are padded to 1 KB.
__device__ __forceinline__ void fm_stage_async(float* __restrict__ lds_dst, const float* __restrict__ src, int floats,
                                               int wave, int lane) {
  const int nch = floats >> 8;
  for (int ch = wave; ch < nch; ch += FM_WAVES)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ch * 256 + lane * 4),
                                     (__attribute__((address_space(3))) void*)(lds_dst + ch * 256), 16, 0, 0);
}
// Staging pipeline shared by the forward and backward kernels: two LDS buffers; while the linears of the group
// in one buffer are computed, the next group (cyclically: the first group of the next tile after the last) is
// in flight into the other.  One workgroup barrier per group.
struct FmPipe {
  float* lds;
  const float* packed;
  const int* goff;
  const int* gfloats;
  int ngroups, buf_floats, cur, par, wave, lane, started, off;
  __device__ __forceinline__ void prefetch_next() {
    if (off) return;
    const int nxt = cur + 1 == ngroups ? 0 : cur + 1;
    fm_stage_async(lds + (par ^ 1) * buf_floats, packed + goff[nxt], gfloats[nxt], wave, lane);
  }
  __device__ __forceinline__ void init(float* lds_, const float* packed_, const int* goff_, const int* gfloats_,
                                       int ngroups_, int buf_floats_, int wave_, int lane_, int off_) {
    lds = lds_; packed = packed_; goff = goff_; gfloats = gfloats_; ngroups = ngroups_; buf_floats = buf_floats_;
    cur = 0; par = 0; wave = wave_; lane = lane_; started = 0; off = off_;
    fm_stage_async(lds, packed + goff[0], gfloats[0], wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    prefetch_next();
  }
  // Called before a linear that opens a group, in two halves so that a kernel can consume values it loaded
  // into registers during the previous stage BETWEEN them: after enter_wait nothing is outstanding (the
  // compiler's own conservative vmcnt(0) at the first use costs nothing); once enter_prefetch has issued the
  // next group's global_load_lds, any ordinary load result would have to wait for those as well.
  int pending;
  __device__ __forceinline__ void enter_wait(int opens_group) {
    pending = 0;
    if (opens_group) {
      if (started) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next group has landed ...
        __syncthreads();                                   // ... for every wave, and all are done with this one
        cur = cur + 1 == ngroups ? 0 : cur + 1;
        par ^= 1;
        pending = 1;
      }
      started = 1;
    }
  }
  __device__ __forceinline__ void enter_prefetch() {
    if (pending) prefetch_next();
    pending = 0;
  }
  __device__ __forceinline__ void enter(int opens_group) {
    enter_wait(opens_group);
    enter_prefetch();
  }
  __device__ __forceinline__ const float* base() const { return lds + par * buf_floats; }
  __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

// stash blocks: natural [row][16 feats] (one b128 per lane) and transposed [feat][row]
__device__ __forceinline__ void st_nat(float* __restrict__ wtb, int blk, int c, int g, f4 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<f4*>(wtb + blk * 256 + c * 16 + 4 * g));
}
__device__ __forceinline__ f4 ld_nat(const float* __restrict__ wtb, int blk, int c, int g) {
  return __builtin_nontemporal_load(reinterpret_cast<const f4*>(wtb + blk * 256 + c * 16 + 4 * g));
}
__device__ __forceinline__ void st_tr(float* __restrict__ wtb, int blk, int c, int g, f4 v) {
  float* p = wtb + blk * 256 + (4 * g) * 16 + c;
  p[0] = v[0]; p[16] = v[1]; p[32] = v[2]; p[48] = v[3];
}

// acc[ob] += W[16*ob.., 16*kb..] * (one 16-feature block of B held as f4)
template <int OB>
__device__ __forceinline__ void gemm_blk(const float* __restrict__ wl /* lds + lw + c*ld + 4*g */, int ld, int kb,
                                         f4 b, f4 (&acc)[OB]) {
  f4 a[OB];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) a[ob] = *reinterpret_cast<const f4*>(wl + ob * 16 * ld + 16 * kb);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) acc[ob] = MFMA16(a[ob][r], b[r], acc[ob]);
}

// acc[ob] += W * B for B = KB register blocks; A fragments of block kb+1 are loaded under the MFMAs of block kb
template <int OB, int KB>
__device__ __forceinline__ void gemm_rr(const float* __restrict__ wl, int ld, const f4 (&b)[KB], f4 (&acc)[OB]) {
  f4 a[2][OB];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) a[0][ob] = *reinterpret_cast<const f4*>(wl + ob * 16 * ld);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 1 < KB) {
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
        a[(kb + 1) & 1][ob] = *reinterpret_cast<const f4*>(wl + ob * 16 * ld + 16 * (kb + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) acc[ob] = MFMA16(a[kb & 1][ob][r], b[kb][r], acc[ob]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// same, with a callback after the MFMAs of each K block (b[kb] is dead from then on: the caller may refill it)
template <int OB, int KB, class F>
__device__ __forceinline__ void gemm_rr_cb(const float* __restrict__ wl, int ld, f4 (&b)[KB], f4 (&acc)[OB], F after_kb) {
  f4 a[2][OB];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) a[0][ob] = *reinterpret_cast<const f4*>(wl + ob * 16 * ld);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 1 < KB) {
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
        a[(kb + 1) & 1][ob] = *reinterpret_cast<const f4*>(wl + ob * 16 * ld + 16 * (kb + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) acc[ob] = MFMA16(a[kb & 1][ob][r], b[kb][r], acc[ob]);
    __builtin_amdgcn_sched_barrier(0);
    after_kb(kb);
  }
}

__device__ __forceinline__ float sum_over_g(float v) {   // lanes c, c+16, c+32, c+48
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
// sum over the 16 lanes of one g (= one DPP row): four row rotations on the VALU (v_add_f32 ... row_ror:n)
// instead of four ds_bpermute round trips through the LDS pipeline; every lane ends with the total
template <int CTRL>
__device__ __forceinline__ float dpp_rot(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sum_over_c(float v) {
  v += dpp_rot<0x128>(v);   // row_ror:8
  v += dpp_rot<0x124>(v);   // row_ror:4
  v += dpp_rot<0x122>(v);   // row_ror:2
  v += dpp_rot<0x121>(v);   // row_ror:1
  return v;
}

struct FmArgs {
  const float* packed; const float* zstats; const float* theta; const float* x; const float* times;
  const float* noise; const float* row_weight; float uniform_weight;
  long long n; int x_rows, t_rows;
  float* loss_out; float* v_out; float* stash; float* ln_part; float* div_out;
  int ntiles;
  long long* timeline;   // debug (env SBI_AMD_FM_TIMELINE): s_memtime stamps of workgroup 0, wave 0
};

// LDS tail after the weight group: mean_0[D] std_0[D] vstd[D] xmean[C] xinv[C] (floats)
#define FM_ZS_FLOATS (3 * 128 + 2 * 128)

// ---------------------------------------------------------------- forward
// MODE 0: velocity, 1: loss only, 2: loss + stash (training)
template <int HB, int MODE>
__global__ void __launch_bounds__(FM_THREADS, 1) fm_fwd_kernel(const FmPlan pl, const FmArgs a) {
  extern __shared__ __align__(16) float lds[];
  float* zs = lds + 2 * pl.lds_fwd_floats;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int D = pl.D, C = pl.C, H = pl.H;
  float* z_mean = zs; float* z_std = zs + 128; float* z_vstd = zs + 256; float* z_xm = zs + 384; float* z_xi = zs + 512;
  for (int i = tid; i < 128; i += FM_THREADS) {
    const float m = i < D ? a.zstats[i] : 0.f, s = i < D ? a.zstats[D + i] : 1.f;
    z_mean[i] = m; z_std[i] = s; z_vstd[i] = sqrtf(1.0f + s * s);
    z_xm[i] = i < C ? a.zstats[2 * D + i] : 0.f;
    z_xi[i] = i < C ? 1.0f / a.zstats[2 * D + C + i] : 0.f;
  }
  const float invH = 1.0f / (float)H;
  FmPipe pipe;
  pipe.init(lds, a.packed, pl.fgrp_off, pl.fgrp_floats, pl.nfg, pl.lds_fwd_floats, wave, lane, NSF_DBG_ABL(pl.ablate, 2));
  const float* wb = lds;
  int titer = -1;
  // theta / noise / t of the first four blocks travel one tile ahead (loaded during the previous tile's output
  // stage); x one stage ahead
  f4 thv[4], nzv[4], blk4[4];
  float t_pref = 0.f;
  auto row_of = [&](int tile_) {
    const long long r = ((long long)tile_ * FM_WAVES + wave) * 16 + c;
    return r < a.n ? r : a.n - 1;
  };
  auto load_theta_noise = [&](long long row_, f4 (&tv)[4], f4 (&nv)[4]) {
    const float* th_ = a.theta + row_ * D;
    const float* nz_ = MODE == 0 ? nullptr : a.noise + row_ * D;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = 16 * kb + 4 * g + i;
        tv[kb][i] = f < D ? th_[f] : 0.f;
        nv[kb][i] = (MODE != 0 && f < D) ? nz_[f] : 0.f;
      }
    }
  };
  if ((int)blockIdx.x < a.ntiles) {
    const long long r0 = row_of(blockIdx.x);
    load_theta_noise(r0, thv, nzv);
    t_pref = a.times[a.t_rows == 1 ? 0 : r0];
  }
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    ++titer;
    const long long wt = (long long)tile * FM_WAVES + wave;
    const long long row_raw = wt * 16 + c;
    const bool valid = row_raw < a.n;
    const long long row = valid ? row_raw : a.n - 1;
    float* wtb = MODE == 2 ? a.stash + wt * (long long)pl.SB * 256 : nullptr;
    const float t = t_pref;
    const float om = 1.0f - t;
    const float* th = a.theta + row * D;
    const float* nz = MODE == 0 ? nullptr : a.noise + row * D;
    const float* xr = a.x + (a.x_rows == 1 ? 0 : row) * C;

#define FM_TS(K) if (a.timeline && blockIdx.x == 0 && tid == 0) a.timeline[titer * 32 + (K)] = __builtin_readcyclecounter();
#define FM_ENTER(J)                 \
  {                                 \
    pipe.enter(pl.lin[J].fg_first); \
    wb = pipe.base();               \
  }
    f4 acc[HB], temb[HB], h[HB];
    auto in_block = [&](int kb, const f4& tv, const f4& nv) {
      f4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = 16 * kb + 4 * g + i;
        float val = 0.f;
        if (f < D) {
          float tt = tv[i];
          if (MODE != 0) tt = om * tt + (t + pl.noise_scale) * nv[i];
          const float sd = om * z_std[f];
          val = (tt - om * z_mean[f]) / sqrtf(sd * sd + t * t + 1e-6f);
        }
        v[i] = val;
      }
      return v;
    };
    // ---- input layer: theta_t -> time-dependent z-score -> Linear(D, H); then the first half of the merge
    FM_TS(0);
    pipe.enter_wait(pl.lin[J_IN].fg_first);
    wb = pipe.base();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) blk4[kb] = in_block(kb, thv[kb], nzv[kb]);
    pipe.enter_prefetch();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {   // x of this tile: consumed after the next stage entry
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = 16 * kb + 4 * g + i;
        thv[kb][i] = f < C ? xr[f] : 0.f;
      }
    }
    FM_TS(1);
    {
      const FmLin& q = pl.lin[J_IN];
      f4 ie[HB];
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) ie[ob] = *reinterpret_cast<const f4*>(wb + q.lb + 16 * ob + 4 * g);
      const float* wl = wb + q.lw + c * q.ldk + 4 * g;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        if (kb < pl.DB) {
          if (MODE == 2 && !NSF_DBG_ABL(pl.ablate, 1)) st_nat(wtb, pl.s_in + kb, c, g, blk4[kb]);
          gemm_blk<HB>(wl, q.ldk, kb, blk4[kb], ie);
        }
      }
      for (int kb = 4; kb < pl.DB; ++kb) {
        f4 tv, nv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = 16 * kb + 4 * g + i;
          tv[i] = f < D ? th[f] : 0.f;
          nv[i] = (MODE != 0 && f < D) ? nz[f] : 0.f;
        }
        const f4 v = in_block(kb, tv, nv);
        if (MODE == 2 && !NSF_DBG_ABL(pl.ablate, 1)) st_nat(wtb, pl.s_in + kb, c, g, v);
        gemm_blk<HB>(wl, q.ldk, kb, v, ie);
      }
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) {
        if (MODE == 2 && !NSF_DBG_ABL(pl.ablate, 1)) st_nat(wtb, pl.s_ie + ob, c, g, ie[ob]);
        h[ob] = gelu4(ie[ob]);
      }
    }
    // merge: Linear(2H, H) on GELU([ie, ce]) as two K = H products
    FM_TS(2);
    pipe.enter_wait(pl.lin[J_MA].fg_first);
    wb = pipe.base();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = 16 * kb + 4 * g + i;
        blk4[kb][i] = f < C ? (thv[kb][i] - z_xm[f]) * z_xi[f] : 0.f;
      }
    }
    pipe.enter_prefetch();
    FM_TS(3);
    {
      const FmLin& q = pl.lin[J_MA];
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) acc[ob] = *reinterpret_cast<const f4*>(wb + q.lb + 16 * ob + 4 * g);
      gemm_rr<HB, HB>(wb + q.lw + c * q.ldk + 4 * g, q.ldk, h, acc);
    }
    // ---- condition layer: standardised x -> Linear(C, H); second half of the merge
    FM_TS(4);
    FM_ENTER(J_CT);
    FM_TS(5);
    {
      const FmLin& q = pl.lin[J_CT];
      f4 ce[HB];
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) ce[ob] = *reinterpret_cast<const f4*>(wb + q.lb + 16 * ob + 4 * g);
      const float* wl = wb + q.lw + c * q.ldk + 4 * g;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        if (kb < pl.CB) {
          if (MODE == 2 && !NSF_DBG_ABL(pl.ablate, 1)) st_nat(wtb, pl.s_c + kb, c, g, blk4[kb]);
          gemm_blk<HB>(wl, q.ldk, kb, blk4[kb], ce);
        }
      }
      for (int kb = 4; kb < pl.CB; ++kb) {
        f4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = 16 * kb + 4 * g + i;
          v[i] = f < C ? (xr[f] - z_xm[f]) * z_xi[f] : 0.f;
        }
        if (MODE == 2 && !NSF_DBG_ABL(pl.ablate, 1)) st_nat(wtb, pl.s_c + kb, c, g, v);
        gemm_blk<HB>(wl, q.ldk, kb, v, ce);
      }
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) {
        if (MODE == 2 && !NSF_DBG_ABL(pl.ablate, 1)) st_nat(wtb, pl.s_ce + ob, c, g, ce[ob]);
        h[ob] = gelu4(ce[ob]);
      }
    }
    FM_TS(6);
    FM_ENTER(J_MB);
    FM_TS(7);
    {
      const FmLin& q = pl.lin[J_MB];
      gemm_rr<HB, HB>(wb + q.lw + c * q.ldk + 4 * g, q.ldk, h, acc);
    }
#pragma unroll
    for (int ob = 0; ob < HB; ++ob) {
      if (MODE == 2 && !NSF_DBG_ABL(pl.ablate, 1)) st_nat(wtb, pl.s_h0 + ob, c, g, acc[ob]);
      h[ob] = gelu4(acc[ob]);
    }
    // ---- time embedding: sin/cos features -> Linear(E, H)
    FM_TS(8);
    FM_ENTER(J_TM);
    FM_TS(9);
    {
      const FmLin& q = pl.lin[J_TM];
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) temb[ob] = *reinterpret_cast<const f4*>(wb + q.lb + 16 * ob + 4 * g);
      const float* wl = wb + q.lw + c * q.ldk + 4 * g;
      for (int kb = 0; kb < pl.EB; ++kb) {
        f4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = 16 * kb + 4 * g + i;
          float val = 0.f;
          if (e < pl.E) {
            const float w = expf(-(float)(e & ~1) * pl.log_max_freq_over_E);
            const float ang = t * w;
            val = (e & 1) ? cosf(ang) : sinf(ang);
          }
          v[i] = val;
        }
        if (MODE == 2 && !NSF_DBG_ABL(pl.ablate, 1)) st_nat(wtb, pl.s_te + kb, c, g, v);
        gemm_blk<HB>(wl, q.ldk, kb, v, temb);
      }
    }
    // ---- residual blocks: h <- LayerNorm(GELU(W h + b) + temb + h)
    for (int l = 0; l < pl.L; ++l) {
      FM_TS(10 + 2 * l);
      FM_ENTER(J_L0 + l);
      FM_TS(11 + 2 * l);
      const FmLin& q = pl.lin[J_L0 + l];
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) acc[ob] = *reinterpret_cast<const f4*>(wb + q.lb + 16 * ob + 4 * g);
      if (!NSF_DBG_ABL(pl.ablate, 8)) gemm_rr<HB, HB>(wb + q.lw + c * q.ldk + 4 * g, q.ldk, h, acc);
      float s1 = 0.f;
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) {
        if (MODE == 2 && !NSF_DBG_ABL(pl.ablate, 1)) st_nat(wtb, pl.s_u + l * HB + ob, c, g, acc[ob]);
        acc[ob] = (NSF_DBG_ABL(pl.ablate, 4) ? acc[ob] : gelu4(acc[ob])) + temb[ob] + h[ob];
        s1 += (acc[ob][0] + acc[ob][1]) + (acc[ob][2] + acc[ob][3]);
      }
      const float mu = sum_over_g(s1) * invH;
      float s2 = 0.f;
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d = (16 * ob + 4 * g + i) < H ? acc[ob][i] - mu : 0.f;
          acc[ob][i] = d;
          s2 += d * d;
        }
      }
      const float rstd = 1.0f / sqrtf(sum_over_g(s2) * invH + pl.ln_eps);
      if (MODE == 2 && g == 0) wtb[pl.s_rstd * 256 + l * 16 + c] = rstd;
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) {
        const f4 sh = acc[ob] * rstd;
        if (MODE == 2 && !NSF_DBG_ABL(pl.ablate, 1)) st_nat(wtb, pl.s_sh + l * HB + ob, c, g, sh);
        const f4 gam = *reinterpret_cast<const f4*>(wb + q.lb + 16 * HB + 16 * ob + 4 * g);
        const f4 bet = *reinterpret_cast<const f4*>(wb + q.lb + 32 * HB + 16 * ob + 4 * g);
        h[ob] = sh * gam + bet;
      }
    }
    // ---- output layer + loss / velocity
    FM_TS(10 + 2 * pl.L);
    if (MODE != 0) load_theta_noise(row, thv, nzv);   // L2 hits: the same rows were read for the input stage
    pipe.enter_wait(pl.lin[J_L0 + pl.L].fg_first);
    wb = pipe.base();
    if (MODE != 0) {   // normalised velocity targets of the prefetched blocks
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = 16 * kb + 4 * g + i;
          blk4[kb][i] = f < D ? ((nzv[kb][i] - thv[kb][i]) + z_mean[f]) / z_vstd[f] : 0.f;
        }
      }
    }
    pipe.enter_prefetch();
    if (tile + (int)gridDim.x < a.ntiles) {   // next tile's theta / noise / t
      const long long rn = row_of(tile + gridDim.x);
      load_theta_noise(rn, thv, nzv);
      t_pref = a.times[a.t_rows == 1 ? 0 : rn];
    }
    FM_TS(11 + 2 * pl.L);
    {
      const FmLin& q = pl.lin[J_L0 + pl.L];
      const float* wl = wb + q.lw + c * q.ldk + 4 * g;
      float lsum = 0.f;
      for (int ob = 0; ob < pl.DB; ++ob) {
        f4 o0 = *reinterpret_cast<const f4*>(wb + q.lb + 16 * ob + 4 * g), o1 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < HB; ++kb) {
          const f4 av = *reinterpret_cast<const f4*>(wl + ob * 16 * q.ldk + 16 * kb);
          if (kb & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o1 = MFMA16(av[r], h[kb][r], o1);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) o0 = MFMA16(av[r], h[kb][r], o0);
          }
        }
        o0 += o1;
        f4 diff;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = 16 * ob + 4 * g + i;
          if (MODE == 0) {
            if (f < D && valid) a.v_out[row * D + f] = o0[i] * z_vstd[f] - z_mean[f];
          } else {
            float d = 0.f;
            if (f < D) {
              float tgt;
              switch (ob) {   // uniform: ob is a loop counter
                case 0: tgt = blk4[0][i]; break;
                case 1: tgt = blk4[1][i]; break;
                case 2: tgt = blk4[2][i]; break;
                case 3: tgt = blk4[3][i]; break;
                default: tgt = ((nz[f] - th[f]) + z_mean[f]) / z_vstd[f];
              }
              d = o0[i] - tgt;
            }
            diff[i] = d;
            lsum += d * d;
          }
        }
        if (MODE == 2 && !NSF_DBG_ABL(pl.ablate, 1)) st_nat(wtb, pl.s_diff + ob, c, g, diff);
      }
      if (MODE != 0) {
        lsum = sum_over_g(lsum);
        if (g == 0 && valid) a.loss_out[row] = lsum / (float)D;
      }
    }
    FM_TS(12 + 2 * pl.L);
  }
  pipe.drain();
#undef FM_ENTER
#undef FM_TS
}

// ---------------------------------------------------------------- velocity + divergence (log_prob of the flow)
// d v_f / d theta_f summed over f, exactly (what zuko's FreeFormJacobianTransform computes with a batched autograd
// identity for sbi's VectorFieldPosterior.log_prob, samplers/ode_solvers/zuko_ode.py:19-124 with exact=True), by
// FORWARD-mode propagation: the 16 columns of a wave's MFMA tile are not 16 batch rows but ONE row's primal
// (column 15) next to the tangents of 15 input directions (column c <-> direction 15 chunk + c).  Every linear is
// the same W X^T product for all columns (bias / time embedding on the primal column only); GELU and LayerNorm
// need the primal's values in every column, so each lane keeps a copy of the primal's activations of ITS features
// (hp), refreshed from column 15 by one ds_bpermute per value and layer:
//     GELU:       a_p = u_p Phi(u_p)                     a_tau = (Phi(u_p) + u_p phi(u_p)) u_tau
//     LayerNorm:  h_p = gamma s + beta, s = (y_p - mu) r   h_tau = gamma r (y_tau - mean(y_tau) - s mean(s y_tau))
// A row with more than 15 theta dims takes ceil(D / 15) passes (the primal is recomputed in each).  One workgroup =
// 8 waves = 8 rows per pass; the weight groups stream through LDS exactly as in the forward kernel.
#define FM_DIV_PC 15
__device__ __forceinline__ f4 bcast_primal(f4 v, int g) {
  const int src = 16 * g + FM_DIV_PC;
  return f4{__shfl(v[0], src), __shfl(v[1], src), __shfl(v[2], src), __shfl(v[3], src)};
}
// primal column: GELU(up); tangent columns: GELU'(up) * u; `ap` receives GELU(up) for every lane
__device__ __forceinline__ f4 gelu_primal_tangent(f4 up, f4 u, bool primal, f4& ap) {
  f4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float cdf, ex;
    gelu_core(up[i], cdf, ex);
    ap[i] = up[i] * cdf;
    out[i] = primal ? ap[i] : (cdf + up[i] * 0.3989422804014327f * ex) * u[i];
  }
  return out;
}

template <int HB>
__global__ void __launch_bounds__(FM_THREADS, 1) fm_div_kernel(const FmPlan pl, const FmArgs a) {
  extern __shared__ __align__(16) float lds[];
  float* zs = lds + 2 * pl.lds_fwd_floats;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int D = pl.D, C = pl.C, H = pl.H;
  const bool primal = c == FM_DIV_PC;
  float* z_mean = zs; float* z_std = zs + 128; float* z_vstd = zs + 256; float* z_xm = zs + 384; float* z_xi = zs + 512;
  for (int i = tid; i < 128; i += FM_THREADS) {
    const float m = i < D ? a.zstats[i] : 0.f, s = i < D ? a.zstats[D + i] : 1.f;
    z_mean[i] = m; z_std[i] = s; z_vstd[i] = sqrtf(1.0f + s * s);
    z_xm[i] = i < C ? a.zstats[2 * D + i] : 0.f;
    z_xi[i] = i < C ? 1.0f / a.zstats[2 * D + C + i] : 0.f;
  }
  const float invH = 1.0f / (float)H;
  const int nchunks = (D + FM_DIV_PC - 1) / FM_DIV_PC;
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  FmPipe pipe;
  pipe.init(lds, a.packed, pl.fgrp_off, pl.fgrp_floats, pl.nfg, pl.lds_fwd_floats, wave, lane, 0);
  const float* wb = lds;
#define FM_ENTER(J)                 \
  {                                 \
    pipe.enter(pl.lin[J].fg_first); \
    wb = pipe.base();               \
  }
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const long long row_raw = (long long)tile * FM_WAVES + wave;
    const bool valid = row_raw < a.n;
    const long long row = valid ? row_raw : a.n - 1;
    const float t = a.times[a.t_rows == 1 ? 0 : row];
    const float om = 1.0f - t;
    const float* th = a.theta + row * D;
    const float* xr = a.x + (a.x_rows == 1 ? 0 : row) * C;
    float div = 0.f;
    for (int ch = 0; ch < nchunks; ++ch) {
      const int dir = FM_DIV_PC * ch + c;      // this column's input direction (tangent columns)
      f4 acc[HB], temb[HB], h[HB], hp[HB];
      // ---- input layer: primal = time-dependent z-score of theta_t; tangent of direction f = e_f / scale_f
      FM_ENTER(J_IN);
      {
        const FmLin& q = pl.lin[J_IN];
        f4 ie[HB];
#pragma unroll
        for (int ob = 0; ob < HB; ++ob)
          ie[ob] = primal ? *reinterpret_cast<const f4*>(wb + q.lb + 16 * ob + 4 * g) : zero4;
        const float* wl = wb + q.lw + c * q.ldk + 4 * g;
        for (int kb = 0; kb < pl.DB; ++kb) {
          f4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int f = 16 * kb + 4 * g + i;
            float val = 0.f;
            if (f < D) {
              const float sd = om * z_std[f];
              const float inv = 1.0f / sqrtf(sd * sd + t * t + 1e-6f);
              val = primal ? (th[f] - om * z_mean[f]) * inv : (f == dir ? inv : 0.f);
            }
            v[i] = val;
          }
          gemm_blk<HB>(wl, q.ldk, kb, v, ie);
        }
#pragma unroll
        for (int ob = 0; ob < HB; ++ob) {
          f4 ap;
          h[ob] = gelu_primal_tangent(bcast_primal(ie[ob], g), ie[ob], primal, ap);
        }
      }
      // ---- merge, first half (theta embedding)
      FM_ENTER(J_MA);
      {
        const FmLin& q = pl.lin[J_MA];
#pragma unroll
        for (int ob = 0; ob < HB; ++ob)
          acc[ob] = primal ? *reinterpret_cast<const f4*>(wb + q.lb + 16 * ob + 4 * g) : zero4;
        gemm_rr<HB, HB>(wb + q.lw + c * q.ldk + 4 * g, q.ldk, h, acc);
      }
      // ---- condition layer: no theta dependence, the tangent columns stay zero
      FM_ENTER(J_CT);
      {
        const FmLin& q = pl.lin[J_CT];
        f4 ce[HB];
#pragma unroll
        for (int ob = 0; ob < HB; ++ob)
          ce[ob] = primal ? *reinterpret_cast<const f4*>(wb + q.lb + 16 * ob + 4 * g) : zero4;
        const float* wl = wb + q.lw + c * q.ldk + 4 * g;
        for (int kb = 0; kb < pl.CB; ++kb) {
          f4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int f = 16 * kb + 4 * g + i;
            v[i] = (primal && f < C) ? (xr[f] - z_xm[f]) * z_xi[f] : 0.f;
          }
          gemm_blk<HB>(wl, q.ldk, kb, v, ce);
        }
#pragma unroll
        for (int ob = 0; ob < HB; ++ob) h[ob] = primal ? gelu4(ce[ob]) : zero4;
      }
      FM_ENTER(J_MB);
      {
        const FmLin& q = pl.lin[J_MB];
        gemm_rr<HB, HB>(wb + q.lw + c * q.ldk + 4 * g, q.ldk, h, acc);
      }
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) h[ob] = gelu_primal_tangent(bcast_primal(acc[ob], g), acc[ob], primal, hp[ob]);
      // ---- time embedding (the same for every column: one row per wave)
      FM_ENTER(J_TM);
      {
        const FmLin& q = pl.lin[J_TM];
#pragma unroll
        for (int ob = 0; ob < HB; ++ob) temb[ob] = *reinterpret_cast<const f4*>(wb + q.lb + 16 * ob + 4 * g);
        const float* wl = wb + q.lw + c * q.ldk + 4 * g;
        for (int kb = 0; kb < pl.EB; ++kb) {
          f4 v;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = 16 * kb + 4 * g + i;
            float val = 0.f;
            if (e < pl.E) {
              const float w = expf(-(float)(e & ~1) * pl.log_max_freq_over_E);
              const float ang = t * w;
              val = (e & 1) ? cosf(ang) : sinf(ang);
            }
            v[i] = val;
          }
          gemm_blk<HB>(wl, q.ldk, kb, v, temb);
        }
      }
      // ---- residual blocks: h <- LayerNorm(GELU(W h + b) + temb + h), primal and tangents
      for (int l = 0; l < pl.L; ++l) {
        FM_ENTER(J_L0 + l);
        const FmLin& q = pl.lin[J_L0 + l];
#pragma unroll
        for (int ob = 0; ob < HB; ++ob)
          acc[ob] = primal ? *reinterpret_cast<const f4*>(wb + q.lb + 16 * ob + 4 * g) : zero4;
        gemm_rr<HB, HB>(wb + q.lw + c * q.ldk + 4 * g, q.ldk, h, acc);
        float s1 = 0.f, sp1 = 0.f;
#pragma unroll
        for (int ob = 0; ob < HB; ++ob) {
          f4 ap;
          const f4 at = gelu_primal_tangent(bcast_primal(acc[ob], g), acc[ob], primal, ap);
          hp[ob] = ap + temb[ob] + hp[ob];                          // y_p (every lane, its features)
          acc[ob] = primal ? hp[ob] : at + h[ob];                   // y of this column
          sp1 += (hp[ob][0] + hp[ob][1]) + (hp[ob][2] + hp[ob][3]);
          s1 += (acc[ob][0] + acc[ob][1]) + (acc[ob][2] + acc[ob][3]);
        }
        const float mu = sum_over_g(sp1) * invH;
        const float m1 = sum_over_g(s1) * invH;
        float s2 = 0.f;
#pragma unroll
        for (int ob = 0; ob < HB; ++ob)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float d = (16 * ob + 4 * g + i) < H ? hp[ob][i] - mu : 0.f;
            hp[ob][i] = d;
            s2 += d * d;
          }
        const float rstd = 1.0f / sqrtf(sum_over_g(s2) * invH + pl.ln_eps);
        float s3 = 0.f;
#pragma unroll
        for (int ob = 0; ob < HB; ++ob) {
          hp[ob] = hp[ob] * rstd;                                    // s
          s3 += (hp[ob][0] * acc[ob][0] + hp[ob][1] * acc[ob][1]) + (hp[ob][2] * acc[ob][2] + hp[ob][3] * acc[ob][3]);
        }
        const float m2 = sum_over_g(s3) * invH;
#pragma unroll
        for (int ob = 0; ob < HB; ++ob) {
          const f4 gam = *reinterpret_cast<const f4*>(wb + q.lb + 16 * HB + 16 * ob + 4 * g);
          const f4 bet = *reinterpret_cast<const f4*>(wb + q.lb + 32 * HB + 16 * ob + 4 * g);
          f4 ht;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            ht[i] = (16 * ob + 4 * g + i) < H ? gam[i] * rstd * (acc[ob][i] - m1 - hp[ob][i] * m2) : 0.f;
          hp[ob] = hp[ob] * gam + bet;
          h[ob] = primal ? hp[ob] : ht;
        }
      }
      // ---- output layer: velocity from the primal column, the Jacobian diagonal from the tangent columns
      FM_ENTER(J_L0 + pl.L);
      {
        const FmLin& q = pl.lin[J_L0 + pl.L];
        const float* wl = wb + q.lw + c * q.ldk + 4 * g;
        for (int ob = 0; ob < pl.DB; ++ob) {
          f4 o0 = primal ? *reinterpret_cast<const f4*>(wb + q.lb + 16 * ob + 4 * g) : zero4, o1 = zero4;
#pragma unroll
          for (int kb = 0; kb < HB; ++kb) {
            const f4 av = *reinterpret_cast<const f4*>(wl + ob * 16 * q.ldk + 16 * kb);
            if (kb & 1) {
#pragma unroll
              for (int r = 0; r < 4; ++r) o1 = MFMA16(av[r], h[kb][r], o1);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) o0 = MFMA16(av[r], h[kb][r], o0);
            }
          }
          o0 += o1;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int f = 16 * ob + 4 * g + i;
            if (f < D) {
              if (primal) {
                if (ch == 0 && valid && a.v_out) a.v_out[row * D + f] = o0[i] * z_vstd[f] - z_mean[f];
              } else if (f == dir) {
                div += o0[i] * z_vstd[f];
              }
            }
          }
        }
      }
    }
    div = sum_over_c(sum_over_g(div));
    if (lane == 0 && valid) a.div_out[row] = div;
  }
  pipe.drain();
#undef FM_ENTER
}

// ---------------------------------------------------------------- backward (dX chain)
template <int HB>
__global__ void __launch_bounds__(FM_THREADS, 1) fm_bwd_kernel(const FmPlan pl, const FmArgs a) {
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int D = pl.D, H = pl.H, L = pl.L;
  const float invH = 1.0f / (float)H;
  float* lnp = a.ln_part + ((long long)blockIdx.x * FM_WAVES + wave) * (long long)(L * 2 * 16 * HB);
  FmPipe pipe;
  pipe.init(lds, a.packed, pl.bgrp_off, pl.bgrp_floats, pl.nbg, pl.lds_bwd_floats, wave, lane, NSF_DBG_ABL(pl.ablate, 2));
  const float* wb = lds;
  // Stash reads are issued one stage ahead, in place, as soon as the registers they refill are dead (see
  // FmPipe::enter_wait for where their first use must sit); the head of the next tile is fetched during the
  // last stage of the current one.
  f4 sh[HB], u[HB], dv[4];
  float rstd = 0.f;
  auto tile_base = [&](int tile_) { return a.stash + ((long long)tile_ * FM_WAVES + wave) * (long long)pl.SB * 256; };
  auto load_head = [&](const float* wtb_) {
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) dv[ob] = ld_nat(wtb_, pl.s_diff + (ob < pl.DB ? ob : 0), c, g);
#pragma unroll
    for (int ob = 0; ob < HB; ++ob) sh[ob] = ld_nat(wtb_, pl.s_sh + (L - 1) * HB + ob, c, g);
    rstd = wtb_[pl.s_rstd * 256 + (L - 1) * 16 + c];
  };
  if ((int)blockIdx.x < a.ntiles) {
    const float* w0 = tile_base(blockIdx.x);
    load_head(w0);
#pragma unroll
    for (int ob = 0; ob < HB; ++ob) u[ob] = ld_nat(w0, pl.s_u + (L - 1) * HB + ob, c, g);
  }

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const long long wt = (long long)tile * FM_WAVES + wave;
    const long long row_raw = wt * 16 + c;
    const bool valid = row_raw < a.n;
    float* wtb = a.stash + wt * (long long)pl.SB * 256;
    float wrow = 0.f;
    if (valid) wrow = (a.row_weight ? a.row_weight[row_raw] : a.uniform_weight) * (2.0f / (float)D);

    f4 gh[HB], gte[HB], acc[HB];
    // ---- output layer: g_v = 2 w (out - target) / D ; g_h = W_o^T g_v
    pipe.enter_wait(pl.lin[J_L0 + L].bg_first);
    wb = pipe.base();
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) dv[ob] = dv[ob] * wrow;
    pipe.enter_prefetch();
    {
      const FmLin& q = pl.lin[J_L0 + L];
      const float* wl = wb + q.ltw + c * q.ldt + 4 * g;
#pragma unroll
      for (int ib = 0; ib < HB; ++ib) { gh[ib] = f4{0.f, 0.f, 0.f, 0.f}; gte[ib] = f4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) {
        if (ob < pl.DB) {
          if (!NSF_DBG_ABL(pl.ablate, 1)) st_tr(wtb, pl.g_v + ob, c, g, dv[ob]);
          gemm_blk<HB>(wl, q.ldt, ob, dv[ob], gh);
        }
      }
      for (int ob = 4; ob < pl.DB; ++ob) {
        const f4 gv = ld_nat(wtb, pl.s_diff + ob, c, g) * wrow;
        if (!NSF_DBG_ABL(pl.ablate, 1)) st_tr(wtb, pl.g_v + ob, c, g, gv);
        gemm_blk<HB>(wl, q.ldt, ob, gv, gh);
      }
    }
    // ---- residual blocks in reverse
    for (int l = L - 1; l >= 0; --l) {
      pipe.enter_wait(pl.lin[J_L0 + l].bg_first);
      wb = pipe.base();
      const FmLin& q = pl.lin[J_L0 + l];
      float m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) {
        const f4 gam = *reinterpret_cast<const f4*>(wb + q.ltg + 16 * ob + 4 * g);
        // LayerNorm parameter gradients: reduce over this wave's 16 rows, one add per feature into the
        // wave's private partial (single writer: deterministic)
        const f4 pg = gh[ob] * sh[ob];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float tg = sum_over_c(pg[i]), tb = sum_over_c(gh[ob][i]);
          if (c == 0) {
            unsafeAtomicAdd(lnp + (l * 2 + 0) * 16 * HB + 16 * ob + 4 * g + i, tg);
            unsafeAtomicAdd(lnp + (l * 2 + 1) * 16 * HB + 16 * ob + 4 * g + i, tb);
          }
        }
        gh[ob] = gh[ob] * gam;      // g_hat (zero on padded features: gamma is zero padded)
        m1 += (gh[ob][0] + gh[ob][1]) + (gh[ob][2] + gh[ob][3]);
        const f4 p2 = gh[ob] * sh[ob];
        m2 += (p2[0] + p2[1]) + (p2[2] + p2[3]);
      }
      m1 = sum_over_g(m1) * invH;
      m2 = sum_over_g(m2) * invH;
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) {
        f4 gs;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          gs[i] = (16 * ob + 4 * g + i) < H ? rstd * (gh[ob][i] - m1 - sh[ob][i] * m2) : 0.f;
        gte[ob] += gs;
        acc[ob] = gs;                                   // skip connection
        u[ob] = NSF_DBG_ABL(pl.ablate, 4) ? gs * u[ob] : gs * gelu_grad4(u[ob]);     // g_u, in place
        if (!NSF_DBG_ABL(pl.ablate, 1)) st_tr(wtb, pl.g_u + l * HB + ob, c, g, u[ob]);
      }
      // refill for the next stage: layer l-1's (s_hat, u, rstd), or (ie, h0) after the first block; the u
      // blocks are refilled inside the GEMM as soon as it has consumed them
#pragma unroll
      for (int ob = 0; ob < HB; ++ob)
        sh[ob] = ld_nat(wtb, (l > 0 ? pl.s_sh + (l - 1) * HB : pl.s_ie) + ob, c, g);
      rstd = wtb[pl.s_rstd * 256 + (l > 0 ? l - 1 : 0) * 16 + c];
      pipe.enter_prefetch();
      const int u_next = l > 0 ? pl.s_u + (l - 1) * HB : pl.s_h0;
      gemm_rr_cb<HB, HB>(wb + q.ltw + c * q.ldt + 4 * g, q.ldt, u, acc,
                         [&](int kb) { u[kb] = ld_nat(wtb, u_next + kb, c, g); });
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) gh[ob] = acc[ob];
    }
    // ---- h = GELU(h0); merge layer; input / condition layers     (u = h0, sh = ie)
    pipe.enter_wait(pl.lin[J_MA].bg_first);
    wb = pipe.base();
    f4 gh0[HB];
#pragma unroll
    for (int ob = 0; ob < HB; ++ob) {
      gh0[ob] = gh[ob] * gelu_grad4(u[ob]);
      st_tr(wtb, pl.g_h0 + ob, c, g, gh0[ob]);
      st_tr(wtb, pl.g_te + ob, c, g, gte[ob]);
      u[ob] = ld_nat(wtb, pl.s_ce + ob, c, g);
      sh[ob] = gelu_grad4(sh[ob]);
    }
    pipe.enter_prefetch();
    {
      const FmLin& q = pl.lin[J_MA];
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) acc[ob] = f4{0.f, 0.f, 0.f, 0.f};
      gemm_rr<HB, HB>(wb + q.ltw + c * q.ldt + 4 * g, q.ldt, gh0, acc);
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) st_tr(wtb, pl.g_ie + ob, c, g, acc[ob] * sh[ob]);
    }
    pipe.enter_wait(pl.lin[J_MB].bg_first);
    wb = pipe.base();
#pragma unroll
    for (int ob = 0; ob < HB; ++ob) u[ob] = gelu_grad4(u[ob]);
    pipe.enter_prefetch();
    const bool more = tile + (int)gridDim.x < a.ntiles;
    const float* wnext = tile_base(more ? tile + (int)gridDim.x : tile);
    if (more) load_head(wnext);
    {
      const FmLin& q = pl.lin[J_MB];
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) acc[ob] = f4{0.f, 0.f, 0.f, 0.f};
      gemm_rr<HB, HB>(wb + q.ltw + c * q.ldt + 4 * g, q.ldt, gh0, acc);
#pragma unroll
      for (int ob = 0; ob < HB; ++ob) {
        st_tr(wtb, pl.g_ce + ob, c, g, acc[ob] * u[ob]);
        if (more) u[ob] = ld_nat(wnext, pl.s_u + (L - 1) * HB + ob, c, g);
      }
    }
  }
  pipe.drain();
}

// ---------------------------------------------------------------- weight gradients: dW_j = G_j^T X_j over a row chunk
// One workgroup per (row chunk, linear).  Each of its 4 waves accumulates the WHOLE dW (OB x KB blocks, up to
// 8 x 8 f4 accumulators: one wave per SIMD) over every fourth wave-tile of the chunk, so each stash block is
// read once: A fragments by one 16-byte load per lane from the transposed G blocks, B fragments from the natural
// X blocks.  Operands of the next wave-tile are in flight under the MFMAs of the current one.  The four waves'
// sums are combined through LDS in a fixed order and written as the chunk's partial.
template <int OBT, int KBT>
__device__ __forceinline__ void fm_dw_body(const FmPlan& pl, const float* __restrict__ stash, long long nwt,
                                           float* __restrict__ partials, const FmLin& q, float* lds) {
  // OBT x KBT is the linear's block shape rounded up to an instantiated one: the MFMA loops are branch free
  // (a guard per MFMA costs a basic block and an s_waitcnt each); slots past the real OB / KB re-read the last
  // real block and accumulate values that are never stored.
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int OB = q.OB, KB = q.KB;
  const bool want_bias = q.g_b >= 0;
  constexpr bool PF = OBT * KBT <= 49;     // 8 x 8 accumulators leave no room for a second operand set
  f4 acc[OBT][KBT];
  float accb[OBT];                 // bias gradient: per-lane row sums of the A fragments (feature c, rows 4g..4g+3)
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob) {
    accb[ob] = 0.f;
#pragma unroll
    for (int kb = 0; kb < KBT; ++kb) acc[ob][kb] = f4{0.f, 0.f, 0.f, 0.f};
  }
  const long long wt0 = (long long)blockIdx.x * FM_DW_TILES + wave;
  const long long wt1 = (long long)(blockIdx.x + 1) * FM_DW_TILES < nwt ? (long long)(blockIdx.x + 1) * FM_DW_TILES : nwt;
  const long long wstride = (long long)pl.SB * 256;
  const float* gbase = stash + q.s_g * 256 + c * 16 + 4 * g;        // transposed G: [feature c][rows 4g..]
  const float* xbase = stash + q.s_x * 256 + (4 * g) * 16 + c;      // natural X: [rows 4g..][feature c]
  f4 av[OBT], bv[KBT], avn[PF ? OBT : 1], bvn[PF ? KBT : 1];
  auto load_ops = [&](long long wt, f4 (&a_)[OBT], f4 (&b_)[KBT]) {
    const float* ga = gbase + wt * wstride;
    const float* xb = xbase + wt * wstride;
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) a_[ob] = *reinterpret_cast<const f4*>(ga + (ob < OB ? ob : OB - 1) * 256);
#pragma unroll
    for (int kb = 0; kb < KBT; ++kb) {
      const float* xk = xb + (kb < KB ? kb : KB - 1) * 256;
      b_[kb] = f4{xk[0], xk[16], xk[32], xk[48]};
    }
  };
  if (PF && wt0 < wt1) load_ops(wt0, av, bv);
  for (long long wt = wt0; wt < wt1; wt += 4) {
    if constexpr (PF) {
      if (wt + 4 < wt1 && !NSF_DBG_ABL(pl.ablate, 32)) load_ops(wt + 4, avn, bvn);
    } else {
      load_ops(wt, av, bv);
    }
    if (q.x_gelu) {
#pragma unroll
      for (int kb = 0; kb < KBT; ++kb) bv[kb] = gelu4(bv[kb]);
    }
    if (!NSF_DBG_ABL(pl.ablate, 16)) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ob = 0; ob < OBT; ++ob)
#pragma unroll
          for (int kb = 0; kb < KBT; ++kb) acc[ob][kb] = MFMA16(av[ob][r], bv[kb][r], acc[ob][kb]);
    }
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) accb[ob] += (av[ob][0] + av[ob][1]) + (av[ob][2] + av[ob][3]);
    if constexpr (PF) {
#pragma unroll
      for (int b = 0; b < OBT; ++b) av[b] = avn[b];
#pragma unroll
      for (int b = 0; b < KBT; ++b) bv[b] = bvn[b];
    }
  }
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob) accb[ob] = sum_over_g(accb[ob]);   // every lane: total of feature 16*ob + c
  // ---- combine the four waves: (2,3) -> LDS, (0,1) add; 1 -> LDS, 0 adds and stores
  const int per_wave = OBT * KBT * 256 + OBT * 64;     // floats: acc blocks [blk][lane][4] + bias [ob][lane]
  auto spill = [&](float* dst) {
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) {
#pragma unroll
      for (int kb = 0; kb < KBT; ++kb) *reinterpret_cast<f4*>(dst + ((ob * KBT + kb) * 64 + lane) * 4) = acc[ob][kb];
      dst[OBT * KBT * 256 + ob * 64 + lane] = accb[ob];
    }
  };
  auto absorb = [&](const float* src) {
#pragma unroll
    for (int ob = 0; ob < OBT; ++ob) {
#pragma unroll
      for (int kb = 0; kb < KBT; ++kb)
        acc[ob][kb] += *reinterpret_cast<const f4*>(src + ((ob * KBT + kb) * 64 + lane) * 4);
      accb[ob] += src[OBT * KBT * 256 + ob * 64 + lane];
    }
  };
  if (wave >= 2) spill(lds + (wave - 2) * per_wave);
  __syncthreads();
  if (wave < 2) absorb(lds + wave * per_wave);
  __syncthreads();
  if (wave == 1) spill(lds);
  __syncthreads();
  if (wave != 0) return;
  absorb(lds);
  // partial in FRAGMENT layout (one 1 KB store per block); fm_reduce_kernel maps it to the flat gradient
  float* part = partials + (long long)blockIdx.x * pl.PF;
#pragma unroll
  for (int ob = 0; ob < OBT; ++ob) {
    if (ob >= OB) continue;
#pragma unroll
    for (int kb = 0; kb < KBT; ++kb)
      if (kb < KB) *reinterpret_cast<f4*>(part + q.pf_w + ((ob * KB + kb) * 64 + lane) * 4) = acc[ob][kb];
    if (g == 0) part[q.pf_b + ob * 16 + c] = accb[ob];
  }
}

// One launch for all linears: blockIdx.y walks them largest first (pl.dw_order) and branches, uniformly, to the
// body instantiated for the linear's rounded block shape.
__global__ void __launch_bounds__(FM_DW_THREADS) fm_dw_kernel(const FmPlan pl, const float* __restrict__ stash,
                                                           long long nwt, float* __restrict__ partials) {
  extern __shared__ __align__(16) float lds[];
  const FmLin& q = pl.lin[pl.dw_order[blockIdx.y]];
  const int ot = q.OB <= 4 ? 4 : (q.OB <= 7 ? 7 : 8), kt = q.KB <= 4 ? 4 : (q.KB <= 7 ? 7 : 8);
#define FM_DW_CASE(OT, KT) \
  if (ot == OT && kt == KT) return fm_dw_body<OT, KT>(pl, stash, nwt, partials, q, lds);
  FM_DW_CASE(7, 7) FM_DW_CASE(7, 4) FM_DW_CASE(4, 7) FM_DW_CASE(4, 4) FM_DW_CASE(8, 8)
  FM_DW_CASE(8, 4) FM_DW_CASE(4, 8) FM_DW_CASE(7, 8) FM_DW_CASE(8, 7)
#undef FM_DW_CASE
}

// Sums the chunks' fragment-layout partials (coalesced) and scatters each element to its place in the flat
// gradient: block (ob, kb), lane (c, g), register i  ->  dW[16*ob + 4*g + i][16*kb + c]
__global__ void __launch_bounds__(256) fm_reduce_kernel(const FmPlan pl, const float* __restrict__ partials, int nchunk,
                                                        float* __restrict__ grad) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= pl.PF) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = 0;
  for (; i + 3 < nchunk; i += 4) {
    s0 += partials[(long long)i * pl.PF + e];
    s1 += partials[(long long)(i + 1) * pl.PF + e];
    s2 += partials[(long long)(i + 2) * pl.PF + e];
    s3 += partials[(long long)(i + 3) * pl.PF + e];
  }
  for (; i < nchunk; ++i) s0 += partials[(long long)i * pl.PF + e];
  const float sum = (s0 + s1) + (s2 + s3);
  int j = 0;
  while (j + 1 < pl.NL && e >= pl.lin[j + 1].pf_w) ++j;
  const FmLin& q = pl.lin[j];
  if (e < q.pf_b) {
    const int r = e - q.pf_w, blk = r >> 8, lane = (r & 255) >> 2, ii = r & 3;
    const int ob = blk / q.KB, kb = blk - ob * q.KB;
    const int o = 16 * ob + 4 * (lane >> 4) + ii, in = 16 * kb + (lane & 15);
    if (o < q.out && in < q.in) grad[q.g_w + o * q.g_ld + in] = sum;
  } else {
    const int o = e - q.pf_b;
    if (o < q.out && q.g_b >= 0) grad[q.g_b + o] = sum;
  }
}

// LayerNorm gamma / beta gradients: one workgroup per (layer, gamma|beta, 16-feature block) sums the backward
// waves' partials: thread (f = tid & 15, pg = tid >> 4) takes partials pg, pg + 16, ...; fixed-order LDS finish
__global__ void __launch_bounds__(256) fm_ln_reduce_kernel(const FmPlan pl, const float* __restrict__ ln_part, int nln,
                                                           float* __restrict__ grad) {
  __shared__ float red[16][17];
  const int blk = blockIdx.x;                    // (l * 2 + k) * HB + ob
  const int lk = blk / pl.HB, ob = blk - lk * pl.HB;
  const int f = threadIdx.x & 15, pg = threadIdx.x >> 4;
  const long long stride = (long long)pl.L * 2 * 16 * pl.HB;
  const float* p = ln_part + lk * 16 * pl.HB + 16 * ob + f;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = pg;
  for (; i + 48 < nln; i += 64) {
    s0 += p[i * stride];
    s1 += p[(i + 16) * stride];
    s2 += p[(i + 32) * stride];
    s3 += p[(i + 48) * stride];
  }
  for (; i < nln; i += 16) s0 += p[i * stride];
  red[pg][f] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (pg == 0) {
    float t = 0.f;
    for (int q = 0; q < 16; ++q) t += red[q][f];
    const int feat = 16 * ob + f;
    if (feat < pl.H) grad[pl.g_ln + (lk >> 1) * 2 * pl.H + (lk & 1) * pl.H + feat] = t;
  }
}

// linears fed by a LayerNorm output: X was s_hat, so  dW[o][i] = gamma[i] M[o][i] + beta[i] db[o]
__global__ void __launch_bounds__(256) fm_lnfix_kernel(const FmPlan pl, const float* __restrict__ params,
                                                       float* __restrict__ grad) {
  const FmLin& q = pl.lin[blockIdx.y];
  if (q.ln_fix < 0) return;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= q.out * q.in) return;
  const int o = idx / q.in, i = idx - o * q.in;
  const float gam = params[pl.g_ln + q.ln_fix * 2 * pl.H + i], bet = params[pl.g_ln + q.ln_fix * 2 * pl.H + pl.H + i];
  float* w = grad + q.g_w + o * q.g_ld + i;
  *w = gam * *w + bet * grad[q.g_b + o];
}

// ---------------------------------------------------------------- packing
__global__ void __launch_bounds__(256) fm_pack_kernel(const FmPlan pl, const float* __restrict__ params,
                                                      float* __restrict__ packed) {
  const int j = blockIdx.x;
  const FmLin& q = pl.lin[j];
  const int tid = blockIdx.y * 256 + threadIdx.x, nth = gridDim.y * 256;
  const int rows = 16 * q.OB;
  for (int idx = tid; idx < rows * q.ldk; idx += nth) {
    const int o = idx / q.ldk, i = idx - o * q.ldk;
    packed[q.w_off + idx] = (o < q.out && i < q.in) ? params[q.g_w + o * q.g_ld + i] : 0.f;
  }
  const bool is_layer = j >= J_L0 && j < J_L0 + pl.L;
  const int l = j - J_L0;
  for (int idx = tid; idx < 3 * rows; idx += nth) {
    const int k = idx / rows, o = idx - k * rows;
    float v = 0.f;
    if (o < q.out) {
      if (k == 0) v = q.g_b >= 0 ? params[q.g_b + o] : 0.f;
      else if (is_layer) v = params[pl.g_ln + l * 2 * pl.H + (k - 1) * pl.H + o];
    }
    packed[q.w_off + rows * q.ldk + idx] = v;
  }
  if (j == J_IN || j == J_CT || j == J_TM) return;
  const int trows = 16 * q.KB;
  for (int idx = tid; idx < trows * q.ldt; idx += nth) {
    const int i = idx / q.ldt, o = idx - i * q.ldt;
    packed[q.t_off + idx] = (o < q.out && i < q.in) ? params[q.g_w + o * q.g_ld + i] : 0.f;
  }
  for (int o = tid; o < rows; o += nth)
    packed[q.t_off + trows * q.ldt + o] = (is_layer && o < q.out) ? params[pl.g_ln + l * 2 * pl.H + o] : 0.f;
}

// ---------------------------------------------------------------- host side
static int fm_grid(int ntiles) {
  static int cus_of[64] = {0};        // per device; filled on first use
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cus_of[dev] == 0) {
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    cus_of[dev] = cus;
  }
  const int cus = cus_of[dev];
  return ntiles < cus ? ntiles : cus;   // one 8-wave workgroup per CU
}

template <int MODE>
static int fm_launch_fwd(const FmPlan& pl, const FmArgs& a, hipStream_t st) {
  const size_t lds = 4ull * (2 * pl.lds_fwd_floats + FM_ZS_FLOATS);
  const int grid = fm_grid(a.ntiles);
#define FM_FWD_CASE(HBV)                                                                                        \
  case HBV: {                                                                                                   \
    hipError_t e = hipFuncSetAttribute((const void*)fm_fwd_kernel<HBV, MODE>,                                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    if (e != hipSuccess) return (int)e;                                                                         \
    hipLaunchKernelGGL((fm_fwd_kernel<HBV, MODE>), dim3(grid), dim3(FM_THREADS), lds, st, pl, a);               \
    break;                                                                                                      \
  }
  switch (pl.HB) {
    FM_FWD_CASE(4)
    FM_FWD_CASE(7)
    FM_FWD_CASE(8)
    default: return SBI_AMD_E_UNSUPPORTED;
  }
#undef FM_FWD_CASE
  return (int)hipGetLastError();
}

static int fm_launch_div(const FmPlan& pl, const FmArgs& a, hipStream_t st) {
  const size_t lds = 4ull * (2 * pl.lds_fwd_floats + FM_ZS_FLOATS);
  const int grid = fm_grid(a.ntiles);
#define FM_DIV_CASE(HBV)                                                                                        \
  case HBV: {                                                                                                   \
    hipError_t e = hipFuncSetAttribute((const void*)fm_div_kernel<HBV>,                                         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    if (e != hipSuccess) return (int)e;                                                                         \
    hipLaunchKernelGGL((fm_div_kernel<HBV>), dim3(grid), dim3(FM_THREADS), lds, st, pl, a);                     \
    break;                                                                                                      \
  }
  switch (pl.HB) {
    FM_DIV_CASE(4)
    FM_DIV_CASE(7)
    FM_DIV_CASE(8)
    default: return SBI_AMD_E_UNSUPPORTED;
  }
#undef FM_DIV_CASE
  return (int)hipGetLastError();
}

static int fm_launch_bwd(const FmPlan& pl, const FmArgs& a, int grid, hipStream_t st) {
  const size_t lds = 4ull * 2 * pl.lds_bwd_floats;
#define FM_BWD_CASE(HBV)                                                                                        \
  case HBV: {                                                                                                   \
    hipError_t e = hipFuncSetAttribute((const void*)fm_bwd_kernel<HBV>,                                         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    if (e != hipSuccess) return (int)e;                                                                         \
    hipLaunchKernelGGL((fm_bwd_kernel<HBV>), dim3(grid), dim3(FM_THREADS), lds, st, pl, a);                     \
    break;                                                                                                      \
  }
  switch (pl.HB) {
    FM_BWD_CASE(4)
    FM_BWD_CASE(7)
    FM_BWD_CASE(8)
    default: return SBI_AMD_E_UNSUPPORTED;
  }
#undef FM_BWD_CASE
  return (int)hipGetLastError();
}

struct FmWs { long long stash, partials, ln_part, total; int nchunk, nln_max; long long nwt; int ntiles; };
static FmWs fm_ws_layout(const FmPlan& pl, long long n) {
  FmWs w;
  w.ntiles = (int)((n + FM_ROWS - 1) / FM_ROWS);
  w.nwt = (long long)w.ntiles * FM_WAVES;
  w.nchunk = (int)((w.nwt + FM_DW_TILES - 1) / FM_DW_TILES);
  w.nln_max = FM_WAVES * 1024;   // backward waves: 8 per workgroup, one workgroup per CU, CUs <= 1024
  w.stash = 0;
  w.partials = w.stash + w.nwt * (long long)pl.SB * 256;
  w.ln_part = w.partials + (long long)w.nchunk * pl.PF;
  w.total = w.ln_part + (long long)w.nln_max * pl.L * 2 * 16 * pl.HB;
  return w;
}

extern "C" {

int64_t sbi_amd_fmpe_param_count(const sbi_amd_fmpe_config* cfg) {
  FmPlan pl;
  int rc = fm_build_plan(cfg, &pl);
  return rc ? rc : pl.P;
}

int64_t sbi_amd_fmpe_param_offset(const sbi_amd_fmpe_config* cfg, int32_t kind, int32_t layer) {
  FmPlan pl;
  int rc = fm_build_plan(cfg, &pl);
  if (rc) return rc;
  if (layer < 0 || layer >= pl.L) return SBI_AMD_E_BADARG;
  switch (kind) {
    case 0: return pl.lin[J_IN].g_w;
    case 1: return pl.lin[J_IN].g_b;
    case 2: return pl.lin[J_CT].g_w;
    case 3: return pl.lin[J_CT].g_b;
    case 4: return pl.lin[J_MA].g_w;
    case 5: return pl.lin[J_MA].g_b;
    case 6: return pl.lin[J_TM].g_w;
    case 7: return pl.lin[J_TM].g_b;
    case 8: return pl.lin[J_L0 + layer].g_w;
    case 9: return pl.lin[J_L0 + layer].g_b;
    case 10: return pl.g_ln + layer * 2 * pl.H;
    case 11: return pl.g_ln + layer * 2 * pl.H + pl.H;
    case 12: return pl.lin[J_L0 + pl.L].g_w;
    case 13: return pl.lin[J_L0 + pl.L].g_b;
    default: return SBI_AMD_E_BADARG;
  }
}

int64_t sbi_amd_fmpe_packed_floats(const sbi_amd_fmpe_config* cfg) {
  FmPlan pl;
  int rc = fm_build_plan(cfg, &pl);
  return rc ? rc : pl.packed_floats;
}

int sbi_amd_fmpe_pack(const sbi_amd_fmpe_config* cfg, const float* params, float* packed, void* stream) {
  FmPlan pl;
  int rc = fm_build_plan(cfg, &pl);
  if (rc) return rc;
  if (!params || !packed) return SBI_AMD_E_BADARG;
  hipLaunchKernelGGL(fm_pack_kernel, dim3(pl.NL, 8), dim3(256), 0, (hipStream_t)stream, pl, params, packed);
  return (int)hipGetLastError();
}

int sbi_amd_fmpe_velocity(const sbi_amd_fmpe_config* cfg, const float* packed, const float* zstats,
                          const float* theta_t, const float* x, int64_t x_rows, const float* times,
                          int64_t t_rows, int64_t n, float* v_out, void* stream) {
  FmPlan pl;
  int rc = fm_build_plan(cfg, &pl);
  if (rc) return rc;
  if (!packed || !zstats || !theta_t || !x || !times || !v_out || n < 0) return SBI_AMD_E_BADARG;
  if ((x_rows != 1 && x_rows != n) || (t_rows != 1 && t_rows != n)) return SBI_AMD_E_BADARG;
  if (n == 0) return 0;
  FmArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = packed; a.zstats = zstats; a.theta = theta_t; a.x = x; a.times = times; a.n = n;
  a.x_rows = (int)(x_rows == 1 ? 1 : 2); a.t_rows = (int)(t_rows == 1 ? 1 : 2);
  if (n == 1) { a.x_rows = 1; a.t_rows = 1; }
  a.v_out = v_out; a.ntiles = (int)((n + FM_ROWS - 1) / FM_ROWS);
  return fm_launch_fwd<0>(pl, a, (hipStream_t)stream);
}

int sbi_amd_fmpe_velocity_div(const sbi_amd_fmpe_config* cfg, const float* packed, const float* zstats,
                              const float* theta_t, const float* x, int64_t x_rows, const float* times,
                              int64_t t_rows, int64_t n, float* v_out, float* div_out, void* stream) {
  FmPlan pl;
  int rc = fm_build_plan(cfg, &pl);
  if (rc) return rc;
  if (!packed || !zstats || !theta_t || !x || !times || !div_out || n < 0) return SBI_AMD_E_BADARG;
  if ((x_rows != 1 && x_rows != n) || (t_rows != 1 && t_rows != n)) return SBI_AMD_E_BADARG;
  if (n == 0) return 0;
  FmArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = packed; a.zstats = zstats; a.theta = theta_t; a.x = x; a.times = times; a.n = n;
  a.x_rows = (int)(x_rows == 1 ? 1 : 2); a.t_rows = (int)(t_rows == 1 ? 1 : 2);
  if (n == 1) { a.x_rows = 1; a.t_rows = 1; }
  a.v_out = v_out; a.div_out = div_out; a.ntiles = (int)((n + FM_WAVES - 1) / FM_WAVES);
  return fm_launch_div(pl, a, (hipStream_t)stream);
}

int sbi_amd_fmpe_loss(const sbi_amd_fmpe_config* cfg, const float* packed, const float* zstats, const float* theta,
                      const float* x, int64_t x_rows, const float* times, const float* noise, int64_t n,
                      float* loss_out, void* stream) {
  FmPlan pl;
  int rc = fm_build_plan(cfg, &pl);
  if (rc) return rc;
  if (!packed || !zstats || !theta || !x || !times || !noise || !loss_out || n < 0) return SBI_AMD_E_BADARG;
  if (x_rows != 1 && x_rows != n) return SBI_AMD_E_BADARG;
  if (n == 0) return 0;
  FmArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = packed; a.zstats = zstats; a.theta = theta; a.x = x; a.times = times; a.noise = noise; a.n = n;
  a.x_rows = (x_rows == 1 || n == 1) ? 1 : 2; a.t_rows = n == 1 ? 1 : 2;
  a.loss_out = loss_out; a.ntiles = (int)((n + FM_ROWS - 1) / FM_ROWS);
  return fm_launch_fwd<1>(pl, a, (hipStream_t)stream);
}

int64_t sbi_amd_fmpe_train_workspace_floats(const sbi_amd_fmpe_config* cfg, int64_t n) {
  FmPlan pl;
  int rc = fm_build_plan(cfg, &pl);
  if (rc) return rc;
  if (n <= 0) return SBI_AMD_E_BADARG;
  return fm_ws_layout(pl, n).total;
}

int sbi_amd_fmpe_loss_fwd_bwd(const sbi_amd_fmpe_config* cfg, const float* params, const float* packed,
                              const float* zstats, const float* theta, const float* x, int64_t x_rows,
                              const float* times, const float* noise, int64_t n, const float* row_weight,
                              float uniform_weight, float* loss_out, float* grad_out, float* workspace,
                              void* stream) {
  FmPlan pl;
  int rc = fm_build_plan(cfg, &pl);
  if (rc) return rc;
  if (!params || !packed || !zstats || !theta || !x || !times || !noise || !loss_out || !grad_out || !workspace ||
      n <= 0)
    return SBI_AMD_E_BADARG;
  if (x_rows != 1 && x_rows != n) return SBI_AMD_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const FmWs w = fm_ws_layout(pl, n);
  FmArgs a;
  memset(&a, 0, sizeof(a));
  a.packed = packed; a.zstats = zstats; a.theta = theta; a.x = x; a.times = times; a.noise = noise; a.n = n;
  a.x_rows = (x_rows == 1 || n == 1) ? 1 : 2; a.t_rows = n == 1 ? 1 : 2;
  a.row_weight = row_weight; a.uniform_weight = uniform_weight;
  a.loss_out = loss_out; a.stash = workspace + w.stash; a.ln_part = workspace + w.ln_part; a.ntiles = w.ntiles;
  static long long* tl_dev = nullptr;
  if (sbi_amd_dbg_fm_timeline()) {
    if (!tl_dev) { hipMalloc(&tl_dev, 8 * 32 * 8); }
    hipMemsetAsync(tl_dev, 0, 8 * 32 * 8, st);
    a.timeline = tl_dev;
  }
  rc = fm_launch_fwd<2>(pl, a, st);
  if (rc) return rc;
  if (a.timeline) {
    static int shown = 0;
    if (++shown == 20) {
      long long h[8 * 32];
      hipStreamSynchronize(st);
      hipMemcpy(h, tl_dev, sizeof(h), hipMemcpyDeviceToHost);
      for (int it = 0; it < 3; ++it) {
        fprintf(stderr, "fwd timeline tile-iter %d:", it);
        for (int k = 1; k < 32 && h[it * 32 + k]; ++k) fprintf(stderr, " %lld", h[it * 32 + k] - h[it * 32 + k - 1]);
        fprintf(stderr, "\n");
      }
    }
    a.timeline = nullptr;
  }
  const int bgrid = fm_grid(w.ntiles);
  const int nln = bgrid * FM_WAVES;
  if (nln > w.nln_max) return SBI_AMD_E_UNSUPPORTED;
  hipError_t e = hipMemsetAsync(workspace + w.ln_part, 0, 4ull * nln * pl.L * 2 * 16 * pl.HB, st);
  if (e != hipSuccess) return (int)e;
  rc = fm_launch_bwd(pl, a, bgrid, st);
  if (rc) return rc;
  {
    int mo = 4, mk = 4;
    for (int j = 0; j < pl.NL; ++j) {
      const int ot = pl.lin[j].OB <= 4 ? 4 : (pl.lin[j].OB <= 7 ? 7 : 8);
      const int kt = pl.lin[j].KB <= 4 ? 4 : (pl.lin[j].KB <= 7 ? 7 : 8);
      if (ot * kt * 256 + ot * 64 > mo * mk * 256 + mo * 64) { mo = ot; mk = kt; }
    }
    const size_t dlds = 4ull * 2 * (mo * mk * 256 + mo * 64);
    hipError_t e2 = hipFuncSetAttribute((const void*)fm_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)dlds);
    if (e2 != hipSuccess) return (int)e2;
    hipLaunchKernelGGL(fm_dw_kernel, dim3(w.nchunk, pl.NL), dim3(FM_DW_THREADS), dlds, st, pl, workspace + w.stash,
                       w.nwt, workspace + w.partials);
  }
  hipLaunchKernelGGL(fm_reduce_kernel, dim3((pl.PF + 255) / 256), dim3(256), 0, st, pl, workspace + w.partials,
                     w.nchunk, grad_out);
  hipLaunchKernelGGL(fm_ln_reduce_kernel, dim3(pl.L * 2 * pl.HB), dim3(256), 0, st, pl, workspace + w.ln_part, nln,
                     grad_out);
  const int wmax = (pl.H > pl.D ? pl.H : pl.D) * pl.H;
  hipLaunchKernelGGL(fm_lnfix_kernel, dim3((wmax + 255) / 256, pl.NL), dim3(256), 0, st, pl, params, grad_out);
  return (int)hipGetLastError();
}

}  // extern "C"
