// nsf_plan_layout.h -- the integer layout of an NsfPlan as constexpr functions (see nsf_plan.h for what a plan is).
#pragma once
#include "nsf_plan.h"

constexpr int nsf_round_up(int v, int m) { return (v + m - 1) / m * m; }
constexpr int nsf_two_odd_at_least(int v) {   // smallest 2*odd >= v
  int x = (v + 1) / 2;                 // ceil(v/2)
  if ((x & 1) == 0) x += 1;
  return 2 * x;
}

constexpr int nsf_supported_bins(int K) { return K == 4 || K == 5 || K == 8 || K == 10 || K == 16; }

constexpr int nsf_check_cfg(const sbi_amd_nsf_config* c) {
  if (!c) return SBI_AMD_E_BADARG;
  if (c->D < 1 || c->C < 1 || c->H < 1 || c->T < 1 || c->NB < 0) return SBI_AMD_E_BADARG;
  if (c->H > 16 * NSF_HT_WIDE || c->T > NSF_MAX_T || c->NB > NSF_MAX_NB || !nsf_supported_bins(c->K))
    return SBI_AMD_E_UNSUPPORTED;
  if (c->D > 64 || c->C > 256) return SBI_AMD_E_UNSUPPORTED;
  if (c->min_bin_width * c->K > 1.0f || c->min_bin_height * c->K > 1.0f) return SBI_AMD_E_BADARG;
  if (c->ctx_layers < -1) return SBI_AMD_E_BADARG;                    // -1: no hidden layer at all; 0: the default (1)
  if (c->D == 1 && c->ctx_layers > 4) return SBI_AMD_E_UNSUPPORTED;
  return 0;
}

constexpr void nsf_set_lin(LinDesc* L, int* g, int* l, int out, int in, int bias_pad, int ksteps_fixed,
                    int min_rows = 0) {
  L->out = out;
  L->in = in;
  L->ksteps = ksteps_fixed > 0 ? ksteps_fixed : nsf_round_up((in + 3) / 4, 4);
  // B-from-LDS layers: both operands are zero padded up to 4*ksteps columns.  Hidden-K
  // layers (B = activation fragments, exact zeros beyond H via the zero row): the
  // tail K-steps may run into the following row; those weights meet a zero B.
  L->ldk = nsf_two_odd_at_least(ksteps_fixed > 0 ? in : 4 * L->ksteps);
  L->g_w = *g;
  *g += out * in;
  L->g_b = *g;
  *g += out;
  L->l_w = *l;
  // rows [out, rows_alloc) are all-zero: row `out` is what out-of-range MFMA A rows read, and the
  // backward's transposed K loop walks rows 0 .. 4*KSH-1 of the hidden x hidden layers unpredicated
  int rows_alloc = out + 1;
  if (min_rows > rows_alloc) rows_alloc = min_rows;
  L->rows = rows_alloc;
  *l += rows_alloc * L->ldk;
  L->l_b = *l;
  *l += bias_pad;
}

// Integer part of the plan (offsets, strides, tile counts): a constexpr function, so that the benchmark configuration's
// kernels can be instantiated with the whole layout folded into immediates (nsf_static_plan.h); the floating-point
// constants and the debug switches are filled in by nsf_build_plan (nsf_plan.cpp).
constexpr int nsf_build_layout(const sbi_amd_nsf_config* cfg, int nw, NsfPlan* pl) {
  int rc = nsf_check_cfg(cfg);
  if (rc) return rc;
  *pl = NsfPlan{};
  const int D = cfg->D, C = cfg->C, H = cfg->H, K = cfg->K, T = cfg->T;
  const int ctx_mlp = (D == 1);
  const int NB = ctx_mlp ? 0 : cfg->NB;   // the context-only conditioner has no residual blocks
  pl->ctx_mlp = ctx_mlp;
  pl->ctx_reps = ctx_mlp ? (cfg->ctx_layers > 0 ? cfg->ctx_layers : (cfg->ctx_layers < 0 ? 0 : 1)) : 0;
  pl->D = D; pl->C = C; pl->H = H; pl->K = K; pl->T = T; pl->NB = NB;
  pl->P = 3 * K - 1;
  pl->PT = (pl->P + 15) / 16;
  // kernels are instantiated for 13 (H=49..52) and 16; hidden > 64 (wide cooperative kernels only): whole K-quads
  pl->KSH = H > 16 * NSF_HT ? 4 * NSF_HT_WIDE : (((H + 3) / 4 == 13) ? 13 : 16);

  for (int par = 0; par < 2; ++par) {
    ShapeDesc* s = &pl->shape[par];
    // create_alternating_binary_mask (torchutils.py:396-410): even transforms
    // transform the even feature indices, odd ones the odd indices.
    s->d_tr = ctx_mlp ? 1 : ((par == 0) ? (D + 1) / 2 : D / 2);   // D == 1: dummy mask [1] every transform
    s->d_id = D - s->d_tr;
    s->in0 = s->d_id + C;
    int g = 0, l = 0;
    const int hb = 16 * NSF_HT;
    const int tr_rows = 4 * pl->KSH + 1;   // transposed (backward) K loops walk 4*KSH rows
    nsf_set_lin(&s->lin[0], &g, &l, H, s->in0, hb, 0, tr_rows);
    if (ctx_mlp) {
      if (pl->ctx_reps > 0) nsf_set_lin(&s->lin[1], &g, &l, H, H, hb, pl->KSH, tr_rows);   // (none: lin[1] stays empty)
      s->fin = 2;
    } else {
      for (int b = 0; b < NB; ++b) {
        nsf_set_lin(&s->lin[1 + 3 * b], &g, &l, H, C, hb, 0);
        nsf_set_lin(&s->lin[2 + 3 * b], &g, &l, H, H, hb, pl->KSH, tr_rows);
        nsf_set_lin(&s->lin[3 + 3 * b], &g, &l, H, H, hb, pl->KSH, tr_rows);
      }
      s->fin = 1 + 3 * NB;
    }
    l = nsf_round_up(l, 4);   // the final layer (+ LU) is staged on its own by the backward kernel's overlay mode
    if (l > pl->hidden_img_floats) pl->hidden_img_floats = l;
    s->final_off = l;
    nsf_set_lin(&s->lin[s->fin], &g, &l, s->d_tr * pl->P, H, s->d_tr * 16 * pl->PT, pl->KSH);
    s->g_lu = g;
    if (!ctx_mlp) g += D * (D - 1) + 2 * D;   // lower, upper, unconstrained diag, bias
    s->n_params = g;
    { const int lus = D <= 16 ? 16 : D;   // dense U, L padded to 16 x 16 for D <= 16
      l = nsf_round_up(l, 4);   // 16-byte aligned: the backward kernel reads matrix rows as float4
      s->l_U = l; l += lus * lus;
      s->l_L = l; l += lus * lus; }
    s->l_lub = l; l += D + 1;   // bias, then sum_i log U_ii
    // everything above is what the training kernels stage; the explicit inverses (sampling direction only)
    // come last so that the backward kernel can leave them out of its LDS image
    l = nsf_round_up(l, 4);
    if (l > pl->lds_w_train_floats) pl->lds_w_train_floats = l;
    s->l_Ui = s->l_Li = -1;
    if (!ctx_mlp && D <= 16) { s->l_Ui = l; l += 256; s->l_Li = l; l += 256; }
    s->lds_floats = nsf_round_up(l + 8, 4);
    if (s->lds_floats > pl->lds_w_floats) pl->lds_w_floats = s->lds_floats;
  }
  pl->img_floats = pl->lds_w_floats;
  int off = 0;
  for (int t = 0; t < T; ++t) {
    pl->g_layer[t] = off;
    off += pl->shape[t & 1].n_params;
  }
  pl->n_params = off;

  // per-wave scratch; every row stride is 2*odd (bank-conflict-free b32 access
  // by (row = lane&15, k-slot = lane>>4) lane pairs)
  pl->ZW = nsf_two_odd_at_least(D);
  pl->CW = nsf_two_odd_at_least(C);
  int d_id_max = pl->shape[0].d_id > pl->shape[1].d_id ? pl->shape[0].d_id : pl->shape[1].d_id;
  int need = d_id_max + 4 * nsf_round_up((C + 3) / 4, 4);       // context-layer K-steps read past C
  int need2 = 4 * nsf_round_up((d_id_max + C + 3) / 4, 4);      // initial-layer K-steps
  pl->CINW = nsf_two_odd_at_least(need > need2 ? need : need2);
  // spline-parameter staging: P (made odd) floats per row => conflict-free per-row reads;
  // slot stride == 16 (mod 32) puts the second dim slot of a 32-lane half on the other banks
  pl->PSW = 16 * pl->PT + 1;   // odd (conflict-free per-row reads) and wide enough for all 16*PT outputs
  pl->DS = 16 * pl->PSW;
  while ((pl->DS & 31) != 16) pl->DS += 1;
  // scratch = flow state rows, context rows, and TWO spline-parameter buffers (the final-layer GEMM
  // of chunk c+1 is issued under the spline of chunk c).  The conditioner-input rows alias the
  // second buffer and the LU temporaries the first: both are dead while those are live.
  int d_tr_max = pl->shape[0].d_tr;
  pl->DCH = 2 < d_tr_max ? 2 : d_tr_max;   // a spline task occupies a lane PAIR: 16 rows x 2 dims per pass
  int pst_sz = pl->DCH * pl->DS;
  if (pst_sz < 16 * pl->CINW) pst_sz = 16 * pl->CINW;
  if (pst_sz < 16 * pl->ZW + 16) pst_sz = 16 * pl->ZW + 16;
  int o = 0;
  pl->sc_zs = o; o += 16 * pl->ZW;
  pl->sc_cs = o;
  if (C > 16) o += 16 * pl->CW;   // C <= 16: the standardized context lives in 4 registers per lane
  pl->sc_pst = o; o += pst_sz;
  // more than two waves per SIMD (nw > 8): ONE staging buffer per wave (4.9 instead of 9.1 KB at the defaults), the
  // final-layer GEMM of a chunk then runs before that chunk's spline instead of under the previous one's; latency
  // is covered by the third wave of the SIMD instead of by the second buffer
  if (nw > 8) {
    pl->sc_pst2 = pl->sc_pst;
  } else {
    pl->sc_pst2 = o; o += pst_sz;
  }
  pl->sc_us = pl->sc_pst;
  pl->sc_cin = pl->sc_pst2;
  pl->sc_total = nsf_round_up(o, 4);
  if (4ll * ((int64_t)pl->lds_w_floats + (int64_t)nw * pl->sc_total) > NSF_LDS_LIMIT_BYTES) return SBI_AMD_E_LDS;
  // hidden > 64: the throughput kernels (four hidden m-tiles per wave, one transform's image in LDS) never take the
  // shape, whatever its image happens to weigh; the flat-parameter part of the plan is what the wide kernels use
  if (H > 16 * NSF_HT) {
    pl->img_floats = 0;        // no throughput image is ever packed for it
    return SBI_AMD_E_LDS;
  }
  return 0;
}

