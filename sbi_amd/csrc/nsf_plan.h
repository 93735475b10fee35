// nsf_plan.h -- host-computed execution plan shared by every NSF kernel.
//
// The plan turns an sbi_amd_nsf_config (the hyper-parameters build_nsf takes,
// /root/reference/sbi/neural_nets/net_builders/flow.py:333-460) into
//   * offsets into the flat fp32 parameter buffer (nflows' natural order),
//   * the LDS image one workgroup stages per coupling layer (MFMA A-operand
//     layout: row-major [out+1][ldk], ldk = 2*odd so that the 16x16x4 f32
//     MFMA operand reads are LDS-bank-conflict free, zero padded), and
//   * the per-wave LDS scratch (flow state, conditioner input, spline params).
// It is passed to kernels by value (~1.3 KB of kernarg).
#pragma once
#include <stdint.h>
#include "../../include/sbi_amd_nsf.h"

#define NSF_MAX_T 16
#define NSF_MAX_NB 4
#define NSF_MAX_LIN (2 + 3 * NSF_MAX_NB)
#define NSF_HT 4          // hidden tiles of 16 of the throughput kernels (weights of a transform in LDS) -> H <= 64
#define NSF_HT_WIDE 8     // hidden tiles the wide cooperative kernels take (weights from L2, nsf_coop_wide_kernel.h) -> H <= 128
#define NSF_MAX_DCH 2     // spline dims per chunk: one (row, dim) task per lane pair (lane, lane^32)
#define NSF_LDS_LIMIT_BYTES (160 * 1024)

#include "debug_env.h"   // NSF_DBG_ABL and the (debug-build-only) environment switches
struct LinDesc {
  int g_w, g_b;   // float offsets relative to the layer's block in the flat buffer
  int l_w, l_b;   // float offsets inside the LDS weight image
  int out, in;    // natural dims of nn.Linear(in, out)
  int rows;       // rows stored in the image (>= out+1; rows >= out are zero)
  int ldk;        // LDS row stride (floats), 2*odd, >= 4*ksteps
  int ksteps;     // MFMA K-steps the kernels run: ceil(in/4), rounded up to a multiple of 4
                  // for the layers whose B operand comes from LDS (initial / context layers)
                  // and to the compile-time KSH for the hidden->* layers
};

struct ShapeDesc {   // one per mask parity (even / odd transform index)
  int d_id, d_tr, in0;
  LinDesc lin[NSF_MAX_LIN];   // 0 initial | 1+3b ctx_b | 2+3b lin0_b | 3+3b lin1_b | 1+3NB final
                              // ctx_mlp (D == 1): 0 input C->H | 1 hidden H->H | 2 final
  int fin;                    // index of the final layer in lin[]
  int g_lu;                   // LULinear block offset relative to the layer block
  int l_U, l_L, l_lub;        // LDS offsets of expanded U[D][D], L[D][D], bias[D]
  int l_Ui, l_Li;             // explicit inverses U^-1, L^-1 (16 x 16, D <= 16 only; else -1)
  int final_off;              // image offset (multiple of 4) where the final layer, U, L, LU bias start
  int n_params;               // floats in this layer block
  int lds_floats;             // size of the LDS image
};

struct NsfPlan {
  int D, C, H, K, T, NB, P, PT;   // P = 3K-1, PT = ceil(P/16)
  int ctx_mlp;                    // D == 1: sbi's ContextSplineMap conditioner (flow.py:1419-1478): params from
                                  // the context only (C->H relu, H->H relu, H->P), mask [1], no LULinear
  int ctx_reps;                   // ctx_mlp: applications of the one shared hidden layer (hidden_layers_spline_context)
  int KSH;                        // hidden-layer K-steps the kernel template is instantiated for (13 or 16; 32: H > 64)
  float B, min_w, min_h, min_d, lu_eps, sqrt_h, inv_sqrt_h;
  float one_minus_kw, one_minus_kh;   // 1 - min_w*K, 1 - min_h*K
  float d_const;                      // log(exp(1-min_d)-1): boundary derivative pre-activation
  float log_z;                        // 0.5*D*log(2*pi), fp32 (flow.py:1486-1487)
  ShapeDesc shape[2];
  int g_layer[NSF_MAX_T];
  int n_params;
  int img_floats;               // floats per layer in the packed weight image (= lds_w_floats)
  int lds_w_floats;             // LDS weight image size (max over parities)
  int lds_w_train_floats;       // its prefix without the explicit LU inverses: what the backward kernel stages
  int hidden_img_floats;        // its prefix up to the final layer (max over parities): the hidden-layer images
  // per-wave scratch (float offsets relative to the wave's scratch base)
  int ZW, CW, CINW, PSW, DS, DCH;
  int sc_zs, sc_us, sc_cs, sc_cin, sc_pst, sc_pst2, sc_total;
  int ablate;                   // -DNSF_DEBUG builds only (env SBI_AMD_ABLATE): skip phases, results invalid; else 0
};

// Builds the plan for nw waves per workgroup; returns 0 or SBI_AMD_E_*.
int nsf_build_plan(const sbi_amd_nsf_config* cfg, int nw, NsfPlan* pl);
// Largest nw in {8,4,2,1} (<= nw_max) whose LDS footprint fits 160 KiB.
// `wide`: allow 12-wave workgroups (3 per SIMD, one staging buffer per wave) when rows and LDS permit -- measured to
// pay for the sampling direction only
int nsf_plan_for_rows(const sbi_amd_nsf_config* cfg, int64_t n, NsfPlan* pl, int* nw_out, bool wide = false);
// Packed weight image: T consecutive LDS images (img_floats each), written by
// nsf_pack_kernel from the flat parameters; kernels stage a layer with a float4 copy.
// activation-stash slots per (transform, 16-row tile): h_0 | per block t1 t2 sigmoid(gate) h; ctx_mlp: h_1 ... h_{reps+1}
constexpr int nsf_ast_slots(const NsfPlan& pl) { return pl.ctx_mlp ? 1 + pl.ctx_reps : 1 + 4 * pl.NB; }
// spline-parameter stash of the training pass (round 6): the forward kernel keeps the final layer's outputs -- the raw
// spline parameters -- per (transform, 16-row wave-tile, transformed dim, parameter m-tile) as one 16-byte word per lane
// in MFMA D-fragment order (lane (j, g), register r = parameter 16 pt + 4 r + g of row j), and the wave-specialised
// backward reloads them instead of recomputing the final layer (130 of ~1000 MFMAs per 16 rows at the defaults).
constexpr int nsf_pst_tile_floats(const NsfPlan& pl) {
  return (pl.shape[0].d_tr > pl.shape[1].d_tr ? pl.shape[0].d_tr : pl.shape[1].d_tr) * pl.PT * 256;
}
static inline int64_t nsf_packed_floats(const NsfPlan& pl) { return (int64_t)pl.T * pl.img_floats; }
static inline int64_t nsf_lds_bytes(const NsfPlan& pl, int nw) {
  return 4ll * ((int64_t)pl.lds_w_floats + (int64_t)nw * pl.sc_total);
}
