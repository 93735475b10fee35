#pragma once
// nsf_coop.h -- plan and weight image of the LATENCY-oriented ("cooperative") NSF kernels for small batches.
//
// Why a second kernel family ("Small batches"): the throughput kernels give one wavefront a
// 16-row tile and ALL output features of every layer, with the transform's weights staged in LDS -- right for
// 65 536 rows, but at <= 8 192 rows the step is the latency of one tile through T transforms (five dependent
// backward launches of 46 us, a 116 us forward).  Here FOUR wavefronts (one per SIMD) share a tile of 16 * NT rows:
//   * wave w owns output features [16 w, 16 w + 16) of every hidden layer: a 64 -> 64 layer is 13 MFMAs per wave
//     instead of 52, followed by an all-gather of the four D fragments through LDS (one ds_write_b128, one
//     barrier, three ds_read_b128 per wave: lane L of wave w needs exactly what lane L of the other waves holds);
//   * weights never pass through LDS: every weight is used by exactly one wave per tile, so the A operands are
//     read straight from L2 as 16-byte words out of a FRAGMENT-ORDERED image (block (m-tile, K-quad) = 64 lanes x
//     4 K-steps, 1 KiB contiguous per wave load), a layer ahead of their use;
//   * the spline's (row, dim) tasks and LULinear (two chained 16 x 16 MFMA mat-vecs) are spread over the waves;
//   * the backward pass walks all T transforms of its tile in ONE launch (the state gradient stays in LDS), computes
//     the weight gradients of its tile as K = rows MFMA tiles from transposed LDS tiles, and writes them as one
//     partial slab per workgroup and transform, summed by the same fixed-order reduction as the throughput path.
// Reference path replaced: the same as nsf_flow_kernel.h / nsf_train_kernel.h (nflows Flow.log_prob and its
// autograd backward behind NFlowsFlow.loss, sbi/neural_nets/estimators/nflows_flow.py:77-109, called per
// minibatch by sbi/inference/trainers/base.py:1150-1193).
#include "nsf_plan.h"

#define CO_WAVES 4
#define CO_MAX_NT 2          // row tiles (of 16) per workgroup
#define CO_MAX_MATS 40
#define CO_MAX_BIAS 16

// one matrix of the fragment-ordered image: block (mt, q) at off + (mt * quads + q) * 256 floats; lane l, element r
// of a block holds  M[m = 16 mt + iperm(l & 15)][k = 4 (4 q + r) + (l >> 4)]  -- the A operand of K-step 4 q + r.
// (iperm(i) = 4 (i & 3) + (i >> 2): with it the D fragment of a GEMM (lane (g, j), register r <-> row 16 mt + 4 r + g)
//  is the B fragment of the next GEMM's K-step 4 mt + r, as in nsf_device.h.)
struct CoMat {
  int off, mtiles, quads;
  int kind;      // CO_K_*: how (m, k) maps to the flat parameters
  int lin;       // index into ShapeDesc::lin (or -1)
};
enum {
  CO_K_W0 = 0,     // initial layer, K order [context: KCQ quads ; identity features: one quad (<= 8 used)]
  CO_K_PLAIN,      // W[m][k]
  CO_K_WF,         // final layer: m-tile = (dim, 16-param tile), K = hidden
  CO_K_WFT,        // final layer transposed: m = hidden, k = (dim, param padded to 16 PT)
  CO_K_PLAIN_T,    // W[k][m]
  CO_K_W0T,        // initial layer transposed, identity columns only: m = identity slot, k = hidden
  CO_K_U, CO_K_L, CO_K_UT, CO_K_LT,
  CO_K_CTX_T,      // context columns of a layer, transposed: m = context feature, k = hidden (d loss / d embedded x)
  CO_K_UI, CO_K_LI // explicit inverses U^-1, L^-1 (sampling direction of the wide kernels)
};
struct CoBias {     // D-fragment-ordered bias block: [mtile][16], element 4 g + r <-> row 16 mt + 4 r + g
  int off, mtiles, kind, lin;   // kind: 0 plain, 1 final layer (per (dim, tile)), 2 LULinear bias
};

struct CoShape {    // per mask parity
  CoMat W0, WC[NSF_MAX_NB], W1[NSF_MAX_NB], W2[NSF_MAX_NB], WF, U, L;           // forward
  CoMat WFT, W1T[NSF_MAX_NB], W2T[NSF_MAX_NB], W0T, UT, LT;                      // backward
  CoMat WCT[NSF_MAX_NB], W0CT;                                                   // backward, d loss / d context
  CoMat UI, LI;                                                                  // sampling direction (wide nets only)
  CoBias b0, bc[NSF_MAX_NB], b1[NSF_MAX_NB], b2[NSF_MAX_NB], bf, blu;
  int o_bias;       // first float behind the matrices (256-aligned): bias blocks, then
  int o_ld;         // the slot holding sum_i log U_ii
  int KCQ;          // K-quads (16 context features each) of the context part of the initial / gate layers (1 or 2)
  int nft;          // final-layer m-tiles = d_tr * PT
  // partial weight-gradient slab of one (workgroup, transform): 256-float tiles in D-fragment order (lane l holds four
  // consecutive inputs 16 nt + 4 (l >> 4) + r of output 16 mt + (l & 15); the bias is input index L.in), one
  // 16-byte store per lane and tile; linear k's tile (mt, nt) is dw_tb[k] + mt * dw_nnt[k] + nt
  int dw_tb[NSF_MAX_LIN], dw_nnt[NSF_MAX_LIN];
  int dw_tail;      // float offset of the LULinear block (natural order: lower, upper, diag, bias, sum d/d logabsdet)
};

struct CoopPlan {
  CoShape sh[2];
  int img_floats;           // floats per transform in the coop image
  int MT;                   // hidden m-tiles per wave: 1 (hidden <= 64), 2 (hidden <= 128: nsf_coop_wide_kernel.h)
  int NT, R, RS;            // row tiles per workgroup, rows = 16 NT, row stride of the transposed tiles
  int ZS;                   // row stride of the state rows (odd)
  int PSW, DSTR;            // spline-parameter staging: floats per (row, dim), per row
  int slots;                // stash slots (256 floats each) per (transform, 16-row tile)
  int s_blk, s_par;         // slot of block 0's first entry / of the first parameter tile
  // LDS offsets (floats)
  int o_zs, o_gys, o_gzs, o_w, o_pst, o_ex, o_ldp, o_gt, o_at, o_ct, o_lut, o_ctx, lds_floats;
  int ct_rows;              // rows of the static conditioner-input tile
  int grid;                 // workgroups
  int PLP;                  // floats per (workgroup, transform) partial-gradient slab (tile order, see CoShape)
};

// Compact kernel constants of the forward / backward kernels (~0.5 KB of kernarg instead of the 3.5 KB of both plans):
// everything those kernels read, as plain scalars the compiler keeps in SGPRs.  A kernel that walks the plan structs
// pays one scalar-cache round trip (~200 cycles, and an lgkmcnt wait that also drains its LDS reads) per field it
// touches -- measured at a third of the backward kernel's time.
struct CoKP {      // per mask parity
  int d_id, d_tr, in0, nft;
  int nnt0;        // n-tiles of d W0 (conditioner input + bias column)
  int dw_tail;     // slab offset of the LULinear block
  int w0, wc0, w10, w20, wf, u, l, wft, w1t0, w2t0, w0t, ut, lt, wct0, w0ct;   // image offsets (block b: + b * stride)
  int ui, li;      // U^-1, L^-1 (wide nets)
  int b0, bc0, b10, b20, bf, blu, ld;
};
struct CoK {
  int D, C, H, NB, T, P, KCQ;
  int HT, HQ;                  // hidden m-tiles (4 or 8) and K-quads of a hidden-K matrix
  int sA, sT, sC, sB;          // per-block strides of [WC W1 W2], [W1T W2T], [WCT] and of the bias blocks [bc b1 b2]
  int img_floats;
  int ZS, RS, PSW, DSTR;
  int o_zs, o_gys, o_gzs, o_w, o_pst, o_ex, o_ldp, o_gt, o_at, o_ct, o_lut, o_ctx, ct_rows;
  int slots, s_blk, s_par;
  int PLP, ntc, nnh;           // slab: floats per (workgroup, transform); n-tiles of d Wc and of the hidden-input layers
  int ablate;
  float B, min_w, min_h, min_d, inv_sqrt_h, one_minus_kw, one_minus_kh, d_const, log_z;
  CoKP p[2];
};
void coop_make_consts(const NsfPlan& pl, const CoopPlan& cp, CoK* k);

// 0 or SBI_AMD_E_*: which shapes the cooperative kernels take (everything else keeps the throughput kernels)
int coop_build_plan(const NsfPlan& pl, int64_t n, int nt_force, bool training, CoopPlan* cp);
// rows up to which the cooperative path is preferred (env SBI_AMD_COOP_MAX_ROWS overrides; 0 disables)
int64_t coop_max_rows();
int64_t coop_train_rows();  // the same for the training pass (default 8 192)
bool coop_lean_forward();   // forward pass of > 4096-row calls as one-tile workgroups, two to a CU (nsf_coop_plan.cpp)
static inline int64_t coop_image_floats(const NsfPlan& pl, const CoopPlan& cp) { return (int64_t)pl.T * cp.img_floats; }
