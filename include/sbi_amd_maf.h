/*
 * sbi_amd_maf.h -- C ABI of the MI355X (gfx950) `maf_rqs` path: masked autoregressive flow with
 * rational-quadratic-spline transforms (SURVEY.md section 8 row a19 "MADE masked linear" and (f)4).
 * Same library (libsbi_amd_nsf.so), same conventions as sbi_amd_nsf.h (device pointers, fp32 row-major,
 * asynchronous on `stream`, return 0 / SBI_AMD_E_* / hipError_t, condition row = x[n % x_rows]).
 *
 * Reference path replaced (pure Python, no FFI): NFlowsFlow(build_maf_rqs(...)).log_prob / loss / sample
 *     sbi/neural_nets/net_builders/flow.py:212-330       (wiring: T x [MaskedPiecewiseRationalQuadratic-
 *                                                          AutoregressiveTransform, RandomPermutation])
 *     sbi/neural_nets/estimators/nflows_flow.py:77-151   (estimator surface)
 * whose arithmetic lives in nflows 0.14 (transforms/made.py: MaskedLinear / MADE, transforms/autoregressive.py,
 * transforms/permutations.py, transforms/splines/rational_quadratic.py).  INTEGRATION.md shows the ctypes stub.
 *
 * Flat parameter layout (`params`, nflows' natural order), per transform t = 0..T-1:
 *     autoregressive_net.initial_layer.weight (H, D), .bias (H)        MaskedLinear, hidden degrees
 *     autoregressive_net.context_layer.weight (H, C), .bias (H)        nn.Linear
 *     per block b: autoregressive_net.blocks.b.linear.weight (H, H), .bias (H)   MaskedLinear
 *     autoregressive_net.final_layer.weight (D*(3K-1), H), .bias (D*(3K-1))      MaskedLinear, output degrees
 * The degree masks are static (made.py: inputs 1..D, hidden `i % max(1, D-1) + min(1, D-1)`, outputs
 * `repeat(1..D, 3K-1)`; hidden `>=`, output `>`): sbi_amd_maf_pack multiplies them into the packed image, the
 * gradient kernels multiply them into the weight gradients -- masked entries of `params` never matter.
 * `perms`: T x D int32, RandomPermutation._permutation of every transform (forward: out[d] = in[perm[d]]).
 */
#ifndef SBI_AMD_MAF_H
#define SBI_AMD_MAF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sbi_amd_maf_config {
  int32_t D;          /* theta features (1..16)                  flow.py:281  x_numel        */
  int32_t C;          /* embedded condition features (1..32)     flow.py:286  y_numel        */
  int32_t H;          /* hidden_features (<= 64)                 flow.py:222                 */
  int32_t K;          /* num_bins (4,5,8,10,16)                  flow.py:226                 */
  int32_t T;          /* num_transforms (<= 16)                  flow.py:223                 */
  int32_t NB;         /* num_blocks: feed-forward blocks (<= 4)  flow.py:225                 */
  float tail_bound;   /* flow.py:228 */
  float min_bin_width, min_bin_height, min_derivative;   /* flow.py:231-233 */
  int32_t scale_by_sqrt_hidden;   /* 0: nflows' MADE has no `hidden_features` attribute, the spline logits are used
                                     as produced (default); 1: divide width / height logits by sqrt(H) */
  int32_t variant;    /* 0: nflows maf_rqs (above).  1: zuko NSF (sbi build_zuko_nsf, flow.py:578-640 ->
                         zuko.flows.NSF): per transform a masked MLP hyper-net on [z ; context]
                         (Linear(D+C, H), NB x Linear(H, H), Linear(H, D*(3K-1)); ReLU; adjacency masks given by the
                         caller in `masks`), no context layer, no permutation (`perms` holds each transform's
                         autoregressive ORDER: arange / reversed), zuko's MonotonicRQSTransform parametrisation
                         (soft-clipped logits, exp slopes, tail_bound 5); min_* and scale_by_sqrt_hidden unused.
                         Flat layout: hyper.0 (H, D+C), hyper.2 .. (H, H), last (D*(3K-1), H), each weight then bias */
} sbi_amd_maf_config;

/* Floats in the flat parameter buffer / in the packed weight image; <0 = SBI_AMD_E_*. */
int64_t sbi_amd_maf_param_count(const sbi_amd_maf_config* cfg);
int64_t sbi_amd_maf_packed_floats(const sbi_amd_maf_config* cfg);
/* Float offset of linear `which` (0 initial, 1 context, 2+b block b, 2+NB final) of transform t; `bias` != 0
 * selects its bias vector. */
int64_t sbi_amd_maf_param_offset(const sbi_amd_maf_config* cfg, int32_t t, int32_t which, int32_t bias);

/* flat params (+ the permutations) -> packed image (masked weights in the MFMA operand layout). */
/* `masks`: optional 0/1 floats in the layout of `params` (bias slots ignored); NULL = the MADE degree formulas of
 * variant 0; required for variant 1. */
int sbi_amd_maf_pack(const sbi_amd_maf_config* cfg, const float* params, const int32_t* perms, const float* masks,
                     float* packed, void* stream);

/* Flow.log_prob: logp_out[n] = log p(theta_n | x_{n % x_rows}); noise_out (n, D) optional (transform output).
 * One launch: z-scoring, T x [MADE on MFMA -> RQ spline on all D dims -> permutation], base density. */
int sbi_amd_maf_log_prob(const sbi_amd_maf_config* cfg, const float* packed, const float* zstats,
                         const float* theta, const float* x, int64_t n, int64_t x_rows, float* logp_out,
                         float* noise_out, void* stream);

/* Flow._sample's inverse for GIVEN noise: theta_out (n, D) = transform^{-1}(noise | x); logabsdet_out optional.
 * The autoregressive inverse runs D conditioner passes per transform (pass i finalises dimension i). */
int sbi_amd_maf_sample(const sbi_amd_maf_config* cfg, const float* packed, const float* zstats, const float* noise,
                       const float* x, int64_t n, int64_t x_rows, float* theta_out, float* logabsdet_out,
                       void* stream);

/* Training pass: loss_out[n] = -log p_n (optional), grad_out (param_count) = d( sum_n w_n loss_n ) / d params with
 * w_n = row_weight[n] (or uniform_weight when row_weight is NULL), grad_theta_out (n, D) optional.
 * Forward with the per-transform input stash, then per transform (last -> first) a row-parallel backward kernel
 * (conditioner recompute, spline forward + reverse mode, back-propagation through the masked layers) and a
 * split-K MFMA GEMM for the weight gradients, then a fixed-order reduction (deterministic, no atomics). */
int64_t sbi_amd_maf_train_workspace_floats(const sbi_amd_maf_config* cfg, int64_t n);
int sbi_amd_maf_loss_fwd_bwd(const sbi_amd_maf_config* cfg, const float* packed, const float* zstats,
                             const float* masks, const float* theta, const float* x, int64_t n, int64_t x_rows,
                             const float* row_weight, float uniform_weight, float* loss_out, float* grad_out,
                             float* grad_theta_out, float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif
