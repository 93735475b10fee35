/* sbi_amd_fmpe.h -- C ABI of the MI355X-native FMPE (flow-matching) vector-field path.
 *
 * Second hot path of libsbi_amd_nsf.so (BASELINE configs[4]): the default
 * vector-field MLP of sbi's flow-matching posterior estimator, its conditional-flow-matching loss with
 * gradients, and the velocity evaluation ODE samplers call.  Plain pointers and sizes only; every pointer is
 * a DEVICE pointer (fp32) unless noted; `stream` is a hipStream_t (NULL = default stream).
 *
 * What each entry point replaces in the reference (file:line under /root/reference/sbi):
 *   sbi_amd_fmpe_velocity        FlowMatchingEstimator.forward / ode_fn
 *                                neural_nets/estimators/flowmatching_estimator.py:206-274, 349-372
 *                                (VectorFieldMLP.forward  net_builders/vector_field_nets.py:683-719)
 *   sbi_amd_fmpe_velocity_div    the augmented right-hand side of VectorFieldPosterior.log_prob: velocity AND the exact
 *                                trace of its Jacobian wrt theta_t -- inference/posteriors/vector_field_posterior.py:467-504
 *                                -> potentials/vector_field_potential.py:149-207 -> samplers/ode_solvers/zuko_ode.py:19-124
 *                                (zuko's FreeFormJacobianTransform with exact=True: a batched-identity autograd trace;
 *                                here forward-mode tangents ride in the MFMA tile next to the primal)
 *   sbi_amd_fmpe_loss            FlowMatchingEstimator.loss (validation: no gradient)   :276-347
 *   sbi_amd_fmpe_loss_fwd_bwd    the same loss + loss.backward() of the training loop
 *                                inference/trainers/vfpe/base_vf_inference.py:443-470, trainers/base.py:1160-1190
 *   sbi_amd_fmpe_pack            (no counterpart) parameters -> MFMA operand images, once per optimizer step
 *
 * Flat parameter order (fp32, natural nn.Linear [out][in] layouts):
 *   input_layer.weight [H][D], .bias [H] | condition_layer.weight [H][C], .bias [H] |
 *   input_merge_layer.weight [H][2H], .bias [H] | time_linear_layer.weight [H][E], .bias [H] |
 *   for l < L: layers.l.weight [H][H], .bias [H] | for l < L: layers_norm.l.weight [H], .bias [H] |
 *   output_layer.weight [D][H], .bias [D]
 * zstats (fp32, 2D + 2C): mean_0 [D], std_0 [D] (time-dependent z-scoring of theta,
 *   flowmatching_estimator.py:120-147), x mean [C], x std [C] (Standardize, sbiutils.py:418-428).
 */
#ifndef SBI_AMD_FMPE_H
#define SBI_AMD_FMPE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sbi_amd_fmpe_config {
  int32_t D;          /* theta dim, 1..128 */
  int32_t C;          /* (embedded) x dim, 1..128 */
  int32_t H;          /* hidden_features, 16..128 (sbi default 100) */
  int32_t L;          /* num_layers, 1..8 (default 5) */
  int32_t E;          /* time embedding dim, even, 2..64 (default 32) */
  float max_freq;     /* sinusoidal_max_freq (1000) */
  float noise_scale;  /* sigma_min (1e-3) */
  float ln_eps;       /* LayerNorm eps (1e-5) */
} sbi_amd_fmpe_config;

/* Number of fp32 parameters / offset of one block in the flat buffer.
 * kind: 0 W_in 1 b_in 2 W_c 3 b_c 4 W_m 5 b_m 6 W_t 7 b_t 8 W_l 9 b_l 10 ln_w_l 11 ln_b_l 12 W_o 13 b_o. */
int64_t sbi_amd_fmpe_param_count(const sbi_amd_fmpe_config* cfg);
int64_t sbi_amd_fmpe_param_offset(const sbi_amd_fmpe_config* cfg, int32_t kind, int32_t layer);

/* Size of, and writer for, the packed operand images (forward W and backward W^T per linear, zero padded). */
int64_t sbi_amd_fmpe_packed_floats(const sbi_amd_fmpe_config* cfg);
int sbi_amd_fmpe_pack(const sbi_amd_fmpe_config* cfg, const float* params, float* packed, void* stream);

/* v_out[n][D] = velocity at (theta_t[n][D], x, times).  x has x_rows rows (1 = one observation for every
 * row, else n); times has t_rows entries (1 or n). */
int sbi_amd_fmpe_velocity(const sbi_amd_fmpe_config* cfg, const float* packed, const float* zstats,
                          const float* theta_t, const float* x, int64_t x_rows, const float* times,
                          int64_t t_rows, int64_t n, float* v_out, void* stream);

/* The same velocity (v_out may be NULL) and div_out[n] = sum_f d v_out[n][f] / d theta_t[n][f] (exact, fp32). */
int sbi_amd_fmpe_velocity_div(const sbi_amd_fmpe_config* cfg, const float* packed, const float* zstats,
                              const float* theta_t, const float* x, int64_t x_rows, const float* times,
                              int64_t t_rows, int64_t n, float* v_out, float* div_out, void* stream);

/* loss_out[n] = per-row conditional-flow-matching loss for theta[n][D], x, times[n] ~ U[0,1],
 * noise[n][D] ~ N(0, I) (the draws FlowMatchingEstimator.loss makes internally, made explicit). */
int sbi_amd_fmpe_loss(const sbi_amd_fmpe_config* cfg, const float* packed, const float* zstats,
                      const float* theta, const float* x, int64_t x_rows, const float* times,
                      const float* noise, int64_t n, float* loss_out, void* stream);

/* Workspace (floats) for one training pass over n rows: activation stash, gradient partials. */
int64_t sbi_amd_fmpe_train_workspace_floats(const sbi_amd_fmpe_config* cfg, int64_t n);

/* Per-row losses and grad_out[param_count] = d/dparams sum_i w_i loss_i, with w_i = row_weight[i] if
 * row_weight != NULL else uniform_weight (1/n for the mean loss). */
int sbi_amd_fmpe_loss_fwd_bwd(const sbi_amd_fmpe_config* cfg, const float* params, const float* packed,
                              const float* zstats, const float* theta, const float* x, int64_t x_rows,
                              const float* times, const float* noise, int64_t n, const float* row_weight,
                              float uniform_weight, float* loss_out, float* grad_out, float* workspace,
                              void* stream);

/* ---- ODE sampler: device-resident Dormand-Prince 5(4) step machinery (csrc/ode.hip) ---------------------------
 * Replaces, for the probability-flow ODE of VectorFieldPosterior.sample (inference/posteriors/
 * vector_field_posterior.py:436-466 -> samplers/ode_solvers/zuko_ode.py:19-124, zuko's adaptive odeint: a
 * third-party solver, its controller is the textbook one here), everything of a step that is not the velocity
 * evaluation.  `state` is a device buffer of SBI_AMD_DOPRI5_STATE_FLOATS fp32 slots owned by the caller:
 *   slot 16: signed step h of the current attempt;  slots 17..22: times of stages 2..7 (hand slot 16+i to
 *   sbi_amd_fmpe_velocity as its 1-element `times` for stage i+1);  slot 23: current t;  slot 24: 1.0 once t has
 *   reached t1 (the only value the host ever reads; an attempt after that is a no-op);  slots 25/26: accepted /
 *   rejected attempts;  slot 27: last error ratio;  slot 28: 1.0 if the current attempt reaches t1 when accepted.
 *   Slots 0..15 are the controller's doubles.
 * One attempt = for i = 1..6 { sbi_amd_dopri5_stage(i) -> y_stage; k[i] = velocity(y_stage, time slot 16+i) };
 * sbi_amd_dopri5_finish (y5 = the stage-6 state; on acceptance y <- y5 and k[0] <- k[6] in place).
 * k: HOST array of 7 DEVICE pointers k1..k7 (n floats each), n = rows x D. scratch: 256 doubles (device). */
#define SBI_AMD_DOPRI5_STATE_FLOATS 32
int sbi_amd_dopri5_init(float* state, double t0, double t1, double first_step, double atol, double rtol,
                        void* stream);
int sbi_amd_dopri5_stage(const float* y, const float* const* k, int32_t stage, const float* state, float* y_stage,
                         int64_t n, void* stream);
int sbi_amd_dopri5_finish(float* y, const float* y5, const float* const* k, float* state, double* scratch,
                          int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
