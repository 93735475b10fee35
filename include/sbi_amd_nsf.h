/*
 * sbi_amd_nsf.h -- C ABI of the MI355X (gfx950) NSF / NPE hot-path library
 * (libsbi_amd_nsf.so, built from sbi_amd/csrc/ by hipcc).
 *
 * The reference (sbi-dev/sbi) is pure Python and has NO FFI for this path; the
 * arithmetic it executes lives in nflows 0.14 behind
 *     sbi/neural_nets/estimators/nflows_flow.py:77-151  (NFlowsFlow.log_prob /
 *     loss / sample / inverse_transform)
 * and the optimiser step in
 *     sbi/inference/trainers/base.py:1150-1193           (_train_epoch).
 * Each entry point below replaces the eager-op sequence behind ONE of those
 * Python calls; INTEGRATION.md shows the ctypes stub a maintainer would add to
 * nflows_flow.py to bind them.
 *
 * Conventions
 *   - all pointers except `cfg` and `*_host` are DEVICE pointers (HBM), fp32,
 *     row-major, contiguous; `stream` is a hipStream_t (NULL = default stream)
 *   - every call is asynchronous on `stream`; nothing allocates or syncs
 *   - return value: 0 = launched; <0 = SBI_AMD_E_* (nothing launched);
 *                   >0 = hipError_t from the launch
 *   - `params` is ONE flat fp32 buffer in nflows' natural parameter order; `packed`
 *     is its kernel-side image (sbi_amd_nsf_pack)
 *     (see sbi_amd_nsf_param_count / DESIGN.md "flat parameter layout"):
 *     per transform t = 0..T-1:
 *        initial_layer.weight (H, d_id+C), .bias (H)
 *        per block b: context_layer.weight (H,C), .bias (H),
 *                     linear_layers.0.weight (H,H), .bias (H),
 *                     linear_layers.1.weight (H,H), .bias (H)
 *        final_layer.weight (d_tr*(3K-1), H), .bias (d_tr*(3K-1))
 *        LULinear: lower_entries, upper_entries (D(D-1)/2 each),
 *                  unconstrained_upper_diag (D), bias (D)
 *   - `zstats` is 2*D + 2*C floats: theta shift (D), theta scale (D)
 *     [PointwiseAffineTransform buffers, sbiutils.py:226-247], x mean (C),
 *     x std (C) [Standardize buffers, sbiutils.py:418-428]; no z-scoring =
 *     shift 0 / scale 1 / mean 0 / std 1
 *   - condition broadcasting: `x` has `x_rows` rows; row n of the batch uses
 *     x[n % x_rows] (x_rows == n: paired; x_rows == 1: single x_o; x_rows == B
 *     with n == S*B: NFlowsFlow's (S,B) flattening, nflows_flow.py:91-93)
 */
#ifndef SBI_AMD_NSF_H
#define SBI_AMD_NSF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBI_AMD_E_UNSUPPORTED (-1) /* config outside what the kernels are instantiated for */
#define SBI_AMD_E_BADARG (-2)
#define SBI_AMD_E_LDS (-3) /* config needs more than 160 KiB LDS per workgroup */

typedef struct sbi_amd_nsf_config {
  int32_t D;          /* theta features (>= 2)                 flow.py:393  x_numel          */
  int32_t C;          /* embedded condition features           flow.py:394  y_numel          */
  int32_t H;          /* hidden_features (<= 128; > 64: the wide cooperative kernels at every batch size) flow.py:343 */
  int32_t K;          /* num_bins (4,5,8,10,16)                flow.py:345                   */
  int32_t T;          /* num_transforms (<= 16)                flow.py:344                   */
  int32_t NB;         /* num_blocks of the ResidualNet (<= 4)  flow.py:349                   */
  float tail_bound;   /* spline tail bound                     flow.py:347                   */
  float min_bin_width, min_bin_height, min_derivative; /* nflows defaults 1e-3 (estimator_configs.py:49-51) */
  float lu_eps;       /* LULinear eps, nflows default 1e-3 */
  int32_t ctx_layers; /* theta-dim 1 only (ContextSplineMap, flow.py:1419-1478): hidden_layers_spline_context, the number
                       * of times the ONE shared hidden Linear + ReLU is applied (flow.py:346; 1 ... 4); 0 is read as 1, so a
                       * zero-initialised trailing field keeps the default; -1: no hidden layer (hidden_layers_spline_context
                       * = 0: the flat buffer then holds no hidden Linear); ignored for theta-dim >= 2 */
} sbi_amd_nsf_config;

/* Number of floats in the flat parameter buffer for `cfg` (98 025 for the
 * default D=C=10,H=50,K=10,T=5,NB=2); <0 on unsupported config. */
int64_t sbi_amd_nsf_param_count(const sbi_amd_nsf_config* cfg);

/* Float offset of transform t's block inside the flat buffer, and of its
 * LULinear sub-block (host helper for state_dict <-> flat conversion). */
int64_t sbi_amd_nsf_layer_offset(const sbi_amd_nsf_config* cfg, int32_t t);
int64_t sbi_amd_nsf_lu_offset(const sbi_amd_nsf_config* cfg, int32_t t);

/* The compute kernels read the weights from a PACKED image (per transform: the
 * MFMA A-operand layout one workgroup stages into LDS, with LULinear's L/U
 * expanded -- what nflows' LULinear._create_lower_upper rebuilds on every call).
 * sbi_amd_nsf_pack converts flat `params` -> `packed`
 * (sbi_amd_nsf_packed_floats(cfg) floats); call it whenever params changed. */
int64_t sbi_amd_nsf_packed_floats(const sbi_amd_nsf_config* cfg);
int sbi_amd_nsf_pack(const sbi_amd_nsf_config* cfg, const float* params, float* packed, void* stream);

/* `packed` holds two images: the throughput kernels' (one LDS image per transform) and, for the shapes they take
 * (theta-dim 2..16, x-dim <= 32; x-dim <= 64 with hidden_features 65..128), the fragment-ordered image of the latency-oriented kernels that calls of at most
 * 12 288 rows (training passes: 8 192) are routed to (four cooperating wavefronts per 16-row tile, all transforms of the backward pass in one
 * launch: csrc/nsf_coop.h; sbi's default training_batch_size = 200, npe_base.py:301-316, lives here).
 * sbi_amd_nsf_image_kind says which image an n-row call reads (0 throughput, 1 cooperative; `training` != 0 for
 * train_forward / train_backward / loss_fwd_bwd); sbi_amd_nsf_pack_images re-packs only the images named in the
 * bit mask (1 throughput, 2 cooperative; 8 / 4: the explicit U^-1 / L^-1 of the throughput / cooperative image, which
 * only sbi_amd_nsf_sample reads: an n-row sampling call needs bit 4 (with bit 2) whenever
 * sbi_amd_nsf_image_kind(cfg, n, 0) == 1 -- small calls of narrow nets and EVERY call at hidden_features > 64 -- and
 * bit 8 (with bit 1) otherwise) -- a training loop at a fixed batch size
 * needs ONE of bits 1 / 2 per step and never the inverses.  sbi_amd_nsf_pack packs everything. */
int sbi_amd_nsf_image_kind(const sbi_amd_nsf_config* cfg, int64_t n, int32_t training);
/* Host-side answer (tests, tuning): waves of 16 rows per workgroup the throughput forward kernel (sampling != 0: the
 * sampling-direction kernel) launches for an n-row call -- chosen to minimise the rounds over the CUs -- or SBI_AMD_E_*. */
int sbi_amd_nsf_plan_waves(const sbi_amd_nsf_config* cfg, int64_t n, int32_t sampling);
/* Tuning / test hook: calls of at most `rows` rows take the cooperative kernels (0: never; default 12 288, and 8 192
 * for training passes; the environment variable SBI_AMD_COOP_MAX_ROWS, like this call, sets both); returns the
 * previous value.  Process-wide; images packed before a
 * change stay valid (both images live in `packed`), but a training workspace belongs to the path that sized it. */
int64_t sbi_amd_nsf_set_coop_max_rows(int64_t rows);
int sbi_amd_nsf_pack_images(const sbi_amd_nsf_config* cfg, const float* params, float* packed, int32_t images,
                            void* stream);

/* log p(theta | x) for n rows: replaces nflows Flow.log_prob behind
 * NFlowsFlow.log_prob (nflows_flow.py:77-97).  noise_out (n,D) is optional
 * (NULL) and receives transform(theta) = NFlowsFlow.inverse_transform
 * (nflows_flow.py:43-75). */
int sbi_amd_nsf_log_prob(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats,
                         const float* theta, const float* x, int64_t n, int64_t x_rows,
                         float* logp_out, float* noise_out, void* stream);

/* theta = transform^{-1}(noise | x) for n rows: the arithmetic of
 * Flow._sample behind NFlowsFlow.sample (nflows_flow.py:111-128) for GIVEN
 * base noise (the caller draws it with torch's generator, see DESIGN.md RNG).
 * logabsdet_out (n) optional: log|det d theta/d noise|. */
int sbi_amd_nsf_sample(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats,
                       const float* noise, const float* x, int64_t n, int64_t x_rows,
                       float* theta_out, float* logabsdet_out, void* stream);

/* Training pass for one minibatch: per-row loss = -log p (NFlowsFlow.loss,
 * nflows_flow.py:99-109) and d(sum_n w_n * loss_n)/d params accumulated into
 * grad_out (P floats, OVERWRITTEN), w_n = row_weight[n] or `uniform_weight`
 * when row_weight is NULL (1/B gives the gradient of the batch mean,
 * trainers/base.py:1178-1181).  grad_theta_out (n,D) optional: w_n * d loss_n /
 * d theta_n (needed by MAP / gradient_ascent, base_posterior.py:216-323).  grad_x_out (n,C) optional, needs
 * x_rows == n: w_n * d loss_n / d x_n, the gradient a trainable embedding net in front of the flow
 * back-propagates (flow.py:1395-1416 puts `standardizing_net -> embedding_net` there).
 * `workspace` must hold sbi_amd_nsf_train_workspace_floats(cfg, n) floats.
 * Which shapes train is decided on the HOST by sbi_amd_nsf_train_workspace_floats (a negative return is the
 * refusal: no device call has happened): theta-dim <= 15, <= 2 blocks, identity features + x-dim <= 32 take the
 * wave-specialised backward kernel; wider shapes (theta-dim up to 24 with x-dim 32, x-dim up to 94 at theta-dim 10
 * with hidden 50 / 10 bins / 2 blocks; 3-4 blocks) take the generic pass (row-parallel backward kernel + split-K
 * weight-gradient GEMMs, more workspace, no grad_x_out); SBI_AMD_E_LDS once one transform's weight image plus the
 * kernel's tiles exceed 160 KiB of LDS. */
int64_t sbi_amd_nsf_train_workspace_floats(const sbi_amd_nsf_config* cfg, int64_t n);
int sbi_amd_nsf_loss_fwd_bwd(const sbi_amd_nsf_config* cfg, const float* params, const float* packed,
                             const float* zstats,
                             const float* theta, const float* x, int64_t n, int64_t x_rows,
                             const float* row_weight, float uniform_weight, float* loss_out,
                             float* grad_out, float* grad_theta_out, float* grad_x_out, float* workspace,
                             void* stream);

/* The same pass in two calls, for losses whose row weights depend on the log-probabilities themselves
 * (the atomic proposal-posterior loss of multi-round NPE-C, npe_c.py:356-440: w = d loss / d log p needs the
 * softmax over each row's atoms).  `train_forward` writes log p (n) and leaves the state / activation stash
 * in `workspace`; `train_backward` consumes that stash: same cfg, x, n, x_rows, packed image, stream order.
 * grad_out (P, OVERWRITTEN) = d( sum_n w_n * (-log p_n) ) / d params. */
int sbi_amd_nsf_train_forward(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats,
                              const float* theta, const float* x, int64_t n, int64_t x_rows,
                              float* logp_out, float* workspace, void* stream);
int sbi_amd_nsf_train_backward(const sbi_amd_nsf_config* cfg, const float* params, const float* packed,
                               const float* zstats, const float* x, int64_t n, int64_t x_rows,
                               const float* row_weight, float uniform_weight, float* grad_out,
                               float* grad_theta_out, float* grad_x_out, float* workspace, void* stream);

/* The training loop's minibatch sampler (csrc/shuffle.hip): rows offset .. offset + count of a fresh pseudo-random
 * order of n_perm training rows, gathered in ONE launch: a_out[i] = a[base_idx[pi(offset + i)]] (rows of da floats),
 * b_out likewise; idx_out (int64) receives the source rows.  Replaces SubsetRandomSampler + DataLoader collation of
 * trainers/base.py:541-560 (torch.randperm of the training split per epoch, drop_last batches).  pi is a keyed
 * permutation of [0, n_perm) (6-round Feistel + cycle walking), one `key` per epoch; base_idx NULL = identity (the
 * split is the first n_perm rows); any of a_out / b_out / idx_out may be NULL. */
int sbi_amd_shuffled_gather(const float* a, int32_t da, const float* b, int32_t db, const int64_t* base_idx,
                            int64_t n_perm, uint64_t key, int64_t offset, int64_t count, float* a_out, float* b_out,
                            int64_t* idx_out, void* stream);

/* Fused global-norm clip + Adam on the flat buffer: replaces
 * clip_grad_norm_(max_norm) + torch.optim.Adam.step (trainers/base.py:1181-1187,
 * :1097).  `step` is the 1-based step count; max_norm <= 0 disables clipping.
 * scratch: >= 130 floats (scratch[0] receives the pre-clip gradient norm). */
int sbi_amd_adam_clip_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq,
                           int64_t count, int64_t step, float lr, float beta1, float beta2, float eps,
                           float max_norm, float* scratch, void* stream);

/* clip_grad_norm_ needs |grad|^2; the training pass's gradient reduction leaves it as partial sums in its workspace (a
 * rider of the reduction kernel: each workgroup squares what it writes), which saves the separate pass over the gradient:
 *   parts = sbi_amd_nsf_train_sqnorm_parts(cfg, n, workspace, &n_parts)   after loss_fwd_bwd / train_backward of n rows
 *     (NULL / 0: that pass leaves none -- the generic training pass -- use sbi_amd_adam_clip_step)
 *   sbi_amd_adam_clip_step_parts(..., parts, n_parts, scratch, stream)    instead of sbi_amd_adam_clip_step
 * Only valid while grad_out is what the reduction wrote: a data-parallel all-reduce over MORE THAN ONE rank changes the
 * gradient, and the norm must then be taken from the reduced gradient (sbi_amd_adam_clip_step). */
const float* sbi_amd_nsf_train_sqnorm_parts(const sbi_amd_nsf_config* cfg, int64_t n, const float* workspace,
                                            int64_t* n_parts);
int sbi_amd_adam_clip_step_parts(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                                 int64_t step, float lr, float beta1, float beta2, float eps, float max_norm,
                                 const float* sqnorm_parts, int64_t n_parts, float* scratch, void* stream);

/* Table-driven re-pack (csrc/step_tail.hip).  A training step ends with the weight image of the next step's kernels
 * being rebuilt from the flat parameters -- what nflows redoes inside every forward call (LULinear._create_lower_upper
 * and the .t() views, nflows transforms/lu.py) and sbi_amd_nsf_pack_images does in 11 - 13 us by re-deriving where
 * every parameter goes.  That never changes between steps:
 *   sbi_amd_nsf_build_step_map(cfg, images, params, packed, map, workspace, stream)   -- ONCE per network
 *     images: 1 (throughput image), 2 (cooperative image) or 3 -- what sbi_amd_nsf_image_kind says the training
 *     batches read; map: sbi_amd_nsf_step_map_ints(cfg) int32; workspace: sbi_amd_nsf_step_map_workspace_floats(cfg)
 *     floats (scratch, free afterwards).  The table (image position -> parameter | softplus flag, constant, or a
 *     transform's logabsdet) is MEASURED by running the pack kernels on probe vectors and verified bit for bit
 *     against them on a random vector before it is returned (SBI_AMD_E_UNSUPPORTED otherwise: keep using
 *     sbi_amd_nsf_pack_images).  `packed` is fully packed from `params` on return.  Synchronises the stream.
 *   sbi_amd_nsf_table_pack(cfg, params, packed, map, stream)                          -- per optimizer step
 *     rewrites the parameter-dependent positions of the table's image(s): bit-identical to
 *     sbi_amd_nsf_pack_images(cfg, params, packed, images) on that buffer.  The other image and the explicit LU
 *     inverses are NOT touched (stale after a parameter update: sbi_amd_nsf_pack_images before a call that reads them). */
int64_t sbi_amd_nsf_step_map_ints(const sbi_amd_nsf_config* cfg);
int64_t sbi_amd_nsf_step_map_workspace_floats(const sbi_amd_nsf_config* cfg);
int sbi_amd_nsf_build_step_map(const sbi_amd_nsf_config* cfg, int32_t images, const float* params, float* packed,
                               int32_t* map, float* workspace, void* stream);
int sbi_amd_nsf_table_pack(const sbi_amd_nsf_config* cfg, const float* params, float* packed, const int32_t* map,
                           void* stream);
/* sbi_amd_nsf_table_pack only accepts a `map` that sbi_amd_nsf_build_step_map completed for the same configuration in
 * this process (SBI_AMD_E_BADARG otherwise: the kernel trusts the table's header and gather indices).  Before freeing
 * or reusing the map buffer, tell the library: a later allocation at the same address must not pass for a table. */
int sbi_amd_nsf_release_step_map(const int32_t* map);

/* Data-parallel exchange (SURVEY 8b `allreduce_flat`, 8e; the reference trains on one device and has none,
 * sbi/inference/trainers/base.py:1150-1193): ONE in-place SUM all-reduce of the flat fp32 gradient buffer per step over
 * RCCL (xGMI inside a node), enqueued on `stream` between the backward pass (sbi_amd_nsf_train_backward) and
 * sbi_amd_adam_clip_step*: every rank then clips and steps identically (csrc/adam_math.h) and replicas stay bit-identical.
 * librccl.so is resolved with dlopen on first use (SBI_AMD_E_UNSUPPORTED when it is absent); return codes >= 10000 are
 * 10000 + ncclResult_t.  One communicator per process and device:
 *   rank 0: sbi_amd_rccl_unique_id(id) -> ship the sbi_amd_rccl_unique_id_bytes() bytes to every rank (any channel) ->
 *   every rank, with its device current: sbi_amd_rccl_comm_init(&comm, world, rank, id) -> per step
 *   sbi_amd_allreduce_flat(comm, grad, count, stream) -> sbi_amd_rccl_comm_destroy(comm).
 * Python's default remains torch.distributed's `nccl` backend (the same RCCL): sbi_amd/utils/collectives.py. */
int32_t sbi_amd_rccl_unique_id_bytes(void);
int sbi_amd_rccl_unique_id(void* id_out);
int sbi_amd_rccl_comm_init(void** comm_out, int32_t world, int32_t rank, const void* id_bytes);
int sbi_amd_rccl_comm_destroy(void* comm);
int sbi_amd_allreduce_flat(void* comm, float* grad_bucket, int64_t count, void* stream);

/* Multi-round NPE-C, atomic proposal-posterior loss (sbi/inference/trainers/npe/npe_c.py:356-440): the host-side
 * arithmetic around the batched log_prob as two launches (csrc/atomic.hip).
 *   sbi_amd_atomic_atoms: for every row b of theta (batch, dim) draws num_atoms - 1 contrasting rows != b, uniform without
 *     replacement (npe_c.py:387-392; Philox keyed by `seed`, or taken from choices_in (batch, num_atoms - 1) when given),
 *     optionally returns them (choices_out) and writes the atom tensor atoms_out (num_atoms, batch, dim), atom 0 = the
 *     row itself -- atoms-major, so sbi_amd_nsf_train_forward(..., n = num_atoms * batch, x_rows = batch) pairs row r
 *     with x[r % batch] and the context is never repeated.  num_atoms <= 64 (SBI_AMD_E_UNSUPPORTED beyond).
 *   sbi_amd_atomic_weights: log_q, log_prior (num_atoms, batch) -> log_prob_out (batch) = u[0] - logsumexp_a u[a],
 *     u = log_q - log_prior (+ masks * log_q[0] for the combined loss, npe_c.py:425-436), and weights_out (num_atoms,
 *     batch) = scale * d log_prob_out[b] / d log_q[a, b]: the row weights of sbi_amd_nsf_train_backward, which
 *     differentiates sum_n w_n * (-log q_n) (scale = 1 / global_batch for the loss -log_prob_out). */
int sbi_amd_atomic_atoms(const float* theta, int32_t batch, int32_t num_atoms, int32_t dim, uint64_t seed,
                         const int64_t* choices_in, int64_t* choices_out, float* atoms_out, void* stream);
int sbi_amd_atomic_weights(const float* log_q, const float* log_prior, const float* masks, int32_t batch,
                           int32_t num_atoms, float scale, float* log_prob_out, float* weights_out, void* stream);

/* Stable stream compaction of accepted proposal draws: the body of accept_reject_sample's loop
 * (sbi/samplers/rejection/rejection.py:368-409 -- `candidates[are_accepted]` per condition, appended in order, plus the
 * running counts of the batch-size rule) as ONE launch (csrc/compact.hip: acceptance test, single-pass scan with
 * decoupled look-back, order-preserving scatter).
 *   candidates (batch_rows, num_xos, event_floats) fp32; acceptance from `accepted` (batch_rows, num_xos) bytes, or --
 *   accepted == NULL -- from the box lo <= theta <= hi (box_low / box_high: event_floats floats each; what torch's
 *   interval support check of a BoxUniform prior evaluates; NaN fails);
 *   out (num_samples, num_xos, event_floats): row r of condition x lands at state[x] + #accepted rows before r, rows
 *   past num_samples are dropped;  state int64[3 num_xos]: [0, X) rows filled (in/out, clamped to num_samples),
 *   [X, 2X) accepted so far (in/out), [2X, 3X) accepted by this call (out) -- the one read-back of an iteration;
 *   control int32[2 num_xos], zero before the first call (left zero by every call);  scan uint64[
 *   sbi_amd_accept_compact_scan_words(batch_rows, num_xos)], zero-initialised once;  generation != 0, different from
 *   the previous call's on the same scan buffer (a counter).  batch_rows < 2^31. */
int64_t sbi_amd_accept_compact_scan_words(int64_t batch_rows, int32_t num_xos);
int sbi_amd_accept_compact(const float* candidates, const uint8_t* accepted, const float* box_low, const float* box_high,
                           int64_t batch_rows, int32_t num_xos, int32_t event_floats, float* out, int64_t num_samples,
                           int64_t* state, int32_t* control, uint64_t* scan, uint32_t generation, void* stream);

/* One tick of the vectorised slice sampler for all chains (the loop body of SliceSamplerVectorized.run,
 * sbi/samplers/mcmc/slice_numpy.py:353-587): consumes the log-probabilities of `next_param` (what the batched
 * log_prob kernel just produced), advances every chain's BEGIN/LOWER/UPPER/SAMPLE_SLICE state, writes the next
 * evaluation points back into `next_param`, stores accepted sweeps into `samples`
 * (num_chains, num_samples, dim) after `tuning` width-tuning sweeps and counts finished chains in *done_count.
 * uniforms: (num_chains, 4 + dim) U[0,1) draws per tick from the caller's generator, or NULL: the kernel draws its own
 *   (Philox4x32-10, counter = (tick_no, chain, block), key = seed; the reference's sampler uses NumPy's global generator:
 *   there is no stream to reproduce, `seed` comes from torch's generator so that torch.manual_seed fixes a run).
 * istate: (num_chains, 4) int32 {state, dim index, sweep, -}; fstate: (num_chains, 8) {cxi, wi, lx, ux, xi, logu}.
 * theta_next / logabsdet_next (both or neither): the NEXT evaluation point mapped to constrained space by the
 *   transform `kind` (p0, p1: see sbi_amd_mcmc_to_constrained) and its log|det| -- what the batched log_prob kernel and
 *   this call's `logp_offset` read in the next tick, so a tick is two launches: log_prob, this. */
int sbi_amd_mcmc_slice_tick(int32_t num_chains, int32_t dim, int32_t num_samples, int32_t tuning, float max_width,
                            const float* logp, const float* logp_offset /* optional (num_chains): subtracted */,
                            const float* uniforms, float* x, float* next_param, float* width,
                            int32_t* order, int32_t* istate, float* fstate, float* samples, int32_t* done_count,
                            uint64_t seed, uint64_t tick_no, int32_t kind, const float* p0, const float* p1,
                            float* theta_next, float* logabsdet_next, void* stream);

/* The PERSISTENT form of the same loop: `nticks` ticks of all chains in ONE launch.  A workgroup of the cooperative
 * forward kernel owns 16 chains and alternates the log-density of their next evaluation points (one x_o: x_o is the
 * embedded, not yet standardized condition row, C floats) with the tick of their state machines; chains never interact,
 * so nothing is exchanged between workgroups and nothing returns to the host in between.  Requirements: a configuration
 * the cooperative kernels take (theta-dim 2 ... 16, hidden <= 64); `packed` holds the cooperative image
 * (sbi_amd_nsf_pack_images bit 2); theta_next (num_chains, dim) holds the constrained image of next_param on entry (and
 * on exit), logabsdet_next (num_chains) its log|det|; logp_scratch: num_chains floats; uniforms always in-kernel
 * (Philox: seed, tick0 + tick).  Poll *done_count between launches. */
int sbi_amd_mcmc_slice_run(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats, const float* x_o,
                           int32_t num_chains, int32_t num_samples, int32_t tuning, float max_width, float* x,
                           float* next_param, float* width, int32_t* order, int32_t* istate, float* fstate,
                           float* samples, int32_t* done_count, uint64_t seed, uint64_t tick0, int32_t nticks,
                           int32_t kind, const float* p0, const float* p1, float* theta_next, float* logabsdet_next,
                           float* logp_scratch, void* stream);

/* Unconstrained -> constrained parameters for the transforms of mcmc_transform (sbi/utils/sbiutils.py:867-980)
 * and the log|det| term of transformed_potential (sbi/utils/potentialutils.py:15-51) in one launch:
 * kind 0 identity; kind 1 theta = p0 + p1 * u (z-scoring with the prior's mean p0 / std p1);
 * kind 2 theta = p0 + p1 * sigmoid(u) (box [p0, p0 + p1]).  logabsdet_out = log|det d u / d theta|. */
int sbi_amd_mcmc_to_constrained(int32_t kind, int32_t num_chains, int32_t dim, const float* p0, const float* p1,
                                const float* u, float* theta_out, float* logabsdet_out, void* stream);

/* The coupling transform's spline on its own (SURVEY 8b `spline_coupling_fwd / inv`): nflows 0.14
 * unconstrained_rational_quadratic_spline (tails = "linear") exactly as the flow kernels evaluate it (the same device
 * routine: bin search on the fp32 knots, the selected bin re-derived beyond fp32 in the forward direction, clamped
 * discriminant / root in the inverse direction).  params (n, 3 K - 1) = [K width logits | K height logits | K - 1 interior
 * derivative pre-activations] per task; width / height logits are multiplied by `logit_scale` first (sbi's couplings:
 * 1 / sqrt(hidden_features); plain nflows splines: 1).  outputs (n); logabsdet (n, optional) = log|d output / d input|
 * of the direction that ran.  num_bins in {4, 5, 8, 10, 16}.  A test hook and a convenience for binders; the flow
 * kernels run the spline inside their own launches. */
int sbi_amd_rq_spline(int32_t num_bins, int32_t inverse, float tail_bound, float min_bin_width, float min_bin_height,
                      float min_derivative, float logit_scale, const float* params, const float* inputs, int64_t n,
                      float* outputs, float* logabsdet, void* stream);

/* Host-side consistency check of the cooperative kernels' address arithmetic against the plan tables, and of the
 * compile-time default layout against the run-time plan (no device work; used by the CPU tests).  0 = consistent,
 * -1 = the configuration has no cooperative image. */
int sbi_amd_nsf_coop_selfcheck(const sbi_amd_nsf_config* cfg);

/* Library/ABI version (major*100 + minor) and the gfx arch string it was built for. */
#define SBI_AMD_NSF_ABI_VERSION 114
int sbi_amd_nsf_abi_version(void);
const char* sbi_amd_nsf_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* SBI_AMD_NSF_H */
