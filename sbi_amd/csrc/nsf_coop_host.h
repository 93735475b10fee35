#pragma once
// nsf_coop_host.h -- host entry points of the cooperative (small-batch) path (nsf_coop.hip), used by the C ABI
// functions in nsf_flow.hip / nsf_train.hip to route calls of <= coop_max_rows() rows.
#include "nsf_coop.h"

bool coop_applies(const sbi_amd_nsf_config* cfg, int64_t n, bool training, NsfPlan* pl, CoopPlan* cp);
bool coop_shape_ok(const sbi_amd_nsf_config* cfg, NsfPlan* pl, CoopPlan* cp);
int64_t coop_packed_floats(const sbi_amd_nsf_config* cfg);
int coop_pack(const sbi_amd_nsf_config* cfg, const float* params, float* cimg, int which, void* stream);
int64_t coop_workspace_floats(const NsfPlan& pl, const CoopPlan& cp, int64_t n);
const float* coop_sqnorm_parts(const NsfPlan& pl, const CoopPlan& cp, int64_t n, const float* workspace, int64_t* n_parts);
int coop_log_prob(const sbi_amd_nsf_config* cfg, const NsfPlan& pl, const CoopPlan& cp, const float* cimg,
                  const float* zstats, const float* theta, const float* x, int64_t n, int64_t x_rows, float* logp,
                  float* noise, void* stream);
int coop_sample(const sbi_amd_nsf_config* cfg, const NsfPlan& pl, const CoopPlan& cp, const float* cimg,
                const float* zstats, const float* noise, const float* x, int64_t n, int64_t x_rows, float* theta_out,
                float* logabsdet_out, void* stream);
int coop_train_forward(const sbi_amd_nsf_config* cfg, const NsfPlan& pl, const CoopPlan& cp, const float* cimg,
                       const float* zstats, const float* theta, const float* x, int64_t n, int64_t x_rows,
                       float* logp_out, float* workspace, void* stream);
int coop_train_backward(const sbi_amd_nsf_config* cfg, const NsfPlan& pl, const CoopPlan& cp, const float* params,
                        const float* cimg, const float* zstats, const float* x, int64_t n, int64_t x_rows,
                        const float* row_weight, float uniform_weight, float* grad_out, float* grad_theta_out,
                        float* grad_x_out, float* loss_out, float* workspace, void* stream);
