// nsf_train.hip -- NPE training pass (loss + gradients); placeholder until the
// backward kernels land (returns SBI_AMD_E_UNSUPPORTED, loudly, never a fallback).
#include <hip/hip_runtime.h>
#include "nsf_plan.h"

extern "C" int64_t sbi_amd_nsf_train_workspace_floats(const sbi_amd_nsf_config* cfg, int64_t n) {
  (void)cfg; (void)n;
  return SBI_AMD_E_UNSUPPORTED;
}
extern "C" int sbi_amd_nsf_loss_fwd_bwd(const sbi_amd_nsf_config* cfg, const float* params, const float* packed,
                                        const float* zstats,
                                        const float* theta, const float* x, int64_t n, int64_t x_rows,
                                        const float* row_weight, float uniform_weight, float* loss_out,
                                        float* grad_out, float* grad_theta_out, float* workspace, void* stream) {
  return SBI_AMD_E_UNSUPPORTED;
}
