// nsf_coop_k5.hip -- num_bins = 5 instantiations of the cooperative (small-batch) kernels.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "nsf_train_kernel.h"
#include "nsf_coop_wide_kernel.h"

template int co_fwd_k<5>(const NsfPlan&, const CoopPlan&, const CoFwdArgs&, hipStream_t);
template int co_bwd_k<5>(const NsfPlan&, const CoopPlan&, const CoBwdArgs&, hipStream_t);
template int co_inv_k<5>(const NsfPlan&, const CoopPlan&, const float*, const float*, const float*, const float*,
                          long long, long long, float*, float*, hipStream_t);
