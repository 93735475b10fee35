// maf_k5.hip -- num_bins = 5 instantiations of the maf_rqs kernels (separate translation unit: parallel build)
#include "maf_kernel.h"
template int maf_dispatch_k<5>(const MafPlan&, int, int, const float*, const float*, const float*, const float*,
                                int64_t, int64_t, float*, float*, float*, const MafBwdArgs*, hipStream_t);
