// maf_k16.hip -- num_bins = 16 instantiations of the maf_rqs kernels (separate translation unit: parallel build)
#include "maf_kernel.h"
template int maf_dispatch_k<16>(const MafPlan&, int, int, const float*, const float*, const float*, const float*,
                                int64_t, int64_t, float*, float*, float*, const MafBwdArgs*, hipStream_t);
