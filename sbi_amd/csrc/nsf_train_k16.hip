// nsf_train_k16.hip -- num_bins = 16 instantiations of the backward kernel.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "nsf_train_kernel.h"

template int launch_bwd_k<16>(const NsfPlan&, const TrainPlan&, const BwdIo&, hipStream_t);
