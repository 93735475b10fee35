// nsf_train_k5.hip -- num_bins = 5 instantiations of the backward kernel.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "nsf_train_kernel.h"

template int launch_bwd_k<5>(const NsfPlan&, const TrainPlan&, const BwdIo&, hipStream_t);
