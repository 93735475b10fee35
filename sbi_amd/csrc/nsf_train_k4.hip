// nsf_train_k4.hip -- num_bins = 4 instantiations of the backward kernel.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "nsf_train_kernel.h"

template int launch_bwd_k<4>(const NsfPlan&, const TrainPlan&, const BwdIo&, hipStream_t);
