// nsf_train_k8.hip -- num_bins = 8 instantiations of the backward kernel.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "nsf_train_kernel.h"

template int launch_bwd_k<8>(const NsfPlan&, const TrainPlan&, const BwdIo&, hipStream_t);
