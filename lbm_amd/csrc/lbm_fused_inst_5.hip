// Multi-step sweep kernels, family set 5 of lbm_fused_instances.h (compiled next to the other sets).
#include <hip/hip_runtime.h>

#include "lbm_fused_instances.h"

#pragma clang fp contract(off)

LBM_FUSED_SET_5(LBM_FUSED_DEFINE)
