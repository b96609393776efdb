// MSM kernels + launch sequence instantiated for mnt4753_g1 (see msm_impl.h, msm_kernels.h).
#include "msm_impl.h"
GH_DEFINE_MSM_OPS(gh::Mnt4G1, msm_ops_mnt4753_g1)
