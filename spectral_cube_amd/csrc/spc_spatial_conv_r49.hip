// ring size 49 instantiation of the separable spatial stencil (the mask-array forms of the general kernel: spc_spatial_conv_r49m.hip, the all-valid pass: r49f.hip)
#define SPC_SPLIT_MASKED 49
#include "spc_spatial_conv_impl.h"
namespace spc_spconv { template int launch_sep<49>(const SpArgs&, hipStream_t, dim3, bool); }
