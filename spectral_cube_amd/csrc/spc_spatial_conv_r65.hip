// ring size 65 instantiation of the separable spatial stencil (the mask-array forms of the general kernel: spc_spatial_conv_r65m.hip, the all-valid pass: r65f.hip)
#define SPC_SPLIT_MASKED 65
#include "spc_spatial_conv_impl.h"
namespace spc_spconv { template int launch_sep<65>(const SpArgs&, hipStream_t, dim3, bool); }
