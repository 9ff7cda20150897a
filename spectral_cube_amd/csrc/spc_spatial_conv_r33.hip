// ring size 33 instantiation of the separable spatial stencil
#include "spc_spatial_conv_impl.h"
namespace spc_spconv { template int launch_sep<33>(const SpArgs&, hipStream_t, dim3, bool); }
