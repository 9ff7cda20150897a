// ring size 49 instantiation of the separable spatial stencil
#include "spc_spatial_conv_impl.h"
namespace spc_spconv { template int launch_sep<49>(const SpArgs&, hipStream_t, dim3, bool); }
