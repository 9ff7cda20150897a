// ring size 29 instantiation of the separable spatial stencil
#include "spc_spatial_conv_impl.h"
namespace spc_spconv { template int launch_sep<29>(const SpArgs&, hipStream_t, dim3, bool); }
