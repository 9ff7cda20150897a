// ring size 65: the general separable spatial stencil with a mask ARRAY, in a translation unit of its own (build time)
#include "spc_spatial_conv_impl.h"
namespace spc_spconv { template int launch_sep_general<65, true>(const SpArgs&, hipStream_t, dim3, bool, bool); }
