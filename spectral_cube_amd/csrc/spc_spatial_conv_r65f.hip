// ring size 65: the all-valid pass of the separable spatial stencil, in a translation unit of its own (build time)
#include "spc_spatial_conv_impl.h"
namespace spc_spconv { template int launch_sep_fast<65>(const SpArgs&, hipStream_t, bool); }
