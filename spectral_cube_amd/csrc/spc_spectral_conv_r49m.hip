// 49-tap ring of the spectral stencil, general form (masks / NaNs; see spectral_conv_ring_wide_kernel in spc_spectral_conv_impl.h)
#include "spc_spectral_conv_impl.h"
namespace spc_sconv { template int launch_ring_wide<49>(const ConvArgs&, hipStream_t); }
