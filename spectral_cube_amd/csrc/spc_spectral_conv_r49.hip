// 49-tap ring of the spectral stencil: all-valid pass only (see launch_fast_only in spc_spectral_conv_impl.h)
#include "spc_spectral_conv_impl.h"
namespace spc_sconv { template int launch_fast_only<49>(const ConvArgs&, hipStream_t); }
