// ring size 33 instantiation of the spectral stencil (see spc_spectral_conv_impl.h)
#include "spc_spectral_conv_impl.h"
namespace spc_sconv { template int launch<33>(const ConvArgs&, hipStream_t, int, bool); }
