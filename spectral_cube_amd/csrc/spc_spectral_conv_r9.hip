// ring size 9 instantiation of the spectral stencil (see spc_spectral_conv_impl.h)
#include "spc_spectral_conv_impl.h"
namespace spc_sconv { template int launch<9>(const ConvArgs&, hipStream_t, int, bool); }
