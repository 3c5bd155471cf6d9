// r2l_api.hip -- libr2l_isp.so: the gfx950 build of the C ABI in include/r2l_isp.h.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC r2l_api.hip -o libr2l_isp.so -ldl
#include <rocfft/rocfft.h>  // TYPES only: fft_denoising's two transforms are resolved with dlopen at first use (r2l_rocfft), not linked
#include "r2l_api_impl.h"
