// r2l_api.hip -- libr2l_isp.so: the gfx950 build of the C ABI in include/r2l_isp.h.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC r2l_api.hip -o libr2l_isp.so -lrocfft
#include <rocfft/rocfft.h>  // fft_denoising's two transforms (the one stage of the path that is a library call)
#include "r2l_api_impl.h"
