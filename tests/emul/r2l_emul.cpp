// r2l_emul.cpp -- HOST EMULATION of the HIP kernels, for the CPU-only test suite.
//
// TEST INFRASTRUCTURE ONLY.  Compiles the very same workgroup programs (raw2logit_amd/csrc/*.h) with
// g++ and runs their phases for tid = 0..255 in a loop over plain host memory, behind the same C ABI
// (include/r2l_isp.h).  It exists so that `pytest -m "not gpu"` can check the tiling / halo / border /
// reduction logic of the real kernel source against the oracle without a GPU.  It is never loaded by
// the product: raw2logit_amd/_lib.py refuses any library whose r2l_is_device_build() is 0.
#define R2L_EMUL 1
#include "../../raw2logit_amd/csrc/r2l_api_impl.h"
