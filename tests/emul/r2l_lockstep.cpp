// r2l_lockstep.cpp -- LOCK-STEP HOST EMULATION of the HIP kernels (one fiber per lane, cooperatively scheduled on one host thread), for the CPU-only test suite under
// -fsanitize=address,undefined.  TEST INFRASTRUCTURE ONLY: see r2l_lockstep_rt.h.  Compiles the same workgroup programs as the
// device library (raw2logit_amd/csrc/*.h) in their DEVICE forms -- including the row-streaming forward, the passes over planes
// and the branch-free static loops, which the serial emulation (r2l_emul.cpp) cannot run -- behind the same C ABI.
#define R2L_EMUL 1
#define R2L_LOCKSTEP 1
#include "../../raw2logit_amd/csrc/r2l_api_impl.h"
