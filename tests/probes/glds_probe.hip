// glds_probe.hip -- does `global_load_lds_dwordx4` (M0 = LDS base) reach LDS addresses above 64 KiB on gfx950, as the asm
// statement of r2l_glds16_nt and as the compiler builtin?   hipcc -O2 --offload-arch=gfx950 glds_probe.hip -o glds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../raw2logit_amd/csrc/r2l_common.h"

template <bool BUILTIN>
__global__ __launch_bounds__(256) void probe(const float* src, float* out, int lds_float_off) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 40000; i += 256) lds[i] = -1.f;
  __syncthreads();
  float* slot = lds + lds_float_off + tid * 4;
  if (BUILTIN)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + tid * 4),
                                     (__attribute__((address_space(3))) void*)slot, 16, 0, 0);
  else
    r2l_glds16(src, 16u * tid, slot);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const r2l_f4 v = r2l_lds_f4(slot);
  out[tid * 4 + 0] = v.x;
  out[tid * 4 + 1] = v.y;
  out[tid * 4 + 2] = v.z;
  out[tid * 4 + 3] = v.w;
  // where did it land, if not there?  first 40000 floats
  if (tid == 0) {
    int found = -1;
    for (int i = 0; i < 40000; ++i)
      if (lds[i] == src[0] && i != lds_float_off) { found = i; break; }
    out[1024] = (float)found;
  }
}

int main() {
  const int n = 1024;
  std::vector<float> h(n), o(n + 8);
  for (int i = 0; i < n; ++i) h[i] = 1000.f + i;
  float *d, *r;
  hipMalloc(&d, n * 4);
  hipMalloc(&r, (n + 8) * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
  hipFuncSetAttribute((const void*)probe<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
  const int offs[] = {0, 4096, 15000, 16380, 16384, 17408, 24000, 30000, 36000};
  for (int builtin = 0; builtin < 2; ++builtin)
    for (int off : offs) {
      hipMemset(r, 0, (n + 8) * 4);
      if (builtin)
        hipLaunchKernelGGL(probe<true>, dim3(1), dim3(256), 160000, 0, d, r, off);
      else
        hipLaunchKernelGGL(probe<false>, dim3(1), dim3(256), 160000, 0, d, r, off);
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(o.data(), r, (n + 8) * 4, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < n; ++i) bad += o[i] != h[i];
      printf("%s lds byte offset %6d: %s (%d of %d wrong, first %g, stray copy at float %d) %s\n", builtin ? "builtin" : "asm    ",
             off * 4, bad ? "WRONG" : "ok", bad, n, o[0], (int)o[1024], hipGetErrorString(e));
    }
  return 0;
}
