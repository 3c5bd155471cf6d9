// HBM bandwidth probe: calibrates the ceiling for the ISP's 4 B in : 12 B out access pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void r1w3(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = a[i];
    b[i] = v; v.x += 1.f; b[i + n] = v; v.y += 1.f; b[i + 2 * n] = v;
  }
}
__global__ void w_only(float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = make_float4(1, 2, 3, 4);
}
__global__ void r_only(const float4* __restrict__ a, float* out, size_t n) {
  float s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[0] = s;
}
int main() {
  const size_t n = (size_t)256 * 1024 * 1024 / 4;  // float4 elements of a 1 GiB plane... 268M px / 4
  float4 *a, *b; float* o;
  hipMalloc(&a, n * 16); hipMalloc(&b, 3 * n * 16); hipMalloc(&o, 4);
  hipMemset(a, 0, n * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {2048, 8192, 32768}) {
    for (int k = 0; k < 4; ++k) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (k == 0) copy4<<<grid, 256>>>(a, b, n);
        if (k == 1) r1w3<<<grid, 256>>>(a, b, n);
        if (k == 2) w_only<<<grid, 256>>>(b, 3 * n);
        if (k == 3) r_only<<<grid, 256>>>(b, o, 3 * n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      const double bytes = (k == 0 ? 2.0 : k == 1 ? 4.0 : 3.0) * n * 16;
      printf("grid %6d %-7s %8.1f us  %7.1f GB/s\n", grid, k == 0 ? "copy" : k == 1 ? "r1w3" : k == 2 ? "write" : "read", best * 1e3, bytes / best / 1e6);
    }
  }
  return 0;
}
