// VALU issue-rate probe: v_fma_f32 vs v_pk_fma_f32 vs v_pk_mov_b32, v_log/v_exp, and the clock the chip holds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 2048
template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, unsigned long long* clk, float w) {
  float a[16];
  v2f p[8];
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int i = 0; i < 8; ++i) p[i] = (v2f){a[2 * i], a[2 * i + 1]};
  const v2f w2 = {w, w * 1.0001f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], w, 0.5f);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], w2, (v2f){0.5f, 0.25f});
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], w2, (v2f){0.5f, 0.25f});
    } else if (MODE == 2) {  // 8 pk_fma + 8 cross-pair shuffles
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], w2, (v2f){0.5f, 0.25f});
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v2f q = {p[i].y, p[(i + 1) & 7].x};
        asm volatile("" : "+v"(q));
        p[i] = q;
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(a[i]) * w);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
  const int grid = 256 * 8;  // 8 WGs of 4 waves per CU = 8 waves per SIMD
  float* o; unsigned long long* c;
  hipMalloc(&o, grid * 256 * 4); hipMalloc(&c, grid * 16);
  unsigned long long* hc = new unsigned long long[2 * grid];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"v_fma_f32 x16", "v_pk_fma_f32 x16", "pk_fma x8 + shuffle x8", "log+mul+exp x16"};
  for (int m = 0; m < 4; ++m) {
    float best = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
      hipEventRecord(e0);
      if (m == 0) probe<0><<<grid, 256>>>(o, c, 0.999f);
      if (m == 1) probe<1><<<grid, 256>>>(o, c, 0.999f);
      if (m == 2) probe<2><<<grid, 256>>>(o, c, 0.999f);
      if (m == 3) probe<3><<<grid, 256>>>(o, c, 0.999f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipMemcpy(hc, c, grid * 16, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0; for (int i = 0; i < grid; ++i) { cyc += hc[2 * i]; rt += hc[2 * i + 1]; }
    cyc /= grid; rt /= grid;
    // wave-instructions per SIMD: grid*4 waves / (256*4 SIMDs) * ITERS * 16
    const double winst = (double)grid * 4 / 1024 * ITERS * 16;
    printf("%-24s %8.1f us  wg cycles %9.0f  clock %.2f GHz  cycles per wave-instr per SIMD %.2f (wall-based @clock)\n", names[m], best * 1e3,
           cyc, cyc / rt * 0.1, best * 1e-3 * (cyc / rt * 1e8) / winst);
  }
  return 0;
}
