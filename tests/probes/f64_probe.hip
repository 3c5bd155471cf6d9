// Issue cost of the float64 / conversion / transcendental instructions the static chain is made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 1024
template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, unsigned long long* clk, double w) {
  double a[16];
  float f[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001 + i + 1; f[i] = (float)a[i]; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(w));
      if (MODE == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(w));
      if (MODE == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(w));
      if (MODE == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
      if (MODE == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
      if (MODE == 5) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(w));
      if (MODE == 6) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a[i]));
      if (MODE == 7) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
      if (MODE == 8) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
      if (MODE == 9) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 15]));
      if (MODE == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(f[(i + 1) & 15]));
      if (MODE == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(w));
      if (MODE == 12) asm volatile("v_med3_f32 %0, %0, 0, 1.0" : "+v"(f[i]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += (float)a[i] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
  const int grid = 256 * 8;
  float* o; unsigned long long* c;
  (void)hipMalloc(&o, grid * 256 * 4); (void)hipMalloc(&c, grid * 16);
  unsigned long long* hc = new unsigned long long[2 * grid];
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_max_f64", "v_ldexp_f64",
                         "v_log_f32", "v_exp_f32", "v_fma_f32", "v_cndmask_b32", "v_pk_fma_f32", "v_med3_f32"};
  for (int m = 0; m < 13; ++m) {
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      (void)hipEventRecord(e0);
      switch (m) {
#define C(M) case M: probe<M><<<grid, 256>>>(o, c, 0.999); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12)
      }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    (void)hipMemcpy(hc, c, grid * 16, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0; for (int i = 0; i < grid; ++i) { cyc += hc[2 * i]; rt += hc[2 * i + 1]; }
    cyc /= grid; rt /= grid;
    const double winst = (double)grid * 4 / 1024 * ITERS * 16;
    printf("%-16s %8.1f us  clock %.2f GHz  cycles per wave-instr per SIMD %.2f\n", names[m], best * 1e3, cyc / rt * 0.1,
           best * 1e-3 * (cyc / rt * 1e8) / winst);
  }
  return 0;
}
