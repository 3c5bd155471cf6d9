// VALU issue-rate probe for gfx950: cycles per wave-instruction of v_fma_f32 / v_pk_fma_f32 / v_mov_b32 /
// v_cndmask_b32 / v_mov_b32 dpp at 1, 2, 3, 4 wavefronts per SIMD (independent instructions, 16 accumulators).
// hipcc --offload-arch=gfx950 -O3 tests/microbench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, long long* cyc) {
  float a[16];
  f2 p[16];
  for (int i = 0; i < 16; ++i) {
    a[i] = threadIdx.x * 0.001f + i;
    p[i] = f2{a[i], a[i] + 1.f};
  }
  const float m = out[0], c = out[1];
  const f2 m2 = {m, m}, c2 = {c, c};
  unsigned long long msk = __builtin_amdgcn_ballot_w64(threadIdx.x & 1);
  asm volatile("" : "+s"(msk));
  float sm = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
  asm volatile("" : "+s"(sm));
  int sacc = 0;
  f2 smm = {sm, sm};
  asm volatile("" : "+s"(smm));
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define PK(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
#define DPP(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 8) & 15]));
#define PKS(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "s"(m2), "v"(c2));
#define CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(msk));
#define CNDZ(i) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(a[i]) : "s"(msk));
#define CNDD(i) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(m), "s"(msk));
#define MULM(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
#define EXPF(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define RDL(i) { int t_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t_) : "v"(a[i])); sacc ^= t_; }
#define FMAS(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sm), "v"(c));
#define PKMOV(i) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(p[i]) : "v"(p[(i + 3) & 15]), "v"(p[(i + 7) & 15]));
#define CMPCND(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
#define CND64V(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
#define CMPCNDS(i) asm volatile("v_cmp_lt_f32_e64 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(msk));
#define PKSG(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "s"(smm), "v"(c2));
#define FMACS(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sm), "v"(c));
#define ADDS(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sm));
#define MAXV(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define FMAK(i) asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(a[i]) : "v"(c));
#define CNDVSET(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
#define CMP1(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
#define MIX(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %4, %5" : "+v"(p[i]), "+v"(a[i]) : "v"(m2), "v"(c2), "v"(m), "v"(c));
    if (KIND == 0) { REP16(FMA) REP16(FMA) REP16(FMA) REP16(FMA) }
    if (KIND == 1) { REP16(PK) REP16(PK) REP16(PK) REP16(PK) }
    if (KIND == 2) { REP16(MOV) REP16(MOV) REP16(MOV) REP16(MOV) }
    if (KIND == 3) { REP16(CND) REP16(CND) REP16(CND) REP16(CND) }
    if (KIND == 4) { REP16(DPP) REP16(DPP) REP16(DPP) REP16(DPP) }
    if (KIND == 5) { REP16(MIX) REP16(MIX) }
    if (KIND == 6) { REP16(CNDS) REP16(CNDS) REP16(CNDS) REP16(CNDS) }
    if (KIND == 7) { REP16(CNDZ) REP16(CNDZ) REP16(CNDZ) REP16(CNDZ) }
    if (KIND == 8) { REP16(CNDD) REP16(CNDD) REP16(CNDD) REP16(CNDD) }
    if (KIND == 9) { REP16(MULM) REP16(MULM) REP16(MULM) REP16(MULM) }
    if (KIND == 10) { REP16(MED3) REP16(MED3) REP16(MED3) REP16(MED3) }
    if (KIND == 11) { REP16(EXPF) REP16(EXPF) REP16(EXPF) REP16(EXPF) }
    if (KIND == 12) { REP16(RDL) REP16(RDL) REP16(RDL) REP16(RDL) }
    if (KIND == 13) { REP16(FMAS) REP16(FMAS) REP16(FMAS) REP16(FMAS) }
    if (KIND == 14) { REP16(PKMOV) REP16(PKMOV) REP16(PKMOV) REP16(PKMOV) }
    if (KIND == 15) { REP16(CMPCND) REP16(CMPCND) }
    if (KIND == 16) { REP16(CND64V) REP16(CND64V) REP16(CND64V) REP16(CND64V) }
    if (KIND == 17) { REP16(CMPCNDS) REP16(CMPCNDS) }
    if (KIND == 18) { REP16(PKSG) REP16(PKSG) REP16(PKSG) REP16(PKSG) }
    if (KIND == 19) { REP16(FMACS) REP16(FMACS) REP16(FMACS) REP16(FMACS) }
    if (KIND == 20) { REP16(ADDS) REP16(ADDS) REP16(ADDS) REP16(ADDS) }
    if (KIND == 21) { REP16(MAXV) REP16(MAXV) REP16(MAXV) REP16(MAXV) }
    if (KIND == 22) { REP16(FMAK) REP16(FMAK) REP16(FMAK) REP16(FMAK) }
    if (KIND == 24) { CMP1(0) CNDVSET(1) CNDVSET(2) CNDVSET(3) CNDVSET(4) CNDVSET(5) CNDVSET(6) CNDVSET(7) CMP1(8) CNDVSET(9) CNDVSET(10) CNDVSET(11) CNDVSET(12) CNDVSET(13) CNDVSET(14) CNDVSET(15)
                      CMP1(0) CNDVSET(1) CNDVSET(2) CNDVSET(3) CNDVSET(4) CNDVSET(5) CNDVSET(6) CNDVSET(7) CMP1(8) CNDVSET(9) CNDVSET(10) CNDVSET(11) CNDVSET(12) CNDVSET(13) CNDVSET(14) CNDVSET(15)
                      CMP1(0) CNDVSET(1) CNDVSET(2) CNDVSET(3) CNDVSET(4) CNDVSET(5) CNDVSET(6) CNDVSET(7) CMP1(8) CNDVSET(9) CNDVSET(10) CNDVSET(11) CNDVSET(12) CNDVSET(13) CNDVSET(14) CNDVSET(15)
                      CMP1(0) CNDVSET(1) CNDVSET(2) CNDVSET(3) CNDVSET(4) CNDVSET(5) CNDVSET(6) CNDVSET(7) CMP1(8) CNDVSET(9) CNDVSET(10) CNDVSET(11) CNDVSET(12) CNDVSET(13) CNDVSET(14) CNDVSET(15) }
    if (KIND == 25) { if (it == 0) { CMP1(0) } REP16(CNDVSET) REP16(CNDVSET) REP16(CNDVSET) REP16(CNDVSET) }
    if (KIND == 26) { asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"); REP16(CND64V) REP16(CND64V) REP16(CND64V) REP16(CND64V) }
    if (KIND == 23) { asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"); REP16(CNDVSET) REP16(CNDVSET) REP16(CNDVSET) REP16(CNDVSET) }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1];
  s += (float)sacc;
  out[2 + blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 64 << 20);
  hipMalloc(&cyc, 64);
  float h[2] = {1.0001f, 0.5f};
  hipMemcpy(out, h, 8, hipMemcpyHostToDevice);
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_mov_b32", "v_cndmask_b32 vcc", "v_mov_b32_dpp", "pk_fma+fma pairs",
                         "v_cndmask e64 sgpr", "v_cndmask e64 0,v,s", "v_cndmask e64 3-addr", "v_mul_f32", "v_med3_f32", "v_exp_f32",
                         "v_readlane_b32", "v_fma_f32 sgpr src", "v_pk_mov_b32", "v_cmp vcc + cndmask vcc (2)", "v_cndmask e64 vcc",
                         "v_cmp e64 s + cndmask e64 (2)", "v_pk_fma_f32 sgpr src", "v_fmac_f32 e32 sgpr", "v_add_f32 sgpr", "v_max_f32",
                         "v_fma_f32 inline const", "s_mov vcc; cndmask vcc", "1 v_cmp + 7 cndmask vcc", "v_cmp once; cndmask vcc", "s_mov vcc; cndmask e64 vcc"};
  const int iters = 5000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int kind = 0; kind < 27; ++kind)
    for (int wps = 1; wps <= 4; wps += (kind >= 15 ? 1 : 1)) {
      // 256 CUs x wps workgroups of 256 threads (4 wavefronts = one per SIMD)
      const int grid = 256 * wps;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        switch (kind) {
          case 0: probe<0><<<grid, 256>>>(out, iters, cyc); break;
          case 1: probe<1><<<grid, 256>>>(out, iters, cyc); break;
          case 2: probe<2><<<grid, 256>>>(out, iters, cyc); break;
          case 3: probe<3><<<grid, 256>>>(out, iters, cyc); break;
          case 4: probe<4><<<grid, 256>>>(out, iters, cyc); break;
          case 5: probe<5><<<grid, 256>>>(out, iters, cyc); break;
          case 6: probe<6><<<grid, 256>>>(out, iters, cyc); break;
          case 7: probe<7><<<grid, 256>>>(out, iters, cyc); break;
          case 8: probe<8><<<grid, 256>>>(out, iters, cyc); break;
          case 9: probe<9><<<grid, 256>>>(out, iters, cyc); break;
          case 10: probe<10><<<grid, 256>>>(out, iters, cyc); break;
          case 11: probe<11><<<grid, 256>>>(out, iters, cyc); break;
          case 12: probe<12><<<grid, 256>>>(out, iters, cyc); break;
          case 13: probe<13><<<grid, 256>>>(out, iters, cyc); break;
          case 14: probe<14><<<grid, 256>>>(out, iters, cyc); break;
          case 15: probe<15><<<grid, 256>>>(out, iters, cyc); break;
          case 16: probe<16><<<grid, 256>>>(out, iters, cyc); break;
          case 17: probe<17><<<grid, 256>>>(out, iters, cyc); break;
          case 18: probe<18><<<grid, 256>>>(out, iters, cyc); break;
          case 19: probe<19><<<grid, 256>>>(out, iters, cyc); break;
          case 20: probe<20><<<grid, 256>>>(out, iters, cyc); break;
          case 21: probe<21><<<grid, 256>>>(out, iters, cyc); break;
          case 22: probe<22><<<grid, 256>>>(out, iters, cyc); break;
          case 23: probe<23><<<grid, 256>>>(out, iters, cyc); break;
          case 24: probe<24><<<grid, 256>>>(out, iters, cyc); break;
          case 25: probe<25><<<grid, 256>>>(out, iters, cyc); break;
          default: probe<26><<<grid, 256>>>(out, iters, cyc); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      long long c;
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      const double insts = 64.0 * iters;  // per wavefront
      printf("%-22s waves/SIMD %d: %8.3f ms  %6.2f ns per instruction and SIMD\n", names[kind], wps, ms, ms * 1e6 / (insts * wps));
    }
  return 0;
}
