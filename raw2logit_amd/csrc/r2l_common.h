// r2l_common.h -- shared definitions of the raw2logit ISP kernels (gfx950 / CDNA4).
//
// Every kernel body is written as a sequence of PHASES: functions of (tid, LDS, per-thread
// registers) separated by workgroup barriers.  libr2l_isp.so compiles them with hipcc for gfx950.
// The same phase functions also compile as plain C++ when R2L_EMUL is defined: tests/_build/
// libr2l_emul.so then runs every phase for tid = 0..511 in a loop, which lets the CPU-only test
// suite check tiling / halo / indexing logic of the REAL kernel source against the oracle.  The
// emulation is test infrastructure: the product loader (raw2logit_amd/_lib.py) refuses it.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/r2l_isp.h"

#define R2L_NT 512  // threads per workgroup: 8 wavefronts of 64 (two per SIMD)

// Host builds (test infrastructure, tests/emul/):
//   R2L_EMUL                  the SERIAL emulation (tests/emul/r2l_emul.cpp): every phase as a loop over tid, no operation
//                             between lanes -- it compiles the tile forms of the kernels only (R2L_SERIAL below);
//   R2L_EMUL + R2L_LOCKSTEP   the LOCK-STEP emulation (tests/emul/r2l_lockstep.cpp): one FIBER per lane, cooperatively
//                             scheduled on a single host thread (one lane runs at a time), the workgroups of a launch one after
//                             the other; wave shifts / readfirstlane / shuffles / s_barrier are rendezvous between the fibers of a
//                             wavefront resp. workgroup (tests/emul/r2l_lockstep_rt.h).  It checks addresses and undefined
//                             behaviour; races between lanes are NOT modelled (nothing runs concurrently).
//                             It compiles EVERY kernel -- the row-streaming forward, the passes over planes, the branch-free
//                             static loops -- in its device form, for -fsanitize=address,undefined runs of the CPU suite.
// Both take the host forms of the leaf helpers (packed pairs, LDS reads, coherent loads: `#ifdef R2L_EMUL`).
#if defined(R2L_EMUL) && !defined(R2L_LOCKSTEP)
#define R2L_SERIAL 1
#endif
#ifdef R2L_LOCKSTEP
#include "../../tests/emul/r2l_lockstep_rt.h"
#endif
// a lane whose EXEC bit goes off for the rest of the kernel while its wavefront goes on exchanging values between lanes (the
// lock-step emulation has to be told; nothing on the device)
#ifdef R2L_LOCKSTEP
#define R2L_LANE_RETIRES() r2l_ls::retire_lane()
#else
#define R2L_LANE_RETIRES()
#endif
// DIAGNOSTIC BUILDS, TIMING ONLY (-DR2L_EXP_NO_HALO; results are wrong): the band passes fetch their halo rows from inside the band
// (the row clamped to [y0, y1 - 1]: a line the wavefront itself touched a few steps ago), i.e. every plane is fetched exactly once --
// the upper bound of what bands that share their halo rows in time (odd bands walking bottom-up) could gain
#ifdef R2L_EXP_NO_HALO
#define R2L_NH(r) ((r) < y0 ? y0 : ((r) > y1 - 1 ? y1 - 1 : (r)))
#else
#define R2L_NH(r) (r)
#endif
// workgroup barrier that orders LDS traffic only (see R2L_PHASE_END)
#if defined(R2L_LOCKSTEP)
#define R2L_LDS_BARRIER() r2l_ls::wg_barrier()
#elif defined(R2L_EMUL)
#define R2L_LDS_BARRIER()
#else
#define R2L_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

#ifdef R2L_EMUL
#define R2L_HD static inline
#define R2L_HOSTDEV static inline
#define R2L_MEMBER inline
#define R2L_BLOCKFN static inline
struct alignas(16) r2l_f4 {
  float x, y, z, w;
};
struct alignas(8) r2l_f2 {
  float x, y;
};
R2L_HD float r2l_log2(float x) { return log2f(x); }
R2L_HD float r2l_exp2(float x) { return exp2f(x); }
R2L_HD float r2l_rcp(float x) { return 1.0f / x; }
R2L_HD r2l_f4 r2l_stream_load_f4(const float* p) { return *(const r2l_f4*)p; }
R2L_HD void r2l_stream_store_f4(float* p, const r2l_f4& v) { *(r2l_f4*)p = v; }
R2L_HD r2l_f4 r2l_load_f4_nt(const float* p) { return *(const r2l_f4*)p; }
R2L_HD void r2l_store_f4_nt(float* p, const r2l_f4& v) { *(r2l_f4*)p = v; }
#ifdef R2L_SERIAL
// the serial emulation runs one lane at a time: kernels take their every-lane-loads form there
#define R2L_HAVE_LANE_SHIFTS false
#define R2L_LANE_ID 0
R2L_HD float r2l_wave_shr1(float x) { return x; }
R2L_HD float r2l_wave_shl1(float x) { return x; }
R2L_HD float r2l_row_shr1(float x, float edge) { return edge; }
R2L_HD float r2l_row_shl1(float x, float edge) { return edge; }
#define R2L_PHASE_BEGIN for (int tid = 0; tid < R2L_NT; ++tid) {
#define R2L_PHASE_BEGIN_L R2L_PHASE_BEGIN
#define R2L_PHASE_BEGIN_IF(L) R2L_PHASE_BEGIN
#define R2L_PHASE_BEGIN_N(NT) for (int tid = 0; tid < (NT); ++tid) {
#define R2L_PHASE_END }
#define R2L_TREG_DECL(type, name) type name##_all[R2L_NT]
#define R2L_TREG(name) name##_all[tid]
#else
// the lock-step emulation: one fiber per lane (a single host thread, one lane at a time) -- the device's forms, the lane-to-lane moves as rendezvous
#define R2L_HAVE_LANE_SHIFTS true
#define R2L_LANE_ID ((int)(threadIdx.x & 63))
R2L_HD float r2l_wave_shr1(float x) { return r2l_ls::dpp(x, x, 0x138); }
R2L_HD float r2l_wave_shl1(float x) { return r2l_ls::dpp(x, x, 0x130); }
R2L_HD float r2l_row_shr1(float x, float edge) { return r2l_ls::dpp(edge, x, 0x111); }
R2L_HD float r2l_row_shl1(float x, float edge) { return r2l_ls::dpp(edge, x, 0x101); }
#define R2L_PHASE_BEGIN \
  {                     \
    const int tid = threadIdx.x;
#define R2L_PHASE_BEGIN_L R2L_PHASE_BEGIN
#define R2L_PHASE_BEGIN_IF(L) R2L_PHASE_BEGIN
#define R2L_PHASE_BEGIN_N(NT) R2L_PHASE_BEGIN
#define R2L_PHASE_END \
  }                   \
  R2L_LDS_BARRIER();
#define R2L_TREG_DECL(type, name) type name
#define R2L_TREG(name) name
#endif
#define R2L_PRAGMA_UNROLL
#define R2L_PRAGMA_NOUNROLL
#define R2L_SCHED_FENCE()
#define R2L_PRIO(n)
#else
#include <hip/hip_runtime.h>
#define R2L_HD static __device__ __forceinline__
#define R2L_HOSTDEV static __host__ __device__ __forceinline__
#define R2L_MEMBER __device__ __forceinline__
#define R2L_BLOCKFN static __device__ __forceinline__
typedef float4 r2l_f4;
typedef float2 r2l_f2;
// v_log_f32 / v_exp_f32 are base-2 and 1 ULP; inputs here are >= 1e-5 (or exactly 0) so the
// denormal pre-scaling of logf()/expf() is not needed.  v_rcp_f32 is 1 ULP.
#ifdef R2L_EXP_NO_TRANS  // DIAGNOSTIC BUILDS, TIMING ONLY (results are wrong): one multiply instead of each transcendental
R2L_HD float r2l_log2(float x) { return x * 1.0001f; }
R2L_HD float r2l_exp2(float x) { return x * 0.9999f; }
R2L_HD float r2l_rcp(float x) { return x * 1.0002f; }
#else
R2L_HD float r2l_log2(float x) { return __builtin_amdgcn_logf(x); }
R2L_HD float r2l_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
R2L_HD float r2l_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif
// value of the previous / next lane of the 64-lane wavefront (DPP wave shifts; lane 0 / 63 keep their own)
#ifdef R2L_HAVE_LANE_SHIFTS_OFF  // A/B builds: every lane loads its neighbour columns
#define R2L_HAVE_LANE_SHIFTS false
#else
#define R2L_HAVE_LANE_SHIFTS true
#endif
#define R2L_LANE_ID ((int)(threadIdx.x & 63))
R2L_HD float r2l_wave_shr1(float x) {
  const int i = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x138, 0xf, 0xf, false));
}
R2L_HD float r2l_wave_shl1(float x) {
  const int i = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x130, 0xf, 0xf, false));
}
// value of x in the previous / next lane of the 16-lane DPP row (row_shr:1 / row_shl:1); lane 0 resp. lane 15 of
// the row, which has no such neighbour, gets its own `edge`
R2L_HD float r2l_row_shr1(float x, float edge) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x),
                                                               0x111, 0xf, 0xf, false));
}
R2L_HD float r2l_row_shl1(float x, float edge) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x),
                                                               0x101, 0xf, 0xf, false));
}
// Touched-once streams (a frame read once, an output written once) can take the nontemporal policy.  A pure
// 4 B : 12 B copy of the static chain's shape gains 5 % from it (tests/probes/stream_probe.hip: 5.2 -> 5.5-5.7
// TB/s).  Round 1's tile kernels did not (profiles/r01_e_static_ab.txt); the row-streaming kernels do, for their
// STORES (round 5, same buffers, interleaved: bilinear 767 -> 758 us, Malvar2004 830 -> 810, default luma chain 941 -> 927,
// Malvar2004 + median 1084 -> 1050-1073; profiles/r05_nt_stores.txt), and lose 9-22 % with nontemporal LOADS (the halo rows a
// neighbouring band re-reads must stay cached): stores nontemporal by default (-DR2L_NT_STORES=0: plain), loads plain
// (-DR2L_NT_LOADS: the A/B form).
typedef float r2l_v4 __attribute__((ext_vector_type(4)));
R2L_HD r2l_f4 r2l_stream_load_f4(const float* p) {
#ifndef R2L_NT_LOADS
  return *(const r2l_f4*)p;
#else
  const r2l_v4 v = __builtin_nontemporal_load((const r2l_v4*)p);
  r2l_f4 o;
  o.x = v.x;
  o.y = v.y;
  o.z = v.z;
  o.w = v.w;
  return o;
#endif
}
// read-once operand of a pure reduction (nontemporal: 6.0 -> 7.0 TB/s on a read-only stream, stream_probe)
R2L_HD r2l_f4 r2l_load_f4_nt(const float* p) {
  const r2l_v4 v = __builtin_nontemporal_load((const r2l_v4*)p);
  r2l_f4 o;
  o.x = v.x;
  o.y = v.y;
  o.z = v.z;
  o.w = v.w;
  return o;
}
// a 16-byte store around the caches (nontemporal): for planes written once and read much later, or not by this step at all
R2L_HD void r2l_store_f4_nt(float* p, const r2l_f4& s) {
  r2l_v4 v;
  v.x = s.x;
  v.y = s.y;
  v.z = s.z;
  v.w = s.w;
  __builtin_nontemporal_store(v, (r2l_v4*)p);
}
#ifndef R2L_NT_STORES
#define R2L_NT_STORES 1
#endif
R2L_HD void r2l_stream_store_f4(float* p, const r2l_f4& s) {
#if !R2L_NT_STORES
  *(r2l_f4*)p = s;
#else
  r2l_v4 v;
  v.x = s.x;
  v.y = s.y;
  v.z = s.z;
  v.w = s.w;
  __builtin_nontemporal_store(v, (r2l_v4*)p);
#endif
}
#define R2L_PHASE_BEGIN \
  {                     \
    const int tid = threadIdx.x;
// R2L_PHASE_BEGIN_L: tid is laundered, so everything the phase derives from it (item maps, LDS addresses, edge
// flags) is recomputed inside the phase -- a few integer instructions -- instead of being hoisted out of the tile
// loop and kept live across all other phases.  Worth ~40-80 VGPRs in the tile kernels (fwd 128 -> 99, bwd1 250 ->
// 211, bwd2 240 -> 163) but ~5 % of their time (profiles/r02_b_lds_ab.txt): used where the registers buy a second
// workgroup per CU (bwd2), not elsewhere.
R2L_HD int r2l_phase_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
#define R2L_PHASE_BEGIN_L \
  {                       \
    const int tid = r2l_phase_tid();
#define R2L_PHASE_BEGIN_IF(L) \
  {                           \
    const int tid = (L) ? r2l_phase_tid() : (int)threadIdx.x;
#define R2L_PHASE_BEGIN_N(NT) R2L_PHASE_BEGIN
// A phase boundary only orders LDS traffic between the waves of the workgroup.  __syncthreads() would
// also wait for every outstanding global store (s_waitcnt vmcnt(0)), i.e. expose the full HBM write
// latency of the previous tile's output at every barrier; the raw barrier below waits for LDS only.
// (Global loads feeding a ds_write are waited for by the compiler at the ds_write itself.)
#define R2L_PHASE_END \
  }                   \
  R2L_LDS_BARRIER();
#define R2L_TREG_DECL(type, name) type name
#define R2L_TREG(name) name
#define R2L_PRAGMA_UNROLL _Pragma("unroll")
#define R2L_PRAGMA_NOUNROLL _Pragma("unroll 1")
// keeps the scheduler from hoisting the scalar loads of LATER weights above this point (their live
// ranges would overflow the SGPR file and be spilled to VGPR lanes)
#define R2L_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// wave issue priority (0..3)
#define R2L_PRIO(n) __builtin_amdgcn_s_setprio(n)
#endif

// 128-bit LDS read that stays one ds_read_b128: without the empty asm hipcc scalarises the vector load
// (it is only used element-wise), then re-merges the pieces with neighbouring scalar reads into
// ds_read2_b32 / ds_read_b96 groups whose lane stride of 16 B is a 4-way bank conflict (measured:
// SQ_LDS_BANK_CONFLICT = 68 % of SQ_LDS_IDX_ACTIVE).
#ifdef R2L_EMUL
R2L_HD r2l_f4 r2l_lds_f4(const float* p) { return *(const r2l_f4*)p; }
R2L_HD float r2l_lds_f1(const float* p) { return *p; }
struct alignas(16) r2l_d2 {
  double x, y;
};
R2L_HD r2l_d2 r2l_lds_d2(const double* p) { return *(const r2l_d2*)p; }
#else
// two float64 as one ds_read_b128
struct alignas(16) r2l_d2 {
  double x, y;
};
typedef double r2l_vd2 __attribute__((ext_vector_type(2)));
R2L_HD r2l_d2 r2l_lds_d2(const double* p) {
  const r2l_vd2 v = *(const volatile __attribute__((address_space(3))) r2l_vd2*)p;
  r2l_d2 o;
  o.x = v.x;
  o.y = v.y;
  return o;
}
// one float, as its own ds_read_b32 (never merged with a neighbour)
R2L_HD float r2l_lds_f1(const float* p) { return *(const volatile __attribute__((address_space(3))) float*)p; }
typedef float r2l_v4 __attribute__((ext_vector_type(4)));
R2L_HD r2l_f4 r2l_lds_f4(const float* p) {
  // volatile: the access may neither be split nor merged with neighbours, but several of them can
  // still be in flight together (an asm pin on the value would serialise load -> wait -> load)
  const r2l_v4 v = *(const volatile __attribute__((address_space(3))) r2l_v4*)p;
  r2l_f4 o;
  o.x = v.x;
  o.y = v.y;
  o.z = v.z;
  o.w = v.w;
  return o;
}
#endif

// Asynchronous global -> LDS copy of 16 bytes per lane (gfx950 `global_load_lds_dwordx4`): no destination registers, the
// data lands in LDS while the wave computes.  The hardware writes lane l of the wavefront at (wave-uniform LDS base) +
// 16 l: a lane-linear staging area, 1 KiB per wave-instruction; the source is (wave-uniform base pointer) + (per-lane byte
// offset), i.e. one scalar register pair and ONE vector register per copy.
// Written as an asm statement on purpose: hipcc keeps no account of it, so no `s_waitcnt vmcnt` is inserted in front of
// the NEXT LDS read (through __builtin_amdgcn_global_load_lds every later ds_read waits for the copy -- its whole memory
// latency, in the middle of the phase that was supposed to hide it).  The consumer therefore waits itself:
// r2l_glds_wait(), then a barrier if other lanes read the data.  (Vector-memory operations complete in issue order, so the
// compiler's own counted waits for ITS loads stay correct: an untracked operation in between only makes them wait for
// more, never for less.)  Default cache policy: a nontemporal copy of a plane another kernel of the step re-writes made
// that kernel slower (profiles/r03_b1_staging_modes.txt).
#ifdef R2L_EMUL
R2L_HD void r2l_glds16(const float* base, unsigned lane_byte_off, float* lane_slot) {
  *(r2l_f4*)lane_slot = *(const r2l_f4*)((const char*)base + lane_byte_off);
}
R2L_HD void r2l_glds_wait() {}
#else
#ifndef R2L_GLDS_POLICY
#define R2L_GLDS_POLICY ""  // " nt": A/B builds
#endif
R2L_HD void r2l_glds16(const float* base, unsigned lane_byte_off, float* lane_slot) {
  // wave-uniform LDS base = the slot of lane 0, whether or not lane 0 takes part (v_readfirstlane reads the first ACTIVE lane)
  const unsigned lbase = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) float*)lane_slot - 16u * (unsigned)R2L_LANE_ID);
  const unsigned long long gb = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)gb), hi = __builtin_amdgcn_readfirstlane((unsigned)(gb >> 32));
  const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" R2L_GLDS_POLICY "\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(lane_byte_off), "s"(sb), "s"(lbase)
               : "memory");
}
R2L_HD void r2l_glds_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

#define R2L_LN2 0.69314718055994530942

// diagnostic builds (-DR2L_EXP_STAMPS): s_memrealtime (100 MHz) at the stations of a launch's tail -- end of the item loop,
// partials stored, tickets, tree levels, unfold -- as seen by thread 0 of whichever workgroup passes a station last
// (= the launch's last workgroup); slots 0-9 statistics pass, 10-19 bn_reduce, 20-29 B2's sums pass.  tests/tail_timeline.py
#if defined(R2L_EXP_STAMPS) && !defined(R2L_EMUL)
__device__ unsigned long long r2l_tail_ts[32];
#define R2L_TAILST(k)                                                                 \
  do {                                                                                \
    if (threadIdx.x == 0) r2l_tail_ts[k] = __builtin_amdgcn_s_memrealtime();          \
  } while (0)
#else
#define R2L_TAILST(k)
#endif
#define R2L_TAIL_BASE(NSLOTS) ((NSLOTS) == 12 ? 0 : ((NSLOTS) == 6 ? 10 : 20))

// Primitives of the in-kernel final reductions.  Partials travel between workgroups (possibly on different
// XCDs, whose L2s are not coherent with each other) through device-coherent accesses: relaxed agent-scope
// atomic stores / loads, which write through / read past the local L2.  A full agent-scope fence would work
// too but writes back and invalidates the whole L2 of the XCD (measured: every kernel 2-7x slower, the frames
// being streamed lose their L2 lines); instead a producer only waits for its own coherent stores to be
// acknowledged (s_waitcnt vmcnt(0)) before it takes its arrival ticket.
// (the emulation runs the workgroups of a launch one after the other, so plain memory operations do)
#ifdef R2L_EMUL
#define R2L_STORES_DONE()
R2L_HD unsigned r2l_ticket(unsigned* c) { return (*c)++; }
R2L_HD void r2l_store_coherent(float* p, float v) { *p = v; }
R2L_HD void r2l_store_coherent(double* p, double v) { *p = v; }
R2L_HD float r2l_load_coherent(const float* p) { return *p; }
R2L_HD double r2l_load_coherent(const double* p) { return *p; }
#else
#define R2L_STORES_DONE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
R2L_HD unsigned r2l_ticket(unsigned* c) {
  return __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
R2L_HD void r2l_store_coherent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
R2L_HD void r2l_store_coherent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
R2L_HD float r2l_load_coherent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
R2L_HD double r2l_load_coherent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

// ---- raw frames: float32 in [0,1], or the sensor's 16-bit containers normalised on the fly ---------------
// The reference divides the loaded container values by 2**bits - 1 in float32 (dataset.py:86-87, :140-143;
// utils/dataset_utils.py:18-26).  With u16 != null the kernels read the containers themselves (2 B/px instead
// of 4) and apply that division when a value enters LDS / the register window: q0 = u * (1/d), one residual
// step e = fma(-q0, d, u), q = fma(e, 1/d, q0) -- equal to the correctly rounded float32 quotient u / d for
// every 16-bit u and d = 65535, 4095, 1023, 255 (checked exhaustively by the tests).
// A third container, float64 frames, exists for the static chains only (a float64 ndarray handed to
// processing(), e.g. a DNG's uint16 / (2**bits - 1), stays float64 through remove_blacklv).
enum { R2L_RAW_F32 = 0, R2L_RAW_U16 = 1, R2L_RAW_F64 = 2 };
struct R2LRaw {
  const float* f32;
  const unsigned short* u16;
  const double* f64;
  float denom, rdenom;
};
R2L_HOSTDEV R2LRaw r2l_raw_f32(const float* p) {
  R2LRaw r;
  r.f32 = p;
  r.u16 = nullptr;
  r.f64 = nullptr;
  r.denom = r.rdenom = 1.f;
  return r;
}
R2L_HOSTDEV R2LRaw r2l_raw_f64(const double* p) {
  R2LRaw r;
  r.f32 = nullptr;
  r.u16 = nullptr;
  r.f64 = p;
  r.denom = r.rdenom = 1.f;
  return r;
}
R2L_HOSTDEV R2LRaw r2l_raw_u16(const unsigned short* p, float denom) {
  R2LRaw r;
  r.f32 = nullptr;
  r.f64 = nullptr;
  r.u16 = p;
  r.denom = denom;
  r.rdenom = 1.0f / denom;
  return r;
}
R2L_HD float r2l_raw_decode(unsigned u, const R2LRaw& r) {
  const float uf = (float)u;
  const float q0 = uf * r.rdenom;
  const float e = fmaf(-q0, r.denom, uf);
  return fmaf(e, r.rdenom, q0);
}
// element i / the aligned 4-element chunk starting at element i of a raw buffer, as float32
R2L_HD float r2l_raw_elem(const R2LRaw& r, size_t i) { return r.u16 ? r2l_raw_decode(r.u16[i], r) : r.f32[i]; }
R2L_HD r2l_f4 r2l_raw_vec4(const R2LRaw& r, size_t i) {
  r2l_f4 v;
  if (r.u16) {
    const r2l_f2 b = *(const r2l_f2*)(r.u16 + i);
    unsigned lo, hi;
    memcpy(&lo, &b.x, 4);
    memcpy(&hi, &b.y, 4);
    v.x = r2l_raw_decode(lo & 0xffffu, r);
    v.y = r2l_raw_decode(lo >> 16, r);
    v.z = r2l_raw_decode(hi & 0xffffu, r);
    v.w = r2l_raw_decode(hi >> 16, r);
  } else {
    v = *(const r2l_f4*)(r.f32 + i);
  }
  return v;
}
R2L_HD unsigned r2l_f2u(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  return u;
}
R2L_HD float r2l_u2f(unsigned u) {
  float x;
  memcpy(&x, &u, 4);
  return x;
}

// ---- packed pairs ---------------------------------------------------------------------------------
// Two horizontally adjacent pixels share one v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: on gfx950 a packed
// f32 instruction issues in the same 4 cycles as a scalar one (measured, tests/probes/valu_probe.hip), and
// the fused kernels are bound by VALU issue, not by HBM.  Element h of a pair is the pixel in column 2p+h
// of the thread's 4-column micro-tile; each half goes through exactly the operations (and the operation
// order) of the unpacked code.
#ifdef R2L_EMUL
typedef float r2l_p2 __attribute__((vector_size(8)));
R2L_HD r2l_p2 r2l_mk2(float a, float b) {
  r2l_p2 r = {a, b};
  return r;
}
R2L_HD r2l_p2 r2l_pfma(r2l_p2 a, r2l_p2 b, r2l_p2 c) { return r2l_mk2(fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])); }
R2L_HD r2l_p2 r2l_pmul(r2l_p2 a, r2l_p2 b) { return r2l_mk2(a[0] * b[0], a[1] * b[1]); }
R2L_HD r2l_p2 r2l_padd(r2l_p2 a, r2l_p2 b) { return r2l_mk2(a[0] + b[0], a[1] + b[1]); }
#else
typedef float r2l_p2 __attribute__((ext_vector_type(2)));
R2L_HD r2l_p2 r2l_mk2(float a, float b) {
  r2l_p2 r = {a, b};
  return r;
}
R2L_HD r2l_p2 r2l_pfma(r2l_p2 a, r2l_p2 b, r2l_p2 c) { return __builtin_elementwise_fma(a, b, c); }
R2L_HD r2l_p2 r2l_pmul(r2l_p2 a, r2l_p2 b) { return a * b; }
R2L_HD r2l_p2 r2l_padd(r2l_p2 a, r2l_p2 b) { return a + b; }
#endif
R2L_HD r2l_p2 r2l_splat2(float a) { return r2l_mk2(a, a); }
// {a[1], b[0]}: the pair that straddles two aligned pairs, as ONE v_pk_mov_b32 (left to itself hipcc emits two
// v_mov_b32 whenever both halves come out of the same 128-bit load)
R2L_HD r2l_p2 r2l_straddle(r2l_p2 a, r2l_p2 b) {
#ifdef R2L_EMUL
  return r2l_mk2(a[1], b[0]);
#else
  r2l_p2 d;
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
  return d;
#endif
}

// torch 'reflect' (mirror without repeating the edge: c b | a b c), clamped for far-out indices
R2L_HD int r2l_mirror(int i, int n) {
  i = i < 0 ? -i : i;
  i = i >= n ? 2 * (n - 1) - i : i;
  i = i < 0 ? 0 : i;
  return i > n - 1 ? n - 1 : i;
}
// scipy 'reflect' (symmetric: b a | a b), clamped
R2L_HD int r2l_symmetric(int i, int n) {
  i = i < 0 ? -1 - i : i;
  i = i >= n ? 2 * n - 1 - i : i;
  i = i < 0 ? 0 : i;
  return i > n - 1 ? n - 1 : i;
}
// channel of the RGGB site with row/column parities (py,px): R=0, G1=G2=1, B=2
// (pipeline_torch.py:256-259, :274-277)
R2L_HD int r2l_site_channel(int py, int px) { return (py & 1) + (px & 1); }

// ---- output epilogue: the weak augmentation as part of the forward's stores ----------------------------------------
// utils/augmentation.py:70-74 applied to the processor's output (model.py:79-81): RandomHorizontalFlip, RandomVerticalFlip,
// RandomRotate90 = rot90^k(vflip(hflip(x))), a permutation of every (H, W) plane.  It is affine in the pixel coordinates:
// pixel (y, x) of the ISP's own layout goes to element  s0 + sr * y + sc * x  of its output plane (k even: sc = +-1, the
// lane's 4 pixels stay one 16-byte vector, reversed for a horizontal flip; k odd: sc = +-H, four scalar accesses).  The
// apply pass writes there directly, the backward reads grad_out from there (bn_reduce pairs grad_out with the saved output
// element by element: both are in the permuted layout), so the separate permutation kernel (24 B/px) and its inverse in the
// backward disappear.  r2l_aug_map (r2l_staged_kernels.h) is the definition the host derives (s0, sr, sc) from.
struct R2LEpi {
  int on, s0, sr, sc;
};

// ---- folded parameters -------------------------------------------------------------------------
// The chain  mosaic -> Debayer conv -> white balance -> CCM -> RGB->YUV  (pipeline_torch.py:183-194)
// is linear in the black-level-corrected raw value v: because each mosaic plane is non-zero only on
// its own Bayer sites, YUV[k](p) = sum_t A[k][parity(p)][t] * v(p+t) over the 3x3 taps t, with
//   A[k][par][t] = sum_j T[k][j] * debayer.weight[j][channel(site(par,t))][t],
//   T = M_RGB_2_YUV * colour_correction * diag(white_balance).
// A tiny prologue kernel computes the block below (in float64) from the packed parameters on the
// device, so parameters never visit the host.
struct R2LFolded {
  float bl[4];  // black level per RGGB site
  float AY[4][9];  // [parity = (y&1)*2 + (x&1)][tap = (dy+1)*3 + (dx+1)]
  float AU[4][9];
  float AV[4][9];
  float sharp[9];
  float blur[25];
  float M2[9];  // M_YUV_2_RGB [k][c]
  float inv_gamma;
  float gamma;
  float pad[2];
  // the parity-indexed stencils again, laid out [row parity][tap][column parity]: the weights of a pixel
  // pair (even column, odd column) sit side by side, i.e. in one aligned SGPR pair after an s_load_dwordx2
  float AY2[2][9][2];
  float AU2[2][9][2];
  float AV2[2][9][2];
  // the 5x5 blur for the first / last two image rows, mirror padding folded into the weights (row-streaming forward:
  // its window holds rows y-2..y+2; a row outside the image gives its weights to its mirror image y' = -y resp.
  // 2(H-1) - y, both inside the window for H >= 4).  [0]: y = 0, [1]: y = 1, [2]: y = H-2, [3]: y = H-1
  float blur_edge[4][25];
};

// The kernels read the folded block through the CONSTANT address space so that every weight is a
// scalar load (s_load -> SGPR operand of v_fmac), never a per-lane VGPR.  The SGPR file holds ~100
// values, fewer than the ~150 weights: the pixel code therefore works one output row at a time and
// launders the pointer (r2l_opaque) per row, so that hipcc re-issues the scalar loads of that row's
// weights instead of hoisting all of them to the top and spilling SGPRs into VGPR lanes.
#ifdef R2L_EMUL
typedef const R2LFolded& R2LFoldedRef;
#define R2L_FOLDED_REF(ptr) (*(ptr))
#define R2L_CONSTAS
#elif defined(R2L_EXP_CONST_WEIGHTS)
// DIAGNOSTIC BUILD, TIMING ONLY (results are wrong): every folded weight is a compile-time constant, so the kernels issue no
// scalar loads for them at all -- the upper bound of what hiding the scalar-load waits of a row step could buy
// (profiles/r05_const_weights.txt)
constexpr R2LFolded r2l_exp_make_folded() {
  R2LFolded f{};
  float v = 0.0131f;
  for (int i = 0; i < 4; ++i) f.bl[i] = (v += 0.0007f);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 9; ++j) {
      f.AY[i][j] = (v += 0.0007f);
      f.AU[i][j] = (v += 0.0007f);
      f.AV[i][j] = (v += 0.0007f);
    }
  for (int j = 0; j < 9; ++j) f.sharp[j] = (v += 0.0007f);
  for (int j = 0; j < 25; ++j) f.blur[j] = (v += 0.0007f);
  for (int j = 0; j < 9; ++j) f.M2[j] = (v += 0.0007f);
  f.inv_gamma = 0.4545f;
  f.gamma = 2.2f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 9; ++j)
      for (int k = 0; k < 2; ++k) {
        f.AY2[i][j][k] = (v += 0.0007f);
        f.AU2[i][j][k] = (v += 0.0007f);
        f.AV2[i][j][k] = (v += 0.0007f);
      }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 25; ++j) f.blur_edge[i][j] = (v += 0.0007f);
  return f;
}
static __device__ constexpr R2LFolded r2l_exp_folded = r2l_exp_make_folded();
typedef const R2LFolded& R2LFoldedRef;
#define R2L_FOLDED_REF(ptr) (((void)(ptr)), r2l_exp_folded)
#define R2L_CONSTAS
#else
typedef const __attribute__((address_space(4))) R2LFolded& R2LFoldedRef;
#define R2L_FOLDED_REF(ptr) (*(const __attribute__((address_space(4))) R2LFolded*)(ptr))
#define R2L_CONSTAS __attribute__((address_space(4)))
#endif
#define R2L_FOLDED_FLOATS 0
R2L_HD const R2LFolded* r2l_opaque(const R2LFolded* p) {
#ifndef R2L_EMUL
  asm volatile("" : "+s"(p));
#endif
  return p;
}

// ... and not before `dep` exists: in a loop body without branches (r2l_fwd_apply_block) nothing else keeps hipcc from
// floating the invariant scalar loads of ALL the unrolled steps to the top of the block (__builtin_amdgcn_sched_barrier
// does not order them: 280 spilled SGPRs, 1 KB of scratch)
R2L_HD const R2LFolded* r2l_opaque_after(const R2LFolded* p, float dep) {
#ifndef R2L_EMUL
  asm volatile("" : "+s"(p) : "v"(dep));
#else
  (void)dep;
#endif
  return p;
}

// Workgroup ids are dealt round-robin over the 8 XCDs (id % 8), each with its own 4 MiB L2.  Kernels whose consecutive
// work items share rows (the band passes: a band re-reads the halo rows of its neighbours) can walk the items through this
// map: inside every window of 8 M consecutive ids, XCD x takes the M CONTIGUOUS virtual ids [x M, (x + 1) M) -- M
// neighbouring bands then run on one XCD at about the same time and their shared halo rows are L2 hits instead of second
// fetches from HBM, while the chip as a whole still works on one compact region (handing each XCD a contiguous EIGHTH of
// the whole launch is slower than no map at all: eight regions in flight, profiles/r04_ab_static_xcd.txt).
// R2L_XCD_REMAP = M (0: off).
#ifndef R2L_XCD_REMAP
#define R2L_XCD_REMAP 0
#endif
R2L_HD int r2l_xcd_contiguous(int bid, int nblk) {
#if R2L_XCD_REMAP
  constexpr int M = R2L_XCD_REMAP, G = 8 * M;
  const int w0 = bid - bid % G;
  return (w0 + G <= nblk) ? w0 + (bid & 7) * M + ((bid - w0) >> 3) : bid;
#else
  (void)nblk;
  return bid;
#endif
}

// The same map with the window size as a launch argument (the band passes of the parametrized step: `m` = neighbouring
// workgroups per XCD, a power of two, 0 = off).  Only the WORK ITEMS follow the mapped id; partial-sum slots and the
// reduction trees keep the hardware's workgroup id.
R2L_HD int r2l_xcd_window(int bid, int nblk, int m) {
  if (m <= 0) return bid;
  const int G = 8 * m, w0 = bid & ~(G - 1);
  return (w0 + G <= nblk) ? w0 + (bid & 7) * m + ((bid - w0) >> 3) : bid;
}

// Issue priority by PROGRESS.  The wavefronts of a SIMD start together (the band passes are sized for one round of resident
// wavefronts) and the hardware favours the oldest one whenever several are ready: left alone they finish one after the
// other, and the last one walks its rows alone with every scalar-load and memory wait exposed (the plane kernels'
// wavefronts end between 25 and 68 us after a common start, profiles/r03_z_timeline_fwd.txt).  A wavefront that is
// further BEHIND in its band gets the higher priority -- quarter of the band done -> s_setprio 3, 2, 1, 0 -- so the
// wavefronts of a SIMD advance together and overlap until the end: 64x512x512 step 0.3940 -> 0.3873 ms, the sums pass of
// the backward 54.9 -> 52.5 us (profiles/r04_progress_prio.txt).  s_setprio takes an immediate: a scalar if-chain.
// R2L_PROGRESS_PRIO = 0: off; 1: per group of 6 rows; 2: per row step.
#ifndef R2L_PROGRESS_PRIO_MODE
#define R2L_PROGRESS_PRIO_MODE 1
#endif
#if R2L_PROGRESS_PRIO_MODE && !defined(R2L_EMUL)
R2L_HD void r2l_progress_prio(int done, int total) {
  const int q4 = done * 4;
  if (q4 < total) __builtin_amdgcn_s_setprio(3);
  else if (q4 < 2 * total) __builtin_amdgcn_s_setprio(2);
  else if (q4 < 3 * total) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}
#define R2L_PROGRESS_PRIO(done, total) r2l_progress_prio((done), (total))
#if R2L_PROGRESS_PRIO_MODE == 2
#define R2L_PROGRESS_PRIO_STEP(done, total) r2l_progress_prio((done), (total))
#else
#define R2L_PROGRESS_PRIO_STEP(done, total)
#endif
#else
#define R2L_PROGRESS_PRIO(done, total)
#define R2L_PROGRESS_PRIO_STEP(done, total)
#endif

// A kernel's argument block, re-read from the kernarg segment at the point of use: arguments that only the end of a kernel
// needs (reduction tree, BatchNorm bookkeeping: 32 scalar registers' worth) otherwise stay live through the whole main
// loop, where scalar registers are what the streaming kernels run out of (SGPR spills into VGPR lanes: 44 -> 39 with this;
// the rest are masks and row bookkeeping of the loop itself).  The kernels here take ONE struct by value, which is the
// start of the segment.
#ifndef R2L_EMUL
template <class T>
R2L_HD const __attribute__((address_space(4))) T* r2l_kernargs() {
  const __attribute__((address_space(4))) T* p =
      (const __attribute__((address_space(4))) T*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
#elif defined(R2L_LOCKSTEP)
template <class T>
R2L_HD const T* r2l_kernargs() {
  return (const T*)r2l_ls::kernarg();
}
#endif

// element (k, c) of T = M_RGB_2_YUV * colour_correction * diag(white_balance), float64
R2L_HD double r2l_fold_T_one(const float* P, int k, int c) {
  double s = 0;
  R2L_PRAGMA_UNROLL
  for (int j = 0; j < 3; ++j) s += (double)P[R2L_P_M_RGB2YUV + k * 3 + j] * (double)P[R2L_P_CCM + j * 3 + c];
  return s * (double)P[R2L_P_WHITE_BALANCE + c];
}
// A[k][par][t] = sum_j T[k][j] * debayer.weight[j][channel(site(par,t))][t], float64
R2L_HD double r2l_fold_A_one(const float* P, int k, int par, int t) {
  const int py = par >> 1, px = par & 1, dy = t / 3 - 1, dx = t % 3 - 1;
  const int c = r2l_site_channel(py + dy + 2, px + dx + 2);
  double s = 0;
  R2L_PRAGMA_UNROLL
  for (int j = 0; j < 3; ++j) s += r2l_fold_T_one(P, k, j) * (double)P[R2L_P_DEBAYER + (j * 3 + c) * 9 + t];
  return s;
}
// element `idx` (float index into R2LFolded) of the folded block; one lane per element
R2L_HD void r2l_fold_one(const float* P, R2LFolded* F, int idx) {
  float* out = (float*)F;
  const int o_ay = 4, o_sharp = 4 + 108, o_blur = o_sharp + 9, o_m2 = o_blur + 25, o_ig = o_m2 + 9;
  const int o_pair = o_ig + 4;
  const int o_edge = o_pair + 108;
  if (idx >= o_edge) {
    const int e = idx - o_edge, set = e / 25, i = (e % 25) / 5, j = e % 5;
    // window row i = image row y + i - 2; rows outside (top sets: i < 2 - y; bottom sets: i > 2 + (H-1-y)) are zero,
    // their mirror images take their weights
    const int yrel = (set == 0) ? 0 : (set == 1 ? 1 : (set == 2 ? -2 : -1));  // y, or y - H for the bottom sets
    float w = 0.f;
    for (int k = 0; k < 5; ++k) {  // source window row k lands on window row t
      const int r = yrel + k - 2;  // row index relative to row 0 (top sets) or to row H (bottom sets)
      int t;
      if (set < 2)
        t = (r < 0) ? (-r) - yrel + 2 : k;
      else
        t = (r >= 0) ? (-2 - r) - yrel + 2 : k;  // H + r -> 2(H-1) - (H + r) = H - 2 - r
      if (t == i) w += P[R2L_P_BLUR + k * 5 + j];
    }
    out[idx] = w;
  } else if (idx < 4) {
    out[idx] = P[R2L_P_BLACK_LEVEL + idx];
  } else if ((idx >= o_ay && idx < o_sharp) || (idx >= o_pair && idx < o_pair + 108)) {
    int k, par, t;
    if (idx < o_sharp) {
      const int e = idx - o_ay;
      k = e / 36;
      par = (e % 36) / 9;
      t = e % 9;
    } else {  // [k][py][t][px]
      const int e = idx - o_pair;
      k = e / 36;
      par = ((e % 36) / 18) * 2 + (e & 1);
      t = (e % 18) / 2;
    }
    // (no lane-private T[3][3]: a dynamically indexed array would live in scratch memory)
    out[idx] = (float)r2l_fold_A_one(P, k, par, t);
  } else if (idx < o_blur) {
    out[idx] = P[R2L_P_SHARPEN + idx - o_sharp];
  } else if (idx < o_m2) {
    out[idx] = P[R2L_P_BLUR + idx - o_blur];
  } else if (idx < o_ig) {
    out[idx] = P[R2L_P_M_YUV2RGB + idx - o_m2];
  } else if (idx == o_ig) {
    out[idx] = (float)(1.0 / (double)P[R2L_P_GAMMA]);
  } else if (idx == o_ig + 1) {
    out[idx] = P[R2L_P_GAMMA];
  } else {
    out[idx] = 0.f;
  }
}
#define R2L_FOLDED_NFLOATS ((int)(sizeof(R2LFolded) / sizeof(float)))

// ---- reduction slots of the backward kernels ----------------------------------------------------
// B1 (pixel kernel): sums over pixels p of the tile interior
enum {
  R2L_B1_GBLUR = 0,   // [25]  sum gY''(p) * Y'_ext(p+t)           -> d/d gaussian_blur.weight
  R2L_B1_GAU = 25,    // [4][9] sum_{par(p)} gU(p) * v_ext(p+t)      -> folded debayer/CCM/WB grads
  R2L_B1_GAV = 61,    // [4][9]
  R2L_B1_SU = 97,     // [4]   sum_{par(p)} gU(p)                   -> black level
  R2L_B1_SV = 101,    // [4]
  R2L_B1_GGAM = 105,  // [1]   sum g * x^(1/gamma) * log2(x_clipped) -> gamma_correct
  R2L_B1_NACC = 106
};
// B2 (luma adjoint kernel)
enum {
  R2L_B2_GSHARP = 0,  // [9]   sum gY'(p) * Y_zero_ext(p+t)         -> d/d sharpening_filter.weight
  R2L_B2_GAY = 9,     // [4][9] sum_{par(p)} gY(p) * v_ext(p+t)
  R2L_B2_SY = 45,     // [4]
  R2L_B2_NACC = 49
};
#define R2L_NSUMS (R2L_B1_NACC + R2L_B2_NACC)

// Unfold the reduced sums into the gradient of trainable parameter `o` (index into the packed block);
// float64 throughout, one lane per parameter.
// gT[k][j] = sum_{par,t} GA[k][par][t] * debayer.weight[j][chan(par,t)][t]  (shared by the white-balance
// and colour-matrix gradients)
R2L_HD double r2l_unfold_gT(const float* P, const double* S, int k, int j) {
  const double* b1 = S;
  const double* b2 = S + R2L_B1_NACC;
  const double* GA = (k == 0) ? (b2 + R2L_B2_GAY) : (k == 1 ? b1 + R2L_B1_GAU : b1 + R2L_B1_GAV);
  double s = 0;
  R2L_PRAGMA_UNROLL
  for (int par = 0; par < 4; ++par)
    R2L_PRAGMA_UNROLL
  for (int t = 0; t < 9; ++t) {  // (unrolled: the channel of a tap is a constant then, the 72 LDS reads go out as a batch)
    const int c = r2l_site_channel((par >> 1) + t / 3 + 1, (par & 1) + t % 3 + 1);
    s += GA[par * 9 + t] * (double)P[R2L_P_DEBAYER + (j * 3 + c) * 9 + t];
  }
  return s;
}
// black_level[site] gets  - sum_{k, par, t : site(par, t) == site} A[k][par][t] * S_k[par]  (S_k[par] = sum of gK over
// the pixels of parity par; the raw value at p + t has the black level of ITS site subtracted): the k-th part of it
R2L_HD double r2l_unfold_bl_part(const double* S, const double* TG, int site, int k) {
  const double* b1 = S;
  const double* b2 = S + R2L_B1_NACC;
  const double* SS = (k == 0) ? (b2 + R2L_B2_SY) : (k == 1 ? b1 + R2L_B1_SU : b1 + R2L_B1_SV);
  // (unrolled, every product formed and the ones of other sites dropped by a select: the LDS reads go out as one batch;
  // as a loop with a `continue` this was a chain of 36 dependent read + fma steps, 2 us of the launch's tail)
  const double* A = TG + 18 + k * 36;
  double g = 0;
  R2L_PRAGMA_UNROLL
  for (int par = 0; par < 4; ++par) {
    const double ss = SS[par];
    R2L_PRAGMA_UNROLL
    for (int t = 0; t < 9; ++t) {
      const int py = par >> 1, px = par & 1, dy = t / 3 - 1, dx = t % 3 - 1;
      const double prod = A[par * 9 + t] * ss;  // folded A[k][par][t], float64
      g -= ((((py + dy + 2) & 1) * 2 + ((px + dx + 2) & 1)) == site) ? prod : 0.0;
    }
  }
  return g;
}
// Unfold the reduced sums into the gradient of trainable parameter `o` (index into the packed block);
// float64 throughout, one lane per parameter.  TG = T[9] (r2l_fold_T_one), gT[9] (r2l_unfold_gT) and the
// folded stencils A[3][4][9] (r2l_fold_A_one), computed once per launch by 126 lanes.  P, S and TG are read in place (they sit in LDS): no lane-private
// arrays, which dynamic indexing would send to scratch memory (measured: 17 us for this 132-lane step).
R2L_HD float r2l_unfold_one(const float* P, const double* S, int o, const double* TG) {
  const double* b1 = S;
  const double* b2 = S + R2L_B1_NACC;
  const double* T = TG;
  const double* gT = TG + 9;
  if (o >= R2L_P_SHARPEN && o < R2L_P_BLUR) return (float)b2[R2L_B2_GSHARP + o - R2L_P_SHARPEN];
  if (o >= R2L_P_BLUR && o < R2L_P_NTRAIN) return (float)b1[R2L_B1_GBLUR + o - R2L_P_BLUR];
  if (o == R2L_P_GAMMA) {
    const double gamma = P[R2L_P_GAMMA];
    return (float)(-b1[R2L_B1_GGAM] * R2L_LN2 / (gamma * gamma));
  }
  if (o >= R2L_P_DEBAYER && o < R2L_P_SHARPEN) {  // debayer.weight[j][c][t]
    const int e = o - R2L_P_DEBAYER, j = e / 27, c = (e % 27) / 9, t = e % 9;
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    double g = 0;
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      const double* GA = (k == 0) ? (b2 + R2L_B2_GAY) : (k == 1 ? b1 + R2L_B1_GAU : b1 + R2L_B1_GAV);
      const double tk = T[k * 3 + j];
      R2L_PRAGMA_UNROLL
      for (int par = 0; par < 4; ++par) {
        const double prod = tk * GA[par * 9 + t];
        g += (r2l_site_channel((par >> 1) + dy + 2, (par & 1) + dx + 2) == c) ? prod : 0.0;
      }
    }
    return (float)g;
  }
  if (o < R2L_P_WHITE_BALANCE)  // black_level[site]: the three per-k partial sums (r2l_unfold_bl_part)
    return (float)(TG[126 + o * 3] + TG[126 + o * 3 + 1] + TG[126 + o * 3 + 2]);
  // white balance / colour matrix: through gT
  if (o < R2L_P_CCM) {
    const int c = o - R2L_P_WHITE_BALANCE;
    double g = 0;
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < 3; ++j)
        g += gT[k * 3 + c] * (double)P[R2L_P_M_RGB2YUV + k * 3 + j] * (double)P[R2L_P_CCM + j * 3 + c];
    return (float)g;
  }
  const int j = (o - R2L_P_CCM) / 3, c = (o - R2L_P_CCM) % 3;
  double g = 0;
  for (int k = 0; k < 3; ++k)
    g += gT[k * 3 + c] * (double)P[R2L_P_M_RGB2YUV + k * 3 + j] * (double)P[R2L_P_WHITE_BALANCE + c];
  return (float)g;
}
