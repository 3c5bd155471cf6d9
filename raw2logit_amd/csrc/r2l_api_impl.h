// r2l_api_impl.h -- the C ABI of include/r2l_isp.h, written once.
//
// Included by r2l_api.hip (hipcc, gfx950: launches real kernels on a HIP stream) and by
// tests/emul/r2l_emul.cpp (g++, R2L_EMUL: runs the same workgroup programs on host memory, for the
// CPU-only test suite).
#pragma once
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "r2l_simple_kernels.h"
#include "r2l_param_stream.h"
#include "r2l_param_plane_bwd.h"
#include "r2l_static_kernels.h"
#include "r2l_static_stream.h"
#include "r2l_static_chain.h"
#include "r2l_static_planes.h"
#include "r2l_static_menon.h"
#include "r2l_staged_kernels.h"
#include "r2l_aux_kernels.h"

static thread_local std::string r2l_err;
static int r2l_fail(int code, const std::string& msg) {
  r2l_err = msg;
  return code;
}

#ifdef R2L_LOCKSTEP
// the lock-step emulation (tests/emul/r2l_lockstep_rt.h): NT host threads per workgroup, the workgroups one after the other; a
// launch record (name -> count) stands in for the device build's event timing, so that tests can assert which kernels ran
static std::mutex r2l_ls_record_mutex;
static bool r2l_ls_record_on = false;
static std::map<std::string, int> r2l_ls_record;
static void r2l_ls_note(const char* name) {
  std::lock_guard<std::mutex> g(r2l_ls_record_mutex);
  if (r2l_ls_record_on) r2l_ls_record[std::string(name) + "_kernel"] += 1;
}
#define R2L_LS_KERNEL(name, ArgsT, NT_, LDSF, ...)                                                        \
  static int name(const ArgsT& a, int grid, void* stream) {                                               \
    (void)stream;                                                                                         \
    r2l_ls_note(#name);                                                                                   \
    r2l_ls::launch(#name, grid, (NT_), (size_t)(LDSF), &a,                                                \
                   [&](int b_, float* lds_) { __VA_ARGS__(a, b_, grid, lds_); });                         \
    return 0;                                                                                             \
  }
#define R2L_KERNEL_V(name, ArgsT, LDS_FLOATS, W, ...) R2L_LS_KERNEL(name, ArgsT, R2L_NT, LDS_FLOATS, __VA_ARGS__)
#define R2L_KERNEL_OCC(name, ArgsT, blockfn, LDS_FLOATS, W) R2L_LS_KERNEL(name, ArgsT, R2L_NT, LDS_FLOATS, blockfn)
#define R2L_KERNEL(name, ArgsT, blockfn, LDS_FLOATS) R2L_LS_KERNEL(name, ArgsT, R2L_NT, LDS_FLOATS, blockfn)
#define R2L_KERNEL_NT(name, ArgsT, blockfn, NT, W) R2L_LS_KERNEL(name, ArgsT, NT, 0, blockfn)
#define R2L_KERNEL_NT_LDS(name, ArgsT, NT, LDS_FLOATS, W, ...) R2L_LS_KERNEL(name, ArgsT, NT, LDS_FLOATS, __VA_ARGS__)
#elif defined(R2L_EMUL)
#define R2L_KERNEL_V(name, ArgsT, LDS_FLOATS, W, ...)                        \
  static int name(const ArgsT& a, int grid, void* stream) {                  \
    (void)stream;                                                            \
    std::vector<float> buf((size_t)(LDS_FLOATS) + 8);                        \
    float* lds = (float*)(((uintptr_t)buf.data() + 15) & ~(uintptr_t)15);    \
    for (int b = 0; b < grid; ++b) __VA_ARGS__(a, b, grid, lds);             \
    return 0;                                                                \
  }
#define R2L_KERNEL_OCC(name, ArgsT, blockfn, LDS_FLOATS, W) R2L_KERNEL(name, ArgsT, blockfn, LDS_FLOATS)
#define R2L_KERNEL_NT(name, ArgsT, blockfn, NT, W) R2L_KERNEL(name, ArgsT, blockfn, 4)
#define R2L_KERNEL(name, ArgsT, blockfn, LDS_FLOATS)                         \
  static int name(const ArgsT& a, int grid, void* stream) {                  \
    (void)stream;                                                            \
    std::vector<float> buf((size_t)(LDS_FLOATS) + 8);                        \
    float* lds = (float*)(((uintptr_t)buf.data() + 15) & ~(uintptr_t)15);    \
    for (int b = 0; b < grid; ++b) blockfn(a, b, grid, lds);                 \
    return 0;                                                                \
  }
#else
// Optional per-kernel timing (bench.py's roofline leg): when enabled, every launch is bracketed by
// hipEvents recorded on the stream the kernel is launched on; r2l_timing_report() synchronises the
// events and returns "name count total_ms" lines.  Off by default; costs one branch per launch.
#include <map>
#include <mutex>
struct R2LTimedLaunch {
  const char* name;
  hipEvent_t e0, e1;
};
static std::mutex r2l_timing_mutex;
static bool r2l_timing_on = false;
static std::vector<R2LTimedLaunch> r2l_timed;
static void r2l_time_begin(const char* name, hipStream_t s, R2LTimedLaunch& t) {
  t.name = name;
  (void)hipEventCreate(&t.e0);
  (void)hipEventCreate(&t.e1);
  (void)hipEventRecord(t.e0, s);
}
static void r2l_time_end(hipStream_t s, R2LTimedLaunch& t) {
  (void)hipEventRecord(t.e1, s);
  std::lock_guard<std::mutex> g(r2l_timing_mutex);
  r2l_timed.push_back(t);
}
// Diagnostic builds (-DR2L_TEST_HOOKS) can put something in front of a launch whose name contains one of the comma-separated
// substrings of an environment variable, outside the launch's timing events (tests/experiments/mall_xcd_probe.py):
//   R2L_EXP_FLUSH  a pass that reads and re-writes a 768 MB scratch allocation (evicts the L2s and the 256 MB memory-side
//                  cache: what the kernel costs when its predecessor left it nothing);
//   R2L_EXP_TWICE  an untimed launch of the same kernel (what it costs when everything it touches was touched just now).
#ifdef R2L_TEST_HOOKS
__global__ __launch_bounds__(256) void r2l_exp_flush_kernel(float4* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float4 v = p[i];
    v.x += 1.f;
    p[i] = v;
  }
}
static bool r2l_exp_match(const char* env, const char* name) {
  const char* s = getenv(env);
  if (!s || !*s) return false;
  std::string list(s), nm(name);
  size_t pos = 0;
  while (pos <= list.size()) {
    size_t c = list.find(',', pos);
    if (c == std::string::npos) c = list.size();
    const std::string tok = list.substr(pos, c - pos);
    if (!tok.empty() && (tok == "all" || nm.find(tok) != std::string::npos)) return true;
    pos = c + 1;
  }
  return false;
}
static void r2l_exp_pre_launch(const char* name, hipStream_t s) {
  if (!r2l_exp_match("R2L_EXP_FLUSH", name)) return;
  static float4* buf = nullptr;
  const size_t n = ((size_t)768 << 20) / sizeof(float4);
  if (!buf && hipMalloc((void**)&buf, n * sizeof(float4)) != hipSuccess) return;
  hipLaunchKernelGGL(r2l_exp_flush_kernel, dim3(4096), dim3(256), 0, s, buf, n);
}
#define R2L_PRE_LAUNCH(name, stream) r2l_exp_pre_launch(name, (hipStream_t)(stream))
#define R2L_TWICE(name) r2l_exp_match("R2L_EXP_TWICE", name)
#else
#define R2L_PRE_LAUNCH(name, stream)
#define R2L_TWICE(name) false
#endif
#define R2L_KERNEL(name, ArgsT, blockfn, LDS_FLOATS) R2L_KERNEL_OCC(name, ArgsT, blockfn, LDS_FLOATS, 1)
// LDS-free kernels with their own workgroup size (independent wavefronts)
#define R2L_KERNEL_NT(name, ArgsT, blockfn, NT, WAVES_PER_SIMD)                                 \
  __global__ __launch_bounds__(NT, WAVES_PER_SIMD) void name##_kernel(const ArgsT a) {         \
    blockfn(a, (int)blockIdx.x, (int)gridDim.x, nullptr);                                      \
  }                                                                                            \
  static int name(const ArgsT& a, int grid, void* stream) {                                    \
    R2LTimedLaunch t_;                                                                         \
    const bool timed_ = r2l_timing_on;                                                         \
    if (R2L_TWICE(#name)) hipLaunchKernelGGL(name##_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, a); \
    R2L_PRE_LAUNCH(#name, stream);                                                             \
    if (timed_) r2l_time_begin(#name "_kernel", (hipStream_t)stream, t_);                      \
    hipLaunchKernelGGL(name##_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, a);       \
    if (timed_) r2l_time_end((hipStream_t)stream, t_);                                         \
    const hipError_t e = hipGetLastError();                                                    \
    if (e != hipSuccess) return r2l_fail(-10, std::string(#name ": ") + hipGetErrorString(e)); \
    return 0;                                                                                  \
  }
// kernels with their own workgroup size AND static LDS
#define R2L_KERNEL_NT_LDS(name, ArgsT, NT, LDS_FLOATS, WAVES_PER_SIMD, ...)                    \
  __global__ __launch_bounds__(NT, WAVES_PER_SIMD) void name##_kernel(const ArgsT a) {         \
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];                             \
    __VA_ARGS__(a, (int)blockIdx.x, (int)gridDim.x, lds);                                      \
  }                                                                                            \
  static int name(const ArgsT& a, int grid, void* stream) {                                    \
    R2LTimedLaunch t_;                                                                         \
    const bool timed_ = r2l_timing_on;                                                         \
    if (R2L_TWICE(#name)) hipLaunchKernelGGL(name##_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, a); \
    R2L_PRE_LAUNCH(#name, stream);                                                             \
    if (timed_) r2l_time_begin(#name "_kernel", (hipStream_t)stream, t_);                      \
    hipLaunchKernelGGL(name##_kernel, dim3(grid), dim3(NT), 0, (hipStream_t)stream, a);       \
    if (timed_) r2l_time_end((hipStream_t)stream, t_);                                         \
    const hipError_t e = hipGetLastError();                                                    \
    if (e != hipSuccess) return r2l_fail(-10, std::string(#name ": ") + hipGetErrorString(e)); \
    return 0;                                                                                  \
  }
#define R2L_KERNEL_V(name, ArgsT, LDS_FLOATS, WAVES_PER_SIMD, ...)                             \
  __global__ __launch_bounds__(R2L_NT, WAVES_PER_SIMD) void name##_kernel(const ArgsT a) {     \
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];                             \
    __VA_ARGS__(a, (int)blockIdx.x, (int)gridDim.x, lds);                                      \
  }                                                                                            \
  static int name(const ArgsT& a, int grid, void* stream) {                                    \
    R2LTimedLaunch t_;                                                                         \
    const bool timed_ = r2l_timing_on;                                                         \
    if (R2L_TWICE(#name)) hipLaunchKernelGGL(name##_kernel, dim3(grid), dim3(R2L_NT), 0, (hipStream_t)stream, a); \
    R2L_PRE_LAUNCH(#name, stream);                                                             \
    if (timed_) r2l_time_begin(#name "_kernel", (hipStream_t)stream, t_);                      \
    hipLaunchKernelGGL(name##_kernel, dim3(grid), dim3(R2L_NT), 0, (hipStream_t)stream, a);   \
    if (timed_) r2l_time_end((hipStream_t)stream, t_);                                         \
    const hipError_t e = hipGetLastError();                                                    \
    if (e != hipSuccess) return r2l_fail(-10, std::string(#name ": ") + hipGetErrorString(e)); \
    return 0;                                                                                  \
  }
#define R2L_KERNEL_OCC(name, ArgsT, blockfn, LDS_FLOATS, WAVES_PER_SIMD)                       \
  __global__ __launch_bounds__(R2L_NT, WAVES_PER_SIMD) void name##_kernel(const ArgsT a) {                     \
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];                             \
    blockfn(a, (int)blockIdx.x, (int)gridDim.x, lds);                                          \
  }                                                                                            \
  static int name(const ArgsT& a, int grid, void* stream) {                                    \
    R2LTimedLaunch t_;                                                                         \
    const bool timed_ = r2l_timing_on;                                                         \
    if (R2L_TWICE(#name)) hipLaunchKernelGGL(name##_kernel, dim3(grid), dim3(R2L_NT), 0, (hipStream_t)stream, a); \
    R2L_PRE_LAUNCH(#name, stream);                                                             \
    if (timed_) r2l_time_begin(#name "_kernel", (hipStream_t)stream, t_);                      \
    hipLaunchKernelGGL(name##_kernel, dim3(grid), dim3(R2L_NT), 0, (hipStream_t)stream, a);   \
    if (timed_) r2l_time_end((hipStream_t)stream, t_);                                         \
    const hipError_t e = hipGetLastError();                                                    \
    if (e != hipSuccess) return r2l_fail(-10, std::string(#name ": ") + hipGetErrorString(e)); \
    return 0;                                                                                  \
  }
#endif

typedef R2LGeom<64, 64> GFwd;
typedef R2LGeom<64, 64> GBwd1;
typedef R2LGeom<64, 64> GBwd2;

#define R2L_LDS3(G) (R2L_FOLDED_FLOATS + 2 * G::PAD + 3 * G::PLANE)
#define R2L_LDS4(G) (R2L_FOLDED_FLOATS + 2 * G::PAD + 4 * G::PLANE)
static_assert(R2L_LDS3(GFwd) >= R2L_RED_FLOATS, "reduction scratch must fit");
static_assert(R2L_LDS3(GBwd2) >= R2L_RED_FLOATS, "reduction scratch must fit");

R2L_KERNEL(r2l_launch_fold, R2LFoldArgs, r2l_fold_block, 4)
R2L_KERNEL(r2l_launch_unfold, R2LUnfoldArgs, r2l_unfold_block, 4 + 2 * R2L_NSUMS + 2 * R2L_UNFOLD_TG + R2L_P_COUNT + 4)
R2L_KERNEL(r2l_launch_bn_finalize, R2LBnFinalizeArgs, r2l_bn_finalize_block, 4)
R2L_KERNEL(r2l_launch_bn_bwd_means, R2LBnBwdMeansArgs, r2l_bn_bwd_means_block, 4)
R2L_KERNEL(r2l_launch_reduce_rows, R2LReduceRowsArgs, r2l_reduce_rows_block, 2 * R2L_NT + 64)
#ifndef R2L_OCC_FWD
#define R2L_OCC_FWD 4
#endif
#ifndef R2L_OCC_BWD1
#define R2L_OCC_BWD1 2
#endif
#ifndef R2L_OCC_BWD2
#define R2L_OCC_BWD2 4
#endif
#if R2L_OCC_BWD2 >= 4 && R2L_B2_PREFETCH
#error "bwd2 at two workgroups per CU needs -DR2L_B2_PREFETCH=0 -DR2L_RPI_ADJ=3 (128 VGPRs)"
#endif
// hot instantiation (frames that tile exactly, no additive layer) + the general ones; each again for 16-bit
// container frames (compile-time, so that the float32 kernels carry no decode code)
R2L_KERNEL_V(r2l_launch_fwd, R2LFwdArgs, R2L_LDS3(GFwd), R2L_OCC_FWD, r2l_fwd_block<GFwd, false, false, false>)
R2L_KERNEL_V(r2l_launch_fwd_ragged, R2LFwdArgs, R2L_LDS3(GFwd), 2, r2l_fwd_block<GFwd, false, true, false>)
R2L_KERNEL_V(r2l_launch_fwd_add, R2LFwdArgs, R2L_LDS3(GFwd), 2, r2l_fwd_block<GFwd, true, true, false>)
// the additive layer on frames that tile exactly -- the reference's only case: its layer is 256 x 256 (pipeline_torch.py:130)
R2L_KERNEL_V(r2l_launch_fwd_add_exact, R2LFwdArgs, R2L_LDS3(GFwd), R2L_OCC_FWD, r2l_fwd_block<GFwd, true, false, false>)
#ifndef R2L_SERIAL
// the forward as a row-streaming kernel (r2l_param_stream.h): NW wavefronts side by side cover 256 * NW columns
#ifndef R2L_FS_OCC
#define R2L_FS_OCC 3
#endif
#ifndef R2L_FS_STATS_KERNEL
#define R2L_FS_STATS_KERNEL 1
#endif
#ifndef R2L_FS_MINBAND
#define R2L_FS_MINBAND 16  // rows: shortest band of the row-streaming forward (7 halo rows of luma per band)
#endif
// (8 wavefronts side by side -- frames 1024 < W <= 2048 -- need 99 KB of LDS: one workgroup per CU, 2 wavefronts per SIMD)
#define R2L_FS_OCC_NW(NW) ((NW) == 8 ? 2 : R2L_FS_OCC)
#define R2L_FS_KERNEL(name, NW, U16)                                                                    \
  R2L_KERNEL_NT_LDS(name, R2LFwdStreamArgs, (NW) * 64, R2L_FS_LDS_FLOATS(NW), R2L_FS_OCC_NW(NW), r2l_fwd_stream_block<NW, U16>)
R2L_FS_KERNEL(r2l_launch_fwd_stream_w1, 1, false)
R2L_FS_KERNEL(r2l_launch_fwd_stream_w2, 2, false)
R2L_FS_KERNEL(r2l_launch_fwd_stream_w4, 4, false)
R2L_FS_KERNEL(r2l_launch_fwd_stream_w8, 8, false)
R2L_FS_KERNEL(r2l_launch_fwd_stream_w1_u16, 1, true)
R2L_FS_KERNEL(r2l_launch_fwd_stream_w2_u16, 2, true)
R2L_FS_KERNEL(r2l_launch_fwd_stream_w4_u16, 4, true)
R2L_FS_KERNEL(r2l_launch_fwd_stream_w8_u16, 8, true)
// ... the statistics pass of train-mode BatchNorm (no output; keeps Y' when asked): its own instantiation
#define R2L_FS_KERNEL_STATS(name, NW, U16)                                                              \
  R2L_KERNEL_NT_LDS(name, R2LFwdStreamArgs, (NW) * 64, R2L_FS_LDS_FLOATS(NW), R2L_FS_OCC_NW(NW), r2l_fwd_stream_block<NW, U16, false, true>)
R2L_FS_KERNEL_STATS(r2l_launch_fwd_stream_stats_w1, 1, false)
R2L_FS_KERNEL_STATS(r2l_launch_fwd_stream_stats_w2, 2, false)
R2L_FS_KERNEL_STATS(r2l_launch_fwd_stream_stats_w4, 4, false)
R2L_FS_KERNEL_STATS(r2l_launch_fwd_stream_stats_w8, 8, false)
R2L_FS_KERNEL_STATS(r2l_launch_fwd_stream_stats_w1_u16, 1, true)
R2L_FS_KERNEL_STATS(r2l_launch_fwd_stream_stats_w2_u16, 2, true)
R2L_FS_KERNEL_STATS(r2l_launch_fwd_stream_stats_w4_u16, 4, true)
R2L_FS_KERNEL_STATS(r2l_launch_fwd_stream_stats_w8_u16, 8, true)
// ... with the output epilogue (flip / flip / rot90 of the output planes as part of the stores, R2LEpi)
#define R2L_FS_KERNEL_EPI(name, NW, U16)                                                                \
  R2L_KERNEL_NT_LDS(name, R2LFwdStreamArgs, (NW) * 64, R2L_FS_LDS_FLOATS(NW), R2L_FS_OCC_NW(NW), r2l_fwd_stream_block<NW, U16, true>)
R2L_FS_KERNEL_EPI(r2l_launch_fwd_stream_epi_w1, 1, false)
R2L_FS_KERNEL_EPI(r2l_launch_fwd_stream_epi_w2, 2, false)
R2L_FS_KERNEL_EPI(r2l_launch_fwd_stream_epi_w4, 4, false)
R2L_FS_KERNEL_EPI(r2l_launch_fwd_stream_epi_w8, 8, false)
R2L_FS_KERNEL_EPI(r2l_launch_fwd_stream_epi_w1_u16, 1, true)
R2L_FS_KERNEL_EPI(r2l_launch_fwd_stream_epi_w2_u16, 2, true)
R2L_FS_KERNEL_EPI(r2l_launch_fwd_stream_epi_w4_u16, 4, true)
R2L_FS_KERNEL_EPI(r2l_launch_fwd_stream_epi_w8_u16, 8, true)
// the apply pass of train-mode BatchNorm on the Y' plane the statistics pass kept: independent wavefronts, no LDS
#ifndef R2L_FA_OCC
#define R2L_FA_OCC 3
#endif
#ifndef R2L_FA_NWV
#define R2L_FA_NWV 4  // wavefronts (= work items) per workgroup of the apply and luma passes
#endif
#define R2L_FA_KERNEL(name, U16, EPI) \
  R2L_KERNEL_NT_LDS(name, R2LFwdStreamArgs, 64 * R2L_FA_NWV, 4, R2L_FA_OCC, r2l_fwd_apply_block<U16, EPI, false, R2L_FA_NWV>)
R2L_FA_KERNEL(r2l_launch_fwd_apply, false, false)
R2L_FA_KERNEL(r2l_launch_fwd_apply_u16, true, false)
R2L_FA_KERNEL(r2l_launch_fwd_apply_epi, false, true)
R2L_FA_KERNEL(r2l_launch_fwd_apply_epi_u16, true, true)
// ... the same walk without output: the BatchNorm statistics from the kept plane (2 wavefronts per workgroup, each with
// its own work items; <= R2L_MAX_BLOCKS workgroups = partials of the reduction tree)
#define R2L_FA_STATS_NWV 4
#ifndef R2L_FA_STATS_OCC
#define R2L_FA_STATS_OCC 3
#endif
R2L_KERNEL_NT_LDS(r2l_launch_fwd_stats, R2LFwdStreamArgs, 64 * R2L_FA_STATS_NWV, R2L_FA_LDS_FLOATS(R2L_FA_STATS_NWV, true),
                  R2L_FA_STATS_OCC, r2l_fwd_apply_block<false, false, true, R2L_FA_STATS_NWV>)
R2L_KERNEL_NT_LDS(r2l_launch_fwd_stats_u16, R2LFwdStreamArgs, 64 * R2L_FA_STATS_NWV,
                  R2L_FA_LDS_FLOATS(R2L_FA_STATS_NWV, true), R2L_FA_STATS_OCC,
                  r2l_fwd_apply_block<true, false, true, R2L_FA_STATS_NWV>)
// the luma pass in front of them: raw -> Y' (independent wavefronts, no LDS)
#ifndef R2L_FL_OCC
#define R2L_FL_OCC 4
#endif
R2L_KERNEL_NT_LDS(r2l_launch_fwd_luma, R2LFwdStreamArgs, 64 * R2L_FA_NWV, 4, R2L_FL_OCC, r2l_fwd_luma_block<false, R2L_FA_NWV>)
R2L_KERNEL_NT_LDS(r2l_launch_fwd_luma_u16, R2LFwdStreamArgs, 64 * R2L_FA_NWV, 4, R2L_FL_OCC, r2l_fwd_luma_block<true, R2L_FA_NWV>)
#endif
R2L_KERNEL_V(r2l_launch_bwd1, R2LBwd1Args, R2L_LDS3(GBwd1), R2L_OCC_BWD1, r2l_bwd1_block<GBwd1, false, false, false>)
R2L_KERNEL_V(r2l_launch_bwd1_ragged, R2LBwd1Args, R2L_LDS3(GBwd1), 2, r2l_bwd1_block<GBwd1, false, true, false>)
// ... with Y' taken from the plane the streaming forward kept (one workgroup per CU: the second prefetch frame
// takes the kernel past 256 VGPRs)
#ifndef R2L_OCC_BWD1S
#define R2L_OCC_BWD1S 1
#endif
R2L_KERNEL_V(r2l_launch_bwd1_saved, R2LBwd1Args, R2L_LDS3(GBwd1) + GBwd1::PAD + R2L_B1_FRAME_FLOATS, R2L_OCC_BWD1S, r2l_bwd1_block<GBwd1, false, false, false, true>)
R2L_KERNEL_V(r2l_launch_bwd1_saved_u16, R2LBwd1Args, R2L_LDS3(GBwd1) + GBwd1::PAD + R2L_B1_FRAME_FLOATS, R2L_OCC_BWD1S, r2l_bwd1_block<GBwd1, false, false, true, true>)
// (frames that do not tile by 64 below 4 Mi px take r2l_launch_bwd1_ragged -- Y' recomputed in LDS -- also when the forward kept
// Y': the kept-plane form of the general instantiation needed 76 B of scratch per lane, round 6)
R2L_KERNEL_V(r2l_launch_bwd1_add, R2LBwd1Args, R2L_LDS3(GBwd1), 2, r2l_bwd1_block<GBwd1, true, true, false>)
R2L_KERNEL_V(r2l_launch_bwd1_add_exact, R2LBwd1Args, R2L_LDS3(GBwd1), 2, r2l_bwd1_block<GBwd1, true, false, false>)
#ifndef R2L_SERIAL
// kernel B1 as two passes over planes (r2l_param_plane_bwd.h): where the forward kept Y' and no epilogue / additive layer
R2L_KERNEL_NT_LDS(r2l_launch_bwd1_plane, R2LBwd1Args, R2L_BP_NT, R2L_BP_LDS_FLOATS, 2, r2l_bwd1_plane_block<false, false>)
R2L_KERNEL_NT_LDS(r2l_launch_bwd1_plane_u16, R2LBwd1Args, R2L_BP_NT, R2L_BP_LDS_FLOATS, 2, r2l_bwd1_plane_block<true, false>)
R2L_KERNEL_NT_LDS(r2l_launch_bwd1_plane_epi, R2LBwd1Args, R2L_BP_NT, R2L_BP_LDS_FLOATS, 2, r2l_bwd1_plane_block<false, true>)
R2L_KERNEL_NT_LDS(r2l_launch_bwd1_plane_epi_u16, R2LBwd1Args, R2L_BP_NT, R2L_BP_LDS_FLOATS, 2, r2l_bwd1_plane_block<true, true>)
R2L_KERNEL_NT_LDS(r2l_launch_bwd1_blur, R2LBwd1Args, R2L_BP_NT, R2L_BP_RED_FLOATS, 3, r2l_bwd1_blur_block)
R2L_KERNEL_NT_LDS(r2l_launch_bwd1_blur_hp, R2LBwd1Args, R2L_BP_NT, R2L_BP_RED_FLOATS, R2L_HB_OCC, r2l_bwd1_blur_hp_block)
// kernel B2 likewise: the blur's adjoint into a plane, then the sums + the final reduction and unfold
R2L_KERNEL_NT_LDS(r2l_launch_bwd2_hp, R2LBwd2Args, R2L_BP_NT, 4, 4, r2l_bwd2_hp_block)
R2L_KERNEL_NT_LDS(r2l_launch_bwd2_sums, R2LBwd2Args, R2L_B2S_NT, R2L_B2S_LDS_FLOATS, 3, r2l_bwd2_sums_block<false>)
R2L_KERNEL_NT_LDS(r2l_launch_bwd2_sums_u16, R2LBwd2Args, R2L_B2S_NT, R2L_B2S_LDS_FLOATS, 3, r2l_bwd2_sums_block<true>)
// BatchNorm's backward sums from the raw frame, Y' and grad_out (xhat recomputed, the output not read back): r2l_bnr_planes_block
#ifndef R2L_BNR_OCC
#define R2L_BNR_OCC 3
#endif
#define R2L_BNR_NWV 4
#define R2L_BNR_KERNEL(name, U16, EPI)                                                                          \
  R2L_KERNEL_NT_LDS(name, R2LBnrArgs, 64 * R2L_BNR_NWV, R2L_FA_LDS_FLOATS(R2L_BNR_NWV, true), R2L_BNR_OCC,      \
                    r2l_bnr_planes_block<U16, EPI, R2L_BNR_NWV>)
R2L_BNR_KERNEL(r2l_launch_bnr_planes, false, false)
R2L_BNR_KERNEL(r2l_launch_bnr_planes_u16, true, false)
R2L_BNR_KERNEL(r2l_launch_bnr_planes_epi, false, true)
R2L_BNR_KERNEL(r2l_launch_bnr_planes_epi_u16, true, true)
#endif
R2L_KERNEL_V(r2l_launch_bwd2, R2LBwd2Args, R2L_LDS3(GBwd2), R2L_OCC_BWD2, r2l_bwd2_block<GBwd2, false>)
R2L_KERNEL_V(r2l_launch_fwd_u16, R2LFwdArgs, R2L_LDS3(GFwd), R2L_OCC_FWD, r2l_fwd_block<GFwd, false, false, true>)
R2L_KERNEL_V(r2l_launch_fwd_ragged_u16, R2LFwdArgs, R2L_LDS3(GFwd), 2, r2l_fwd_block<GFwd, false, true, true>)
R2L_KERNEL_V(r2l_launch_fwd_add_u16, R2LFwdArgs, R2L_LDS3(GFwd), 2, r2l_fwd_block<GFwd, true, true, true>)
R2L_KERNEL_V(r2l_launch_fwd_add_exact_u16, R2LFwdArgs, R2L_LDS3(GFwd), R2L_OCC_FWD, r2l_fwd_block<GFwd, true, false, true>)
R2L_KERNEL_V(r2l_launch_bwd1_u16, R2LBwd1Args, R2L_LDS3(GBwd1), R2L_OCC_BWD1, r2l_bwd1_block<GBwd1, false, false, true>)
R2L_KERNEL_V(r2l_launch_bwd1_ragged_u16, R2LBwd1Args, R2L_LDS3(GBwd1), 2, r2l_bwd1_block<GBwd1, false, true, true>)
R2L_KERNEL_V(r2l_launch_bwd1_add_u16, R2LBwd1Args, R2L_LDS3(GBwd1), 2, r2l_bwd1_block<GBwd1, true, true, true>)
R2L_KERNEL_V(r2l_launch_bwd1_add_exact_u16, R2LBwd1Args, R2L_LDS3(GBwd1), 2, r2l_bwd1_block<GBwd1, true, false, true>)
R2L_KERNEL_V(r2l_launch_bwd2_u16, R2LBwd2Args, R2L_LDS3(GBwd2), R2L_OCC_BWD2, r2l_bwd2_block<GBwd2, true>)
// 14 KB of LDS instead of 68 KB: 8 workgroups' worth of loads in flight per CU instead of 2
R2L_KERNEL_OCC(r2l_launch_bn_reduce, R2LBnReduceArgs, r2l_bn_reduce_block, R2L_RED_FLOATS_N(6), 8)
R2L_KERNEL(r2l_launch_add_bwd, R2LAddBwdArgs, r2l_add_bwd_block, 4)
R2L_KERNEL(r2l_launch_raw2rgb_fwd, R2LRaw2RgbArgs, r2l_raw2rgb_fwd_block, 4)
R2L_KERNEL(r2l_launch_raw2rgb_bwd, R2LRaw2RgbArgs, r2l_raw2rgb_bwd_block, R2L_RED_FLOATS)
R2L_KERNEL(r2l_launch_static_full, R2LStaticArgs, r2l_static_block<GStatic>, R2L_STATIC_LDS_FLOATS)
// 3 wavefronts per SIMD with 5 rows in flight each beat 4 with 3 by 1 % (same-buffer A/B): the depth is what counts
#ifndef R2L_STREAM_OCC_BILINEAR
#define R2L_STREAM_OCC_BILINEAR 3
#endif
#ifndef R2L_STREAM_OCC_MALVAR
#define R2L_STREAM_OCC_MALVAR 3
#endif
#define R2L_STREAM_KERNEL(name, DEB, RAWK, LUMA, OCC)                                                    \
  R2L_BLOCKFN void name##_block(const R2LStaticStreamArgs& sa, int bid, int nblk, float* lds) {          \
    r2l_static_stream_block<DEB, RAWK, LUMA>(sa, bid, nblk, lds);                                        \
  }                                                                                                      \
  R2L_KERNEL_NT(name, R2LStaticStreamArgs, name##_block, R2L_STREAM_NT, OCC)
// demosaic x frame container (float32 | 16-bit | float64) x {whole short chain, luma-plane passes}
R2L_STREAM_KERNEL(r2l_launch_static_stream_bilinear, 0, R2L_RAW_F32, false, R2L_STREAM_OCC_BILINEAR)
R2L_STREAM_KERNEL(r2l_launch_static_stream_malvar, 1, R2L_RAW_F32, false, R2L_STREAM_OCC_MALVAR)
R2L_STREAM_KERNEL(r2l_launch_static_stream_bilinear_u16, 0, R2L_RAW_U16, false, R2L_STREAM_OCC_BILINEAR)
R2L_STREAM_KERNEL(r2l_launch_static_stream_malvar_u16, 1, R2L_RAW_U16, false, R2L_STREAM_OCC_MALVAR)
R2L_STREAM_KERNEL(r2l_launch_static_stream_bilinear_f64, 0, R2L_RAW_F64, false, 3)
R2L_STREAM_KERNEL(r2l_launch_static_stream_malvar_f64, 1, R2L_RAW_F64, false, 2)
R2L_STREAM_KERNEL(r2l_launch_static_luma_bilinear, 0, R2L_RAW_F32, true, R2L_STREAM_OCC_BILINEAR)
R2L_STREAM_KERNEL(r2l_launch_static_luma_malvar, 1, R2L_RAW_F32, true, R2L_STREAM_OCC_MALVAR)
R2L_STREAM_KERNEL(r2l_launch_static_luma_bilinear_u16, 0, R2L_RAW_U16, true, R2L_STREAM_OCC_BILINEAR)
R2L_STREAM_KERNEL(r2l_launch_static_luma_malvar_u16, 1, R2L_RAW_U16, true, R2L_STREAM_OCC_MALVAR)
R2L_STREAM_KERNEL(r2l_launch_static_luma_bilinear_f64, 0, R2L_RAW_F64, true, 3)
R2L_STREAM_KERNEL(r2l_launch_static_luma_malvar_f64, 1, R2L_RAW_F64, true, 2)
R2L_KERNEL(r2l_launch_plane_filter, R2LPlaneArgs, r2l_plane_filter_block, 4)
R2L_KERNEL(r2l_launch_spec_mask, R2LSpecMaskArgs, r2l_spec_mask_block, 4)
R2L_KERNEL(r2l_launch_static_finish, R2LStaticFinishArgs, r2l_static_finish_block, 4)
R2L_KERNEL(r2l_launch_static_menon, R2LMenonArgs, r2l_static_menon_block, 4)
#ifndef R2L_SERIAL
// row-streaming luma chains (r2l_static_chain.h): NW wavefronts side by side cover frames up to 256 * NW columns
#ifndef R2L_CHAIN_OCC
#define R2L_CHAIN_OCC 2
#endif
// behind unsharp_masking the 28 KB chroma ring per strip leaves one wavefront per SIMD on 1024-wide frames anyway:
// 512 registers instead of spills
#ifndef R2L_CHAIN_OCC_SH
#define R2L_CHAIN_OCC_SH 1
#endif
// (the workgroup size and the LDS size follow the frame width at launch time: 64 threads and 16.1 KB per strip)
#define R2L_MAX_DEVICES 64
#ifdef R2L_LOCKSTEP
#define R2L_CHAIN_KERNEL(name, RAWK, DEB, SH, DN)                                                             \
  static int name(const R2LStaticChainArgs& a, int grid, void* stream) {                                      \
    (void)stream;                                                                                             \
    r2l_ls_note(#name);                                                                                       \
    r2l_ls::launch(#name, grid, a.nw * 64, 2 * (size_t)R2L_CHAIN_LDS_DOUBLES(a.nw, SH), &a,                   \
                   [&](int b_, float* lds_) { r2l_static_chain_block<RAWK, DEB, SH, DN>(a, b_, grid, lds_); }); \
    return 0;                                                                                                 \
  }
#else
#define R2L_CHAIN_KERNEL(name, RAWK, DEB, SH, DN)                                                             \
  __global__ __launch_bounds__((SH) ? 256 : 512, (SH) ? R2L_CHAIN_OCC_SH : R2L_CHAIN_OCC) void name##_kernel(const R2LStaticChainArgs a) { \
    extern __shared__ __attribute__((aligned(16))) float r2l_chain_lds[];                                     \
    r2l_static_chain_block<RAWK, DEB, SH, DN>(a, (int)blockIdx.x, (int)gridDim.x, r2l_chain_lds);             \
  }                                                                                                           \
  static int name(const R2LStaticChainArgs& a, int grid, void* stream) {                                      \
    const size_t lds_bytes = sizeof(double) * R2L_CHAIN_LDS_DOUBLES(a.nw, SH);                                \
    {                                                                                                         \
      /* the attribute is per device: remember what each device of this process was granted */               \
      static std::mutex mu_;                                                                                  \
      static size_t lds_ok[R2L_MAX_DEVICES];                                                                  \
      int dev_ = 0;                                                                                           \
      (void)hipGetDevice(&dev_);                                                                              \
      std::lock_guard<std::mutex> g_(mu_);                                                                    \
      size_t& ok_ = lds_ok[(unsigned)dev_ % R2L_MAX_DEVICES];                                                 \
      if (lds_bytes > (ok_ ? ok_ : (size_t)48 * 1024)) {                                                      \
        const hipError_t ea = hipFuncSetAttribute((const void*)name##_kernel,                                 \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);\
        if (ea != hipSuccess) return r2l_fail(-10, std::string(#name ": ") + hipGetErrorString(ea));          \
        ok_ = lds_bytes;                                                                                      \
      }                                                                                                       \
    }                                                                                                         \
    R2LTimedLaunch t_;                                                                                        \
    const bool timed_ = r2l_timing_on;                                                                        \
    if (timed_) r2l_time_begin(#name "_kernel", (hipStream_t)stream, t_);                                     \
    hipLaunchKernelGGL(name##_kernel, dim3(grid), dim3(a.nw * 64), lds_bytes, (hipStream_t)stream, a);        \
    if (timed_) r2l_time_end((hipStream_t)stream, t_);                                                        \
    const hipError_t e = hipGetLastError();                                                                   \
    if (e != hipSuccess) return r2l_fail(-10, std::string(#name ": ") + hipGetErrorString(e));                \
    return 0;                                                                                                 \
  }
#endif
// [16-bit / float64 frames] x [Malvar2004] x [unsharp_masking] x [median_denoising]
#define R2L_CHAIN_KERNELS(sfx, RAWK)                                               \
  R2L_CHAIN_KERNEL(r2l_launch_static_chain##sfx, RAWK, 0, 0, 0)                    \
  R2L_CHAIN_KERNEL(r2l_launch_static_chain_median##sfx, RAWK, 0, 0, 1)             \
  R2L_CHAIN_KERNEL(r2l_launch_static_chain_unsharp##sfx, RAWK, 0, 1, 0)            \
  R2L_CHAIN_KERNEL(r2l_launch_static_chain_unsharp_median##sfx, RAWK, 0, 1, 1)     \
  R2L_CHAIN_KERNEL(r2l_launch_static_chain_malvar##sfx, RAWK, 1, 0, 0)             \
  R2L_CHAIN_KERNEL(r2l_launch_static_chain_malvar_median##sfx, RAWK, 1, 0, 1)      \
  R2L_CHAIN_KERNEL(r2l_launch_static_chain_malvar_unsharp##sfx, RAWK, 1, 1, 0)     \
  R2L_CHAIN_KERNEL(r2l_launch_static_chain_malvar_unsharp_median##sfx, RAWK, 1, 1, 1)
R2L_CHAIN_KERNELS(, R2L_RAW_F32)
R2L_CHAIN_KERNELS(_u16, R2L_RAW_U16)
R2L_CHAIN_KERNELS(_f64, R2L_RAW_F64)
#endif
R2L_KERNEL(r2l_launch_static_short, R2LStaticArgs, r2l_static_short_block<GStatic>,
           R2L_STATIC_SHORT_LDS_FLOATS)

R2L_KERNEL(r2l_launch_conv33_fwd, R2LStageArgs, r2l_conv33_fwd_block, 4)
R2L_KERNEL(r2l_launch_conv33_bwd, R2LStageArgs, r2l_conv33_bwd_block, R2L_RED_FLOATS)
R2L_KERNEL(r2l_launch_mix3_fwd, R2LStageArgs, r2l_mix3_fwd_block, 4)
R2L_KERNEL(r2l_launch_mix3_bwd, R2LStageArgs, r2l_mix3_bwd_block, R2L_RED_FLOATS)
R2L_KERNEL(r2l_launch_pconv_fwd, R2LStageArgs, r2l_pconv_fwd_block, 4)
R2L_KERNEL(r2l_launch_pconv_bwd, R2LStageArgs, r2l_pconv_bwd_block, R2L_RED_FLOATS)
R2L_KERNEL(r2l_launch_point, R2LPointArgs, r2l_point_block, R2L_RED_FLOATS)
R2L_KERNEL(r2l_launch_aug, R2LAugArgs, r2l_aug_block, 4)
R2L_KERNEL(r2l_launch_aug_tiled, R2LAugTiledArgs, r2l_aug_tiled_block, R2L_AUG_LDS_FLOATS)
R2L_KERNEL(r2l_launch_axpy, R2LAxpyArgs, r2l_axpy_block, 4)
R2L_KERNEL(r2l_launch_philox_noise, R2LPhiloxArgs, r2l_philox_noise_block, 4)
R2L_KERNEL(r2l_launch_ssim, R2LSsimArgs, r2l_ssim_block, R2L_SSIM_LDS_FLOATS)
R2L_KERNEL(r2l_launch_ssim_bwd, R2LSsimBwdArgs, r2l_ssim_bwd_block, R2L_SSIM_BWD_LDS_FLOATS)
R2L_KERNEL(r2l_launch_l2, R2LL2Args, r2l_l2_block, R2L_RED_FLOATS_N(1))
R2L_KERNEL(r2l_launch_pack_fold, R2LPackFoldArgs, r2l_pack_fold_block, R2L_P_COUNT + 2)

// ---- grid sizing ------------------------------------------------------------------------------
static_assert(R2L_MAX_BLOCKS == 2048 && 1 + R2L_MAX_GROUPS <= R2L_NT, "partials / arrival counters are laid out for at most 2048 workgroups");
// Launch shapes are compile-time choices of the product build.  Diagnostic builds (-DR2L_TEST_HOOKS: the host
// emulation and tests/_build/libr2l_isp_hooks.so, never the shipped libr2l_isp.so) can override them through the
// environment -- that is how the tests show that no result depends on the workgroup count, and how A/B runs sweep
// band heights.
#ifdef R2L_TEST_HOOKS
static int r2l_env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  if (!s || !*s) return dflt;
  const int v = atoi(s);
  return v > 0 ? v : dflt;
}
#else
static inline int r2l_env_int(const char*, int dflt) { return dflt; }
#endif
// persistent tile-walking kernels: at most `cap` workgroups (256 CUs x resident workgroups per CU),
// a multiple of 8 when possible so that the XCD-grouped walk applies
static int r2l_tile_grid(int ntiles, int cap) {
  if (cap > R2L_MAX_BLOCKS) cap = R2L_MAX_BLOCKS;
  int g = ntiles < cap ? ntiles : cap;
  if (g >= 8) g -= g % 8;
  return g < 1 ? 1 : g;
}

// Band height of the plane passes (r2l_param_stream.h, r2l_param_plane_bwd.h): a multiple of 6 rows (bands start on
// multiples of 6: the ring slot of a row is the unroll position of its step), the one that needs the fewest rounds of
// `slots` resident wavefronts x the rows a wavefront walks (+ 4 load-only warm-up steps + its start) -- 64x512x512 apply
// pass: 24 rows = 2,816 wavefronts, one round at 3 per SIMD (65 us; 18 rows: 68.5; 12: 67.5; 36: 71.5,
// profiles/r03_apply_kept.txt; re-swept under the progress-priority build, profiles/r04_bands.txt); 64x256x256: 6 rows.
// `env`: override of diagnostic builds.
static int r2l_band_rows(int B, int H, int W, long slots, const char* env) {
  const long nstrip = (W + 255) / 256;
  int bh = 6;
  long best = -1;
  for (int c = 6; c <= 48; c += 6) {
    const long items = (long)B * nstrip * ((H + c - 1) / c);
    const long cost = ((items + slots - 1) / slots) * (10L * c + 32);
    if (best < 0 || cost < best) {
      best = cost;
      bh = c;
    }
  }
  return (r2l_env_int(env, bh) + 5) / 6 * 6;
}

// XCD windows of the band passes (r2l_xcd_window): neighbouring workgroups per XCD; measured per pass (profiles/r05_mall_xcd.txt)
#ifndef R2L_XCDM_FS
#define R2L_XCDM_FS 0   // statistics pass (row-streaming forward)
#endif
#ifndef R2L_XCDM_FA
#define R2L_XCDM_FA 0   // apply pass / luma pass / statistics from the plane
#endif
#ifndef R2L_XCDM_BP
#define R2L_XCDM_BP 0   // kernel B1's plane pass
#endif
#ifndef R2L_XCDM_HB
#define R2L_XCDM_HB 0   // blur sums + blur adjoint
#endif
#ifndef R2L_XCDM_B2S
#define R2L_XCDM_B2S 0  // kernel B2's sums pass
#endif
static int r2l_xcdm(const char* env, int dflt) {
#ifdef R2L_TEST_HOOKS
  const char* s = getenv(env);
  if (s && *s) dflt = atoi(s);
#else
  (void)env;
#endif
  return (dflt > 0 && (dflt & (dflt - 1)) == 0) ? dflt : 0;  // a power of two, or off
}

// ---- workspace ----------------------------------------------------------------------------------
struct R2LWorkspace {
  R2LFolded* folded;
  float* part_b1;
  float* part_b2;
  float* part_small;
  double* sums;
  double* gpartial;    // [R2L_MAX_GROUPS][R2L_NSUMS] group partials of the in-kernel final reductions
  unsigned* counters;  // [1 + R2L_MAX_GROUPS] arrival counters: zeroed by the fold kernel, zero after every launch
  float* gypp;
  float* yp;     // (B,H,W): the sharpened luma Y' of the forward, for kernel B1 (R2L_F_KEEP_LUMA)
  float* hp;     // (B,H,W): the blur's adjoint of dL/dY'' (plane passes of kernel B2, r2l_param_plane_bwd.h)
  float* debug;  // 3 x [R2L_MAX_BLOCKS][8] floats: per-phase cycle stamps of diagnostic builds (fwd, bwd1, bwd2)
  // step block (r2l_isp_step_fwd / _bwd): what one training step keeps between its launches
  float* packed;    // [R2L_P_COUNT] the parameter values the forward saw
  float* bn;        // [6] mean, istd
  float* bn_bwd;    // [6] mean(g), mean(g * xhat)
  double* stats;    // [7] this rank's statistics sums (+ pixel count)
  double* moments;  // [7] mean, biased var, pixel count of the global batch
  double* bsums;    // [6] this rank's BatchNorm backward sums
  size_t total;
};
static size_t r2l_align_up(size_t x) { return (x + 255) & ~(size_t)255; }
static R2LWorkspace r2l_carve(void* base, int B, int H, int W) {
  R2LWorkspace w;
  size_t off = 0;
  char* p = (char*)base;
  w.folded = (R2LFolded*)(p + off);
  off += r2l_align_up(sizeof(R2LFolded));
  w.part_b1 = (float*)(p + off);
  off += r2l_align_up(sizeof(float) * R2L_B1_NACC * R2L_MAX_BLOCKS);
  w.part_b2 = (float*)(p + off);
  off += r2l_align_up(sizeof(float) * R2L_B2_NACC * R2L_MAX_BLOCKS);
  w.part_small = (float*)(p + off);
  off += r2l_align_up(sizeof(float) * 12 * R2L_MAX_BLOCKS);
  w.sums = (double*)(p + off);
  off += r2l_align_up(sizeof(double) * R2L_NSUMS);
  w.gpartial = (double*)(p + off);
  off += r2l_align_up(sizeof(double) * R2L_NSUMS * R2L_MAX_GROUPS);
  w.counters = (unsigned*)(p + off);
  off += r2l_align_up(sizeof(unsigned) * (1 + R2L_MAX_GROUPS));
  w.debug = (float*)(p + off);
#ifdef R2L_EXP_STAMPS
  off += r2l_align_up(sizeof(float) * 4 * 8 * R2L_MAX_BLOCKS);  // + the wavefront placement records (tests/timeline_fwd.py)
#else
  off += r2l_align_up(sizeof(float) * 3 * 8 * R2L_MAX_BLOCKS);
#endif
  w.packed = (float*)(p + off);
  off += r2l_align_up(sizeof(float) * R2L_P_COUNT);
  w.bn = (float*)(p + off);
  w.bn_bwd = w.bn + 8;
  off += r2l_align_up(sizeof(float) * 16);
  w.stats = (double*)(p + off);
  w.moments = w.stats + 8;
  w.bsums = w.stats + 16;
  off += r2l_align_up(sizeof(double) * 24);
  w.gypp = (float*)(p + off);
  off += r2l_align_up(sizeof(float) * (size_t)B * H * W);
  w.yp = (float*)(p + off);
  off += r2l_align_up(sizeof(float) * (size_t)B * H * W);
  w.hp = (float*)(p + off);
  off += r2l_align_up(sizeof(float) * (size_t)B * H * W);
  w.total = off;
  return w;
}

static int r2l_check_dims(int B, int H, int W) {
  if (B < 1) return r2l_fail(-1, "B must be >= 1");
  if (H < 4 || W < 4 || (H & 1) || (W & 1))
    return r2l_fail(-1, "H and W must be even and >= 4 (Bayer quads; 5x5 mirror padding)");
  if ((size_t)H * W > ((size_t)1 << 29)) return r2l_fail(-1, "frames above 2^29 pixels are not supported");
  if ((size_t)B * H * W > ((size_t)1 << 40)) return r2l_fail(-1, "batch too large");
  return 0;
}

extern "C" {

int r2l_abi_version(void) { return R2L_ABI_VERSION; }
const char* r2l_last_error(void) { return r2l_err.c_str(); }
int r2l_is_device_build(void) {
#ifdef R2L_EMUL
  return 0;
#else
  return 1;
#endif
}

void r2l_timing_enable(int on) {
#ifndef R2L_EMUL
  std::lock_guard<std::mutex> g(r2l_timing_mutex);
  r2l_timing_on = on != 0;
#elif defined(R2L_LOCKSTEP)
  std::lock_guard<std::mutex> g(r2l_ls_record_mutex);
  r2l_ls_record_on = on != 0;
  if (on) r2l_ls_record.clear();
#else
  (void)on;
#endif
}

int r2l_timing_report(char* buf, size_t n) {
  std::string out;
#ifdef R2L_LOCKSTEP
  {  // the launch record: "name count 0" lines (no clock in the emulation)
    std::lock_guard<std::mutex> g(r2l_ls_record_mutex);
    for (auto& kv : r2l_ls_record) out += kv.first + " " + std::to_string(kv.second) + " 0\n";
    r2l_ls_record.clear();
  }
#endif
#ifndef R2L_EMUL
  std::vector<R2LTimedLaunch> v;
  {
    std::lock_guard<std::mutex> g(r2l_timing_mutex);
    v.swap(r2l_timed);
  }
  std::map<std::string, std::pair<int, double>> acc;
  for (auto& t : v) {
    (void)hipEventSynchronize(t.e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, t.e0, t.e1);
    (void)hipEventDestroy(t.e0);
    (void)hipEventDestroy(t.e1);
    auto& a = acc[t.name];
    a.first += 1;
    a.second += ms;
  }
  for (auto& kv : acc)
    out += kv.first + " " + std::to_string(kv.second.first) + " " + std::to_string(kv.second.second) + "\n";
#endif
  if (!buf || n == 0) return (int)out.size();
  const size_t m = out.size() < n - 1 ? out.size() : n - 1;
  memcpy(buf, out.data(), m);
  buf[m] = 0;
  return (int)m;
}

size_t r2l_isp_workspace_bytes(int B, int H, int W) {
  if (B < 1 || H < 1 || W < 1) return 0;
  return r2l_carve(nullptr, B, H, W).total;
}

static int r2l_check_raw(const R2LRaw& raw, int W, const char* who) {
  if (!raw.f32 && !raw.u16 && !raw.f64) return r2l_fail(-1, std::string(who) + ": null pointer");
  if (raw.f64 && (W & 3)) return r2l_fail(-4, std::string(who) + ": float64 frames need W % 4 == 0");
  if (raw.u16 && !(raw.denom >= 1.f)) return r2l_fail(-1, std::string(who) + ": denom must be >= 1 (2**bits - 1)");
  if (raw.u16 && (W & 3)) return r2l_fail(-1, std::string(who) + ": 16-bit frames need W % 4 == 0");
  return 0;
}

// internal flag of r2l_isp_fwd_impl (r2l_isp_step_fwd sets it): the workspace's Y' plane is this batch's, written by the
// statistics pass with R2L_F_KEEP_LUMA -- the apply pass reads it instead of computing it again
#define R2L_F_LUMA_VALID 1024
// ... and of the statistics pass: luma pass (raw -> Y' into the workspace) + statistics from that plane, instead of the
// streaming forward without output (r2l_isp_step_fwd sets it in train mode)
#define R2L_F_SPLIT_STATS 2048
#define R2L_F_INTERNAL (R2L_F_LUMA_VALID | R2L_F_SPLIT_STATS)
// where the row-streaming forward (r2l_param_stream.h) runs -- and with R2L_F_KEEP_LUMA leaves Y' for kernel B1
static bool r2l_fwd_streams(const float* additive, int W) {
#ifdef R2L_SERIAL
  (void)additive; (void)W;
  return false;
#else
  return !additive && (W & 3) == 0 && W <= 2048 && !r2l_env_int("R2L_FWD_TILED", 0);
#endif
}
static int r2l_isp_fwd_impl(const R2LRaw& raw, const float* params, const float* additive,
                            const float* bn_mean_istd, float* out, double* stats, void* workspace,
                            size_t workspace_bytes, int B, int H, int W, int flags, void* stream,
                            const R2LBnFinalizeArgs* fin = nullptr, const R2LEpi* ep = nullptr) {
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (int e = r2l_check_raw(raw, W, "r2l_isp_fwd")) return e;
  if (!params || !workspace) return r2l_fail(-1, "r2l_isp_fwd: null pointer");
  if (additive && (H != 256 || W != 256))
    return r2l_fail(-1, "additive_layer is (1,3,256,256): needs 256x256 frames");
  if (r2l_env_int("R2L_FORCE_SPLIT", 0)) flags |= R2L_F_INTERNAL;  // (diagnostic builds: tests/timeline_fwd.py)
  const bool stats_only = (flags & R2L_F_STATS_ONLY) != 0;
  if (stats_only) out = nullptr;
  if (!out && !stats) return r2l_fail(-1, "r2l_isp_fwd: nothing to compute (no out, no stats)");
  const R2LWorkspace ws = r2l_carve(workspace, B, H, W);
  if (workspace_bytes < ws.total) return r2l_fail(-2, "r2l_isp_fwd: workspace too small");
  if (!(flags & R2L_F_FOLDED_VALID)) {
    R2LFoldArgs fa{params, ws.folded, ws.counters};
    if (int e = r2l_launch_fold(fa, 1, stream)) return e;
  }
#ifndef R2L_SERIAL
  if (r2l_fwd_streams(additive, W)) {
    // row-streaming forward: work item = (image, band of rows); short bands are cheap here (the 8 halo rows of a
    // band only compute their luma), so aim at ~2048 items (three wavefronts per SIMD: the kernel's row step is a
    // chain of scalar-load waits, which only other wavefronts can fill), bands of >= 16 rows
    R2LFwdStreamArgs fa;
    fa.raw = raw;
    fa.F = ws.folded;
    fa.bn = bn_mean_istd;
    fa.out = out;
    // (a statistics pass of the streaming kernel keeps Y' too: for kernel B1 and for the apply pass)
    fa.yp_out = (flags & (R2L_F_KEEP_LUMA | R2L_F_SPLIT_STATS)) ? ws.yp : nullptr;
    fa.yp_in = nullptr;
#ifdef R2L_EXP_STAMPS
    fa.tl = r2l_env_int("R2L_TL_STREAM", 0) ? (unsigned long long*)ws.debug : nullptr;  // (tests/timeline_fwd.py)
#endif
    fa.stat_partial = stats ? ws.part_small : nullptr;
    fa.B = B;
    fa.H = H;
    fa.W = W;
    // one round of resident workgroups: 256 CUs x 12 wavefronts (three per SIMD) / wavefronts per workgroup
    const int nwv = W <= 256 ? 1 : (W <= 512 ? 2 : (W <= 1024 ? 4 : 8));
    const long resident = 256L * (12 / nwv);
    long nband = r2l_env_int("R2L_FS_BAND", 0) ? (H + r2l_env_int("R2L_FS_BAND", 32) - 1) / r2l_env_int("R2L_FS_BAND", 32)
                                               : resident / B;
    const int minband = r2l_env_int("R2L_FS_MINBAND", R2L_FS_MINBAND);
    if (nband > H / minband) nband = H / minband;
    if (nband < 1) nband = 1;
    fa.band_h = (int)((H + nband - 1) / nband);
    fa.band_h += fa.band_h & 1;
    fa.nband = (H + fa.band_h - 1) / fa.band_h;
    const long nitems = (long)B * fa.nband;
    if (nitems > (1L << 30)) return r2l_fail(-1, "r2l_isp_fwd: batch too large");
    fa.nitems = (int)nitems;
    long cap = r2l_env_int("R2L_GRID_FWD", (int)(resident < R2L_MAX_BLOCKS ? resident : R2L_MAX_BLOCKS));
    if (cap > R2L_MAX_BLOCKS) cap = R2L_MAX_BLOCKS;
    const int sgrid = (int)(nitems < cap ? nitems : cap);
    fa.tree = R2LTree{ws.part_small, nullptr, ws.gpartial, (stats && !(r2l_env_int("R2L_EXP_NO_TREE", 0) & 1)) ? ws.counters : nullptr, 12, 0};
    fa.stats_out = stats;
    if (fin)
      fa.fin = *fin;
    else
      fa.fin.bn = nullptr;
    const int nw = W <= 256 ? 0 : (W <= 512 ? 1 : (W <= 1024 ? 2 : 3));
    typedef int (*launch_t)(const R2LFwdStreamArgs&, int, void*);
    static const launch_t table[2][2][4] = {
        {{r2l_launch_fwd_stream_w1, r2l_launch_fwd_stream_w2, r2l_launch_fwd_stream_w4, r2l_launch_fwd_stream_w8},
         {r2l_launch_fwd_stream_w1_u16, r2l_launch_fwd_stream_w2_u16, r2l_launch_fwd_stream_w4_u16,
          r2l_launch_fwd_stream_w8_u16}},
        {{r2l_launch_fwd_stream_epi_w1, r2l_launch_fwd_stream_epi_w2, r2l_launch_fwd_stream_epi_w4,
          r2l_launch_fwd_stream_epi_w8},
         {r2l_launch_fwd_stream_epi_w1_u16, r2l_launch_fwd_stream_epi_w2_u16, r2l_launch_fwd_stream_epi_w4_u16,
          r2l_launch_fwd_stream_epi_w8_u16}}};
    const bool epi = ep && ep->on && out;
    fa.ep = epi ? *ep : R2LEpi{0, 0, 0, 0};
    fa.xcdm = r2l_xcdm("R2L_XCD_FS", R2L_XCDM_FS);
    // The passes on the kept luma plane (r2l_param_stream.h: r2l_fwd_luma_block, r2l_fwd_apply_block): independent
    // wavefronts, one per (image, band, 256-column strip); band heights: r2l_band_rows
    const long nstrip = (W + 255) / 256;
    auto band_rows = [&](long slots, const char* env) { return r2l_band_rows(B, H, W, slots, env); };
    const bool kept_ok = !r2l_env_int("R2L_FWD_APPLY_RECOMPUTE", 0);
    // statistics pass = luma pass + statistics from the plane, where that is faster than the streaming forward without
    // output: frames one strip wide (64x256x256: 12 + 26 us against 46; 128x256x256: 17 + 36 against 61).  On 512-wide
    // frames the two kernels issue as many vector instructions as the one (7.1 M + 15.9 M against 23.9 M at 64x512x512)
    // and take as long (28.5 + 55 us against 79.7): profiles/r03_split_stats.txt
    const bool split = r2l_env_int("R2L_FWD_STATS_SPLIT", 0) || (W <= 256 && !r2l_env_int("R2L_FWD_STATS_STREAM", 0));
    if (!out && stats && (flags & R2L_F_SPLIT_STATS) && kept_ok && split) {
      R2LFwdStreamArgs la = fa;
      la.stat_partial = nullptr;
      la.xcdm = fa.xcdm = r2l_xcdm("R2L_XCD_FA", R2L_XCDM_FA);
#ifdef R2L_EXP_STAMPS
      la.tl = (unsigned long long*)ws.debug;
      fa.tl = (unsigned long long*)ws.debug + 8192;
#endif
      la.band_h = band_rows(256L * 4 * R2L_FL_OCC, "R2L_FL_BAND");
      la.nband = (H + la.band_h - 1) / la.band_h;
      const long lgrid = (long)B * la.nband * nstrip;
      if (lgrid > (1L << 30)) return r2l_fail(-1, "r2l_isp_fwd: batch too large");
      la.nitems = (int)lgrid;
      const int lg = (int)((lgrid + R2L_FA_NWV - 1) / R2L_FA_NWV);
      if (int e = raw.u16 ? r2l_launch_fwd_luma_u16(la, lg, stream) : r2l_launch_fwd_luma(la, lg, stream)) return e;
      fa.yp_in = ws.yp;
      fa.yp_out = nullptr;
      fa.band_h = band_rows(256L * 4 * R2L_FA_STATS_OCC, "R2L_FST_BAND");
      fa.nband = (H + fa.band_h - 1) / fa.band_h;
      const long items = (long)B * fa.nband * nstrip;
      if (items > (1L << 30)) return r2l_fail(-1, "r2l_isp_fwd: batch too large");
      fa.nitems = (int)items;
      long g = (items + R2L_FA_STATS_NWV - 1) / R2L_FA_STATS_NWV;
      const long gcap = r2l_env_int("R2L_GRID_FWD", R2L_MAX_BLOCKS);
      if (g > gcap) g = gcap;
      if (g > R2L_MAX_BLOCKS) g = R2L_MAX_BLOCKS;
      return raw.u16 ? r2l_launch_fwd_stats_u16(fa, (int)g, stream) : r2l_launch_fwd_stats(fa, (int)g, stream);
    }
    if (out && !stats && (flags & R2L_F_LUMA_VALID) && kept_ok) {
      // apply pass on the kept Y'
#ifdef R2L_EXP_STAMPS
      fa.tl = (unsigned long long*)ws.debug + 16384;
#endif
      fa.yp_in = ws.yp;
      fa.yp_out = nullptr;
      fa.stat_partial = nullptr;
      fa.xcdm = r2l_xcdm("R2L_XCD_FA", R2L_XCDM_FA);
      fa.band_h = band_rows(256L * 4 * R2L_FA_OCC, "R2L_FA_BAND");
      fa.nband = (H + fa.band_h - 1) / fa.band_h;
      const long grid = (long)B * fa.nband * nstrip;
      if (grid > (1L << 30)) return r2l_fail(-1, "r2l_isp_fwd: batch too large");
      fa.nitems = (int)grid;
      static const launch_t atable[2][2] = {{r2l_launch_fwd_apply, r2l_launch_fwd_apply_u16},
                                            {r2l_launch_fwd_apply_epi, r2l_launch_fwd_apply_epi_u16}};
      return atable[epi ? 1 : 0][raw.u16 ? 1 : 0](fa, (int)((grid + R2L_FA_NWV - 1) / R2L_FA_NWV), stream);
    }
#if R2L_FS_STATS_KERNEL
    if (!out && stats) {  // the statistics pass: its own instantiation (no output code, fewer live scalars)
      static const launch_t stable[2][4] = {
          {r2l_launch_fwd_stream_stats_w1, r2l_launch_fwd_stream_stats_w2, r2l_launch_fwd_stream_stats_w4,
           r2l_launch_fwd_stream_stats_w8},
          {r2l_launch_fwd_stream_stats_w1_u16, r2l_launch_fwd_stream_stats_w2_u16, r2l_launch_fwd_stream_stats_w4_u16,
           r2l_launch_fwd_stream_stats_w8_u16}};
      return stable[raw.u16 ? 1 : 0][nw](fa, sgrid, stream);
    }
#endif
    return table[epi ? 1 : 0][raw.u16 ? 1 : 0][nw](fa, sgrid, stream);
  }
#endif
  if (ep && ep->on && out && additive) return r2l_fail(-3, "r2l_isp_fwd: no output epilogue with an additive layer");
  const int ntiles = B * ((H + GFwd::TH - 1) / GFwd::TH) * ((W + GFwd::TW - 1) / GFwd::TW);
  const int grid = r2l_tile_grid(ntiles, r2l_env_int("R2L_GRID_FWD", 512));
  R2LFwdArgs a;
  a.raw = raw;
  a.additive = additive;
  a.F = ws.folded;
  a.bn = bn_mean_istd;
  a.out = out;
  a.stat_partial = stats ? ws.part_small : nullptr;
  a.B = B;
  a.H = H;
  a.W = W;
  a.debug = ws.debug;
  // the statistics are reduced by the last workgroups of the same launch (the workspace's arrival counters
  // are valid: this call or an earlier one on this workspace ran the fold kernel)
  a.tree = R2LTree{ws.part_small, nullptr, ws.gpartial, stats ? ws.counters : nullptr, 12, 0};
  a.stats_out = stats;
  if (fin)
    a.fin = *fin;
  else
    a.fin.bn = nullptr;
  a.ep = (ep && ep->on && out) ? *ep : R2LEpi{0, 0, 0, 0};
  const bool exact = (H % GFwd::TH == 0) && (W % GFwd::TW == 0);
  int e;
  if (raw.u16)
    e = additive ? (exact ? r2l_launch_fwd_add_exact_u16(a, grid, stream) : r2l_launch_fwd_add_u16(a, grid, stream))
                 : (exact ? r2l_launch_fwd_u16(a, grid, stream) : r2l_launch_fwd_ragged_u16(a, grid, stream));
  else
    e = additive ? (exact ? r2l_launch_fwd_add_exact(a, grid, stream) : r2l_launch_fwd_add(a, grid, stream))
                 : (exact ? r2l_launch_fwd(a, grid, stream) : r2l_launch_fwd_ragged(a, grid, stream));
  if (e) return e;
  return 0;
}

int r2l_bn_finalize(const double* stats, int nranks, float* bn_mean_istd, double* moments, float* running_mean,
                    float* running_var, long long* num_batches_tracked, double eps, double momentum,
                    void* stream) {
  const double* totals = stats;
  if (!totals || !bn_mean_istd || nranks < 1) return r2l_fail(-1, "r2l_bn_finalize: null pointer / nranks < 1");
  if ((running_mean == nullptr) != (running_var == nullptr))
    return r2l_fail(-1, "r2l_bn_finalize: running_mean and running_var go together");
  R2LBnFinalizeArgs a{totals, nranks, bn_mean_istd, moments, running_mean, running_var, eps, momentum,
                      num_batches_tracked};
  return r2l_launch_bn_finalize(a, 1, stream);
}

int r2l_bn_bwd_means(const double* gathered_sums, int nranks, const double* n, float* bn_bwd, void* stream) {
  if (!gathered_sums || !n || !bn_bwd || nranks < 1) return r2l_fail(-1, "r2l_bn_bwd_means: null pointer / nranks < 1");
  R2LBnBwdMeansArgs a{gathered_sums, nranks, n, bn_bwd};
  return r2l_launch_bn_bwd_means(a, 1, stream);
}

int r2l_bn_bwd_reduce(const float* grad_out, const float* out, const double* totals, double* sums,
                      float* bn_bwd, void* workspace, size_t workspace_bytes, int B, int H, int W, int flags,
                      void* stream) {
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (!grad_out || !out || !sums || !workspace) return r2l_fail(-1, "r2l_bn_bwd_reduce: null pointer");
  if (bn_bwd && !totals) return r2l_fail(-1, "r2l_bn_bwd_reduce: bn_bwd needs totals (the pixel count)");
  const R2LWorkspace ws = r2l_carve(workspace, B, H, W);
  if (workspace_bytes < ws.total) return r2l_fail(-2, "r2l_bn_bwd_reduce: workspace too small");
  const size_t hw = (size_t)H * W;
  const size_t nitems = (size_t)3 * B * ((hw + R2L_SEG - 1) / R2L_SEG);
  const int cap = r2l_tile_grid(R2L_MAX_BLOCKS, r2l_env_int("R2L_GRID_BNR", 512));
  int grid = nitems < (size_t)cap ? (int)nitems : cap;
  // a workspace that went through r2l_isp_fwd / r2l_isp_bwd has valid arrival counters: the last workgroups
  // of the launch finish the reduction; otherwise a second, tiny launch does
  const bool in_kernel = (flags & R2L_F_FOLDED_VALID) != 0 && !(r2l_env_int("R2L_EXP_NO_TREE", 0) & 2);
  R2LBnReduceArgs a{grad_out, out, ws.part_small, B, H, W,
                    R2LTree{ws.part_small, nullptr, ws.gpartial, in_kernel ? ws.counters : nullptr, 6, 0},
                    sums, totals, bn_bwd, r2l_env_int("R2L_BNR_ORDER", 0)};
  if (int e = r2l_launch_bn_reduce(a, grid, stream)) return e;
  if (in_kernel) return 0;
  R2LReduceRowsArgs r{ws.part_small, sums, grid, 1.0, bn_bwd, nullptr, 0.0, bn_bwd ? totals + 6 : nullptr};
  return r2l_launch_reduce_rows(r, 6, stream);
}

static int r2l_isp_bwd_impl(const R2LRaw& raw, const float* params, const float* additive,
                            const float* bn_mean_istd, const float* bn_bwd, const float* grad_out,
                            float* grad_params, float* grad_raw, void* workspace, size_t workspace_bytes, int B,
                            int H, int W, int flags, void* stream, const R2LEpi* ep = nullptr) {
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (int e = r2l_check_raw(raw, W, "r2l_isp_bwd")) return e;
  if (!params || !grad_out || !grad_params || !workspace)
    return r2l_fail(-1, "r2l_isp_bwd: null pointer");
  if (bn_bwd && !bn_mean_istd) return r2l_fail(-1, "r2l_isp_bwd: bn_bwd given without bn_mean_istd");
  if (additive && (H != 256 || W != 256))
    return r2l_fail(-1, "additive_layer is (1,3,256,256): needs 256x256 frames");
  if (grad_raw)
    return r2l_fail(-3, "r2l_isp_bwd: grad_raw is produced by the staged path, not the fused kernels");
  const R2LWorkspace ws = r2l_carve(workspace, B, H, W);
  if (workspace_bytes < ws.total) return r2l_fail(-2, "r2l_isp_bwd: workspace too small");
  if (!(flags & R2L_F_FOLDED_VALID)) {
    R2LFoldArgs fa{params, ws.folded, ws.counters};
    if (int e = r2l_launch_fold(fa, 1, stream)) return e;
  }
  const int ntiles = B * ((H + GBwd1::TH - 1) / GBwd1::TH) * ((W + GBwd1::TW - 1) / GBwd1::TW);
  const int g1 = r2l_tile_grid(ntiles, r2l_env_int("R2L_GRID_BWD1", 256));
  R2LBwd1Args a1;
  a1.raw = raw;
  a1.additive = additive;
  a1.F = ws.folded;
  a1.bn = bn_mean_istd;
  a1.bn_bwd = bn_bwd;
  a1.gout = grad_out;
  a1.gypp = ws.gypp;
  a1.partial = ws.part_b1;
  a1.B = B;
  a1.H = H;
  a1.W = W;
  a1.debug = ws.debug + 8 * R2L_MAX_BLOCKS;
  const bool saved = (flags & R2L_F_KEEP_LUMA) && r2l_fwd_streams(additive, W) && !r2l_env_int("R2L_BWD1_RECOMPUTE", 0);
  a1.yp = saved ? ws.yp : nullptr;
  a1.ep = (ep && ep->on) ? *ep : R2LEpi{0, 0, 0, 0};
  a1.band_h = 0;
  a1.hp = nullptr;
  a1.band_hb = 0;
  a1.xcdm = r2l_xcdm("R2L_XCD_BP", R2L_XCDM_BP);
  a1.xcdm_hb = r2l_xcdm("R2L_XCD_HB", R2L_XCDM_HB);
  if (a1.ep.on && additive) return r2l_fail(-3, "r2l_isp_bwd: no output epilogue with an additive layer");
  const bool exact = (H % GBwd1::TH == 0) && (W % GBwd1::TW == 0);
  int g1p = 0;  // workgroups of the plane passes, when they run
  bool blur_hp = false;  // ... with r2l_bwd1_blur_hp_block doing kernel B2's first pass
#ifndef R2L_SERIAL
  // ... where there is enough work for their launch tails: 128x256x256 (8.4 Mpx) 102 us against the tile kernels' 110+,
  // 64x256x256 (4.2 Mpx) 77.7 against 80.5 since the tails were shortened (profiles/r04_small.txt; round 3: 99 against 85,
  // and the threshold was 6 Mi px)
  const bool planes = r2l_env_int("R2L_BWD_PLANES", 0) || (size_t)B * H * W >= ((size_t)4 << 20);
  if (saved && planes && !r2l_env_int("R2L_BWD1_TILED", 0)) {
    // persistent workgroups of 4 independent wavefronts, two per CU (<= 256 VGPRs), not more workgroups than kernel B2
    // runs (its last workgroups reduce both kernels' partials); band height as for the forward's plane passes
    const long nstrip = (W + 255) / 256;
    a1.band_h = r2l_band_rows(B, H, W, 256L * 4 * 2, "R2L_BP_BAND");
    const long items = (long)B * nstrip * ((H + a1.band_h - 1) / a1.band_h);
    long g = (items + R2L_BP_NWV - 1) / R2L_BP_NWV;
    const long cap = r2l_env_int("R2L_GRID_BWD1", 512);
    if (g > cap) g = cap;
    if (g > R2L_MAX_BLOCKS) g = R2L_MAX_BLOCKS;
    g1p = (int)g;
  }
#endif
  int e1;
  if (g1p) {
#ifndef R2L_SERIAL
    e1 = a1.ep.on ? (raw.u16 ? r2l_launch_bwd1_plane_epi_u16(a1, g1p, stream) : r2l_launch_bwd1_plane_epi(a1, g1p, stream))
                  : (raw.u16 ? r2l_launch_bwd1_plane_u16(a1, g1p, stream) : r2l_launch_bwd1_plane(a1, g1p, stream));
    // its second pass (the blur-weight sums) and kernel B2's first (the blur's adjoint) read the same plane: one pass
    // when B2 runs as plane passes too
    blur_hp = !r2l_env_int("R2L_BWD2_TILED", 0) && !r2l_env_int("R2L_BWD_SPLIT_BLUR", 0);
    a1.hp = ws.hp;
    // (its own band height: R2L_HB_OCC wavefronts per SIMD; not more workgroups than wrote the first pass's partials)
    a1.band_hb = r2l_band_rows(B, H, W, 256L * 4 * R2L_HB_OCC, "R2L_HB_BAND");
    if (!e1) e1 = blur_hp ? r2l_launch_bwd1_blur_hp(a1, g1p, stream) : r2l_launch_bwd1_blur(a1, g1p, stream);
#else
    e1 = 0;
#endif
  } else if (saved && exact)
    e1 = raw.u16 ? r2l_launch_bwd1_saved_u16(a1, g1, stream) : r2l_launch_bwd1_saved(a1, g1, stream);
  else if (raw.u16)   // (frames that do not tile by 64: Y' recomputed in LDS, kept or not -- a1.yp is not read)
    e1 = additive ? (exact ? r2l_launch_bwd1_add_exact_u16(a1, g1, stream) : r2l_launch_bwd1_add_u16(a1, g1, stream))
                  : (exact ? r2l_launch_bwd1_u16(a1, g1, stream) : r2l_launch_bwd1_ragged_u16(a1, g1, stream));
  else
    e1 = additive ? (exact ? r2l_launch_bwd1_add_exact(a1, g1, stream) : r2l_launch_bwd1_add(a1, g1, stream))
                  : (exact ? r2l_launch_bwd1(a1, g1, stream) : r2l_launch_bwd1_ragged(a1, g1, stream));
  if (e1) return e1;
  const int ntiles2 = B * ((H + GBwd2::TH - 1) / GBwd2::TH) * ((W + GBwd2::TW - 1) / GBwd2::TW);
  // bwd2 fits two workgroups per CU (<= 128 VGPRs, 68 KB of LDS): 512 workgroups
  const int g2 = r2l_tile_grid(ntiles2, r2l_env_int("R2L_GRID_BWD2", R2L_OCC_BWD2 >= 4 ? 512 : 256));
  R2LBwd2Args a2;
  a2.nmain = 0;
  a2.xcdm = r2l_xcdm("R2L_XCD_B2S", R2L_XCDM_B2S);
  a2.b1_partial = nullptr;
  a2.b1_n = 0;
  a2.b1_tot = nullptr;
  a2.raw = raw;
  a2.F = ws.folded;
  a2.gypp = ws.gypp;
  a2.partial = ws.part_b2;
  a2.B = B;
  a2.H = H;
  a2.W = W;
  a2.debug = ws.debug + 16 * R2L_MAX_BLOCKS;
  // B2's last workgroups reduce both kernels' partials (B1's were written by g1 workgroups: every level-1 group
  // of B2's grid adds the B1 partials of its own 16 workgroup ids, as far as they exist) and unfold them into the
  // 132 gradients; if B1 ran MORE workgroups than B2 (R2L_GRID_* overrides of diagnostic builds) three tiny
  // launches do it
  const int g1w = g1p ? g1p : g1;  // workgroups that wrote B1's partials
#ifndef R2L_SERIAL
  if (g1p && !r2l_env_int("R2L_BWD2_TILED", 0)) {
    // kernel B2 as two passes over planes (r2l_param_plane_bwd.h)
    const long nstrip = (W + 255) / 256;
    auto band_rows = [&](long slots, const char* env) { return r2l_band_rows(B, H, W, slots, env); };
    a2.hp = ws.hp;
    a2.band_h = band_rows(256L * 4 * 4, "R2L_HP_BAND");
    const long hitems = (long)B * nstrip * ((H + a2.band_h - 1) / a2.band_h);
    a2.tree = R2LTree{nullptr, nullptr, nullptr, nullptr, 0, 0};
    a2.params = nullptr;
    a2.grad_params = nullptr;
    a2.asym = 0;
    if (!blur_hp)
      if (int e = r2l_launch_bwd2_hp(a2, (int)((hitems + R2L_BP_NWV - 1) / R2L_BP_NWV), stream)) return e;
    a2.band_h = band_rows(256L * 4 * 3, "R2L_B2S_BAND");
    const long sitems = (long)B * nstrip * ((H + a2.band_h - 1) / a2.band_h);
    long gs = (sitems + R2L_B2S_NWV - 1) / R2L_B2S_NWV;
    const long cap = r2l_env_int("R2L_GRID_BWD2", 768);  // 3 workgroups of 4 wavefronts per CU
    if (gs > cap) gs = cap;
    if (gs > R2L_MAX_BLOCKS) gs = R2L_MAX_BLOCKS;
    // its last workgroups reduce B2's partials; R2L_B2S_HELPERS more workgroups (the grid leaves room for them beside one
    // round of the others) add B1's meanwhile; the last arrival of all unfolds the 155 totals into the 132 gradients
    const bool tree = !(r2l_env_int("R2L_EXP_NO_TREE", 0) & 4);
    a2.tree = R2LTree{nullptr, ws.part_b2, ws.gpartial, tree ? ws.counters : nullptr, 0, 0};
    a2.nmain = (int)gs;
    a2.b1_partial = ws.part_b1;
    a2.b1_n = g1w;
    a2.b1_tot = ws.sums;
    a2.params = params;
    a2.grad_params = grad_params;
    const int grid = (int)gs + (tree ? R2L_B2S_HELPERS : 0);
    if (int e = raw.u16 ? r2l_launch_bwd2_sums_u16(a2, grid, stream) : r2l_launch_bwd2_sums(a2, grid, stream)) return e;
    if (tree) return 0;
    // (diagnostic builds with the in-kernel tree switched off: the three tiny launches of the tile kernels' fallback finish
    // the sums, so that grad_params is never returned uninitialised)
    R2LReduceRowsArgs r1{ws.part_b1, ws.sums, g1w, 1.0, nullptr};
    if (int e = r2l_launch_reduce_rows(r1, R2L_B1_NACC, stream)) return e;
    R2LReduceRowsArgs r2{ws.part_b2, ws.sums + R2L_B1_NACC, (int)gs, 1.0, nullptr};
    if (int e = r2l_launch_reduce_rows(r2, R2L_B2_NACC, stream)) return e;
    R2LUnfoldArgs ua{params, ws.sums, grad_params, 1.0f};
    return r2l_launch_unfold(ua, 1, stream);
  }
#endif
  const bool in_kernel = g1w <= g2;
  a2.tree = R2LTree{ws.part_b1, ws.part_b2, ws.gpartial, in_kernel ? ws.counters : nullptr, R2L_B1_NACC, g1w};
  a2.params = params;
  a2.grad_params = grad_params;
  // uneven tile shares for the two workgroups of a CU (r2l_walk_init): measured, no gain -- the younger workgroup is
  // starved while the older one runs and catches up afterwards, the CU finishes its 16 tiles at the same time whatever the
  // split (profiles/r03_bwd2_tile_shares.txt).  Even shares (0) are the default; diagnostic builds can sweep it.
#ifndef R2L_B2_ASYM
#define R2L_B2_ASYM 0
#endif
  a2.asym = (R2L_OCC_BWD2 >= 4 && g2 == 512 && ntiles2 >= 4 * g2) ? r2l_env_int("R2L_B2_ASYM", R2L_B2_ASYM) : 0;
  if (a2.asym == 1) a2.asym = 0;
  if (int e = raw.u16 ? r2l_launch_bwd2_u16(a2, g2, stream) : r2l_launch_bwd2(a2, g2, stream)) return e;
  if (in_kernel) return 0;
  R2LReduceRowsArgs r1{ws.part_b1, ws.sums, g1w, 1.0, nullptr};
  if (int e = r2l_launch_reduce_rows(r1, R2L_B1_NACC, stream)) return e;
  R2LReduceRowsArgs r2{ws.part_b2, ws.sums + R2L_B1_NACC, g2, 1.0, nullptr};
  if (int e = r2l_launch_reduce_rows(r2, R2L_B2_NACC, stream)) return e;
  R2LUnfoldArgs ua{params, ws.sums, grad_params, 1.0f};
  return r2l_launch_unfold(ua, 1, stream);
}

int r2l_additive_bwd(const float* grad_out, const float* out, const float* bn_mean_istd,
                     const float* bn_bwd, float* grad_additive, int B, int H, int W, void* stream) {
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (!grad_out || !grad_additive) return r2l_fail(-1, "r2l_additive_bwd: null pointer");
  if (bn_mean_istd && !out) return r2l_fail(-1, "r2l_additive_bwd: BatchNorm needs the saved output");
  R2LAddBwdArgs a{grad_out, out, bn_mean_istd, bn_bwd, grad_additive, B, H, W};
  const size_t nchunk = (size_t)3 * H * W / 4;
  int grid = (int)((nchunk + R2L_NT - 1) / R2L_NT);
  if (grid > R2L_MAX_BLOCKS) grid = R2L_MAX_BLOCKS;
  return r2l_launch_add_bwd(a, grid, stream);
}

// ---- one training step in two calls (r2l_isp_step_fwd / r2l_isp_step_bwd) --------------------------------
// the output epilogue travels in the bits of `phase` above R2L_STEP_KEEP_LUMA (R2L_STEP_EPI_*); the affine map of
// R2LEpi follows from r2l_aug_map, the definition the stand-alone permutation kernel (r2l_augment) uses
static int r2l_epi_from_phase(int phase, int H, int W, R2LEpi& ep) {
  const int hflip = (phase & R2L_STEP_EPI_HFLIP) != 0, vflip = (phase & R2L_STEP_EPI_VFLIP) != 0;
  const int k = (phase >> R2L_STEP_EPI_ROT_SHIFT) & 3;
  ep = R2LEpi{0, 0, 0, 0};
  if (!(hflip || vflip || k)) return 0;
  if ((k & 1) && H != W) return r2l_fail(-1, "output epilogue: a rotation by 90 degrees needs square frames");
  const int Wo = (k & 1) ? H : W;
  int r0, c0, r1, c1, r2, c2;
  r2l_aug_map(H, W, hflip, vflip, k, 0, 0, r0, c0);
  r2l_aug_map(H, W, hflip, vflip, k, 1, 0, r1, c1);
  r2l_aug_map(H, W, hflip, vflip, k, 0, 1, r2, c2);
  ep.on = 1;
  ep.s0 = r0 * Wo + c0;
  ep.sr = (r1 * Wo + c1) - ep.s0;
  ep.sc = (r2 * Wo + c2) - ep.s0;
  return 0;
}
#define R2L_STEP_EPI_MASK (R2L_STEP_EPI_HFLIP | R2L_STEP_EPI_VFLIP | (3 << R2L_STEP_EPI_ROT_SHIFT))
static R2LRaw r2l_raw_any(const void* raw, int raw_u16, float denom) {
  return raw_u16 ? r2l_raw_u16((const unsigned short*)raw, denom) : r2l_raw_f32((const float*)raw);
}
size_t r2l_isp_step_offset(int which, int B, int H, int W) {
  if (B < 1 || H < 1 || W < 1) return 0;
  const R2LWorkspace ws = r2l_carve(nullptr, B, H, W);
  switch (which) {
    case R2L_STEP_STATS: return (size_t)((char*)ws.stats - (char*)nullptr);
    case R2L_STEP_MOMENTS: return (size_t)((char*)ws.moments - (char*)nullptr);
    case R2L_STEP_BN_SUMS: return (size_t)((char*)ws.bsums - (char*)nullptr);
    case R2L_STEP_PACKED: return (size_t)((char*)ws.packed - (char*)nullptr);
    case R2L_STEP_BN: return (size_t)((char*)ws.bn - (char*)nullptr);
    case R2L_STEP_LUMA: return (size_t)((char*)ws.yp - (char*)nullptr);
    default: return 0;
  }
}
int r2l_isp_step_fwd(const void* raw, int raw_u16, float denom, const float* const* params_host,
                     const float* additive, int bn_mode, float* running_mean, float* running_var,
                     long long* num_batches_tracked, double eps, double momentum, float* out, void* workspace,
                     size_t workspace_bytes, int B, int H, int W, int nranks, int phase,
                     const double* gathered_stats, void* stream) {
  const int keep = (phase & R2L_STEP_KEEP_LUMA) ? R2L_F_KEEP_LUMA : 0;
  R2LEpi ep;
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (int e = r2l_epi_from_phase(phase, H, W, ep)) return e;
  phase &= ~(R2L_STEP_KEEP_LUMA | R2L_STEP_EPI_MASK);
  if (!raw || !out || !workspace) return r2l_fail(-1, "r2l_isp_step_fwd: null pointer");
  if (bn_mode != R2L_BN_NONE && bn_mode != R2L_BN_TRAIN && bn_mode != R2L_BN_EVAL)
    return r2l_fail(-1, "r2l_isp_step_fwd: bn_mode must be R2L_BN_NONE, R2L_BN_TRAIN or R2L_BN_EVAL");
  if (phase != R2L_STEP_ALL && phase != R2L_STEP_A && phase != R2L_STEP_B)
    return r2l_fail(-1, "r2l_isp_step_fwd: unknown phase");
  if (nranks < 1 || (nranks > 1 && phase == R2L_STEP_ALL && bn_mode == R2L_BN_TRAIN))
    return r2l_fail(-1, "r2l_isp_step_fwd: several ranks exchange the statistics between phase A and phase B");
  if (phase != R2L_STEP_ALL && bn_mode != R2L_BN_TRAIN)
    return r2l_fail(-1, "r2l_isp_step_fwd: only train-mode BatchNorm has two phases");
  if (phase == R2L_STEP_B && !gathered_stats) return r2l_fail(-1, "r2l_isp_step_fwd: phase B needs the gathered statistics");
  if (bn_mode == R2L_BN_EVAL && (!running_mean || !running_var))
    return r2l_fail(-1, "r2l_isp_step_fwd: eval-mode BatchNorm needs the running statistics");
  if ((running_mean == nullptr) != (running_var == nullptr))
    return r2l_fail(-1, "r2l_isp_step_fwd: running_mean and running_var go together");
  const R2LRaw rw = r2l_raw_any(raw, raw_u16, denom);
  const R2LWorkspace ws = r2l_carve(workspace, B, H, W);
  if (workspace_bytes < ws.total) return r2l_fail(-2, "r2l_isp_step_fwd: workspace too small (r2l_isp_workspace_bytes)");
  if (phase != R2L_STEP_B) {
    if (!params_host) return r2l_fail(-1, "r2l_isp_step_fwd: null parameter table");
    R2LPackFoldArgs pa;
    for (int i = 0; i < 9; ++i) {
      if (!params_host[i]) return r2l_fail(-1, "r2l_isp_step_fwd: null parameter pointer");
      pa.src[i] = params_host[i];
    }
    pa.packed = ws.packed;
    pa.F = ws.folded;
    pa.counters = ws.counters;
    pa.running_mean = bn_mode == R2L_BN_EVAL ? running_mean : nullptr;
    pa.running_var = running_var;
    pa.bn = ws.bn;
    pa.eps = eps;
    if (int e = r2l_launch_pack_fold(pa, 1, stream)) return e;
  }
  if (bn_mode == R2L_BN_TRAIN && phase != R2L_STEP_B) {
    // statistics pass; one rank: the last workgroup also does the BatchNorm bookkeeping
    R2LBnFinalizeArgs f{ws.stats, 1, ws.bn, ws.moments, running_mean, running_var, eps, momentum, num_batches_tracked};
    if (int e = r2l_isp_fwd_impl(rw, ws.packed, additive, nullptr, nullptr, ws.stats, workspace, workspace_bytes, B, H,
                                 W, R2L_F_STATS_ONLY | R2L_F_FOLDED_VALID | R2L_F_SPLIT_STATS | keep, stream,
                                 phase == R2L_STEP_ALL ? &f : nullptr))
      return e;
    if (phase == R2L_STEP_A) return 0;
  }
  if (phase == R2L_STEP_B) {
    R2LBnFinalizeArgs f{gathered_stats, nranks, ws.bn, ws.moments, running_mean, running_var, eps, momentum,
                        num_batches_tracked};
    if (int e = r2l_launch_bn_finalize(f, 1, stream)) return e;
  }
  // (train mode: the statistics pass of this call -- or of phase A of this step -- has left Y' in the workspace wherever the
  // row-streaming forward runs, and the apply pass reads it)
  return r2l_isp_fwd_impl(rw, ws.packed, additive, bn_mode == R2L_BN_NONE ? nullptr : ws.bn, out, nullptr, workspace,
                          workspace_bytes, B, H, W,
                          R2L_F_FOLDED_VALID | keep | (bn_mode == R2L_BN_TRAIN ? R2L_F_LUMA_VALID : 0), stream, nullptr, &ep);
}
// BatchNorm's backward sums of a step whose forward kept Y' (R2L_F_KEEP_LUMA on the row-streaming path): recomputed from the raw
// frame and Y' (r2l_bnr_planes_block) on batches of >= 6 Mi px where the plane passes run -- the forward's output is then
// not read at all by the backward.  Returns 1 when it did not run (the caller falls back to r2l_bn_bwd_reduce).
static int r2l_bn_bwd_reduce_planes(const R2LRaw& raw, const float* additive, const float* grad_out, const R2LWorkspace& ws,
                                    const R2LEpi& ep, float* bn_bwd, int B, int H, int W, int keep, void* stream) {
#ifdef R2L_SERIAL
  (void)raw; (void)additive; (void)grad_out; (void)ws; (void)ep; (void)bn_bwd; (void)B; (void)H; (void)W; (void)keep; (void)stream;
  return 1;
#else
  // (from 6 Mi px: at 64x256x256 = 4 Mi px the whole step, output included, lives in the memory-side cache and reading the output back
  //  is cheaper than recomputing it -- bn_reduce 22.4-23.8 us against 27.1; at 128x256x256 37.2 against 37.5, the step 2 % faster)
  const bool planes = r2l_env_int("R2L_BWD_PLANES", 0) || (size_t)B * H * W >= ((size_t)6 << 20);
  if (!keep || !r2l_fwd_streams(additive, W) || !planes || r2l_env_int("R2L_BNR_READ_OUT", 0) ||
      (r2l_env_int("R2L_EXP_NO_TREE", 0) & 2))
    return 1;
  R2LBnrArgs a;
  a.s.raw = raw;
  a.s.F = ws.folded;
  a.s.bn = ws.bn;
  a.s.out = nullptr;
  a.s.yp_out = nullptr;
  a.s.yp_in = ws.yp;
#ifdef R2L_EXP_STAMPS
  a.s.tl = nullptr;
#endif
  a.s.stat_partial = ws.part_small;
  a.s.B = B;
  a.s.H = H;
  a.s.W = W;
  // (band height as kernel B1's: 36 rows at 64x512x512 -- 61.4 us against 63.2 at 24 rows, 71-73 at 12 / 18 / 30, 68.4 at 48)
  a.s.band_h = r2l_band_rows(B, H, W, 256L * 4 * 2, "R2L_BNR_BAND");
  a.s.nband = (H + a.s.band_h - 1) / a.s.band_h;
  const long nstrip = (W + 255) / 256;
  const long items = (long)B * a.s.nband * nstrip;
  if (items > (1L << 30)) return r2l_fail(-1, "r2l_isp_step_bwd: batch too large");
  a.s.nitems = (int)items;
  a.s.tree = R2LTree{ws.part_small, nullptr, ws.gpartial, ws.counters, 12, 0};
  a.s.stats_out = nullptr;
  a.s.fin.bn = nullptr;
  a.s.ep = ep.on ? ep : R2LEpi{0, 0, 0, 0};
  a.s.xcdm = 0;
  a.gout = grad_out;
  a.sums = ws.bsums;
  a.totals = ws.moments;
  a.bn_bwd = bn_bwd;
  long g = (items + R2L_BNR_NWV - 1) / R2L_BNR_NWV;
  const long gcap = r2l_env_int("R2L_GRID_BNR", R2L_MAX_BLOCKS);
  if (g > gcap) g = gcap;
  if (g > R2L_MAX_BLOCKS) g = R2L_MAX_BLOCKS;
  return ep.on ? (raw.u16 ? r2l_launch_bnr_planes_epi_u16(a, (int)g, stream) : r2l_launch_bnr_planes_epi(a, (int)g, stream))
               : (raw.u16 ? r2l_launch_bnr_planes_u16(a, (int)g, stream) : r2l_launch_bnr_planes(a, (int)g, stream));
#endif
}

int r2l_isp_step_bwd(const void* raw, int raw_u16, float denom, const float* additive, const float* grad_out,
                     const float* out, float* grad_params, float* grad_additive, int bn_mode, void* workspace,
                     size_t workspace_bytes, int B, int H, int W, int nranks, int phase,
                     const double* gathered_sums, void* stream) {
  const int keep = (phase & R2L_STEP_KEEP_LUMA) ? R2L_F_KEEP_LUMA : 0;
  R2LEpi ep;
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (int e = r2l_epi_from_phase(phase, H, W, ep)) return e;
  phase &= ~(R2L_STEP_KEEP_LUMA | R2L_STEP_EPI_MASK);
  if (ep.on && grad_additive) return r2l_fail(-3, "r2l_isp_step_bwd: no output epilogue with an additive layer");
  if (!raw || !grad_out || !workspace) return r2l_fail(-1, "r2l_isp_step_bwd: null pointer");
  if (phase != R2L_STEP_ALL && phase != R2L_STEP_A && phase != R2L_STEP_B)
    return r2l_fail(-1, "r2l_isp_step_bwd: unknown phase");
  if (nranks < 1 || (nranks > 1 && phase == R2L_STEP_ALL && bn_mode == R2L_BN_TRAIN))
    return r2l_fail(-1, "r2l_isp_step_bwd: several ranks exchange the BatchNorm sums between phase A and phase B");
  if (phase != R2L_STEP_ALL && bn_mode != R2L_BN_TRAIN)
    return r2l_fail(-1, "r2l_isp_step_bwd: only train-mode BatchNorm has two phases");
  if (phase == R2L_STEP_B && !gathered_sums) return r2l_fail(-1, "r2l_isp_step_bwd: phase B needs the gathered sums");
  if (bn_mode == R2L_BN_TRAIN && !out) return r2l_fail(-1, "r2l_isp_step_bwd: train-mode BatchNorm needs the saved output");
  const R2LRaw rw = r2l_raw_any(raw, raw_u16, denom);
  // (before ANY launch: the recomputing BatchNorm sums below read the raw frames, r2l_isp_bwd_impl's own check comes later)
  if (int e = r2l_check_raw(rw, W, "r2l_isp_step_bwd")) return e;
  const R2LWorkspace ws = r2l_carve(workspace, B, H, W);
  if (workspace_bytes < ws.total) return r2l_fail(-2, "r2l_isp_step_bwd: workspace too small (r2l_isp_workspace_bytes)");
  const float* bn = bn_mode == R2L_BN_NONE ? nullptr : ws.bn;
  const float* bn_bwd = bn_mode == R2L_BN_TRAIN ? ws.bn_bwd : nullptr;
  if (bn_mode == R2L_BN_TRAIN && phase != R2L_STEP_B) {
    float* means = phase == R2L_STEP_ALL ? ws.bn_bwd : nullptr;
    int e = r2l_bn_bwd_reduce_planes(rw, additive, grad_out, ws, ep, means, B, H, W, keep, stream);
    if (e == 1) e = r2l_bn_bwd_reduce(grad_out, out, ws.moments, ws.bsums, means, workspace, workspace_bytes, B, H, W,
                                      R2L_F_FOLDED_VALID, stream);
    if (e) return e;
    if (phase == R2L_STEP_A) return 0;
  }
  if (phase == R2L_STEP_B) {
    R2LBnBwdMeansArgs m{gathered_sums, nranks, ws.moments + 6, ws.bn_bwd};
    if (int e = r2l_launch_bn_bwd_means(m, 1, stream)) return e;
  }
  if (grad_params) {
    if (int e = r2l_isp_bwd_impl(rw, ws.packed, additive, bn, bn_bwd, grad_out, grad_params, nullptr, workspace,
                                 workspace_bytes, B, H, W, R2L_F_FOLDED_VALID | keep, stream, &ep))
      return e;
  }
  if (grad_additive) return r2l_additive_bwd(grad_out, out, bn, bn_bwd, grad_additive, B, H, W, stream);
  return 0;
}

static int r2l_raw2rgb_fwd_impl(const R2LRaw& raw, const float* black_level, float* out, int B, int H, int W,
                                int reduce_size, int out_channels, void* stream) {
  if (B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return r2l_fail(-1, "raw2rgb: H and W must be even");
  if (out_channels != 3 && out_channels != 4) return r2l_fail(-1, "raw2rgb: out_channels in {3,4}");
  if (int e = r2l_check_raw(raw, 4, "raw2rgb")) return e;
  if (!out) return r2l_fail(-1, "raw2rgb: null pointer");
  R2LRaw2RgbArgs a{raw, black_level, out, nullptr, nullptr, nullptr, B, H, W, reduce_size, out_channels};
  const size_t nitems = (size_t)B * (H / 2) * ((W + 3) / 4);
  size_t grid = (nitems + R2L_NT - 1) / R2L_NT;
  if (grid > 4096) grid = 4096;
  return r2l_launch_raw2rgb_fwd(a, (int)grid, stream);
}

size_t r2l_raw2rgb_bwd_workspace_bytes(int B, int H, int W) {
  (void)B;
  (void)H;
  (void)W;
  return sizeof(float) * 4 * R2L_MAX_BLOCKS;
}

int r2l_raw2rgb_bwd(const float* grad_out, float* grad_raw, double* grad_black_level, void* workspace,
                    size_t workspace_bytes, int B, int H, int W, int reduce_size, int out_channels,
                    void* stream) {
  if (B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return r2l_fail(-1, "raw2rgb: H and W must be even");
  if (out_channels != 3 && out_channels != 4) return r2l_fail(-1, "raw2rgb: out_channels in {3,4}");
  if (!grad_out) return r2l_fail(-1, "raw2rgb_bwd: null pointer");
  if (grad_black_level && (!workspace || workspace_bytes < r2l_raw2rgb_bwd_workspace_bytes(B, H, W)))
    return r2l_fail(-2, "raw2rgb_bwd: workspace too small");
  R2LRaw2RgbArgs a{r2l_raw_f32(nullptr), nullptr, nullptr, grad_out, grad_raw,
                   grad_black_level ? (float*)workspace : nullptr, B, H, W, reduce_size, out_channels};
  const size_t nitems = (size_t)B * (H / 2) * ((W + 3) / 4);
  size_t grid = (nitems + R2L_NT - 1) / R2L_NT;
  if (grid > R2L_MAX_BLOCKS) grid = R2L_MAX_BLOCKS;
  if (int e = r2l_launch_raw2rgb_bwd(a, (int)grid, stream)) return e;
  if (grad_black_level) {
    R2LReduceRowsArgs r{(const float*)workspace, grad_black_level, (int)grid, 1.0, nullptr};
    return r2l_launch_reduce_rows(r, 4, stream);
  }
  return 0;
}

// chains the single-launch kernels cover: the short chain (any demosaic) and bilinear + sharpening_filter +
// gaussian_denoising; everything else runs as luma-plane passes and needs two float64 planes of workspace
// chains the row-streaming luma-chain kernel covers (r2l_static_chain.h): bilinear + [sharpening_filter] +
// [gaussian_denoising], frames up to 2048 columns, W % 4 == 0
static bool r2l_static_is_chain(int W, int debayer, int sharpening, int denoising, int median_size = 3) {
  if (debayer == R2L_DEBAYER_MENON2007) return false;
  if (denoising == R2L_DENOISE_MEDIAN && median_size != 3) return false;  // the 5x5 median runs as a luma-plane pass
#ifdef R2L_SERIAL
  (void)W; (void)debayer; (void)sharpening; (void)denoising;
  return false;  // (lane shifts and wave-level exchange: not expressible in the one-lane-at-a-time emulation)
#else
  if (r2l_env_int("R2L_STATIC_TILED", 0)) return false;
  (void)debayer;  // every demosaic, sharpening and denoising the library knows -- but for the chain without a luma stage
  // (behind unsharp_masking the chroma waits 7 rows for its luma: 28 KB of LDS per 256-column strip, 4 strips at most)
  return (W & 3) == 0 && W <= (sharpening == R2L_SHARPEN_UNSHARP ? 1024 : 2048) && denoising != R2L_DENOISE_FFT &&
         !(sharpening == R2L_SHARPEN_NONE && denoising == R2L_DENOISE_NONE);
#endif
}
static bool r2l_static_is_fused(int W, int debayer, int sharpening, int denoising, bool f64_frames = false, int median_size = 3) {
  if (debayer == R2L_DEBAYER_MENON2007) return false;  // always plane passes (r2l_static_menon.h)
  if (denoising == R2L_DENOISE_FFT) return false;
  if (denoising == R2L_DENOISE_MEDIAN && median_size != 3) return false;
  if (sharpening == R2L_SHARPEN_NONE && denoising == R2L_DENOISE_NONE) return true;
  if (r2l_static_is_chain(W, debayer, sharpening, denoising)) return true;
  if (f64_frames) return false;  // the tile kernel of the default chain stages float32 frames in LDS
  return debayer == R2L_DEBAYER_BILINEAR && sharpening == R2L_SHARPEN_FILTER && denoising == R2L_DENOISE_GAUSSIAN;
}
static void r2l_stream_shape(R2LStaticStreamArgs& sa, int B, int H, int W, int debayer) {
  // One wavefront per (image, 256-column strip, row band), dealt in that order, so the ~3,000 wavefronts resident
  // at any moment work on ADJACENT bands.  Short bands keep that active region of the frames and of the output
  // compact in HBM (a few tens of MiB instead of a slice of every image of the batch), which is worth far more
  // than the halo rows every band re-reads (they come from L2): same-buffer A/B on 256x1024x1024, bilinear,
  // 128 rows per band 862 us, 32 rows 830, 16 rows 800, 8 rows 735-756, 5 rows 754, 4 rows 778, 2 rows 1012.
  // Malvar (4 halo rows, 5-row window) is flat between 24 and 48 rows per band -- and, round 5, 6 % FASTER at 10 rows (two
  // full groups of its 5-step unrolled loop): 883 -> 826 us on 256x1024x1024, same buffers (profiles/r05_static_ab.txt;
  // 9 rows 835, 13 rows 848, 15 rows 836, 5 rows 867, 6 rows -- 4 wasted steps of 10 -- 1000): the same compactness, at
  // 40 % more fetched rows.  Rounds 1-4 had only swept 24 .. 48.
  // Round 2, other frame widths (profiles/r02_j_stream_bands.txt): 6-row bands are as good on 1024-wide frames
  // (0.711 vs 0.709 of the HBM peak) and better on 512- and 256-wide ones (0.718 vs 0.692, 0.724 vs 0.700).
  sa.nseg = (W + 255) / 256;
  const int rows = (debayer == R2L_DEBAYER_MALVAR2004) ? 10 : 6;
  long nband = (H + rows - 1) / rows;
  nband = r2l_env_int("R2L_STREAM_BANDS", (int)nband);
  if (nband > H / 2) nband = H / 2;
  if (nband < 1) nband = 1;
  sa.band_h = (int)((H + nband - 1) / nband);
  sa.nband = (H + sa.band_h - 1) / sa.band_h;
  (void)B;
}

// ---- fft_denoising: the two transforms.  Device build: rocFFT plans (real <-> Hermitian, float64, rows of length W),
// cached per (W, rows); the work buffer comes out of the caller's workspace like everything else.
struct R2LFftPlans {
#ifndef R2L_EMUL
  rocfft_plan fwd = nullptr, inv = nullptr;
#endif
  size_t work_bytes = 0;
};
#ifndef R2L_EMUL
// rocFFT is NOT a link-time dependency of this library (round 6): fft_denoising is the one stage of the path that is a library
// call, and an alternate the reference's defaults never take (train.py:100-101 offers it).  The library is opened the first time
// an fft_denoising chain is asked for; every other entry point works on a box without it.
#include <dlfcn.h>
struct R2LRocfft {
  decltype(&rocfft_setup) setup = nullptr;
  decltype(&rocfft_plan_create) plan_create = nullptr;
  decltype(&rocfft_plan_get_work_buffer_size) plan_get_work_buffer_size = nullptr;
  decltype(&rocfft_execution_info_create) execution_info_create = nullptr;
  decltype(&rocfft_execution_info_set_stream) execution_info_set_stream = nullptr;
  decltype(&rocfft_execution_info_set_work_buffer) execution_info_set_work_buffer = nullptr;
  decltype(&rocfft_execute) execute = nullptr;
  decltype(&rocfft_execution_info_destroy) execution_info_destroy = nullptr;
  bool ok = false;
};
static const R2LRocfft* r2l_rocfft() {
  static R2LRocfft api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    for (const char* name : {"librocfft.so.0", "librocfft.so", "/opt/rocm/lib/librocfft.so.0", "/opt/rocm/lib/librocfft.so"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return;
#define R2L_FFT_SYM(field, sym) api.field = (decltype(api.field))dlsym(h, #sym)
    R2L_FFT_SYM(setup, rocfft_setup);
    R2L_FFT_SYM(plan_create, rocfft_plan_create);
    R2L_FFT_SYM(plan_get_work_buffer_size, rocfft_plan_get_work_buffer_size);
    R2L_FFT_SYM(execution_info_create, rocfft_execution_info_create);
    R2L_FFT_SYM(execution_info_set_stream, rocfft_execution_info_set_stream);
    R2L_FFT_SYM(execution_info_set_work_buffer, rocfft_execution_info_set_work_buffer);
    R2L_FFT_SYM(execute, rocfft_execute);
    R2L_FFT_SYM(execution_info_destroy, rocfft_execution_info_destroy);
#undef R2L_FFT_SYM
    api.ok = api.setup && api.plan_create && api.plan_get_work_buffer_size && api.execution_info_create &&
             api.execution_info_set_stream && api.execution_info_set_work_buffer && api.execute && api.execution_info_destroy;
  });
  return api.ok ? &api : nullptr;
}
static int r2l_fft_plans(int W, size_t rows, R2LFftPlans& out) {
  const R2LRocfft* fft = r2l_rocfft();
  if (!fft) return r2l_fail(-10, "fft_denoising needs librocfft.so (ROCm), which could not be opened; every other chain works without it");
  static std::mutex mu;
  static std::map<std::pair<int, std::pair<int, size_t>>, R2LFftPlans> cache;  // (device, (W, rows))
  static bool setup = false;
  std::lock_guard<std::mutex> lk(mu);
  if (!setup) {
    if (fft->setup() != rocfft_status_success) return r2l_fail(-10, "rocfft_setup failed");
    setup = true;
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  const std::pair<int, std::pair<int, size_t>> key{dev, {W, rows}};
  auto it = cache.find(key);
  if (it == cache.end()) {
    // (plans live as long as the process: a caller on another thread may be executing one, so none is destroyed here)
    R2LFftPlans p;
    const size_t len = (size_t)W;
    if (fft->plan_create(&p.fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                           rocfft_precision_double, 1, &len, rows, nullptr) != rocfft_status_success ||
        fft->plan_create(&p.inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                           rocfft_precision_double, 1, &len, rows, nullptr) != rocfft_status_success)
      return r2l_fail(-10, "rocfft_plan_create failed");
    size_t w0 = 0, w1 = 0;
    fft->plan_get_work_buffer_size(p.fwd, &w0);
    fft->plan_get_work_buffer_size(p.inv, &w1);
    p.work_bytes = w0 > w1 ? w0 : w1;
    it = cache.emplace(key, p).first;
  }
  out = it->second;
  return 0;
}
static int r2l_fft_exec(rocfft_plan plan, void* in, void* outp, void* work, size_t work_bytes, void* stream) {
  const R2LRocfft* fft = r2l_rocfft();
  if (!fft) return r2l_fail(-10, "librocfft.so could not be opened");
  rocfft_execution_info info = nullptr;
  if (fft->execution_info_create(&info) != rocfft_status_success) return r2l_fail(-10, "rocfft_execution_info_create failed");
  rocfft_status st = fft->execution_info_set_stream(info, stream);
  if (st == rocfft_status_success && work_bytes) st = fft->execution_info_set_work_buffer(info, work, work_bytes);
  void* ins[1] = {in};
  void* outs[1] = {outp};
  if (st == rocfft_status_success) st = fft->execute(plan, ins, outs, info);
  fft->execution_info_destroy(info);
  return st == rocfft_status_success ? 0 : r2l_fail(-10, "rocfft_execute failed");
}
#endif
// workspace of the fft_denoising chain behind the two luma planes: linear RGB planes, spectrum, rocFFT work buffer
struct R2LFftLayout {
  size_t rgb_off, spec_off, work_off, total, rows;
  R2LFftPlans plans;
};
static int r2l_fft_layout(int B, int H, int W, R2LFftLayout& L) {
  const size_t px = (size_t)B * H * W;
  L.rows = (size_t)3 * B * H;
  L.rgb_off = r2l_align_up(2 * sizeof(double) * px);
  L.spec_off = L.rgb_off + r2l_align_up(3 * sizeof(double) * px);
  L.work_off = L.spec_off + r2l_align_up(2 * sizeof(double) * L.rows * (size_t)(W / 2 + 1));
#ifndef R2L_EMUL
  if (int e = r2l_fft_plans(W, L.rows, L.plans)) return e;
#endif
  L.total = L.work_off + r2l_align_up(L.plans.work_bytes);
  return 0;
}

// ---- Menon2007 chains (r2l_static_menon.h): the (B,3,H,W) float64 image + five (B,H,W) planes; behind them, for fft_denoising,
// the spectrum and rocFFT's work buffer
struct R2LMenonLayout {
  size_t spec_off, work_off, total, rows;
  R2LFftPlans plans;
};
static int r2l_menon_layout(int B, int H, int W, bool fft, R2LMenonLayout& L) {
  const size_t px = (size_t)B * H * W;
  L.rows = (size_t)3 * B * H;
  L.spec_off = r2l_align_up(8 * sizeof(double) * px);
  L.work_off = L.total = L.spec_off;
  if (fft) {
    L.work_off = L.spec_off + r2l_align_up(2 * sizeof(double) * L.rows * (size_t)(W / 2 + 1));
#ifndef R2L_EMUL
    if (int e = r2l_fft_plans(W, L.rows, L.plans)) return e;
#endif
    L.total = L.work_off + r2l_align_up(L.plans.work_bytes);
  }
  return 0;
}
static int r2l_static_menon_impl(const R2LStaticArgs& a, int B, int H, int W, int sharpening, int denoising,
                                 const R2LStaticOpts& opt, void* workspace, size_t workspace_bytes, void* stream) {
  if ((W & 3) || H < 4 || W < 4)
    return r2l_fail(-4, "r2l_static_fwd: menon2007 runs as plane passes, which need W % 4 == 0 (and frames of at least 4 x 4)");
  const bool fft = denoising == R2L_DENOISE_FFT;
  R2LMenonLayout L;
  if (int e = r2l_menon_layout(B, H, W, fft, L)) return e;
  if (!workspace || workspace_bytes < L.total)
    return r2l_fail(-2, "r2l_static_fwd: workspace too small (r2l_static_workspace_bytes)");
  const size_t px = (size_t)B * H * W;
  R2LMenonArgs ma;
  ma.s = a;
  ma.rgb = (double*)workspace;
  ma.gh = ma.rgb + 3 * px;
  ma.gv = ma.gh + px;
  ma.ch = ma.gv + px;
  ma.cv = ma.ch + px;
  ma.m = ma.cv + px;
  ma.luma = ma.gh;
  const int ops[2] = {sharpening == R2L_SHARPEN_FILTER ? 1 : (sharpening == R2L_SHARPEN_UNSHARP ? 4 : 0),
                      denoising == R2L_DENOISE_GAUSSIAN ? 2 : (denoising == R2L_DENOISE_MEDIAN ? (opt.median_kernel_size == 5 ? 5 : 3) : 0)};
  ma.want_luma = (ops[0] || ops[1]) ? 1 : 0;
  for (int i = 0; i < 9; ++i) ma.M1[i] = R2L_YUV_FROM_RGB[i];
  size_t g = (px + R2L_NT - 1) / R2L_NT;
  if (g > 16384) g = 16384;
  for (int st = 0; st <= 7; ++st) {
    ma.stage = st;
    if (int e = r2l_launch_static_menon(ma, (int)g, stream)) return e;
  }
  if (ma.want_luma) {
    double* cur = ma.gh;     // G_H / G_V are dead behind stage 1: the luma plane and its ping-pong partner
    double* other = ma.gv;
    for (int i = 0; i < 2; ++i) {
      if (!ops[i]) continue;
      R2LPlaneArgs pa;
      pa.src = cur;
      pa.dst = other;
      pa.B = B;
      pa.H = H;
      pa.W = W;
      pa.op = ops[i];
      for (int k = 0; k < 5; ++k) pa.gk[k] = a.gk[k];
      for (int k = 0; k < 5; ++k) pa.uk[k] = a.uk[k];
      pa.amount = a.amount;
      size_t gp = (px / 2 + R2L_NT - 1) / R2L_NT;
      if (gp > 16384) gp = 16384;
      if (int e = r2l_launch_plane_filter(pa, (int)gp, stream)) return e;
      double* t = cur;
      cur = other;
      other = t;
    }
    ma.stage = 8;
    ma.luma = cur;
    if (int e = r2l_launch_static_menon(ma, (int)g, stream)) return e;
  }
  if (fft) {
    const int cut0 = (int)(W * opt.fft_fraction), cut1 = (int)(W * (1 - opt.fft_fraction));
#ifdef R2L_EMUL
    r2l_fft_lowpass_rows_host(ma.rgb, L.rows, W, cut0, cut1);
#else
    double* spec = (double*)((char*)workspace + L.spec_off);
    void* work = (char*)workspace + L.work_off;
    if (int e = r2l_fft_exec(L.plans.fwd, ma.rgb, spec, work, L.plans.work_bytes, stream)) return e;
    R2LSpecMaskArgs sm{spec, L.rows, W, cut0, cut1};
    size_t gm = (L.rows * (size_t)(W / 2 + 1) + R2L_NT - 1) / R2L_NT;
    if (gm > 16384) gm = 16384;
    if (int e = r2l_launch_spec_mask(sm, (int)gm, stream)) return e;
    if (int e = r2l_fft_exec(L.plans.inv, spec, ma.rgb, work, L.plans.work_bytes, stream)) return e;
#endif
  }
  R2LStaticFinishArgs fa{a, ma.rgb};
  size_t gf = (px / 4 + R2L_NT - 1) / R2L_NT;
  if (gf > 16384) gf = 16384;
  return r2l_launch_static_finish(fa, (int)gf, stream);
}

static int r2l_static_fwd_impl(const R2LRaw& raw, float* out, int B, int H, int W, const double* camera_host,
                               int debayer, int sharpening, int denoising, double gamma, void* workspace,
                               size_t workspace_bytes, void* stream, const float* mean_std_host = nullptr,
                               const R2LStaticOpts& opt = R2LStaticOpts()) {
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (int e = r2l_check_raw(raw, W, "r2l_static_fwd")) return e;
  if (mean_std_host)
    for (int k = 0; k < 3; ++k)
      if (!(mean_std_host[3 + k] != 0.f)) return r2l_fail(-1, "r2l_static_fwd_norm: std must be non-zero");
  if (!out || !camera_host) return r2l_fail(-1, "r2l_static_fwd: null pointer");
  if (debayer != R2L_DEBAYER_BILINEAR && debayer != R2L_DEBAYER_MALVAR2004 && debayer != R2L_DEBAYER_MENON2007)
    return r2l_fail(-1, "r2l_static_fwd: unknown debayer");
  if (sharpening != R2L_SHARPEN_NONE && sharpening != R2L_SHARPEN_FILTER && sharpening != R2L_SHARPEN_UNSHARP)
    return r2l_fail(-1, "r2l_static_fwd: unknown sharpening");
  if (denoising != R2L_DENOISE_NONE && denoising != R2L_DENOISE_GAUSSIAN && denoising != R2L_DENOISE_MEDIAN &&
      denoising != R2L_DENOISE_FFT)
    return r2l_fail(-4, "r2l_static_fwd: denoising must be none, gaussian_denoising, median_denoising or fft_denoising");
  if (!(gamma > 0)) return r2l_fail(-1, "r2l_static_fwd: gamma must be > 0");
  if (const char* why = r2l_static_opts_problem(opt, sharpening, denoising)) return r2l_fail(-4, std::string("r2l_static_fwd: ") + why);
  R2LStaticArgs a;
  r2l_static_setup(a, raw, out, B, H, W, camera_host, debayer, sharpening, denoising, gamma, mean_std_host, opt);
  if (debayer == R2L_DEBAYER_MENON2007)
    return r2l_static_menon_impl(a, B, H, W, sharpening, denoising, opt, workspace, workspace_bytes, stream);
  const int ntiles = B * ((H + GStatic::TH - 1) / GStatic::TH) * ((W + GStatic::TW - 1) / GStatic::TW);
#ifndef R2L_SERIAL
  if (r2l_static_is_chain(W, debayer, sharpening, denoising, opt.median_kernel_size)) {
    R2LStaticChainArgs ca;
    ca.s = a;
    // Bands: every band re-computes 7 rows of halo, so tall bands are cheaper (256x1024x1024: 64 rows 1134 us,
    // 128 rows 1112, 256 rows 1086; profiles/r02_f_chain_bands.txt) -- as tall as leaves ~1024 workgroups (two
    // rounds of two per CU), but not below 64 rows
    long nband = (1024 + B - 1) / B;
    if (nband > H / 64) nband = H / 64;
    nband = r2l_env_int("R2L_CHAIN_BAND", 0) ? (H + r2l_env_int("R2L_CHAIN_BAND", 64) - 1) / r2l_env_int("R2L_CHAIN_BAND", 64) : nband;
    if (nband < 1) nband = 1;
    ca.band_h = (int)((H + nband - 1) / nband);
    ca.band_h += ca.band_h & 1;  // bands start on even rows
    ca.nband = (H + ca.band_h - 1) / ca.band_h;
    const long grid = (long)B * ca.nband;
    if (grid > (1L << 30)) return r2l_fail(-1, "r2l_static_fwd: batch too large");
    ca.nw = W <= 256 ? 1 : (W <= 512 ? 2 : (W <= 1024 ? 4 : 8));
    const int kind = raw.u16 ? 1 : (raw.f64 ? 2 : 0);
    const int deb = debayer == R2L_DEBAYER_MALVAR2004 ? 1 : 0, dn = denoising == R2L_DENOISE_MEDIAN ? 1 : 0;
    typedef int (*launch_t)(const R2LStaticChainArgs&, int, void*);
    const int sh = sharpening == R2L_SHARPEN_UNSHARP ? 1 : 0;
#define R2L_CHAIN_ROW(sfx)                                                                                          \
  {{{r2l_launch_static_chain##sfx, r2l_launch_static_chain_median##sfx},                                            \
    {r2l_launch_static_chain_unsharp##sfx, r2l_launch_static_chain_unsharp_median##sfx}},                           \
   {{r2l_launch_static_chain_malvar##sfx, r2l_launch_static_chain_malvar_median##sfx},                              \
    {r2l_launch_static_chain_malvar_unsharp##sfx, r2l_launch_static_chain_malvar_unsharp_median##sfx}}}
    static const launch_t table[3][2][2][2] = {R2L_CHAIN_ROW(), R2L_CHAIN_ROW(_u16), R2L_CHAIN_ROW(_f64)};
#undef R2L_CHAIN_ROW
    return table[kind][deb][sh][dn](ca, (int)grid, stream);
  }
#endif
  if (!r2l_static_is_fused(W, debayer, sharpening, denoising, raw.f64 != nullptr, opt.median_kernel_size)) {
    // luma-plane passes: raw -> Y | sharpen | denoise | raw + Y'' -> RGB
    if (W & 3) return r2l_fail(-4, "r2l_static_fwd: this chain runs as plane passes, which need W % 4 == 0");
    const size_t plane_bytes = sizeof(double) * (size_t)B * H * W;
    const bool fft = denoising == R2L_DENOISE_FFT;
    R2LFftLayout L;
    if (fft)
      if (int e = r2l_fft_layout(B, H, W, L)) return e;
    if (!workspace || workspace_bytes < (fft ? L.total : 2 * plane_bytes))
      return r2l_fail(-2, "r2l_static_fwd: workspace too small (r2l_static_workspace_bytes)");
    double* p0 = (double*)workspace;
    double* p1 = p0 + (size_t)B * H * W;
    R2LStaticStreamArgs sa;
    sa.s = a;
    r2l_stream_shape(sa, B, H, W, debayer);
    const long nitems = (long)B * sa.nseg * sa.nband;
    if (nitems > (1L << 30)) return r2l_fail(-1, "r2l_static_fwd: batch too large");
    sa.nitems = (int)nitems;
    const int wpb = R2L_STREAM_NT / 64;
    const int sgrid = (int)((nitems + wpb - 1) / wpb);
    auto stream_pass = [&](const R2LStaticStreamArgs& x) {
      if (raw.u16)
        return debayer == R2L_DEBAYER_MALVAR2004 ? r2l_launch_static_luma_malvar_u16(x, sgrid, stream)
                                                 : r2l_launch_static_luma_bilinear_u16(x, sgrid, stream);
      if (raw.f64)
        return debayer == R2L_DEBAYER_MALVAR2004 ? r2l_launch_static_luma_malvar_f64(x, sgrid, stream)
                                                 : r2l_launch_static_luma_bilinear_f64(x, sgrid, stream);
      return debayer == R2L_DEBAYER_MALVAR2004 ? r2l_launch_static_luma_malvar(x, sgrid, stream)
                                               : r2l_launch_static_luma_bilinear(x, sgrid, stream);
    };
    sa.luma_out = p0;
    sa.luma_in = nullptr;
    sa.lin_out = nullptr;
    if (int e = stream_pass(sa)) return e;
    double* cur = p0;
    double* other = p1;
    const int ops[2] = {sharpening == R2L_SHARPEN_FILTER ? 1 : (sharpening == R2L_SHARPEN_UNSHARP ? 4 : 0),
                        denoising == R2L_DENOISE_GAUSSIAN ? 2 : (denoising == R2L_DENOISE_MEDIAN ? (opt.median_kernel_size == 5 ? 5 : 3) : 0)};
    for (int i = 0; i < 2; ++i) {
      if (!ops[i]) continue;
      R2LPlaneArgs pa;
      pa.src = cur;
      pa.dst = other;
      pa.B = B;
      pa.H = H;
      pa.W = W;
      pa.op = ops[i];
      for (int k = 0; k < 5; ++k) pa.gk[k] = a.gk[k];
      for (int k = 0; k < 5; ++k) pa.uk[k] = a.uk[k];
      pa.amount = a.amount;
      size_t g = ((size_t)B * H * W / 2 + R2L_NT - 1) / R2L_NT;
      if (g > 16384) g = 16384;
      if (int e = r2l_launch_plane_filter(pa, (int)g, stream)) return e;
      double* t = cur;
      cur = other;
      other = t;
    }
    sa.luma_out = nullptr;
    sa.luma_in = cur;
    if (!fft) return stream_pass(sa);
    // fft_denoising: sharpened RGB as float64 planes -> low-pass along the columns -> clip, gamma
    double* rgb = (double*)((char*)workspace + L.rgb_off);
    double* spec = (double*)((char*)workspace + L.spec_off);
    sa.lin_out = rgb;
    if (int e = stream_pass(sa)) return e;
    // int(c * keep_fraction), int(c * (1 - keep_fraction)) (pipeline_numpy.py:229-230; default 0.3)
    const int cut0 = (int)(W * opt.fft_fraction), cut1 = (int)(W * (1 - opt.fft_fraction));
#ifdef R2L_EMUL
    (void)spec;
    r2l_fft_lowpass_rows_host(rgb, L.rows, W, cut0, cut1);
#else
    void* work = (char*)workspace + L.work_off;
    if (int e = r2l_fft_exec(L.plans.fwd, rgb, spec, work, L.plans.work_bytes, stream)) return e;
    R2LSpecMaskArgs ma{spec, L.rows, W, cut0, cut1};
    size_t gm = (L.rows * (size_t)(W / 2 + 1) + R2L_NT - 1) / R2L_NT;
    if (gm > 16384) gm = 16384;
    if (int e = r2l_launch_spec_mask(ma, (int)gm, stream)) return e;
    if (int e = r2l_fft_exec(L.plans.inv, spec, rgb, work, L.plans.work_bytes, stream)) return e;
#endif
    R2LStaticFinishArgs fa{a, rgb};
    size_t gf = ((size_t)B * H * W / 4 + R2L_NT - 1) / R2L_NT;
    if (gf > 16384) gf = 16384;
    return r2l_launch_static_finish(fa, (int)gf, stream);
  }
  if (a.full) {
    const int grid = r2l_tile_grid(ntiles, r2l_env_int("R2L_GRID_STATIC_FULL", 256));
    return r2l_launch_static_full(a, grid, stream);
  }
  if ((W & 3) == 0 && (raw.f64 || !r2l_env_int("R2L_STATIC_TILED", 0))) {
    R2LStaticStreamArgs sa;
    sa.s = a;
    sa.luma_out = nullptr;
    sa.luma_in = nullptr;
    r2l_stream_shape(sa, B, H, W, debayer);
    const long nitems = (long)B * sa.nseg * sa.nband;
    if (nitems > (1L << 30)) return r2l_fail(-1, "r2l_static_fwd: batch too large");
    sa.nitems = (int)nitems;
    const int wpb = R2L_STREAM_NT / 64;
    const int grid = (int)((nitems + wpb - 1) / wpb);
    if (raw.u16)
      return debayer == R2L_DEBAYER_MALVAR2004 ? r2l_launch_static_stream_malvar_u16(sa, grid, stream)
                                               : r2l_launch_static_stream_bilinear_u16(sa, grid, stream);
    if (raw.f64)
      return debayer == R2L_DEBAYER_MALVAR2004 ? r2l_launch_static_stream_malvar_f64(sa, grid, stream)
                                               : r2l_launch_static_stream_bilinear_f64(sa, grid, stream);
    return debayer == R2L_DEBAYER_MALVAR2004 ? r2l_launch_static_stream_malvar(sa, grid, stream)
                                             : r2l_launch_static_stream_bilinear(sa, grid, stream);
  }
  const int grid = r2l_tile_grid(ntiles, r2l_env_int("R2L_GRID_STATIC", 1024));
  return r2l_launch_static_short(a, grid, stream);
}

int r2l_isp_fwd(const float* raw, const float* params, const float* additive,
                const float* bn_mean_istd, float* out, double* stats, void* workspace,
                size_t workspace_bytes, int B, int H, int W, int flags, void* stream) {
  return r2l_isp_fwd_impl(r2l_raw_f32(raw), params, additive, bn_mean_istd, out, stats, workspace, workspace_bytes,
                          B, H, W, flags & ~R2L_F_INTERNAL, stream);
}
// statistics pass + BatchNorm bookkeeping in one launch (one rank: no exchange between the two)
static int r2l_isp_fwd_stats_bn_impl(const R2LRaw& raw, const float* params, const float* additive, double* stats,
                                     float* bn_mean_istd, double* moments, float* running_mean, float* running_var,
                                     long long* num_batches_tracked, double eps, double momentum, void* workspace,
                                     size_t workspace_bytes, int B, int H, int W, void* stream) {
  if (!stats || !bn_mean_istd) return r2l_fail(-1, "r2l_isp_fwd_stats_bn: null pointer");
  if ((running_mean == nullptr) != (running_var == nullptr))
    return r2l_fail(-1, "r2l_isp_fwd_stats_bn: running_mean and running_var go together");
  R2LBnFinalizeArgs f{stats, 1, bn_mean_istd, moments, running_mean, running_var, eps, momentum, num_batches_tracked};
  return r2l_isp_fwd_impl(raw, params, additive, nullptr, nullptr, stats, workspace, workspace_bytes, B, H, W,
                          R2L_F_STATS_ONLY, stream, &f);
}
int r2l_isp_fwd_stats_bn(const float* raw, const float* params, const float* additive, double* stats,
                         float* bn_mean_istd, double* moments, float* running_mean, float* running_var,
                         long long* num_batches_tracked, double eps, double momentum, void* workspace,
                         size_t workspace_bytes, int B, int H, int W, void* stream) {
  return r2l_isp_fwd_stats_bn_impl(r2l_raw_f32(raw), params, additive, stats, bn_mean_istd, moments, running_mean,
                                   running_var, num_batches_tracked, eps, momentum, workspace, workspace_bytes, B,
                                   H, W, stream);
}
int r2l_isp_fwd_stats_bn_u16(const unsigned short* raw, float denom, const float* params, const float* additive,
                             double* stats, float* bn_mean_istd, double* moments, float* running_mean,
                             float* running_var, long long* num_batches_tracked, double eps, double momentum,
                             void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream) {
  return r2l_isp_fwd_stats_bn_impl(r2l_raw_u16(raw, denom), params, additive, stats, bn_mean_istd, moments,
                                   running_mean, running_var, num_batches_tracked, eps, momentum, workspace,
                                   workspace_bytes, B, H, W, stream);
}
int r2l_isp_fwd_u16(const unsigned short* raw, float denom, const float* params, const float* additive,
                    const float* bn_mean_istd, float* out, double* stats, void* workspace,
                    size_t workspace_bytes, int B, int H, int W, int flags, void* stream) {
  return r2l_isp_fwd_impl(r2l_raw_u16(raw, denom), params, additive, bn_mean_istd, out, stats, workspace,
                          workspace_bytes, B, H, W, flags & ~R2L_F_INTERNAL, stream);
}
int r2l_isp_bwd(const float* raw, const float* params, const float* additive,
                const float* bn_mean_istd, const float* bn_bwd, const float* grad_out,
                float* grad_params, float* grad_raw, void* workspace, size_t workspace_bytes, int B,
                int H, int W, int flags, void* stream) {
  return r2l_isp_bwd_impl(r2l_raw_f32(raw), params, additive, bn_mean_istd, bn_bwd, grad_out, grad_params, grad_raw,
                          workspace, workspace_bytes, B, H, W, flags, stream);
}
int r2l_isp_bwd_u16(const unsigned short* raw, float denom, const float* params, const float* additive,
                    const float* bn_mean_istd, const float* bn_bwd, const float* grad_out,
                    float* grad_params, void* workspace, size_t workspace_bytes, int B, int H, int W, int flags,
                    void* stream) {
  return r2l_isp_bwd_impl(r2l_raw_u16(raw, denom), params, additive, bn_mean_istd, bn_bwd, grad_out, grad_params,
                          nullptr, workspace, workspace_bytes, B, H, W, flags, stream);
}
int r2l_raw2rgb_fwd(const float* raw, const float* black_level, float* out, int B, int H, int W,
                    int reduce_size, int out_channels, void* stream) {
  return r2l_raw2rgb_fwd_impl(r2l_raw_f32(raw), black_level, out, B, H, W, reduce_size, out_channels, stream);
}
int r2l_raw2rgb_fwd_u16(const unsigned short* raw, float denom, const float* black_level, float* out, int B,
                        int H, int W, int reduce_size, int out_channels, void* stream) {
  return r2l_raw2rgb_fwd_impl(r2l_raw_u16(raw, denom), black_level, out, B, H, W, reduce_size, out_channels,
                              stream);
}
static size_t r2l_static_ws(int B, int H, int W, int debayer, int sharpening, int denoising, bool f64, int median_size = 3) {
  if (B < 1 || H < 1 || W < 1) return 0;
  if (debayer == R2L_DEBAYER_MENON2007) {
    R2LMenonLayout L;
    if (r2l_menon_layout(B, H, W, denoising == R2L_DENOISE_FFT, L)) return 0;
    return L.total;
  }
  if (r2l_static_is_fused(W, debayer, sharpening, denoising, f64, median_size)) return 0;
  if (denoising == R2L_DENOISE_FFT) {
    R2LFftLayout L;
    if (r2l_fft_layout(B, H, W, L)) return 0;
    return L.total;
  }
  return 2 * sizeof(double) * (size_t)B * H * W;
}
size_t r2l_static_workspace_bytes(int B, int H, int W, int debayer, int sharpening, int denoising) {
  return r2l_static_ws(B, H, W, debayer, sharpening, denoising, false);
}
size_t r2l_static_workspace_bytes_f64(int B, int H, int W, int debayer, int sharpening, int denoising) {
  return r2l_static_ws(B, H, W, debayer, sharpening, denoising, true);
}
size_t r2l_static_workspace_bytes_opts(int frames, int B, int H, int W, int debayer, int sharpening, int denoising,
                                       const double* options_host) {
  int med = 3;
  if (options_host) {
    const double m = options_host[R2L_SOPT_MEDIAN_SIZE];
    med = (m == (double)(int)m) ? (int)m : 3;
  }
  return r2l_static_ws(B, H, W, debayer, sharpening, denoising, frames == R2L_FRAMES_F64, med);
}
int r2l_static_fwd_f64(const double* raw, float* out, int B, int H, int W, const double* camera_host,
                       int debayer, int sharpening, int denoising, double gamma, void* workspace,
                       size_t workspace_bytes, void* stream) {
  return r2l_static_fwd_impl(r2l_raw_f64(raw), out, B, H, W, camera_host, debayer, sharpening, denoising, gamma,
                             workspace, workspace_bytes, stream);
}
int r2l_static_fwd(const float* raw, float* out, int B, int H, int W, const double* camera_host,
                   int debayer, int sharpening, int denoising, double gamma, void* workspace,
                   size_t workspace_bytes, void* stream) {
  return r2l_static_fwd_impl(r2l_raw_f32(raw), out, B, H, W, camera_host, debayer, sharpening, denoising, gamma,
                             workspace, workspace_bytes, stream);
}
int r2l_static_fwd_u16(const unsigned short* raw, float denom, float* out, int B, int H, int W,
                       const double* camera_host, int debayer, int sharpening, int denoising, double gamma,
                       void* workspace, size_t workspace_bytes, void* stream) {
  return r2l_static_fwd_impl(r2l_raw_u16(raw, denom), out, B, H, W, camera_host, debayer, sharpening, denoising,
                             gamma, workspace, workspace_bytes, stream);
}
static int r2l_static_fwd_any(const void* raw, int frames, float denom, float* out, int B, int H, int W,
                              const double* camera_host, int debayer, int sharpening, int denoising, double gamma,
                              const float* mean_std_host, const R2LStaticOpts& opt, void* workspace, size_t workspace_bytes,
                              void* stream);
int r2l_static_fwd_norm(const void* raw, int frames, float denom, float* out, int B, int H, int W,
                        const double* camera_host, int debayer, int sharpening, int denoising, double gamma,
                        const float* mean_std_host, void* workspace, size_t workspace_bytes, void* stream) {
  return r2l_static_fwd_any(raw, frames, denom, out, B, H, W, camera_host, debayer, sharpening, denoising, gamma, mean_std_host,
                            R2LStaticOpts(), workspace, workspace_bytes, stream);
}
int r2l_static_fwd_opts(const void* raw, int frames, float denom, float* out, int B, int H, int W,
                        const double* camera_host, int debayer, int sharpening, int denoising, double gamma,
                        const double* options_host, const float* mean_std_host, void* workspace, size_t workspace_bytes,
                        void* stream) {
  R2LStaticOpts o;
  if (options_host) {
    o.sharp_radius = options_host[R2L_SOPT_SHARP_RADIUS];
    o.sharp_amount = options_host[R2L_SOPT_SHARP_AMOUNT];
    o.gaussian_sigma = options_host[R2L_SOPT_GAUSSIAN_SIGMA];
    o.fft_fraction = options_host[R2L_SOPT_FFT_FRACTION];
    const double m = options_host[R2L_SOPT_MEDIAN_SIZE];
    o.median_kernel_size = (m == (double)(int)m) ? (int)m : -1;
  }
  return r2l_static_fwd_any(raw, frames, denom, out, B, H, W, camera_host, debayer, sharpening, denoising, gamma, mean_std_host,
                            o, workspace, workspace_bytes, stream);
}
static int r2l_static_fwd_any(const void* raw, int frames, float denom, float* out, int B, int H, int W,
                              const double* camera_host, int debayer, int sharpening, int denoising, double gamma,
                              const float* mean_std_host, const R2LStaticOpts& opt, void* workspace, size_t workspace_bytes,
                              void* stream) {
  R2LRaw rw;
  if (frames == R2L_FRAMES_F32)
    rw = r2l_raw_f32((const float*)raw);
  else if (frames == R2L_FRAMES_U16)
    rw = r2l_raw_u16((const unsigned short*)raw, denom);
  else if (frames == R2L_FRAMES_F64)
    rw = r2l_raw_f64((const double*)raw);
  else
    return r2l_fail(-1, "r2l_static_fwd_norm: frames must be R2L_FRAMES_F32, _U16 or _F64");
  return r2l_static_fwd_impl(rw, out, B, H, W, camera_host, debayer, sharpening, denoising, gamma, workspace,
                             workspace_bytes, stream, mean_std_host, opt);
}

// ---- staged (track_stages=True) entry points -------------------------------------------------------
size_t r2l_stage_workspace_bytes(void) { return sizeof(float) * 81 * R2L_MAX_BLOCKS + 256; }

static int r2l_stage_grid(int B, int H, int W, int per_thread) {
  const size_t n = (size_t)B * H * W / per_thread;
  size_t g = (n + R2L_NT - 1) / R2L_NT;
  if (g > R2L_MAX_BLOCKS) g = R2L_MAX_BLOCKS;
  return g < 1 ? 1 : (int)g;
}
static int r2l_stage_finish(const float* partial, int nslots, int grid, float* out, void* stream) {
  R2LReduceRowsArgs r{partial, nullptr, grid, 1.0, out};
  return r2l_launch_reduce_rows(r, nslots, stream);
}

int r2l_stage_conv33_fwd(const float* x, const float* w, float* y, int B, int H, int W, void* stream) {
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (!x || !w || !y) return r2l_fail(-1, "r2l_stage_conv33_fwd: null pointer");
  R2LStageArgs a{x, nullptr, w, nullptr, y, nullptr, B, H, W, 3, 1};
  return r2l_launch_conv33_fwd(a, r2l_stage_grid(B, H, W, 1), stream);
}
int r2l_stage_conv33_bwd(const float* x, const float* w, const float* g, float* gx, float* gw, void* workspace,
                         size_t workspace_bytes, int B, int H, int W, void* stream) {
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (!x || !w || !g || !gw || !workspace) return r2l_fail(-1, "r2l_stage_conv33_bwd: null pointer");
  if (workspace_bytes < r2l_stage_workspace_bytes()) return r2l_fail(-2, "r2l_stage: workspace too small");
  const int grid = r2l_stage_grid(B, H, W, 1);
  R2LStageArgs a{x, g, w, nullptr, gx, (float*)workspace, B, H, W, 3, 1};
  if (int e = r2l_launch_conv33_bwd(a, grid, stream)) return e;
  return r2l_stage_finish((const float*)workspace, 81, grid, gw, stream);
}
int r2l_stage_mix3_fwd(const float* x, const float* m, float* y, int B, int H, int W, void* stream) {
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (!x || !m || !y) return r2l_fail(-1, "r2l_stage_mix3_fwd: null pointer");
  R2LStageArgs a{x, nullptr, m, nullptr, y, nullptr, B, H, W, 0, 0};
  return r2l_launch_mix3_fwd(a, r2l_stage_grid(B, H, W, 1), stream);
}
int r2l_stage_mix3_bwd(const float* x, const float* m, const float* g, float* gx, float* gm, void* workspace,
                       size_t workspace_bytes, int B, int H, int W, void* stream) {
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (!m || !g || !workspace) return r2l_fail(-1, "r2l_stage_mix3_bwd: null pointer");
  if (gm && !x) return r2l_fail(-1, "r2l_stage_mix3_bwd: the matrix gradient needs the forward input");
  if (workspace_bytes < r2l_stage_workspace_bytes()) return r2l_fail(-2, "r2l_stage: workspace too small");
  const int grid = r2l_stage_grid(B, H, W, 1);
  R2LStageArgs a{gm ? x : nullptr, g, m, nullptr, gx, (float*)workspace, B, H, W, 0, 0};
  if (int e = r2l_launch_mix3_bwd(a, grid, stream)) return e;
  return gm ? r2l_stage_finish((const float*)workspace, 9, grid, gm, stream) : 0;
}
int r2l_stage_pconv_fwd(const float* x, const float* k, float* y, int K, int mirror, int B, int H, int W,
                        void* stream) {
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (!x || !k || !y || (K != 3 && K != 5)) return r2l_fail(-1, "r2l_stage_pconv_fwd: bad argument");
  R2LStageArgs a{x, nullptr, k, nullptr, y, nullptr, B, H, W, K, mirror};
  return r2l_launch_pconv_fwd(a, r2l_stage_grid(B, H, W, 1), stream);
}
int r2l_stage_pconv_bwd(const float* x, const float* k, const float* g, float* gx, float* gk25, int K,
                        int mirror, void* workspace, size_t workspace_bytes, int B, int H, int W,
                        void* stream) {
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (!x || !k || !g || !gk25 || !workspace || (K != 3 && K != 5))
    return r2l_fail(-1, "r2l_stage_pconv_bwd: bad argument");
  if (workspace_bytes < r2l_stage_workspace_bytes()) return r2l_fail(-2, "r2l_stage: workspace too small");
  const int grid = r2l_stage_grid(B, H, W, 1);
  R2LStageArgs a{x, g, k, nullptr, gx, (float*)workspace, B, H, W, K, mirror};
  if (int e = r2l_launch_pconv_bwd(a, grid, stream)) return e;
  return r2l_stage_finish((const float*)workspace, 25, grid, gk25, stream);
}
int r2l_stage_point(int op, const float* x, const float* g, const float* w, const float* aux,
                    const float* aux2, float* y, float* sums6, void* workspace, size_t workspace_bytes,
                    int B, int H, int W, void* stream) {
  if (int e = r2l_check_dims(B, H, W)) return e;
  if (op < 0 || op > 8) return r2l_fail(-1, "r2l_stage_point: unknown op");
  const bool reduces = (op == 3 || op == 7);
  if (reduces && (!sums6 || !workspace || workspace_bytes < r2l_stage_workspace_bytes()))
    return r2l_fail(-2, "r2l_stage_point: reduction needs sums + workspace");
  const int grid = r2l_stage_grid(B, H, W * 3, 4);
  R2LPointArgs a{x, g, w, aux, aux2, y, reduces ? (float*)workspace : nullptr, B, H, W, op};
  if (int e = r2l_launch_point(a, grid, stream)) return e;
  return reduces ? r2l_stage_finish((const float*)workspace, 6, grid, sums6, stream) : 0;
}

// ---- augmentation after the ISP (utils/augmentation.py) -------------------------------------------------
int r2l_augment(const float* x, float* y, int N, int H, int W, int hflip, int vflip, int k, int inverse,
                void* stream) {
  if (!x || !y || N < 1 || H < 1 || W < 1) return r2l_fail(-1, "r2l_augment: null pointer / bad dimensions");
  if ((H & 3) == 0 && (W & 3) == 0 && (hflip || vflip || (k & 3))) {
    // 16 bytes at a time, transposes through LDS tiles (r2l_aug_tiled_block)
    const int Wo = (k & 1) ? H : W;
    int r0, c0, r1, c1, r2, c2;
    r2l_aug_map(H, W, hflip != 0, vflip != 0, k & 3, 0, 0, r0, c0);
    r2l_aug_map(H, W, hflip != 0, vflip != 0, k & 3, 1, 0, r1, c1);
    r2l_aug_map(H, W, hflip != 0, vflip != 0, k & 3, 0, 1, r2, c2);
    R2LAugTiledArgs t;
    t.x = x;
    t.y = y;
    t.N = N;
    t.H = H;
    t.W = W;
    t.s0 = r0 * Wo + c0;
    t.sr = (r1 * Wo + c1) - t.s0;
    t.sc = (r2 * Wo + c2) - t.s0;
    t.odd = k & 1;
    t.inverse = inverse != 0;
    t.ntr = (H + R2L_AUG_TS - 1) / R2L_AUG_TS;
    t.ntc = (W + R2L_AUG_TS - 1) / R2L_AUG_TS;
    size_t nt = (size_t)N * t.ntr * t.ntc;
    if (nt > (size_t)1 << 30) return r2l_fail(-1, "r2l_augment: batch too large");
    return r2l_launch_aug_tiled(t, (int)(nt < 4096 ? nt : 4096), stream);
  }
  R2LAugArgs a{x, y, N, H, W, hflip != 0, vflip != 0, k & 3, inverse != 0};
  size_t g = ((size_t)N * H * W + R2L_NT - 1) / R2L_NT;
  if (g > 8192) g = 8192;
  return r2l_launch_aug(a, (int)g, stream);
}
int r2l_add_noise(const float* x, const float* noise, float std, float* y, size_t n, void* stream) {
  if (!x || !noise || !y || n == 0) return r2l_fail(-1, "r2l_add_noise: null pointer / empty");
  R2LAxpyArgs a{x, noise, y, std, n};
  size_t g = (n + R2L_NT - 1) / R2L_NT;
  if (g > 8192) g = 8192;
  return r2l_launch_axpy(a, (int)g, stream);
}
int r2l_add_noise_philox(const float* x, float* y, float std, unsigned long long seed, unsigned long long offset,
                         size_t n, void* stream) {
  if (!x || !y || n == 0) return r2l_fail(-1, "r2l_add_noise_philox: null pointer / empty");
  R2LPhiloxArgs a{x, y, std, seed, offset, n};
  size_t g = ((n + 3) / 4 + R2L_NT - 1) / R2L_NT;
  if (g > 8192) g = 8192;
  return r2l_launch_philox_noise(a, (int)g, stream);
}

// ---- adversarial auxiliary losses (utils/ssim.py, utils/base.py:342-358) -------------------------------
static void r2l_ssim_gauss(float* g) {  // utils/ssim.py:9-11, float32 like torch.Tensor([...]) / sum
  float w[R2L_SSIM_K], sum = 0.f;
  for (int x = 0; x < R2L_SSIM_K; ++x) {
    const double d = x - R2L_SSIM_K / 2;
    w[x] = (float)exp(-(d * d) / (2.0 * 1.5 * 1.5));
    sum += w[x];
  }
  for (int x = 0; x < R2L_SSIM_K; ++x) g[x] = w[x] / sum;
}
size_t r2l_aux_workspace_bytes(int B, int C, int H, int W) {
  if (B < 1 || C < 1 || H < 1 || W < 1) return 0;
  return r2l_align_up(sizeof(float) * R2L_MAX_BLOCKS) + sizeof(float) * 3 * (size_t)B * C * H * W;
}
static int r2l_aux_check(const void* a, const void* b, int B, int C, int H, int W, const char* who) {
  if (!a || !b) return r2l_fail(-1, std::string(who) + ": null pointer");
  if (B < 1 || C < 1 || H < 1 || W < 1 || (size_t)B * C > (1u << 24) || (size_t)H * W > ((size_t)1 << 29))
    return r2l_fail(-1, std::string(who) + ": bad dimensions");
  return 0;
}
static int r2l_ssim_launch(const float* img1, const float* img2, float* partial, float* dmaps, int mode, int B,
                           int C, int H, int W, void* stream, int* grid_out) {
  R2LSsimArgs a;
  a.img1 = img1;
  a.img2 = img2;
  r2l_ssim_gauss(a.g);
  a.partial = partial;
  a.dmaps = dmaps;
  a.nplanes = B * C;
  a.H = H;
  a.W = W;
  a.mode = mode;
  const int ntiles = B * C * ((H + R2L_SSIM_T - 1) / R2L_SSIM_T) * ((W + R2L_SSIM_T - 1) / R2L_SSIM_T);
  *grid_out = ntiles < 512 ? ntiles : 512;
  return r2l_launch_ssim(a, *grid_out, stream);
}
int r2l_ssim_fwd(const float* img1, const float* img2, double* ssim_mean, void* workspace, size_t workspace_bytes,
                 int keep_for_backward, int B, int C, int H, int W, void* stream) {
  if (int e = r2l_aux_check(img1, img2, B, C, H, W, "r2l_ssim_fwd")) return e;
  if (!ssim_mean || !workspace || workspace_bytes < r2l_aux_workspace_bytes(B, C, H, W))
    return r2l_fail(-2, "r2l_ssim_fwd: workspace too small / null output");
  float* partial = (float*)workspace;
  float* dmaps = (float*)((char*)workspace + r2l_align_up(sizeof(float) * R2L_MAX_BLOCKS));
  int grid = 0;
  if (int e = r2l_ssim_launch(img1, img2, partial, dmaps, keep_for_backward ? 3 : 1, B, C, H, W, stream, &grid))
    return e;
  R2LReduceRowsArgs r{partial, ssim_mean, grid, 1.0 / ((double)B * C * H * W), nullptr};
  return r2l_launch_reduce_rows(r, 1, stream);
}
int r2l_ssim_bwd(const float* img1, const float* img2, const float* grad_ssim, float* grad_img2, void* workspace,
                 size_t workspace_bytes, int workspace_has_dmaps, int B, int C, int H, int W, void* stream) {
  if (int e = r2l_aux_check(img1, img2, B, C, H, W, "r2l_ssim_bwd")) return e;
  if (!grad_ssim || !grad_img2 || !workspace || workspace_bytes < r2l_aux_workspace_bytes(B, C, H, W))
    return r2l_fail(-2, "r2l_ssim_bwd: workspace too small / null pointer");
  float* dmaps = (float*)((char*)workspace + r2l_align_up(sizeof(float) * R2L_MAX_BLOCKS));
  int grid = 0;
  if (!workspace_has_dmaps) {
    if (int e = r2l_ssim_launch(img1, img2, nullptr, dmaps, 2, B, C, H, W, stream, &grid)) return e;
  } else {
    const int ntiles = B * C * ((H + R2L_SSIM_T - 1) / R2L_SSIM_T) * ((W + R2L_SSIM_T - 1) / R2L_SSIM_T);
    grid = ntiles < 512 ? ntiles : 512;
  }
  R2LSsimBwdArgs b;
  b.img1 = img1;
  b.img2 = img2;
  b.dmaps = dmaps;
  b.gup = grad_ssim;
  b.scale = (float)(1.0 / ((double)B * C * H * W));
  b.grad = grad_img2;
  r2l_ssim_gauss(b.g);
  b.nplanes = B * C;
  b.H = H;
  b.W = W;
  return r2l_launch_ssim_bwd(b, grid, stream);
}
int r2l_l2_fwd(const float* x, const float* y, double* sum, void* workspace, size_t workspace_bytes, size_t n,
               void* stream) {
  if (!x || !y || !sum || !workspace) return r2l_fail(-1, "r2l_l2_fwd: null pointer");
  if (n == 0 || (n & 3)) return r2l_fail(-1, "r2l_l2_fwd: the element count must be a positive multiple of 4");
  if (workspace_bytes < sizeof(float) * R2L_MAX_BLOCKS) return r2l_fail(-2, "r2l_l2_fwd: workspace too small");
  size_t g = (n / 4 + R2L_NT - 1) / R2L_NT;
  if (g > R2L_MAX_BLOCKS) g = R2L_MAX_BLOCKS;
  R2LL2Args a{x, y, nullptr, nullptr, (float*)workspace, n};
  if (int e = r2l_launch_l2(a, (int)g, stream)) return e;
  R2LReduceRowsArgs r{a.partial, sum, (int)g, 1.0, nullptr};
  return r2l_launch_reduce_rows(r, 1, stream);
}
int r2l_l2_bwd(const float* x, const float* y, const float* grad_sum, float* grad_y, size_t n, void* stream) {
  if (!x || !y || !grad_sum || !grad_y) return r2l_fail(-1, "r2l_l2_bwd: null pointer");
  if (n == 0 || (n & 3)) return r2l_fail(-1, "r2l_l2_bwd: the element count must be a positive multiple of 4");
  size_t g = (n / 4 + R2L_NT - 1) / R2L_NT;
  if (g > 4096) g = 4096;
  R2LL2Args a{x, y, grad_sum, grad_y, nullptr, n};
  return r2l_launch_l2(a, (int)g, stream);
}

#ifdef R2L_TEST_HOOKS
// diagnostic builds: where the per-phase cycle stamps of -DR2L_EXP_STAMPS builds land in the workspace (tests/stamps.py)
#if defined(R2L_EXP_STAMPS) && !defined(R2L_EMUL)
int r2l_test_tail_stamps(unsigned long long* out32) {  // (tests/tail_timeline.py)
  return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(r2l_tail_ts), sizeof(unsigned long long) * 32);
}
#endif
size_t r2l_test_debug_offset(int B, int H, int W) {
  const R2LWorkspace ws = r2l_carve((void*)0, B, H, W);
  return (size_t)((char*)ws.debug - (char*)0);
}
#ifdef R2L_EMUL
// the tile walk of the persistent kernels, replayed on the host: owner[tile] = workgroup id that visits it (or -1), and
// the number of visits per tile in visits[tile]; returns the largest number of tiles any workgroup takes
int r2l_test_walk(int B, int H, int W, int nblk, int asym, int* owner, int* visits) {
  const int ntx = (W + 63) / 64, nty = (H + 63) / 64, ntiles = B * ntx * nty;
  for (int i = 0; i < ntiles; ++i) {
    owner[i] = -1;
    visits[i] = 0;
  }
  int most = 0;
  for (int bid = 0; bid < nblk; ++bid) {
    R2LTileWalk w = r2l_walk_init(B, H, W, 64, 64, bid, nblk, asym);
    R2LTile t;
    int n = 0;
    while (r2l_walk_next(w, H, W, 64, 64, t)) {
      const int tile = (t.b * nty + t.oy / 64) * ntx + t.ox / 64;
      owner[tile] = bid;
      visits[tile] += 1;
      n += 1;
    }
    most = n > most ? n : most;
  }
  return most;
}
#endif
#endif
}  // extern "C"
