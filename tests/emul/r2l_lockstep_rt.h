// r2l_lockstep_rt.h -- LOCK-STEP HOST EMULATION of a gfx950 workgroup: TEST INFRASTRUCTURE ONLY (never part of the product;
// raw2logit_amd/_lib.py refuses any library whose r2l_is_device_build() is 0).
//
// The serial emulation (r2l_emul.cpp) runs a phase for tid = 0, 1, ... in a loop and therefore cannot run the kernels that
// talk between lanes -- the row-streaming forward, the passes over planes, the branch-free static loops: exactly the kernels
// behind the headline numbers, whose unconditional fetches from clamped addresses are where an off-by-one reads or writes out
// of bounds without changing a checked output (missing #2).  Here every lane of a workgroup is a host THREAD that
// runs the kernel's device form; the operations between lanes are rendezvous:
//   wave shifts / row shifts (update_dpp), readfirstlane, shuffles   -> exchange through a slot array + a WAVE barrier
//   s_barrier                                                        -> a WORKGROUP barrier (lanes that have returned leave it)
// Workgroups of a launch run one after the other (as in the serial emulation: arrival tickets and coherent loads stay plain
// memory operations).  LDS is a heap block of exactly the kernel's size, filled with signalling garbage (NaN), so that
// -fsanitize=address sees every access past it and nothing can rely on LDS being zero.  Global buffers are the caller's
// (numpy / torch CPU tensors: under LD_PRELOAD=libasan.so their malloc redzones make out-of-bounds lanes fault).
//
// Divergence: an exchange must be reached by all 64 lanes of a wavefront, like the DPP instruction it stands for, and a
// workgroup barrier by every lane that has not returned -- the kernels only call them in wave-uniform resp. workgroup-uniform
// control flow, which is also what the hardware requires.  A lane that never arrives shows up as a hang; the launcher's
// watchdog aborts after R2L_LS_TIMEOUT_S (default 120 s) with the kernel's name.
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

struct R2LLsDim {
  unsigned x, y, z;
};
static thread_local R2LLsDim threadIdx = {0, 0, 0};

namespace r2l_ls {

// a barrier whose participants can leave for good (a lane that returns from the kernel)
struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  int alive = 0, waiting = 0;
  unsigned gen = 0;
  void reset(int n) {
    alive = n;
    waiting = 0;
  }
  void wait(const char* what) {
    std::unique_lock<std::mutex> lk(m);
    const unsigned g = gen;
    if (++waiting >= alive) {
      waiting = 0;
      ++gen;
      cv.notify_all();
      return;
    }
    if (!cv.wait_for(lk, std::chrono::seconds(timeout_s()), [&] { return gen != g; })) {
      fprintf(stderr, "r2l lock-step emulation: a lane never reached a %s (divergent rendezvous?) -- aborting\n", what);
      abort();
    }
  }
  void leave() {
    std::unique_lock<std::mutex> lk(m);
    --alive;
    if (alive > 0 && waiting >= alive) {
      waiting = 0;
      ++gen;
      cv.notify_all();
    }
  }
  static int timeout_s() {
    static const int t = [] {
      const char* s = getenv("R2L_LS_TIMEOUT_S");
      return (s && atoi(s) > 0) ? atoi(s) : 120;
    }();
    return t;
  }
};

struct Group {  // the workgroup in flight
  int nt = 0;
  Barrier wg;
  std::vector<Barrier> wave;          // one per 64 lanes
  std::vector<uint64_t> slot[2];      // exchange slots, double-buffered by the parity of the lane's exchange count
  const void* kernarg = nullptr;
};
static Group* g_group = nullptr;              // (one launch at a time: the launcher holds a mutex)
static thread_local unsigned t_xcount = 0;    // exchanges this lane has taken part in (equal over a wavefront)

inline void wg_barrier() { g_group->wg.wait("workgroup barrier"); }
inline const void* kernarg() { return g_group->kernarg; }

// every lane deposits `mine`; lane l receives the deposit of lane src(l) of its wavefront, or `keep` if src(l) < 0
template <class T, class SRC>
inline T exchange(T mine, T keep, SRC&& src) {
  static_assert(sizeof(T) <= 8, "one 64-bit slot per lane");
  Group& g = *g_group;
  const unsigned tid = threadIdx.x, lane = tid & 63u, base = tid & ~63u;
  std::vector<uint64_t>& s = g.slot[t_xcount & 1u];
  uint64_t bits = 0;
  memcpy(&bits, &mine, sizeof(T));
  s[tid] = bits;
  g.wave[tid >> 6].wait("wave exchange (DPP / readfirstlane / shuffle)");
  const int from = src((int)lane);
  T out = keep;
  if (from >= 0) {
    unsigned f = base + (unsigned)from;
    if (f >= (unsigned)g.nt) f = tid;  // (a partial last wavefront: lanes that do not exist hold the lane's own value)
    const uint64_t b = s[f];
    memcpy(&out, &b, sizeof(T));
  }
  ++t_xcount;
  return out;
}
// update_dpp(old, src, ctrl): wave_shr:1 (0x138), wave_shl:1 (0x130), row_shr:1 (0x111), row_shl:1 (0x101); row / bank masks
// 0xf, bound_ctrl off -- a lane without a source keeps `old`
template <class T>
inline T dpp(T old, T src, int ctrl) {
  switch (ctrl) {
    case 0x138: return exchange(src, old, [](int l) { return l > 0 ? l - 1 : -1; });
    case 0x130: return exchange(src, old, [](int l) { return l < 63 ? l + 1 : -1; });
    case 0x111: return exchange(src, old, [](int l) { return (l & 15) > 0 ? l - 1 : -1; });
    case 0x101: return exchange(src, old, [](int l) { return (l & 15) < 15 ? l + 1 : -1; });
    default:
      fprintf(stderr, "r2l lock-step emulation: DPP control 0x%x is not modelled\n", ctrl);
      abort();
  }
}
template <class T>
inline T readfirstlane(T x) {
  return exchange(x, x, [](int) { return 0; });
}
template <class T>
inline T shfl_xor(T x, int mask) {
  return exchange(x, x, [mask](int l) { return l ^ mask; });
}

// Run `grid` workgroups of `nt` lanes one after the other: body(bid, lds) is the kernel's workgroup program, called by every
// lane thread with threadIdx.x set.  lds_floats: the kernel's LDS size (a fresh, poisoned heap block per launch).
template <class BODY>
inline void launch(const char* name, int grid, int nt, size_t lds_floats, const void* kernarg, BODY&& body) {
  static std::mutex one_launch;
  std::lock_guard<std::mutex> hold(one_launch);
  (void)name;
  Group g;
  g.nt = nt;
  g.wave = std::vector<Barrier>((size_t)(nt + 63) / 64);
  g.slot[0].assign((size_t)nt, 0);
  g.slot[1].assign((size_t)nt, 0);
  g.kernarg = kernarg;
  g_group = &g;
  // exactly the kernel's LDS (16-byte aligned like the device's), NaN-filled
  float* lds = nullptr;
  const size_t nl = lds_floats ? lds_floats : 4;
  if (posix_memalign((void**)&lds, 16, nl * sizeof(float)) != 0) abort();
  Barrier round;  // all lane threads, every workgroup: start / end of a workgroup
  round.reset(nt);
  auto lane_thread = [&](int tid) {
    threadIdx.x = (unsigned)tid;
    for (int bid = 0; bid < grid; ++bid) {
      if (tid == 0) {
        g.wg.reset(nt);
        for (size_t w = 0; w < g.wave.size(); ++w) {
          const int lanes = nt - (int)w * 64;
          g.wave[w].reset(lanes < 64 ? lanes : 64);
        }
        for (size_t i = 0; i < nl; ++i) {
          const uint32_t nan = 0x7fa00000u + (uint32_t)(i & 0xffff);  // (signalling NaNs: LDS holds garbage at launch)
          memcpy(&lds[i], &nan, 4);
        }
      }
      round.wait("start of a workgroup");
      t_xcount = 0;
      body(bid, lds);
      g.wave[(size_t)tid >> 6].leave();  // this lane has returned: the others' rendezvous no longer count it
      g.wg.leave();
      round.wait("end of a workgroup");
    }
  };
  std::vector<std::thread> th;
  th.reserve((size_t)nt);
  for (int t = 1; t < nt; ++t) th.emplace_back(lane_thread, t);
  lane_thread(0);
  for (auto& t : th) t.join();
  free(lds);
  g_group = nullptr;
}

}  // namespace r2l_ls

// ---- the device builtins the kernels' device forms use ------------------------------------------------------------------
#define __builtin_amdgcn_update_dpp(old, src, ctrl, row_mask, bank_mask, bound) r2l_ls::dpp((old), (src), (ctrl))
#define __builtin_amdgcn_readfirstlane(x) r2l_ls::readfirstlane(x)
#define __builtin_amdgcn_s_setprio(n) ((void)0)
#define __builtin_amdgcn_sched_barrier(n) ((void)0)
#define __builtin_amdgcn_s_memrealtime() 0ull
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_s_getreg(x) 0u
#ifndef __clang__
#define __builtin_assume(x) ((void)0)
#endif
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  return r2l_ls::shfl_xor(v, mask);
}
