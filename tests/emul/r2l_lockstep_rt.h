// r2l_lockstep_rt.h -- LOCK-STEP HOST EMULATION of a gfx950 workgroup: TEST INFRASTRUCTURE ONLY (never part of the product;
// raw2logit_amd/_lib.py refuses any library whose r2l_is_device_build() is 0).
//
// The serial emulation (r2l_emul.cpp) runs a phase for tid = 0, 1, ... in a loop and therefore cannot run the kernels that
// talk between lanes -- the row-streaming forward, the passes over planes, the branch-free static loops: exactly the kernels
// behind the headline numbers, whose unconditional fetches from clamped addresses are where an off-by-one reads or writes out
// of bounds without changing a checked output (missing #2).  Here every lane of a workgroup is a FIBER (its own
// stack, cooperatively scheduled on the calling thread) that runs the kernel's device form; the operations between lanes are
// rendezvous:
//   wave shifts / row shifts (update_dpp), readfirstlane, shuffles   -> exchange through a slot array + a WAVE barrier
//   s_barrier                                                        -> a WORKGROUP barrier (lanes that have returned leave it)
// A lane runs until its next rendezvous, then the next lane runs: one context switch per lane and rendezvous (~50 ns; a first
// version with one OS thread per lane and futex barriers spent 95 % of its time in the kernel).  Workgroups of a launch run one
// after the other (as in the serial emulation: arrival tickets and coherent loads stay plain memory operations).  LDS is a heap
// block of exactly the kernel's size, filled with signalling NaNs, so that -fsanitize=address sees every access past it and
// nothing can rely on LDS being zero.  Global buffers are the caller's (numpy / torch CPU tensors: under LD_PRELOAD=libasan.so
// their malloc redzones make out-of-bounds lanes fault).
//
// Divergence: an exchange must be reached by every lane of a wavefront that has not retired (R2L_LANE_RETIRES: EXEC off for
// good), like the DPP instruction it stands for, and a workgroup barrier by every lane that has not returned -- the kernels only
// call them in wave-uniform resp. workgroup-uniform control flow, which is also what the hardware requires.  A rendezvous that
// cannot complete (every remaining lane is waiting somewhere else) is detected at once and reported with the kernel's name.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#include <mutex>
#include <type_traits>
#include <vector>

#if defined(__SANITIZE_ADDRESS__)
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
#define R2L_LS_ASAN 1
#else
#define R2L_LS_ASAN 0
#endif

struct R2LLsDim {
  unsigned x, y, z;
};
static R2LLsDim threadIdx = {0, 0, 0};  // the lane that is running (one lane runs at a time)

// callee-saved registers + stack pointer: the whole context of a cooperative switch on x86-64 (System V)
extern "C" void r2l_ls_switch(void** save_sp, void* new_sp);
#if defined(__x86_64__)
asm(R"(
.text
.hidden r2l_ls_switch
.globl r2l_ls_switch
.type r2l_ls_switch,@function
r2l_ls_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size r2l_ls_switch, .-r2l_ls_switch
)");
#else
#error "the lock-step emulation's context switch is written for x86-64"
#endif

namespace r2l_ls {

struct Barrier {  // participants can leave for good (a lane that returns from the kernel / retires from its wavefront)
  int alive = 0, waiting = 0;
  unsigned gen = 0;
  void reset(int n) {
    alive = n;
    waiting = 0;
  }
};

struct Fiber {
  void* sp = nullptr;        // saved stack pointer while switched out
  char* stack = nullptr;     // [stack, stack + STACK): grows down
  void* fake = nullptr;      // AddressSanitizer's fake-stack handle while switched out
  bool done = true, retired = false;
  unsigned xcount = 0;       // exchanges this lane has taken part in (equal over a wavefront)
};
constexpr size_t STACK = (size_t)1 << 20;  // per lane (-O0 frames of the unrolled row loops are tens of KB)

struct Group {  // the workgroup in flight
  int nt = 0, cur = 0, running = 0;
  Barrier wg;
  std::vector<Barrier> wave;      // one per 64 lanes
  std::vector<uint64_t> slot[2];  // exchange slots, double-buffered by the parity of the lane's exchange count
  std::vector<Fiber> fib;
  const void* kernarg = nullptr;
  const char* name = "?";
  void* main_sp = nullptr;
  void* main_fake = nullptr;
  const void* main_bottom = nullptr;
  size_t main_size = 0;
  unsigned long stalled = 0;      // consecutive yields without any rendezvous completing or lane finishing
  void (*entry)(void*) = nullptr;
  void* entry_arg = nullptr;
};
static Group* g_group = nullptr;  // (one launch at a time: the launcher holds a mutex)

inline const void* kernarg() { return g_group->kernarg; }

inline void switch_to(Group& g, int from, int to) {  // from / to: lane index, or -1 = the launcher's own context
  void** save = from >= 0 ? &g.fib[(size_t)from].sp : &g.main_sp;
  void* next = to >= 0 ? g.fib[(size_t)to].sp : g.main_sp;
#if R2L_LS_ASAN
  void** fake = from >= 0 ? &g.fib[(size_t)from].fake : &g.main_fake;
  const bool dying = from >= 0 && g.fib[(size_t)from].done;
  if (to >= 0)
    __sanitizer_start_switch_fiber(dying ? nullptr : fake, g.fib[(size_t)to].stack, STACK);
  else
    __sanitizer_start_switch_fiber(dying ? nullptr : fake, g.main_bottom, g.main_size);
#endif
  g.cur = to;
  if (to >= 0) threadIdx.x = (unsigned)to;
  r2l_ls_switch(save, next);
  // (back in `from`)
#if R2L_LS_ASAN
  __sanitizer_finish_switch_fiber(from >= 0 ? g.fib[(size_t)from].fake : g.main_fake, nullptr, nullptr);
#endif
}
// hand the processor to the next lane that has not returned
inline void yield(const char* what) {
  Group& g = *g_group;
  const int me = g.cur;
  if (++g.stalled > 4ul * (unsigned long)g.nt + 16) {
    fprintf(stderr, "r2l lock-step emulation: %s, lane %d: no lane can make progress while this one waits for a %s -- a rendezvous "
                    "in divergent control flow?  aborting\n", g.name, me, what);
    abort();
  }
  int nx = me;
  for (int i = 0; i < g.nt; ++i) {
    nx = nx + 1 == g.nt ? 0 : nx + 1;
    if (!g.fib[(size_t)nx].done) break;
  }
  if (nx != me) switch_to(g, me, nx);
}
inline void barrier_wait(Barrier& b, const char* what) {
  Group& g = *g_group;
  const unsigned gen = b.gen;
  if (++b.waiting >= b.alive) {
    b.waiting = 0;
    ++b.gen;
    g.stalled = 0;
    return;
  }
  while (b.gen == gen) yield(what);
}
inline void barrier_leave(Barrier& b) {
  --b.alive;
  if (b.alive > 0 && b.waiting >= b.alive) {  // the lanes still waiting were waiting for this one only
    b.waiting = 0;
    ++b.gen;
  }
  g_group->stalled = 0;
}
inline void wg_barrier() { barrier_wait(g_group->wg, "workgroup barrier"); }
// A lane whose EXEC bit goes off for the rest of the kernel (`if (x0 >= W) return;` in a device function whose caller goes on to a
// workgroup barrier): its wavefront's DPP moves no longer wait for it.  It still counts for the workgroup barrier, where the
// hardware counts wavefronts, not lanes.
inline void retire_lane() {
  Group& g = *g_group;
  Fiber& f = g.fib[(size_t)g.cur];
  if (!f.retired) {
    f.retired = true;
    barrier_leave(g.wave[(size_t)g.cur >> 6]);
  }
}

// every lane deposits `mine`; lane l receives the deposit of lane src(l) of its wavefront, or `keep` if src(l) < 0
template <class T, class SRC>
inline T exchange(T mine, T keep, SRC&& src) {
  static_assert(sizeof(T) <= 8, "one 64-bit slot per lane");
  Group& g = *g_group;
  const unsigned tid = (unsigned)g.cur, lane = tid & 63u, base = tid & ~63u;
  Fiber& f = g.fib[tid];
  std::vector<uint64_t>& s = g.slot[f.xcount & 1u];
  uint64_t bits = 0;
  memcpy(&bits, &mine, sizeof(T));
  s[tid] = bits;
  barrier_wait(g.wave[tid >> 6], "wave exchange (DPP / readfirstlane / shuffle)");
  const int from = src((int)lane);
  T out = keep;
  if (from >= 0) {
    unsigned fl = base + (unsigned)from;
    if (fl >= (unsigned)g.nt) fl = tid;  // (a partial last wavefront: lanes that do not exist hold the lane's own value)
    const uint64_t b = s[fl];
    memcpy(&out, &b, sizeof(T));
  }
  ++f.xcount;
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

// first instruction of every lane's fiber
inline void fiber_main() {
  Group& g = *g_group;
#if R2L_LS_ASAN
  {
    const void* bottom = nullptr;
    size_t size = 0;
    __sanitizer_finish_switch_fiber(nullptr, &bottom, &size);
    if (!g.main_bottom) {  // (the first lane is entered from the launcher: that is its stack)
      g.main_bottom = bottom;
      g.main_size = size;
    }
  }
#endif
  const int me = g.cur;
  g.entry(g.entry_arg);
  Fiber& f = g.fib[(size_t)me];
  if (!f.retired) barrier_leave(g.wave[(size_t)me >> 6]);  // this lane has returned: the others' rendezvous no longer count it
  barrier_leave(g.wg);
  f.done = true;
  g.stalled = 0;
  if (--g.running == 0) {
    switch_to(g, me, -1);
  } else {
    int nx = me;
    for (int i = 0; i < g.nt; ++i) {
      nx = nx + 1 == g.nt ? 0 : nx + 1;
      if (!g.fib[(size_t)nx].done) break;
    }
    switch_to(g, me, nx);
  }
  abort();  // (a finished lane is never resumed)
}
extern "C" inline void r2l_ls_fiber_trampoline() { fiber_main(); }

// Run `grid` workgroups of `nt` lanes one after the other: body(bid, lds) is the kernel's workgroup program, run by every lane
// with threadIdx.x set.  lds_floats: the kernel's LDS size (a fresh, poisoned heap block per launch).
template <class BODY>
inline void launch(const char* name, int grid, int nt, size_t lds_floats, const void* kernarg, BODY&& body) {
  static std::mutex one_launch;
  std::lock_guard<std::mutex> hold(one_launch);
  Group g;
  g.name = name;
  g.nt = nt;
  g.wave = std::vector<Barrier>((size_t)(nt + 63) / 64);
  g.slot[0].assign((size_t)nt, 0);
  g.slot[1].assign((size_t)nt, 0);
  g.fib.resize((size_t)nt);
  g.kernarg = kernarg;
  // the lanes' stacks: one mapping, a guard page at the low end of each
  const size_t span = STACK + 4096;
  char* stacks = (char*)mmap(nullptr, span * (size_t)nt, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (stacks == (char*)MAP_FAILED) abort();
  for (int t = 0; t < nt; ++t) {
    mprotect(stacks + span * (size_t)t, 4096, PROT_NONE);
    g.fib[(size_t)t].stack = stacks + span * (size_t)t + 4096;
  }
  // exactly the kernel's LDS (16-byte aligned like the device's)
  float* lds = nullptr;
  const size_t nl = lds_floats ? lds_floats : 4;
  if (posix_memalign((void**)&lds, 16, nl * sizeof(float)) != 0) abort();
  typedef typename std::remove_reference<BODY>::type BodyT;
  struct Call {
    BodyT* body;
    int bid;
    float* lds;
  } call{&body, 0, lds};
  g.entry = [](void* p) {
    Call* c = (Call*)p;
    (*c->body)(c->bid, c->lds);
  };
  g.entry_arg = &call;
  g_group = &g;
  for (int bid = 0; bid < grid; ++bid) {
    call.bid = bid;
    g.wg.reset(nt);
    for (size_t w = 0; w < g.wave.size(); ++w) {
      const int lanes = nt - (int)w * 64;
      g.wave[w].reset(lanes < 64 ? lanes : 64);
    }
    for (size_t i = 0; i < nl; ++i) {
      const uint32_t nan = 0x7fa00000u + (uint32_t)(i & 0xffff);  // (signalling NaNs: LDS holds garbage at launch)
      memcpy(&lds[i], &nan, 4);
    }
#if R2L_LS_ASAN
    // a lane that has finished never unwinds its last frames (it is switched away from for good): their redzones would stay
    // poisoned under the next workgroup's frames -- and under whatever the allocator puts at these addresses after munmap
    __asan_unpoison_memory_region(stacks, span * (size_t)nt);
#endif
    for (int t = 0; t < nt; ++t) {
      Fiber& f = g.fib[(size_t)t];
      f.done = false;
      f.retired = false;
      f.xcount = 0;
      f.fake = nullptr;
      // initial frame: six callee-saved registers (zero), then the trampoline as the return address; after `ret` the stack
      // pointer is 8 mod 16, as at any function entry
      uint64_t* top = (uint64_t*)(f.stack + STACK);
      top[-1] = 0;
      top[-2] = (uint64_t)(uintptr_t)&r2l_ls_fiber_trampoline;
      for (int k = 3; k <= 8; ++k) top[-k] = 0;
      f.sp = (void*)(top - 8);
    }
    g.running = nt;
    g.stalled = 0;
    switch_to(g, -1, 0);  // returns when the last lane of the workgroup has returned
  }
  g_group = nullptr;
  free(lds);
#if R2L_LS_ASAN
  __asan_unpoison_memory_region(stacks, span * (size_t)nt);
#endif
  munmap(stacks, span * (size_t)nt);
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
