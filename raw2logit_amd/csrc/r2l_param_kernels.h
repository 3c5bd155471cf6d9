// r2l_param_kernels.h -- fused parametrized ISP (torch semantics), forward and backward.
//
// Replaces ParametrizedProcessing.forward (processing/pipeline_torch.py:175-225) and the autograd
// graph behind it.  One workgroup (512 threads = 8 wavefronts) owns a TW x TH output tile and a
// FRAME of (TW+8) x (TH+8) positions around it (halo 4 = 1 debayer + 1 sharpen + 2 blur).  Three
// float planes of the frame live in LDS:
//     V   black-level-corrected raw value, mirror-extended outside the image      (:183, :233)
//     Y   luma after debayer/WB/CCM/RGB->YUV, ZERO outside the image                (:187-195 padding=1)
//     YP  sharpened luma, mirror-extended outside the image                         (:195, :165 reflect)
// Only the luma plane carries a halo, so the recomputed-halo cost is 2 x 9 FMA per frame pixel; the
// heavy per-pixel work (5x5 blur, chroma, YUV->RGB, clip, gamma, BatchNorm) runs once per pixel, 4 columns
// x 2 rows per thread, two adjacent pixels per packed-f32 instruction, reading LDS with 128-bit accesses.
// DESIGN.md section 3.2 has the whole picture (phases, pipeline, tile walk, reductions).
#pragma once
#include "r2l_common.h"

// diagnostic builds only (-DR2L_EXP_STAMPS): per-phase s_memtime totals of each workgroup's wave 0
#if defined(R2L_EXP_STAMPS) && !defined(R2L_EMUL)
#define R2L_STAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_amdgcn_s_memtime(), t1_; \
  const unsigned long long r0_ = __builtin_amdgcn_s_memrealtime();
#define R2L_STAMP(k) t1_ = __builtin_amdgcn_s_memtime(); st_[k] += t1_ - t0_; t0_ = t1_;
#define R2L_STAMP_FLUSH(dbg, bid) st_[7] = __builtin_amdgcn_s_memrealtime() - r0_; /* 100 MHz ticks: slot 7 */ \
  if (threadIdx.x == 0 && (dbg)) for (int k_ = 0; k_ < 8; ++k_) (dbg)[(size_t)(bid) * 8 + k_] = (float)st_[k_];
// stage stamps inside kernel B1's pixel phase (slots 1, 2 -- unused when Y' comes from the forward: blur + chroma | pointwise)
#define R2L_SUB_ARG , unsigned long long* sub_
#define R2L_SUB_PASS , (SAVED ? st_ : (unsigned long long*)nullptr)
#define R2L_SUB_PASS_FWD , sub_
#define R2L_SUB_PASS_NONE , (unsigned long long*)nullptr
#define R2L_SUB_BEGIN unsigned long long s0_ = __builtin_amdgcn_s_memtime(), s1_;
#define R2L_SUB(k) if (sub_) { s1_ = __builtin_amdgcn_s_memtime(); sub_[k] += s1_ - s0_; s0_ = s1_; }
#else
#define R2L_STAMP_DECL
#define R2L_STAMP(k)
#define R2L_STAMP_FLUSH(dbg, bid)
#define R2L_SUB_ARG
#define R2L_SUB_PASS
#define R2L_SUB_PASS_FWD
#define R2L_SUB_PASS_NONE
#define R2L_SUB_BEGIN
#define R2L_SUB(k)
#endif

template <int TW_, int TH_>
struct R2LGeom {
  static constexpr int TW = TW_, TH = TH_;
  static constexpr int FW = TW + 8, FH = TH + 8;
  // row stride in floats: room for the +2-shifted planes, and a multiple of 16 so that the 128-bit
  // LDS reads of a 16x4-thread wavefront (rows 4 apart) fall on distinct banks
  static constexpr int FS = ((FW + 2 + 15) / 16) * 16;
  static constexpr int PLANE = FH * FS;
  static constexpr int PAD = 16;  // leading floats so that index -1 of plane 0 stays inside LDS
  // Pixel phase: each thread owns 4 columns x 2 rows OF THE SAME BAYER ROW PARITY (rows r and r+2).
  // A wavefront is 16 threads across x 4 down; its 16-lane groups sit 4 pixel rows apart, so with FS a
  // multiple of 16 floats their 128-bit LDS reads fall on distinct banks.  The row parity is uniform per
  // wavefront, which halves the per-thread accumulators of the parity-indexed weight gradients.
  static constexpr int TXN = TW / 4;
  static_assert(TW == 64 && TH == 64 && R2L_NT == 512, "thread -> micro-tile map");
  static R2L_MEMBER void thread_tile(int tid, int& tx, int& row0, int& py) {
    tx = tid & 15;
    const int q = (tid >> 4) & 3, w = tid >> 6;
    py = w & 1;
    row0 = 16 * (w >> 1) + 4 * q + py;  // second row: row0 + 2
  }
};

template <class G, int NPLANES>
constexpr size_t r2l_lds_bytes() {
  return sizeof(float) * (size_t)(R2L_FOLDED_FLOATS + 2 * G::PAD + NPLANES * G::PLANE);
}

struct R2LTile {
  int b, oy, ox;  // image index, tile origin (global coordinates of tile pixel (0,0))
  bool border;    // the frame leaves the image somewhere (loads / extensions need the padding rules)
  bool ragged;    // the tile itself leaves the image, or W % 4 != 0 (stores need per-pixel predicates)
};

// XCD-aware tile walk: workgroup ids are dealt round-robin over the 8 XCDs (each with a private L2),
// so ids with equal (bid % 8) share an L2.  Each such group walks its own contiguous 1/8 of the tile
// list, which keeps halo rows/columns shared by neighbouring tiles inside one L2.
// Division of a tile index by a launch constant (tiles per row, rows per image, tiles per walk block): a runtime
// u32 division costs ~35 scalar instructions and the walk needs five per tile, on the critical path between two
// barriers (measured: 750 cycles per tile, 4 % of the backward kernels).  Round-up multiplier (Granlund &
// Montgomery): l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1, n / d = (mulhi(m, n) + n) >> l for n < 2^31.
// m comes from one float64 division: 2^32 (2^l - d) / d is an integer only for d a power of two (where it is 0,
// exactly), otherwise its fractional part is at least 1/d > 2^-31, far above the float64 rounding error.
struct R2LDiv {
  unsigned m, l, d;
};
R2L_HD R2LDiv r2l_div_init(int d_) {
  R2LDiv r;
  r.d = (unsigned)d_;
  r.l = 0;
  while ((1u << r.l) < r.d) ++r.l;
  r.m = (unsigned)(4294967296.0 * ((double)((1u << r.l) - r.d) / (double)r.d)) + 1u;
  return r;
}
R2L_HD unsigned r2l_mulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
R2L_HD int r2l_div(int n, const R2LDiv& dv) { return (int)((r2l_mulhi(dv.m, (unsigned)n) + (unsigned)n) >> dv.l); }

struct R2LTileWalk {
  int ntx, nty, ntiles, nper, group, w, k, jstep;
  int asym;  // > 0: every asym-th round of the walk is a HALF round, served by the older half of the workgroups only
  R2LDiv dx, dy, dj, dh;
};
// asym (experiment, off by default): two workgroups share a CU and the hardware's issue arbitration favours the waves of
// the one dispatched first (by age): with equal shares the older workgroup of kernel B2 finishes its tiles in ~140 kcycles,
// the younger one needs ~178.  Workgroups are dispatched in id order, so ids below nblk / 2 are the older ones: with
// asym = n they take one tile more every n rounds (n = 4, 8 tiles per workgroup on average: 9 vs 7).  The tile ->
// workgroup map stays a pure function of (bid, nblk): sums do not depend on the schedule.  Measured: the launch takes as
// long as before -- the CU's throughput, not the split, sets the time (profiles/r03_bwd2_tile_shares.txt).
R2L_HD R2LTileWalk r2l_walk_init(int B, int H, int W, int TW, int TH, int bid, int nblk, int asym = 0) {
  R2LTileWalk w;
  w.ntx = (W + TW - 1) / TW;
  w.nty = (H + TH - 1) / TH;
  w.ntiles = B * w.ntx * w.nty;
  const int ngroups = (nblk % 8 == 0) ? 8 : 1;
  w.nper = (w.ntiles + ngroups - 1) / ngroups;
  w.group = bid % ngroups;
  w.w = bid / ngroups;
  w.k = 0;
  w.jstep = nblk / ngroups;
  w.asym = (asym > 1 && (w.jstep & 1) == 0 && w.jstep >= 2) ? asym : 0;
  w.dx = r2l_div_init(w.ntx);
  w.dy = r2l_div_init(w.nty);
  w.dj = r2l_div_init(w.jstep);
  w.dh = r2l_div_init(w.jstep > 1 ? w.jstep / 2 : 1);
  return w;
}
// k-th tile of a workgroup: block k of `jstep` consecutive tiles, rotated by k tile rows + k tile columns
// so that a workgroup walks a diagonal of tile positions: border tiles cost more than interior ones (44 %
// of the tiles of a 512x512 frame are border tiles), and with a rotation by columns only the workgroups
// that start in the first or last tile row would see nothing but border tiles (measured: slowest workgroup
// 28 % above the mean).
R2L_HD bool r2l_walk_next(R2LTileWalk& w, int H, int W, int TW, int TH, R2LTile& t) {
  for (;;) {
    // first tile of round k: rounds before it hold jstep tiles each, but for the half rounds (every asym-th)
    int base = w.k * w.jstep;
    bool half = false;
    if (w.asym) {
      const int q = w.k / w.asym;
      base -= q * (w.jstep / 2);
      half = (w.k - q * w.asym) == w.asym - 1;
    }
    if (base >= w.nper) return false;
    const int rot = w.w + w.k * (w.ntx + 1);
    int j;
    if (half) {
      const int hs = w.jstep / 2;
      j = base + (rot - r2l_div(rot, w.dh) * hs);
      if (w.w >= hs) j = w.nper;  // the younger half sits this round out
    } else {
      j = base + (rot - r2l_div(rot, w.dj) * w.jstep);
    }
    w.k += 1;
    if (j >= w.nper) continue;
    const int tile = w.group * w.nper + j;
    if (tile >= w.ntiles) continue;
    const int r = r2l_div(tile, w.dx), tx = tile - r * w.ntx;
    t.b = r2l_div(r, w.dy);
    t.oy = (r - t.b * w.nty) * TH;
    t.ox = tx * TW;
    t.border = (t.oy < 4) || (t.ox < 4) || (t.oy + TH + 4 > H) || (t.ox + TW + 4 > W) || ((W & 3) != 0);
    t.ragged = (t.oy + TH > H) || (t.ox + TW > W) || ((W & 3) != 0);
    return true;
  }
}

// ---- phase A: raw tile + halo -> V -------------------------------------------------------------
// Work item = one float4 chunk of the frame; item ids tid, tid+256, ... are walked with incremental
// (row, chunk) updates instead of a division per item.
template <class G>
struct R2LChunkWalk {
  static constexpr int CPR = G::FW / 4;            // chunks per frame row
  static constexpr int DROW = R2L_NT / CPR, DCOL = R2L_NT % CPR;
  int fy, cx;
  R2L_MEMBER void init(int tid) {
    fy = tid / CPR;
    cx = tid - fy * CPR;
  }
  R2L_MEMBER void next() {
    fy += DROW;
    cx += DCOL;
    if (cx >= CPR) {
      cx -= CPR;
      fy += 1;
    }
  }
};

// Software pipeline of the tile loop: the raw chunks of the NEXT tile are fetched into registers at the
// start of the current tile's pixel phase (their HBM latency hides behind ~1400 VALU instructions) and
// written to LDS after the phase barrier.
template <class G>
struct R2LPrefetch {
  static constexpr int NIT = (R2LChunkWalk<G>::CPR * G::FH + R2L_NT - 1) / R2L_NT;
  r2l_f4 v[NIT];
};

// MODE 0: raw frame values with mirror coordinates outside the image (the V plane before black level)
// MODE 1: plain plane, zero outside the image
template <class G, bool BORDER, int MODE>
R2L_HD void r2l_fetch_frame(int tid, const float* gb, int oy, int ox, int H, int W, R2LPrefetch<G>& pf,
                            int only = -1 /* >= 0: that chunk of the lane alone (loads spread over a phase) */) {
  R2LChunkWalk<G> w;
  w.init(tid);
  const bool vec_ok = (W & 3) == 0;
  R2L_PRAGMA_UNROLL
  for (int it = 0; it < R2LPrefetch<G>::NIT; ++it) {
    if (only >= 0 && it != only) {
      w.next();
      continue;
    }
    r2l_f4 v;
    v.x = v.y = v.z = v.w = 0.f;
    if (w.fy < G::FH) {
      const int fy = w.fy, cx = w.cx;
      const int gx0 = ox - 4 + 4 * cx;
      if (!BORDER) {
        v = *(const r2l_f4*)(gb + (size_t)(oy - 4 + fy) * W + gx0);
      } else if (MODE == 0) {
        const int gy = r2l_mirror(oy - 4 + fy, H);
        const float* row = gb + (size_t)gy * W;
        if (vec_ok && gx0 >= 0 && gx0 + 3 < W) {
          v = *(const r2l_f4*)(row + gx0);
        } else {
          v.x = row[r2l_mirror(gx0, W)];
          v.y = row[r2l_mirror(gx0 + 1, W)];
          v.z = row[r2l_mirror(gx0 + 2, W)];
          v.w = row[r2l_mirror(gx0 + 3, W)];
        }
      } else {
        const int gy = oy - 4 + fy;
        if ((unsigned)gy < (unsigned)H) {
          const float* row = gb + (size_t)gy * W;
          if (vec_ok && gx0 >= 0 && gx0 + 3 < W) {
            v = *(const r2l_f4*)(row + gx0);
          } else {
            if ((unsigned)(gx0) < (unsigned)W) v.x = row[gx0];
            if ((unsigned)(gx0 + 1) < (unsigned)W) v.y = row[gx0 + 1];
            if ((unsigned)(gx0 + 2) < (unsigned)W) v.z = row[gx0 + 2];
            if ((unsigned)(gx0 + 3) < (unsigned)W) v.w = row[gx0 + 3];
          }
        }
      }
    }
    pf.v[it] = v;
    w.next();
  }
}
template <class G, int MODE>
R2L_HD void r2l_fetch_tile(int tid, const float* gb, const R2LTile& t, int H, int W, R2LPrefetch<G>& pf, int only = -1) {
  const float* base = gb + (size_t)t.b * H * W;
  if (t.border)
    r2l_fetch_frame<G, true, MODE>(tid, base, t.oy, t.ox, H, W, pf, only);
  else
    r2l_fetch_frame<G, false, MODE>(tid, base, t.oy, t.ox, H, W, pf, only);
}

// 16-bit containers: the 4 values of a chunk travel as raw bits in .x/.y of the prefetch register (decoded
// when they are written to LDS, so that the fetch stays a fire-and-forget load)
template <class G, bool BORDER>
R2L_HD void r2l_fetch_frame_u16(int tid, const unsigned short* gb, int oy, int ox, int H, int W,
                                R2LPrefetch<G>& pf, int only = -1) {
  R2LChunkWalk<G> w;
  w.init(tid);
  const bool vec_ok = (W & 3) == 0;
  R2L_PRAGMA_UNROLL
  for (int it = 0; it < R2LPrefetch<G>::NIT; ++it) {
    if (only >= 0 && it != only) {
      w.next();
      continue;
    }
    r2l_f2 v;
    v.x = v.y = 0.f;
    if (w.fy < G::FH) {
      const int fy = w.fy, cx = w.cx;
      const int gx0 = ox - 4 + 4 * cx;
      if (!BORDER) {
        v = *(const r2l_f2*)(gb + (size_t)(oy - 4 + fy) * W + gx0);
      } else {
        const int gy = r2l_mirror(oy - 4 + fy, H);
        const unsigned short* row = gb + (size_t)gy * W;
        if (vec_ok && gx0 >= 0 && gx0 + 3 < W) {
          v = *(const r2l_f2*)(row + gx0);
        } else {
          const unsigned a0 = row[r2l_mirror(gx0, W)], a1 = row[r2l_mirror(gx0 + 1, W)];
          const unsigned a2 = row[r2l_mirror(gx0 + 2, W)], a3 = row[r2l_mirror(gx0 + 3, W)];
          v.x = r2l_u2f(a0 | (a1 << 16));
          v.y = r2l_u2f(a2 | (a3 << 16));
        }
      }
    }
    pf.v[it].x = v.x;
    pf.v[it].y = v.y;
    w.next();
  }
}
// raw frame of tile t -> prefetch registers (either container type)
template <class G, bool U16>
R2L_HD void r2l_fetch_raw_tile(int tid, const R2LRaw& raw, const R2LTile& t, int H, int W, R2LPrefetch<G>& pf,
                               int only = -1) {
  if (U16) {
    const unsigned short* base = raw.u16 + (size_t)t.b * H * W;
    if (t.border)
      r2l_fetch_frame_u16<G, true>(tid, base, t.oy, t.ox, H, W, pf, only);
    else
      r2l_fetch_frame_u16<G, false>(tid, base, t.oy, t.ox, H, W, pf, only);
  } else {
    r2l_fetch_tile<G, 0>(tid, raw.f32, t, H, W, pf, only);
  }
}

// registers -> V plane, black level removed (mirror padding keeps the Bayer parity, so the site follows
// from the frame coordinates)
template <class G, bool U16>
R2L_HD void r2l_store_v(int tid, float* V, R2LFoldedRef F, const R2LPrefetch<G>& pf, const R2LRaw& raw) {
  R2LChunkWalk<G> w;
  w.init(tid);
  R2L_PRAGMA_UNROLL
  for (int it = 0; it < R2LPrefetch<G>::NIT; ++it) {
    if (w.fy < G::FH) {
      const int fy = w.fy, cx = w.cx;
      r2l_f4 v = pf.v[it];
      if (U16) {
        const unsigned lo = r2l_f2u(v.x), hi = r2l_f2u(v.y);
        v.x = r2l_raw_decode(lo & 0xffffu, raw);
        v.y = r2l_raw_decode(lo >> 16, raw);
        v.z = r2l_raw_decode(hi & 0xffffu, raw);
        v.w = r2l_raw_decode(hi >> 16, raw);
      }
      const float b0 = (fy & 1) ? F.bl[2] : F.bl[0], b1 = (fy & 1) ? F.bl[3] : F.bl[1];
      v.x -= b0;
      v.y -= b1;
      v.z -= b0;
      v.w -= b1;
      *(r2l_f4*)(V + fy * G::FS + 4 * cx) = v;
    }
    w.next();
  }
}
// registers -> plane stored shifted by +2 columns
template <class G>
R2L_HD void r2l_store_plane_s2(int tid, float* Pl, const R2LPrefetch<G>& pf) {
  R2LChunkWalk<G> w;
  w.init(tid);
  R2L_PRAGMA_UNROLL
  for (int it = 0; it < R2LPrefetch<G>::NIT; ++it) {
    if (w.fy < G::FH) {
      const r2l_f4 v = pf.v[it];
      float* d = Pl + w.fy * G::FS + 4 * w.cx + 2;
      r2l_f2 lo, hi;
      lo.x = v.x;
      lo.y = v.y;
      hi.x = v.z;
      hi.y = v.w;
      *(r2l_f2*)d = lo;
      *(r2l_f2*)(d + 2) = hi;
    }
    w.next();
  }
}

// (RPI + 2) rows x 6 columns window around a 4-wide x RPI-tall item at (fy0, fx): rows fy0-1..fy0+RPI,
// columns fx-1..fx+4 of an UNSHIFTED plane.  Rows past the end of the plane (items of the last row group)
// are clamped: their values only feed output rows that are not stored.
template <class G, int RPI>
R2L_HD void r2l_window_rows6(const float* Pl, int fy0, int fx, float w[RPI + 2][6]) {
  // The two outer columns are read as single floats.  As 128-bit reads (cols fx-4..fx-1 and fx+4..fx+7) three of
  // their four registers are dead, hipcc overlaps those dead registers with the destination of the NEXT read,
  // and the write-after-write hazard on a register with a load in flight makes it wait for every row's reads
  // before issuing the next row's: one LDS round trip per row instead of one per window.
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < RPI + 2; ++i) {
    const int fy = (fy0 - 1 + i < G::FH - 1) ? fy0 - 1 + i : G::FH - 1;
    const float* r = Pl + fy * G::FS + fx;
    const r2l_f4 m = r2l_lds_f4(r);
    w[i][0] = r2l_lds_f1(r - 1);
    w[i][1] = m.x;
    w[i][2] = m.y;
    w[i][3] = m.z;
    w[i][4] = m.w;
    w[i][5] = r2l_lds_f1(r + 4);
  }
}
// The stencil phases work on items of 4 columns x RPI rows, one item per lane, ONE pass (with 4 x 2 items the
// 630 items needed a second pass in which two wavefronts worked and six waited at the barrier).  4-row items
// (18 x 18 = 324 for Y, 18 x 17 = 306 for YP / the adjoint blur) fill 5 wavefronts, so SIMD 0 issues two streams
// while two SIMDs idle; taller items make exactly 4 wavefronts, one per SIMD.  That pays for the adjoint blur
// (5-row items, 252 of them: bwd2 -3 us); for Y and YP the taller register windows spill in the forward kernel
// (128 VGPRs) and cost more than the balance gains (measured: forward 78 -> 85 us with 5-row YP items, 92 us with
// 6-row Y items), so they stay at 4.  An even row count keeps the Bayer row parity of every output row of the Y
// phase a compile-time constant.
#ifndef R2L_RPI_Y
#define R2L_RPI_Y 4
#endif
#ifndef R2L_RPI_YP
#define R2L_RPI_YP 4
#endif
#ifndef R2L_RPI_ADJ
#define R2L_RPI_ADJ 3  // (5-row items: one wavefront per SIMD, -3 us at one workgroup per CU; 3-row items: a 7 x 8
                       //  window, which is what lets bwd2 fit 128 VGPRs = two workgroups per CU: 109 -> 97 us)
#endif
static_assert(R2L_RPI_Y % 2 == 0, "row parity of the Y phase");

// ---- phase B: Y on frame rows/cols [1, F-1) ------------------------------------------------------
template <class G, bool BORDER>
R2L_HD void r2l_compute_y(int tid, const float* V, float* Y, R2LFoldedRef F, int oy, int ox, int H,
                          int W) {
  constexpr int CPR = G::FW / 4, RPI = R2L_RPI_Y, NRG = (G::FH - 2 + RPI - 1) / RPI;
  static_assert(CPR * NRG <= R2L_NT, "one pass");
  const int rg = tid / CPR, col = tid - rg * CPR;
  if (rg >= NRG) return;
  const int fy0 = 1 + RPI * rg, fx = 4 * col;  // fy0 is odd
  float w[RPI + 2][6];
  r2l_window_rows6<G, RPI>(V, fy0, fx, w);
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < RPI; ++r) {
    const int py = (1 + r) & 1;  // parity of frame row fy0 + r (tile origins are even)
    r2l_p2 o[2];
    o[0] = o[1] = r2l_splat2(0.f);
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      const r2l_p2 wy = r2l_mk2(F.AY2[py][i * 3 + j][0], F.AY2[py][i * 3 + j][1]);
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) o[p] = r2l_pfma(wy, r2l_mk2(w[r + i][2 * p + j], w[r + i][2 * p + j + 1]), o[p]);
    }
    r2l_f4 st;
    st.x = o[0][0];
    st.y = o[0][1];
    st.z = o[1][0];
    st.w = o[1][1];
    if (BORDER) {  // zero padding of the sharpen conv: Y is 0 outside the image
      const int gy = oy - 4 + fy0 + r, gx = ox - 4 + fx;
      const bool yin = (unsigned)gy < (unsigned)H;
      st.x = (yin && (unsigned)gx < (unsigned)W) ? st.x : 0.f;
      st.y = (yin && (unsigned)(gx + 1) < (unsigned)W) ? st.y : 0.f;
      st.z = (yin && (unsigned)(gx + 2) < (unsigned)W) ? st.z : 0.f;
      st.w = (yin && (unsigned)(gx + 3) < (unsigned)W) ? st.w : 0.f;
    }
    if (fy0 + r <= G::FH - 2) *(r2l_f4*)(Y + (fy0 + r) * G::FS + fx) = st;
  }
}

// ---- phase C: YP = sharpen(Y) on frame rows/cols [2, F-2), stored shifted by +2 columns ----------
template <class G>
R2L_HD void r2l_compute_yp(int tid, const float* Y, float* YP, R2LFoldedRef F) {
  constexpr int CPR = G::FW / 4, RPI = R2L_RPI_YP, NRG = (G::FH - 4 + RPI - 1) / RPI;
  static_assert(CPR * NRG <= R2L_NT, "one pass");
  const int rg = tid / CPR, col = tid - rg * CPR;
  if (rg >= NRG) return;
  const int fy0 = 2 + RPI * rg, fx = 4 * col;
  float w[RPI + 2][6];
  r2l_window_rows6<G, RPI>(Y, fy0, fx, w);
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < RPI; ++r) {
    r2l_p2 o[2];
    o[0] = o[1] = r2l_splat2(0.f);
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      const r2l_p2 ws = r2l_splat2(F.sharp[i * 3 + j]);
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) o[p] = r2l_pfma(ws, r2l_mk2(w[r + i][2 * p + j], w[r + i][2 * p + j + 1]), o[p]);
    }
    float* d = YP + (fy0 + r) * G::FS + fx + 2;
    r2l_f2 lo, hi;
    lo.x = o[0][0];
    lo.y = o[0][1];
    hi.x = o[1][0];
    hi.y = o[1][1];
    if ((G::FH - 4) % RPI == 0 || fy0 + r < G::FH - 2) {  // ragged last row group
      *(r2l_f2*)d = lo;
      *(r2l_f2*)(d + 2) = hi;
    }
  }
}

// ---- phase C2 (border tiles): mirror-extend YP outside the image (padding_mode='reflect', :165) ---
template <class G>
R2L_HD void r2l_fill_yp_one(float* YP, int fy, int fx, int oy, int ox, int H, int W) {
  const int gy = oy - 4 + fy, gx = ox - 4 + fx;
  const int my = r2l_mirror(gy, H) - (oy - 4), mx = r2l_mirror(gx, W) - (ox - 4);
  if (my >= 2 && my < G::FH - 2 && mx >= 2 && mx < G::FW - 2)
    YP[fy * G::FS + fx + 2] = YP[my * G::FS + mx + 2];
}
R2L_HD int r2l_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// Only the strips outside the image are visited: columns [2,cl) u [cr,FW-2) over all rows, then rows
// [2,rt) u [rb,FH-2) over the in-image columns [cl,cr).
template <class G>
R2L_HD void r2l_fill_yp_mirror(int tid, float* YP, int oy, int ox, int H, int W) {
  const int cl = r2l_clampi(4 - ox, 2, G::FW - 2), cr = r2l_clampi(W - ox + 4, 2, G::FW - 2);
  const int rt = r2l_clampi(4 - oy, 2, G::FH - 2), rb = r2l_clampi(H - oy + 4, 2, G::FH - 2);
  const int wl = cl - 2, wa = wl + (G::FW - 2 - cr);
  if (wa > 0)
    for (int i = tid; i < wa * (G::FH - 4); i += R2L_NT) {
      const int r = i / wa, k = i - r * wa;
      r2l_fill_yp_one<G>(YP, 2 + r, k < wl ? 2 + k : cr + (k - wl), oy, ox, H, W);
    }
  const int ht = rt - 2, hb = ht + (G::FH - 2 - rb), wc = cr - cl;
  if (hb > 0 && wc > 0)
    for (int i = tid; i < hb * wc; i += R2L_NT) {
      const int k = i / wc, c = i - k * wc;
      r2l_fill_yp_one<G>(YP, k < ht ? 2 + k : rb + (k - ht), cl + c, oy, ox, H, W);
    }
}

// ---- phase D helpers: each thread owns a 4x4 micro-tile and walks it ONE OUTPUT ROW AT A TIME -------
// (a fully unrolled 4x4 body keeps all ~150 weights and a 8x8 + 6x6 window live at once, which hipcc
// answers with SGPR->VGPR-lane spills; per row the working set is 5x8 + 3x6 values and <= 70 weights)

// 5 rows x 8 columns of YP around output row `frow` (frame row of the output pixel row): rows
// frow-2..frow+2, columns fx0-2..fx0+5 with fx0 = 4*tx+4 (the plane is stored shifted by +2)
template <class G>
R2L_HD void r2l_rows_yp(const float* YP, int tx, int frow, float yw[5][8]) {
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 5; ++i) {
    const float* r = YP + (frow - 2 + i) * G::FS + 4 * tx + 4;
    const r2l_f4 a = r2l_lds_f4(r);
    const r2l_f4 b = r2l_lds_f4(r + 4);
    yw[i][0] = a.x;
    yw[i][1] = a.y;
    yw[i][2] = a.z;
    yw[i][3] = a.w;
    yw[i][4] = b.x;
    yw[i][5] = b.y;
    yw[i][6] = b.z;
    yw[i][7] = b.w;
  }
}
// 3 rows x 6 columns of an unshifted plane around output row `frow`: rows frow-1..frow+1, columns
// fx0-1..fx0+4, fetched as one aligned 128-bit read and two single floats per row
template <class G>
R2L_HD void r2l_rows_3x6(const float* Pl, int tx, int frow, float w[3][6]) {
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i) {  // outer columns as single floats: see r2l_window_rows6
    const float* r = Pl + (frow - 1 + i) * G::FS + 4 * tx + 4;
    const r2l_f4 m = r2l_lds_f4(r);
    w[i][0] = r2l_lds_f1(r - 1);
    w[i][1] = m.x;
    w[i][2] = m.y;
    w[i][3] = m.z;
    w[i][4] = m.w;
    w[i][5] = r2l_lds_f1(r + 4);
  }
}

// ---- final reduction inside the producing launch ------------------------------------------------------
// The per-workgroup partials are summed in a fixed two-level order by whichever workgroups arrive last:
// the last of every 16 consecutive workgroups adds its group's 16 partials per slot (float64) into a group
// partial, and the last group to finish adds the <= 64 group partials.  The result does not depend on
// arrival order (bitwise reproducible), no extra launch is needed, and the long part of the work (level 1)
// overlaps with workgroups that are still computing.  counters[1 + g] / counters[0] count arrivals; they are
// zero before the launch (the fold kernel initialises a fresh workspace) and are reset by the last arriver.
#ifndef R2L_MAX_BLOCKS
#define R2L_MAX_BLOCKS 2048
#endif
#define R2L_TREE_GROUP 16
#define R2L_MAX_GROUPS (R2L_MAX_BLOCKS / R2L_TREE_GROUP)
struct R2LTree {
  const float* partial;   // [split][nblk]
  const float* partial2;  // [nslots - split][nblk] (slots >= split), may be null when split == nslots
  double* gpartial;       // [group][nslots] (room for R2L_MAX_GROUPS x R2L_NSUMS)
  unsigned* counters;     // [1 + R2L_MAX_GROUPS]; null: no in-kernel reduction
  int split;
  int nblk1;  // workgroups that wrote `partial` (slots < split), when that was another launch with another grid
              // (bwd1: one workgroup per CU, bwd2: two); 0 = the grid of this launch
};
// Returns true (uniformly over the workgroup) in the ONE workgroup that arrives last; out[0..NSLOTS) then
// holds the totals.  `out` may be LDS or global memory; lds4: 4 floats of LDS scratch for the tickets;
// scratch: LDS staging area of scratch_n doubles (>= 16 NSLOTS).
// This is the tail of its launch -- the last workgroup runs it with the rest of the chip idle -- so it is written for
// latency: a lane issues all the coherent loads of its slot as one batch into registers and adds them in a fixed order
// from there (parked in LDS and added by a loop of dependent LDS reads, level 2 alone took 5.5 / 2.6 / 7.6 us of the
// statistics / bn_reduce / sums launches: profiles/r04_tails.txt); group partials are laid out [group][slot], so the
// level-2 loads of neighbouring lanes are neighbours; with few slots, P lanes share a slot's groups (group q goes to
// lane q mod P) and their P sums are added in lane order.
#ifndef R2L_EMUL
template <int CTRL>
R2L_HD double r2l_dpp_add_f64(double d) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, d);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
  return d + __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
R2L_HD double r2l_row16_sum(double d) {  // every lane: the sum over its row of 16 lanes (the order depends on the lane only)
  d = r2l_dpp_add_f64<0xb1>(d);          // quad_perm:[1,0,3,2]
  d = r2l_dpp_add_f64<0x4e>(d);          // quad_perm:[2,3,0,1]
  d = r2l_dpp_add_f64<0x124>(d);         // row_ror:4
  return r2l_dpp_add_f64<0x128>(d);      // row_ror:8
}
#endif
template <int NSLOTS, int NT>
struct R2LTreeShape {
  static constexpr int P0 = NT / NSLOTS;
  static constexpr int P = P0 >= 16 ? 16 : (P0 >= 8 ? 8 : (P0 >= 4 ? 4 : (P0 >= 2 ? 2 : 1)));  // lanes per slot, level 2
  // loads in flight per lane, level 2: one batch up to 768 workgroups (many slots) / all of a lane's groups (few)
  static constexpr int CH = NSLOTS >= 64 ? 48 : (R2L_MAX_GROUPS / P < 16 ? R2L_MAX_GROUPS / P : 16);
};
struct R2LNoWork {
#ifdef R2L_EMUL
  void operator()(int) const {}
#else
  __device__ void operator()(int) const {}
#endif
};
// Level 1: true (uniformly over the workgroup) in the workgroup that arrives last in its group of 16, after it has stored the
// group's partial of every slot.  `meanwhile(tid)`: work of every thread that does not depend on the totals, run while
// the ticket travels.
template <int NSLOTS, int NT = R2L_NT, class MEANWHILE = R2LNoWork>
R2L_BLOCKFN bool r2l_tree_level1(const R2LTree& tr, int bid, int nblk, float* lds4, MEANWHILE&& meanwhile = R2LNoWork()) {
  unsigned* lu = (unsigned*)lds4;
  const int g = bid / R2L_TREE_GROUP;
  const int g0 = g * R2L_TREE_GROUP;
  const int gsize = (nblk - g0 < R2L_TREE_GROUP) ? nblk - g0 : R2L_TREE_GROUP;
  R2L_PHASE_BEGIN
  unsigned t1 = 0;
  if (tid == 0) t1 = r2l_ticket(tr.counters + 1 + g);
  meanwhile(tid);
  if (tid == 0) lu[0] = t1;
  R2L_PHASE_END
  if (lu[0] + 1 != (unsigned)gsize) return false;
  R2L_TAILST(R2L_TAIL_BASE(NSLOTS) + 2);
  R2L_PHASE_BEGIN
  if (tid == 0) tr.counters[1 + g] = 0;
  {
    const int n1 = tr.nblk1 > 0 ? tr.nblk1 : nblk;
#ifdef R2L_EMUL
    for (int sl = tid; sl < NSLOTS; sl += NT) {
      const bool first = sl < tr.split;
      const float* src = first ? tr.partial + (size_t)sl * n1 + g0 : tr.partial2 + (size_t)(sl - tr.split) * nblk + g0;
      int lim = first ? n1 - g0 : gsize;  // (the other launch may have written fewer workgroups' partials)
      if (lim > gsize) lim = gsize;
      double acc = 0.0;
      for (int m = 0; m < lim; ++m) acc += (double)r2l_load_coherent(src + m);
      r2l_store_coherent(tr.gpartial + (size_t)g * NSLOTS + sl, acc);
    }
#else
    // sixteen neighbouring lanes take the sixteen workgroups of a slot (one 64-byte run of the slot's row), all loads of a
    // lane in one batch -- unconditional, from clamped addresses, zeros selected afterwards: a load under a condition is a
    // branch with its own wait -- and the sixteen are added in registers (DPP: pairs, quads, the row's quads)
    static_assert(NT % 16 == 0 && R2L_TREE_GROUP == 16, "a row of 16 lanes per slot");
    constexpr int NB1 = (NSLOTS * 16 + NT - 1) / NT;
    float v1[NB1];
    R2L_PRAGMA_UNROLL
    for (int it = 0; it < NB1; ++it) {
      const int idx = tid + it * NT, m = idx & 15;
      const int sl = (idx >> 4) < NSLOTS ? (idx >> 4) : NSLOTS - 1;
      const bool first = sl < tr.split;
      const float* src = first ? tr.partial + (size_t)sl * n1 + g0 : tr.partial2 + (size_t)(sl - tr.split) * nblk + g0;
      int lim = first ? n1 - g0 : gsize;  // (the other launch may have written fewer workgroups' partials)
      if (lim > gsize) lim = gsize;
      const int last = lim > 0 ? lim - 1 : 0;
      v1[it] = r2l_load_coherent(src + (m < last ? m : last));
      if (m >= lim) v1[it] = 0.f;
    }
    R2L_PRAGMA_UNROLL
    for (int it = 0; it < NB1; ++it) {
      const int idx = tid + it * NT;
      const double d = r2l_row16_sum((double)v1[it]);
      if ((idx & 15) == 0 && (idx >> 4) < NSLOTS) r2l_store_coherent(tr.gpartial + (size_t)g * NSLOTS + (idx >> 4), d);
    }
#endif
  }
  R2L_STORES_DONE();
  R2L_PHASE_END
  R2L_TAILST(R2L_TAIL_BASE(NSLOTS) + 4);
  return true;
}
// Level 2, for the workgroups level 1 let through (and `extra` more arrivals: workgroups that contribute something else the
// last one needs): true (uniformly) in the ONE workgroup that arrives last; out[0..NSLOTS) then holds the totals.
struct R2LNoExtra {
#ifdef R2L_EMUL
  double fetch(int) const { return 0.0; }
  void put(int, double) const {}
#else
  R2L_MEMBER double fetch(int) const { return 0.0; }
  R2L_MEMBER void put(int, double) const {}
#endif
};
// `also`: something else the last workgroup reads from global memory -- also.fetch(tid) goes out with the batch of loads,
// also.put(tid, value) files it once the batch is in
template <int NSLOTS, int NT = R2L_NT, class ALSO = R2LNoExtra>
R2L_BLOCKFN bool r2l_tree_level2(const R2LTree& tr, int nblk, int extra, float* lds4, double* out, double* scratch,
                                 ALSO&& also = R2LNoExtra()) {
  unsigned* lu = (unsigned*)lds4;
  const int ngroups = (nblk + R2L_TREE_GROUP - 1) / R2L_TREE_GROUP;
  R2L_PHASE_BEGIN
  if (tid == 0) lu[1] = r2l_ticket(tr.counters);
  R2L_PHASE_END
  if (lu[1] + 1 != (unsigned)(ngroups + extra)) return false;
  R2L_TAILST(R2L_TAIL_BASE(NSLOTS) + 5);
  // the group partials of every slot, in group order (lane `part` of a slot: groups part, part + P, ...)
  constexpr int P = R2LTreeShape<NSLOTS, NT>::P, CH = R2LTreeShape<NSLOTS, NT>::CH;
  R2L_PHASE_BEGIN
  if (tid == 0) tr.counters[0] = 0;
  const double also_v = also.fetch(tid);
  for (int w = tid; w < NSLOTS * P; w += NT) {
    const int part = w / NSLOTS, sl = w - part * NSLOTS;
    double acc = 0.0;
    for (int q0 = part; q0 < ngroups; q0 += CH * P) {
      double v[CH];
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < CH; ++k) {
        const int q = q0 + k * P;
        v[k] = r2l_load_coherent(tr.gpartial + (size_t)(q < ngroups ? q : ngroups - 1) * NSLOTS + sl);
      }
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < CH; ++k) acc += (q0 + k * P < ngroups) ? v[k] : 0.0;
    }
    if (P == 1)
      out[sl] = acc;
    else
      scratch[w] = acc;  // [part][slot]
  }
  also.put(tid, also_v);
  R2L_PHASE_END
  if (P > 1) {
    R2L_PHASE_BEGIN
    for (int sl = tid; sl < NSLOTS; sl += NT) {
      double acc = 0.0;
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < P; ++p) acc += scratch[p * NSLOTS + sl];
      out[sl] = acc;
    }
    R2L_PHASE_END
  }
  R2L_TAILST(R2L_TAIL_BASE(NSLOTS) + 6);
  return true;
}
template <int NSLOTS, int NT = R2L_NT, class MEANWHILE = R2LNoWork>
R2L_BLOCKFN bool r2l_tree_finish(const R2LTree& tr, int bid, int nblk, float* lds4, double* out, double* scratch,
                                 int scratch_n, MEANWHILE&& meanwhile = R2LNoWork()) {
  (void)scratch_n;
  if (!r2l_tree_level1<NSLOTS, NT>(tr, bid, nblk, lds4, meanwhile)) return false;
  return r2l_tree_level2<NSLOTS, NT>(tr, nblk, 0, lds4, out, scratch);
}

// sums (float64 in LDS) -> the 132 parameter gradients; tg (R2L_UNFOLD_TG doubles) and pl (R2L_P_COUNT floats)
// are LDS
#define R2L_UNFOLD_TG (126 + 12)  // T[9], gT[9], folded A[3][4][9], black-level partial sums [site][k]
// (every phase walks its work items with a stride of NT threads, so workgroups smaller than R2L_NT can run it too)
// r2l_unfold_from_lds: the packed parameters already sit in pl.  TABLES_DONE: and the tables that depend on the parameters
// only (r2l_unfold_tables_lane: T, the folded stencils A) in tg -- computed while the tree's first ticket travelled
template <int NT = R2L_NT>
R2L_HD void r2l_unfold_tables_lane(int tid, double* tg, const float* pl) {
  for (int t = tid; t < 128 + 108; t += NT) {
    if (t < 9) {
      tg[t] = r2l_fold_T_one(pl, t / 3, t % 3);
    } else if (t >= 128) {
      const int e = t - 128;
      tg[18 + e] = r2l_fold_A_one(pl, e / 36, (e % 36) / 9, e % 9);
    }
  }
}
template <int NT = R2L_NT, bool TABLES_DONE = false>
R2L_BLOCKFN void r2l_unfold_from_lds(const double* sums, double* tg, const float* pl, float* grad_params) {
  R2L_TAILST(23);
  R2L_PHASE_BEGIN
  if (!TABLES_DONE) r2l_unfold_tables_lane<NT>(tid, tg, pl);
  for (int t = tid; t < 128; t += NT) {
    if (t >= 64 && t < 73) tg[9 + t - 64] = r2l_unfold_gT(pl, sums, (t - 64) / 3, (t - 64) % 3);
    // the black-level gradient sums 108 products per site: 12 lanes take one (site, k) each (36 candidates, fixed order)
    // instead of 4 lanes walking all 108 -- this runs in the launch's last workgroup with the rest of the chip idle
    if (TABLES_DONE && t < 12) tg[126 + t] = r2l_unfold_bl_part(sums, tg, t / 3, t % 3);
  }
  R2L_PHASE_END
  R2L_TAILST(28);
  if (!TABLES_DONE) {
    R2L_PHASE_BEGIN
    for (int t = tid; t < 12; t += NT) tg[126 + t] = r2l_unfold_bl_part(sums, tg, t / 3, t % 3);
    R2L_PHASE_END
  }
  R2L_TAILST(29);
  R2L_PHASE_BEGIN
  for (int t = tid; t < R2L_P_NTRAIN; t += NT) grad_params[t] = r2l_unfold_one(pl, sums, t, tg);
  R2L_PHASE_END
}
template <int NT = R2L_NT>
R2L_BLOCKFN void r2l_unfold_phases(const float* params, const double* sums, double* tg, float* pl,
                                   float* grad_params) {
  R2L_PHASE_BEGIN
  for (int t = tid; t < R2L_P_COUNT; t += NT) pl[t] = params[t];
  R2L_PHASE_END
  r2l_unfold_from_lds<NT>(sums, tg, pl, grad_params);
}

// packed forms: pair p = columns (2p, 2p+1) of the micro-tile row
template <class FT>
R2L_HD void r2l_blur_row2(const float yw[5][8], const FT& F, r2l_p2 ypp[2]) {
  ypp[0] = ypp[1] = r2l_splat2(0.f);
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 5; ++i) {
    // aligned pairs P[k] = columns (2k, 2k+1) of the window row, straddling pairs O[k] = (2k+1, 2k+2)
    r2l_p2 P[4], O[3];
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 4; ++k) P[k] = r2l_mk2(yw[i][2 * k], yw[i][2 * k + 1]);
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) O[k] = r2l_straddle(P[k], P[k + 1]);
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 5; ++j) {
      const r2l_p2 w = r2l_splat2(F.blur[i * 5 + j]);
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) ypp[p] = r2l_pfma(w, (j & 1) ? O[p + j / 2] : P[p + j / 2], ypp[p]);
    }
  }
}
// the same with the 25 weights behind a pointer (interior weights, or the border-row sets of the streaming forward)
template <class WT>
R2L_HD void r2l_blur_row2w(const float yw[5][8], WT w25, r2l_p2 ypp[2]) {
  ypp[0] = ypp[1] = r2l_splat2(0.f);
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 5; ++i) {
    r2l_p2 P[4], O[3];
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 4; ++k) P[k] = r2l_mk2(yw[i][2 * k], yw[i][2 * k + 1]);
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) O[k] = r2l_straddle(P[k], P[k + 1]);
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 5; ++j) {
      const r2l_p2 w = r2l_splat2(w25[i * 5 + j]);
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) ypp[p] = r2l_pfma(w, (j & 1) ? O[p + j / 2] : P[p + j / 2], ypp[p]);
    }
  }
}
template <int PY, class FT>
R2L_HD void r2l_chroma_row2(const float vw[3][6], const FT& F, r2l_p2 u[2], r2l_p2 v[2]) {
  u[0] = u[1] = v[0] = v[1] = r2l_splat2(0.f);
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 3; ++j) {
    const r2l_p2 wu = r2l_mk2(F.AU2[PY][i * 3 + j][0], F.AU2[PY][i * 3 + j][1]);
    const r2l_p2 wv = r2l_mk2(F.AV2[PY][i * 3 + j][0], F.AV2[PY][i * 3 + j][1]);
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) {
      const r2l_p2 x = r2l_mk2(vw[i][2 * p + j], vw[i][2 * p + j + 1]);
      u[p] = r2l_pfma(wu, x, u[p]);
      v[p] = r2l_pfma(wv, x, v[p]);
    }
  }
}

// ---- BatchNorm bookkeeping on the device (no host round trip) ---------------------------------------
// tot[7] = sum_c(x-.5)[3], sum_c((x-.5)^2)[3], n  (already summed over ranks)  ->
//   bn[6]       = mean[3], 1/sqrt(var_biased + eps)[3]                (what the apply pass consumes)
//   moments[6]  = mean[3], var_biased[3] (float64)
//   running_mean / running_var (optional): nn.BatchNorm2d update with `momentum`, unbiased variance
struct R2LBnFinalizeArgs {
  const double* tot;  // [nranks][7]: the statistics vectors of all ranks, added here in rank order
  int nranks;
  float* bn;
  double* moments;
  float* running_mean;
  float* running_var;
  double eps, momentum;  // momentum < 0: cumulative moving average, 1 / num_batches_tracked (after its increment)
  long long* num_batches_tracked;  // optional, incremented by one
};
// what the bookkeeping reads from global memory (lanes 0-2): asked for early by callers whose last workgroup runs it at the
// end of a launch, so that it arrives behind waits that happen anyway
struct R2LBnPre {
  long long nbt;  // num_batches_tracked before this step
  float rm, rv;   // running_mean / running_var of the lane's channel
};
R2L_HD R2LBnPre r2l_bn_finalize_fetch(const R2LBnFinalizeArgs& a, int tid) {
  R2LBnPre p;
  p.nbt = 0;
  p.rm = p.rv = 0.f;
  if (tid < 3) {
    if (a.num_batches_tracked) p.nbt = *a.num_batches_tracked;
    if (a.running_mean) {
      p.rm = a.running_mean[tid];
      p.rv = a.running_var[tid];
    }
  }
  return p;
}
R2L_HD void r2l_bn_finalize_lane(const R2LBnFinalizeArgs& a, int tid, const R2LBnPre& pre) {
  if (tid < 3) {
    const long long nbt = pre.nbt + 1;
    const double mom = a.momentum < 0.0 ? 1.0 / (double)nbt : a.momentum;
    double n = 0.0, s1 = 0.0, s2 = 0.0;  // rank order: every rank computes bit-identical statistics
    for (int r = 0; r < a.nranks; ++r) {
      n += a.tot[r * 7 + 6];
      s1 += a.tot[r * 7 + tid];
      s2 += a.tot[r * 7 + 3 + tid];
    }
    const double m1 = s1 / n;
    double var = s2 / n - m1 * m1;
    var = var < 0.0 ? 0.0 : var;
    const double mean = m1 + 0.5;
    a.bn[tid] = (float)mean;
    a.bn[3 + tid] = (float)(1.0 / sqrt(var + a.eps));
    if (a.moments) {
      a.moments[tid] = mean;
      a.moments[3 + tid] = var;
      a.moments[6] = n;  // pixel count of the global batch (the BatchNorm backward divides by it)
    }
    if (a.running_mean) {
      const double unb = var * (n / (n > 1.0 ? n - 1.0 : 1.0));
      a.running_mean[tid] = (float)((1.0 - mom) * (double)pre.rm + mom * mean);
      a.running_var[tid] = (float)((1.0 - mom) * (double)pre.rv + mom * unb);
    }
  }
}
R2L_BLOCKFN void r2l_bn_finalize_phases(const R2LBnFinalizeArgs& a) {
  R2L_PHASE_BEGIN
  r2l_bn_finalize_lane(a, tid, r2l_bn_finalize_fetch(a, tid));
  R2L_PHASE_END
  R2L_PHASE_BEGIN  // after every lane has read the counter
  if (tid == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;
  R2L_PHASE_END
}

// ================================================================================================
// forward
// ================================================================================================
struct R2LFwdArgs {
  R2LRaw raw;             // (B,H,W)
  const float* additive;  // (3,256,256) or null
  const R2LFolded* F;
  const float* bn;      // mean[3], istd[3] or null
  float* out;           // (B,3,H,W) or null (stats only)
  float* stat_partial;  // [12][nblk] or null: (high, low) float32 halves of the workgroups' float64 totals
  int B, H, W;
  float* debug;  // diagnostic builds
  R2LTree tree;  // in-kernel final reduction of the statistics -> stats_out[0..6), stats_out[6] = B*H*W
  double* stats_out;
  R2LBnFinalizeArgs fin;  // fin.bn != null (one rank): the last workgroup also runs the BatchNorm bookkeeping
  R2LEpi ep;              // ep.on: the output goes to its augmented position (R2LEpi)
};

// Two forward workgroups share a CU; the hardware favours the older one's waves, so left alone the younger
// workgroup finishes ~40 % later and runs the tail at half occupancy.  Waves in the short LDS-bound stencil
// phases get a higher issue priority than waves in the long VALU-bound pixel phase: the two workgroups then
// interleave phase by phase instead of by age (measured: 82 -> 78 us).
#ifndef R2L_PRIO_PIXELS
#define R2L_PRIO_PIXELS 0
#endif
#ifndef R2L_PRIO_STENCIL
#define R2L_PRIO_STENCIL 2
#endif
// BatchNorm statistics of a thread: sum (x - p), sum (x - p)^2 per channel and pair half about the thread's own pivot p
// (the first pixel it sees), npx pixels.  Re-based to the common pivot 0.5 in float64 when the workgroup reduces
// (r2l_fwd_stats_reduce; see R2LFsState in r2l_param_stream.h for why).
struct R2LFwdRegs {
  r2l_p2 acc[6];
  float piv[3];
  float npx;
};

template <class G, int PY, bool RAGGED, bool ADD>
R2L_HD void r2l_fwd_row(const float* V, const float* YP, const R2LFwdArgs& a, int tx, int frow, int gx0,
                        unsigned off0, unsigned plane, float* ob, const float mean[3],
                        const float istd[3], R2LFwdRegs& regs) {
  R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque(a.F));
  r2l_p2 ypp[2], u[2], v[2];
  {
    float yw[5][8];
    r2l_rows_yp<G>(YP, tx, frow, yw);
    r2l_blur_row2(yw, F, ypp);
  }
  {
    float vw[3][6];
    r2l_rows_3x6<G>(V, tx, frow, vw);
    r2l_chroma_row2<PY>(vw, F, u, v);
  }
  const bool vec_ok = !RAGGED || (((a.W & 3) == 0) && (gx0 + 3 < a.W));
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    r2l_p2 x[2];
    const unsigned off = (unsigned)k * plane + off0;
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) {
      r2l_p2 rgb = r2l_pmul(r2l_splat2(F.M2[k * 3]), ypp[p]);
      rgb = r2l_pfma(r2l_splat2(F.M2[k * 3 + 1]), u[p], rgb);
      rgb = r2l_pfma(r2l_splat2(F.M2[k * 3 + 2]), v[p], rgb);
      const r2l_p2 lg = r2l_mk2(r2l_log2(fminf(fmaxf(rgb[0], 1e-5f), 1.0f)),     // :206
                                r2l_log2(fminf(fmaxf(rgb[1], 1e-5f), 1.0f)));
      const r2l_p2 e = r2l_pmul(lg, r2l_splat2(F.inv_gamma));                    // :209
      x[p] = r2l_mk2(r2l_exp2(e[0]), r2l_exp2(e[1]));
      if (ADD) {                                                                 // :213 (H == W == 256)
        const float a0 = (!RAGGED || gx0 + 2 * p < a.W) ? a.additive[off + 2 * p] : 0.f;
        const float a1 = (!RAGGED || gx0 + 2 * p + 1 < a.W) ? a.additive[off + 2 * p + 1] : 0.f;
        x[p] = r2l_padd(x[p], r2l_mk2(a0, a1));
      }
      if (a.stat_partial) {
        if (p == 0) regs.piv[k] = (regs.npx == 0.f) ? x[0][0] : regs.piv[k];  // (pixel 0 of an active thread is inside the image)
        r2l_p2 d = r2l_padd(x[p], r2l_splat2(-regs.piv[k]));
        if (RAGGED) d = r2l_mk2(gx0 + 2 * p < a.W ? d[0] : 0.f, gx0 + 2 * p + 1 < a.W ? d[1] : 0.f);
        regs.acc[k] = r2l_padd(regs.acc[k], d);
        regs.acc[3 + k] = r2l_pfma(d, d, regs.acc[3 + k]);
      }
    }
    if (ob) {
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p)
        x[p] = r2l_pmul(r2l_padd(x[p], r2l_splat2(-mean[k])), r2l_splat2(istd[k]));  // :217
      if (a.ep.on) {  // output epilogue (this tile kernel is the fallback forward: element by element)
        float* o = ob + (unsigned)k * plane + (a.ep.s0 + a.ep.sr * (int)((off0 - (unsigned)gx0) / (unsigned)a.W) + a.ep.sc * gx0);
        R2L_PRAGMA_UNROLL
        for (int c = 0; c < 4; ++c)
          if (!RAGGED || gx0 + c < a.W) o[c * a.ep.sc] = x[c >> 1][c & 1];
      } else if (vec_ok) {
        r2l_f4 st;
        st.x = x[0][0];
        st.y = x[0][1];
        st.z = x[1][0];
        st.w = x[1][1];
        *(r2l_f4*)(ob + off) = st;  // (a nontemporal store here costs the apply pass 4 %)
      } else {
        R2L_PRAGMA_UNROLL
        for (int c = 0; c < 4; ++c)
          if (gx0 + c < a.W) ob[off + c] = x[c >> 1][c & 1];
      }
    }
  }
  if (a.stat_partial) {
    float n = 4.f;
    if (RAGGED) n = (float)((gx0 < a.W) + (gx0 + 1 < a.W) + (gx0 + 2 < a.W) + (gx0 + 3 < a.W));
    regs.npx += n;
  }
}

template <class G, bool RAGGED, bool ADD>
R2L_HD void r2l_fwd_pixels(int tid, const float* V, const float* YP, const R2LFwdArgs& a,
                           const R2LTile& t, R2LFwdRegs& regs) {
  int tx, row0, py;
  G::thread_tile(tid, tx, row0, py);
  const int gy0 = t.oy + row0, gx0 = t.ox + 4 * tx;
  if (RAGGED && (gy0 >= a.H || gx0 >= a.W)) return;  // micro-tile entirely outside a ragged image edge
  const unsigned plane = (unsigned)a.H * (unsigned)a.W;           // < 2^29 (checked by the ABI)
  float* ob = a.out ? a.out + (size_t)t.b * 3 * plane : nullptr;  // wave-uniform image base
  const unsigned pix0 = (unsigned)gy0 * (unsigned)a.W + (unsigned)gx0;
  float mean[3] = {0.f, 0.f, 0.f}, istd[3] = {1.f, 1.f, 1.f};
  if (a.bn) {
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      mean[k] = a.bn[k];
      istd[k] = a.bn[3 + k];
    }
  }
  // (the stage-major form of r2l_bwd1_rows2 does not pay here: at 128 VGPRs it spills, 78 -> 173 us; the 5x5 blur
  // of both rows first, the rest row by row: 77 -> 79.5 us -- four waves per SIMD already hide the scalar loads)
  R2L_PRAGMA_NOUNROLL
  for (int rr = 0; rr < 4; rr += 2) {  // rows row0 and row0 + 2
    if (RAGGED && gy0 + rr >= a.H) break;
    const unsigned off0 = pix0 + (unsigned)rr * (unsigned)a.W;
    if (py)
      r2l_fwd_row<G, 1, RAGGED, ADD>(V, YP, a, tx, row0 + 4 + rr, gx0, off0, plane, ob, mean, istd, regs);
    else
      r2l_fwd_row<G, 0, RAGGED, ADD>(V, YP, a, tx, row0 + 4 + rr, gx0, off0, plane, ob, mean, istd, regs);
  }
}

// per-thread accumulators -> one partial per slot per workgroup, in a fixed order (bitwise
// reproducible).  Slots go through LDS 32 at a time: every thread parks its 32 values, then 16 threads
// per slot add 32 of the R2L_NT values each (lane `part` takes threads part, part+16, ...: at most a 2-way
// bank conflict), and one thread per slot adds the 16 partial sums.
#define R2L_RED_ROWS(NACC) ((NACC) < 32 ? (NACC) : 32)
#define R2L_RED_FLOATS_N(NACC) (R2L_RED_ROWS(NACC) * (R2L_NT + 1) + 32 * 16)  // LDS floats of a NACC-slot reduction
#define R2L_RED_FLOATS R2L_RED_FLOATS_N(32)
#define R2L_ACC_DIRECT(regs, i) R2L_TREG(regs).acc[i]
#define R2L_BLOCK_REDUCE(NACC, regs, lds, partial, bid, nblk) \
  R2L_BLOCK_REDUCE_F(NACC, R2L_ACC_DIRECT, regs, lds, partial, bid, nblk)
#define R2L_BLOCK_REDUCE_F(NACC, VAL, regs, lds, partial, bid, nblk)                        \
  R2L_PRAGMA_UNROLL                                                                         \
  for (int base_ = 0; base_ < (NACC); base_ += 32) {                                        \
    R2L_PHASE_BEGIN                                                                         \
    R2L_PRAGMA_UNROLL                                                                       \
    for (int i_ = 0; i_ < 32; ++i_)                                                         \
      if (base_ + i_ < (NACC)) (lds)[i_ * (R2L_NT + 1) + tid] = VAL(regs, base_ + i_);     \
    R2L_PHASE_END                                                                           \
    R2L_PHASE_BEGIN                                                                         \
    {                                                                                       \
      const int slot_ = tid >> 4, part_ = tid & 15;                                         \
      float s_ = 0.f;                                                                       \
      if (base_ + slot_ < (NACC)) {                                                         \
        R2L_PRAGMA_UNROLL                                                                   \
        for (int j_ = 0; j_ < R2L_NT / 16; ++j_)                                            \
          s_ += (lds)[slot_ * (R2L_NT + 1) + part_ + 16 * j_];                              \
      }                                                                                     \
      (lds)[R2L_RED_ROWS(NACC) * (R2L_NT + 1) + tid] = s_;                                                  \
    }                                                                                       \
    R2L_PHASE_END                                                                           \
    R2L_PHASE_BEGIN                                                                         \
    if (tid < 32 && base_ + tid < (NACC)) {                                                 \
      float s_ = 0.f;                                                                       \
      R2L_PRAGMA_UNROLL                                                                     \
      for (int j_ = 0; j_ < 16; ++j_) s_ += (lds)[R2L_RED_ROWS(NACC) * (R2L_NT + 1) + tid * 16 + j_];       \
      r2l_store_coherent(&(partial)[(size_t)(base_ + tid) * (nblk) + (bid)], s_);           \
    }                                                                                       \
    R2L_STORES_DONE(); /* a later workgroup may finish the reduction inside this launch */  \
    R2L_PHASE_END                                                                           \
  }


#ifndef R2L_FWD_L
#define R2L_FWD_L 17  // phases of the forward tile loop that launder tid (bit 0: the store phase, bit 4: the pixel phase): 121 VGPRs, no scratch
                      // (round 6: 16 alone had crept back to 128 VGPRs + 12 B of scratch)
#endif
template <class G, bool ADD, bool MAYBE_RAGGED, bool U16>
R2L_BLOCKFN void r2l_fwd_block(const R2LFwdArgs& a, int bid, int nblk, float* lds) {
  float* V = lds + R2L_FOLDED_FLOATS + G::PAD;
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  float* Y = V + G::PLANE;
  float* YP = Y + G::PLANE;
  R2L_TREG_DECL(R2LFwdRegs, regs);
  R2L_TREG_DECL(R2LPrefetch<G>, pre);
  R2LTileWalk w = r2l_walk_init(a.B, a.H, a.W, G::TW, G::TH, bid, nblk);
  R2LTile t, tn;
  bool have = r2l_walk_next(w, a.H, a.W, G::TW, G::TH, t);
  R2L_PHASE_BEGIN
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 6; ++i) R2L_TREG(regs).acc[i] = r2l_splat2(0.f);
  R2L_TREG(regs).piv[0] = R2L_TREG(regs).piv[1] = R2L_TREG(regs).piv[2] = 0.5f;
  R2L_TREG(regs).npx = 0.f;
  if (have) r2l_fetch_raw_tile<G, U16>(tid, a.raw, t, a.H, a.W, R2L_TREG(pre));
  R2L_PHASE_END
  R2L_STAMP_DECL
  while (have) {
    R2L_PHASE_BEGIN_IF(R2L_FWD_L & 1)
    r2l_store_v<G, U16>(tid, V, F, R2L_TREG(pre), a.raw);
    R2L_PHASE_END
    R2L_STAMP(0)
    const bool haven = r2l_walk_next(w, a.H, a.W, G::TW, G::TH, tn);
    R2L_PHASE_BEGIN_IF(R2L_FWD_L & 2)
    if (t.border)
      r2l_compute_y<G, true>(tid, V, Y, F, t.oy, t.ox, a.H, a.W);
    else
      r2l_compute_y<G, false>(tid, V, Y, F, t.oy, t.ox, a.H, a.W);
    R2L_PHASE_END
    R2L_STAMP(1)
    R2L_PHASE_BEGIN_IF(R2L_FWD_L & 4)
    r2l_compute_yp<G>(tid, Y, YP, F);
    R2L_PHASE_END
    R2L_STAMP(2)
    if (t.border) {
      R2L_PHASE_BEGIN_IF(R2L_FWD_L & 8)
      r2l_fill_yp_mirror<G>(tid, YP, t.oy, t.ox, a.H, a.W);
      R2L_PHASE_END
    R2L_STAMP(3)
    }
    R2L_PHASE_BEGIN_IF(R2L_FWD_L & 16)
    if (haven) r2l_fetch_raw_tile<G, U16>(tid, a.raw, tn, a.H, a.W, R2L_TREG(pre));  // next tile, in flight
    R2L_PRIO(R2L_PRIO_PIXELS);
    if (MAYBE_RAGGED && t.ragged)
      r2l_fwd_pixels<G, MAYBE_RAGGED, ADD>(tid, V, YP, a, t, R2L_TREG(regs));
    else
      r2l_fwd_pixels<G, false, ADD>(tid, V, YP, a, t, R2L_TREG(regs));
    R2L_PRIO(R2L_PRIO_STENCIL);
    R2L_PHASE_END
    R2L_STAMP(4)
    t = tn;
    have = haven;
  }
  R2L_STAMP_FLUSH(a.debug, bid)
  if (a.stat_partial) {
    // threads -> one float64 total per slot and workgroup, in a fixed order: every thread re-bases its sums to the pivot
    // 0.5 and parks them in LDS (6 x R2L_NT doubles), 16 threads per slot add 32 of them each, one adds the 16.  The
    // workgroup's totals travel as (high, low) float32 pairs -- slots i and 6 + i -- which the tree adds up separately.
    double* dl = (double*)lds;
    R2L_PHASE_BEGIN
    const R2LFwdRegs& r = R2L_TREG(regs);
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      const double s1 = (double)r.acc[k][0] + (double)r.acc[k][1], s2 = (double)r.acc[3 + k][0] + (double)r.acc[3 + k][1];
      const double dp = (double)r.piv[k] - 0.5, n = (double)r.npx;
      dl[k * R2L_NT + tid] = fma(n, dp, s1);
      dl[(3 + k) * R2L_NT + tid] = fma(dp, fma(n, dp, 2.0 * s1), s2);
    }
    R2L_PHASE_END
    R2L_PHASE_BEGIN
    if (tid < 96) {
      const int slot = tid >> 4, part = tid & 15;
      double acc = 0.0;
      for (int j = 0; j < R2L_NT / 16; ++j) acc += dl[slot * R2L_NT + part + 16 * j];
      dl[6 * R2L_NT + tid] = acc;
    }
    R2L_PHASE_END
    R2L_PHASE_BEGIN
    if (tid < 6) {
      double acc = 0.0;
      for (int j = 0; j < 16; ++j) acc += dl[6 * R2L_NT + tid * 16 + j];
      const float hi = (float)acc;
      r2l_store_coherent(&a.stat_partial[(size_t)tid * nblk + bid], hi);
      r2l_store_coherent(&a.stat_partial[(size_t)(6 + tid) * nblk + bid], (float)(acc - (double)hi));
    }
    R2L_STORES_DONE();
    R2L_PHASE_END
    double* sl = (double*)(lds + 4);  // totals in LDS: the bookkeeping below reads them back
    if (a.tree.counters && r2l_tree_finish<12>(a.tree, bid, nblk, lds, sl, (double*)(lds + 512),
                                               (R2L_RED_FLOATS - 512) / 2)) {
      R2L_PHASE_BEGIN
      if (tid < 6) sl[tid] += sl[6 + tid];
      R2L_PHASE_END
      R2L_PHASE_BEGIN
      if (tid == 0) sl[6] = (double)a.B * (double)a.H * (double)a.W;
      R2L_PHASE_END
      R2L_PHASE_BEGIN
      if (tid < 7) a.stats_out[tid] = sl[tid];
      R2L_PHASE_END
      if (a.fin.bn) {
        R2LBnFinalizeArgs f = a.fin;
        f.tot = sl;
        f.nranks = 1;
        r2l_bn_finalize_phases(f);
      }
    }
  }
}

// ================================================================================================
// backward, kernel B1: everything that is pointwise in the pixel + the blur-weight / chroma sums
// ================================================================================================
struct R2LBwd1Args {
  R2LRaw raw;
  const float* additive;
  const R2LFolded* F;
  const float* bn;      // mean[3], istd[3] or null
  const float* bn_bwd;  // mean_g[3], mean_gxhat[3] or null
  const float* gout;    // (B,3,H,W)
  float* gypp;          // (B,H,W): d loss / d Y'' (blurred luma)
  float* partial;       // [R2L_B1_NACC][nblk]
  int B, H, W;
  float* debug;
  const float* yp;      // (B,H,W) or null: the sharpened luma Y' the forward kept (SAVED instantiations)
  R2LEpi ep;            // ep.on: grad_out arrives in the augmented layout the forward's epilogue wrote (R2LEpi)
  int band_h;           // r2l_bwd1_plane_block: rows per work item (a multiple of 6)
  float* hp;            // r2l_bwd1_blur_hp_block: (B,H,W) the blur's adjoint of dL/dY'' (kernel B2's plane)
  int band_hb;          // r2l_bwd1_blur_hp_block: rows per work item (a multiple of 6)
  int xcdm, xcdm_hb;    // plane passes: neighbouring workgroups per XCD (r2l_xcd_window; 0 = off)
};

// per-thread accumulators of B1, as pairs: element h of a pair belongs to the pixels in the even (h = 0) or
// odd (h = 1) columns of the micro-tile.  A thread only sees pixels of ONE row parity, so for the
// parity-indexed tables the pair IS the two column parities of its row parity; for the other sums the halves
// are added when the workgroup reduces.
enum { R2L_L1_GBLUR = 0, R2L_L1_GAU = 25, R2L_L1_GAV = 34, R2L_L1_SU = 43, R2L_L1_SV = 44,
       R2L_L1_GGAM = 45, R2L_L1_NACC = 46 };
struct R2LBwd1Regs {
  r2l_p2 acc[R2L_L1_NACC];
  int py;
};
// value of global slot i (layout R2L_B1_*) held by a thread of row parity py
R2L_HD float r2l_b1_slot(const R2LBwd1Regs& r, int i) {
  if (i < R2L_B1_GAU) return r.acc[R2L_L1_GBLUR + i][0] + r.acc[R2L_L1_GBLUR + i][1];
  if (i < R2L_B1_SU) {
    const int tbl = (i - R2L_B1_GAU) / 36, k = (i - R2L_B1_GAU) % 36, par = k / 9, t = k % 9;
    const r2l_p2 v = r.acc[(tbl ? R2L_L1_GAV : R2L_L1_GAU) + t];
    return ((par >> 1) == r.py) ? ((par & 1) ? v[1] : v[0]) : 0.f;
  }
  if (i < R2L_B1_GGAM) {
    const int tbl = (i - R2L_B1_SU) / 4, par = (i - R2L_B1_SU) % 4;
    const r2l_p2 v = r.acc[tbl ? R2L_L1_SV : R2L_L1_SU];
    return ((par >> 1) == r.py) ? ((par & 1) ? v[1] : v[0]) : 0.f;
  }
  return r.acc[R2L_L1_GGAM][0] + r.acc[R2L_L1_GGAM][1];
}
#define R2L_ACC_B1(regs, i) r2l_b1_slot(R2L_TREG(regs), i)

struct R2LBnConsts {
  float mean[3], istd[3], mg[3], mgx[3];  // mg, mgx: istd * mean(g), istd * mean(g * xhat) (r2l_bn_bwd_pair)
};
// BatchNorm2d backward of one pixel pair, train mode: istd * (g - mean(g) - xhat * mean(g * xhat)), evaluated as
// fma(istd, g, -istd * mean(g)) - xhat * (istd * mean(g * xhat)).  Not (g - mean(g)) first: mean(g) is small against g, so
// that difference rounds by the SAME amount for every g of a binade (-(mean(g) mod ulp)) -- a bias of ~ N ulp / 4 in every sum
// over the N pixels of a plane, which the black-level gradient (a sum that cancels to 1/1000 of its terms when BatchNorm sits
// on low-variance frames) showed as 3e-4 of its scale; the product istd * g has fresh low bits for every pixel.  In eval mode
// mg = mgx = 0 and only the scaling remains; without BatchNorm istd = 1 too, which leaves g unchanged (exactly).
R2L_HD r2l_p2 r2l_bn_bwd_pair(r2l_p2 g, r2l_p2 xhat, float istd, float c0, float c1) {
  return r2l_pfma(xhat, r2l_splat2(-c1), r2l_pfma(r2l_splat2(istd), g, r2l_splat2(-c0)));
}


// 4 consecutive pixels of the ISP's layout read back from the augmented layout: p = the position of the first one
R2L_HD r2l_f4 r2l_epi_load4(const float* p, int sc) {
  r2l_f4 v;
  if (sc == 1) {
    v = *(const r2l_f4*)p;
  } else if (sc == -1) {
    const r2l_f4 q = *(const r2l_f4*)(p - 3);
    v.x = q.w;
    v.y = q.z;
    v.z = q.y;
    v.w = q.x;
  } else {
    v.x = p[0];
    v.y = p[sc];
    v.z = p[2 * sc];
    v.w = p[3 * sc];
  }
  return v;
}
// grad_out of this thread's 2 rows x 4 columns x 3 channels, fetched two phases ahead of its use: in the
// pixel phase a wave has one other wave per SIMD to hide behind (VGPR-bound, 1 workgroup per CU), which
// does not cover an HBM round trip; issued before the Y phase the loads have ~2 phases to land.
struct R2LGoutPre {
  r2l_f4 g[2][3];
};
template <class G>
R2L_HD void r2l_bwd1_fetch_gout(int tid, const R2LBwd1Args& a, const R2LTile& t, R2LGoutPre& gp, int r) {
  int tx, row0, py;
  G::thread_tile(tid, tx, row0, py);
  const unsigned plane = (unsigned)a.H * (unsigned)a.W;
  const float* gb = a.gout + (size_t)t.b * 3 * plane;
  const unsigned pix0 = (unsigned)(t.oy + row0) * (unsigned)a.W + (unsigned)(t.ox + 4 * tx);
  if (a.ep.on) {  // (store phase: registers and branches are cheap here)
    const float* o = gb + (a.ep.s0 + a.ep.sr * (t.oy + row0 + 2 * r) + a.ep.sc * (t.ox + 4 * tx));
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) gp.g[r][k] = r2l_epi_load4(o + (unsigned)k * plane, a.ep.sc);
    return;
  }
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k)
    // read once here: nontemporal, so that it does not evict the raw frames / dL/dY'' (A/B: -1 % on the step)
    gp.g[r][k] = r2l_load_f4_nt(gb + (unsigned)k * plane + pix0 + (unsigned)(2 * r) * (unsigned)a.W);
}

// The kept-luma frame of the NEXT tile through an LDS staging area instead of prefetch registers (hot instantiations:
// frames that tile exactly, Y' kept by the forward): asynchronous global -> LDS copies (r2l_glds16, no destination
// registers) issued from inside the current tile's pixel phase, one per lane behind a window row of the gradient
// correlations, and converted to the Y' plane in the next store phase.  The staging area is a dense (FH x FW) image of
// the frame: chunk c = it * R2L_NT + tid of the chunk walk is frame row c / (FW/4), columns 4 (c % (FW/4)) .. + 3, and
// lands at float 4 c.  Straight-line on purpose -- no branch, no predicate: the copies sit between the window rows of
// unrolled loops and every extra basic block there costs the register allocator dearly (89-134 spilled registers with an
// `if` per copy, none without).  So chunks outside the image copy a clamped in-image chunk (replaced by zeros when the
// plane is built), and the wavefronts past the end of the frame in the last round are pointed at a dump area behind it.
// What this buys: 12 registers (kernel B1: 255 -> 245, no scratch, no spilled registers) and 2 us; what was measured on
// the way (profiles/r03_b1_staging_modes.txt): grad_out and the raw frame staged the same way make the kernel and -- through
// the chip's clocks -- every other kernel of the step slower, a nontemporal policy on these copies slows the forward.
#define R2L_B1_FRAME_FLOATS (72 * 72 + 256)  // a 64 x 64 tile's frame + 1 KiB: dump area / overhang of the last round's wavefront
#ifndef R2L_B1_GLDS
#define R2L_B1_GLDS 1  // 0: A/B builds, the Y' frame prefetched into registers at the end of the pixel phase (round 2)
#endif
template <class G>
R2L_HD void r2l_stage_frame(int tid, const float* gb, int oy, int ox, int H, int W, float* FSG, int only = -1) {
  constexpr int NCH = G::FH * (G::FW / 4);
  R2LChunkWalk<G> w;
  w.init(tid);
  R2L_PRAGMA_UNROLL
  for (int it = 0; it < R2LPrefetch<G>::NIT; ++it) {
    if (only < 0 || only == it) {
      int gy = oy - 4 + w.fy, gx0 = ox - 4 + 4 * w.cx;
      gy = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy);
      gx0 = gx0 < 0 ? 0 : (gx0 > W - 4 ? W - 4 : gx0);
      const int c = it * R2L_NT + tid;
      // (wave-uniform: does this wavefront's first chunk exist?)
      float* slot = (it * R2L_NT + (tid & ~63) < NCH) ? FSG + 4 * c : FSG + 4 * NCH + 4 * (tid & 63);
      r2l_glds16(gb, 4u * ((unsigned)gy * (unsigned)W + (unsigned)gx0), slot);
    }
    w.next();
  }
}
// staged Y' frame -> plane stored shifted by +2 columns (r2l_store_plane_s2 with the chunks read back from LDS; zero
// outside the image)
template <class G>
R2L_HD void r2l_store_plane_s2_staged(int tid, float* Pl, const float* FSG, int oy, int ox, int H, int W) {
  R2LChunkWalk<G> w;
  w.init(tid);
  R2L_PRAGMA_UNROLL
  for (int it = 0; it < R2LPrefetch<G>::NIT; ++it) {
    if (w.fy < G::FH) {
      const r2l_f4 v = r2l_lds_f4(FSG + 4 * (it * R2L_NT + tid));
      const int gy = oy - 4 + w.fy, gx0 = ox - 4 + 4 * w.cx;
      const bool in = (unsigned)gy < (unsigned)H && gx0 >= 0 && gx0 + 3 < W;
      float* d = Pl + w.fy * G::FS + 4 * w.cx + 2;
      r2l_f2 lo, hi;
      lo.x = in ? v.x : 0.f;
      lo.y = in ? v.y : 0.f;
      hi.x = in ? v.z : 0.f;
      hi.y = in ? v.w : 0.f;
      *(r2l_f2*)d = lo;
      *(r2l_f2*)(d + 2) = hi;
    }
    w.next();
  }
}
// one output row (4 pixels of this thread) of kernel B1; PY = row parity
R2L_HD float r2l_pick(bool second, float a, float b) {
#ifndef R2L_EMUL
  asm("" : "+v"(a), "+v"(b));  // a select of two register values (never a load through a selected address)
#endif
  return second ? b : a;
}
template <class G, int PY, bool RAGGED, bool ADD, bool PRE>
R2L_HD void r2l_bwd1_row(const float* V, const float* YP, const R2LBwd1Args& a, int tx, int frow, int gx0,
                         unsigned off0, unsigned plane, const float* gb, const R2LGoutPre& gpre, bool second,
                         float* gyb, const R2LBnConsts& bc, R2LBwd1Regs& regs) {
  R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque(a.F));
  r2l_p2 ypp[2], u[2], v[2];
  {
    float yw[5][8];
    r2l_rows_yp<G>(YP, tx, frow, yw);
    r2l_blur_row2(yw, F, ypp);
  }
  {
    float vw[3][6];
    r2l_rows_3x6<G>(V, tx, frow, vw);
    r2l_chroma_row2<PY>(vw, F, u, v);
  }
  const bool vec_ok = !RAGGED || (((a.W & 3) == 0) && (gx0 + 3 < a.W));
  r2l_p2 grgb[3][2];
  r2l_p2 ggam = r2l_splat2(0.f);
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    const unsigned off = (unsigned)k * plane + off0;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (PRE) {  // register select, not an indexed access (which would go through scratch)
      g[0] = r2l_pick(second, gpre.g[0][k].x, gpre.g[1][k].x);
      g[1] = r2l_pick(second, gpre.g[0][k].y, gpre.g[1][k].y);
      g[2] = r2l_pick(second, gpre.g[0][k].z, gpre.g[1][k].z);
      g[3] = r2l_pick(second, gpre.g[0][k].w, gpre.g[1][k].w);
    } else if (vec_ok) {
      // (with an output epilogue the frames are W % 4 == 0: this branch; gy = off0 / W)
      const r2l_f4 q = a.ep.on ? r2l_epi_load4(gb + (unsigned)k * plane + (a.ep.s0 + a.ep.sr * (int)((off0 - (unsigned)gx0) / (unsigned)a.W) + a.ep.sc * gx0), a.ep.sc)
                               : *(const r2l_f4*)(gb + off);
      g[0] = q.x;
      g[1] = q.y;
      g[2] = q.z;
      g[3] = q.w;
    } else {
      const float* o = a.ep.on ? gb + (unsigned)k * plane + (a.ep.s0 + a.ep.sr * (int)((off0 - (unsigned)gx0) / (unsigned)a.W) + a.ep.sc * gx0)
                               : gb + off;
      const int sc = a.ep.on ? a.ep.sc : 1;
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 4; ++c)
        if (gx0 + c < a.W) g[c] = o[c * sc];
    }
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) {
      const bool valid0 = !RAGGED || (gx0 + 2 * p < a.W), valid1 = !RAGGED || (gx0 + 2 * p + 1 < a.W);
      r2l_p2 rgb = r2l_pmul(r2l_splat2(F.M2[k * 3]), ypp[p]);
      rgb = r2l_pfma(r2l_splat2(F.M2[k * 3 + 1]), u[p], rgb);
      rgb = r2l_pfma(r2l_splat2(F.M2[k * 3 + 2]), v[p], rgb);
      const r2l_p2 xc = r2l_mk2(fminf(fmaxf(rgb[0], 1e-5f), 1.0f), fminf(fmaxf(rgb[1], 1e-5f), 1.0f));
      const r2l_p2 lg = r2l_mk2(r2l_log2(xc[0]), r2l_log2(xc[1]));
      const r2l_p2 e = r2l_pmul(lg, r2l_splat2(F.inv_gamma));
      const r2l_p2 og = r2l_mk2(r2l_exp2(e[0]), r2l_exp2(e[1]));
      r2l_p2 x = og;
      if (ADD) x = r2l_padd(x, r2l_mk2(valid0 ? a.additive[off + 2 * p] : 0.f, valid1 ? a.additive[off + 2 * p + 1] : 0.f));
      const r2l_p2 xhat = r2l_pmul(r2l_padd(x, r2l_splat2(-bc.mean[k])), r2l_splat2(bc.istd[k]));
      r2l_p2 gx = r2l_bn_bwd_pair(r2l_mk2(g[2 * p], g[2 * p + 1]), xhat, bc.istd[k], bc.mg[k], bc.mgx[k]);
      if (RAGGED) gx = r2l_mk2(valid0 ? gx[0] : 0.f, valid1 ? gx[1] : 0.f);
      const r2l_p2 gxo = r2l_pmul(gx, og);
      ggam = r2l_pfma(gxo, lg, ggam);
      const r2l_p2 gc = r2l_pmul(r2l_pmul(gxo, r2l_splat2(F.inv_gamma)), r2l_mk2(r2l_rcp(xc[0]), r2l_rcp(xc[1])));
      // torch.clip backward: the gradient passes where 1e-5 <= rgb <= 1, i.e. where clipping left rgb unchanged
      grgb[k][p] = r2l_mk2((rgb[0] == xc[0]) ? gc[0] : 0.f, (rgb[1] == xc[1]) ? gc[1] : 0.f);
    }
    if (RAGGED) R2L_SCHED_FENCE();  // one channel at a time: the general instantiations have no register to spare for overlap
  }
  r2l_p2 gy2[2], gu[2], gv[2];  // d loss / d (Y'', U, V)
  R2L_PRAGMA_UNROLL
  for (int p = 0; p < 2; ++p) {
    gy2[p] = r2l_pfma(r2l_splat2(F.M2[6]), grgb[2][p],
                      r2l_pfma(r2l_splat2(F.M2[3]), grgb[1][p], r2l_pmul(r2l_splat2(F.M2[0]), grgb[0][p])));
    gu[p] = r2l_pfma(r2l_splat2(F.M2[7]), grgb[2][p],
                     r2l_pfma(r2l_splat2(F.M2[4]), grgb[1][p], r2l_pmul(r2l_splat2(F.M2[1]), grgb[0][p])));
    gv[p] = r2l_pfma(r2l_splat2(F.M2[8]), grgb[2][p],
                     r2l_pfma(r2l_splat2(F.M2[5]), grgb[1][p], r2l_pmul(r2l_splat2(F.M2[2]), grgb[0][p])));
  }
  if (vec_ok) {
    r2l_f4 st;
    st.x = gy2[0][0];
    st.y = gy2[0][1];
    st.z = gy2[1][0];
    st.w = gy2[1][1];
    *(r2l_f4*)(gyb + off0) = st;
  } else {
    R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c)
      if (gx0 + c < a.W) gyb[off0 + c] = gy2[c >> 1][c & 1];
  }
  R2L_SCHED_FENCE();
  regs.acc[R2L_L1_GGAM] = r2l_padd(regs.acc[R2L_L1_GGAM], ggam);
  // d/d gaussian_blur.weight[i][j] = sum_p gY''(p) * YP_ext(p + (i-2, j-2)).  The two windows are read from
  // LDS a second time here rather than kept in 58 registers across the pointwise part (VGPR-bound kernel).
  float yw[5][8];
  r2l_rows_yp<G>(YP, tx, frow, yw);
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 5; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 5; ++j) {
    r2l_p2 s = regs.acc[R2L_L1_GBLUR + i * 5 + j];
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) s = r2l_pfma(gy2[p], r2l_mk2(yw[i][2 * p + j], yw[i][2 * p + j + 1]), s);
    regs.acc[R2L_L1_GBLUR + i * 5 + j] = s;
  }
  R2L_SCHED_FENCE();
  // folded chroma stencils: GA[par][t] = sum_{p of parity par} gU(p) * v_ext(p+t); pair half = column parity
  float vw[3][6];
  r2l_rows_3x6<G>(V, tx, frow, vw);
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 3; ++j) {
    r2l_p2 su = regs.acc[R2L_L1_GAU + i * 3 + j];
    r2l_p2 sv = regs.acc[R2L_L1_GAV + i * 3 + j];
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) {
      const r2l_p2 x = r2l_mk2(vw[i][2 * p + j], vw[i][2 * p + j + 1]);
      su = r2l_pfma(gu[p], x, su);
      sv = r2l_pfma(gv[p], x, sv);
    }
    regs.acc[R2L_L1_GAU + i * 3 + j] = su;
    regs.acc[R2L_L1_GAV + i * 3 + j] = sv;
  }
  regs.acc[R2L_L1_SU] = r2l_padd(regs.acc[R2L_L1_SU], r2l_padd(gu[0], gu[1]));
  regs.acc[R2L_L1_SV] = r2l_padd(regs.acc[R2L_L1_SV], r2l_padd(gv[0], gv[1]));
}

// Both rows of a thread (frame rows frow0 and frow0 + 2, the same Bayer row parity), STAGE by stage instead of row
// by row: the ~85 weights of a row body do not fit the scalar registers, so the row-by-row form re-loads them
// for every row (8 scalar-load waits per row, each a full `s_waitcnt lgkmcnt(0)` with one other wave per SIMD to
// hide behind); stage-major, every group of weights is fetched once per two rows.  Hot instantiation only (tiles
// inside the image, prefetched grad_out).
// mid(slot): 16 slots behind the window rows of the two gradient correlations, where the block issues the next tile's
// asynchronous copies one at a time (r2l_stage_frame).
template <class G, int PY, bool ADD, class MID>
R2L_HD void r2l_bwd1_rows2(const float* V, const float* YP, const R2LBwd1Args& a, int tid, int tx, int frow0,
                           unsigned off00, unsigned plane, const R2LGoutPre& gpre, float* gyb,
                           const R2LBnConsts& bc, R2LBwd1Regs& regs, MID&& mid R2L_SUB_ARG) {
  R2L_SUB_BEGIN
  R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque(a.F));
  r2l_p2 ypp[2][2], u[2][2], v[2][2];
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 2; ++r) {
    float yw[5][8];
    r2l_rows_yp<G>(YP, tx, frow0 + 2 * r, yw);
    r2l_blur_row2(yw, F, ypp[r]);
  }
  R2L_SCHED_FENCE();
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 2; ++r) {
    float vw[3][6];
    r2l_rows_3x6<G>(V, tx, frow0 + 2 * r, vw);
    r2l_chroma_row2<PY>(vw, F, u[r], v[r]);
  }
  R2L_SCHED_FENCE();
  R2L_SUB(1)
  r2l_p2 gy2[2][2], gu[2][2], gv[2][2];  // d loss / d (Y'', U, V)
  r2l_p2 ggam = r2l_splat2(0.f);
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 2; ++r) {
    const unsigned off0 = off00 + (unsigned)(2 * r) * (unsigned)a.W;
    r2l_p2 grgb[3][2];
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      const unsigned off = (unsigned)k * plane + off0;
      const float g[4] = {gpre.g[r][k].x, gpre.g[r][k].y, gpre.g[r][k].z, gpre.g[r][k].w};
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) {
        r2l_p2 rgb = r2l_pmul(r2l_splat2(F.M2[k * 3]), ypp[r][p]);
        rgb = r2l_pfma(r2l_splat2(F.M2[k * 3 + 1]), u[r][p], rgb);
        rgb = r2l_pfma(r2l_splat2(F.M2[k * 3 + 2]), v[r][p], rgb);
        const r2l_p2 xc = r2l_mk2(fminf(fmaxf(rgb[0], 1e-5f), 1.0f), fminf(fmaxf(rgb[1], 1e-5f), 1.0f));
        const r2l_p2 lg = r2l_mk2(r2l_log2(xc[0]), r2l_log2(xc[1]));
        const r2l_p2 e = r2l_pmul(lg, r2l_splat2(F.inv_gamma));
        const r2l_p2 og = r2l_mk2(r2l_exp2(e[0]), r2l_exp2(e[1]));
        r2l_p2 x = og;
        if (ADD) x = r2l_padd(x, r2l_mk2(a.additive[off + 2 * p], a.additive[off + 2 * p + 1]));
        const r2l_p2 xhat = r2l_pmul(r2l_padd(x, r2l_splat2(-bc.mean[k])), r2l_splat2(bc.istd[k]));
        r2l_p2 gx = r2l_bn_bwd_pair(r2l_mk2(g[2 * p], g[2 * p + 1]), xhat, bc.istd[k], bc.mg[k], bc.mgx[k]);
        const r2l_p2 gxo = r2l_pmul(gx, og);
        ggam = r2l_pfma(gxo, lg, ggam);
        const r2l_p2 gc = r2l_pmul(r2l_pmul(gxo, r2l_splat2(F.inv_gamma)), r2l_mk2(r2l_rcp(xc[0]), r2l_rcp(xc[1])));
        grgb[k][p] = r2l_mk2((rgb[0] == xc[0]) ? gc[0] : 0.f, (rgb[1] == xc[1]) ? gc[1] : 0.f);  // clip backward
      }
    }
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) {
      gy2[r][p] = r2l_pfma(r2l_splat2(F.M2[6]), grgb[2][p],
                           r2l_pfma(r2l_splat2(F.M2[3]), grgb[1][p], r2l_pmul(r2l_splat2(F.M2[0]), grgb[0][p])));
      gu[r][p] = r2l_pfma(r2l_splat2(F.M2[7]), grgb[2][p],
                          r2l_pfma(r2l_splat2(F.M2[4]), grgb[1][p], r2l_pmul(r2l_splat2(F.M2[1]), grgb[0][p])));
      gv[r][p] = r2l_pfma(r2l_splat2(F.M2[8]), grgb[2][p],
                          r2l_pfma(r2l_splat2(F.M2[5]), grgb[1][p], r2l_pmul(r2l_splat2(F.M2[2]), grgb[0][p])));
    }
    r2l_f4 st;
    st.x = gy2[r][0][0];
    st.y = gy2[r][0][1];
    st.z = gy2[r][1][0];
    st.w = gy2[r][1][1];
    *(r2l_f4*)(gyb + off0) = st;
  }
  R2L_SCHED_FENCE();
  R2L_SUB(2)
  regs.acc[R2L_L1_GGAM] = r2l_padd(regs.acc[R2L_L1_GGAM], ggam);
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 2; ++r) {  // d/d gaussian_blur.weight (windows read from LDS a second time, see r2l_bwd1_row)
    float yw[5][8];
    r2l_rows_yp<G>(YP, tx, frow0 + 2 * r, yw);
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 5; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 5; ++j) {
      r2l_p2 sacc = regs.acc[R2L_L1_GBLUR + i * 5 + j];
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) sacc = r2l_pfma(gy2[r][p], r2l_mk2(yw[i][2 * p + j], yw[i][2 * p + j + 1]), sacc);
      regs.acc[R2L_L1_GBLUR + i * 5 + j] = sacc;
      if (j == 4) mid(r * 5 + i);  // one of the next tile's loads behind every window row (10 slots here, 6 below)
    }
    R2L_SCHED_FENCE();
  }
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 2; ++r) {  // folded chroma stencils
    float vw[3][6];
    r2l_rows_3x6<G>(V, tx, frow0 + 2 * r, vw);
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      r2l_p2 su = regs.acc[R2L_L1_GAU + i * 3 + j];
      r2l_p2 sv = regs.acc[R2L_L1_GAV + i * 3 + j];
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) {
        const r2l_p2 x = r2l_mk2(vw[i][2 * p + j], vw[i][2 * p + j + 1]);
        su = r2l_pfma(gu[r][p], x, su);
        sv = r2l_pfma(gv[r][p], x, sv);
      }
      regs.acc[R2L_L1_GAU + i * 3 + j] = su;
      regs.acc[R2L_L1_GAV + i * 3 + j] = sv;
      if (j == 2) mid(10 + r * 3 + i);
    }
    regs.acc[R2L_L1_SU] = r2l_padd(regs.acc[R2L_L1_SU], r2l_padd(gu[r][0], gu[r][1]));
    regs.acc[R2L_L1_SV] = r2l_padd(regs.acc[R2L_L1_SV], r2l_padd(gv[r][0], gv[r][1]));
    R2L_SCHED_FENCE();
  }
}

template <class G, bool RAGGED, bool ADD, bool PRE, class MID>
R2L_HD void r2l_bwd1_pixels(int tid, const float* V, const float* YP, const R2LBwd1Args& a,
                            const R2LTile& t, const R2LGoutPre& gp, R2LBwd1Regs& regs, MID&& mid R2L_SUB_ARG) {
  int tx, row0, py;
  G::thread_tile(tid, tx, row0, py);
  const int gy0 = t.oy + row0, gx0 = t.ox + 4 * tx;
  if (RAGGED && (gy0 >= a.H || gx0 >= a.W)) return;
  const unsigned plane = (unsigned)a.H * (unsigned)a.W;
  const float* gb = a.gout + (size_t)t.b * 3 * plane;
  float* gyb = a.gypp + (size_t)t.b * plane;
  const unsigned pix0 = (unsigned)gy0 * (unsigned)a.W + (unsigned)gx0;
  R2LBnConsts bc;
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    bc.mean[k] = a.bn ? a.bn[k] : 0.f;
    bc.istd[k] = a.bn ? a.bn[3 + k] : 1.f;
    bc.mg[k] = a.bn_bwd ? bc.istd[k] * a.bn_bwd[k] : 0.f;
    bc.mgx[k] = a.bn_bwd ? bc.istd[k] * a.bn_bwd[3 + k] : 0.f;
  }
#ifndef R2L_B1_ROW_MAJOR
  if (!RAGGED && PRE) {
    if (py)
      r2l_bwd1_rows2<G, 1, ADD>(V, YP, a, tid, tx, row0 + 4, pix0, plane, gp, gyb, bc, regs, mid R2L_SUB_PASS_FWD);
    else
      r2l_bwd1_rows2<G, 0, ADD>(V, YP, a, tid, tx, row0 + 4, pix0, plane, gp, gyb, bc, regs, mid R2L_SUB_PASS_FWD);
    return;
  }
#endif
  R2L_PRAGMA_UNROLL
  for (int sl = 0; sl < 16; ++sl) mid(sl);
  R2L_PRAGMA_NOUNROLL
  for (int rr = 0; rr < 4; rr += 2) {
    if (RAGGED && gy0 + rr >= a.H) break;
    const unsigned off0 = pix0 + (unsigned)rr * (unsigned)a.W;
    if (py)
      r2l_bwd1_row<G, 1, RAGGED, ADD, PRE>(V, YP, a, tx, row0 + 4 + rr, gx0, off0, plane, gb, gp, rr != 0, gyb,
                                           bc, regs);
    else
      r2l_bwd1_row<G, 0, RAGGED, ADD, PRE>(V, YP, a, tx, row0 + 4 + rr, gx0, off0, plane, gb, gp, rr != 0, gyb,
                                           bc, regs);
  }
}

// SAVED: the forward kept Y' (a.yp): its frame is prefetched like the raw frame and stored to the YP plane (zero outside
// the image, then the mirror fill of border tiles) instead of being recomputed raw -> Y -> Y' -- two of the five
// phases, 29 % of the kernel (profiles/r01_f_phase_stamps.txt), for 4.5 B/px more traffic.
template <class G, bool ADD, bool MAYBE_RAGGED, bool U16, bool SAVED = false>
R2L_BLOCKFN void r2l_bwd1_block(const R2LBwd1Args& a, int bid, int nblk, float* lds) {
  float* V = lds + R2L_FOLDED_FLOATS + G::PAD;
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  float* Y = V + G::PLANE;
  float* YP = Y + G::PLANE;
  R2L_TREG_DECL(R2LBwd1Regs, regs);
  R2L_TREG_DECL(R2LPrefetch<G>, pre);
  R2L_TREG_DECL(R2LPrefetch<G>, pre_yp);
  R2L_TREG_DECL(R2LGoutPre, gpre);
  // hot instantiations: the next tile's Y' frame travels through an LDS staging area behind the three planes
  // (r2l_stage_frame; these kernels are launched with R2L_B1_FRAME_FLOATS more LDS)
  constexpr bool GLDS = SAVED && !MAYBE_RAGGED && !ADD && (R2L_B1_GLDS != 0);
  static_assert(!GLDS || G::FH * G::FW + 256 == R2L_B1_FRAME_FLOATS, "staging area of a frame");
  float* YS = GLDS ? YP + G::PLANE + G::PAD : nullptr;
  // additive layer on frames that do not tile by 64 (never the reference's: its layer is 256 x 256, pipeline_torch.py:130): no
  // register to spare across the pixel phase -- the raw frame of a tile is fetched when the tile starts, not a tile ahead
  constexpr bool LATE_RAW = ADD && MAYBE_RAGGED;
  R2LTileWalk w = r2l_walk_init(a.B, a.H, a.W, G::TW, G::TH, bid, nblk);
  R2LTile t, tn;
  bool have = r2l_walk_next(w, a.H, a.W, G::TW, G::TH, t);
  R2L_PHASE_BEGIN
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < R2L_L1_NACC; ++i) R2L_TREG(regs).acc[i] = r2l_splat2(0.f);
  {
    int tx_, row_;
    G::thread_tile(tid, tx_, row_, R2L_TREG(regs).py);
  }
  if (!LATE_RAW && have) r2l_fetch_raw_tile<G, U16>(tid, a.raw, t, a.H, a.W, R2L_TREG(pre));
  if (SAVED && !GLDS && have) r2l_fetch_tile<G, 1>(tid, a.yp, t, a.H, a.W, R2L_TREG(pre_yp));
  if (GLDS && have) {
    r2l_stage_frame<G>(tid, a.yp + (size_t)t.b * a.H * a.W, t.oy, t.ox, a.H, a.W, YS);
    r2l_glds_wait();  // this wave's copies have landed; behind the phase barrier every wave's have
  }
  R2L_PHASE_END
  R2L_STAMP_DECL
  while (have) {
    R2L_PHASE_BEGIN_IF(ADD || MAYBE_RAGGED)
    if (LATE_RAW) r2l_fetch_raw_tile<G, U16>(tid, a.raw, t, a.H, a.W, R2L_TREG(pre));  // (no prefetch: see LATE_RAW)
    r2l_store_v<G, U16>(tid, V, F, R2L_TREG(pre), a.raw);
    if (GLDS)
      r2l_store_plane_s2_staged<G>(tid, YP, YS, t.oy, t.ox, a.H, a.W);
    else if (SAVED)
      r2l_store_plane_s2<G>(tid, YP, R2L_TREG(pre_yp));
    // (measured: the grad_out loads issued here, 119 us; a tile ahead at the head of the pixel phase, where 18 loads
    // per lane then queue up in the texture-address path, 133 us; a tile ahead behind the pixel arithmetic, 124 us;
    // a tile ahead from this store phase, held across the pixel phase, 136 us; a tile ahead through an LDS staging
    // area like the Y' frame, 115 us -- but the step as a whole slower, profiles/r03_b1_staging_modes.txt)
    if (SAVED && !MAYBE_RAGGED) {
      r2l_bwd1_fetch_gout<G>(tid, a, t, R2L_TREG(gpre), 0);
      r2l_bwd1_fetch_gout<G>(tid, a, t, R2L_TREG(gpre), 1);
    }
    R2L_PHASE_END
    R2L_STAMP(0)
    const bool haven = r2l_walk_next(w, a.H, a.W, G::TW, G::TH, tn);
    if (!SAVED) {
    R2L_PHASE_BEGIN_IF(ADD || MAYBE_RAGGED)
    // consumed in the pixel phase; one row per stencil phase (six loads per lane at once queue up in the
    // texture-address path and hold every wave at its next instruction)
    if (!MAYBE_RAGGED) r2l_bwd1_fetch_gout<G>(tid, a, t, R2L_TREG(gpre), 0);
    if (t.border)
      r2l_compute_y<G, true>(tid, V, Y, F, t.oy, t.ox, a.H, a.W);
    else
      r2l_compute_y<G, false>(tid, V, Y, F, t.oy, t.ox, a.H, a.W);
    R2L_PHASE_END
    R2L_STAMP(1)
    R2L_PHASE_BEGIN_IF(ADD || MAYBE_RAGGED)
    if (!MAYBE_RAGGED) r2l_bwd1_fetch_gout<G>(tid, a, t, R2L_TREG(gpre), 1);
    r2l_compute_yp<G>(tid, Y, YP, F);
    R2L_PHASE_END
    R2L_STAMP(2)
    }
    if (t.border) {
      R2L_PHASE_BEGIN_IF(ADD || MAYBE_RAGGED)
      r2l_fill_yp_mirror<G>(tid, YP, t.oy, t.ox, a.H, a.W);
      R2L_PHASE_END
    R2L_STAMP(3)
    }
    R2L_PHASE_BEGIN_IF(ADD || MAYBE_RAGGED)
    if (!LATE_RAW && haven) r2l_fetch_raw_tile<G, U16>(tid, a.raw, tn, a.H, a.W, R2L_TREG(pre));
    if (GLDS) {
      // the next tile's Y' frame: one copy per lane behind each of the first window rows of the blur-weight correlation
      // (as a burst all 8 wavefronts would reach them together and queue in the texture-address path).  After the last
      // tile the copies run once more on the tile itself: cheaper than a branch per slot.
      const R2LTile tq = haven ? tn : t;
      auto mid = [&](int slot) {
        if (slot < R2LPrefetch<G>::NIT)
          r2l_stage_frame<G>(tid, a.yp + (size_t)tq.b * a.H * a.W, tq.oy, tq.ox, a.H, a.W, YS, slot);
      };
      r2l_bwd1_pixels<G, false, ADD, true>(tid, V, YP, a, t, R2L_TREG(gpre), R2L_TREG(regs), mid R2L_SUB_PASS);
      r2l_glds_wait();  // this wave's copies have landed; behind the phase barrier every wave's have
    } else {
      auto nomid = [](int) {};
      if (MAYBE_RAGGED)
        r2l_bwd1_pixels<G, MAYBE_RAGGED, ADD, false>(tid, V, YP, a, t, R2L_TREG(gpre), R2L_TREG(regs), nomid R2L_SUB_PASS);
      else
        r2l_bwd1_pixels<G, false, ADD, !MAYBE_RAGGED>(tid, V, YP, a, t, R2L_TREG(gpre), R2L_TREG(regs), nomid R2L_SUB_PASS);
#ifndef R2L_EMUL
      asm volatile("" ::: "memory");
#endif
      if (SAVED && haven) r2l_fetch_tile<G, 1>(tid, a.yp, tn, a.H, a.W, R2L_TREG(pre_yp));
    }
    R2L_PHASE_END
    R2L_STAMP(4)
    t = tn;
    have = haven;
  }
  R2L_STAMP_FLUSH(a.debug, bid)
  R2L_BLOCK_REDUCE_F(R2L_B1_NACC, R2L_ACC_B1, regs, lds, a.partial, bid, nblk)
}

// ================================================================================================
// backward, kernel B2: adjoint of blur (mirror pad) and sharpen (zero pad) on the luma plane
// ================================================================================================
struct R2LBwd2Args {
  R2LRaw raw;
  const R2LFolded* F;
  const float* gypp;  // (B,H,W) from B1
  float* partial;     // [R2L_B2_NACC][nblk]
  int B, H, W;
  float* debug;
  R2LTree tree;         // in-kernel final reduction of B1's and B2's partials + unfold -> grad_params
  const float* params;  // packed parameters (for the unfold)
  float* grad_params;   // [R2L_P_NTRAIN]
  int asym;             // r2l_walk_init: uneven tile shares for the two workgroups of a CU (0 = even)
  float* hp;            // plane passes (r2l_param_plane_bwd.h): (B,H,W) the blur's adjoint of dL/dY''
  int band_h;           // plane passes: rows per work item (a multiple of 6)
  // the sums pass (r2l_bwd2_sums_block) with helper workgroups: workgroups [0, nmain) walk the work items and reduce B2's
  // partials (tree.partial2, tree.split = 0); workgroups nmain + h add B1's partials b1_partial[slot][b1_n] of slots
  // 8 h ... 8 h + 7 into b1_tot[slot] while the others compute
  int nmain;
  const float* b1_partial;
  int b1_n;
  double* b1_tot;  // [R2L_B1_NACC]
  int xcdm;        // the sums pass: neighbouring workgroups per XCD (r2l_xcd_window; 0 = off)
};

enum { R2L_L2_GSHARP = 0, R2L_L2_GAY = 9, R2L_L2_SY = 27, R2L_L2_NACC = 29 };
struct R2LBwd2Regs {
  float acc[R2L_L2_NACC];
  int py;
};
R2L_HD float r2l_b2_slot(const R2LBwd2Regs& r, int i) {
  if (i < R2L_B2_GAY) return r.acc[R2L_L2_GSHARP + i];
  if (i < R2L_B2_SY) {
    const int k = i - R2L_B2_GAY, par = k / 9, t = k % 9;
    return ((par >> 1) == r.py) ? r.acc[R2L_L2_GAY + (par & 1) * 9 + t] : 0.f;
  }
  const int par = i - R2L_B2_SY;
  return ((par >> 1) == r.py) ? r.acc[R2L_L2_SY + (par & 1)] : 0.f;
}
#define R2L_ACC_B2(regs, i) r2l_b2_slot(R2L_TREG(regs), i)

// phase: HP(q') = sum_t blur[t] * G2_ext0(q' - t) on frame rows/cols [2, F-2); G2 is stored shifted
template <class G>
R2L_HD void r2l_adjoint_blur(int tid, const float* G2, float* HP, R2LFoldedRef F) {
  constexpr int CPR = G::FW / 4, RPI = R2L_RPI_ADJ, NRG = (G::FH - 4 + RPI - 1) / RPI;
  static_assert(CPR * NRG <= R2L_NT, "one pass");
  const int rg = tid / CPR, cx = tid - rg * CPR;
  if (rg >= NRG) return;
  const int fy0 = 2 + RPI * rg, fx = 4 * cx;
  float w[RPI + 4][8];  // rows fy0-2..fy0+RPI+1, cols fx-2..fx+5
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < RPI + 4; ++i) {
    // rows past the end of the plane (ragged last row group) are clamped: they only feed rows that are not stored
    const int fy = (fy0 - 2 + i < G::FH - 1) ? fy0 - 2 + i : G::FH - 1;
    const float* r = G2 + fy * G::FS + fx;  // (+2 shift) - 2
    const r2l_f4 a = r2l_lds_f4(r);
    const r2l_f4 b = r2l_lds_f4(r + 4);
    w[i][0] = a.x;
    w[i][1] = a.y;
    w[i][2] = a.z;
    w[i][3] = a.w;
    w[i][4] = b.x;
    w[i][5] = b.y;
    w[i][6] = b.z;
    w[i][7] = b.w;
  }
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < RPI; ++r) {
    r2l_p2 o[2];
    o[0] = o[1] = r2l_splat2(0.f);
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 5; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 5; ++j) {  // source p = q' - (i-2, j-2)  ->  window index (r+4-i, c+4-j)
      const r2l_p2 wb = r2l_splat2(F.blur[i * 5 + j]);
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p)
        o[p] = r2l_pfma(wb, r2l_mk2(w[r + 4 - i][2 * p + 4 - j], w[r + 4 - i][2 * p + 5 - j]), o[p]);
    }
    r2l_f4 st;
    st.x = o[0][0];
    st.y = o[0][1];
    st.z = o[1][0];
    st.w = o[1][1];
    if ((G::FH - 4) % RPI == 0 || fy0 + r < G::FH - 2) *(r2l_f4*)(HP + (fy0 + r) * G::FS + fx) = st;  // ragged group
  }
}

// A frame whose image ends exactly 2 rows (columns) below (right of) its tile -- H % TH == 2 -- owns image
// rows H-3 and H-2 as tile row TH-1 and sharpen halo, and their mirror images H+1, H sit in frame rows
// FH-1, FH-2, which the pass above does not cover.  Those two rows (columns) are computed here, tap by tap;
// frame positions beyond the frame are outside the image there, where dL/dY'' is zero.
template <class G>
R2L_HD bool r2l_tail_rows(int oy, int H) { return H - oy == G::TH + 2; }
template <class G>
R2L_HD bool r2l_tail_cols(int ox, int W) { return W - ox == G::TW + 2; }
template <class G>
R2L_HD float r2l_adjoint_blur_one(const float* G2, R2LFoldedRef F, int fy, int fx) {
  float s = 0.f;
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) {
      const int sy = fy - (i - 2), sx = fx - (j - 2);
      if (sy >= 0 && sy < G::FH && sx >= 0 && sx < G::FW) s = fmaf(F.blur[i * 5 + j], G2[sy * G::FS + sx + 2], s);
    }
  return s;
}
template <class G>
R2L_HD void r2l_adjoint_blur_tail(int tid, const float* G2, float* HP, R2LFoldedRef F, int oy, int ox, int H, int W) {
  const bool rows = r2l_tail_rows<G>(oy, H), cols = r2l_tail_cols<G>(ox, W);
  if (rows) {
    const int xlim = cols ? G::FW : G::FW - 2;
    for (int i = tid; i < 2 * (xlim - 2); i += R2L_NT) {
      const int fy = G::FH - 2 + i / (xlim - 2), fx = 2 + i % (xlim - 2);
      HP[fy * G::FS + fx] = r2l_adjoint_blur_one<G>(G2, F, fy, fx);
    }
  }
  if (cols)
    for (int i = tid; i < 2 * (G::FH - 4); i += R2L_NT) {
      const int fx = G::FW - 2 + i / (G::FH - 4), fy = 2 + i % (G::FH - 4);
      HP[fy * G::FS + fx] = r2l_adjoint_blur_one<G>(G2, F, fy, fx);
    }
}

// phase (border tiles): fold the contributions that the mirror padding sent outside the image back
// onto their sources, IN PLACE: an in-image position adds the values of its out-of-image mirror images
// (which nobody writes in this phase).  Out-of-image entries keep their values; the pixel phase masks
// them (the zero padding of the sharpen conv has no adjoint contribution there).
template <class G>
R2L_HD void r2l_fold_one(float* HP, int fy, int fx, int oy, int ox, int H, int W) {
  const int gy = oy - 4 + fy, gx = ox - 4 + fx;
  if (!((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)) return;
  int ey[3], ex[3];
  ey[0] = fy;
  ey[1] = (gy >= 1 && gy <= 2) ? fy - 2 * gy : -1;                    // image row -gy
  ey[2] = (gy >= H - 3 && gy <= H - 2) ? fy + 2 * (H - 1 - gy) : -1;  // image row 2(H-1)-gy
  ex[0] = fx;
  ex[1] = (gx >= 1 && gx <= 2) ? fx - 2 * gx : -1;
  ex[2] = (gx >= W - 3 && gx <= W - 2) ? fx + 2 * (W - 1 - gx) : -1;
  // rows / columns of HP that hold adjoint-blur values: [2, F-2), plus the two tail rows / columns
  const int ylim = r2l_tail_rows<G>(oy, H) ? G::FH : G::FH - 2, xlim = r2l_tail_cols<G>(ox, W) ? G::FW : G::FW - 2;
  float s = 0.f;
  R2L_PRAGMA_UNROLL
  for (int p = 0; p < 3; ++p)
    R2L_PRAGMA_UNROLL
  for (int q = 0; q < 3; ++q)
    if ((p | q) != 0 && ey[p] >= 2 && ey[p] < ylim && ex[q] >= 2 && ex[q] < xlim)
      s += HP[ey[p] * G::FS + ex[q]];
  HP[fy * G::FS + fx] += s;
}
// Only image rows / columns {1, 2, n-3, n-2} have mirror images: visit those columns over all rows, then
// those rows over the remaining columns (each position once).
template <class G>
R2L_HD void r2l_fold_mirror(int tid, float* HP, int oy, int ox, int H, int W) {
  constexpr int NW = G::FW - 4, NH = G::FH - 4;
  const int gc[4] = {1, 2, W - 3, W - 2}, gr[4] = {1, 2, H - 3, H - 2};
  for (int i = tid; i < 4 * NH; i += R2L_NT) {
    const int k = i / NH, fy = 2 + (i - k * NH);
    const int fx = gc[k] - ox + 4;
    const bool dup = (k >= 2) && (gc[k] == gc[k - 2]);  // W == 4: columns 1, 2 listed twice
    if (!dup && fx >= 2 && fx < G::FW - 2) r2l_fold_one<G>(HP, fy, fx, oy, ox, H, W);
  }
  for (int i = tid; i < 4 * NW; i += R2L_NT) {
    const int k = i / NW, fx = 2 + (i - k * NW);
    const int fy = gr[k] - oy + 4;
    const int gx = ox - 4 + fx;
    const bool dup = (k >= 2) && (gr[k] == gr[k - 2]);
    const bool colcand = gx == 1 || gx == 2 || gx == W - 3 || gx == W - 2;  // done by the first loop
    if (!dup && !colcand && fy >= 2 && fy < G::FH - 2) r2l_fold_one<G>(HP, fy, fx, oy, ox, H, W);
  }
}

// one output row of kernel B2; PY = row parity; BORDER: mask adjoint values outside the image
template <class G, int PY, bool BORDER>
R2L_HD void r2l_bwd2_row(const float* V, const float* Y, const float* HP, const R2LBwd2Args& a, int tx,
                         int frow, int gy, int gx0, R2LBwd2Regs& regs) {
  R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque(a.F));
  // one window at a time (HP, then Y, then V), each consumed before the next is read: at 128 VGPRs (two workgroups
  // per CU) there is no room for the three of them side by side
  float gy1[4];  // d loss / d Y (pre-sharpen luma)
  float gyp[4];  // d loss / d Y' at the 4 pixels
  {
    float hw[3][6];
    r2l_rows_3x6<G>(HP, tx, frow, hw);
    if (BORDER) {
      R2L_PRAGMA_UNROLL
      for (int i = 0; i < 3; ++i) {
        const bool rin = (unsigned)(gy - 1 + i) < (unsigned)a.H;
        R2L_PRAGMA_UNROLL
        for (int j = 0; j < 6; ++j)
          hw[i][j] = (rin && (unsigned)(gx0 - 1 + j) < (unsigned)a.W) ? hw[i][j] : 0.f;
      }
    }
    R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c) {
      float s = 0.f;
      R2L_PRAGMA_UNROLL
      for (int i = 0; i < 3; ++i)
        R2L_PRAGMA_UNROLL
      for (int j = 0; j < 3; ++j)  // source q = p - (i-1, j-1) -> window index (2-i, c+2-j)
        s = fmaf(F.sharp[i * 3 + j], hw[2 - i][c + 2 - j], s);
      const bool valid = !BORDER || ((unsigned)gy < (unsigned)a.H && (unsigned)(gx0 + c) < (unsigned)a.W);
      gy1[c] = valid ? s : 0.f;
      gyp[c] = hw[1][c + 1];
    }
  }
  R2L_SCHED_FENCE();
  {
    // d/d sharpening_filter.weight[i][j] = sum_p gY'(p) * Y_zero_ext(p + (i-1, j-1))
    float yw[3][6];
    r2l_rows_3x6<G>(Y, tx, frow, yw);
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      float s = regs.acc[R2L_L2_GSHARP + i * 3 + j];
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 4; ++c) s = fmaf(gyp[c], yw[i][c + j], s);
      regs.acc[R2L_L2_GSHARP + i * 3 + j] = s;
    }
  }
  R2L_SCHED_FENCE();
  float vw[3][6];
  r2l_rows_3x6<G>(V, tx, frow, vw);
  R2L_PRAGMA_UNROLL
  for (int px = 0; px < 2; ++px) {
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      float s = regs.acc[R2L_L2_GAY + px * 9 + i * 3 + j];
      R2L_PRAGMA_UNROLL
      for (int c = px; c < 4; c += 2) s = fmaf(gy1[c], vw[i][c + j], s);
      regs.acc[R2L_L2_GAY + px * 9 + i * 3 + j] = s;
    }
    regs.acc[R2L_L2_SY + px] += gy1[px] + gy1[px + 2];
  }
}

template <class G, bool BORDER>
R2L_HD void r2l_bwd2_pixels(int tid, const float* V, const float* Y, const float* HP,
                            const R2LBwd2Args& a, const R2LTile& t, R2LBwd2Regs& regs) {
  int tx, row0, py;
  G::thread_tile(tid, tx, row0, py);
  const int gy0 = t.oy + row0, gx0 = t.ox + 4 * tx;
  if (BORDER && (gy0 >= a.H || gx0 >= a.W)) return;
  R2L_PRAGMA_NOUNROLL
  for (int rr = 0; rr < 4; rr += 2) {
    if (py)
      r2l_bwd2_row<G, 1, BORDER>(V, Y, HP, a, tx, row0 + 4 + rr, gy0 + rr, gx0, regs);
    else
      r2l_bwd2_row<G, 0, BORDER>(V, Y, HP, a, tx, row0 + 4 + rr, gy0 + rr, gx0, regs);
  }
}

// bwd2 runs TWO workgroups per CU (-DR2L_OCC_BWD2=4, the default): that needs <= 128 VGPRs, which it gets from
// (i) no software prefetch of the next tile (24 registers; the other workgroup's phases hide the HBM round trip),
// (ii) 3-row items in the adjoint blur, (iii) one register window at a time in the pixel phase, (iv) R2L_PHASE_BEGIN_L.
// profiles/r02_c_bwd2_ab.txt: 109 us (round 1) -> 116 us with (iii)+(iv) alone at one workgroup per CU -> 97 us.
#ifndef R2L_B2_PREFETCH
#define R2L_B2_PREFETCH 0
#endif

template <class G, bool U16>
R2L_BLOCKFN void r2l_bwd2_block(const R2LBwd2Args& a, int bid, int nblk, float* lds) {
  float* V = lds + R2L_FOLDED_FLOATS + G::PAD;
  float* G2 = V + G::PLANE;  // dL/dY'' (shifted); dead after the adjoint blur, then holds Y
  float* HP = G2 + G::PLANE;
  float* Y = G2;
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  R2L_TREG_DECL(R2LBwd2Regs, regs);
  R2L_TREG_DECL(R2LPrefetch<G>, pre_v);
  R2L_TREG_DECL(R2LPrefetch<G>, pre_g);
  R2LTileWalk w = r2l_walk_init(a.B, a.H, a.W, G::TW, G::TH, bid, nblk, a.asym);
  R2LTile t, tn;
  bool have = r2l_walk_next(w, a.H, a.W, G::TW, G::TH, t);
  R2L_PHASE_BEGIN
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < R2L_L2_NACC; ++i) R2L_TREG(regs).acc[i] = 0.f;
  {
    int tx_, row_;
    G::thread_tile(tid, tx_, row_, R2L_TREG(regs).py);
  }
#if R2L_B2_PREFETCH
  if (have) {
    r2l_fetch_raw_tile<G, U16>(tid, a.raw, t, a.H, a.W, R2L_TREG(pre_v));
    r2l_fetch_tile<G, 1>(tid, a.gypp, t, a.H, a.W, R2L_TREG(pre_g));
  }
#endif
  R2L_PHASE_END
  R2L_STAMP_DECL
  while (have) {
    R2L_PHASE_BEGIN_L
#if !R2L_B2_PREFETCH
    // two workgroups per CU: the other workgroup's phases hide this tile's HBM round trip, and the 24 registers
    // of a software prefetch are what keeps the kernel above 128 VGPRs
    r2l_fetch_raw_tile<G, U16>(tid, a.raw, t, a.H, a.W, R2L_TREG(pre_v));
    r2l_fetch_tile<G, 1>(tid, a.gypp, t, a.H, a.W, R2L_TREG(pre_g));
#endif
    r2l_store_v<G, U16>(tid, V, F, R2L_TREG(pre_v), a.raw);
    r2l_store_plane_s2<G>(tid, G2, R2L_TREG(pre_g));
    R2L_PHASE_END
    R2L_STAMP(0)
    const bool haven = r2l_walk_next(w, a.H, a.W, G::TW, G::TH, tn);
    R2L_PHASE_BEGIN_L
    r2l_adjoint_blur<G>(tid, G2, HP, F);
    R2L_PHASE_END
    if (r2l_tail_rows<G>(t.oy, a.H) || r2l_tail_cols<G>(t.ox, a.W)) {  // uniform over the workgroup
      R2L_PHASE_BEGIN_L
      r2l_adjoint_blur_tail<G>(tid, G2, HP, F, t.oy, t.ox, a.H, a.W);
      R2L_PHASE_END
    }
    R2L_STAMP(1)
    R2L_PHASE_BEGIN_L
    if (t.border) {
      r2l_compute_y<G, true>(tid, V, Y, F, t.oy, t.ox, a.H, a.W);
      r2l_fold_mirror<G>(tid, HP, t.oy, t.ox, a.H, a.W);
    } else {
      r2l_compute_y<G, false>(tid, V, Y, F, t.oy, t.ox, a.H, a.W);
    }
    R2L_PHASE_END
    R2L_STAMP(2)
    R2L_PHASE_BEGIN_L
#if R2L_B2_PREFETCH
    if (haven) {
      r2l_fetch_raw_tile<G, U16>(tid, a.raw, tn, a.H, a.W, R2L_TREG(pre_v));
      r2l_fetch_tile<G, 1>(tid, a.gypp, tn, a.H, a.W, R2L_TREG(pre_g));
    }
#endif
    if (t.border)
      r2l_bwd2_pixels<G, true>(tid, V, Y, HP, a, t, R2L_TREG(regs));
    else
      r2l_bwd2_pixels<G, false>(tid, V, Y, HP, a, t, R2L_TREG(regs));
    R2L_PHASE_END
    R2L_STAMP(3)
    t = tn;
    have = haven;
  }
  R2L_BLOCK_REDUCE_F(R2L_B2_NACC, R2L_ACC_B2, regs, lds, a.partial, bid, nblk)
  R2L_STAMP(4)
  R2L_STAMP_FLUSH(a.debug, bid)
  if (a.tree.counters) {
    // LDS scratch of the epilogue: 4 floats of tickets, then float64 sums[R2L_NSUMS], T[9] + gT[9], and the
    // packed parameters as floats (all below float 512, where the staging area of the reduction starts)
    double* sums = (double*)(lds + 4);
    double* tg = sums + R2L_NSUMS;
    float* pl = (float*)(tg + R2L_UNFOLD_TG);
    const bool last_ = r2l_tree_finish<R2L_NSUMS>(a.tree, bid, nblk, lds, sums, (double*)(lds + 512),
                                                  (R2L_RED_FLOATS - 512) / 2);
    R2L_STAMP(5)
    R2L_STAMP_FLUSH(a.debug, bid)
    if (!last_) return;
#if defined(R2L_TEST_HOOKS) && !defined(R2L_EMUL)
    if (a.debug && threadIdx.x < R2L_NSUMS) ((double*)a.debug)[threadIdx.x] = sums[threadIdx.x];  // (tests: the 155 totals)
#endif
    r2l_unfold_phases(a.params, sums, tg, pl, a.grad_params);
    R2L_STAMP(6)
    R2L_STAMP_FLUSH(a.debug, bid)
  }
}
