// r2l_param_stream.h -- the parametrized forward (ParametrizedProcessing.forward, pipeline_torch.py:175-225) as a
// ROW-STREAMING kernel: the same arithmetic as r2l_fwd_block (r2l_param_kernels.h), organised like the static
// luma-chain kernel (r2l_static_chain.h) instead of as LDS tiles.
//
// A wavefront owns a strip of 256 columns (4 per lane) and a band of rows and walks down the band with, in
// registers (all float32, two horizontally adjacent pixels per packed instruction):
//     V   the last 3 rows of black-level-corrected raw values, 6 wide          -> Y, U, V of the middle row (:183-194)
//     Y   the last 3 luma rows, 6 wide (zero outside the image, :195 padding=1)  -> Y' = sharpen(Y)
//     Y'  a ring of 6 sharpened rows, 8 wide (mirror-extended columns, :165)      -> Y'' = 5x5 blur (:202)
// and (U, V) of the last rows in a wave-private LDS ring until their luma arrives.  Neighbour columns come from
// the neighbouring lanes (DPP wave shifts); the strip edges go through LDS between the wavefronts of the
// workgroup, which cover one image row side by side: per row and wavefront 6 floats and ONE barrier (the Y' edge
// columns travel one row late, together with the next row's Y edges).  Raw row q+1 gives Y(q), Y'(q-1) and the
// finished output row q-4.  Row borders: mirrored raw rows are fetched as such (mirror padding keeps the Bayer
// parity), luma rows outside the image are zero, and the blur of the first / last two image rows uses weight
// sets with the mirror padding folded in (R2LFolded::blur_edge), so the window never needs a row it does not have.
// Compared with the tile kernel: no halo recompute in x, 7 rows of halo per band in y, no phases, no plane
// round trips through LDS.
#pragma once
#include "r2l_param_kernels.h"

#ifndef R2L_SERIAL  // (needs lane-to-lane moves: the device, or the lock-step emulation)

#ifndef R2L_FS_PF
#define R2L_FS_PF 2
#endif
#ifndef R2L_FS_BF_FETCH
#define R2L_FS_BF_FETCH 1
#endif
#ifndef R2L_FS_FULL_GROUPS
#define R2L_FS_FULL_GROUPS 1
#endif

struct R2LFwdStreamArgs {
  R2LRaw raw;
  const R2LFolded* F;
  const float* bn;      // mean[3], istd[3] or null
  float* out;           // (B,3,H,W) or null (statistics only)
  float* yp_out;        // (B,H,W) or null: the sharpened luma Y', kept for the backward and for the apply pass
  const float* yp_in;   // r2l_fwd_apply_block: the plane an earlier pass kept
  float* stat_partial;  // [12][nblk] or null: (high, low) float32 halves of the workgroups' float64 totals
  int B, H, W;
  int nband, band_h, nitems;  // work item = (image, band); workgroup bid takes items bid, bid + nblk, ...
  R2LTree tree;
  double* stats_out;
  R2LBnFinalizeArgs fin;
  R2LEpi ep;  // EPI instantiations: where the output goes (R2LEpi)
  int xcdm;   // neighbouring workgroups per XCD (r2l_xcd_window; 0 = off): the work items follow the mapped workgroup id
#ifdef R2L_EXP_STAMPS
  unsigned long long* tl;  // diagnostic builds: (start, end) s_memrealtime of every workgroup's first wavefront
#endif
};
#if defined(R2L_EXP_STAMPS) && !defined(R2L_EMUL)
#define R2L_TL_BEGIN(a, bid) const unsigned long long tl0_ = __builtin_amdgcn_s_memrealtime();
#define R2L_TL_END(a, bid)                                                       \
  if ((a).tl && (bid) < 4096 && threadIdx.x == 0) {                              \
    (a).tl[2 * (bid)] = tl0_;                                                    \
    (a).tl[2 * (bid) + 1] = __builtin_amdgcn_s_memrealtime();                    \
  }                                                                              \
  if ((a).tl && (a).stat_partial && (bid) < 1024 && (threadIdx.x & 63) == 0) { /* where every wavefront of the statistics kernel ran: HW_ID (4), XCC_ID (20) */ \
    (a).tl[16384 + (bid) * 4 + (threadIdx.x >> 6)] = /* (tl = area + 8192 in this kernel) */                             \
        ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) << 32) | \
        (unsigned)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));         \
  }
#else
#define R2L_TL_BEGIN(a, bid)
#define R2L_TL_END(a, bid)
#endif

// the lane's index in its wavefront from the hardware (v_mbcnt), opaque to the optimiser: every call re-derives it (two
// instructions) instead of keeping one value alive from the kernel's first instruction to its last
R2L_HD int r2l_lane_id() {
#ifdef R2L_EMUL
  return (int)(threadIdx.x & 63u);
#else
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
#endif
}
R2L_HD float r2l_wshr(float x, float edge) {  // previous lane's x; lane 0 of the wavefront gets `edge`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x),
                                                               0x138, 0xf, 0xf, false));
}
R2L_HD float r2l_wshl(float x, float edge) {  // next lane's x; lane 63 gets `edge`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x),
                                                               0x130, 0xf, 0xf, false));
}

#define R2L_FS_EX 8                          // floats per wavefront and buffer in the exchange area
#define R2L_FS_FIFO_ROWS 6                   // (chroma waits 4 rows for its luma; slot = row mod 6 = unroll position)
#define R2L_FS_FIFO_F4 (R2L_FS_FIFO_ROWS * 2 * 64)  // (U[4], V[4]) of 6 rows x 64 lanes = 12 KB per wavefront
// reduction scratch ((6 NT + 96) doubles, >= the tree's scratch): it reuses the chroma rings, which are idle by then
#define R2L_FS_RED_FLOATS(NW) 3072  // (<= one wavefront's ring: 12 slots x 128 groups of the tree in one pass)
#define R2L_FS_RING_FLOATS(NW) ((NW) * R2L_FS_FIFO_F4 * 4 > R2L_FS_RED_FLOATS(NW) ? (NW) * R2L_FS_FIFO_F4 * 4 : R2L_FS_RED_FLOATS(NW))
#define R2L_FS_LDS_FLOATS(NW) (2 * (NW) * R2L_FS_EX + 16 + R2L_FS_RING_FLOATS(NW) + 12 * (NW))  // + 6 doubles per wave

// one raw row in flight: the lane's 4 values (undecoded bits for 16-bit containers) + the strip-edge neighbour
struct R2LFsStage {
  r2l_f4 c;
  float e;
  int ym;  // mirrored source row
};
template <bool U16>
R2L_HD void r2l_fs_fetch(const R2LFwdStreamArgs& a, size_t img0, int ym, int x0, bool le, bool re, int lane,
                         R2LFsStage& s) {
  const size_t e = img0 + (size_t)ym * a.W + x0;
  s.ym = ym;
  s.e = 0.f;
  if (U16) {
    const unsigned short* r = a.raw.u16 + e;
    const r2l_f2 b = *(const r2l_f2*)r;
    s.c.x = b.x;
    s.c.y = b.y;
    if ((lane == 0 && !le) || (lane == 63 && !re)) s.e = r2l_u2f((unsigned)(lane == 0 ? r[-1] : r[4]));
  } else {
    const float* r = a.raw.f32 + e;
    s.c = r2l_stream_load_f4(r);
    if ((lane == 0 && !le) || (lane == 63 && !re)) s.e = (lane == 0) ? r[-1] : r[4];
  }
}
// ... branch-free: every lane loads an edge value from an in-row address, only the strip's first / last lane use theirs
// (a load under a lane-dependent or uniform condition in the row loop makes hipcc wait with vmcnt(0) in every step)
template <bool U16>
R2L_HD void r2l_fs_fetch_bf(const R2LFwdStreamArgs& a, size_t img0, int ym, int x0, bool le, bool re, int lane,
                            R2LFsStage& s) {
  const size_t e = img0 + (size_t)ym * a.W + x0;
  const int eo = (lane < 32) ? (le ? 0 : -1) : (re ? 3 : 4);
  s.ym = ym;
  if (U16) {
    const unsigned short* r = a.raw.u16 + e;
    const r2l_f2 b = *(const r2l_f2*)r;
    s.c.x = b.x;
    s.c.y = b.y;
    s.e = r2l_u2f((unsigned)r[eo]);
  } else {
    const float* r = a.raw.f32 + e;
    s.c = r2l_stream_load_f4(r);
    s.e = r[eo];
  }
}
// staged row -> 6 black-level-corrected values, columns x0-1 .. x0+4 (mirror padding at the image edges: column
// -1 is column 1, column W is column W-2; the black level follows the source site)
template <bool U16>
R2L_HD void r2l_fs_convert(const R2LFwdStreamArgs& a, R2LFoldedRef F, const R2LFsStage& s, bool le, bool re,
                           float v[6]) {
  float c0, c1, c2, c3, ee = s.e;
  if (U16) {
    const unsigned lo = r2l_f2u(s.c.x), hi = r2l_f2u(s.c.y);
    c0 = r2l_raw_decode(lo & 0xffffu, a.raw);
    c1 = r2l_raw_decode(lo >> 16, a.raw);
    c2 = r2l_raw_decode(hi & 0xffffu, a.raw);
    c3 = r2l_raw_decode(hi >> 16, a.raw);
    ee = r2l_raw_decode(r2l_f2u(s.e) & 0xffffu, a.raw);
  } else {
    c0 = s.c.x;
    c1 = s.c.y;
    c2 = s.c.z;
    c3 = s.c.w;
  }
  const float be = (s.ym & 1) ? F.bl[2] : F.bl[0], bo = (s.ym & 1) ? F.bl[3] : F.bl[1];
  c0 -= be;
  c1 -= bo;
  c2 -= be;
  c3 -= bo;
  const float l = r2l_wshr(c3, ee - bo), r = r2l_wshl(c0, ee - be);  // column x0-1 is odd, x0+4 even
  v[0] = le ? c1 : l;
  v[1] = c0;
  v[2] = c1;
  v[3] = c2;
  v[4] = c3;
  v[5] = re ? c2 : r;
}

struct R2LFsState {
  float v[3][6];   // V rows (slot = row mod 3)
  float y[3][6];   // Y rows with their left / right neighbours (slot = row mod 3)
  float yp[6][8];  // Y' rows with two neighbours each side (slot = row mod 6)
  // statistics: sum (x - p), sum (x - p)^2 per channel, per pair half, about a PIVOT p of the lane's own: the value of its
  // first pixel of the band.  Images are locally smooth, so the float32 sums stay small and exact-ish; with a fixed pivot
  // (0.5, rounds 1-2) a channel near saturation (mean 0.99, variance 1.4e-3) lost 170x of the sums' precision to the
  // cancellation in E[d^2] - E[d]^2: 7e-6 relative in the variance, 2e-5 in the normalised output -- five times the
  // distance of the reference's own float32 run from its float64 run (tests: "out vs reference float64").  The lane
  // re-bases its sums to the common pivot 0.5 in float64 when the band is done.
  r2l_p2 acc[6];
  float piv[3];
};

// 3x3 stencil with per-column-parity weights on a 6-wide 3-row window -> 4 outputs as 2 pairs.  A window row is three
// aligned pairs P0 = (w0,w1), P1 = (w2,w3), P2 = (w4,w5); the middle tap's operands straddle them: ONE v_pk_mov_b32 each
// (r2l_straddle) -- written as r2l_mk2(row[1], row[2]) hipcc copies both halves with a v_mov_b32 each, at every use.
#ifndef R2L_STENCIL_STRADDLE
#define R2L_STENCIL_STRADDLE 1
#endif
// the pairs of a 6-wide window row at column offsets 0, 1, 2: x[j][p] = (row[2p + j], row[2p + j + 1])
R2L_HD void r2l_row_pairs(const float* row, r2l_p2 x[3][2]) {
#if R2L_STENCIL_STRADDLE
  const r2l_p2 P0 = r2l_mk2(row[0], row[1]), P1 = r2l_mk2(row[2], row[3]), P2 = r2l_mk2(row[4], row[5]);
  x[0][0] = P0;
  x[0][1] = P1;
  x[1][0] = r2l_straddle(P0, P1);
  x[1][1] = r2l_straddle(P1, P2);
  x[2][0] = P1;
  x[2][1] = P2;
#else
  R2L_PRAGMA_UNROLL
  for (int j = 0; j < 3; ++j)
    R2L_PRAGMA_UNROLL
  for (int p = 0; p < 2; ++p) x[j][p] = r2l_mk2(row[2 * p + j], row[2 * p + j + 1]);
#endif
}
template <class WT>
R2L_HD void r2l_fs_stencil_parity(const float* r0, const float* r1, const float* r2, WT w /* [9][2] */, r2l_p2 o[2]) {
  o[0] = o[1] = r2l_splat2(0.f);
  const float* rows[3] = {r0, r1, r2};
#if R2L_STENCIL_STRADDLE
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i) {
    const r2l_p2 P0 = r2l_mk2(rows[i][0], rows[i][1]), P1 = r2l_mk2(rows[i][2], rows[i][3]), P2 = r2l_mk2(rows[i][4], rows[i][5]);
    const r2l_p2 O0 = r2l_straddle(P0, P1), O1 = r2l_straddle(P1, P2);
    const r2l_p2 x[3][2] = {{P0, P1}, {O0, O1}, {P1, P2}};
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      const r2l_p2 wy = r2l_mk2(w[i * 3 + j][0], w[i * 3 + j][1]);
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) o[p] = r2l_pfma(wy, x[j][p], o[p]);
    }
  }
#else
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 3; ++j) {
    const r2l_p2 wy = r2l_mk2(w[i * 3 + j][0], w[i * 3 + j][1]);
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) o[p] = r2l_pfma(wy, r2l_mk2(rows[i][2 * p + j], rows[i][2 * p + j + 1]), o[p]);
  }
#endif
}
// the same with ONE weight per tap for both halves (the sharpen): 3x3 cross-correlation on a 6-wide 3-row window
template <class WT>
R2L_HD void r2l_fs_stencil_plain(const float* r0, const float* r1, const float* r2, WT w /* [9] */, r2l_p2 o[2]) {
  o[0] = o[1] = r2l_splat2(0.f);
  const float* rows[3] = {r0, r1, r2};
#if R2L_STENCIL_STRADDLE
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i) {
    const r2l_p2 P0 = r2l_mk2(rows[i][0], rows[i][1]), P1 = r2l_mk2(rows[i][2], rows[i][3]), P2 = r2l_mk2(rows[i][4], rows[i][5]);
    const r2l_p2 O0 = r2l_straddle(P0, P1), O1 = r2l_straddle(P1, P2);
    const r2l_p2 x[3][2] = {{P0, P1}, {O0, O1}, {P1, P2}};
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      const r2l_p2 ws = r2l_splat2(w[i * 3 + j]);
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) o[p] = r2l_pfma(ws, x[j][p], o[p]);
    }
  }
#else
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 3; ++j) {
    const r2l_p2 ws = r2l_splat2(w[i * 3 + j]);
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) o[p] = r2l_pfma(ws, r2l_mk2(rows[i][2 * p + j], rows[i][2 * p + j + 1]), o[p]);
  }
#endif
}

// the colour code of one output row (:203-217): Y'', U, V of the lane's 4 pixels -> RGB, clip, gamma, [statistics about the
// lane's pivot], [BatchNorm], store (EPI: at the augmented position, R2LEpi)
// STATS: 0 none; 1 the streaming kernel's form (under `a.stat_partial && store_ok`); 2 branch-free, weighted with smask
// (1 for the pixels that count, 0 for the others), the pivot taken in the band's first row (`first`); 3 the sums of BatchNorm's
// backward over grad_out (gk: this row's grad_out as pairs)
#ifndef R2L_OUT_NT
#define R2L_OUT_NT 1
#endif
template <bool EPI, int STATS>
R2L_HD void r2l_fs_colour(const R2LFwdStreamArgs& a, R2LFoldedRef F, r2l_p2* acc, float* piv, const r2l_p2 ypp[2],
                          const r2l_p2 u[2], const r2l_p2 v[2], int y, int y0, int x0, float* ob, unsigned plane,
                          bool store_ok, const float mean[3], const float istd[3], float smask = 0.f,
                          bool first = false, const r2l_p2 (*gk)[2] = nullptr) {
  const unsigned off0 = (unsigned)y * (unsigned)a.W + (unsigned)x0;
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    r2l_p2 x[2];
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) {
      r2l_p2 rgb = r2l_pmul(r2l_splat2(F.M2[k * 3]), ypp[p]);
      rgb = r2l_pfma(r2l_splat2(F.M2[k * 3 + 1]), u[p], rgb);
      rgb = r2l_pfma(r2l_splat2(F.M2[k * 3 + 2]), v[p], rgb);
      const r2l_p2 lg = r2l_mk2(r2l_log2(fminf(fmaxf(rgb[0], 1e-5f), 1.0f)),     // :206
                                r2l_log2(fminf(fmaxf(rgb[1], 1e-5f), 1.0f)));
      const r2l_p2 e = r2l_pmul(lg, r2l_splat2(F.inv_gamma));                    // :209
      x[p] = r2l_mk2(r2l_exp2(e[0]), r2l_exp2(e[1]));
      if (STATS == 1 && a.stat_partial && store_ok) {
        if (p == 0) piv[k] = (y == y0) ? x[0][0] : piv[k];
        const r2l_p2 d = r2l_padd(x[p], r2l_splat2(-piv[k]));
        acc[k] = r2l_padd(acc[k], d);
        acc[3 + k] = r2l_pfma(d, d, acc[3 + k]);
      }
      if (STATS == 3) {
        // BatchNorm's backward sums (:217 backward): sum g and sum g * xhat, xhat = the value the apply pass stored -- the same
        // expression on the same numbers -- RECOMPUTED from the raw frame and Y' instead of read back (r2l_bnr_planes_block);
        // smask: the wave-uniform 1 / 0 of the row (lanes beyond the frame are taken out once, after the band)
        const r2l_p2 xh = r2l_pmul(r2l_padd(x[p], r2l_splat2(-mean[k])), r2l_splat2(istd[k]));
        const r2l_p2 t = r2l_pmul(gk[k][p], r2l_splat2(smask));
        acc[k] = r2l_padd(acc[k], t);
        acc[3 + k] = r2l_pfma(t, xh, acc[3 + k]);
      }
      if (STATS == 2) {
        if (p == 0) piv[k] = first ? x[0][0] : piv[k];
        const r2l_p2 d = r2l_padd(x[p], r2l_splat2(-piv[k]));
        const r2l_p2 t = r2l_pmul(d, r2l_splat2(smask));
        acc[k] = r2l_padd(acc[k], t);
        acc[3 + k] = r2l_pfma(t, d, acc[3 + k]);
      }
    }
    if (ob && store_ok) {
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p)
        x[p] = r2l_pmul(r2l_padd(x[p], r2l_splat2(-mean[k])), r2l_splat2(istd[k]));  // :217
      r2l_f4 s4;
      s4.x = x[0][0];
      s4.y = x[0][1];
      s4.z = x[1][0];
      s4.w = x[1][1];
      if (!EPI) {
        // the output goes AROUND the caches (nontemporal): 12 B/px that this pass never reads again would otherwise push the raw
        // frames and Y' -- which the neighbouring bands and the backward re-read -- out of the memory-side cache: apply pass
        // 72.8 -> 64.9 us at 64x512x512, bn_reduce (which reads the output later) +0.5 (profiles/r05_nt_stores.txt;
        // -DR2L_OUT_NT=0: the plain store).  The kept planes Y', dL/dY'', HP stay cached: their readers follow at once.
#if R2L_OUT_NT
        r2l_store_f4_nt(ob + (unsigned)k * plane + off0, s4);
#else
        *(r2l_f4*)(ob + (unsigned)k * plane + off0) = s4;
#endif
      } else {  // the augmented position of this lane's 4 pixels (R2LEpi)
        float* o = ob + (unsigned)k * plane + (a.ep.s0 + a.ep.sr * y + a.ep.sc * x0);
        if (a.ep.sc == 1) {
#if R2L_OUT_NT
          r2l_store_f4_nt(o, s4);
#else
          *(r2l_f4*)o = s4;
#endif
        } else if (a.ep.sc == -1) {
          r2l_f4 r4;
          r4.x = s4.w;
          r4.y = s4.z;
          r4.z = s4.y;
          r4.w = s4.x;
#if R2L_OUT_NT
          r2l_store_f4_nt(o - 3, r4);
#else
          *(r2l_f4*)(o - 3) = r4;
#endif
        } else {
          o[0] = s4.x;
          o[a.ep.sc] = s4.y;
          o[2 * a.ep.sc] = s4.z;
          o[3 * a.ep.sc] = s4.w;
        }
      }
    }
  }
}

// K: ring position of the step (window / chroma-ring slots); PYQ: parity of row q; FULL: the driver runs every group of 6
// steps in full (r2l_fwd_stream_block) -- rows past the band's end are computed and neither stored nor counted
template <int NW, bool U16, int K, bool EPI, int PYQ = (K & 1), bool FULL = false>
R2L_HD void r2l_fs_step(const R2LFwdStreamArgs& a, R2LFsState& st, int q, int y0, int y1, bool le, bool re,
                        int wave, int lane, float* ex, r2l_f4* fifo, float* ob, float* ypb, unsigned plane, int x0,
                        bool store_ok, const float mean[3], const float istd[3], float smask) {
  R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque(a.F));
  R2LFoldedRef Fh = R2L_FOLDED_REF(a.F);  // not laundered: the 25 blur weights stay in scalar registers (-1 %)
  constexpr int PY = PYQ;
  const int H = a.H;
  const float* vu = st.v[(K + 2) % 3];  // V(q-1)
  const float* vm = st.v[K % 3];        // V(q)
  const float* vl = st.v[(K + 1) % 3];  // V(q+1)
  float* yq = st.y[K % 3];
  const bool qin = (unsigned)q < (unsigned)H;
  // ---- Y(q) into the window, (U, V)(q) into the ring ----------------------------------------------------------
  {
    r2l_p2 o[2];
    r2l_fs_stencil_parity(vu, vm, vl, F.AY2[PY], o);
    yq[1] = o[0][0];
    yq[2] = o[0][1];
    yq[3] = o[1][0];
    yq[4] = o[1][1];
    if (!qin) yq[1] = yq[2] = yq[3] = yq[4] = 0.f;  // zero padding of the sharpen conv (:162 padding=1)
  }
  // ---- strip edges: Y(q) (1 column each side) and Y'(q-2) (2 columns each side) ------------------------------
  float* ypq2 = st.yp[(K + 4) % 6];  // Y'(q-2): own columns in [2..5], neighbours still missing
  float rl_y = 0.f, rl_p2 = 0.f, rl_p3 = 0.f, rr_y = 0.f, rr_p0 = 0.f, rr_p1 = 0.f;
#ifdef R2L_EXP_NO_FS_EXCH  // (timing-only ablation: no strip-edge exchange at all)
  constexpr bool EXCH = false;
#else
  constexpr bool EXCH = NW > 1;
#endif
  if (EXCH) {
    float* mine = ex + ((q & 1) * NW + wave) * R2L_FS_EX;
    if (lane == 0) {
      mine[0] = yq[1];
      mine[1] = ypq2[2];
      mine[2] = ypq2[3];
    }
    if (lane == 63) {
      mine[3] = yq[4];
      mine[4] = ypq2[4];
      mine[5] = ypq2[5];
    }
  }
  {
    // (U, V)(q) into the ring while the edge values are on their way (the halo rows of a band only need luma)
    if (q >= y0 && q < y1) {
      r2l_p2 u[2], v[2];
      r2l_fs_stencil_parity(vu, vm, vl, F.AU2[PY], u);
      r2l_fs_stencil_parity(vu, vm, vl, F.AV2[PY], v);
      r2l_f4* f = fifo + (K % R2L_FS_FIFO_ROWS) * 2 * 64 + lane;
      r2l_f4 fu, fv;
      fu.x = u[0][0];
      fu.y = u[0][1];
      fu.z = u[1][0];
      fu.w = u[1][1];
      fv.x = v[0][0];
      fv.y = v[0][1];
      fv.z = v[1][0];
      fv.w = v[1][1];
      f[0] = fu;
      f[64] = fv;
    }
  }
  if (EXCH) {
    float* mine = ex + ((q & 1) * NW + wave) * R2L_FS_EX;
#ifndef R2L_EXP_NO_FS_BARRIER  // (timing-only ablation: the strip edges read whatever the neighbour wrote last)
    R2L_LDS_BARRIER();
#endif
    if (lane == 0 && wave > 0) {
      const float* o = mine - R2L_FS_EX;
      rl_y = o[3];
      rl_p2 = o[4];
      rl_p3 = o[5];
    }
    if (lane == 63 && wave < NW - 1) {
      const float* o = mine + R2L_FS_EX;
      rr_y = o[0];
      rr_p0 = o[1];
      rr_p1 = o[2];
    }
  }
  {
    const float l = r2l_wshr(yq[4], rl_y), r = r2l_wshl(yq[1], rr_y);
    yq[0] = le ? 0.f : l;  // zero padding
    yq[5] = re ? 0.f : r;
    const float l2 = r2l_wshr(ypq2[4], rl_p2), l1 = r2l_wshr(ypq2[5], rl_p3);
    const float r1 = r2l_wshl(ypq2[2], rr_p0), r2 = r2l_wshl(ypq2[3], rr_p1);
    // mirror padding of the blur (:165 reflect): column -1 = column 1, -2 = 2; W = W-2, W+1 = W-3
    const float m1 = ypq2[3], m2 = ypq2[4];
    ypq2[0] = le ? m2 : l2;
    ypq2[1] = le ? m1 : l1;
    ypq2[6] = re ? m2 : r1;
    ypq2[7] = re ? m1 : r2;
  }
  // ---- Y'(q-1) = sharpen(Y): 3x3 cross-correlation over rows q-2, q-1, q ---------------------------------------
  {
    float* ypn = st.yp[(K + 5) % 6];
    const float* yu = st.y[(K + 1) % 3];  // Y(q-2)
    const float* ym = st.y[(K + 2) % 3];  // Y(q-1)
    r2l_p2 o[2];
    o[0] = o[1] = r2l_splat2(0.f);
    const float* rows[3] = {yu, ym, yq};
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      const r2l_p2 ws = r2l_splat2(F.sharp[i * 3 + j]);
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) o[p] = r2l_pfma(ws, r2l_mk2(rows[i][2 * p + j], rows[i][2 * p + j + 1]), o[p]);
    }
    ypn[2] = o[0][0];
    ypn[3] = o[0][1];
    ypn[4] = o[1][0];
    ypn[5] = o[1][1];
  }
  // ---- output row y = q-4 ------------------------------------------------------------------------------------
  const int y = q - 4;
  const bool row_ok = y < y1;  // FULL: rows past the band's end (the padding steps of its last group)
  if (FULL) {
    store_ok = store_ok && row_ok;
    smask = row_ok ? smask : 0.f;
  }
  if (y >= y0 && (FULL || row_ok)) {
    r2l_p2 ypp[2];
    {
      // window rows y-2 .. y+2 = q-6 .. q-2 sit in ring slots K .. K+4
      float yw[5][8];
      R2L_PRAGMA_UNROLL
      for (int i = 0; i < 5; ++i)
        R2L_PRAGMA_UNROLL
      for (int j = 0; j < 8; ++j) yw[i][j] = st.yp[(K + i) % 6][j];
      // the first / last two image rows take the weight sets with the mirror padding folded in (a padding row past the
      // image's last row: any finite weights)
      const int set = (y < 2) ? y : (y - (H - 2)) + 2;
#ifdef R2L_EXP_CONST_WEIGHTS
    const float* w25 = &Fh.blur[0];  // (timing only: no border sets)
    (void)set;
#else
      const R2L_CONSTAS float* w25 =
          ((y >= 2 && y < H - 2) || y >= H) ? &Fh.blur[0] : &F.blur_edge[0][0] + 25 * set;
#endif
      r2l_blur_row2w(yw, w25, ypp);
    }
    const r2l_f4* f = fifo + ((K + 2) % R2L_FS_FIFO_ROWS) * 2 * 64 + lane;  // row q-4
    const r2l_f4 fu = f[0], fv = f[64];
    const r2l_p2 u[2] = {r2l_mk2(fu.x, fu.y), r2l_mk2(fu.z, fu.w)}, v[2] = {r2l_mk2(fv.x, fv.y), r2l_mk2(fv.z, fv.w)};
    // (statistics: the branch-free form, weighted with the lane's 0 / 1 mask -- 6 v_pk_mul_f32 per row step where the
    // conditional form `if (a.stat_partial && store_ok)` is if-converted into 24 v_cndmask_b32)
    r2l_fs_colour<EPI, 2>(a, F, st.acc, st.piv, ypp, u, v, y, y0, x0, ob, plane, store_ok, mean, istd, smask, y == y0);
    const unsigned off0 = (unsigned)y * (unsigned)a.W + (unsigned)x0;
    // Y'(y), kept for kernel B1 of the backward: the middle row of the blur's window, stored last (the step's
    // registers are free here; next to the sharpen, or in front of the colour code, the kernel spills)
    if (ypb && store_ok) {
#ifndef R2L_EMUL
      asm volatile("" ::: "memory");
#endif
      const float* ypy = st.yp[(K + 2) % 6];
      r2l_f4 s4;
      s4.x = ypy[2];
      s4.y = ypy[3];
      s4.z = ypy[4];
      s4.w = ypy[5];
#ifdef R2L_EXP_YP_NT
      r2l_store_f4_nt(ypb + off0, s4);
#else
      *(r2l_f4*)(ypb + off0) = s4;
#endif
    }
  }
  // rows of Y' that lie outside the image only ever meet zero weights, but must stay finite
  if ((unsigned)(q - 1) >= (unsigned)H) {
    float* ypn = st.yp[(K + 5) % 6];
    ypn[2] = ypn[3] = ypn[4] = ypn[5] = 0.f;
  }
}

// one work item's lane sums (float32 pairs about the lane's pivot) -> float64 about the common pivot 0.5, added over the
// wavefront in a fixed butterfly order into its float64 totals in LDS; npx = pixels behind this lane's sums
R2L_HD void r2l_fs_lane_sums(const r2l_p2* acc, const float* piv, double npx, bool ok, int lane, double* tots) {
  double part[6];
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    // from the lane's pivot p to the common pivot 0.5:  x - .5 = (x - p) + dp
    const double s1 = (double)acc[k][0] + (double)acc[k][1];
    const double s2 = (double)acc[3 + k][0] + (double)acc[3 + k][1];
    const double dp = ok ? (double)piv[k] - 0.5 : 0.0;
    double v1 = fma(npx, dp, s1), v2 = fma(dp, fma(npx, dp, 2.0 * s1), s2);
    R2L_PRAGMA_UNROLL
    for (int m = 32; m >= 1; m >>= 1) {
      v1 += __shfl_xor(v1, m, 64);
      v2 += __shfl_xor(v2, m, 64);
    }
    part[k] = v1;
    part[3 + k] = v2;
  }
  if (lane == 0) {
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 6; ++i) tots[i] += part[i];
  }
}

// ---- statistics: the wavefronts' float64 totals -> one partial per slot and workgroup (fixed order), then the shared
// tree; the last workgroup of the launch writes the totals and, one rank, does the BatchNorm bookkeeping
template <int NW, int NT>
R2L_BLOCKFN void r2l_fs_stats_finish(const R2LFwdStreamArgs& a_, int bid, int nblk, int tid, int wave, double* tots,
                                     float* red) {
  (void)a_;
  // the arguments of this part are read here, not carried through the main loop in scalar registers (r2l_kernargs)
  const R2L_CONSTAS R2LFwdStreamArgs* ka = r2l_kernargs<R2LFwdStreamArgs>();
  struct {
    float* stat_partial;
    R2LTree tree;
    double* stats_out;
    R2LBnFinalizeArgs fin;
    int B, H, W;
  } a;
  a.stat_partial = ka->stat_partial;
  a.tree.partial = ka->tree.partial;
  a.tree.partial2 = ka->tree.partial2;
  a.tree.gpartial = ka->tree.gpartial;
  a.tree.counters = ka->tree.counters;
  a.tree.split = ka->tree.split;
  a.tree.nblk1 = ka->tree.nblk1;
  a.stats_out = ka->stats_out;
  a.fin.tot = ka->fin.tot;
  a.fin.nranks = ka->fin.nranks;
  a.fin.bn = ka->fin.bn;
  a.fin.moments = ka->fin.moments;
  a.fin.running_mean = ka->fin.running_mean;
  a.fin.running_var = ka->fin.running_var;
  a.fin.eps = ka->fin.eps;
  a.fin.momentum = ka->fin.momentum;
  a.fin.num_batches_tracked = ka->fin.num_batches_tracked;
  a.B = ka->B;
  a.H = ka->H;
  a.W = ka->W;
  // ---- statistics: lanes -> one partial per slot and workgroup (fixed order), then the shared tree ------------
  {
    R2L_TAILST(0);
    // (the bookkeeping's reads of global memory, asked for now: they arrive behind the partials' store wait)
    R2LBnPre pre;
    pre.nbt = 0;
    pre.rm = pre.rv = 0.f;
    if (a.tree.counters && a.fin.bn) pre = r2l_bn_finalize_fetch(a.fin, tid);
    R2L_LDS_BARRIER();  // every wavefront is done with its chroma ring
    if (tid < 6) {  // the wavefronts' totals in wavefront order; (high, low) float32 halves in slots tid and 6 + tid
      double acc = 0.0;
      for (int w = 0; w < NW; ++w) acc += (tots - wave * 6)[w * 6 + tid];
      const float hi = (float)acc;
      r2l_store_coherent(&a.stat_partial[(size_t)tid * nblk + bid], hi);
      r2l_store_coherent(&a.stat_partial[(size_t)(6 + tid) * nblk + bid], (float)(acc - (double)hi));
    }
    R2L_STORES_DONE();
    R2L_LDS_BARRIER();
    R2L_TAILST(1);
    double* sl = (double*)(red + 4);  // totals in LDS: the bookkeeping below reads them back
    if (a.tree.counters &&
        r2l_tree_finish<12, NT>(a.tree, bid, nblk, red, sl, (double*)(red + 512), (R2L_FS_RED_FLOATS(NW) - 512) / 2)) {
      if (tid < 6) sl[tid] += sl[6 + tid];
      R2L_LDS_BARRIER();
      if (tid == 0) sl[6] = (double)a.B * (double)a.H * (double)a.W;
      R2L_LDS_BARRIER();
      if (tid < 7) a.stats_out[tid] = sl[tid];
      R2L_LDS_BARRIER();
      if (a.fin.bn) {
        R2LBnFinalizeArgs f = a.fin;
        f.tot = sl;
        f.nranks = 1;
        r2l_bn_finalize_lane(f, tid, pre);
        if (tid == 0 && f.num_batches_tracked) *f.num_batches_tracked = pre.nbt + 1;  // (lanes 0-2 read it long ago)
      }
      R2L_TAILST(7);
    }
  }
}

// SONLY: the statistics pass of train-mode BatchNorm as its own instantiation -- no output, no BatchNorm constants, no
// epilogue: fewer live scalars (the general kernel parks ~25 of them per row step in vector lanes, v_readlane_b32)
template <int NW, bool U16, bool EPI = false, bool SONLY = false>
R2L_BLOCKFN void r2l_fwd_stream_block(const R2LFwdStreamArgs& a, int bid, int nblk, float* lds) {
  constexpr int NT = NW * 64;
#if defined(R2L_EXP_STAMPS) && !defined(R2L_EMUL)
  const unsigned long long tls0_ = __builtin_amdgcn_s_memrealtime();
#endif
  // (wave: a scalar; lane: the hardware's lane id, re-derived wherever it is needed -- the thread id the kernel was handed
  // in v0 would otherwise stay live across the row loop, which has no register to spare: 28 B of scratch in round 4)
  // (the statistics instantiation only: in the others the same change ADDS 20 B of scratch)
  const int wave = SONLY ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : (int)threadIdx.x >> 6;
  const int lane = SONLY ? r2l_lane_id() : (int)threadIdx.x & 63;
  float* ex = lds;
  r2l_f4* fifo = (r2l_f4*)(lds + 2 * NW * R2L_FS_EX + 16) + (size_t)wave * R2L_FS_FIFO_F4;
  float* red = lds + 2 * NW * R2L_FS_EX + 16;  // reduction scratch: over the chroma rings, after the last item
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  const int xs = wave * 256 + 4 * lane;
  const bool store_ok = xs < a.W;
  const int x0 = store_ok ? xs : a.W - 4;
  const bool le = x0 == 0, re = x0 + 4 >= a.W;
  const unsigned plane = (unsigned)a.H * (unsigned)a.W;
  float mean[3] = {0.f, 0.f, 0.f}, istd[3] = {1.f, 1.f, 1.f};
  if (!SONLY && a.bn) {
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      mean[k] = a.bn[k];
      istd[k] = a.bn[3 + k];
    }
  }
  const float smask = ((SONLY || a.stat_partial) && store_ok) ? 1.f : 0.f;  // does this lane's pixel count in the statistics?
  R2LFsState st;
  // statistics: float32 pair accumulators per work item (<= band_h x 4 pixels per lane); after every item the
  // wavefront adds its 64 lane sums (float64, fixed butterfly order) into its float64 totals in LDS -- the rounding
  // of a sum then does not grow with the items a workgroup walks, i.e. does not depend on the grid, and the totals
  // cost no registers
  double* tots = (double*)(lds + 2 * NW * R2L_FS_EX + 16 + R2L_FS_RING_FLOATS(NW)) + wave * 6;
  if (lane < 6) tots[lane] = 0.0;
  // the chroma ring starts out as zeros: the padding steps of a band's last group (R2L_FS_FULL_GROUPS) read ring rows this item
  // never wrote -- harmless garbage only as long as fmin(fmax(NaN, 1e-5), 1) clamps a NaN before the 0 / 1 mask multiplies it;
  // with defined contents nothing depends on that (12 LDS stores per lane, once per kernel; later items see earlier items' rows)
  {
    r2l_f4 z;
    z.x = z.y = z.z = z.w = 0.f;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < R2L_FS_FIFO_ROWS * 2; ++i) fifo[i * 64 + lane] = z;
  }
  constexpr int PF = R2L_FS_PF;
  static_assert(6 % PF == 0, "the prefetch ring is indexed by the unroll position");
#if defined(R2L_EXP_STAMPS) && !defined(R2L_EMUL)
  // diagnostic builds (tests/timeline_fwd.py, R2L_TL_STREAM=1): s_memrealtime at the workgroup's entry, in front of its item loop,
  // behind it, and behind the statistics' tail -- 4 records per workgroup at tl[4 * bid]
  const unsigned long long tls1_ = __builtin_amdgcn_s_memrealtime();
#endif
  for (int item = r2l_xcd_window(bid, nblk, a.xcdm); item < a.nitems; item += nblk) {
    const int band = item % a.nband, b = item / a.nband;
    const int y0 = band * a.band_h;
    const int y1 = (y0 + a.band_h < a.H) ? y0 + a.band_h : a.H;
    const size_t img = (size_t)b * plane;
    float* ob = (!SONLY && a.out) ? a.out + (size_t)b * 3 * plane : nullptr;
    float* ypb = a.yp_out ? a.yp_out + (size_t)b * plane : nullptr;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 6; ++i) st.acc[i] = r2l_splat2(0.f);
    st.piv[0] = st.piv[1] = st.piv[2] = 0.5f;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 6; ++j) st.y[i][j] = 0.f;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 6; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 8; ++j) st.yp[i][j] = 0.f;
#if R2L_FS_FULL_GROUPS
    // first luma row computed: qf = y0 - 3 (Y'(y0-2) needs Y(y0-3)); steps s = 0, 1, ... with q = qf + s, unrolled 6-fold:
    // the ring position of a step is s mod 6 (every window slot a compile-time index), the parity of its row (s + 1) mod 2
    // (bands start on even rows).  EVERY group of 6 runs in full and every fetch is unconditional (r2l_fs_fetch_bf, rows
    // past the band's last raw row re-fetch that row): with a conditionally executed step or fetch in the loop hipcc
    // waited with vmcnt(0) right behind each fetch -- for the row just requested and the previous rows' stores.  The
    // padding steps of the last group (band_h + 7 steps: 22-row bands pad 1) finish rows >= y1, which are neither stored
    // nor counted.
    {
      const int qf = y0 - 3, q1 = y1 + 4;  // q1: exclusive, = the last raw row the band consumes
      {
        R2LFsStage s0, s1;
        r2l_fs_fetch_bf<U16>(a, img, r2l_mirror(R2L_NH(qf - 1), a.H), x0, le, re, lane, s0);
        r2l_fs_fetch_bf<U16>(a, img, r2l_mirror(R2L_NH(qf), a.H), x0, le, re, lane, s1);
        r2l_fs_convert<U16>(a, F, s0, le, re, st.v[2]);
        r2l_fs_convert<U16>(a, F, s1, le, re, st.v[0]);
      }
      R2LFsStage pf[PF];  // ring: step s consumes pf[s % PF] (raw row q + 1) and refills it with row q + 1 + PF
      R2L_PRAGMA_UNROLL
      for (int i = 0; i < PF; ++i) {
        const int rr = (qf + 1 + i < q1) ? qf + 1 + i : q1;
        r2l_fs_fetch_bf<U16>(a, img, r2l_mirror(R2L_NH(rr), a.H), x0, le, re, lane, pf[i]);
      }
      const int nsteps = q1 - qf;
#if defined(R2L_EXP_STAMPS) && !defined(R2L_EMUL)
      // (diagnostic builds: the time every step of a few sampled workgroups ends, tl[8192 + 64 * (bid / 128) + step])
#define R2L_FS_STEP_STAMP(s_)                                                                     \
  if (a.tl && (bid & 127) == 5 && bid < 2048 && (s_) < 64 && wave == 0 && r2l_lane_id() == 0)      \
    a.tl[8192 + 64 * (bid >> 7) + (s_)] = __builtin_amdgcn_s_memrealtime();
#else
#define R2L_FS_STEP_STAMP(s_)
#endif
      for (int sb = 0; sb < nsteps; sb += 6) {
        R2L_PROGRESS_PRIO(sb, nsteps);
#define R2L_FS_STEP(K)                                                                                          \
  {                                                                                                             \
    const int q = qf + sb + K;                                                                                  \
    r2l_fs_convert<U16>(a, F, pf[K % PF], le, re, st.v[(K + 1) % 3]);                                           \
    {                                                                                                           \
      const int rr_ = (q + 1 + PF < q1) ? q + 1 + PF : q1;                                                      \
      r2l_fs_fetch_bf<U16>(a, img, r2l_mirror(R2L_NH(rr_), a.H), x0, le, re, lane, pf[K % PF]);                         \
    }                                                                                                           \
    r2l_fs_step<NW, U16, K, EPI, (K + 1) & 1, true>(a, st, q, y0, y1, le, re, wave, lane, ex, fifo,             \
                                                   SONLY ? nullptr : ob, ypb, plane, x0, store_ok, mean, istd, smask); \
    R2L_FS_STEP_STAMP(sb + K)                                                                                   \
  }
        R2L_FS_STEP(0)
        R2L_FS_STEP(1)
        R2L_FS_STEP(2)
        R2L_FS_STEP(3)
        R2L_FS_STEP(4)
        R2L_FS_STEP(5)
#undef R2L_FS_STEP
      }
    }
#else
    // first luma row computed: qf = y0 - 3 (Y'(y0-2) needs Y(y0-3)).  The loop is unrolled 6-fold with q = qb + K,
    // qb a multiple of 6, so that every window slot is a compile-time index; the first pass enters it at K0 = qf - q0
    // (the steps before qf are skipped, not computed), and the warm-up rows go to the slots that position implies.
    const int qf = y0 - 3;
    const int q0 = (qf >= 0) ? qf - qf % 6 : -(((-qf) + 5) / 6) * 6;
    const int k0 = qf - q0;
    const int q1 = y1 + 4;  // exclusive: output row y1-1 leaves at step q = y1+3
    {
      R2LFsStage s0, s1;
      r2l_fs_fetch<U16>(a, img, r2l_mirror(R2L_NH(qf - 1), a.H), x0, le, re, lane, s0);
      r2l_fs_fetch<U16>(a, img, r2l_mirror(R2L_NH(qf), a.H), x0, le, re, lane, s1);
      switch (k0 % 3) {  // rows qf-1, qf -> slots (k0 + 2) % 3, k0 % 3
        case 0:
          r2l_fs_convert<U16>(a, F, s0, le, re, st.v[2]);
          r2l_fs_convert<U16>(a, F, s1, le, re, st.v[0]);
          break;
        case 1:
          r2l_fs_convert<U16>(a, F, s0, le, re, st.v[0]);
          r2l_fs_convert<U16>(a, F, s1, le, re, st.v[1]);
          break;
        default:
          r2l_fs_convert<U16>(a, F, s0, le, re, st.v[1]);
          r2l_fs_convert<U16>(a, F, s1, le, re, st.v[2]);
          break;
      }
    }
    R2LFsStage pf[PF];  // ring: step K consumes pf[K % PF] (raw row q + 1) and refills it with row q + 1 + PF
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < PF; ++i) {
      R2LFsStage t;
      r2l_fs_fetch<U16>(a, img, r2l_mirror(R2L_NH(qf + 1 + i), a.H), x0, le, re, lane, t);
      R2L_PRAGMA_UNROLL
      for (int j = 0; j < PF; ++j)
        if ((k0 + i) % PF == j) pf[j] = t;
    }
    for (int qb = q0; qb < q1; qb += 6) {
      R2L_PROGRESS_PRIO(qb - q0, q1 - q0);
#define R2L_FS_STEP(K)                                                                                          \
  if (qb + K >= qf && qb + K < q1) {                                                                            \
    const int q = qb + K;                                                                                       \
    R2L_PROGRESS_PRIO_STEP(q - q0, q1 - q0);                                                                    \
    r2l_fs_convert<U16>(a, F, pf[K % PF], le, re, st.v[(K + 1) % 3]);                                           \
    if (R2L_FS_BF_FETCH) {                                                                                      \
      const int rr_ = (q + 1 + PF < q1) ? q + 1 + PF : q1;                                                      \
      r2l_fs_fetch_bf<U16>(a, img, r2l_mirror(R2L_NH(rr_), a.H), x0, le, re, lane, pf[K % PF]);                         \
    } else if (q + 1 + PF <= q1)                                                                                \
      r2l_fs_fetch<U16>(a, img, r2l_mirror(R2L_NH(q + 1 + PF), a.H), x0, le, re, lane, pf[K % PF]);                     \
    r2l_fs_step<NW, U16, K, EPI>(a, st, q, y0, y1, le, re, wave, lane, ex, fifo, ob, ypb, plane, x0, store_ok, mean, istd, smask); \
  }
      R2L_FS_STEP(0)
      R2L_FS_STEP(1)
      R2L_FS_STEP(2)
      R2L_FS_STEP(3)
      R2L_FS_STEP(4)
      R2L_FS_STEP(5)
#undef R2L_FS_STEP
    }
#endif
    if (NW > 1) R2L_LDS_BARRIER();  // exchange buffers free for the next item
    if (SONLY || a.stat_partial) r2l_fs_lane_sums(st.acc, st.piv, store_ok ? 4.0 * (double)(y1 - y0) : 0.0, store_ok, lane, tots);
  }
#if defined(R2L_EXP_STAMPS) && !defined(R2L_EMUL)
  const unsigned long long tls2_ = __builtin_amdgcn_s_memrealtime();
#endif
  if (SONLY || a.stat_partial) {
    const int tid = SONLY ? wave * 64 + r2l_lane_id() : (int)threadIdx.x;
    r2l_fs_stats_finish<NW, NT>(a, bid, nblk, tid, wave, tots, red);
  }
#if defined(R2L_EXP_STAMPS) && !defined(R2L_EMUL)
  if (a.tl && bid < 2048 && wave == 0 && r2l_lane_id() == 0) {
    a.tl[4 * bid] = tls0_;
    a.tl[4 * bid + 1] = tls1_;
    a.tl[4 * bid + 2] = tls2_;
    a.tl[4 * bid + 3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ================================================================================================
// The APPLY pass of train-mode BatchNorm when the statistics pass has kept Y' (R2L_F_KEEP_LUMA): everything in front of
// the blur is already there, so the second pass over the batch is  out = BN(gamma(M * (blur5x5(Y'), U, V)))  with (U, V)
// the 3x3 chroma stencils of the raw frame -- no luma stencil, no sharpen, no halo rows to re-compute (a band re-READS 4
// rows of Y' and 2 raw rows, nothing else), no strip-edge exchange: the neighbour columns of a strip's first and last
// lane are two 8-byte loads.  Wavefronts are independent (one per workgroup, no LDS, no barrier); a wavefront owns
// (image, 256-column strip, band of rows) and walks down the band with V (3 rows x 6) and Y' (6-slot ring x 8) in
// registers.  The arithmetic of an output row is the streaming kernel's, function by function (r2l_fs_stencil_parity,
// r2l_blur_row2w with the same edge weight sets, r2l_fs_colour), on the same Y' values: the two apply passes agree bit
// for bit (tests/test_gpu_parity.py: test_apply_pass_reads_the_luma_plane_the_statistics_pass_kept).  20 B/px of traffic
// (4 raw + 4 Y' + 12 out) against 16 B/px and 31 % fewer vector instructions (16.5 M against 23.9 M per launch at
// 64x512x512): the pass is bound by its HBM traffic -- 65 us = 5.1 TB/s; 29 us with the stores taken out, 44 us with the
// loads taken out (profiles/r03_apply_kept.txt) -- where the streaming apply pass is bound by instruction issue (72.5 us).
struct R2LFaStage {  // one Y' row in flight: the lane's 4 values + the pair beyond the strip edge (first / last lane)
  r2l_f4 c;
  r2l_f2 e;
};
// The fetches are BRANCH-FREE (every lane loads an edge value from an in-row address; only the first and last lane use
// theirs) and the loop below has no conditional memory operation but the predicated stores: hipcc's s_waitcnt insertion
// then knows how many younger loads follow the row a step consumes and waits with vmcnt(8 .. 13) -- with a conditional
// fetch or a conditionally executed step anywhere in the loop it falls back to vmcnt(0), which waits for the loads just
// issued AND for the output stores of the previous row.
template <bool U16>
R2L_HD void r2l_fa_fetch_raw(const R2LFwdStreamArgs& a, size_t img0, int ym, int x0, bool le, bool re, int lane,
                             R2LFsStage& s) {
  const size_t e = img0 + (size_t)ym * a.W + x0;
  const int eo = (lane < 32) ? (le ? 0 : -1) : (re ? 3 : 4);
  s.ym = ym;
  if (U16) {
    const unsigned short* r = a.raw.u16 + e;
    const r2l_f2 b = *(const r2l_f2*)r;
    s.c.x = b.x;
    s.c.y = b.y;
    s.e = r2l_u2f((unsigned)r[eo]);
  } else {
    const float* r = a.raw.f32 + e;
    s.c = r2l_stream_load_f4(r);
    s.e = r[eo];
  }
}
// NT: the caller is the LAST reader of this plane in the step -- the row goes around the caches (the edge pair, which the neighbouring
// strip's wavefront also loads as part of its row, stays a plain load)
template <bool NT = false>
R2L_HD void r2l_fa_fetch(const float* ypimg, int r, int H, int W, int x0, bool le, bool re, int lane, R2LFaStage& s) {
  const int rc = r < 0 ? 0 : (r >= H ? H - 1 : r);  // rows outside the image are zeroed when the row is built
  const float* p = ypimg + (size_t)rc * W + x0;
  const int eo = (lane < 32) ? (le ? 0 : -2) : (re ? 2 : 4);
  s.c = NT ? r2l_load_f4_nt(p) : r2l_stream_load_f4(p);
  s.e = *(const r2l_f2*)(p + eo);
}
// staged row -> 8 values, columns x0-2 .. x0+5 (mirror padding of the blur at the image edges, :165 reflect)
R2L_HD void r2l_fa_build(const R2LFaStage& s, bool rin, bool le, bool re, float o[8]) {
  const float c0 = rin ? s.c.x : 0.f, c1 = rin ? s.c.y : 0.f, c2 = rin ? s.c.z : 0.f, c3 = rin ? s.c.w : 0.f;
  const float e0 = rin ? s.e.x : 0.f, e1 = rin ? s.e.y : 0.f;
  const float l2 = r2l_wshr(c2, e0), l1 = r2l_wshr(c3, e1);
  const float r1 = r2l_wshl(c0, e0), r2 = r2l_wshl(c1, e1);
  o[0] = le ? c2 : l2;
  o[1] = le ? c1 : l1;
  o[2] = c0;
  o[3] = c1;
  o[4] = c2;
  o[5] = c3;
  o[6] = re ? c2 : r1;
  o[7] = re ? c1 : r2;
}
// a scalar 1 hipcc cannot see through: `if (r2l_opaque_true())` around a step's arithmetic makes the step a basic block
// of its own (STATS: without a store or a branch the six unrolled steps are ONE block, scheduled as one: 215 VGPRs)
R2L_HD bool r2l_opaque_true() {
  int one = 1;
#ifndef R2L_EMUL
  asm volatile("" : "+s"(one));
#endif
  return one != 0;
}
struct R2LFaState {
  float v[3][6];   // V rows (slot = row mod 3)
  float yp[6][8];  // Y' rows (slot = row mod 6)
};
template <int K, bool EPI, bool STATS, bool BNR = false>
R2L_HD void r2l_fa_step(const R2LFwdStreamArgs& a, R2LFaState& st, r2l_p2* acc, float* piv, int y, int y0,
                        bool store_ok, float* ob, unsigned plane, int x0, const float mean[3], const float istd[3],
                        const r2l_p2 (*gk)[2] = nullptr, float rowf = 0.f) {
  // (the weights of one section at a time: chroma, blur, colour code -- r2l_opaque_after)
  // the previous step's last result (STATS: there is no store to end a step with) / this step's newest window value
  R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque_after(a.F, (STATS || BNR) ? acc[5][1] : st.yp[(K + 2) % 6][2]));
  constexpr int PY = K & 1;
  const int H = a.H;
  const float* vu = st.v[(K + 2) % 3];  // V(y-1)
  const float* vm = st.v[K % 3];        // V(y)
  const float* vl = st.v[(K + 1) % 3];  // V(y+1)
  r2l_p2 u[2], v[2], ypp[2];
  r2l_fs_stencil_parity(vu, vm, vl, F.AU2[PY], u);
  r2l_fs_stencil_parity(vu, vm, vl, F.AV2[PY], v);
  {
    R2LFoldedRef Fb = R2L_FOLDED_REF(r2l_opaque_after(a.F, v[1][1]));
    float yw[5][8];  // window rows y-2 .. y+2 sit in ring slots K+4 .. K+8
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 5; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 8; ++j) yw[i][j] = st.yp[(K + 4 + i) % 6][j];
    const int set = (y < 2) ? y : (y - (H - 2)) + 2;
#ifdef R2L_EXP_CONST_WEIGHTS
    const float* w25 = &Fb.blur[0];  // (timing only: no border sets)
    (void)set;
#else
    const R2L_CONSTAS float* w25 =
        (y >= 2 && y < H - 2) ? &Fb.blur[0] : &Fb.blur_edge[0][0] + 25 * set;
#endif
    r2l_blur_row2w(yw, w25, ypp);
  }
  R2LFoldedRef Fc = R2L_FOLDED_REF(r2l_opaque_after(a.F, ypp[1][1]));
  if (BNR)
    r2l_fs_colour<false, 3>(a, Fc, acc, piv, ypp, u, v, y, y0, x0, nullptr, plane, false, mean, istd, rowf, false, gk);
  else
    r2l_fs_colour<EPI, STATS ? 2 : 0>(a, Fc, acc, piv, ypp, u, v, y, y0, x0, ob, plane, store_ok, mean, istd,
                                      store_ok ? 1.f : 0.f, K == 0 && y == y0);
}

#ifndef R2L_FA_PF
#define R2L_FA_PF 2
#endif
// STATS: no output -- the statistics of train-mode BatchNorm from the kept plane (the colour code's sums about the lane's
// pivot, then the streaming kernel's reduction: r2l_fs_stats_finish); NWV wavefronts per workgroup, every one of them
// walking its own work items (bid * NWV + wave, + nblk * NWV, ...), so that <= R2L_MAX_BLOCKS partials fill the chip.
// The apply pass runs one wavefront per workgroup and one item per wavefront.
#define R2L_FA_LDS_FLOATS(NWV, STATS) ((STATS) ? 16 + R2L_FS_RED_FLOATS(NWV) + 12 * (NWV) : 4)
// one work item = (image, band, strip), strips fastest: neighbouring items share halo rows and strip edges
template <bool U16, bool EPI, bool STATS>
R2L_HD void r2l_fa_item(const R2LFwdStreamArgs& a, int item, int lane, const float mean[3], const float istd[3],
                        double* tots) {
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  const int nstrip = (a.W + 255) >> 8;
  const unsigned plane = (unsigned)a.H * (unsigned)a.W;
  const int strip = item % nstrip, ib = item / nstrip;
  const int band = ib % a.nband, b = ib / a.nband;
  const int xs = strip * 256 + 4 * lane;
  const bool in_w = xs < a.W;
  const int x0 = in_w ? xs : a.W - 4;
  const bool le = x0 == 0, re = x0 + 4 >= a.W;
  // bands start on multiples of 6 rows (the host rounds band_h): the ring slot of a row, row mod 6, is then the unroll
  // position K of its step in every band, and the warm-up is the same four steps everywhere
  const int y0 = band * a.band_h;
  const int y1 = (y0 + a.band_h < a.H) ? y0 + a.band_h : a.H;
  const size_t img = (size_t)b * plane;
  float* ob = STATS ? nullptr : a.out + (size_t)b * 3 * plane;
  if (!STATS) __builtin_assume(ob != nullptr);
  const float* ypimg = a.yp_in + img;
  R2LFaState st;
  r2l_p2 acc[6];
  float piv[3] = {0.5f, 0.5f, 0.5f};
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 6; ++i) acc[i] = r2l_splat2(0.f);
  constexpr int PF = R2L_FA_PF;
  static_assert(6 % PF == 0, "the prefetch ring is indexed by the unroll position");
  R2LFsStage pf[PF];   // ring: step K consumes pf[K % PF] (raw row q + 1) and refills it with row q + 1 + PF
  R2LFaStage pfy[PF];  // likewise Y' row q + 2
  // step q builds V(q+1) and Y'(q+2); from q = y0 on it also finishes output row q.  Warm-up: q = y0-4 .. y0-1 = K 2 .. 5.
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < PF; ++i) {
    r2l_fa_fetch_raw<U16>(a, img, r2l_mirror(R2L_NH(y0 - 3 + i), a.H), x0, le, re, lane, pf[(2 + i) % PF]);
    r2l_fa_fetch(ypimg, R2L_NH(y0 - 2 + i), a.H, a.W, x0, le, re, lane, pfy[(2 + i) % PF]);
  }
#define R2L_FA_LOAD_STEP(K, q)                                                                          \
  {                                                                                                     \
    r2l_fs_convert<U16>(a, F, pf[(K) % PF], le, re, st.v[((K) + 1) % 3]);                               \
    r2l_fa_build(pfy[(K) % PF], (unsigned)((q) + 2) < (unsigned)a.H, le, re, st.yp[((K) + 2) % 6]);     \
    r2l_fa_fetch_raw<U16>(a, img, r2l_mirror(R2L_NH((q) + 1 + PF), a.H), x0, le, re, lane, pf[(K) % PF]);       \
    r2l_fa_fetch(ypimg, R2L_NH((q) + 2 + PF), a.H, a.W, x0, le, re, lane, pfy[(K) % PF]);                       \
  }
  R2L_FA_LOAD_STEP(2, y0 - 4)
  R2L_FA_LOAD_STEP(3, y0 - 3)
  R2L_FA_LOAD_STEP(4, y0 - 2)
  R2L_FA_LOAD_STEP(5, y0 - 1)
  // every group of 6 steps runs in full: rows past the band's end (last band of an image whose height is not a multiple
  // of 6) are computed from clamped fetches and neither stored nor counted
  for (int qb = y0; qb < y1; qb += 6) {
    R2L_PROGRESS_PRIO(qb - y0, y1 - y0);
#define R2L_FA_STEP(K)                                                                                  \
  {                                                                                                     \
    const int q = qb + K;                                                                               \
    R2L_PROGRESS_PRIO_STEP(q - y0, y1 - y0);                                                            \
    R2L_FA_LOAD_STEP(K, q)                                                                              \
    if (!STATS || r2l_opaque_true())                                                                    \
      r2l_fa_step<K, EPI, STATS>(a, st, acc, piv, q, y0, in_w && q < y1, ob, plane, x0, mean, istd);    \
  }
    R2L_FA_STEP(0)
    R2L_FA_STEP(1)
    R2L_FA_STEP(2)
    R2L_FA_STEP(3)
    R2L_FA_STEP(4)
    R2L_FA_STEP(5)
#undef R2L_FA_STEP
  }
#undef R2L_FA_LOAD_STEP
  if (STATS) r2l_fs_lane_sums(acc, piv, in_w ? 4.0 * (double)(y1 - y0) : 0.0, in_w, lane, tots);
}
template <bool U16, bool EPI, bool STATS, int NWV>
R2L_BLOCKFN void r2l_fwd_apply_block(const R2LFwdStreamArgs& a, int bid, int nblk, float* lds) {
  // (the wavefront index as a SCALAR: the work item, its rows and the weight set of a row must not look lane-dependent)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  float mean[3] = {0.f, 0.f, 0.f}, istd[3] = {1.f, 1.f, 1.f};
  R2L_TL_BEGIN(a, bid)
  if (!STATS) {
    if (a.bn) {
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < 3; ++k) {
        mean[k] = a.bn[k];
        istd[k] = a.bn[3 + k];
      }
    }
    // one item per wavefront; NWV wavefronts per workgroup only because the dispatcher starts ~250 workgroups per
    // microsecond: 4,096 single-wavefront workgroups take 15 us to launch (tests/timeline_fwd.py)
    const int item = r2l_xcd_window(bid, nblk, a.xcdm) * NWV + wave;
    if (item < a.nitems) r2l_fa_item<U16, EPI, false>(a, item, lane, mean, istd, nullptr);
    R2L_TL_END(a, bid)
  } else {
    float* red = lds + 16;
    double* tots = (double*)(lds + 16 + R2L_FS_RED_FLOATS(NWV)) + wave * 6;
    if (lane < 6) tots[lane] = 0.0;
    R2L_PRAGMA_NOUNROLL
    for (int item = r2l_xcd_window(bid, nblk, a.xcdm) * NWV + wave; item < a.nitems; item += nblk * NWV)
      r2l_fa_item<U16, false, true>(a, item, lane, mean, istd, tots);
    R2L_TL_END(a, bid)
    r2l_fs_stats_finish<NWV, NWV * 64>(a, bid, nblk, tid, wave, tots, red);
  }
}

// ================================================================================================
// The LUMA pass: raw -> Y' = sharpen(Y) (pipeline_torch.py:183-195), the plane the two passes above (statistics, apply)
// and kernel B1 of the backward read.  Independent wavefronts again: the sharpen needs Y one column beyond the lane's
// four on each side; between lanes that is a DPP shift, and the strip's first and last lane compute the one column
// their neighbour strip owns THEMSELVES -- one extra pair per row for all lanes, (Y(x0+4), Y(x0-1)) = (even, odd column)
// like every other pair, from a raw window 8 columns wide (edge pairs: one 8-byte load per row) -- instead of an
// exchange through LDS and a barrier per row.  Same functions, same order of operations as r2l_fs_step: the plane is
// bit-identical to the one the streaming forward keeps.  8 B/px (4 in, 4 out) + 4 raw halo rows per band.
struct R2LFlStage {
  r2l_f4 c;
  r2l_f2 e;  // columns (x0-2, x0-1) in the first half of the wavefront, (x0+4, x0+5) in the second: (even, odd)
  int ym;
};
template <bool U16, bool NT = false>
R2L_HD void r2l_fl_fetch(const R2LFwdStreamArgs& a, size_t img0, int ym, int x0, bool le, bool re, int lane,
                         R2LFlStage& s) {
  const size_t e = img0 + (size_t)ym * a.W + x0;
  const int eo = (lane < 32) ? (le ? 0 : -2) : (re ? 2 : 4);
  s.ym = ym;
  if (U16) {
    const unsigned short* r = a.raw.u16 + e;
    const r2l_f2 b = *(const r2l_f2*)r;
    s.c.x = b.x;
    s.c.y = b.y;
    s.e.x = *(const float*)(r + eo);  // two 16-bit values
  } else {
    const float* r = a.raw.f32 + e;
    s.c = NT ? r2l_load_f4_nt(r) : r2l_stream_load_f4(r);
    s.e = *(const r2l_f2*)(r + eo);
  }
}
// staged row -> 8 black-level-corrected values, columns x0-2 .. x0+5 (mirror padding at the image edges)
template <bool U16>
R2L_HD void r2l_fl_convert(const R2LFwdStreamArgs& a, R2LFoldedRef F, const R2LFlStage& s, bool le, bool re, float v[6],
                           r2l_p2 xp[3]) {
  float c0, c1, c2, c3, e0, e1;
  if (U16) {
    const unsigned lo = r2l_f2u(s.c.x), hi = r2l_f2u(s.c.y), ee = r2l_f2u(s.e.x);
    c0 = r2l_raw_decode(lo & 0xffffu, a.raw);
    c1 = r2l_raw_decode(lo >> 16, a.raw);
    c2 = r2l_raw_decode(hi & 0xffffu, a.raw);
    c3 = r2l_raw_decode(hi >> 16, a.raw);
    e0 = r2l_raw_decode(ee & 0xffffu, a.raw);
    e1 = r2l_raw_decode(ee >> 16, a.raw);
  } else {
    c0 = s.c.x;
    c1 = s.c.y;
    c2 = s.c.z;
    c3 = s.c.w;
    e0 = s.e.x;
    e1 = s.e.y;
  }
  const float be = (s.ym & 1) ? F.bl[2] : F.bl[0], bo = (s.ym & 1) ? F.bl[3] : F.bl[1];
  c0 -= be;
  c1 -= bo;
  c2 -= be;
  c3 -= bo;
  e0 -= be;
  e1 -= bo;
  const float l2 = r2l_wshr(c2, e0), l1 = r2l_wshr(c3, e1);
  const float r1 = r2l_wshl(c0, e0), r2 = r2l_wshl(c1, e1);
  // column -2 is column 2, -1 is 1; W is W-2, W+1 is W-3
  v[0] = le ? c1 : l1;
  v[1] = c0;
  v[2] = c1;
  v[3] = c2;
  v[4] = c3;
  v[5] = re ? c2 : r1;
  xp[0] = r2l_mk2(c3, le ? c2 : l2);
  xp[1] = r2l_mk2(v[5], v[0]);
  xp[2] = r2l_mk2(re ? c1 : r2, c0);
}
struct R2LFlState {
  float v[3][6];  // V rows (slot = row mod 3), columns x0-1 .. x0+4 (the window of r2l_fs_step)
  // ... and the inputs of the extra pair, (column x0+3+j, column x0-2+j), as PAIRS: read as floats out of an array (one
  // 8-wide row, or two 3-wide ones) hipcc widens the loads into overlapping 8-byte ones, cannot take the array apart any
  // more and parks it in LDS -- whose addressing reads the dispatch packet: workgroups then start 70 per us instead of 1,400
  r2l_p2 xp[3][3];
  float y[3][6];  // Y rows (slot = row mod 3), columns x0-1 .. x0+4
};
// step y (K = y mod 6): V(y+2) is in place; Y(y+1) from V(y .. y+2) [LUMA]; Y'(y) from Y(y-1 .. y+1), stored [OUT]
template <int K, bool LUMA, bool OUT>
R2L_HD void r2l_fl_step(const R2LFwdStreamArgs& a, R2LFlState& st, int y, bool le, bool re, bool store_ok, float* ypb,
                        int x0) {
  R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque(a.F));
  if (LUMA) {
    constexpr int PY = (K + 1) & 1;
    float* yq = st.y[(K + 1) % 3];
    r2l_p2 o[2];
    r2l_fs_stencil_parity(st.v[K % 3], st.v[(K + 1) % 3], st.v[(K + 2) % 3], F.AY2[PY], o);  // V(y), V(y+1), V(y+2)
    // the columns beyond the lane's four, as ONE more pair: (Y(x0+4), Y(x0-1)) -- even column, odd column
    r2l_p2 e = r2l_splat2(0.f);
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j)
      e = r2l_pfma(r2l_mk2(F.AY2[PY][i * 3 + j][0], F.AY2[PY][i * 3 + j][1]), st.xp[(K + i) % 3][j], e);
    const bool qin = (unsigned)(y + 1) < (unsigned)a.H;  // zero padding of the sharpen conv (:162 padding=1)
    yq[0] = (qin && !le) ? e[1] : 0.f;
    yq[1] = qin ? o[0][0] : 0.f;
    yq[2] = qin ? o[0][1] : 0.f;
    yq[3] = qin ? o[1][0] : 0.f;
    yq[4] = qin ? o[1][1] : 0.f;
    yq[5] = (qin && !re) ? e[0] : 0.f;
  }
  if (OUT) {
    r2l_p2 o[2];  // rows Y(y-1), Y(y), Y(y+1) = slots K+2, K, K+1
    r2l_fs_stencil_plain(st.y[(K + 2) % 3], st.y[K % 3], st.y[(K + 1) % 3], F.sharp, o);
    if (store_ok) {
      r2l_f4 s4;
      s4.x = o[0][0];
      s4.y = o[0][1];
      s4.z = o[1][0];
      s4.w = o[1][1];
      *(r2l_f4*)(ypb + (unsigned)y * (unsigned)a.W + (unsigned)x0) = s4;
    }
  }
}
#ifndef R2L_FL_PF
#define R2L_FL_PF 3
#endif
template <bool U16, int NWV>
R2L_BLOCKFN void r2l_fwd_luma_block(const R2LFwdStreamArgs& a, int bid, int nblk, float* lds) {
  (void)lds;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int item = r2l_xcd_window(bid, nblk, a.xcdm) * NWV + wave;  // one item per wavefront (NWV per workgroup: see r2l_fwd_apply_block)
  if (item >= a.nitems) return;
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  const int nstrip = (a.W + 255) >> 8;
  const int strip = item % nstrip, ib = item / nstrip;
  const int band = ib % a.nband, b = ib / a.nband;
  const int xs = strip * 256 + 4 * lane;
  const bool in_w = xs < a.W;
  const int x0 = in_w ? xs : a.W - 4;
  const bool le = x0 == 0, re = x0 + 4 >= a.W;
  const int y0 = band * a.band_h;  // a multiple of 6
  const int y1 = (y0 + a.band_h < a.H) ? y0 + a.band_h : a.H;
  const size_t img = (size_t)b * a.H * a.W;
  float* ypb = a.yp_out + img;
  R2LFlState st;
  R2L_TL_BEGIN(a, bid)
  constexpr int PF = R2L_FL_PF;
  static_assert(6 % PF == 0, "the prefetch ring is indexed by the unroll position");
  R2LFlStage pf[PF];  // ring: step K consumes pf[K % PF] (raw row y + 2) and refills it with row y + 2 + PF
  // warm-up: y = y0-4 (V(y0-2)), y0-3 (V(y0-1)), y0-2 (V(y0), Y(y0-1)), y0-1 (V(y0+1), Y(y0)) = K 2 .. 5
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < PF; ++i)
    r2l_fl_fetch<U16>(a, img, r2l_mirror(R2L_NH(y0 - 2 + i), a.H), x0, le, re, lane, pf[(2 + i) % PF]);
#define R2L_FL_STEP(K, y, LUMA, OUT)                                                                    \
  {                                                                                                     \
    r2l_fl_convert<U16>(a, F, pf[(K) % PF], le, re, st.v[((K) + 2) % 3], st.xp[((K) + 2) % 3]);                      \
    r2l_fl_fetch<U16>(a, img, r2l_mirror(R2L_NH((y) + 2 + PF), a.H), x0, le, re, lane, pf[(K) % PF]);           \
    r2l_fl_step<K, LUMA, OUT>(a, st, y, le, re, in_w && (y) < y1, ypb, x0);                             \
  }
  R2L_FL_STEP(2, y0 - 4, false, false)
  R2L_FL_STEP(3, y0 - 3, false, false)
  R2L_FL_STEP(4, y0 - 2, true, false)
  R2L_FL_STEP(5, y0 - 1, true, false)
  for (int qb = y0; qb < y1; qb += 6) {
    R2L_PROGRESS_PRIO(qb - y0, y1 - y0);
    R2L_FL_STEP(0, qb + 0, true, true)
    R2L_FL_STEP(1, qb + 1, true, true)
    R2L_FL_STEP(2, qb + 2, true, true)
    R2L_FL_STEP(3, qb + 3, true, true)
    R2L_FL_STEP(4, qb + 4, true, true)
    R2L_FL_STEP(5, qb + 5, true, true)
  }
#undef R2L_FL_STEP
  R2L_TL_END(a, bid)
}

#endif  // !R2L_SERIAL
