// r2l_param_stream_bwd.h -- kernel B2 of the backward (r2l_bwd2_block, r2l_param_kernels.h) as a ROW-STREAMING
// kernel: adjoint of the mirror-padded 5x5 blur, adjoint of the zero-padded 3x3 sharpen, d/d sharpening_filter.weight,
// the folded luma-stencil sums, then the final reduction of both backward kernels' partials and the unfold into the
// 132 parameter gradients -- organised like the streaming forward (r2l_param_stream.h) instead of as LDS tiles.
//
// A wavefront owns 256 columns (4 per lane) and a band of rows.  Two inputs stream in, one row each per step g:
// row g of G = dL/dY'' (written by kernel B1) and raw row g-1.  In registers (float32, packed pairs):
//     HP'  five partial rows of dL/dY' = blur^T G: the arriving row of G (8 wide, transient) is scattered into the
//          rows g-2 .. g+2 it contributes to, instead of five rows of G being held for a gather (20 registers
//          instead of 40); row g-2 is complete after step g
//     HP   the last 3 complete rows of dL/dY', 6 wide (zero outside the image)         -> gY(g-3) = dL/dY(g-3)
//     V    ring of the last raw rows minus black level, 6 wide (mirror-extended)        -> Y(g-2), and d/dA_Y
//     Y    the last 3 luma rows, 6 wide (zero outside the image)                        -> d/d sharpen
// Row r = g-3 is the row whose sums are accumulated in step g.  Neighbour columns: G's come from memory (lane 0 / 63
// load the neighbouring strip's pair), HP's and Y's from the neighbouring lanes (DPP) and, at the strip edges, from
// the neighbouring wavefront through LDS: 4 floats per wavefront and row, one barrier per row.
// The mirror padding's adjoint needs no extra rows or columns: rows 1, 2, H-3, H-2 take weight sets with the
// contributions of their mirror images folded in (R2LFolded::adj), and the first / last lane of an image row adds
// the two folded columns as 15 extra multiply-adds.  Needs H >= 6.
#pragma once
#include "r2l_param_stream.h"

#ifndef R2L_EMUL

#ifndef R2L_BS_PF
#define R2L_BS_PF 2
#endif
// A step reads ~70 weights through the scalar cache.  Left alone, hipcc lifts all those loads to the top of the
// (branch-free) step and spills scalar registers into vector lanes.  The weights are therefore staged by hand: each
// group of taps waits for its own weights (R2L_BS_PIN*: an empty asm that needs them in registers), then issues the
// next group's loads (through a pointer laundered by the same kind of asm, so the loads cannot move up), then
// computes -- the loads of group n+1 fly during the arithmetic of group n.
typedef const __attribute__((address_space(4))) float* r2l_cfp;
R2L_HD r2l_cfp r2l_bs_launder(r2l_cfp p) {
  asm volatile("" : "+s"(p));
  return p;
}
#define R2L_BS_S5(w, o) "+s"(w[o]), "+s"(w[o + 1]), "+s"(w[o + 2]), "+s"(w[o + 3]), "+s"(w[o + 4])
#define R2L_BS_S4(w, o) "+s"(w[o]), "+s"(w[o + 1]), "+s"(w[o + 2]), "+s"(w[o + 3])
#define R2L_BS_PIN5(w)                    \
  asm volatile("" : R2L_BS_S5(w, 0));     \
  __builtin_amdgcn_sched_barrier(0)
#define R2L_BS_PIN9(w)                                        \
  asm volatile("" : R2L_BS_S5(w, 0), R2L_BS_S4(w, 5));        \
  __builtin_amdgcn_sched_barrier(0)
#define R2L_BS_PIN18(w)                                                                        \
  asm volatile("" : R2L_BS_S5(w, 0), R2L_BS_S5(w, 5), R2L_BS_S5(w, 10), R2L_BS_S4(w, 14));   \
  __builtin_amdgcn_sched_barrier(0)
#define R2L_BS_PINV2(x, y) asm volatile("" : "+v"(x), "+v"(y))

struct R2LBwd2StreamArgs {
  R2LRaw raw;
  const R2LFolded* F;
  const float* gypp;  // (B,H,W) dL/dY'' from kernel B1
  float* partial;     // [R2L_B2_NACC][nblk]
  int B, H, W;
  int nband, band_h, nitems;
  R2LTree tree;  // in-kernel final reduction of B1's and B2's partials + unfold -> grad_params
  const float* params;
  float* grad_params;
};

#define R2L_BS_EX 8
#define R2L_BS_RED_FLOATS(NW) 3200  // >= 512 + R2L_NSUMS * R2L_TREE_GROUP floats of tree scratch; wave sums [NW][64] before
#define R2L_BS_LDS_FLOATS(NW) (2 * (NW) * R2L_BS_EX + 16 + R2L_BS_RED_FLOATS(NW) + 2 * R2L_NSUMS + 2 * R2L_UNFOLD_TG + R2L_P_COUNT + 16)

struct R2LBsGStage {
  r2l_f4 c;
  r2l_f2 e;  // lane 0: columns x0-2, x0-1; lane 63: columns x0+4, x0+5 (when they exist)
  int row;   // unclamped: rows outside the image convert to zeros
};
R2L_HD void r2l_bs_fetch_g(const R2LBwd2StreamArgs& a, size_t img0, int row, int x0, bool le, bool re, int lane,
                           R2LBsGStage& s) {
  s.e.x = s.e.y = 0.f;
  s.row = row;
  const int rc = row < 0 ? 0 : (row >= a.H ? a.H - 1 : row);  // (no branch: the load is always issued)
  const float* r = a.gypp + img0 + (size_t)rc * a.W + x0;
  s.c = *(const r2l_f4*)r;
  if ((lane == 0 && !le) || (lane == 63 && !re)) s.e = *(const r2l_f2*)(lane == 0 ? r - 2 : r + 4);
}
// staged row -> 8 values, columns x0-2 .. x0+5, zero outside the image
R2L_HD void r2l_bs_convert_g(const R2LBsGStage& s, int H, bool le, bool re, float w[8]) {
  const bool in = (unsigned)s.row < (unsigned)H;
  const float c0 = in ? s.c.x : 0.f, c1 = in ? s.c.y : 0.f, c2 = in ? s.c.z : 0.f, c3 = in ? s.c.w : 0.f;
  const float e0 = in ? s.e.x : 0.f, e1 = in ? s.e.y : 0.f;
  const float l2 = r2l_wshr(c2, e0), l1 = r2l_wshr(c3, e1);
  const float r1 = r2l_wshl(c0, e0), r2 = r2l_wshl(c1, e1);
  w[0] = le ? 0.f : l2;
  w[1] = le ? 0.f : l1;
  w[2] = c0;
  w[3] = c1;
  w[4] = c2;
  w[5] = c3;
  w[6] = re ? 0.f : r1;
  w[7] = re ? 0.f : r2;
}

struct R2LBsState {
  r2l_p2 part[6][2];  // partial rows of dL/dY' (slot = row mod 6), own 4 columns as 2 pairs
  float v[6][6];   // V rows (slot = row mod 6; four are live)
  float y[3][6];   // Y rows (slot = row mod 3)
  float hp[3][6];  // dL/dY' rows (slot = row mod 3)
  float gsh[9];       // d/d sharpen
  r2l_p2 gay[2][9];   // [row parity][tap]: pair half = column parity
  r2l_p2 sy[2];
};

// raw row fetch / conversion of the streaming forward, for this kernel's argument struct
template <bool U16>
R2L_HD void r2l_bs_fetch_v(const R2LBwd2StreamArgs& a, size_t img0, int ym, int x0, bool le, bool re, int lane,
                           R2LFsStage& s) {
  R2LFwdStreamArgs f;
  f.raw = a.raw;
  f.W = a.W;
  r2l_fs_fetch<U16>(f, img0, ym, x0, le, re, lane, s);
}
template <bool U16>
R2L_HD void r2l_bs_convert_v(const R2LBwd2StreamArgs& a, R2LFoldedRef F, const R2LFsStage& s, bool le, bool re,
                             float v[6]) {
  R2LFwdStreamArgs f;
  f.raw = a.raw;
  r2l_fs_convert<U16>(f, F, s, le, re, v);
}

// Packed multiply-adds want their operands in even-aligned register pairs, and a value can sit in one pair only: the
// rows here are kept as pairs (0,1), (2,3), (4,5) of their 6 columns (own columns 1 .. 4), taps that read the pairs
// (1,2), (3,4) are written as two scalar multiply-adds instead of a packed one on copied registers.
R2L_HD void r2l_bs_stencil_parity(const float* r0, const float* r1, const float* r2, const float w[18] /* [9][2] */,
                                  r2l_p2 o[2]) {
  o[0] = o[1] = r2l_splat2(0.f);
  const float* rows[3] = {r0, r1, r2};
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i) {
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; j += 2) {
      const r2l_p2 wy = r2l_mk2(w[(i * 3 + j) * 2], w[(i * 3 + j) * 2 + 1]);
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) o[p] = r2l_pfma(wy, r2l_mk2(rows[i][2 * p + j], rows[i][2 * p + j + 1]), o[p]);
    }
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) {
      o[p][0] = fmaf(w[(i * 3 + 1) * 2], rows[i][2 * p + 1], o[p][0]);
      o[p][1] = fmaf(w[(i * 3 + 1) * 2 + 1], rows[i][2 * p + 2], o[p][1]);
    }
  }
}

// One step g (K = g mod 6).
//  (1) Row g of dL/dY'' (columns x0-2 .. x0+5, zero outside the image) is scattered into the partial rows g+2 .. g+DMIN
//      of dL/dY' = adjoint of the mirror-padded blur.  Target row t = g + d reads this row as its window row
//      k = 2 - d: HP(t)[c] += sum_j A_t[k][j] * w[c + 4 - j], A_t = the weight set of row t (R2LFolded::adj: rows 1, 2,
//      H-3, H-2 carry the rows the padding mirrored onto them).  The first / last lane of an image row also adds the
//      columns the padding mirrored onto columns 1, 2 / W-3, W-2.  No test of t against the image or the band: rows
//      outside are never read (their ring slots are restarted before they are), so the step has no branches around
//      the state.  DMIN > -2 in a band's first steps (rows above the band need no dL/dY').
//  (2) DOV: raw row g-1 enters the V ring.
//  (3) DOYHP: Y(g-2) from V rows g-3 .. g-1; HP(g-2) = the now complete partial row; strip edges of both exchanged.
//  (4) SUMS: row r = g-3: gY = adjoint sharpen of HP; the three families of sums.
template <int NW, bool U16, int K, int DMIN, bool DOV, bool DOYHP, bool SUMS>
R2L_HD void r2l_bs_step(const R2LBwd2StreamArgs& a, R2LBsState& st, const R2LBsGStage& sg, const R2LFsStage& sv, int g,
                        int y1, bool le, bool re, int wave, int lane, float* ex, bool store_ok) {
  const R2LFolded* Fl = r2l_opaque(a.F);
  R2LFoldedRef F = R2L_FOLDED_REF(Fl);
  const int H = a.H;
  float W[2][5];
  float wy[18];
  float ws[9];
  auto adj_of = [&](int d) -> r2l_cfp {
    const int t = g + d;
    const int set = (int)(t == 1) + 2 * (int)(t == 2) + 3 * (int)(t == H - 3) + 4 * (int)(t == H - 2);  // H >= 6
    return r2l_bs_launder((r2l_cfp)&F.adj[0][0] + 25 * set + 5 * (2 - d));
  };
  {
    r2l_cfp A = adj_of(2);
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 5; ++j) W[0][j] = A[j];
  }
  float w[8];
  r2l_bs_convert_g(sg, H, le, re, w);
  r2l_p2 P[4], O[3];
  R2L_PRAGMA_UNROLL
  for (int m = 0; m < 4; ++m) P[m] = r2l_mk2(w[2 * m], w[2 * m + 1]);
  R2L_PRAGMA_UNROLL
  for (int m = 0; m < 3; ++m) O[m] = r2l_straddle(P[m], P[m + 1]);
  const float fl = le ? 1.f : 0.f, fr = re ? 1.f : 0.f;
  constexpr int PY = K & 1;  // parity of g-2
  R2L_PRAGMA_UNROLL
  for (int d = 2; d >= DMIN; --d) {
    float* A = W[(2 - d) & 1];
    R2L_BS_PIN5(A);
    if (d > DMIN) {
      r2l_cfp An = adj_of(d - 1);
      R2L_PRAGMA_UNROLL
      for (int j = 0; j < 5; ++j) W[(3 - d) & 1][j] = An[j];
    } else if (DOYHP) {
      r2l_cfp An = r2l_bs_launder((r2l_cfp)&F.AY2[PY][0][0]);
      R2L_PRAGMA_UNROLL
      for (int j = 0; j < 18; ++j) wy[j] = An[j];
    }
    r2l_p2* pr = st.part[(K + d + 6) % 6];
    if (d == 2) pr[0] = pr[1] = r2l_splat2(0.f);  // row g+2 starts here
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 5; ++j) {  // output column c reads window column c + 4 - j: pair p starts at 2p + 4 - j
      const r2l_p2 wk = r2l_splat2(A[j]);
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) {
        const int s0 = 2 * p + 4 - j;  // 0 .. 6
        pr[p] = r2l_pfma(wk, (s0 & 1) ? O[s0 / 2] : P[s0 / 2], pr[p]);
      }
    }
    // folded columns (zero factors on every other lane): le: HP[1] += A0 w[3] + A1 w[2], HP[2] += A0 w[2];
    // re: HP[2] += A3 w[5] + A4 w[4], HP[1] += A4 w[5]
    const float c1 = fmaf(A[0], w[3], A[1] * w[2]), c2 = A[0] * w[2];
    const float d1 = fmaf(A[3], w[5], A[4] * w[4]), d2 = A[4] * w[5];
    pr[0][1] = fmaf(fl, c1, fmaf(fr, d2, pr[0][1]));
    pr[1][0] = fmaf(fl, c2, fmaf(fr, d1, pr[1][0]));
    R2L_BS_PINV2(pr[0], pr[1]);
  }
  if (DOV) r2l_bs_convert_v<U16>(a, F, sv, le, re, st.v[(K + 5) % 6]);
  if (!DOYHP) return;
  const int rr = g - 2;  // row of the new Y and HP
  const bool rin = (unsigned)rr < (unsigned)H;
  float* yn = st.y[(K + 1) % 3];
  float* hn = st.hp[(K + 1) % 3];
  // ---- Y(g-2) from V rows g-3 .. g-1 (zero outside the image: the sharpen's padding) ---------------------------
  {
    R2L_BS_PIN18(wy);
    if (SUMS) {
      r2l_cfp An = r2l_bs_launder((r2l_cfp)&F.sharp[0]);
      R2L_PRAGMA_UNROLL
      for (int j = 0; j < 9; ++j) ws[j] = An[j];
    }
    r2l_p2 o[2];
    r2l_bs_stencil_parity(st.v[(K + 3) % 6], st.v[(K + 4) % 6], st.v[(K + 5) % 6], wy, o);
    yn[1] = rin ? o[0][0] : 0.f;
    yn[2] = rin ? o[0][1] : 0.f;
    yn[3] = rin ? o[1][0] : 0.f;
    yn[4] = rin ? o[1][1] : 0.f;
  }
  // ---- HP(g-2): the partial row is complete (its last contribution, from row g, was scattered above) -----------
  {
    const r2l_p2* pr = st.part[(K + 4) % 6];
    const bool keep = rin && store_ok;  // (lanes beyond the image edge shadow the last column group: they add nothing)
    hn[1] = keep ? pr[0][0] : 0.f;
    hn[2] = keep ? pr[0][1] : 0.f;
    hn[3] = keep ? pr[1][0] : 0.f;
    hn[4] = keep ? pr[1][1] : 0.f;
  }
  // ---- strip edges of Y(g-2) and HP(g-2) ------------------------------------------------------------------------
  float rl_y = 0.f, rl_h = 0.f, rr_y = 0.f, rr_h = 0.f;
  if (NW > 1) {
    float* mine = ex + ((g & 1) * NW + wave) * R2L_BS_EX;
    if (lane == 0) {
      mine[0] = yn[1];
      mine[1] = hn[1];
    }
    if (lane == 63) {
      mine[2] = yn[4];
      mine[3] = hn[4];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (lane == 0 && wave > 0) {
      const float* o = mine - R2L_BS_EX;
      rl_y = o[2];
      rl_h = o[3];
    }
    if (lane == 63 && wave < NW - 1) {
      const float* o = mine + R2L_BS_EX;
      rr_y = o[0];
      rr_h = o[1];
    }
  }
  {
    const float yl = r2l_wshr(yn[4], rl_y), yr = r2l_wshl(yn[1], rr_y);
    const float hl = r2l_wshr(hn[4], rl_h), hr = r2l_wshl(hn[1], rr_h);
    yn[0] = le ? 0.f : yl;
    yn[5] = re ? 0.f : yr;
    hn[0] = le ? 0.f : hl;
    hn[5] = re ? 0.f : hr;
  }
  // ---- row r = g-3: gY = adjoint sharpen of HP; the three families of sums -------------------------------------
  const int r = g - 3;
  if (SUMS) {
    constexpr int PYR = (K + 1) & 1;
    const float* hu = st.hp[(K + 2) % 3];  // HP(r-1)
    const float* hm = st.hp[K % 3];        // HP(r)
    const float* hd = st.hp[(K + 1) % 3];  // HP(r+1)
    // gY(r)[c] = sum_{i,j} sharp[i][j] * HP(r - (i-1))[c - (j-1)]: row i = 0 is HP(r+1), source column c + 1 - j
    r2l_p2 gy[2];
    gy[0] = gy[1] = r2l_splat2(0.f);
    R2L_BS_PIN9(ws);
    {
      const float* rows[3] = {hd, hm, hu};
      R2L_PRAGMA_UNROLL
      for (int i = 0; i < 3; ++i) {
        R2L_PRAGMA_UNROLL
        for (int j = 0; j < 3; j += 2) {
          const r2l_p2 w2 = r2l_splat2(ws[i * 3 + j]);
          R2L_PRAGMA_UNROLL
          for (int p = 0; p < 2; ++p)
            gy[p] = r2l_pfma(w2, r2l_mk2(rows[i][2 * p + 2 - j], rows[i][2 * p + 3 - j]), gy[p]);
        }
        const float wc = ws[i * 3 + 1];
        R2L_PRAGMA_UNROLL
        for (int p = 0; p < 2; ++p) {
          gy[p][0] = fmaf(wc, rows[i][2 * p + 1], gy[p][0]);
          gy[p][1] = fmaf(wc, rows[i][2 * p + 2], gy[p][1]);
        }
      }
    }
    if (!(store_ok && r < y1)) gy[0] = gy[1] = r2l_splat2(0.f);  // (steps past the band's end: HP(r) is zero, gY is not)
    // d/d sharpening_filter.weight[i][j] += sum_c HP(r)[c] * Y(r + i - 1)[c + j - 1]
    {
      const float* rows[3] = {st.y[(K + 2) % 3], st.y[K % 3], st.y[(K + 1) % 3]};  // Y(r-1), Y(r), Y(r+1)
      R2L_PRAGMA_UNROLL
      for (int i = 0; i < 3; ++i)
        R2L_PRAGMA_UNROLL
      for (int j = 0; j < 3; ++j) {
        float s = st.gsh[i * 3 + j];
        R2L_PRAGMA_UNROLL
        for (int c = 1; c <= 4; ++c) s = fmaf(hm[c], rows[i][c + j - 1], s);
        st.gsh[i * 3 + j] = s;
      }
    }
    // folded luma stencil: GA_Y[parity(p)][t] += gY(p) * V_ext(p + t); pair half = column parity
    {
      const float* rows[3] = {st.v[(K + 2) % 6], st.v[(K + 3) % 6], st.v[(K + 4) % 6]};  // V(r-1), V(r), V(r+1)
      R2L_PRAGMA_UNROLL
      for (int i = 0; i < 3; ++i) {
        R2L_PRAGMA_UNROLL
        for (int j = 0; j < 3; j += 2) {
          r2l_p2 s = st.gay[PYR][i * 3 + j];
          R2L_PRAGMA_UNROLL
          for (int p = 0; p < 2; ++p) s = r2l_pfma(gy[p], r2l_mk2(rows[i][2 * p + j], rows[i][2 * p + j + 1]), s);
          st.gay[PYR][i * 3 + j] = s;
        }
        r2l_p2 s = st.gay[PYR][i * 3 + 1];
        R2L_PRAGMA_UNROLL
        for (int p = 0; p < 2; ++p) {
          s[0] = fmaf(gy[p][0], rows[i][2 * p + 1], s[0]);
          s[1] = fmaf(gy[p][1], rows[i][2 * p + 2], s[1]);
        }
        st.gay[PYR][i * 3 + 1] = s;
      }
      st.sy[PYR] = r2l_padd(st.sy[PYR], r2l_padd(gy[0], gy[1]));
    }
  }
}

template <int NW, bool U16>
R2L_BLOCKFN void r2l_bwd2_stream_block(const R2LBwd2StreamArgs& a, int bid, int nblk, float* lds) {
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float* ex = lds;
  float* red = lds + 2 * NW * R2L_BS_EX + 16;
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  const int xs = wave * 256 + 4 * lane;
  const bool store_ok = xs < a.W;
  const int x0 = store_ok ? xs : a.W - 4;
  const bool le = x0 == 0, re = x0 + 4 >= a.W;
  const size_t plane = (size_t)a.H * a.W;
  R2LBsState st;
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 9; ++i) {
    st.gsh[i] = 0.f;
    st.gay[0][i] = st.gay[1][i] = r2l_splat2(0.f);
  }
  st.sy[0] = st.sy[1] = r2l_splat2(0.f);
  constexpr int PF = R2L_BS_PF;
  static_assert(6 % PF == 0, "the prefetch rings are indexed by the unroll position");
  for (int item = bid; item < a.nitems; item += nblk) {
    const int band = item % a.nband, b = item / a.nband;
    const int y0 = band * a.band_h;
    const int y1 = (y0 + a.band_h < a.H) ? y0 + a.band_h : a.H;
    const size_t img = (size_t)b * plane;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 6; ++j) st.y[i][j] = st.hp[i][j] = 0.f;
    // Steps g = y0-3 .. y1+2 (rounded up to whole groups of six).  band_h is a multiple of 6, so g mod 6 is known at
    // compile time in every step: K = (g - y0 + 3) mod 6 + ... = 3, 4, 5, 0, 1, 2, 3, ...  The first six steps are a
    // static prologue: scatter only (into the rows the band needs), raw rows from g = y0-1 on, Y and HP from y0+1 on;
    // the main loop's steps carry no conditions on the state.
    R2LBsGStage pg[PF];  // step K consumes pg[K % PF] (dL/dY'' row g) and pv[K % PF] (raw row g - 1)
    R2LFsStage pv[PF];
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < PF; ++i) {
      r2l_bs_fetch_g(a, img, y0 - 3 + i, x0, le, re, lane, pg[(3 + i) % PF]);
      pv[(3 + i) % PF] = R2LFsStage{};
    }
    const int hm1 = a.H - 1;
#define R2L_BS_STEP(K, G, DMIN, DOV, DOYHP, SUMS)                                                        \
  {                                                                                                      \
    const int g = (G);                                                                                   \
    const R2LBsGStage sg = pg[K % PF];                                                                   \
    const R2LFsStage sv = pv[K % PF];                                                                    \
    r2l_bs_fetch_g(a, img, g + PF, x0, le, re, lane, pg[K % PF]);                                        \
    {                                                                                                    \
      int vr = g - 1 + PF; /* mirrored; rows past the band's last group only have to be readable */     \
      vr = vr < 0 ? -vr : vr;                                                                            \
      vr = vr > hm1 ? 2 * hm1 - vr : vr;                                                                 \
      r2l_bs_fetch_v<U16>(a, img, vr < 0 ? -vr : vr, x0, le, re, lane, pv[K % PF]);                      \
    }                                                                                                    \
    r2l_bs_step<NW, U16, K, DMIN, DOV, DOYHP, SUMS>(a, st, sg, sv, g, y1, le, re, wave, lane, ex,        \
                                                    store_ok);                                           \
  }
    R2L_BS_STEP(3, y0 - 3, 2, false, false, false)
    R2L_BS_STEP(4, y0 - 2, 1, false, false, false)
    R2L_BS_STEP(5, y0 - 1, 0, true, false, false)
    R2L_BS_STEP(0, y0, -1, true, false, false)
    R2L_BS_STEP(1, y0 + 1, -2, true, true, false)
    R2L_BS_STEP(2, y0 + 2, -2, true, true, false)
    for (int gb = y0 + 3; gb < y1 + 3; gb += 6) {
      R2L_BS_STEP(3, gb, -2, true, true, true)
      R2L_BS_STEP(4, gb + 1, -2, true, true, true)
      R2L_BS_STEP(5, gb + 2, -2, true, true, true)
      R2L_BS_STEP(0, gb + 3, -2, true, true, true)
      R2L_BS_STEP(1, gb + 4, -2, true, true, true)
      R2L_BS_STEP(2, gb + 5, -2, true, true, true)
    }
#undef R2L_BS_STEP
    if (NW > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  // ---- lanes -> one partial per global slot (layout R2L_B2_*) and workgroup ------------------------------------
  {
    float* wsum = red;  // [NW][64] (49 used)
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < R2L_B2_NACC; ++i) {
      float v;
      if (i < R2L_B2_GAY) {
        v = st.gsh[i];
      } else if (i < R2L_B2_SY) {
        const int par = (i - R2L_B2_GAY) / 9, t = (i - R2L_B2_GAY) % 9;
        v = st.gay[par >> 1][t][par & 1];
      } else {
        const int par = i - R2L_B2_SY;
        v = st.sy[par >> 1][par & 1];
      }
      R2L_PRAGMA_UNROLL
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane == 0) wsum[wave * 64 + i] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = tid; i < R2L_B2_NACC; i += NT) {
      float acc = 0.f;
      for (int w = 0; w < NW; ++w) acc += wsum[w * 64 + i];
      r2l_store_coherent(&a.partial[(size_t)i * nblk + bid], acc);
    }
    R2L_STORES_DONE();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (a.tree.counters) {
    float* base = lds + 2 * NW * R2L_BS_EX + 16;
    double* sums = (double*)(base + R2L_BS_RED_FLOATS(NW));
    double* tg = sums + R2L_NSUMS;
    float* pl = (float*)(tg + R2L_UNFOLD_TG);
    static_assert(R2L_BS_RED_FLOATS(NW) >= 512 + R2L_NSUMS * R2L_TREE_GROUP, "tree scratch");
    const bool last_ = r2l_tree_finish<R2L_NSUMS, NT>(a.tree, bid, nblk, base, sums, (double*)(base + 512),
                                                      (R2L_BS_RED_FLOATS(NW) - 512) / 2);
    if (!last_) return;
    r2l_unfold_phases<NT>(a.params, sums, tg, pl, a.grad_params);
  }
}

#endif  // !R2L_EMUL
