// r2l_param_kernels.h -- fused parametrized ISP (torch semantics), forward and backward.
//
// Replaces ParametrizedProcessing.forward (processing/pipeline_torch.py:175-225) and the autograd
// graph behind it.  One workgroup (256 threads = 4 wavefronts) owns a TW x TH output tile and a
// FRAME of (TW+8) x (TH+8) positions around it (halo 4 = 1 debayer + 1 sharpen + 2 blur).  Three
// float planes of the frame live in LDS:
//     V   black-level-corrected raw value, mirror-extended outside the image      (:183, :233)
//     Y   luma after debayer/WB/CCM/RGB->YUV, ZERO outside the image                (:187-195 padding=1)
//     YP  sharpened luma, mirror-extended outside the image                         (:195, :165 reflect)
// Only the luma plane carries a halo, so the recomputed-halo cost is 2 x 9 FMA per frame pixel; the
// heavy per-pixel work (5x5 blur, chroma, YUV->RGB, clip, gamma, BatchNorm) runs once per pixel on a
// 4x4 register micro-tile per thread, reading LDS with 128-bit accesses.
#pragma once
#include "r2l_common.h"

template <int TW_, int TH_>
struct R2LGeom {
  static constexpr int TW = TW_, TH = TH_;
  static constexpr int FW = TW + 8, FH = TH + 8;
  // row stride in floats: room for the +2-shifted planes, and a multiple of 16 so that the 128-bit
  // LDS reads of a 16x4-thread wavefront (rows 4 apart) fall on distinct banks
  static constexpr int FS = ((FW + 2 + 15) / 16) * 16;
  static constexpr int PLANE = FH * FS;
  static constexpr int PAD = 16;  // leading floats so that index -1 of plane 0 stays inside LDS
  static constexpr int TXN = TW / 4, TYN = TH / 4;  // micro-tiles per tile row / column
  static_assert(TXN * TYN == R2L_NT, "one 4x4 micro-tile per thread");
};

template <class G, int NPLANES>
constexpr size_t r2l_lds_bytes() {
  return sizeof(float) * (size_t)(2 * G::PAD + NPLANES * G::PLANE);
}

struct R2LTile {
  int b, oy, ox;  // image index, tile origin (global coordinates of tile pixel (0,0))
  bool border;    // the frame leaves the image somewhere
};

// XCD-aware tile walk: workgroup ids are dealt round-robin over the 8 XCDs (each with a private L2),
// so ids with equal (bid % 8) share an L2.  Each such group walks its own contiguous 1/8 of the tile
// list, which keeps halo rows/columns shared by neighbouring tiles inside one L2.
struct R2LTileWalk {
  int ntx, nty, ntiles, nper, group, j, jstep;
};
R2L_HD R2LTileWalk r2l_walk_init(int B, int H, int W, int TW, int TH, int bid, int nblk) {
  R2LTileWalk w;
  w.ntx = (W + TW - 1) / TW;
  w.nty = (H + TH - 1) / TH;
  w.ntiles = B * w.ntx * w.nty;
  const int ngroups = (nblk % 8 == 0) ? 8 : 1;
  w.nper = (w.ntiles + ngroups - 1) / ngroups;
  w.group = bid % ngroups;
  w.j = bid / ngroups;
  w.jstep = nblk / ngroups;
  return w;
}
R2L_HD bool r2l_walk_next(R2LTileWalk& w, int H, int W, int TW, int TH, R2LTile& t) {
  while (w.j < w.nper) {
    const int tile = w.group * w.nper + w.j;
    w.j += w.jstep;
    if (tile >= w.ntiles) return false;
    const int tx = tile % w.ntx, r = tile / w.ntx;
    t.b = r / w.nty;
    t.oy = (r % w.nty) * TH;
    t.ox = tx * TW;
    t.border = (t.oy < 4) || (t.ox < 4) || (t.oy + TH + 4 > H) || (t.ox + TW + 4 > W);
    return true;
  }
  return false;
}

// ---- phase A: raw tile + halo -> V -------------------------------------------------------------
template <class G>
R2L_HD void r2l_load_v(int tid, float* V, const float* rawb, R2LFoldedRef F, int oy, int ox, int H,
                       int W) {
  constexpr int CPR = G::FW / 4;
  const bool vec_ok = (W & 3) == 0;
  for (int ci = tid; ci < CPR * G::FH; ci += R2L_NT) {
    const int fy = ci / CPR, cx = ci - fy * CPR;
    const int gy = r2l_mirror(oy - 4 + fy, H);
    const int gx0 = ox - 4 + 4 * cx;
    const float* row = rawb + (size_t)gy * W;
    r2l_f4 v;
    if (vec_ok && gx0 >= 0 && gx0 + 3 < W) {
      v = *(const r2l_f4*)(row + gx0);
    } else {
      v.x = row[r2l_mirror(gx0, W)];
      v.y = row[r2l_mirror(gx0 + 1, W)];
      v.z = row[r2l_mirror(gx0 + 2, W)];
      v.w = row[r2l_mirror(gx0 + 3, W)];
    }
    // mirror padding keeps the Bayer parity, so the site follows from the frame coordinates
    const float b0 = F.bl[(fy & 1) * 2], b1 = F.bl[(fy & 1) * 2 + 1];
    v.x -= b0;
    v.y -= b1;
    v.z -= b0;
    v.w -= b1;
    *(r2l_f4*)(V + fy * G::FS + 4 * cx) = v;
  }
}

// a plane of the frame from a (B,H,W) global plane, ZERO outside the image, stored shifted by +2
template <class G>
R2L_HD void r2l_load_plane_zero_s2(int tid, float* Pl, const float* gb, int oy, int ox, int H, int W) {
  constexpr int CPR = G::FW / 4;
  const bool vec_ok = (W & 3) == 0;
  for (int ci = tid; ci < CPR * G::FH; ci += R2L_NT) {
    const int fy = ci / CPR, cx = ci - fy * CPR;
    const int gy = oy - 4 + fy;
    const int gx0 = ox - 4 + 4 * cx;
    r2l_f4 v;
    v.x = v.y = v.z = v.w = 0.f;
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
    float* d = Pl + fy * G::FS + 4 * cx + 2;
    r2l_f2 lo, hi;
    lo.x = v.x;
    lo.y = v.y;
    hi.x = v.z;
    hi.y = v.w;
    *(r2l_f2*)d = lo;
    *(r2l_f2*)(d + 2) = hi;
  }
}

// 4 rows x 6 columns window around a 4-wide x 2-tall item at (fy, fx): rows fy-1..fy+2,
// columns fx-1..fx+4 of an UNSHIFTED plane
template <class G>
R2L_HD void r2l_window_4x6(const float* Pl, int fy, int fx, float w[4][6]) {
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 4; ++i) {
    const float* r = Pl + (fy - 1 + i) * G::FS + fx;
    w[i][0] = r[-1];
    const r2l_f4 m = *(const r2l_f4*)r;
    w[i][1] = m.x;
    w[i][2] = m.y;
    w[i][3] = m.z;
    w[i][4] = m.w;
    w[i][5] = r[4];
  }
}

// ---- phase B: Y on frame rows/cols [1, F-1) ------------------------------------------------------
template <class G>
R2L_HD void r2l_compute_y(int tid, const float* V, float* Y, R2LFoldedRef F, int oy, int ox, int H,
                          int W) {
  constexpr int CPR = G::FW / 4, NRP = (G::FH - 2) / 2;
  for (int it = tid; it < CPR * NRP; it += R2L_NT) {
    const int rp = it / CPR, cx = it - rp * CPR;
    const int fy = 1 + 2 * rp, fx = 4 * cx;  // fy is odd
    float w[4][6];
    r2l_window_4x6<G>(V, fy, fx, w);
    R2L_PRAGMA_UNROLL
    for (int r = 0; r < 2; ++r) {
      const int gy = oy - 4 + fy + r;
      const bool yin = (unsigned)gy < (unsigned)H;
      float o[4];
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 4; ++c) {
        const int par = (((1 + r) & 1) << 1) | (c & 1);
        float s = 0.f;
        R2L_PRAGMA_UNROLL
        for (int i = 0; i < 3; ++i)
          R2L_PRAGMA_UNROLL
        for (int j = 0; j < 3; ++j) s = fmaf(F.AY[par][i * 3 + j], w[r + i][c + j], s);
        const int gx = ox - 4 + fx + c;
        o[c] = (yin && (unsigned)gx < (unsigned)W) ? s : 0.f;  // zero padding of the sharpen conv
      }
      r2l_f4 st;
      st.x = o[0];
      st.y = o[1];
      st.z = o[2];
      st.w = o[3];
      *(r2l_f4*)(Y + (fy + r) * G::FS + fx) = st;
    }
  }
}

// ---- phase C: YP = sharpen(Y) on frame rows/cols [2, F-2), stored shifted by +2 columns ----------
template <class G>
R2L_HD void r2l_compute_yp(int tid, const float* Y, float* YP, R2LFoldedRef F) {
  constexpr int CPR = G::FW / 4, NRP = (G::FH - 4) / 2;
  for (int it = tid; it < CPR * NRP; it += R2L_NT) {
    const int rp = it / CPR, cx = it - rp * CPR;
    const int fy = 2 + 2 * rp, fx = 4 * cx;
    float w[4][6];
    r2l_window_4x6<G>(Y, fy, fx, w);
    R2L_PRAGMA_UNROLL
    for (int r = 0; r < 2; ++r) {
      float o[4];
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 4; ++c) {
        float s = 0.f;
        R2L_PRAGMA_UNROLL
        for (int i = 0; i < 3; ++i)
          R2L_PRAGMA_UNROLL
        for (int j = 0; j < 3; ++j) s = fmaf(F.sharp[i * 3 + j], w[r + i][c + j], s);
        o[c] = s;
      }
      float* d = YP + (fy + r) * G::FS + fx + 2;
      r2l_f2 lo, hi;
      lo.x = o[0];
      lo.y = o[1];
      hi.x = o[2];
      hi.y = o[3];
      *(r2l_f2*)d = lo;
      *(r2l_f2*)(d + 2) = hi;
    }
  }
}

// ---- phase C2 (border tiles): mirror-extend YP outside the image (padding_mode='reflect', :165) ---
template <class G>
R2L_HD void r2l_fill_yp_mirror(int tid, float* YP, int oy, int ox, int H, int W) {
  constexpr int NW = G::FW - 4, NH = G::FH - 4;
  for (int i = tid; i < NW * NH; i += R2L_NT) {
    const int fy = 2 + i / NW, fx = 2 + i % NW;
    const int gy = oy - 4 + fy, gx = ox - 4 + fx;
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) continue;
    const int my = r2l_mirror(gy, H) - (oy - 4), mx = r2l_mirror(gx, W) - (ox - 4);
    if (my >= 2 && my < G::FH - 2 && mx >= 2 && mx < G::FW - 2)
      YP[fy * G::FS + fx + 2] = YP[my * G::FS + mx + 2];
  }
}

// ---- phase D helpers: one 4x4 micro-tile per thread ----------------------------------------------
// 8x8 window of YP (rows fy0-2..fy0+5, cols fx0-2..fx0+5; fx0 = 4*tx+4, plane shifted by +2)
template <class G>
R2L_HD void r2l_window_yp(const float* YP, int tx, int ty, float yw[8][8]) {
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 8; ++i) {
    const float* r = YP + (4 * ty + 2 + i) * G::FS + 4 * tx + 4;
    const r2l_f4 a = *(const r2l_f4*)r;
    const r2l_f4 b = *(const r2l_f4*)(r + 4);
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
// 6x6 window of an unshifted plane (rows fy0-1..fy0+4, cols fx0-1..fx0+4)
template <class G>
R2L_HD void r2l_window_6x6(const float* Pl, int tx, int ty, float w[6][6]) {
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 6; ++i) {
    const float* r = Pl + (4 * ty + 3 + i) * G::FS + 4 * tx + 4;
    w[i][0] = r[-1];
    const r2l_f4 m = *(const r2l_f4*)r;
    w[i][1] = m.x;
    w[i][2] = m.y;
    w[i][3] = m.z;
    w[i][4] = m.w;
    w[i][5] = r[4];
  }
}

R2L_HD void r2l_blur_4x4(const float yw[8][8], R2LFoldedRef F, float ypp[4][4]) {
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 4; ++r)
    R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) {
    float s = 0.f;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 5; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 5; ++j) s = fmaf(F.blur[i * 5 + j], yw[r + i][c + j], s);
    ypp[r][c] = s;
  }
}

R2L_HD void r2l_chroma_4x4(const float vw[6][6], R2LFoldedRef F, float u[4][4], float v[4][4]) {
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 4; ++r)
    R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) {
    const int par = ((r & 1) << 1) | (c & 1);
    float su = 0.f, sv = 0.f;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      su = fmaf(F.AU[par][i * 3 + j], vw[r + i][c + j], su);
      sv = fmaf(F.AV[par][i * 3 + j], vw[r + i][c + j], sv);
    }
    u[r][c] = su;
    v[r][c] = sv;
  }
}

// ================================================================================================
// forward
// ================================================================================================
struct R2LFwdArgs {
  const float* raw;       // (B,H,W)
  const float* additive;  // (3,256,256) or null
  const R2LFolded* F;
  const float* bn;      // mean[3], istd[3] or null
  float* out;           // (B,3,H,W) or null (stats only)
  float* stat_partial;  // [6][nblk] or null
  int B, H, W;
};

struct R2LFwdRegs {
  float acc[6];
};

template <class G>
R2L_HD void r2l_fwd_pixels(int tid, const float* V, const float* YP, const R2LFwdArgs& a,
                           R2LFoldedRef F, const R2LTile& t, R2LFwdRegs& regs) {
  const int tx = tid % G::TXN, ty = tid / G::TXN;
  const int gy0 = t.oy + 4 * ty, gx0 = t.ox + 4 * tx;
  if (gy0 >= a.H || gx0 >= a.W) return;  // micro-tile entirely outside a ragged image edge
  float ypp[4][4], u[4][4], v[4][4];
  {
    float yw[8][8];
    r2l_window_yp<G>(YP, tx, ty, yw);
    r2l_blur_4x4(yw, F, ypp);
  }
  {
    float vw[6][6];
    r2l_window_6x6<G>(V, tx, ty, vw);
    r2l_chroma_4x4(vw, F, u, v);
  }
  const size_t plane = (size_t)a.H * a.W;
  const bool vec_ok = ((a.W & 3) == 0) && (gx0 + 3 < a.W);
  float mean[3] = {0.f, 0.f, 0.f}, istd[3] = {1.f, 1.f, 1.f};
  if (a.bn) {
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      mean[k] = a.bn[k];
      istd[k] = a.bn[3 + k];
    }
  }
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 4; ++r) {
    const int gy = gy0 + r;
    if (gy >= a.H) break;
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      float x[4];
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 4; ++c) {
        float rgb = F.M2[k * 3] * ypp[r][c];
        rgb = fmaf(F.M2[k * 3 + 1], u[r][c], rgb);
        rgb = fmaf(F.M2[k * 3 + 2], v[r][c], rgb);
        const float xc = fminf(fmaxf(rgb, 1e-5f), 1.0f);          // :206
        x[c] = r2l_exp2(r2l_log2(xc) * F.inv_gamma);              // :209
      }
      if (a.additive) {                                          // :213 (H == W == 256)
        const float* ad = a.additive + ((size_t)k * a.H + gy) * a.W + gx0;
        R2L_PRAGMA_UNROLL
        for (int c = 0; c < 4; ++c)
          if (gx0 + c < a.W) x[c] += ad[c];
      }
      if (a.stat_partial) {
        R2L_PRAGMA_UNROLL
        for (int c = 0; c < 4; ++c)
          if (gx0 + c < a.W) {
            const float d = x[c] - 0.5f;
            regs.acc[k] += d;
            regs.acc[3 + k] = fmaf(d, d, regs.acc[3 + k]);
          }
      }
      if (a.out) {
        R2L_PRAGMA_UNROLL
        for (int c = 0; c < 4; ++c) x[c] = (x[c] - mean[k]) * istd[k];  // :217
        float* o = a.out + ((size_t)t.b * 3 + k) * plane + (size_t)gy * a.W + gx0;
        if (vec_ok) {
          r2l_f4 st;
          st.x = x[0];
          st.y = x[1];
          st.z = x[2];
          st.w = x[3];
          *(r2l_f4*)o = st;
        } else {
          R2L_PRAGMA_UNROLL
          for (int c = 0; c < 4; ++c)
            if (gx0 + c < a.W) o[c] = x[c];
        }
      }
    }
  }
}

// per-thread accumulators -> one partial per slot per workgroup, in a fixed order (bitwise
// reproducible): slots go through LDS 32 at a time, thread s < 32 adds the 256 values of slot s.
#define R2L_RED_FLOATS (32 * 257)
#define R2L_BLOCK_REDUCE(NACC, regs, lds, partial, bid, nblk)                               \
  R2L_PRAGMA_UNROLL                                                                         \
  for (int base_ = 0; base_ < (NACC); base_ += 32) {                                        \
    R2L_PHASE_BEGIN                                                                         \
    R2L_PRAGMA_UNROLL                                                                       \
    for (int i_ = 0; i_ < 32; ++i_)                                                         \
      if (base_ + i_ < (NACC)) (lds)[i_ * 257 + tid] = R2L_TREG(regs).acc[base_ + i_];      \
    R2L_PHASE_END                                                                           \
    R2L_PHASE_BEGIN                                                                         \
    if (tid < 32 && base_ + tid < (NACC)) {                                                 \
      float s_ = 0.f;                                                                       \
      for (int j_ = 0; j_ < R2L_NT; ++j_) s_ += (lds)[tid * 257 + j_];                      \
      (partial)[(size_t)(base_ + tid) * (nblk) + (bid)] = s_;                               \
    }                                                                                       \
    R2L_PHASE_END                                                                           \
  }

template <class G>
R2L_BLOCKFN void r2l_fwd_block(const R2LFwdArgs& a, int bid, int nblk, float* lds) {
  float* V = lds + G::PAD;
  float* Y = V + G::PLANE;
  float* YP = Y + G::PLANE;
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  R2L_TREG_DECL(R2LFwdRegs, regs);
  R2L_PHASE_BEGIN
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 6; ++i) R2L_TREG(regs).acc[i] = 0.f;
  R2L_PHASE_END
  R2LTileWalk w = r2l_walk_init(a.B, a.H, a.W, G::TW, G::TH, bid, nblk);
  R2LTile t;
  while (r2l_walk_next(w, a.H, a.W, G::TW, G::TH, t)) {
    const float* rawb = a.raw + (size_t)t.b * a.H * a.W;
    R2L_PHASE_BEGIN
    r2l_load_v<G>(tid, V, rawb, F, t.oy, t.ox, a.H, a.W);
    R2L_PHASE_END
    R2L_PHASE_BEGIN
    r2l_compute_y<G>(tid, V, Y, F, t.oy, t.ox, a.H, a.W);
    R2L_PHASE_END
    R2L_PHASE_BEGIN
    r2l_compute_yp<G>(tid, Y, YP, F);
    R2L_PHASE_END
    if (t.border) {
      R2L_PHASE_BEGIN
      r2l_fill_yp_mirror<G>(tid, YP, t.oy, t.ox, a.H, a.W);
      R2L_PHASE_END
    }
    R2L_PHASE_BEGIN
    r2l_fwd_pixels<G>(tid, V, YP, a, F, t, R2L_TREG(regs));
    R2L_PHASE_END
  }
  if (a.stat_partial) {
    R2L_BLOCK_REDUCE(6, regs, lds, a.stat_partial, bid, nblk)
  }
}

// ================================================================================================
// backward, kernel B1: everything that is pointwise in the pixel + the blur-weight / chroma sums
// ================================================================================================
struct R2LBwd1Args {
  const float* raw;
  const float* additive;
  const R2LFolded* F;
  const float* bn;      // mean[3], istd[3] or null
  const float* bn_bwd;  // mean_g[3], mean_gxhat[3] or null
  const float* gout;    // (B,3,H,W)
  float* gypp;          // (B,H,W): d loss / d Y'' (blurred luma)
  float* partial;       // [R2L_B1_NACC][nblk]
  int B, H, W;
};

struct R2LBwd1Regs {
  float acc[R2L_B1_NACC];
};

template <class G>
R2L_HD void r2l_bwd1_pixels(int tid, const float* V, const float* YP, const R2LBwd1Args& a,
                            R2LFoldedRef F, const R2LTile& t, R2LBwd1Regs& regs) {
  const int tx = tid % G::TXN, ty = tid / G::TXN;
  const int gy0 = t.oy + 4 * ty, gx0 = t.ox + 4 * tx;
  if (gy0 >= a.H || gx0 >= a.W) return;
  float yw[8][8], vw[6][6];
  float ypp[4][4], u[4][4], v[4][4];
  r2l_window_yp<G>(YP, tx, ty, yw);
  r2l_blur_4x4(yw, F, ypp);
  r2l_window_6x6<G>(V, tx, ty, vw);
  r2l_chroma_4x4(vw, F, u, v);
  const size_t plane = (size_t)a.H * a.W;
  const bool vec_ok = ((a.W & 3) == 0) && (gx0 + 3 < a.W);
  float mean[3] = {0.f, 0.f, 0.f}, istd[3] = {1.f, 1.f, 1.f}, mg[3] = {0.f, 0.f, 0.f},
        mgx[3] = {0.f, 0.f, 0.f};
  if (a.bn) {
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      mean[k] = a.bn[k];
      istd[k] = a.bn[3 + k];
    }
  }
  if (a.bn_bwd) {
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      mg[k] = a.bn_bwd[k];
      mgx[k] = a.bn_bwd[3 + k];
    }
  }
  float gy2[4][4], gu[4][4], gv[4][4];  // d loss / d (Y'', U, V)
  float ggam = 0.f;
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 4; ++r) {
    const int gy = gy0 + r;
    const bool rowin = gy < a.H;
    float grgb[3][4];
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      float g[4] = {0.f, 0.f, 0.f, 0.f};
      if (rowin) {
        const float* gp = a.gout + ((size_t)t.b * 3 + k) * plane + (size_t)gy * a.W + gx0;
        if (vec_ok) {
          const r2l_f4 q = *(const r2l_f4*)gp;
          g[0] = q.x;
          g[1] = q.y;
          g[2] = q.z;
          g[3] = q.w;
        } else {
          R2L_PRAGMA_UNROLL
          for (int c = 0; c < 4; ++c)
            if (gx0 + c < a.W) g[c] = gp[c];
        }
      }
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 4; ++c) {
        const bool valid = rowin && (gx0 + c < a.W);
        float rgb = F.M2[k * 3] * ypp[r][c];
        rgb = fmaf(F.M2[k * 3 + 1], u[r][c], rgb);
        rgb = fmaf(F.M2[k * 3 + 2], v[r][c], rgb);
        const float xc = fminf(fmaxf(rgb, 1e-5f), 1.0f);
        const float lg = r2l_log2(xc);
        const float og = r2l_exp2(lg * F.inv_gamma);
        float gx = g[c];
        if (a.bn) {
          float x = og;
          if (a.additive && valid) x += a.additive[((size_t)k * a.H + gy) * a.W + gx0 + c];
          const float xhat = (x - mean[k]) * istd[k];
          // BatchNorm2d backward, train mode: istd * (g - mean(g) - xhat * mean(g*xhat)); in eval
          // mode mg = mgx = 0 and only the scaling remains
          gx = istd[k] * (gx - mg[k] - xhat * mgx[k]);
        }
        gx = valid ? gx : 0.f;
        ggam = fmaf(gx * og, lg, ggam);
        const float gc = gx * og * F.inv_gamma * r2l_rcp(xc);
        grgb[k][c] = (rgb >= 1e-5f && rgb <= 1.0f) ? gc : 0.f;  // torch.clip backward
      }
    }
    R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c) {
      gy2[r][c] = F.M2[0] * grgb[0][c] + F.M2[3] * grgb[1][c] + F.M2[6] * grgb[2][c];
      gu[r][c] = F.M2[1] * grgb[0][c] + F.M2[4] * grgb[1][c] + F.M2[7] * grgb[2][c];
      gv[r][c] = F.M2[2] * grgb[0][c] + F.M2[5] * grgb[1][c] + F.M2[8] * grgb[2][c];
    }
    if (rowin) {
      float* o = a.gypp + (size_t)t.b * plane + (size_t)gy * a.W + gx0;
      if (vec_ok) {
        r2l_f4 st;
        st.x = gy2[r][0];
        st.y = gy2[r][1];
        st.z = gy2[r][2];
        st.w = gy2[r][3];
        *(r2l_f4*)o = st;
      } else {
        R2L_PRAGMA_UNROLL
        for (int c = 0; c < 4; ++c)
          if (gx0 + c < a.W) o[c] = gy2[r][c];
      }
    }
  }
  regs.acc[R2L_B1_GGAM] += ggam;
  // d/d gaussian_blur.weight[i][j] = sum_p gY''(p) * YP_ext(p + (i-2, j-2))
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 5; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 5; ++j) {
    float s = 0.f;
    R2L_PRAGMA_UNROLL
    for (int r = 0; r < 4; ++r)
      R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c) s = fmaf(gy2[r][c], yw[r + i][c + j], s);
    regs.acc[R2L_B1_GBLUR + i * 5 + j] += s;
  }
  // folded chroma stencils: GA[par][t] = sum_{p of parity par} gU(p) * v_ext(p+t)
  R2L_PRAGMA_UNROLL
  for (int par = 0; par < 4; ++par) {
    const int r0 = par >> 1, c0 = par & 1;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      float su = 0.f, sv = 0.f;
      R2L_PRAGMA_UNROLL
      for (int r = r0; r < 4; r += 2)
        R2L_PRAGMA_UNROLL
      for (int c = c0; c < 4; c += 2) {
        su = fmaf(gu[r][c], vw[r + i][c + j], su);
        sv = fmaf(gv[r][c], vw[r + i][c + j], sv);
      }
      regs.acc[R2L_B1_GAU + par * 9 + i * 3 + j] += su;
      regs.acc[R2L_B1_GAV + par * 9 + i * 3 + j] += sv;
    }
    float tu = 0.f, tv = 0.f;
    R2L_PRAGMA_UNROLL
    for (int r = r0; r < 4; r += 2)
      R2L_PRAGMA_UNROLL
    for (int c = c0; c < 4; c += 2) {
      tu += gu[r][c];
      tv += gv[r][c];
    }
    regs.acc[R2L_B1_SU + par] += tu;
    regs.acc[R2L_B1_SV + par] += tv;
  }
}

template <class G>
R2L_BLOCKFN void r2l_bwd1_block(const R2LBwd1Args& a, int bid, int nblk, float* lds) {
  float* V = lds + G::PAD;
  float* Y = V + G::PLANE;
  float* YP = Y + G::PLANE;
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  R2L_TREG_DECL(R2LBwd1Regs, regs);
  R2L_PHASE_BEGIN
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < R2L_B1_NACC; ++i) R2L_TREG(regs).acc[i] = 0.f;
  R2L_PHASE_END
  R2LTileWalk w = r2l_walk_init(a.B, a.H, a.W, G::TW, G::TH, bid, nblk);
  R2LTile t;
  while (r2l_walk_next(w, a.H, a.W, G::TW, G::TH, t)) {
    const float* rawb = a.raw + (size_t)t.b * a.H * a.W;
    R2L_PHASE_BEGIN
    r2l_load_v<G>(tid, V, rawb, F, t.oy, t.ox, a.H, a.W);
    R2L_PHASE_END
    R2L_PHASE_BEGIN
    r2l_compute_y<G>(tid, V, Y, F, t.oy, t.ox, a.H, a.W);
    R2L_PHASE_END
    R2L_PHASE_BEGIN
    r2l_compute_yp<G>(tid, Y, YP, F);
    R2L_PHASE_END
    if (t.border) {
      R2L_PHASE_BEGIN
      r2l_fill_yp_mirror<G>(tid, YP, t.oy, t.ox, a.H, a.W);
      R2L_PHASE_END
    }
    R2L_PHASE_BEGIN
    r2l_bwd1_pixels<G>(tid, V, YP, a, F, t, R2L_TREG(regs));
    R2L_PHASE_END
  }
  R2L_BLOCK_REDUCE(R2L_B1_NACC, regs, lds, a.partial, bid, nblk)
}

// ================================================================================================
// backward, kernel B2: adjoint of blur (mirror pad) and sharpen (zero pad) on the luma plane
// ================================================================================================
struct R2LBwd2Args {
  const float* raw;
  const R2LFolded* F;
  const float* gypp;  // (B,H,W) from B1
  float* partial;     // [R2L_B2_NACC][nblk]
  int B, H, W;
};

struct R2LBwd2Regs {
  float acc[R2L_B2_NACC];
};

// phase: HP(q') = sum_t blur[t] * G2_ext0(q' - t) on frame rows/cols [2, F-2); G2 is stored shifted
template <class G>
R2L_HD void r2l_adjoint_blur(int tid, const float* G2, float* HP, R2LFoldedRef F) {
  constexpr int CPR = G::FW / 4, NRP = (G::FH - 4) / 2;
  for (int it = tid; it < CPR * NRP; it += R2L_NT) {
    const int rp = it / CPR, cx = it - rp * CPR;
    const int fy = 2 + 2 * rp, fx = 4 * cx;
    float w[6][8];  // rows fy-2..fy+3, cols fx-2..fx+5
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 6; ++i) {
      const float* r = G2 + (fy - 2 + i) * G::FS + fx;  // (+2 shift) - 2
      const r2l_f4 a = *(const r2l_f4*)r;
      const r2l_f4 b = *(const r2l_f4*)(r + 4);
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
    for (int r = 0; r < 2; ++r) {
      float o[4];
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 4; ++c) {
        float s = 0.f;
        R2L_PRAGMA_UNROLL
        for (int i = 0; i < 5; ++i)
          R2L_PRAGMA_UNROLL
        for (int j = 0; j < 5; ++j)  // source p = q' - (i-2, j-2)  ->  window index (r+4-i, c+4-j)
          s = fmaf(F.blur[i * 5 + j], w[r + 4 - i][c + 4 - j], s);
        o[c] = s;
      }
      r2l_f4 st;
      st.x = o[0];
      st.y = o[1];
      st.z = o[2];
      st.w = o[3];
      *(r2l_f4*)(HP + (fy + r) * G::FS + fx) = st;
    }
  }
}

// phase (border tiles): fold the contributions that the mirror padding sent outside the image back
// onto their sources, and zero everything outside the image:  H2 <- fold(HP)
template <class G>
R2L_HD void r2l_fold_mirror(int tid, const float* HP, float* H2, int oy, int ox, int H, int W) {
  constexpr int NW = G::FW - 4, NH = G::FH - 4;
  for (int i = tid; i < NW * NH; i += R2L_NT) {
    const int fy = 2 + i / NW, fx = 2 + i % NW;
    const int gy = oy - 4 + fy, gx = ox - 4 + fx;
    float s = 0.f;
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
      int ey[3], ex[3];
      ey[0] = fy;
      ey[1] = (gy >= 1 && gy <= 2) ? fy - 2 * gy : -1;                    // image row -gy
      ey[2] = (gy >= H - 3 && gy <= H - 2) ? fy + 2 * (H - 1 - gy) : -1;  // image row 2(H-1)-gy
      ex[0] = fx;
      ex[1] = (gx >= 1 && gx <= 2) ? fx - 2 * gx : -1;
      ex[2] = (gx >= W - 3 && gx <= W - 2) ? fx + 2 * (W - 1 - gx) : -1;
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 3; ++p)
        R2L_PRAGMA_UNROLL
      for (int q = 0; q < 3; ++q)
        if (ey[p] >= 2 && ey[p] < G::FH - 2 && ex[q] >= 2 && ex[q] < G::FW - 2)
          s += HP[ey[p] * G::FS + ex[q]];
    }
    H2[fy * G::FS + fx] = s;
  }
}

template <class G>
R2L_HD void r2l_bwd2_pixels(int tid, const float* V, const float* Y, const float* HS,
                            const R2LBwd2Args& a, R2LFoldedRef F, const R2LTile& t,
                            R2LBwd2Regs& regs) {
  const int tx = tid % G::TXN, ty = tid / G::TXN;
  const int gy0 = t.oy + 4 * ty, gx0 = t.ox + 4 * tx;
  if (gy0 >= a.H || gx0 >= a.W) return;
  float hw[6][6], yw[6][6], vw[6][6];
  r2l_window_6x6<G>(HS, tx, ty, hw);
  r2l_window_6x6<G>(Y, tx, ty, yw);
  r2l_window_6x6<G>(V, tx, ty, vw);
  float gy1[4][4];  // d loss / d Y (pre-sharpen luma), interior pixels
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 4; ++r)
    R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) {
    const bool valid = (gy0 + r < a.H) && (gx0 + c < a.W);
    float s = 0.f;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j)  // source q = p - (i-1, j-1) -> window index (r+2-i, c+2-j)
      s = fmaf(F.sharp[i * 3 + j], hw[r + 2 - i][c + 2 - j], s);
    gy1[r][c] = valid ? s : 0.f;
    // outside the image gY' (hw centre) is already zero for border tiles; interior tiles are all valid
  }
  // d/d sharpening_filter.weight[i][j] = sum_p gY'(p) * Y_zero_ext(p + (i-1, j-1))
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 3; ++j) {
    float s = 0.f;
    R2L_PRAGMA_UNROLL
    for (int r = 0; r < 4; ++r)
      R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c) s = fmaf(hw[r + 1][c + 1], yw[r + i][c + j], s);
    regs.acc[R2L_B2_GSHARP + i * 3 + j] += s;
  }
  R2L_PRAGMA_UNROLL
  for (int par = 0; par < 4; ++par) {
    const int r0 = par >> 1, c0 = par & 1;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      float s = 0.f;
      R2L_PRAGMA_UNROLL
      for (int r = r0; r < 4; r += 2)
        R2L_PRAGMA_UNROLL
      for (int c = c0; c < 4; c += 2) s = fmaf(gy1[r][c], vw[r + i][c + j], s);
      regs.acc[R2L_B2_GAY + par * 9 + i * 3 + j] += s;
    }
    float ts = 0.f;
    R2L_PRAGMA_UNROLL
    for (int r = r0; r < 4; r += 2)
      R2L_PRAGMA_UNROLL
    for (int c = c0; c < 4; c += 2) ts += gy1[r][c];
    regs.acc[R2L_B2_SY + par] += ts;
  }
}

template <class G>
R2L_BLOCKFN void r2l_bwd2_block(const R2LBwd2Args& a, int bid, int nblk, float* lds) {
  float* V = lds + G::PAD;
  float* Y = V + G::PLANE;
  float* G2 = Y + G::PLANE;   // shifted; reused as H2 by border tiles
  float* HP = G2 + G::PLANE;
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  R2L_TREG_DECL(R2LBwd2Regs, regs);
  R2L_PHASE_BEGIN
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < R2L_B2_NACC; ++i) R2L_TREG(regs).acc[i] = 0.f;
  R2L_PHASE_END
  R2LTileWalk w = r2l_walk_init(a.B, a.H, a.W, G::TW, G::TH, bid, nblk);
  R2LTile t;
  while (r2l_walk_next(w, a.H, a.W, G::TW, G::TH, t)) {
    const size_t off = (size_t)t.b * a.H * a.W;
    R2L_PHASE_BEGIN
    r2l_load_v<G>(tid, V, a.raw + off, F, t.oy, t.ox, a.H, a.W);
    r2l_load_plane_zero_s2<G>(tid, G2, a.gypp + off, t.oy, t.ox, a.H, a.W);
    R2L_PHASE_END
    R2L_PHASE_BEGIN
    r2l_compute_y<G>(tid, V, Y, F, t.oy, t.ox, a.H, a.W);
    r2l_adjoint_blur<G>(tid, G2, HP, F);
    R2L_PHASE_END
    const float* HS = HP;
    if (t.border) {
      R2L_PHASE_BEGIN
      r2l_fold_mirror<G>(tid, HP, G2, t.oy, t.ox, a.H, a.W);
      R2L_PHASE_END
      HS = G2;
    }
    R2L_PHASE_BEGIN
    r2l_bwd2_pixels<G>(tid, V, Y, HS, a, F, t, R2L_TREG(regs));
    R2L_PHASE_END
  }
  R2L_BLOCK_REDUCE(R2L_B2_NACC, regs, lds, a.partial, bid, nblk)
}
