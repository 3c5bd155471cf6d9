// r2l_static_stream.h -- the SHORT static chain (BASELINE config C3) as a row-streaming kernel.
//
//   remove_blacklv -> demosaicing_CFA_Bayer_{bilinear,Malvar2004} -> white balance -> colour matrix ->
//   clip[0,1] -> x ** (1/gamma)        (processing(), pipeline_numpy.py:70-141 without sharpening/denoising)
//
// There is no stencil CHAIN here (halo 1 or 2 of one stage), so nothing needs LDS or a barrier: every
// wavefront is an independent line-buffer ISP.  A wavefront owns a strip of 256 columns (4 per lane) and a
// band of rows; it walks down the band keeping the last 3 (bilinear) or 5 (Malvar) raw rows in a rolling
// register window, so every raw row is fetched once per band (one fully coalesced 1-KiB read per wave)
// and each output row leaves as three 1-KiB stores.  Left/right neighbours come from two 8-byte loads that
// hit the L1 lines the neighbouring lanes just fetched.  Algorithmic traffic: 4 B in + 12 B out per pixel.
// Linear part in float64 like the reference (see r2l_static_kernels.h), log2/exp2 in float32.
#pragma once
#include "r2l_static_kernels.h"

struct R2LStaticStreamArgs {
  R2LStaticArgs s;
  int nseg, nband, band_h, nitems;
  // multi-pass chains (LUMA instantiations): the first pass stops at the luma plane Y = (yuv_from_rgb * CCM *
  // WB * demosaic)[0] in float64; filter passes (r2l_static_planes.h) rewrite it; the last pass recomputes the
  // chroma from the raw frame, takes the filtered luma from the plane and finishes (YUV->RGB, clip, gamma)
  double* luma_out;
  const double* luma_in;
  // fft_denoising works on the sharpened RGB image before clip and gamma: with lin_out the last pass stores that
  // image as float64 planes (B,3,H,W) instead of finishing
  double* lin_out;
};

// raw values of columns x0-2 .. x0+5 of source row `ys` (symmetric extension at the image edges); with 16-bit
// containers the undecoded bits (v[0] = left pair, v[1..2] = the 4 centre values, v[3] = right pair), decoded
// when the row enters the window so that the fetch stays a fire-and-forget load.  RAWK = container of the
// frames: R2L_RAW_F32 | R2L_RAW_U16 | R2L_RAW_F64 (float64 frames keep their float64 values in the stage)
template <int RAWK>
struct R2LRowStageT {
  float v[8];
  int ys;
};
template <>
struct R2LRowStageT<R2L_RAW_F64> {
  double v[8];
  int ys;
};
// LANES: the neighbour columns come from the neighbouring lanes of the wavefront (DPP wave shifts) when the row
// enters the window; only lane 0 / lane 63 fetch theirs from the neighbouring strip (one load instruction for
// both, staged in v[0..1] resp. v[0]).  Otherwise every lane loads its own neighbour pairs (hits in L1).  On the
// bilinear chain (1 neighbour each side) the lane form is 3 % faster, on Malvar (2 each side) 9 % slower.
// BRANCH-FREE form (float32 / 16-bit frames): every lane issues the same loads -- the 16 (8) bytes of its own columns and
// ONE pair beyond them from an in-row address (lanes 0-31: the pair on their left, lanes 32-63: on their right; at the
// image's edges the lane's own columns, whose mirror images convert_row takes from the centre values) [LANES], or both
// pairs [!LANES].  With a load under a lane-dependent or uniform condition anywhere in the row loop hipcc's s_waitcnt
// insertion gives up counting: the row step waited with vmcnt(0) .. vmcnt(3), i.e. for the row it had JUST requested and
// for the previous row's stores, instead of for the row requested PF steps ago (the lesson of r2l_fa_fetch_raw, round 3;
// profiles/r04_static_branch_free.txt).
// Addresses: a wave-uniform row pointer (scalar registers) + the lane's unsigned element offsets xo / xl / xr, constants of
// the work item -- the `global_load v, voffset, s[base]` form, one 32-bit register per stream instead of a 64-bit pointer
// per load.
template <int RAWK, bool LANES>
R2L_HD void r2l_stream_fetch_row_bf(const R2LStaticArgs& a, size_t img0, int ys, unsigned xo, unsigned xl, unsigned xr,
                                    bool le, bool re, R2LRowStageT<RAWK>& st) {
  static_assert(RAWK != R2L_RAW_F64, "float64 frames keep the conditional form");
  const size_t e = img0 + (size_t)ys * a.W;  // wave-uniform
  st.ys = ys;
  float* v = st.v;
  // (xo, xl, xr: BYTE offsets, 32-bit: uniform 64-bit base + zero-extended 32-bit lane offset is the saddr form)
  if (RAWK == R2L_RAW_U16) {
    const char* r = (const char*)(a.raw.u16 + e);
    const r2l_f2 c = *(const r2l_f2*)(r + xo);
    v[1] = c.x;
    v[2] = c.y;
    v[0] = *(const float*)(r + xl);  // LANES: xl = the pair on the lane's side (left for lanes 0-31, right for 32-63)
    v[3] = LANES ? 0.f : *(const float*)(r + xr);
    return;
  }
  const char* r = (const char*)(a.raw.f32 + e);
  const r2l_f4 c = r2l_stream_load_f4((const float*)(r + xo));
  v[2] = c.x;
  v[3] = c.y;
  v[4] = c.z;
  v[5] = c.w;
  if (LANES) {
    const r2l_f2 q = *(const r2l_f2*)(r + xl);
    v[0] = q.x;
    v[1] = q.y;
    v[6] = v[7] = 0.f;
  } else {
    const r2l_f2 l = *(const r2l_f2*)(r + xl), q = *(const r2l_f2*)(r + xr);
    v[0] = le ? c.y : l.x;  // x = -2 -> 1, x = -1 -> 0 (symmetric extension)
    v[1] = le ? c.x : l.y;
    v[6] = re ? c.w : q.x;  // x = W -> W-1, x = W+1 -> W-2
    v[7] = re ? c.z : q.y;
  }
}
template <int RAWK, bool LANES>
R2L_HD void r2l_stream_fetch_row(const R2LStaticArgs& a, size_t img0, int ys, int x0, bool le, bool re,
                                 R2LRowStageT<RAWK>& st) {
  const size_t e = img0 + (size_t)ys * a.W + x0;
  st.ys = ys;
  if constexpr (RAWK == R2L_RAW_F64) {
    const double* r = a.raw.f64 + e;
    double* v = st.v;
    const r2l_d2 c0 = *(const r2l_d2*)r, c1 = *(const r2l_d2*)(r + 2);
    v[2] = c0.x;
    v[3] = c0.y;
    v[4] = c1.x;
    v[5] = c1.y;
    if (le) {  // x = -1 -> 0, x = -2 -> 1
      v[0] = c0.y;
      v[1] = c0.x;
    } else {
      const r2l_d2 l = *(const r2l_d2*)(r - 2);
      v[0] = l.x;
      v[1] = l.y;
    }
    if (re) {  // x = W -> W-1, x = W+1 -> W-2
      v[6] = c1.y;
      v[7] = c1.x;
    } else {
      const r2l_d2 q = *(const r2l_d2*)(r + 4);
      v[6] = q.x;
      v[7] = q.y;
    }
  } else {
  constexpr bool U16 = RAWK == R2L_RAW_U16;
  float* v = st.v;
  if (U16) {
    const unsigned short* r = a.raw.u16 + e;
    const r2l_f2 c = *(const r2l_f2*)r;
    v[1] = c.x;
    v[2] = c.y;
    if (LANES) {
      const int lane = (x0 >> 2) & 63;
      v[0] = v[3] = 0.f;
      if ((lane == 0 && !le) || (lane == 63 && !re)) v[0] = *(const float*)(lane == 0 ? r - 2 : r + 4);
      return;
    }
    v[0] = le ? 0.f : *(const float*)(r - 2);
    v[3] = re ? 0.f : *(const float*)(r + 4);
    return;
  }
  const float* r = a.raw.f32 + e;
  const r2l_f4 c = r2l_stream_load_f4(r);
  v[2] = c.x;
  v[3] = c.y;
  v[4] = c.z;
  v[5] = c.w;
  if (LANES) {
    const int lane = (x0 >> 2) & 63;
    v[0] = v[1] = v[6] = v[7] = 0.f;
    if ((lane == 0 && !le) || (lane == 63 && !re)) {
      const r2l_f2 q = *(const r2l_f2*)(lane == 0 ? r - 2 : r + 4);
      v[0] = q.x;
      v[1] = q.y;
    }
    return;
  }
  if (le) {  // x = -1 -> 0, x = -2 -> 1
    v[0] = c.y;
    v[1] = c.x;
  } else {
    const r2l_f2 l = *(const r2l_f2*)(r - 2);
    v[0] = l.x;
    v[1] = l.y;
  }
  if (re) {  // x = W -> W-1, x = W+1 -> W-2
    v[6] = c.w;
    v[7] = c.z;
  } else {
    const r2l_f2 q = *(const r2l_f2*)(r + 4);
    v[6] = q.x;
    v[7] = q.y;
  }
  }
}
// staged row -> 8 black-level-corrected float64 values (the black level follows the SOURCE site).  The
// subtraction happens in the arithmetic of the frames' dtype, like the reference's in-place remove_blacklv
// (pipeline_numpy.py:152-158): float32 frames (and 16-bit containers, which the datasets normalise in float32,
// dataset.py:86-87) subtract the float32-rounded black level in float32; float64 frames subtract in float64.
// DT = double: the window holds float64 values; DT = float (float32 / 16-bit frames only): the window keeps the float32
// values the subtraction produced and its users widen on the fly -- the same numbers, half the registers.
template <int RAWK, bool LANES, class DT = double>
R2L_HD void r2l_stream_convert_row(const R2LStaticArgs& a, const R2LRowStageT<RAWK>& st, bool le, bool re,
                                   DT dst[8]) {
  const int ys = st.ys;
  if constexpr (RAWK == R2L_RAW_F64) {
    const double be = (ys & 1) ? a.bl[2] : a.bl[0], bo = (ys & 1) ? a.bl[3] : a.bl[1];
    // source column parities: x0-2 even, x0-1 odd, ..., except the mirrored ones (1, 0 | W-1, W-2)
    dst[0] = (DT)(st.v[0] - (le ? bo : be));
    dst[1] = (DT)(st.v[1] - (le ? be : bo));
    dst[2] = (DT)(st.v[2] - be);
    dst[3] = (DT)(st.v[3] - bo);
    dst[4] = (DT)(st.v[4] - be);
    dst[5] = (DT)(st.v[5] - bo);
    dst[6] = (DT)(st.v[6] - (re ? bo : be));
    dst[7] = (DT)(st.v[7] - (re ? be : bo));
  } else {
  constexpr bool U16 = RAWK == R2L_RAW_U16;
  float v[8];
  if (U16) {
    const unsigned c0 = r2l_f2u(st.v[1]), c1 = r2l_f2u(st.v[2]);
    unsigned l = r2l_f2u(st.v[0]), q = r2l_f2u(st.v[3]);
    if (LANES) {
      const int lane = R2L_LANE_ID;
      l = r2l_f2u(r2l_wave_shr1(st.v[2]));
      q = r2l_f2u(r2l_wave_shl1(st.v[1]));
      if (lane == 0) l = r2l_f2u(st.v[0]);
      if (lane == 63) q = r2l_f2u(st.v[0]);
    }
    v[2] = r2l_raw_decode(c0 & 0xffffu, a.raw);
    v[3] = r2l_raw_decode(c0 >> 16, a.raw);
    v[4] = r2l_raw_decode(c1 & 0xffffu, a.raw);
    v[5] = r2l_raw_decode(c1 >> 16, a.raw);
    v[0] = le ? v[3] : r2l_raw_decode(l & 0xffffu, a.raw);
    v[1] = le ? v[2] : r2l_raw_decode(l >> 16, a.raw);
    v[6] = re ? v[5] : r2l_raw_decode(q & 0xffffu, a.raw);
    v[7] = re ? v[4] : r2l_raw_decode(q >> 16, a.raw);
  } else {
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 8; ++i) v[i] = st.v[i];
    if (LANES) {
      const int lane = R2L_LANE_ID;
      v[0] = r2l_wave_shr1(st.v[4]);  // columns x0-2, x0-1 = the left lane's x0+2, x0+3
      v[1] = r2l_wave_shr1(st.v[5]);
      v[6] = r2l_wave_shl1(st.v[2]);  // columns x0+4, x0+5 = the right lane's x0, x0+1
      v[7] = r2l_wave_shl1(st.v[3]);
      if (lane == 0) {
        v[0] = le ? st.v[3] : st.v[0];
        v[1] = le ? st.v[2] : st.v[1];
      }
      if (lane == 63 || re) {
        v[6] = re ? st.v[5] : st.v[0];
        v[7] = re ? st.v[4] : st.v[1];
      }
    }
  }
  const float be = (ys & 1) ? a.blf[2] : a.blf[0], bo = (ys & 1) ? a.blf[3] : a.blf[1];
  dst[0] = (DT)(v[0] - (le ? bo : be));
  dst[1] = (DT)(v[1] - (le ? be : bo));
  dst[2] = (DT)(v[2] - be);
  dst[3] = (DT)(v[3] - bo);
  dst[4] = (DT)(v[4] - be);
  dst[5] = (DT)(v[5] - bo);
  dst[6] = (DT)(v[6] - (re ? bo : be));
  dst[7] = (DT)(v[7] - (re ? be : bo));
  }
}

// float32 window value -> float64, re-done at every use: left alone the compiler keeps the float64 copy of every window
// value alive across the rows that use it -- the float64 window again
R2L_HD double r2l_widen(double x) { return x; }
R2L_HD double r2l_widen(float x) {
#ifndef R2L_EMUL
  double d;
  asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(x));  // (volatile: identical statements are not merged)
  return d;
#else
  return (double)x;
#endif
}

// WB * CCM, clip, gamma and the three 16-byte stores of one output row (4 pixels of this lane)
R2L_HD void r2l_stream_finish_row(const R2LStaticArgs& a, const double d[4][3], float* outb, size_t plane,
                                  size_t off, bool ok = true) {
  float x[3][4];
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c)
    R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    const double rgb = fma(a.wbccm[k * 3], d[c][0], fma(a.wbccm[k * 3 + 1], d[c][1], a.wbccm[k * 3 + 2] * d[c][2]));
    x[k][c] = r2l_clip_gamma(rgb, a.inv_gamma);
  }
  r2l_static_normalize<4>(a, x);
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    r2l_f4 st;
    st.x = x[k][0];
    st.y = x[k][1];
    st.z = x[k][2];
    st.w = x[k][3];
    if (ok) r2l_stream_store_f4(outb + (size_t)k * plane + off, st);
  }
}

// first pass of a multi-pass chain: luma of the 4 pixels -> float64 plane
R2L_HD void r2l_stream_luma_out_row(const R2LStaticArgs& a, const double d[4][3], double* yplane, size_t off) {
  double y[4];
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) y[c] = fma(a.T[0], d[c][0], fma(a.T[1], d[c][1], a.T[2] * d[c][2]));
  double* o = yplane + off;
  o[0] = y[0];
  o[1] = y[1];
  o[2] = y[2];
  o[3] = y[3];
}
// last pass: chroma from the raw frame, filtered luma from the plane; rgb = rgb_from_yuv * (Y'', U, V)
R2L_HD void r2l_stream_luma_in_row(const R2LStaticArgs& a, const double d[4][3], const double* yplane, float* outb,
                                   size_t plane, size_t off) {
  float x[3][4];
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) {
    const double yy = yplane[off + c];
    const double u = fma(a.T[3], d[c][0], fma(a.T[4], d[c][1], a.T[5] * d[c][2]));
    const double v = fma(a.T[6], d[c][0], fma(a.T[7], d[c][1], a.T[8] * d[c][2]));
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      const double rgb = fma(a.M2[k * 3], yy, fma(a.M2[k * 3 + 1], u, a.M2[k * 3 + 2] * v));
      x[k][c] = r2l_clip_gamma(rgb, a.inv_gamma);
    }
  }
  r2l_static_normalize<4>(a, x);
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    r2l_f4 st;
    st.x = x[k][0];
    st.y = x[k][1];
    st.z = x[k][2];
    st.w = x[k][3];
    r2l_stream_store_f4(outb + (size_t)k * plane + off, st);
  }
}

// the same up to yuv2rgb, stored as float64 planes (the input of fft_denoising)
R2L_HD void r2l_stream_lin_out_row(const R2LStaticArgs& a, const double d[4][3], const double* yplane, double* linb,
                                   size_t plane, size_t off) {
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) {
    const double yy = yplane[off + c];
    const double u = fma(a.T[3], d[c][0], fma(a.T[4], d[c][1], a.T[5] * d[c][2]));
    const double v = fma(a.T[6], d[c][0], fma(a.T[7], d[c][1], a.T[8] * d[c][2]));
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k)
      linb[(size_t)k * plane + off + c] = fma(a.M2[k * 3], yy, fma(a.M2[k * 3 + 1], u, a.M2[k * 3 + 2] * v));
  }
}

// bilinear row: w0/w1/w2 = window rows y-1, y, y+1 (8 values each); tpy = their source row parities
template <class WT>
R2L_HD void r2l_stream_bilinear_row(const WT* w0, const WT* w1, const WT* w2, const int tpy[3],
                                    bool le, bool re, double d[4][3]) {
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) {
    double n[3][3];
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      n[0][j] = r2l_widen(w0[1 + c + j]);
      n[1][j] = r2l_widen(w1[1 + c + j]);
      n[2][j] = r2l_widen(w2[1 + c + j]);
    }
    int tpx[3] = {1 - (c & 1), c & 1, 1 - (c & 1)};
    // a tap mirrored back from outside the image keeps the site of the column it came from
    if (c == 0 && le) tpx[0] = 0;
    if (c == 3 && re) tpx[2] = 1;
    r2l_bilinear_px(n, tpy, tpx, d[c]);
  }
}

// The same convolution on rows y-1, y, y+1 inside the image, where the row masks are the plain checkerboard: per
// site the masked sums collapse to the pixel itself, the mean of its 2 row / column neighbours, of its 4 edge or
// of its 4 corner neighbours.  u/m/l = window rows y-1, y, y+1; vertical pair sums are shared by the 4 pixels.
// (Summation order differs from the masked form by round-off of float64 only.)
// Left / right image edge (le: pixel c = 0 is column 0; re: pixel c = 3 is column W-1): the column mirrored back
// from outside is the edge column itself, so its taps carry the EDGE column's sites, not the checkerboard's.
// In the closed form that is: the outer column contributes nothing to what the checkerboard expected there
// (its values are zeroed), and the taps of that column (weights [1 2 1]/4 for R/B, [0 1 0]/4 for G) add to the
// channels the edge column itself holds: the pixel's own channel gets +1/2 (R/B) or +1/4 (G) of the pixel, and
// the channel of the rows above / below gets +1/4 of their sum when it is R or B.
template <int PY, class WT>
R2L_HD void r2l_stream_bilinear_row_interior(const WT* u, const WT* m_, const WT* l, bool le, bool re,
                                             double d[4][3]) {
  double vs[8], hs[4], m[8];
  R2L_PRAGMA_UNROLL
  for (int j = 1; j < 7; ++j) {
    vs[j] = r2l_widen(u[j]) + r2l_widen(l[j]);
    m[j] = r2l_widen(m_[j]);
  }
  const double fl = le ? 1.0 : 0.0, fr = re ? 1.0 : 0.0;
  const double m1 = le ? 0.0 : m[1], m6 = re ? 0.0 : m[6];
  vs[1] = le ? 0.0 : vs[1];
  vs[6] = re ? 0.0 : vs[6];
  hs[0] = m1 + m[3];
  hs[1] = m[2] + m[4];
  hs[2] = m[3] + m[5];
  hs[3] = m[4] + m6;
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) {
    const int j = 2 + c;
    const double corners = (vs[j - 1] + vs[j + 1]) * 0.25, edges = (vs[j] + hs[c]) * 0.25;
    if (PY == 0 && (c & 1) == 0) {  // R site
      d[c][0] = m[j];
      d[c][1] = edges;
      d[c][2] = corners;
    } else if (PY == 0) {  // G site of an R row: R left/right, B above/below
      d[c][0] = hs[c] * 0.5;
      d[c][1] = m[j];
      d[c][2] = vs[j] * 0.5;
    } else if ((c & 1) == 0) {  // G site of a B row
      d[c][0] = vs[j] * 0.5;
      d[c][1] = m[j];
      d[c][2] = hs[c] * 0.5;
    } else {  // B site
      d[c][0] = corners;
      d[c][1] = edges;
      d[c][2] = m[j];
    }
  }
  if (PY == 0) {
    d[0][0] = fma(fl, 0.5 * m[2], d[0][0]);   // column 0 is R here, G2 above / below
    d[3][1] = fma(fr, 0.25 * m[5], d[3][1]);  // column W-1 is G1 here, B above / below
    d[3][2] = fma(fr, 0.25 * vs[5], d[3][2]);
  } else {
    d[0][1] = fma(fl, 0.25 * m[2], d[0][1]);  // column 0 is G2 here, R above / below
    d[0][0] = fma(fl, 0.25 * vs[2], d[0][0]);
    d[3][2] = fma(fr, 0.5 * m[5], d[3][2]);   // column W-1 is B here, G1 above / below
  }
}

template <int PY>
R2L_HD void r2l_stream_malvar_row(const double* w0, const double* w1, const double* w2, const double* w3,
                                  const double* w4, double d[4][3]) {
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) {
    double n[5][5];
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 5; ++j) {
      n[0][j] = w0[c + j];
      n[1][j] = w1[c + j];
      n[2][j] = w2[c + j];
      n[3][j] = w3[c + j];
      n[4][j] = w4[c + j];
    }
    r2l_malvar_px(n, PY, c & 1, d[c]);
  }
}

// Malvar2004 for the 4 pixels of a lane (w0 .. w4 = window rows y-2 .. y+2, 8 values each: columns x0-2 .. x0+5).
// The four 5x5 kernels (r2l_malvar_* in r2l_static_kernels.h) are sums of a few symmetric groups of taps; the
// vertical pair sums of a column serve every pixel whose window holds that column, so they are formed once per
// column instead of once per pixel and kernel: ~60 float64 operations per row and lane instead of ~128.
// (Summation order differs from the per-pixel form by float64 round-off only.)  Used by the luma-chain kernels
// (r2l_static_chain.h: -2..3 %); in the short chain's kernel its 20 more live registers cost the third wavefront per
// SIMD or spill (875 -> 1070-1165 us), so that one keeps the per-pixel form above.
// WT = float: a float32 window (float32 / 16-bit frames), widened here -- exactly, so the results are those of the float64
// window bit for bit.  That is what lets the short chain's kernel use this form: its 5-row float64 window alone was 80 of
// its 167 registers.
template <int PY, class WT = double>
R2L_HD void r2l_stream_malvar_row_shared(const WT* w0, const WT* w1, const WT* w2, const WT* w3,
                                  const WT* w4, double d[4][3]) {
  double v1[8], v2[8];  // rows y-1 + y+1 (columns 1 .. 6 used), rows y-2 + y+2 (columns 2 .. 5 used)
  double wm[8];         // the pixel's own row
  // (r2l_widen: the conversion is re-done per output row; left alone the compiler keeps the float64 copy of every window
  // value alive across the rows that use it -- the float64 window again)
  R2L_PRAGMA_UNROLL
  for (int j = 0; j < 8; ++j) wm[j] = r2l_widen(w2[j]);
  R2L_PRAGMA_UNROLL
  for (int j = 1; j < 7; ++j) v1[j] = r2l_widen(w1[j]) + r2l_widen(w3[j]);
  R2L_PRAGMA_UNROLL
  for (int j = 2; j < 6; ++j) v2[j] = r2l_widen(w0[j]) + r2l_widen(w4[j]);
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) {
    const int m = c + 2;  // window column of the pixel
    const double ctr = wm[m];
    const double h1 = wm[m - 1] + wm[m + 1], h2 = wm[m - 2] + wm[m + 2];
    const double dg = v1[m - 1] + v1[m + 1];  // the four diagonal neighbours
    const double far = v2[m] + h2;
    if ((c & 1) == PY) {  // R site (PY 0, even column) or B site (PY 1, odd column): G and the opposite colour
      const double g = (fma(2.0, v1[m] + h1, 4.0 * ctr) - far) * 0.125;
      const double o = fma(-1.5, far, fma(2.0, dg, 6.0 * ctr)) * 0.125;
      d[c][PY ? 2 : 0] = ctr;
      d[c][1] = g;
      d[c][PY ? 0 : 2] = o;
    } else {  // G site: the colour of its row from the horizontal kernel, the other from the vertical one
      const double hz = (fma(0.5, v2[m], fma(4.0, h1, 5.0 * ctr)) - dg - h2) * 0.125;
      const double vt = (fma(0.5, h2, fma(4.0, v1[m], 5.0 * ctr)) - dg - v2[m]) * 0.125;
      d[c][1] = ctr;
      d[c][PY ? 2 : 0] = hz;   // red row (PY 0): R left / right; blue row: B left / right
      d[c][PY ? 0 : 2] = vt;
    }
  }
}

#ifndef R2L_STREAM_MALVAR_F32WIN
#define R2L_STREAM_MALVAR_F32WIN 1  // 0: A/B builds, float64 window + per-pixel kernels (round 2)
#endif
template <bool F32>
struct R2LWinType {
  typedef double type;
};
template <>
struct R2LWinType<true> {
  typedef float type;
};
// bilinear keeps its float64 window (3 rows); so does Malvar2004 on float64 frames, in the per-pixel form
template <int PY, bool WF32>
R2L_HD void r2l_stream_malvar_rowT(const float* w0, const float* w1, const float* w2, const float* w3, const float* w4,
                                   double d[4][3]) {
  r2l_stream_malvar_row_shared<PY, float>(w0, w1, w2, w3, w4, d);
}
template <int PY, bool WF32>
R2L_HD void r2l_stream_malvar_rowT(const double* w0, const double* w1, const double* w2, const double* w3,
                                   const double* w4, double d[4][3]) {
  r2l_stream_malvar_row<PY>(w0, w1, w2, w3, w4, d);
}
#ifndef R2L_STREAM_PF_BILINEAR
#define R2L_STREAM_PF_BILINEAR 5
#endif
#ifndef R2L_STREAM_PF_MALVAR
#define R2L_STREAM_PF_MALVAR 2
#endif
// one lane's work item: image b, column strip seg (256 columns), row band
template <int DEB, int RAWK, bool LUMA>
R2L_HD void r2l_static_stream_item(const R2LStaticStreamArgs& sa, int item, int lane) {
  const R2LStaticArgs& a = sa.s;
  constexpr int HALO = DEB ? 2 : 1, NR = 2 * HALO + 1;
  constexpr bool LANES = (DEB == 0) && R2L_HAVE_LANE_SHIFTS && RAWK != R2L_RAW_F64;
  const int seg = item % sa.nseg, r = item / sa.nseg;
  const int band = r % sa.nband, b = r / sa.nband;
  const int x0 = seg * 256 + 4 * lane;
  if (x0 >= a.W) {
    R2L_LANE_RETIRES();
    return;
  }
  const int y0 = band * sa.band_h;
  const int y1 = (y0 + sa.band_h < a.H) ? y0 + sa.band_h : a.H;
  const bool le = x0 == 0, re = x0 + 4 >= a.W;
  const size_t plane = (size_t)a.H * a.W;
  const size_t img = (size_t)b * plane;  // element offset of image b
  float* outb = a.out + (size_t)b * 3 * plane;
  // Malvar2004 on float32 / 16-bit frames: a float32 window (r2l_stream_malvar_row_shared widens it)
  constexpr bool WF32 = (DEB == 1) && RAWK != R2L_RAW_F64 && R2L_STREAM_MALVAR_F32WIN;
  typename R2LWinType<WF32>::type win[NR][8];
  int par[NR];  // source row parity of each window slot
  // warm-up: slots 0..NR-2 hold rows y0-HALO .. y0+HALO-1
  R2LRowStageT<RAWK> st;
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < NR - 1; ++i) {
    const int ys = r2l_symmetric(y0 - HALO + i, a.H);
    r2l_stream_fetch_row<RAWK, LANES>(a, img, ys, x0, le, re, st);
    r2l_stream_convert_row<RAWK, LANES, typename R2LWinType<WF32>::type>(a, st, le, re, win[i]);
    par[i] = ys & 1;
  }
  // software pipeline, PF rows deep: the rows needed by the next PF output rows are in flight while this one
  // is computed (pf[i] = row y+HALO+i).  At 3-4 wavefronts per SIMD (the float64 window costs the registers)
  // the bytes in flight come from the depth, not from the occupancy.
  constexpr int PF = (DEB || RAWK == R2L_RAW_F64) ? R2L_STREAM_PF_MALVAR : R2L_STREAM_PF_BILINEAR;
  R2LRowStageT<RAWK> pf[PF];
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < PF; ++i) r2l_stream_fetch_row<RAWK, LANES>(a, img, r2l_symmetric(y0 + HALO + i, a.H), x0, le, re, pf[i]);
  for (int yb = y0; yb < y1; yb += NR) {
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < NR; ++k) {  // unrolled by the window depth: slot indices are compile-time
      const int y = yb + k;
      if (y < y1) {
        // newest row y + HALO (fetched one iteration ago) goes to slot (k + NR - 1) % NR
        r2l_stream_convert_row<RAWK, LANES, typename R2LWinType<WF32>::type>(a, pf[0], le, re, win[(k + NR - 1) % NR]);
        par[(k + NR - 1) % NR] = pf[0].ys & 1;
        R2L_PRAGMA_UNROLL
        for (int i = 0; i + 1 < PF; ++i) pf[i] = pf[i + 1];
        if (y + PF < y1) r2l_stream_fetch_row<RAWK, LANES>(a, img, r2l_symmetric(y + PF + HALO, a.H), x0, le, re, pf[PF - 1]);
        double d[4][3];
        if constexpr (DEB == 0) {
          // interior rows: closed-form sums (left / right image edge included); the first / last image row sees
          // a mirrored row and takes the masked form
          if (y > 0 && y < a.H - 1) {
            if (y & 1)
              r2l_stream_bilinear_row_interior<1>(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], le, re, d);
            else
              r2l_stream_bilinear_row_interior<0>(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], le, re, d);
          } else {
            const int tpy[3] = {par[k % NR], par[(k + 1) % NR], par[(k + 2) % NR]};
            r2l_stream_bilinear_row(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], tpy, le, re, d);
          }
        } else {
          if (y & 1)
            r2l_stream_malvar_rowT<1, WF32>(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], win[(k + 3) % NR],
                                            win[(k + 4) % NR], d);
          else
            r2l_stream_malvar_rowT<0, WF32>(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], win[(k + 3) % NR],
                                            win[(k + 4) % NR], d);
        }
        if (LUMA && sa.luma_out)
          r2l_stream_luma_out_row(a, d, sa.luma_out, img + (size_t)y * a.W + x0);
        else if (LUMA && sa.lin_out)
          r2l_stream_lin_out_row(a, d, sa.luma_in + img, sa.lin_out + 3 * img, plane, (size_t)y * a.W + x0);
        else if (LUMA)
          r2l_stream_luma_in_row(a, d, sa.luma_in + img, outb, plane, (size_t)y * a.W + x0);
        else
          r2l_stream_finish_row(a, d, outb, plane, (size_t)y * a.W + x0);
      }
    }
  }
}

// ---- row stage with SCALAR strip edges (branch-free kernels) ----------------------------------------------------------
// The columns beyond a lane's four come from the neighbouring lanes (DPP wave shifts); only the strip's first / last lane
// need values of the neighbouring strips -- two wave-uniform addresses per row: ONE s_load_dwordx2 each (scalar cache, no
// vector-memory instruction, no per-lane address, no condition), carried in scalar registers until the row enters the
// window.  At the image's left / right edge the scalar loads read in-row dummies and the lanes there mirror their own
// columns (symmetric extension).
#ifndef R2L_EMUL
R2L_HD float r2l_swshr(float x, float edge) {  // previous lane's x; lane 0 of the wavefront gets `edge`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x),
                                                               0x138, 0xf, 0xf, false));
}
R2L_HD float r2l_swshl(float x, float edge) {  // next lane's x; lane 63 gets `edge`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x),
                                                               0x130, 0xf, 0xf, false));
}
template <int RAWK>
struct R2LRowStageS {
  r2l_f4 c;       // the lane's 4 columns (16-bit containers: c.x, c.y = the 4 undecoded values)
  r2l_f2 sl, sr;  // wave-uniform: columns (strip - 2, strip - 1) and (strip + 256, strip + 257) (16-bit: .x = both, undecoded)
  int ys;
};
// sbl / sbr: wave-uniform BYTE offsets of the two edge pairs inside a row; xo: the lane's byte offset
template <int RAWK>
R2L_HD void r2l_stream_fetch_row_s(const R2LStaticArgs& a, size_t img0, int ys, unsigned xo, unsigned sbl, unsigned sbr,
                                   R2LRowStageS<RAWK>& st) {
  static_assert(RAWK != R2L_RAW_F64, "float32 / 16-bit frames");
  const size_t e = img0 + (size_t)ys * a.W;  // wave-uniform
  st.ys = ys;
  if (RAWK == R2L_RAW_U16) {
    const char* r = (const char*)(a.raw.u16 + e);
    const r2l_f2 c = *(const r2l_f2*)(r + xo);
    st.c.x = c.x;
    st.c.y = c.y;
    st.c.z = st.c.w = 0.f;
    st.sl.x = *(const __attribute__((address_space(4))) float*)(r + sbl);
    st.sr.x = *(const __attribute__((address_space(4))) float*)(r + sbr);
    st.sl.y = st.sr.y = 0.f;
    return;
  }
  const char* r = (const char*)(a.raw.f32 + e);
  st.c = r2l_stream_load_f4((const float*)(r + xo));
  typedef float v2_t __attribute__((ext_vector_type(2)));
  const v2_t tl = *(const __attribute__((address_space(4))) v2_t*)(r + sbl);
  const v2_t tr = *(const __attribute__((address_space(4))) v2_t*)(r + sbr);
  st.sl.x = tl.x;
  st.sl.y = tl.y;
  st.sr.x = tr.x;
  st.sr.y = tr.y;
}
// staged row -> 8 black-level-corrected values (columns x0-2 .. x0+5); le0 / re63: the strip starts / ends at the image's
// edge (wave-uniform); le / re: this lane holds the image's first / last 4 columns
template <int RAWK, class DT>
R2L_HD void r2l_stream_convert_row_s(const R2LStaticArgs& a, const R2LRowStageS<RAWK>& st, bool le0, bool le, bool re,
                                     DT dst[8]) {
  float c0, c1, c2, c3, l0, l1, r0, r1;
  if (RAWK == R2L_RAW_U16) {
    const unsigned lo = r2l_f2u(st.c.x), hi = r2l_f2u(st.c.y), sl = r2l_f2u(st.sl.x), sr = r2l_f2u(st.sr.x);
    c0 = r2l_raw_decode(lo & 0xffffu, a.raw);
    c1 = r2l_raw_decode(lo >> 16, a.raw);
    c2 = r2l_raw_decode(hi & 0xffffu, a.raw);
    c3 = r2l_raw_decode(hi >> 16, a.raw);
    l0 = r2l_raw_decode(sl & 0xffffu, a.raw);
    l1 = r2l_raw_decode(sl >> 16, a.raw);
    r0 = r2l_raw_decode(sr & 0xffffu, a.raw);
    r1 = r2l_raw_decode(sr >> 16, a.raw);
  } else {
    c0 = st.c.x;
    c1 = st.c.y;
    c2 = st.c.z;
    c3 = st.c.w;
    l0 = st.sl.x;
    l1 = st.sl.y;
    r0 = st.sr.x;
    r1 = st.sr.y;
  }
  float v[8];
  // columns x0-2, x0-1 = the left lane's x0+2, x0+3; the strip's first lane: the neighbouring strip's last two columns,
  // or (image edge, symmetric extension) its own columns 1, 0
  v[0] = r2l_swshr(c2, le0 ? c1 : l0);
  v[1] = r2l_swshr(c3, le0 ? c0 : l1);
  v[2] = c0;
  v[3] = c1;
  v[4] = c2;
  v[5] = c3;
  const float q0 = r2l_swshl(c0, r0), q1 = r2l_swshl(c1, r1);  // columns x0+4, x0+5 = the right lane's x0, x0+1
  v[6] = re ? c3 : q0;  // x = W -> W-1, x = W+1 -> W-2
  v[7] = re ? c2 : q1;
  const int ys = st.ys;
  const float be = (ys & 1) ? a.blf[2] : a.blf[0], bo = (ys & 1) ? a.blf[3] : a.blf[1];
  dst[0] = (DT)(v[0] - (le ? bo : be));
  dst[1] = (DT)(v[1] - (le ? be : bo));
  dst[2] = (DT)(v[2] - be);
  dst[3] = (DT)(v[3] - bo);
  dst[4] = (DT)(v[4] - be);
  dst[5] = (DT)(v[5] - bo);
  dst[6] = (DT)(v[6] - (re ? bo : be));
  dst[7] = (DT)(v[7] - (re ? be : bo));
}
#endif

// The same work item with a BRANCH-FREE row loop (float32 / 16-bit frames, whole short chain): every fetch is
// unconditional (r2l_stream_fetch_row_bf; rows past the band's last needed row re-fetch that row: L1 hits), the prefetch
// ring is indexed by the unroll position (U = lcm(window depth, PF) steps per group: no register copies), every group runs
// in full and only the stores are predicated (rows past the band's end, lanes past the frame's last column: a clamped
// strip).  hipcc then counts its vmcnt waits: a row step waits for the row it requested PF steps ago and for nothing else.
#ifndef R2L_STREAM_BF
#define R2L_STREAM_BF 2  // 0: off; 1: bilinear and Malvar2004; 2: Malvar2004 only (bilinear is faster in its round-1 form)
#endif
#ifndef R2L_STREAM_BF_PF_BILINEAR
#define R2L_STREAM_BF_PF_BILINEAR 6
#endif
#ifndef R2L_STREAM_BF_PF_MALVAR
#define R2L_STREAM_BF_PF_MALVAR 5
#endif
#ifndef R2L_STREAM_BF_MALVAR_LANES
#define R2L_STREAM_BF_MALVAR_LANES 0  // 1: Malvar2004's two neighbour columns each side by lane shifts (one edge pair per lane and row in flight instead of two)
#endif
#ifndef R2L_STREAM_BF_BILINEAR_F32WIN
#define R2L_STREAM_BF_BILINEAR_F32WIN 1
#endif
#ifndef R2L_STREAM_BF_SCALAR_EDGES
#define R2L_STREAM_BF_SCALAR_EDGES 0  // 1: strip edges through scalar loads + lane shifts (measured slower, r04_ab_static_se.txt)
#endif
template <int DEB, int RAWK>
R2L_HD void r2l_static_stream_item_bf(const R2LStaticStreamArgs& sa, int item, int lane) {
  const R2LStaticArgs& a = sa.s;
  constexpr int HALO = DEB ? 2 : 1, NR = 2 * HALO + 1;
  constexpr bool LANES = (DEB == 0 || R2L_STREAM_BF_MALVAR_LANES) && R2L_HAVE_LANE_SHIFTS;
  constexpr int PF = DEB ? R2L_STREAM_BF_PF_MALVAR : R2L_STREAM_BF_PF_BILINEAR;
  constexpr int U = (PF % NR == 0) ? PF : PF * NR;  // steps per unrolled group: a multiple of both ring sizes
  static_assert(U % NR == 0 && U % PF == 0, "ring slots are compile-time indices");
  const int seg = item % sa.nseg, r = item / sa.nseg;
  const int band = r % sa.nband, b = r / sa.nband;
  const int xs = seg * 256 + 4 * lane;
  const bool in_w = xs < a.W;
  const int x0 = in_w ? xs : a.W - 4;  // lanes past the last column walk the last 4 columns and store nothing
  const int y0 = band * sa.band_h;
  const int y1 = (y0 + sa.band_h < a.H) ? y0 + sa.band_h : a.H;
  const bool le = x0 == 0, re = x0 + 4 >= a.W;
  const size_t plane = (size_t)a.H * a.W;
  const size_t img = (size_t)b * plane;
  float* outb = a.out + (size_t)b * 3 * plane;
  // (float32 window for bilinear too: 24 registers instead of 48, widened exactly where used -- what makes room for the
  // deeper prefetch ring at three wavefronts per SIMD)
  constexpr bool WF32 = DEB ? (R2L_STREAM_MALVAR_F32WIN != 0) : (R2L_STREAM_BF_BILINEAR_F32WIN != 0);
  typedef typename R2LWinType<WF32>::type WT;
  WT win[NR][8];
  int par[NR];
  // the lane's element offsets inside a row: its 4 columns, the pair on its left (x0 - 2; at the image's left edge its
  // own columns) and on its right (x0 + 4; its own); LANES: xl = the pair on the lane's side of the wavefront
  constexpr unsigned ESZ = (RAWK == R2L_RAW_U16) ? 2u : 4u;  // (as byte offsets: r2l_stream_fetch_row_bf / _s)
  const unsigned xo = ESZ * (unsigned)x0, xleft = ESZ * (unsigned)(x0 + (le ? 0 : -2)), xright = ESZ * (unsigned)(x0 + (re ? 2 : 4));
  const unsigned xl = LANES ? (lane < 32 ? xleft : xright) : xleft, xr = xright;
  // scalar strip edges (R2L_STREAM_BF_SCALAR_EDGES): the pairs left of the strip's first and right of its last lane
  const int sx0 = seg * 256;
  const bool le0 = sx0 == 0;
  const unsigned sbl = ESZ * (unsigned)(le0 ? 0 : sx0 - 2), sbr = ESZ * (unsigned)(sx0 + 256 + 2 <= a.W ? sx0 + 256 : a.W - 2);
  (void)xl; (void)xr; (void)sbl; (void)sbr; (void)le0;
  const int ylast = y1 - 1 + HALO;  // last source row (before the symmetric extension) this band needs
  // -DR2L_EXP_STATIC_NO_HALO (A/B builds; WRONG RESULTS, timing only): every halo row is fetched from inside the band -- the upper
  // bound of anything that shares the halo rows of neighbouring bands (the review's "adjacent bands in lock-step", round 6)
#ifdef R2L_EXP_STATIC_NO_HALO
#define R2L_BF_ROW(ys_) ((ys_) < y0 ? y0 : ((ys_) > y1 - 1 ? y1 - 1 : (ys_)))
#else
#define R2L_BF_ROW(ys_) (ys_)
#endif
#if R2L_STREAM_BF_SCALAR_EDGES
  typedef R2LRowStageS<RAWK> StageT;
#define R2L_BF_FETCH(ys_, st_) r2l_stream_fetch_row_s<RAWK>(a, img, R2L_BF_ROW(ys_), xo, sbl, sbr, (st_))
#define R2L_BF_CONVERT(st_, dst_) r2l_stream_convert_row_s<RAWK, WT>(a, (st_), le0, le, re, (dst_))
#else
  typedef R2LRowStageT<RAWK> StageT;
#define R2L_BF_FETCH(ys_, st_) r2l_stream_fetch_row_bf<RAWK, LANES>(a, img, R2L_BF_ROW(ys_), xo, xl, xr, le, re, (st_))
#define R2L_BF_CONVERT(st_, dst_) r2l_stream_convert_row<RAWK, LANES, WT>(a, (st_), le, re, (dst_))
#endif
  {
    StageT st;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < NR - 1; ++i) {
      const int ys = r2l_symmetric(y0 - HALO + i, a.H);
      R2L_BF_FETCH(ys, st);
      R2L_BF_CONVERT(st, win[i]);
      par[i] = ys & 1;
    }
  }
  StageT pf[PF];  // ring: step k consumes pf[k % PF] (row y + HALO) and refills it with row y + HALO + PF
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < PF; ++i) {
    const int yr = (y0 + HALO + i < ylast) ? y0 + HALO + i : ylast;
    R2L_BF_FETCH(r2l_symmetric(yr, a.H), pf[i]);
  }
  for (int yb = y0; yb < y1; yb += U) {
#ifdef R2L_STREAM_PROGRESS_PRIO
    R2L_PROGRESS_PRIO(yb - y0, y1 - y0);
#endif
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < U; ++k) {
      const int y = yb + k;
      R2L_BF_CONVERT(pf[k % PF], win[(k + NR - 1) % NR]);
      par[(k + NR - 1) % NR] = pf[k % PF].ys & 1;
      {
        const int yr = (y + HALO + PF < ylast) ? y + HALO + PF : ylast;
        R2L_BF_FETCH(r2l_symmetric(yr, a.H), pf[k % PF]);
      }
      double d[4][3];
      if constexpr (DEB == 0) {
        if (y > 0 && y < a.H - 1) {
          if (y & 1)
            r2l_stream_bilinear_row_interior<1>(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], le, re, d);
          else
            r2l_stream_bilinear_row_interior<0>(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], le, re, d);
        } else {
          const int tpy[3] = {par[k % NR], par[(k + 1) % NR], par[(k + 2) % NR]};
          r2l_stream_bilinear_row(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], tpy, le, re, d);
        }
      } else {
        if (y & 1)
          r2l_stream_malvar_rowT<1, WF32>(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], win[(k + 3) % NR],
                                          win[(k + 4) % NR], d);
        else
          r2l_stream_malvar_rowT<0, WF32>(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], win[(k + 3) % NR],
                                          win[(k + 4) % NR], d);
      }
      const int yc = y < y1 ? y : y1 - 1;  // (rows past the band's end: computed, not stored)
      r2l_stream_finish_row(a, d, outb, plane, (size_t)yc * a.W + x0, in_w && y < y1);
    }
  }
}

#undef R2L_BF_FETCH
#undef R2L_BF_CONVERT
#undef R2L_BF_ROW

#define R2L_STREAM_NT 256  // 4 independent wavefronts per workgroup
template <int DEB, int RAWK, bool LUMA>
R2L_BLOCKFN void r2l_static_stream_block(const R2LStaticStreamArgs& sa, int bid, int nblk, float* lds) {
  (void)lds;
  (void)nblk;
  bid = r2l_xcd_contiguous(bid, nblk);
  R2L_PHASE_BEGIN_N(R2L_STREAM_NT)
  const int item = bid * (R2L_STREAM_NT / 64) + (tid >> 6);  // one work item per wavefront
#if R2L_STREAM_BF && !defined(R2L_SERIAL)
  if constexpr (!LUMA && RAWK != R2L_RAW_F64 && (R2L_STREAM_BF == 1 || DEB == 1)) {
    // (the wavefront index as a SCALAR: the work item and its rows must not look lane-dependent)
    const int witem = bid * (R2L_STREAM_NT / 64) + __builtin_amdgcn_readfirstlane(tid >> 6);
    if (witem < sa.nitems) r2l_static_stream_item_bf<DEB, RAWK>(sa, witem, tid & 63);
    return;
  }
#endif
  if (item < sa.nitems) r2l_static_stream_item<DEB, RAWK, LUMA>(sa, item, tid & 63);
  R2L_PHASE_END
}
