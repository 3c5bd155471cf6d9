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
};

// raw values of columns x0-2 .. x0+5 of source row `ys` (symmetric extension at the image edges); with 16-bit
// containers the undecoded bits (v[0] = left pair, v[1..2] = the 4 centre values, v[3] = right pair), decoded
// when the row enters the window so that the fetch stays a fire-and-forget load
struct R2LRowStage {
  float v[8];
  int ys;
};
template <bool U16>
R2L_HD void r2l_stream_fetch_row(const R2LStaticArgs& a, size_t img0, int ys, int x0, bool le, bool re,
                                 R2LRowStage& st) {
  const size_t e = img0 + (size_t)ys * a.W + x0;
  float* v = st.v;
  st.ys = ys;
  if (U16) {
    const unsigned short* r = a.raw.u16 + e;
    const r2l_f2 c = *(const r2l_f2*)r;
    v[1] = c.x;
    v[2] = c.y;
    v[0] = le ? 0.f : *(const float*)(r - 2);
    v[3] = re ? 0.f : *(const float*)(r + 4);
    return;
  }
  const float* r = a.raw.f32 + e;
  const r2l_f4 c = *(const r2l_f4*)r;
  v[2] = c.x;
  v[3] = c.y;
  v[4] = c.z;
  v[5] = c.w;
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
// staged row -> 8 black-level-corrected float64 values (the black level follows the SOURCE site)
template <bool U16>
R2L_HD void r2l_stream_convert_row(const R2LStaticArgs& a, const R2LRowStage& st, bool le, bool re,
                                   double dst[8]) {
  float v[8];
  if (U16) {
    const unsigned l = r2l_f2u(st.v[0]), c0 = r2l_f2u(st.v[1]), c1 = r2l_f2u(st.v[2]), q = r2l_f2u(st.v[3]);
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
  }
  const int ys = st.ys;
  const double be = (ys & 1) ? a.bl[2] : a.bl[0], bo = (ys & 1) ? a.bl[3] : a.bl[1];
  // source column parities: x0-2 even, x0-1 odd, ..., except the mirrored ones (1, 0 | W-1, W-2)
  dst[0] = (double)v[0] - (le ? bo : be);
  dst[1] = (double)v[1] - (le ? be : bo);
  dst[2] = (double)v[2] - be;
  dst[3] = (double)v[3] - bo;
  dst[4] = (double)v[4] - be;
  dst[5] = (double)v[5] - bo;
  dst[6] = (double)v[6] - (re ? bo : be);
  dst[7] = (double)v[7] - (re ? be : bo);
}

// WB * CCM, clip, gamma and the three 16-byte stores of one output row (4 pixels of this lane)
R2L_HD void r2l_stream_finish_row(const R2LStaticArgs& a, const double d[4][3], float* outb, size_t plane,
                                  size_t off) {
  float x[3][4];
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c)
    R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    const double rgb = a.wbccm[k * 3] * d[c][0] + a.wbccm[k * 3 + 1] * d[c][1] + a.wbccm[k * 3 + 2] * d[c][2];
    const float xf = (float)fmin(fmax(rgb, 0.0), 1.0);                  // np.clip(img, 0, 1)   :138
    x[k][c] = (xf > 0.f) ? r2l_exp2(r2l_log2(xf) * a.inv_gamma) : 0.f;  // img ** (1 / gamma)   :243
  }
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    r2l_f4 st;
    st.x = x[k][0];
    st.y = x[k][1];
    st.z = x[k][2];
    st.w = x[k][3];
    *(r2l_f4*)(outb + (size_t)k * plane + off) = st;
  }
}

// first pass of a multi-pass chain: luma of the 4 pixels -> float64 plane
R2L_HD void r2l_stream_luma_out_row(const R2LStaticArgs& a, const double d[4][3], double* yplane, size_t off) {
  double y[4];
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) y[c] = a.T[0] * d[c][0] + a.T[1] * d[c][1] + a.T[2] * d[c][2];
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
    const double u = a.T[3] * d[c][0] + a.T[4] * d[c][1] + a.T[5] * d[c][2];
    const double v = a.T[6] * d[c][0] + a.T[7] * d[c][1] + a.T[8] * d[c][2];
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      const double rgb = a.M2[k * 3] * yy + a.M2[k * 3 + 1] * u + a.M2[k * 3 + 2] * v;
      const float xf = (float)fmin(fmax(rgb, 0.0), 1.0);
      x[k][c] = (xf > 0.f) ? r2l_exp2(r2l_log2(xf) * a.inv_gamma) : 0.f;
    }
  }
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    r2l_f4 st;
    st.x = x[k][0];
    st.y = x[k][1];
    st.z = x[k][2];
    st.w = x[k][3];
    *(r2l_f4*)(outb + (size_t)k * plane + off) = st;
  }
}

// bilinear row: w0/w1/w2 = window rows y-1, y, y+1 (8 values each); tpy = their source row parities
R2L_HD void r2l_stream_bilinear_row(const double* w0, const double* w1, const double* w2, const int tpy[3],
                                    bool le, bool re, double d[4][3]) {
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) {
    double n[3][3];
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      n[0][j] = w0[1 + c + j];
      n[1][j] = w1[1 + c + j];
      n[2][j] = w2[1 + c + j];
    }
    int tpx[3] = {1 - (c & 1), c & 1, 1 - (c & 1)};
    // a tap mirrored back from outside the image keeps the site of the column it came from
    if (c == 0 && le) tpx[0] = 0;
    if (c == 3 && re) tpx[2] = 1;
    r2l_bilinear_px(n, tpy, tpx, d[c]);
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

// one lane's work item: image b, column strip seg (256 columns), row band
template <int DEB, bool U16, bool LUMA>
R2L_HD void r2l_static_stream_item(const R2LStaticStreamArgs& sa, int item, int lane) {
  const R2LStaticArgs& a = sa.s;
  constexpr int HALO = DEB ? 2 : 1, NR = 2 * HALO + 1;
  const int seg = item % sa.nseg, r = item / sa.nseg;
  const int band = r % sa.nband, b = r / sa.nband;
  const int x0 = seg * 256 + 4 * lane;
  if (x0 >= a.W) return;
  const int y0 = band * sa.band_h;
  const int y1 = (y0 + sa.band_h < a.H) ? y0 + sa.band_h : a.H;
  const bool le = x0 == 0, re = x0 + 4 >= a.W;
  const size_t plane = (size_t)a.H * a.W;
  const size_t img = (size_t)b * plane;  // element offset of image b
  float* outb = a.out + (size_t)b * 3 * plane;
  double win[NR][8];
  int par[NR];  // source row parity of each window slot
  // warm-up: slots 0..NR-2 hold rows y0-HALO .. y0+HALO-1
  R2LRowStage st;
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < NR - 1; ++i) {
    const int ys = r2l_symmetric(y0 - HALO + i, a.H);
    r2l_stream_fetch_row<U16>(a, img, ys, x0, le, re, st);
    r2l_stream_convert_row<U16>(a, st, le, re, win[i]);
    par[i] = ys & 1;
  }
  // software pipeline, two rows deep: the rows needed by the next TWO output rows are in flight while
  // this one is computed (st = row y+HALO, st2 = row y+HALO+1)
  R2LRowStage st2;
  r2l_stream_fetch_row<U16>(a, img, r2l_symmetric(y0 + HALO, a.H), x0, le, re, st);
  r2l_stream_fetch_row<U16>(a, img, r2l_symmetric(y0 + HALO + 1, a.H), x0, le, re, st2);
  for (int yb = y0; yb < y1; yb += NR) {
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < NR; ++k) {  // unrolled by the window depth: slot indices are compile-time
      const int y = yb + k;
      if (y < y1) {
        // newest row y + HALO (fetched one iteration ago) goes to slot (k + NR - 1) % NR
        r2l_stream_convert_row<U16>(a, st, le, re, win[(k + NR - 1) % NR]);
        par[(k + NR - 1) % NR] = st.ys & 1;
        st = st2;
        if (y + 2 < y1) r2l_stream_fetch_row<U16>(a, img, r2l_symmetric(y + 2 + HALO, a.H), x0, le, re, st2);
        double d[4][3];
        if (DEB == 0) {
          // interior rows: the tap rows have the checkerboard parities (compile-time after the uniform
          // branch); the first / last image row sees a mirrored row and takes the runtime parities
          if (y > 0 && y < a.H - 1 && (y & 1)) {
            const int tpy[3] = {0, 1, 0};
            r2l_stream_bilinear_row(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], tpy, le, re, d);
          } else if (y > 0 && y < a.H - 1) {
            const int tpy[3] = {1, 0, 1};
            r2l_stream_bilinear_row(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], tpy, le, re, d);
          } else {
            const int tpy[3] = {par[k % NR], par[(k + 1) % NR], par[(k + 2) % NR]};
            r2l_stream_bilinear_row(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], tpy, le, re, d);
          }
        } else {
          if (y & 1)
            r2l_stream_malvar_row<1>(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], win[(k + 3) % NR],
                                     win[(k + 4) % NR], d);
          else
            r2l_stream_malvar_row<0>(win[k % NR], win[(k + 1) % NR], win[(k + 2) % NR], win[(k + 3) % NR],
                                     win[(k + 4) % NR], d);
        }
        if (LUMA && sa.luma_out)
          r2l_stream_luma_out_row(a, d, sa.luma_out, img + (size_t)y * a.W + x0);
        else if (LUMA)
          r2l_stream_luma_in_row(a, d, sa.luma_in + img, outb, plane, (size_t)y * a.W + x0);
        else
          r2l_stream_finish_row(a, d, outb, plane, (size_t)y * a.W + x0);
      }
    }
  }
}

#define R2L_STREAM_NT 256  // 4 independent wavefronts per workgroup
template <int DEB, bool U16, bool LUMA>
R2L_BLOCKFN void r2l_static_stream_block(const R2LStaticStreamArgs& sa, int bid, int nblk, float* lds) {
  (void)lds;
  (void)nblk;
  R2L_PHASE_BEGIN_N(R2L_STREAM_NT)
  const int item = bid * (R2L_STREAM_NT / 64) + (tid >> 6);  // one work item per wavefront
  if (item < sa.nitems) r2l_static_stream_item<DEB, U16, LUMA>(sa, item, tid & 63);
  R2L_PHASE_END
}
