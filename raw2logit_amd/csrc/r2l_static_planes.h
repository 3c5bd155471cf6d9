// r2l_static_planes.h -- luma-plane filters of the multi-pass static chains (numpy semantics, float64).
//
// The static alternates the reference offers on the Y channel (pipeline_numpy.py:110-122, :180-209):
//   op 1  sharpening_filter   convolve2d(Y, [[0,-1,0],[-1,5,-1],[0,-1,0]], 'same', fill 0)          :180-191
//   op 2  gaussian_denoising  scipy.ndimage.gaussian_filter(Y, 0.5): 5 taps per axis, 'reflect'        :203-209
//   op 3  median_denoising    scipy.ndimage.median_filter(Y, 3): 3x3, 'reflect'                        :194-200
//   op 5  median_denoising    ... with median_kernel_size=5 (pipeline_numpy.py:119): 5x5, 'reflect' -- plane passes only
//   op 4  unsharp_masking     skimage.filters.unsharp_mask(Y, radius 1, amount 1, multichannel=True) as the
//                             reference calls it (:117, :170-177): every COLUMN is a "channel", so the Gaussian
//                             (sigma 1, 9 taps, 'reflect') runs along the rows only; Y + (Y - blurred)   [unpinned:
//                             scikit-image is absent, see oracle/isp_oracle.py unsharp_mask]
// One lane per pixel pair, neighbours straight from global memory (the plane is L2-resident per row band);
// borders by index arithmetic (scipy 'reflect' = symmetric: b a | a b).  These chains are the long tail of
// `--sp_*` sweeps, not the throughput configuration (that one is fused: r2l_static_stream.h).
#pragma once
#include "r2l_static_kernels.h"

struct R2LPlaneArgs {
  const double* src;  // (B,H,W)
  double* dst;
  int B, H, W, op;
  double gk[5];
  double uk[5];      // op 4: half of the 9-tap Gaussian (sigma 1): uk[0] = centre tap ... uk[4] = offset +-4
  double amount;     // op 4
};
R2L_HD void r2l_cswap(double& a, double& b) {
  const double lo = fmin(a, b), hi = fmax(a, b);
  a = lo;
  b = hi;
}
// median of 9 (19 compare-exchanges)
R2L_HD double r2l_median9(double p[9]) {
  r2l_cswap(p[1], p[2]); r2l_cswap(p[4], p[5]); r2l_cswap(p[7], p[8]);
  r2l_cswap(p[0], p[1]); r2l_cswap(p[3], p[4]); r2l_cswap(p[6], p[7]);
  r2l_cswap(p[1], p[2]); r2l_cswap(p[4], p[5]); r2l_cswap(p[7], p[8]);
  r2l_cswap(p[0], p[3]); r2l_cswap(p[5], p[8]); r2l_cswap(p[4], p[7]);
  r2l_cswap(p[3], p[6]); r2l_cswap(p[1], p[4]); r2l_cswap(p[2], p[5]);
  r2l_cswap(p[4], p[7]); r2l_cswap(p[4], p[2]); r2l_cswap(p[6], p[4]);
  r2l_cswap(p[4], p[2]);
  return p[4];
}
R2L_HD double r2l_plane_px(const R2LPlaneArgs& a, const double* img, int y, int x) {
  if (a.op == 1) {  // zero-filled 5-point stencil (the kernel is symmetric: convolution == correlation)
    const double c = img[(size_t)y * a.W + x];
    const double n = y > 0 ? img[(size_t)(y - 1) * a.W + x] : 0.0;
    const double s = y < a.H - 1 ? img[(size_t)(y + 1) * a.W + x] : 0.0;
    const double w = x > 0 ? img[(size_t)y * a.W + x - 1] : 0.0;
    const double e = x < a.W - 1 ? img[(size_t)y * a.W + x + 1] : 0.0;
    return 5.0 * c - n - s - w - e;
  }
  if (a.op == 2) {  // axis 0 first, then axis 1, each as scipy's symmetric correlate1d
    double t[5];
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 5; ++j) {
      const int xx = r2l_symmetric(x + j - 2, a.W);
      const double c = img[(size_t)y * a.W + xx];
      const double u1 = img[(size_t)r2l_symmetric(y - 1, a.H) * a.W + xx], d1 = img[(size_t)r2l_symmetric(y + 1, a.H) * a.W + xx];
      const double u2 = img[(size_t)r2l_symmetric(y - 2, a.H) * a.W + xx], d2 = img[(size_t)r2l_symmetric(y + 2, a.H) * a.W + xx];
      t[j] = a.gk[2] * c + (u1 + d1) * a.gk[1] + (u2 + d2) * a.gk[0];
    }
    return a.gk[2] * t[2] + (t[1] + t[3]) * a.gk[1] + (t[0] + t[4]) * a.gk[0];
  }
  if (a.op == 4) {
    const double c = img[(size_t)y * a.W + x];
    double blurred = a.uk[0] * c;  // scipy correlate1d, symmetric kernel: w0*c + sum_k (up_k + down_k) * w_k
    R2L_PRAGMA_UNROLL
    for (int k = 1; k <= 4; ++k)
      blurred += (img[(size_t)r2l_symmetric(y - k, a.H) * a.W + x] + img[(size_t)r2l_symmetric(y + k, a.H) * a.W + x]) * a.uk[k];
    return c + (c - blurred) * a.amount;
  }
  if (a.op == 5) {  // scipy.ndimage.median_filter(Y, 5): the 13th smallest of the 5x5 window, 'reflect' (= symmetric) borders
    double q[25];
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 5; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 5; ++j)
      q[i * 5 + j] = img[(size_t)r2l_symmetric(y + i - 2, a.H) * a.W + r2l_symmetric(x + j - 2, a.W)];
    // selection network: pass k leaves the minimum of q[k ..] in q[k]; 13 passes, 234 compare-exchanges, every index a
    // compile-time constant (the window stays in registers)
    R2L_PRAGMA_UNROLL
    for (int k = 0; k <= 12; ++k)
      R2L_PRAGMA_UNROLL
    for (int j = k + 1; j < 25; ++j) r2l_cswap(q[k], q[j]);
    return q[12];
  }
  double p[9];
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 3; ++j)
    p[i * 3 + j] = img[(size_t)r2l_symmetric(y + i - 1, a.H) * a.W + r2l_symmetric(x + j - 1, a.W)];
  return r2l_median9(p);
}
// Interior fast path: the two adjacent pixels (x, x+1) of a lane share their window, which is fetched with
// aligned 16-byte loads (x is even) and no border arithmetic; lanes whose window leaves the image take the
// per-pixel path above.  `r` = window radius of the op.
R2L_HD r2l_d2 r2l_ld2(const double* p) {
#ifdef R2L_EMUL
  r2l_d2 v;
  v.x = p[0];
  v.y = p[1];
  return v;
#else
  const double2 t = *(const double2*)p;
  r2l_d2 v;
  v.x = t.x;
  v.y = t.y;
  return v;
#endif
}
R2L_HD void r2l_plane_px2(const R2LPlaneArgs& a, const double* img, int y, int x, double& o0, double& o1) {
  const size_t W = (size_t)a.W;
  const double* c = img + (size_t)y * W + x;
  if (a.op == 1) {
    const r2l_d2 l = r2l_ld2(c - 2), m = r2l_ld2(c), rr = r2l_ld2(c + 2), n = r2l_ld2(c - W), s = r2l_ld2(c + W);
    o0 = 5.0 * m.x - n.x - s.x - l.y - m.y;
    o1 = 5.0 * m.y - n.y - s.y - m.x - rr.x;
    return;
  }
  if (a.op == 2) {
    double t[6];  // vertical pass at columns x-2 .. x+3
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      const double* q = c + 2 * j - 2;
      const r2l_d2 m = r2l_ld2(q), u1 = r2l_ld2(q - W), d1 = r2l_ld2(q + W), u2 = r2l_ld2(q - 2 * W), d2 = r2l_ld2(q + 2 * W);
      t[2 * j] = a.gk[2] * m.x + (u1.x + d1.x) * a.gk[1] + (u2.x + d2.x) * a.gk[0];
      t[2 * j + 1] = a.gk[2] * m.y + (u1.y + d1.y) * a.gk[1] + (u2.y + d2.y) * a.gk[0];
    }
    o0 = a.gk[2] * t[2] + (t[1] + t[3]) * a.gk[1] + (t[0] + t[4]) * a.gk[0];
    o1 = a.gk[2] * t[3] + (t[2] + t[4]) * a.gk[1] + (t[1] + t[5]) * a.gk[0];
    return;
  }
  if (a.op == 4) {
    const r2l_d2 m = r2l_ld2(c);
    double b0 = a.uk[0] * m.x, b1 = a.uk[0] * m.y;
    R2L_PRAGMA_UNROLL
    for (int k = 1; k <= 4; ++k) {
      const r2l_d2 u = r2l_ld2(c - (size_t)k * W), d = r2l_ld2(c + (size_t)k * W);
      b0 += (u.x + d.x) * a.uk[k];
      b1 += (u.y + d.y) * a.uk[k];
    }
    o0 = m.x + (m.x - b0) * a.amount;
    o1 = m.y + (m.y - b1) * a.amount;
    return;
  }
  double p0[9], p1[9];
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i) {
    const double* q = c + ((size_t)i * W) - W;
    const r2l_d2 l = r2l_ld2(q - 2), m = r2l_ld2(q), rr = r2l_ld2(q + 2);
    p0[i * 3] = l.y, p0[i * 3 + 1] = m.x, p0[i * 3 + 2] = m.y;
    p1[i * 3] = m.x, p1[i * 3 + 1] = m.y, p1[i * 3 + 2] = rr.x;
  }
  o0 = r2l_median9(p0);
  o1 = r2l_median9(p1);
}
R2L_BLOCKFN void r2l_plane_filter_block(const R2LPlaneArgs& a, int bid, int nblk, float* lds) {
  (void)lds;
  const size_t hw = (size_t)a.H * a.W, n2 = (size_t)a.B * hw / 2;  // W is even
  const int r = (a.op == 4) ? 4 : (a.op == 2 ? 2 : 1), rx = (a.op == 4) ? 0 : 2;  // window reach in y / loads in x
  R2L_PHASE_BEGIN
  for (size_t i2 = (size_t)bid * R2L_NT + tid; i2 < n2; i2 += (size_t)nblk * R2L_NT) {
    const size_t e = i2 * 2;
    const size_t b = e / hw, rem = e - b * hw;
    const int y = (int)(rem / a.W), x = (int)(rem - (size_t)y * a.W);
    const double* img = a.src + b * hw;
    double o0, o1;
    if (a.op != 5 && y >= r && y + r < a.H && x >= rx && x + 1 + rx < a.W) {   // (op 5: the per-pixel form everywhere)
      r2l_plane_px2(a, img, y, x, o0, o1);
    } else {
      o0 = r2l_plane_px(a, img, y, x);
      o1 = r2l_plane_px(a, img, y, x + 1);
    }
    a.dst[e] = o0;
    a.dst[e + 1] = o1;
  }
  R2L_PHASE_END
}

// ---- fft_denoising (pipeline_numpy.py:212-238 as processing() calls it, :121-122): per image row and colour channel
// the spectrum along the columns is zeroed for k in [int(W * keep), int(W * (1 - keep))) and the REAL PART of the
// inverse transform is kept.  The input is real, so its spectrum F is Hermitian and Re(ifft(m F)) = ifft(h F) with
// the symmetrised mask h_k = (m_k + m_{W-k}) / 2 (values 0, 1/2, 1): a real-to-complex transform, this mask on the
// W/2 + 1 stored bins (with the 1/W of the unnormalised inverse), a complex-to-real transform.
struct R2LSpecMaskArgs {
  double* spec;  // [rows][W/2 + 1] interleaved complex
  size_t rows;
  int W, cut0, cut1;  // bins cut0 <= k < cut1 are zeroed
};
R2L_HD double r2l_fft_mask(int k, int W, int cut0, int cut1) {
  const int kk = (W - k) % W;
  const double m0 = (k >= cut0 && k < cut1) ? 0.0 : 1.0, m1 = (kk >= cut0 && kk < cut1) ? 0.0 : 1.0;
  return 0.5 * (m0 + m1);
}
R2L_BLOCKFN void r2l_spec_mask_block(const R2LSpecMaskArgs& a, int bid, int nblk, float* lds) {
  (void)lds;
  const int nb = a.W / 2 + 1;
  const size_t n = a.rows * (size_t)nb;
  R2L_PHASE_BEGIN
  for (size_t i = (size_t)bid * R2L_NT + tid; i < n; i += (size_t)nblk * R2L_NT) {
    const int k = (int)(i % (size_t)nb);
    const double h = r2l_fft_mask(k, a.W, a.cut0, a.cut1) / (double)a.W;
    a.spec[2 * i] *= h;
    a.spec[2 * i + 1] *= h;
  }
  R2L_PHASE_END
}
// float64 planes (B,3,H,W) -> clip, gamma (+ the T.Normalize epilogue) -> float32 output; 4 pixels per lane
struct R2LStaticFinishArgs {
  R2LStaticArgs s;
  const double* lin;
};
R2L_BLOCKFN void r2l_static_finish_block(const R2LStaticFinishArgs& fa, int bid, int nblk, float* lds) {
  (void)lds;
  const R2LStaticArgs& a = fa.s;
  const size_t hw4 = (size_t)a.H * a.W / 4, n = (size_t)a.B * hw4;  // W % 4 == 0
  R2L_PHASE_BEGIN
  for (size_t i = (size_t)bid * R2L_NT + tid; i < n; i += (size_t)nblk * R2L_NT) {
    const size_t b = i / hw4, off = (i - b * hw4) * 4;
    float x[3][4];
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k)
      R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c)
      x[k][c] = r2l_clip_gamma(fa.lin[((size_t)b * 3 + k) * hw4 * 4 + off + c], a.inv_gamma);
    r2l_static_normalize<4>(a, x);
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      r2l_f4 st;
      st.x = x[k][0];
      st.y = x[k][1];
      st.z = x[k][2];
      st.w = x[k][3];
      *(r2l_f4*)(a.out + ((size_t)b * 3 + k) * hw4 * 4 + off) = st;
    }
  }
  R2L_PHASE_END
}
#ifdef R2L_EMUL
// host emulation of the two transforms + mask (the device build calls rocFFT): plain O(W^2) DFT per row
static inline void r2l_fft_lowpass_rows_host(double* rows, size_t nrows, int W, int cut0, int cut1) {
  std::vector<double> re(W), im(W), out(W);
  const double w0 = 2.0 * 3.14159265358979323846 / (double)W;
  for (size_t r = 0; r < nrows; ++r) {
    double* x = rows + r * (size_t)W;
    for (int k = 0; k < W; ++k) {
      double sr = 0.0, si = 0.0;
      for (int j = 0; j < W; ++j) {
        const double ang = w0 * (double)(((long long)k * j) % W);
        sr += x[j] * cos(ang);
        si -= x[j] * sin(ang);
      }
      const bool cut = k >= cut0 && k < cut1;
      re[k] = cut ? 0.0 : sr;
      im[k] = cut ? 0.0 : si;
    }
    for (int j = 0; j < W; ++j) {
      double s = 0.0;
      for (int k = 0; k < W; ++k) {
        const double ang = w0 * (double)(((long long)k * j) % W);
        s += re[k] * cos(ang) - im[k] * sin(ang);
      }
      out[j] = s / (double)W;
    }
    for (int j = 0; j < W; ++j) x[j] = out[j];
  }
}
#endif

#ifndef R2L_EMUL
#pragma clang fp contract(fast)  // end of the static chains (see r2l_static_kernels.h)
#endif
