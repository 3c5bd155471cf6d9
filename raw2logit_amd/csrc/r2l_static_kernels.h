// r2l_static_kernels.h -- the static ("numpy semantics") pipeline, batched on the device.
//
// Replaces processing() (processing/pipeline_numpy.py:70-141) applied per image by
// RawProcessingPipeline.__call__ (:55-67):
//   remove_blacklv (:152-158) -> demosaicing_CFA_Bayer_bilinear | _Malvar2004 (:92-95; colour-demosaicing
//   0.1.6: scipy.ndimage.convolve, mode 'reflect' == symmetric) -> white balance (:161-162) -> colour
//   matrix (:165-167) -> [sharpening_filter (:180-191): rgb2yuv, convolve2d(Y, K, 'same', fill 0),
//   yuv2rgb] -> [gaussian_denoising (:203-209): rgb2yuv, ndimage.gaussian_filter(Y, 0.5) (5 taps,
//   symmetric), yuv2rgb] -> clip[0,1] (:138) -> x ** (1/gamma) (:241-244).
// remove_blacklv works in place in the dtype of the frame it is handed -- float32 for every tif / png tile
// (utils/dataset_utils.py:18-26, dataset.py:86-87), float64 for a DNG's uint16 / (2**bits - 1) -- and from
// the demosaic on everything is float64.  x**(1/gamma) has an unbounded derivative at 0, so float32 round-off
// in the linear part would break the 1e-5 parity bar near black.  The kernels therefore subtract the black
// level in the frames' own arithmetic (float32 frames and 16-bit containers: float32-rounded black level,
// float32 subtraction; float64 frames: float64) and run the rest of the linear part in float64 as well (MI355X: half the
// f32 vector rate, still far below the HBM time); only log2/exp2 run in float32, whose RELATIVE error
// is what matters for the power law.
//
// SHORT chain (no sharpening / denoising; BASELINE config C3): one LDS plane (raw tile + halo), one
// read of raw (4 B/px), one write of RGB (12 B/px).
// FULL chain: two more float64 LDS planes for the luma (zero-extended for the sharpen, symmetric-
// extended for the blur), same structure as the parametrized kernel.
#pragma once
#include "r2l_param_kernels.h"

// The float64 arithmetic of the static chains is written out with explicit fma() where a fused operation is
// wanted and compiled WITHOUT automatic contraction (until the end of r2l_static_planes.h): left to the compiler,
// the float32 and the 16-bit instantiations of one kernel can fuse differently, and a value that differs in its
// last float64 bit can round to a different float32 next to the clip at 0 (found by tests/fuzz_more.py: one pixel
// in 10^8).  With this, frames fed as 16-bit containers give bit-identical results to host-normalised ones.
#ifndef R2L_EMUL
#pragma clang fp contract(off)
#endif

typedef R2LGeom<64, 64> GStatic;
#define R2L_STATIC_LDS_FLOATS (2 * GStatic::PAD + 5 * GStatic::PLANE + 160)  // (+ the argument block)
#define R2L_STATIC_SHORT_LDS_FLOATS (2 * GStatic::PAD + GStatic::PLANE)

struct R2LStaticArgs {
  R2LRaw raw;
  float* out;
  int B, H, W;
  int debayer, full, sharpen, denoise;
  double bl[4];   // black level of float64 frames
  float blf[4];   // ... of float32 frames / 16-bit containers: rounded to float32, subtracted in float32
  double wbccm[9];  // colour_matrix * diag(white_balance): RGB_cc = wbccm * RGB_demosaic
  double T[9];      // yuv_from_rgb * wbccm
  double M2[9];     // rgb_from_yuv = inv(yuv_from_rgb)
  double ksharp[9];
  double gk[5];  // gaussian taps, sigma 0.5, radius 2
  double uk[5];  // unsharp_masking: half of the 9-tap Gaussian (sigma 1): uk[0] = centre tap ... uk[4] = offset +-4
  double amount;  // unsharp_masking
  float inv_gamma;
  int normalize;            // epilogue (x - nmean[c]) / nstd[c]: the T.Normalize(mean, std) of train.py:157-171 (2: see r2l_static_normalize)
  float nmean[3], nstd[3];
  float nrcp[3];            // RN(1 / nstd[c])
};

// skimage.color yuv_from_rgb (scikit-image 0.18.1); the reference's own copy is pipeline_torch.py:21-23
static const double R2L_YUV_FROM_RGB[9] = {0.299,       0.587,       0.114,       -0.14714119, -0.28886916,
                                           0.43601035,  0.61497538,  -0.51496512, -0.10001026};
static inline void r2l_inv3(const double* m, double* o) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const double A = e * i - f * h, Bc = -(d * i - f * g), C = d * h - e * g;
  const double det = a * A + b * Bc + c * C;
  o[0] = A / det;
  o[1] = -(b * i - c * h) / det;
  o[2] = (b * f - c * e) / det;
  o[3] = Bc / det;
  o[4] = (a * i - c * g) / det;
  o[5] = -(a * f - c * d) / det;
  o[6] = C / det;
  o[7] = -(a * h - b * g) / det;
  o[8] = (a * e - b * d) / det;
}

// processing()'s numeric arguments (pipeline_numpy.py:70-73, used at :117-122); the reference's defaults
struct R2LStaticOpts {
  double sharp_radius = 1.0, sharp_amount = 1.0, gaussian_sigma = 0.5, fft_fraction = 0.3;
  int median_kernel_size = 3;
};
// scipy.ndimage.gaussian_filter's window radius for a sigma: int(truncate * sigma + 0.5), truncate = 4.0
static inline int r2l_scipy_radius(double sigma) { return (int)(4.0 * sigma + 0.5); }
// what the kernels' windows can hold: a 5-tap Gaussian for gaussian_denoising (radius <= 2: sigma < 0.625), a 9-tap one behind
// unsharp_masking (radius <= 4: sharp_radius < 1.125), the 3x3 median (5x5: as a luma-plane pass); 0 <= fft_fraction <= 0.5.  NULL: fine.
static inline const char* r2l_static_opts_problem(const R2LStaticOpts& o, int sharpening, int denoising) {
  if (denoising == R2L_DENOISE_GAUSSIAN && !(o.gaussian_sigma > 0.0 && r2l_scipy_radius(o.gaussian_sigma) <= 2))
    return "gaussian_sigma must be in (0, 0.625): scipy's window radius int(4 sigma + 0.5) may not exceed the kernels' 2";
  if (sharpening == R2L_SHARPEN_UNSHARP && !(o.sharp_radius > 0.0 && r2l_scipy_radius(o.sharp_radius) <= 4))
    return "sharp_radius must be in (0, 1.125): scipy's window radius int(4 radius + 0.5) may not exceed the kernels' 4";
  if (sharpening == R2L_SHARPEN_UNSHARP && !(o.sharp_amount == o.sharp_amount))
    return "sharp_amount is not a number";
  if (denoising == R2L_DENOISE_MEDIAN && o.median_kernel_size != 3 && o.median_kernel_size != 5)
    return "median_kernel_size must be 3 (fused kernels) or 5 (luma-plane passes)";
  if (denoising == R2L_DENOISE_FFT && !(o.fft_fraction >= 0.0 && o.fft_fraction <= 0.5))
    return "fft_fraction must be in [0, 0.5] (pipeline_numpy.py:212-214)";
  return nullptr;
}

static inline void r2l_static_setup(R2LStaticArgs& a, const R2LRaw& raw, float* out, int B, int H, int W,
                                    const double* cam, int debayer, int sharpening, int denoising,
                                    double gamma, const float* mean_std = nullptr,
                                    const R2LStaticOpts& opt = R2LStaticOpts()) {
  a.normalize = mean_std != nullptr;
  for (int k = 0; k < 3; ++k) {
    a.nmean[k] = mean_std ? mean_std[k] : 0.f;
    a.nstd[k] = mean_std ? mean_std[3 + k] : 1.f;
    a.nrcp[k] = 1.0f / a.nstd[k];
    unsigned bits;
    memcpy(&bits, &a.nstd[k], 4);
    if (mean_std && (bits & 0x7fffffu) == 0x7fffffu) a.normalize = 2;
  }
  // skimage.color yuv_from_rgb (scikit-image 0.18.1); the reference's own copy is
  // pipeline_torch.py:21-23
  const double* M1 = R2L_YUV_FROM_RGB;
  a.raw = raw;
  a.out = out;
  a.B = B;
  a.H = H;
  a.W = W;
  a.debayer = debayer;
  a.sharpen = sharpening;
  a.denoise = denoising;
  a.full = (sharpening != R2L_SHARPEN_NONE) || (denoising != R2L_DENOISE_NONE);
  for (int s = 0; s < 4; ++s) a.bl[s] = cam[s];
  for (int s = 0; s < 4; ++s) a.blf[s] = (float)cam[s];
  for (int k = 0; k < 3; ++k)
    for (int c = 0; c < 3; ++c) a.wbccm[k * 3 + c] = cam[7 + k * 3 + c] * cam[4 + c];
  for (int k = 0; k < 3; ++k)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int j = 0; j < 3; ++j) s += M1[k * 3 + j] * a.wbccm[j * 3 + c];
      a.T[k * 3 + c] = s;
    }
  r2l_inv3(M1, a.M2);
  static const double KS[9] = {0, -1, 0, -1, 5, -1, 0, -1, 0};  // pipeline_numpy.py:180
  static const double ID[9] = {0, 0, 0, 0, 1, 0, 0, 0, 0};
  for (int i = 0; i < 9; ++i) a.ksharp[i] = (sharpening == R2L_SHARPEN_FILTER) ? KS[i] : ID[i];
  if (denoising == R2L_DENOISE_GAUSSIAN) {
    // scipy.ndimage.gaussian_filter(Y, sigma) (pipeline_numpy.py:203-209; default 0.5: radius 2): _gaussian_kernel1d over
    // radius = int(4 sigma + 0.5) <= 2 taps each side, normalised over THAT window; taps beyond it are zero in the 5-tap frame
    const int rad = r2l_scipy_radius(opt.gaussian_sigma);
    const double sg = opt.gaussian_sigma;
    double w[5], s = 0;
    for (int i = 0; i < 5; ++i) {
      const double x = i - 2;
      w[i] = (x < -rad || x > rad) ? 0.0 : exp(-0.5 / (sg * sg) * x * x);
      s += w[i];
    }
    for (int i = 0; i < 5; ++i) a.gk[i] = w[i] / s;
  } else {
    for (int i = 0; i < 5; ++i) a.gk[i] = (i == 2) ? 1.0 : 0.0;
  }
  {  // skimage unsharp_mask(Y, radius, amount): scipy _gaussian_kernel1d(sigma = radius, radius = int(4 sigma + 0.5) <= 4),
     // normalised over its own window (defaults: sigma 1, 9 taps, amount 1)
    const double sg = sharpening == R2L_SHARPEN_UNSHARP ? opt.sharp_radius : 1.0;  // (unused by any other chain)
    const int rad = r2l_scipy_radius(sg);
    double w[9], sum = 0;
    for (int k = -4; k <= 4; ++k) sum += (w[k + 4] = (k < -rad || k > rad) ? 0.0 : exp(-0.5 / (sg * sg) * k * k));
    for (int k = 0; k <= 4; ++k) a.uk[k] = w[4 + k] / sum;
    a.amount = sharpening == R2L_SHARPEN_UNSHARP ? opt.sharp_amount : 1.0;
  }
  a.inv_gamma = (float)(1.0 / gamma);
}

// ---- phase A: raw tile + halo, symmetric ('reflect' of scipy) coordinates, plain float32 -----------
template <class G>
R2L_HD void r2l_load_raw_sym(int tid, float* V, const R2LRaw& raw, size_t img0, int oy, int ox, int H, int W) {
  constexpr int CPR = G::FW / 4;
  const bool vec_ok = (W & 3) == 0;
  for (int ci = tid; ci < CPR * G::FH; ci += R2L_NT) {
    const int fy = ci / CPR, cx = ci - fy * CPR;
    const int gy = r2l_symmetric(oy - 4 + fy, H);
    const int gx0 = ox - 4 + 4 * cx;
    const size_t row = img0 + (size_t)gy * W;
    r2l_f4 v;
    if (vec_ok && gx0 >= 0 && gx0 + 3 < W) {
      v = r2l_raw_vec4(raw, row + gx0);
    } else {
      v.x = r2l_raw_elem(raw, row + r2l_symmetric(gx0, W));
      v.y = r2l_raw_elem(raw, row + r2l_symmetric(gx0 + 1, W));
      v.z = r2l_raw_elem(raw, row + r2l_symmetric(gx0 + 2, W));
      v.w = r2l_raw_elem(raw, row + r2l_symmetric(gx0 + 3, W));
    }
    *(r2l_f4*)(V + fy * G::FS + 4 * cx) = v;
  }
}

// the same in two halves for the software pipeline of the full chain: the next tile's frame is fetched into
// registers at the start of the pixel phase and written to LDS after it (16-bit containers travel undecoded in
// .x/.y when the chunk was fetched as a vector)
template <class G>
struct R2LStaticPre {
  static constexpr int NIT = ((G::FW / 4) * G::FH + R2L_NT - 1) / R2L_NT;
  r2l_f4 v[NIT];
};
template <class G>
R2L_HD void r2l_fetch_raw_sym(int tid, const R2LRaw& raw, size_t img0, int oy, int ox, int H, int W,
                              R2LStaticPre<G>& pf) {
  constexpr int CPR = G::FW / 4;
  const bool vec_ok = (W & 3) == 0;
  R2L_PRAGMA_UNROLL
  for (int it = 0; it < R2LStaticPre<G>::NIT; ++it) {
    const int ci = tid + it * R2L_NT;
    r2l_f4 v;
    v.x = v.y = v.z = v.w = 0.f;
    if (ci < CPR * G::FH) {
      const int fy = ci / CPR, cx = ci - fy * CPR;
      const int gy = r2l_symmetric(oy - 4 + fy, H);
      const int gx0 = ox - 4 + 4 * cx;
      const size_t row = img0 + (size_t)gy * W;
      if (vec_ok && gx0 >= 0 && gx0 + 3 < W) {
        if (raw.u16) {
          const r2l_f2 b = *(const r2l_f2*)(raw.u16 + row + gx0);
          v.x = b.x;
          v.y = b.y;
        } else {
          v = *(const r2l_f4*)(raw.f32 + row + gx0);
        }
      } else {
        v.x = r2l_raw_elem(raw, row + r2l_symmetric(gx0, W));
        v.y = r2l_raw_elem(raw, row + r2l_symmetric(gx0 + 1, W));
        v.z = r2l_raw_elem(raw, row + r2l_symmetric(gx0 + 2, W));
        v.w = r2l_raw_elem(raw, row + r2l_symmetric(gx0 + 3, W));
      }
    }
    pf.v[it] = v;
  }
}
template <class G>
R2L_HD void r2l_store_raw_sym(int tid, float* V, const R2LRaw& raw, int ox, int W, const R2LStaticPre<G>& pf) {
  constexpr int CPR = G::FW / 4;
  const bool vec_ok = (W & 3) == 0;
  R2L_PRAGMA_UNROLL
  for (int it = 0; it < R2LStaticPre<G>::NIT; ++it) {
    const int ci = tid + it * R2L_NT;
    if (ci < CPR * G::FH) {
      const int fy = ci / CPR, cx = ci - fy * CPR;
      const int gx0 = ox - 4 + 4 * cx;
      r2l_f4 v = pf.v[it];
      if (raw.u16 && vec_ok && gx0 >= 0 && gx0 + 3 < W) {
        const unsigned lo = r2l_f2u(v.x), hi = r2l_f2u(v.y);
        v.x = r2l_raw_decode(lo & 0xffffu, raw);
        v.y = r2l_raw_decode(lo >> 16, raw);
        v.z = r2l_raw_decode(hi & 0xffffu, raw);
        v.w = r2l_raw_decode(hi >> 16, raw);
      }
      *(r2l_f4*)(V + fy * G::FS + 4 * cx) = v;
    }
  }
}

// ---- demosaicing of one pixel -----------------------------------------------------------------------
// Bilinear: three MASKED planes convolved with H_RB / H_G / H_RB.  n = black-level-corrected 3x3
// neighbourhood, tpy/tpx = row / column parities of the taps' (symmetric-clamped) coordinates: for taps
// inside the image that is the usual checkerboard; a tap mirrored back from outside keeps the site of
// the pixel it was mirrored from.
R2L_HD void r2l_bilinear_px(const double n[3][3], const int tpy[3], const int tpx[3], double d[3]) {
  double dr = 0.0, dg = 0.0, db = 0.0;
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 3; ++j) {
    const int ch = tpy[i] + tpx[j];
    const double krb = ((i == 1) ? 2.0 : 1.0) * ((j == 1) ? 2.0 : 1.0) * 0.25;
    const double t = krb * n[i][j];
    dr += (ch == 0) ? t : 0.0;
    db += (ch == 2) ? t : 0.0;
    if (i == 1 || j == 1) {
      const double kg = ((i == 1 && j == 1) ? 4.0 : 1.0) * 0.25;
      dg += (ch == 1) ? kg * n[i][j] : 0.0;
    }
  }
  d[0] = dr;
  d[1] = dg;
  d[2] = db;
}

// the same convolution where the 3x3 neighbourhood lies inside the image (plain checkerboard masks): the pixel
// itself, the mean of its 2 row / column neighbours, of its 4 edge or of its 4 corner neighbours; (py, px) =
// site parities of the pixel.  Differs from the masked form by float64 round-off of the summation order only.
R2L_HD void r2l_bilinear_interior_px(const double n[3][3], int py, int px, double d[3]) {
  const double vs = n[0][1] + n[2][1], hs = n[1][0] + n[1][2];
  if (py == px) {
    const double corners = ((n[0][0] + n[2][0]) + (n[0][2] + n[2][2])) * 0.25, edges = (vs + hs) * 0.25;
    d[0] = py ? corners : n[1][1];
    d[1] = edges;
    d[2] = py ? n[1][1] : corners;
  } else {  // G sites: (0,1) has R left/right and B above/below, (1,0) the other way round
    d[0] = (py == 0) ? hs * 0.5 : vs * 0.5;
    d[1] = n[1][1];
    d[2] = (py == 0) ? vs * 0.5 : hs * 0.5;
  }
}

// Malvar-He-Cutler 2004 (colour-demosaicing 0.1.6 coefficients), w = black-level-corrected 5x5
// neighbourhood of the UNMASKED mosaic, (py,px) = parity of the output pixel.
R2L_HD double r2l_malvar_gr_gb(const double w[5][5]) {
  return (4.0 * w[2][2] + 2.0 * (w[1][2] + w[3][2] + w[2][1] + w[2][3]) -
          (w[0][2] + w[4][2] + w[2][0] + w[2][4])) *
         0.125;
}
R2L_HD double r2l_malvar_rg_rb(const double w[5][5]) {  // Rg_RB_Bg_BR (horizontal neighbours x4)
  return (5.0 * w[2][2] + 4.0 * (w[2][1] + w[2][3]) - (w[1][1] + w[1][3] + w[3][1] + w[3][3]) -
          (w[2][0] + w[2][4]) + 0.5 * (w[0][2] + w[4][2])) *
         0.125;
}
R2L_HD double r2l_malvar_rg_br(const double w[5][5]) {  // its transpose (vertical neighbours x4)
  return (5.0 * w[2][2] + 4.0 * (w[1][2] + w[3][2]) - (w[1][1] + w[1][3] + w[3][1] + w[3][3]) -
          (w[0][2] + w[4][2]) + 0.5 * (w[2][0] + w[2][4])) *
         0.125;
}
R2L_HD double r2l_malvar_rb_bb(const double w[5][5]) {
  return (6.0 * w[2][2] + 2.0 * (w[1][1] + w[1][3] + w[3][1] + w[3][3]) -
          1.5 * (w[0][2] + w[4][2] + w[2][0] + w[2][4])) *
         0.125;
}
R2L_HD void r2l_malvar_px(const double w[5][5], int py, int px, double d[3]) {
  if (py == 0 && px == 0) {  // R site
    d[0] = w[2][2];
    d[1] = r2l_malvar_gr_gb(w);
    d[2] = r2l_malvar_rb_bb(w);
  } else if (py == 0 && px == 1) {  // G in a red row, blue column
    d[0] = r2l_malvar_rg_rb(w);
    d[1] = w[2][2];
    d[2] = r2l_malvar_rg_br(w);
  } else if (py == 1 && px == 0) {  // G in a blue row, red column
    d[0] = r2l_malvar_rg_br(w);
    d[1] = w[2][2];
    d[2] = r2l_malvar_rg_rb(w);
  } else {  // B site
    d[0] = r2l_malvar_rb_bb(w);
    d[1] = r2l_malvar_gr_gb(w);
    d[2] = w[2][2];
  }
}

// black-level-corrected float64 window of N x N raw values whose element (i,j) sits at global
// (gy0 - HALO + i, gx0 - HALO + j); rp/cp receive the site parities of the rows / columns.
// PAR0 = parity of gy0 and of gx0 (known at compile time: tiles and micro-tiles start on even pixels)
template <class G, int NR, int NC, int HALO, bool BORDER, int PAR0>
R2L_HD void r2l_static_window(const float* V, int fy0, int fx0, int gy0, int gx0,
                              const R2LStaticArgs& a, double w[NR][NC], int rp[NR], int cp[NC]) {
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < NR; ++i) rp[i] = BORDER ? (r2l_symmetric(gy0 - HALO + i, a.H) & 1) : ((PAR0 + i + HALO) & 1);
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < NC; ++i) cp[i] = BORDER ? (r2l_symmetric(gx0 - HALO + i, a.W) & 1) : ((PAR0 + i + HALO) & 1);
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < NR; ++i) {
    const float* r = V + (fy0 - HALO + i) * G::FS + fx0 - HALO;
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < NC; ++j) {
      float bl;  // the tile kernels serve float32 frames / 16-bit containers: float32 black level arithmetic
      if (BORDER) {
        const float b0 = rp[i] ? a.blf[2] : a.blf[0], b1 = rp[i] ? a.blf[3] : a.blf[1];
        bl = cp[j] ? b1 : b0;
      } else {
        bl = a.blf[(((PAR0 + i + HALO) & 1) << 1) | ((PAR0 + j + HALO) & 1)];
      }
      w[i][j] = (double)(r[j] - bl);
    }
  }
}

// Micro-tile of the pixel stage: NC columns x MR rows per lane.  The short tile chain keeps 4 x 2 (float4 stores);
// the full chain takes 2 x 4: its float64 luma window is then read with a lane stride of 16 bytes, i.e. free of
// LDS bank conflicts (4-column micro-tiles read 8-byte values 32 bytes apart: 8 of the 32 banks, measured as
// more than half of the pixel phase).
template <bool FULL>
struct R2LStaticMT {
  static constexpr int NC = FULL ? 2 : 4, MR = FULL ? 4 : 2;
};
// demosaiced RGB (float64) of the NC-wide x MR-tall micro-tile at frame (fy0, fx0) / global (gy0, gx0)
template <class G, bool BORDER, int NC, int MR>
R2L_HD void r2l_static_demosaic(const float* V, int fy0, int fx0, int gy0, int gx0, const R2LStaticArgs& a,
                                double d[MR][NC][3]) {
  if (a.debayer == R2L_DEBAYER_MALVAR2004) {
    double w[MR + 4][NC + 4];
    int rp[MR + 4], cp[NC + 4];
    r2l_static_window<G, MR + 4, NC + 4, 2, BORDER, 0>(V, fy0, fx0, gy0, gx0, a, w, rp, cp);
    R2L_PRAGMA_UNROLL
    for (int r = 0; r < MR; ++r)
      R2L_PRAGMA_UNROLL
    for (int c = 0; c < NC; ++c) {
      double n[5][5];
      R2L_PRAGMA_UNROLL
      for (int i = 0; i < 5; ++i)
        R2L_PRAGMA_UNROLL
      for (int j = 0; j < 5; ++j) n[i][j] = w[r + i][c + j];
      r2l_malvar_px(n, r & 1, c & 1, d[r][c]);
    }
  } else {
    double w[MR + 2][NC + 2];
    int rp[MR + 2], cp[NC + 2];
    r2l_static_window<G, MR + 2, NC + 2, 1, BORDER, 0>(V, fy0, fx0, gy0, gx0, a, w, rp, cp);
    R2L_PRAGMA_UNROLL
    for (int r = 0; r < MR; ++r)
      R2L_PRAGMA_UNROLL
    for (int c = 0; c < NC; ++c) {
      double n[3][3];
      int tpy[3], tpx[3];
      R2L_PRAGMA_UNROLL
      for (int i = 0; i < 3; ++i) {
        tpy[i] = rp[r + i];
        tpx[i] = cp[c + i];
        R2L_PRAGMA_UNROLL
        for (int j = 0; j < 3; ++j) n[i][j] = w[r + i][c + j];
      }
      if (BORDER)
        r2l_bilinear_px(n, tpy, tpx, d[r][c]);
      else
        r2l_bilinear_interior_px(n, r & 1, c & 1, d[r][c]);  // micro-tiles start on even pixels
    }
  }
}

R2L_HD bool r2l_static_touches_border(int gy0, int gx0, int halo, int H, int W, int nc, int mr) {
  return gy0 - halo < 0 || gx0 - halo < 0 || gy0 + mr - 1 + halo >= H || gx0 + nc - 1 + halo >= W;
}

// ---- full chain, phase B: luma on frame [1, F-1) quads -> Y (float64, zero outside the image) -------
template <class G, bool BORDER>
R2L_HD void r2l_static_y_quad(const float* V, double* Y, int fy, int fx, int gy, int gx,
                              const R2LStaticArgs& a) {
  // 2x2 quad at ODD frame coordinates (fy, fx): window of 4x4 raw values (halo 1)
  double w[4][4];
  int rp[4], cp[4];
  r2l_static_window<G, 4, 4, 1, BORDER, 1>(V, fy, fx, gy, gx, a, w, rp, cp);
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 2; ++r)
    R2L_PRAGMA_UNROLL
  for (int c = 0; c < 2; ++c) {
    double n[3][3], d[3];
    int tpy[3], tpx[3];
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i) {
      tpy[i] = rp[r + i];
      tpx[i] = cp[c + i];
      R2L_PRAGMA_UNROLL
      for (int j = 0; j < 3; ++j) n[i][j] = w[r + i][c + j];
    }
    if (BORDER)
      r2l_bilinear_px(n, tpy, tpx, d);
    else
      r2l_bilinear_interior_px(n, (1 + r) & 1, (1 + c) & 1, d);  // quads start on odd pixels
    const double y = fma(a.T[0], d[0], fma(a.T[1], d[1], a.T[2] * d[2]));
    const bool in = (unsigned)(gy + r) < (unsigned)a.H && (unsigned)(gx + c) < (unsigned)a.W;
    Y[(fy + r) * G::FS + fx + c] = in ? y : 0.0;
  }
}
template <class G>
R2L_HD void r2l_static_compute_y(int tid, const float* V, double* Y, const R2LStaticArgs& a, int oy,
                                 int ox) {
  constexpr int QW = (G::FW - 2) / 2, QH = (G::FH - 2) / 2;
  for (int q = tid; q < QW * QH; q += R2L_NT) {
    const int fy = 1 + 2 * (q / QW), fx = 1 + 2 * (q % QW);
    const int gy = oy - 4 + fy, gx = ox - 4 + fx;
    const bool border = gy - 1 < 0 || gx - 1 < 0 || gy + 2 >= a.H || gx + 2 >= a.W;
    if (border)
      r2l_static_y_quad<G, true>(V, Y, fy, fx, gy, gx, a);
    else
      r2l_static_y_quad<G, false>(V, Y, fy, fx, gy, gx, a);
  }
}
// phase C: YP = convolve2d(Y, K, 'same', fill 0) on frame [2, F-2)
template <class G>
R2L_HD void r2l_static_compute_yp(int tid, const double* Y, double* YP, const R2LStaticArgs& a) {
  // items of 2 columns x 4 rows with a 6 x 6 register window read two float64 at a time (frame column fx - 2 is
  // even, i.e. 16-byte aligned; lane stride 16 bytes: no bank conflicts): 18 LDS reads per 8 pixels instead of 72
  constexpr int NW = G::FW - 4, NH = G::FH - 4, CW = NW / 2, RH = NH / 4;
  static_assert(NW % 2 == 0 && NH % 4 == 0, "whole items");
  for (int it = tid; it < CW * RH; it += R2L_NT) {
    const int fy = 2 + 4 * (it / CW), fx = 2 + 2 * (it % CW);
    double w[6][6];  // rows fy-1..fy+4, columns fx-2..fx+3
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 6; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 6; j += 2) {
      const r2l_d2 q = r2l_lds_d2(Y + (fy - 1 + i) * G::FS + fx - 2 + j);
      w[i][j] = q.x;
      w[i][j + 1] = q.y;
    }
    R2L_PRAGMA_UNROLL
    for (int r = 0; r < 4; ++r) {
      double o[2];
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 2; ++c) {
        double sacc = 0.0;
        R2L_PRAGMA_UNROLL
        for (int p = 0; p < 3; ++p)
          R2L_PRAGMA_UNROLL
        for (int q = 0; q < 3; ++q)  // true convolution: flipped kernel (K is symmetric anyway)
          sacc = fma(a.ksharp[(2 - p) * 3 + (2 - q)], w[r + p][c + q + 1], sacc);
        o[c] = sacc;
      }
      r2l_d2 st;
      st.x = o[0];
      st.y = o[1];
      *(r2l_d2*)(YP + (fy + r) * G::FS + fx) = st;
    }
  }
}
// phase C2 (border tiles): symmetric extension of YP outside the image
template <class G>
R2L_HD void r2l_static_fill_yp(int tid, double* YP, int oy, int ox, int H, int W) {
  constexpr int NW = G::FW - 4, NH = G::FH - 4;
  for (int i = tid; i < NW * NH; i += R2L_NT) {
    const int fy = 2 + i / NW, fx = 2 + i % NW;
    const int gy = oy - 4 + fy, gx = ox - 4 + fx;
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) continue;
    const int my = r2l_symmetric(gy, H) - (oy - 4), mx = r2l_symmetric(gx, W) - (ox - 4);
    if (my >= 2 && my < G::FH - 2 && mx >= 2 && mx < G::FW - 2) YP[fy * G::FS + fx] = YP[my * G::FS + mx];
  }
}

// np.clip(img, 0, 1) (:138) and img ** (1 / gamma) (:243) of one float64 value.  Rounding to float32 is
// monotonic and 0, 1 are float32 numbers, so clipping after the conversion gives the same float32 as clipping
// before it (one v_cvt + a clamp instead of two float64 compares).  0 needs no special case: log2(0) = -inf,
// -inf * (1 / gamma) = -inf, exp2(-inf) = 0 -- in libm and in v_log_f32 / v_exp_f32 alike.
R2L_HD float r2l_clip_gamma(double rgb, float inv_gamma) {
  const float xf = fminf(fmaxf((float)rgb, 0.f), 1.f);
  return r2l_exp2(r2l_log2(xf) * inv_gamma);
}

// The optional T.Normalize epilogue on the NC pixels x 3 channels a lane is about to store -- torchvision's
// tensor.sub_(mean).div_(std) in float32 (train.py:157-171) -- under ONE uniform branch behind the gamma code
// (a test per value would cut the pixel loops into basic blocks; measured +10 % on the luma chains).
// The quotient is the correctly rounded one in 3 instructions instead of the ~10 of a division: with r = RN(1/s),
// q0 = RN(t r), the remainder t - q0 s is exact in one fma and RN(q0 + rem r) is RN(t / s) (Markstein's theorem; it
// excludes divisors whose significand is all ones: normalize = 2 divides; tests/test_oracle_golden.py checks the
// sequence on 3 * 10^6 quotients per std).
template <int NC, class A>
R2L_HD void r2l_static_normalize(const A& a, float x[3][NC]) {
  if (a.normalize == 2) {
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k)
      R2L_PRAGMA_UNROLL
    for (int c = 0; c < NC; ++c) x[k][c] = (x[k][c] - a.nmean[k]) / a.nstd[k];
  } else if (a.normalize) {
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      const float m = a.nmean[k], s = a.nstd[k], r = a.nrcp[k];
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < NC; ++c) {
        const float t = x[k][c] - m;
        const float q0 = t * r;
        x[k][c] = fmaf(fmaf(-q0, s, t), r, q0);
      }
    }
  }
}

// ---- pixel stage ---------------------------------------------------------------------------------
template <class G, bool BORDER, bool FULL>
R2L_HD void r2l_static_pixels_impl(int mt, const float* V, const double* YP, const R2LStaticArgs& a,
                                   const R2LTile& t) {
  constexpr int NC = R2LStaticMT<FULL>::NC, MR = R2LStaticMT<FULL>::MR, TXN = G::TW / NC;
  const int tx = mt % TXN, ty = mt / TXN;
  const int gy0 = t.oy + MR * ty, gx0 = t.ox + NC * tx;
  const int fy0 = MR * ty + 4, fx0 = NC * tx + 4;
  double d[MR][NC][3];
  r2l_static_demosaic<G, BORDER, NC, MR>(V, fy0, fx0, gy0, gx0, a, d);
  double ypp[MR][NC];
  if (FULL) {
    // ndimage.gaussian_filter is two 1-D passes, along axis 0 (rows) first, then along axis 1: 5 + 5 taps per
    // pixel instead of 25 (and the same order of operations as scipy)
    static_assert(!FULL || NC % 2 == 0, "the luma window is read two float64 at a time");
    double vert[MR][NC + 4];
    R2L_PRAGMA_UNROLL
    for (int r = 0; r < MR; ++r)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < NC + 4; ++j) vert[r][j] = 0.0;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < MR + 4; ++i) {  // input row fy0-2+i contributes to output rows r = i-4 .. i
      double row[NC + 4];
      const double* rp = YP + (fy0 - 2 + i) * G::FS + fx0 - 2;  // fx0 is even: 16-byte aligned
      R2L_PRAGMA_UNROLL
      for (int j = 0; j < NC + 4; j += 2) {
        const r2l_d2 q = r2l_lds_d2(rp + j);
        row[j] = q.x;
        row[j + 1] = q.y;
      }
      R2L_PRAGMA_UNROLL
      for (int r = 0; r < MR; ++r) {
        const int ki = i - r;
        if (ki < 0 || ki > 4) continue;
        R2L_PRAGMA_UNROLL
        for (int j = 0; j < NC + 4; ++j) vert[r][j] = fma(a.gk[ki], row[j], vert[r][j]);
      }
    }
    R2L_PRAGMA_UNROLL
    for (int r = 0; r < MR; ++r)
      R2L_PRAGMA_UNROLL
    for (int c = 0; c < NC; ++c) {
      double sacc = 0.0;
      R2L_PRAGMA_UNROLL
      for (int kj = 0; kj < 5; ++kj) sacc = fma(a.gk[kj], vert[r][c + kj], sacc);
      ypp[r][c] = sacc;
    }
  }
  const size_t plane = (size_t)a.H * a.W;
  const bool vec_ok = ((a.W & (NC - 1)) == 0) && (gx0 + NC - 1 < a.W);
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < MR; ++r) {
    const int gy = gy0 + r;
    if (gy >= a.H) break;
    float x[3][NC];
    R2L_PRAGMA_UNROLL
    for (int c = 0; c < NC; ++c) {
      double rgb[3];
      if (FULL) {
        const double u = fma(a.T[3], d[r][c][0], fma(a.T[4], d[r][c][1], a.T[5] * d[r][c][2]));
        const double v = fma(a.T[6], d[r][c][0], fma(a.T[7], d[r][c][1], a.T[8] * d[r][c][2]));
        R2L_PRAGMA_UNROLL
        for (int k = 0; k < 3; ++k)
          rgb[k] = fma(a.M2[k * 3], ypp[r][c], fma(a.M2[k * 3 + 1], u, a.M2[k * 3 + 2] * v));
      } else {
        R2L_PRAGMA_UNROLL
        for (int k = 0; k < 3; ++k)
          rgb[k] = fma(a.wbccm[k * 3], d[r][c][0], fma(a.wbccm[k * 3 + 1], d[r][c][1], a.wbccm[k * 3 + 2] * d[r][c][2]));
      }
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < 3; ++k) x[k][c] = r2l_clip_gamma(rgb[k], a.inv_gamma);
    }
    r2l_static_normalize<NC>(a, x);
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      float* o = a.out + ((size_t)t.b * 3 + k) * plane + (size_t)gy * a.W + gx0;
      if (vec_ok && NC == 4) {
        r2l_f4 st;
        st.x = x[k][0];
        st.y = x[k][1];
        st.z = x[k][2];
        st.w = x[k][NC - 1];
        *(r2l_f4*)o = st;
      } else if (vec_ok && NC == 2) {
        r2l_f2 st;
        st.x = x[k][0];
        st.y = x[k][1];
        *(r2l_f2*)o = st;
      } else {
        R2L_PRAGMA_UNROLL
        for (int c = 0; c < NC; ++c)
          if (gx0 + c < a.W) o[c] = x[k][c];
      }
    }
  }
}

template <class G, bool FULL>
R2L_HD void r2l_static_pixels(int tid, const float* V, const double* YP, const R2LStaticArgs& a,
                              const R2LTile& t) {
  constexpr int NC = R2LStaticMT<FULL>::NC, MR = R2LStaticMT<FULL>::MR, TXN = G::TW / NC;
  static_assert(TXN * (G::TH / MR) == R2L_NT, "one micro-tile per lane");
  const int tx = tid % TXN, ty = tid / TXN;
  const int gy0 = t.oy + MR * ty, gx0 = t.ox + NC * tx;
  if (gy0 >= a.H || gx0 >= a.W) return;
  const int halo = (a.debayer == R2L_DEBAYER_MALVAR2004) ? 2 : 1;
  if (r2l_static_touches_border(gy0, gx0, halo, a.H, a.W, NC, MR))
    r2l_static_pixels_impl<G, true, FULL>(tid, V, YP, a, t);
  else
    r2l_static_pixels_impl<G, false, FULL>(tid, V, YP, a, t);
}

// The chain's constants (45 float64 + the frame geometry) are kernel arguments; kept live across the tile loop they
// overflow the scalar registers and hipcc parks them in VGPR lanes (1,400 v_readlane in the kernel, two per use of
// a constant).  The argument block is therefore copied to LDS once per workgroup, and every phase re-reads from
// there what it uses (broadcast ds_reads into registers for the length of the phase; reading it from the kernarg
// segment again per phase costs a global-memory round trip at the head of every phase).
#define R2L_STATIC_ARGS_FLOATS 160
static_assert(sizeof(R2LStaticArgs) <= 4 * R2L_STATIC_ARGS_FLOATS, "argument block copy in LDS");
#ifdef R2L_EMUL
#define R2L_STATIC_ARGS(a, lds_args) (a)
R2L_HD void r2l_static_args_to_lds(int, const R2LStaticArgs&, float*) {}
#else
R2L_HD void r2l_static_args_to_lds(int tid, const R2LStaticArgs& a, float* lds_args) {
  (void)a;
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc parses this too and has no address spaces)
  const __attribute__((address_space(4))) float* p =
      (const __attribute__((address_space(4))) float*)__builtin_amdgcn_kernarg_segment_ptr();
  if (tid < (int)((sizeof(R2LStaticArgs) + 3) / 4)) lds_args[tid] = p[tid];
#else
  (void)tid;
  (void)lds_args;
#endif
}
R2L_HD R2LStaticArgs r2l_static_args_from_lds(const float* lds_args) {
  R2LStaticArgs r;
  __builtin_memcpy(&r, lds_args, sizeof(r));
  return r;
}
#define R2L_STATIC_ARGS(a, lds_args) r2l_static_args_from_lds(lds_args)
#endif

template <class G>
R2L_BLOCKFN void r2l_static_block(const R2LStaticArgs& a, int bid, int nblk, float* lds) {
  float* V = lds + G::PAD;
  double* Y = (double*)(V + G::PLANE);
  double* YP = Y + G::PLANE;
  float* lds_args = (float*)(YP + G::PLANE);
  R2L_PHASE_BEGIN
  r2l_static_args_to_lds(tid, a, lds_args);
  R2L_PHASE_END
  R2LTileWalk w = r2l_walk_init(a.B, a.H, a.W, G::TW, G::TH, bid, nblk);
  R2LTile t, tn;
  R2L_TREG_DECL(R2LStaticPre<G>, pre);
  bool have = r2l_walk_next(w, a.H, a.W, G::TW, G::TH, t);
  R2L_PHASE_BEGIN
  if (have) r2l_fetch_raw_sym<G>(tid, a.raw, (size_t)t.b * a.H * a.W, t.oy, t.ox, a.H, a.W, R2L_TREG(pre));
  R2L_PHASE_END
  while (have) {
    R2L_PHASE_BEGIN
    r2l_store_raw_sym<G>(tid, V, a.raw, t.ox, a.W, R2L_TREG(pre));
    R2L_PHASE_END
    const bool haven = r2l_walk_next(w, a.H, a.W, G::TW, G::TH, tn);
    R2L_PHASE_BEGIN
    r2l_static_compute_y<G>(tid, V, Y, R2L_STATIC_ARGS(a, lds_args), t.oy, t.ox);
    R2L_PHASE_END
    R2L_PHASE_BEGIN
    r2l_static_compute_yp<G>(tid, Y, YP, R2L_STATIC_ARGS(a, lds_args));
    R2L_PHASE_END
    if (t.border) {
      R2L_PHASE_BEGIN
      r2l_static_fill_yp<G>(tid, YP, t.oy, t.ox, a.H, a.W);
      R2L_PHASE_END
    }
    R2L_PHASE_BEGIN
    // next tile's frame: in flight during the pixel phase
    if (haven) r2l_fetch_raw_sym<G>(tid, a.raw, (size_t)tn.b * a.H * a.W, tn.oy, tn.ox, a.H, a.W, R2L_TREG(pre));
    r2l_static_pixels<G, true>(tid, V, YP, R2L_STATIC_ARGS(a, lds_args), t);
    R2L_PHASE_END
    t = tn;
    have = haven;
  }
}

// short chain: demosaic -> WB -> CCM -> clip -> gamma; a single LDS plane
template <class G>
R2L_BLOCKFN void r2l_static_short_block(const R2LStaticArgs& a, int bid, int nblk, float* lds) {
  float* V = lds + G::PAD;
  R2LTileWalk w = r2l_walk_init(a.B, a.H, a.W, G::TW, G::TH, bid, nblk);
  R2LTile t;
  while (r2l_walk_next(w, a.H, a.W, G::TW, G::TH, t)) {
    const size_t img0 = (size_t)t.b * a.H * a.W;
    R2L_PHASE_BEGIN
    r2l_load_raw_sym<G>(tid, V, a.raw, img0, t.oy, t.ox, a.H, a.W);
    R2L_PHASE_END
    R2L_PHASE_BEGIN
    r2l_static_pixels<G, false>(tid, V, (const double*)nullptr, a, t);
    R2L_PHASE_END
  }
}
