// r2l_static_menon.h -- demosaicing_CFA_Bayer_Menon2007 (pipeline_numpy.py:96-97: `--sp_debayer menon2007`) as passes over
// float64 planes, numpy semantics.
//
// The algorithm lives in a third-party package the reference imports (colour-demosaicing 0.1.6, environment.yml:296), absent from
// the reference tree and from this image: PARITY UNPINNED -- the kernels follow the published DDFAPD algorithm (Menon, Andriani,
// Calvagno, IEEE TIP 2007) as that package implements it, restated in oracle/isp_oracle.py (demosaicing_CFA_Bayer_Menon2007), which
// is what the tests compare with.  Unlike bilinear / Malvar2004 it is a DECISION-directed chain of nine dependent stencils (support
// about +-9 pixels), so it runs as plane passes, not as one streaming kernel: the long tail of `--sp_debayer` sweeps, not a
// throughput configuration.
//
//   stage 0  G_H, G_V (green at the red / blue sites by the directional 5-tap filters h_0 + h_1, 'mirror' borders) and the
//            chrominance differences C_H, C_V = CFA - G_H / G_V at those sites (0 at the green ones)
//   stage 1  D_H(x) = |C_H(x) - C_H(x + 2)| (np.pad 'reflect' on the right), D_V likewise down the rows; the classifiers
//            d_H = convolve(D_H, k, 'constant'), d_V = convolve(D_V, k^T, 'constant') with the 5x5 kernel k; M = (d_V >= d_H);
//            G = M ? G_H : G_V; R / B planes start as the CFA at their own sites
//   stage 2  R, B at the green sites (row resp. column neighbours, bilinear on the colour difference to G)
//   stage 3  R at the blue sites, B at the red sites along the decided direction
//   stage 4  refining: G at the red / blue sites from the 3-tap mean of R - G (B - G) along the decided direction
//   stage 5  refining: R, B at the green sites from the refined differences
//   stage 6  refining: R at the blue sites, B at the red sites from the 3-tap mean of R - B along the decided direction
//   stage 7  white balance, colour matrix (in place: the planes become the linear RGB image) and the luma plane Y
//   stage 8  (chains with a luma stage) RGB <- rgb_from_yuv * (Y'', U, V), in place
// Every stage from 2 on is IN PLACE: it writes one colour at one kind of site and reads that colour only at the other kinds.
// The decision M compares two float64 sums, so every sum here is formed in the order scipy's ndimage forms it (correlate1d's
// symmetric fast path: centre tap first, then the pairs from the farthest inwards; convolve: the flipped kernel's non-zero taps
// in row-major order) without contraction -- a decision that flips on a last-bit difference would change a pixel by far more than
// the parity bar.  Layout: R, G, B as the planes of a (B,3,H,W) float64 image (what the fft_denoising passes and the finish pass
// of r2l_static_planes.h take), five more (B,H,W) planes for G_H, G_V, C_H, C_V (later: the luma ping-pong) and M.
#pragma once
#include "r2l_static_planes.h"

#ifndef R2L_EMUL
#pragma clang fp contract(off)
#endif

struct R2LMenonArgs {
  R2LStaticArgs s;
  int stage;
  double* rgb;            // (B,3,H,W)
  double* gh;             // (B,H,W) each
  double* gv;
  double* ch;
  double* cv;
  double* m;              // 1.0: horizontal
  double* luma;           // stage 7 out / stage 8 in (the filtered plane)
  int want_luma;
  double M1[9];           // skimage yuv_from_rgb (r2l_static_setup)
};

// black-level-corrected CFA sample, in the arithmetic of the frame's dtype (remove_blacklv works in place, pipeline_numpy.py:152-158)
R2L_HD double r2l_menon_cfa(const R2LStaticArgs& a, size_t img, int y, int x) {
  const size_t i = img + (size_t)y * a.W + x;
  const int site = ((y & 1) << 1) | (x & 1);
  if (a.raw.f64) return a.raw.f64[i] - a.bl[site];
  return (double)(r2l_raw_elem(a.raw, i) - a.blf[site]);
}
// scipy correlate1d, symmetric 3-tap kernel (w1, w0, w1): centre first, then the pair
R2L_HD double r2l_sym3(double l, double c, double r, double w0, double w1) {
  double t = c * w0;
  t += (l + r) * w1;
  return t;
}
R2L_BLOCKFN void r2l_static_menon_block(const R2LMenonArgs& ma, int bid, int nblk, float* lds) {
  (void)lds;
  const R2LStaticArgs& a = ma.s;
  const int H = a.H, W = a.W;
  const size_t hw = (size_t)H * W;
#ifndef R2L_MENON_LOOP
#define R2L_MENON_LOOP 1
#endif
  R2L_PHASE_BEGIN
#if R2L_MENON_LOOP == 2
  // a workgroup walks image rows, its lanes the columns (measured 1.8-4x slower than the flat loop: profiles/r06_menon.txt; kept as an A/B switch)
  const long nrows = (long)a.B * H;
  for (long row = bid; row < nrows; row += nblk)
  for (int x = tid; x < W; x += R2L_NT) {
    const size_t b = (size_t)(row / H);
    const int y = (int)(row - (long)b * H);
    const size_t p = (size_t)y * W + x;
    const size_t i = b * hw + p;
#else
  // flat grid-stride loop over the pixels; the index is decomposed in 32-bit arithmetic where the batch allows it
  const size_t n = (size_t)a.B * hw;
  const bool small = R2L_MENON_LOOP == 1 && n <= 0xffffffffull;
  for (size_t i = (size_t)bid * R2L_NT + tid; i < n; i += (size_t)nblk * R2L_NT) {
    size_t b, p;
    int y, x;
    if (small) {
      const unsigned i32 = (unsigned)i, hw32 = (unsigned)hw;
      const unsigned b32 = i32 / hw32, p32 = i32 - b32 * hw32, y32 = p32 / (unsigned)W;
      b = b32;
      p = p32;
      y = (int)y32;
      x = (int)(p32 - y32 * (unsigned)W);
    } else {
      b = i / hw;
      p = i - b * hw;
      y = (int)(p / (size_t)W);
      x = (int)(p - (size_t)y * W);
    }
#endif
    const size_t img = b * hw;          // offset of image b in a (B,H,W) plane
    const size_t img3 = b * 3 * hw;     // ... in the (B,3,H,W) image
    double* Rp = ma.rgb + img3;
    double* Gp = Rp + hw;
    double* Bp = Gp + hw;
    const bool red_row = (y & 1) == 0, g_site = ((y ^ x) & 1) != 0, r_site = red_row && !(x & 1), b_site = !red_row && (x & 1);
    const int xm1 = r2l_mirror(x - 1, W), xp1 = r2l_mirror(x + 1, W), ym1 = r2l_mirror(y - 1, H), yp1 = r2l_mirror(y + 1, H);
#define R2L_MN_AT(P, yy, xx) (P)[(size_t)(yy) * W + (xx)]
    switch (ma.stage) {
      case 0: {
        const double c = r2l_menon_cfa(a, img, y, x);
        double gh = c, gv = c, ch = 0.0, cv = 0.0;
        if (!g_site) {
          const int xm2 = r2l_mirror(x - 2, W), xp2 = r2l_mirror(x + 2, W), ym2 = r2l_mirror(y - 2, H), yp2 = r2l_mirror(y + 2, H);
          // _cnv(CFA, h_0) + _cnv(CFA, h_1), h_0 = [0, .5, 0, .5, 0], h_1 = [-.25, 0, .5, 0, -.25]
          const double a0 = (r2l_menon_cfa(a, img, y, xm1) + r2l_menon_cfa(a, img, y, xp1)) * 0.5;
          double a1 = c * 0.5;
          a1 += (r2l_menon_cfa(a, img, y, xm2) + r2l_menon_cfa(a, img, y, xp2)) * -0.25;
          gh = a0 + a1;
          const double b0 = (r2l_menon_cfa(a, img, ym1, x) + r2l_menon_cfa(a, img, yp1, x)) * 0.5;
          double b1 = c * 0.5;
          b1 += (r2l_menon_cfa(a, img, ym2, x) + r2l_menon_cfa(a, img, yp2, x)) * -0.25;
          gv = b0 + b1;
          ch = c - gh;
          cv = c - gv;
        }
        ma.gh[i] = gh;
        ma.gv[i] = gv;
        ma.ch[i] = ch;
        ma.cv[i] = cv;
        break;
      }
      case 1: {
        const double* CH = ma.ch + img;
        const double* CV = ma.cv + img;
        // D_H(yy, xx) inside the image, 0 outside ('constant'); the sample two ahead by np.pad(..., 'reflect'): W -> W-2, W+1 -> W-3
        auto DH = [&](int yy, int xx) -> double {
          if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) return 0.0;
          const int x2 = xx + 2 < W ? xx + 2 : 2 * (W - 1) - (xx + 2);
          return fabs(R2L_MN_AT(CH, yy, xx) - R2L_MN_AT(CH, yy, x2));
        };
        auto DV = [&](int yy, int xx) -> double {
          if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) return 0.0;
          const int y2 = yy + 2 < H ? yy + 2 : 2 * (H - 1) - (yy + 2);
          return fabs(R2L_MN_AT(CV, yy, xx) - R2L_MN_AT(CV, y2, xx));
        };
        // scipy.ndimage.convolve: the flipped kernel's non-zero taps in row-major order, tmp += value * weight
        double dh = 0.0, dv = 0.0;
        dh += DH(y - 2, x - 2) * 1.0;
        dh += DH(y - 2, x) * 1.0;
        dh += DH(y - 1, x - 1) * 1.0;
        dh += DH(y, x - 2) * 3.0;
        dh += DH(y, x) * 3.0;
        dh += DH(y + 1, x - 1) * 1.0;
        dh += DH(y + 2, x - 2) * 1.0;
        dh += DH(y + 2, x) * 1.0;
        // k^T flipped: rows (i-2: j-2, j+2), (i-1: j-1, j+1) ... in row-major order of the flipped transposed kernel
        dv += DV(y - 2, x - 2) * 1.0;
        dv += DV(y - 2, x) * 3.0;
        dv += DV(y - 2, x + 2) * 1.0;
        dv += DV(y - 1, x - 1) * 1.0;
        dv += DV(y - 1, x + 1) * 1.0;
        dv += DV(y, x - 2) * 1.0;
        dv += DV(y, x) * 3.0;
        dv += DV(y, x + 2) * 1.0;
        const bool horiz = dv >= dh;
        ma.m[i] = horiz ? 1.0 : 0.0;
        const double c = r2l_menon_cfa(a, img, y, x);
        Gp[p] = horiz ? ma.gh[i] : ma.gv[i];
        Rp[p] = r_site ? c : 0.0;
        Bp[p] = b_site ? c : 0.0;
        break;
      }
      case 2: {
        if (!g_site) break;
        const double g = Gp[p];
        const double gh = r2l_sym3(R2L_MN_AT(Gp, y, xm1), 0.0, R2L_MN_AT(Gp, y, xp1), 0.0, 0.5);
        const double gvv = r2l_sym3(R2L_MN_AT(Gp, ym1, x), 0.0, R2L_MN_AT(Gp, yp1, x), 0.0, 0.5);
        const double rh = r2l_sym3(R2L_MN_AT(Rp, y, xm1), 0.0, R2L_MN_AT(Rp, y, xp1), 0.0, 0.5);
        const double rv = r2l_sym3(R2L_MN_AT(Rp, ym1, x), 0.0, R2L_MN_AT(Rp, yp1, x), 0.0, 0.5);
        const double bh = r2l_sym3(R2L_MN_AT(Bp, y, xm1), 0.0, R2L_MN_AT(Bp, y, xp1), 0.0, 0.5);
        const double bv = r2l_sym3(R2L_MN_AT(Bp, ym1, x), 0.0, R2L_MN_AT(Bp, yp1, x), 0.0, 0.5);
        // red rows: R from the row, B from the column; blue rows the other way round (G + cnv(R) - cnv(G))
        Rp[p] = red_row ? (g + rh) - gh : (g + rv) - gvv;
        Bp[p] = red_row ? (g + bv) - gvv : (g + bh) - gh;
        break;
      }
      case 3: {
        if (g_site) break;
        const bool horiz = ma.m[i] == 1.0;
        const double rr = horiz ? r2l_sym3(R2L_MN_AT(Rp, y, xm1), 0.0, R2L_MN_AT(Rp, y, xp1), 0.0, 0.5)
                                : r2l_sym3(R2L_MN_AT(Rp, ym1, x), 0.0, R2L_MN_AT(Rp, yp1, x), 0.0, 0.5);
        const double bb = horiz ? r2l_sym3(R2L_MN_AT(Bp, y, xm1), 0.0, R2L_MN_AT(Bp, y, xp1), 0.0, 0.5)
                                : r2l_sym3(R2L_MN_AT(Bp, ym1, x), 0.0, R2L_MN_AT(Bp, yp1, x), 0.0, 0.5);
        if (b_site)
          Rp[p] = (Bp[p] + rr) - bb;   // B + cnv(R) - cnv(B)
        else
          Bp[p] = (Rp[p] + bb) - rr;   // R + cnv(B) - cnv(R)
        break;
      }
      case 4: {
        if (g_site) break;
        const bool horiz = ma.m[i] == 1.0;
        const double* Cp = r_site ? Rp : Bp;  // the site's own colour: G = C - mean3(C - G) along the decided direction
        const int y0 = horiz ? y : ym1, y1 = horiz ? y : yp1, x0 = horiz ? xm1 : x, x1 = horiz ? xp1 : x;
        const double third = 1.0 / 3.0;
        const double dc = Cp[p] - Gp[p];
        const double dl = R2L_MN_AT(Cp, y0, x0) - R2L_MN_AT(Gp, y0, x0), dr = R2L_MN_AT(Cp, y1, x1) - R2L_MN_AT(Gp, y1, x1);
        Gp[p] = Cp[p] - r2l_sym3(dl, dc, dr, third, third);
        break;
      }
      case 5: {
        if (!g_site) break;
        // red rows (= blue columns): R - G from the row's red sites, B - G from the column's blue sites; blue rows the other way round
        const double g = Gp[p];
        const double rgl = R2L_MN_AT(Rp, red_row ? y : ym1, red_row ? xm1 : x) - R2L_MN_AT(Gp, red_row ? y : ym1, red_row ? xm1 : x);
        const double rgr = R2L_MN_AT(Rp, red_row ? y : yp1, red_row ? xp1 : x) - R2L_MN_AT(Gp, red_row ? y : yp1, red_row ? xp1 : x);
        const double bgl = R2L_MN_AT(Bp, red_row ? ym1 : y, red_row ? x : xm1) - R2L_MN_AT(Gp, red_row ? ym1 : y, red_row ? x : xm1);
        const double bgr = R2L_MN_AT(Bp, red_row ? yp1 : y, red_row ? x : xp1) - R2L_MN_AT(Gp, red_row ? yp1 : y, red_row ? x : xp1);
        Rp[p] = g + r2l_sym3(rgl, 0.0, rgr, 0.0, 0.5);
        Bp[p] = g + r2l_sym3(bgl, 0.0, bgr, 0.0, 0.5);
        break;
      }
      case 6: {
        if (g_site) break;
        const bool horiz = ma.m[i] == 1.0;
        const int y0 = horiz ? y : ym1, y1 = horiz ? y : yp1, x0 = horiz ? xm1 : x, x1 = horiz ? xp1 : x;
        const double third = 1.0 / 3.0;
        const double dc = Rp[p] - Bp[p];
        const double dl = R2L_MN_AT(Rp, y0, x0) - R2L_MN_AT(Bp, y0, x0), dr = R2L_MN_AT(Rp, y1, x1) - R2L_MN_AT(Bp, y1, x1);
        const double mean = r2l_sym3(dl, dc, dr, third, third);
        if (b_site)
          Rp[p] = Bp[p] + mean;
        else
          Bp[p] = Rp[p] - mean;
        break;
      }
      case 7: {
        // img * white_balance, then einsum('ijk,lk->ijl', img, colour_matrix) (pipeline_numpy.py:161-167), rgb2yuv's first row
        // (wbccm = colour_matrix * diag(white_balance), as the other static kernels fold it)
        const double r = Rp[p], g = Gp[p], bl = Bp[p];
        double o[3];
        R2L_PRAGMA_UNROLL
        for (int k = 0; k < 3; ++k) o[k] = r * a.wbccm[k * 3] + g * a.wbccm[k * 3 + 1] + bl * a.wbccm[k * 3 + 2];
        Rp[p] = o[0];
        Gp[p] = o[1];
        Bp[p] = o[2];
        if (ma.want_luma) ma.luma[i] = o[0] * ma.M1[0] + o[1] * ma.M1[1] + o[2] * ma.M1[2];
        break;
      }
      default: {  // 8: yuv = rgb2yuv(rgb); yuv[0] = the filtered luma; rgb = yuv2rgb(yuv)
        const double r = Rp[p], g = Gp[p], bl = Bp[p];
        const double yy = ma.luma[i];
        const double u = r * ma.M1[3] + g * ma.M1[4] + bl * ma.M1[5], v = r * ma.M1[6] + g * ma.M1[7] + bl * ma.M1[8];
        Rp[p] = yy * a.M2[0] + u * a.M2[1] + v * a.M2[2];
        Gp[p] = yy * a.M2[3] + u * a.M2[4] + v * a.M2[5];
        Bp[p] = yy * a.M2[6] + u * a.M2[7] + v * a.M2[8];
        break;
      }
    }
#undef R2L_MN_AT
  }
  R2L_PHASE_END
}

#ifndef R2L_EMUL
#pragma clang fp contract(fast)
#endif
