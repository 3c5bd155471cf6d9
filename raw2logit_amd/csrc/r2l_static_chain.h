// r2l_static_chain.h -- the static chains WITH a luma stencil chain as one row-streaming kernel:
//
//   remove_blacklv -> demosaicing_CFA_Bayer_bilinear | _Malvar2004 -> white balance -> colour matrix -> rgb2yuv ->
//   [sharpening_filter: Y <- convolve2d(Y, K, 'same', fill 0)] ->
//   [gaussian_denoising: Y <- gaussian_filter(Y, 0.5)  |  median_denoising: Y <- median_filter(Y, 3)]
//   -> yuv2rgb -> clip[0,1] -> x ** (1/gamma)
//
// i.e. processing() (pipeline_numpy.py:70-141) with the default chain of train.py:96-101, the chains with only one
// of the two luma filters, and the Malvar2004 / median_denoising alternates of the reference's sweeps
// (figures/train.sh:49-79; pipeline_numpy.py:94-95, :194-200).  Template parameters: DEB (0 bilinear: 3-row raw
// window; 1 Malvar2004: 5-row window in a ring of 6) and DN (0 Gaussian: ring of 6 horizontally blurred rows, output
// row q-3; 1 median: the last 3 sharpened rows with one neighbour column each side, output row q-2).
// Round 1 ran the default chain as an LDS tile kernel (r2l_static_block: float64 planes, 115 KB
// of LDS, one workgroup per CU, 21 % of the HBM peak).  Here, like the short chain (r2l_static_stream.h), every
// wavefront is a line-buffer ISP: it owns a strip of 256 columns (4 per lane) and a band of rows, walks down the
// band and keeps in registers
//     the last 3 raw rows (float64, black level removed)            -> bilinear demosaic of the middle row
//     the last 3 luma rows Y                                         -> 5-point sharpen  Y'
//     a ring of 6 horizontally blurred rows  Hb = g (*)_x Y'          -> vertical 5-tap blur  Y''
// and in a wave-private LDS ring the chroma (U, V) of the last 4 rows, which waits 3 rows for its luma.  The halo
// of the luma chain is 1 + 2 columns each side: inside a wavefront the neighbour columns come from the
// neighbouring lanes (DPP wave shifts); at the strip edges lane 0 / lane 63 exchange them with the neighbouring
// wavefront of the workgroup through LDS -- 6 float64 values per wavefront and row, one barrier per row (the
// wavefronts of a workgroup cover one image row side by side and advance in lock step; the workgroups of a CU
// are at different rows, so the barrier of one is filled by the others).  Row latency: raw row q+1 gives Y(q),
// Y'(q-1), Hb(q-1) and the finished output row q-3; a band re-computes 7 rows of halo.
// Algorithmic traffic: 4 B in + 12 B out per pixel, like the short chain.  Linear part in float64 (see
// r2l_static_kernels.h), log2 / exp2 in float32.
#pragma once
#include "r2l_static_stream.h"
#ifndef R2L_CHAIN_BF
#define R2L_CHAIN_BF 1
#endif
#ifndef R2L_CHAIN_PROGRESS_PRIO
#define R2L_CHAIN_PROGRESS_PRIO 1  // (default chain 963 -> 944 us, profiles/r04_static_chain_ab.txt)
#endif
#if R2L_CHAIN_PROGRESS_PRIO
#define R2L_CHAIN_PRIO(d, t) R2L_PROGRESS_PRIO(d, t)
#else
#define R2L_CHAIN_PRIO(d, t)
#endif

#ifndef R2L_SERIAL

#ifndef R2L_CHAIN_PF
#define R2L_CHAIN_PF 3
#endif

struct R2LStaticChainArgs {
  R2LStaticArgs s;
  int nband, band_h;
  int nw;  // wavefronts per workgroup = 256-column strips per image row (1, 2, 4, 8)
};

// float64 value of x in the previous / next lane of the wavefront; lane 0 / lane 63 get `edge`
R2L_HD double r2l_wave_shr1_d(double x, double edge) {
  const unsigned long long xu = __builtin_bit_cast(unsigned long long, x), eu = __builtin_bit_cast(unsigned long long, edge);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)eu, (int)(unsigned)xu, 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(eu >> 32), (int)(unsigned)(xu >> 32), 0x138, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
R2L_HD double r2l_wave_shl1_d(double x, double edge) {
  const unsigned long long xu = __builtin_bit_cast(unsigned long long, x), eu = __builtin_bit_cast(unsigned long long, edge);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)eu, (int)(unsigned)xu, 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(eu >> 32), (int)(unsigned)(xu >> 32), 0x130, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}

#define R2L_CHAIN_EX 8                                  // doubles per wavefront and buffer in the exchange area
// chroma ring: FR rows x (U[4], V[4]) x 64 lanes: 4 rows = 16 KB per wavefront (7 rows, 28 KB, behind unsharp_masking)
#define R2L_CHAIN_FIFO_ROWS(SH) ((SH) ? 7 : 4)
#define R2L_CHAIN_FIFO_DOUBLES(SH) (R2L_CHAIN_FIFO_ROWS(SH) * 4 * 64 * 2)
#define R2L_CHAIN_LDS_DOUBLES(NW, SH) (2 * (NW) * R2L_CHAIN_EX + (NW) * R2L_CHAIN_FIFO_DOUBLES(SH))

// per-lane state of the luma chain
// (a float32 raw window, widened on the fly as in the short chain's Malvar2004 kernel, does not pay here: the chroma ring
// in LDS, not the registers, keeps these kernels at two wavefronts per SIMD, and the extra conversions cost 2 %:
// profiles/r03_static_windows.txt)
template <int DEB, int SH, int DN>
struct R2LChainState {
  double rw[DEB ? 6 : 3][8];  // raw rows (slot = row mod 3 / mod 6): columns x0-2 .. x0+5, black level removed
  // SH 0: luma rows q-2 .. q (slot = row mod 3).  SH 1 (unsharp_masking: 9 rows, which no ring that divides the
  // 6-fold unroll holds): rows q-8 .. q as a shift register, [8] = row q
  double yr[SH ? 9 : 3][4];
  double yl, yrr;             // SH 0: left / right neighbour of the luma row q-1 (zero outside the image)
  // DN 0: horizontally blurred sharpened luma, own columns (slot = row mod 6)
  // DN 1: sharpened luma, columns x0-1 .. x0+4 (symmetric at the image edges) (slot = row mod 3)
  double hb[DN == 0 ? 6 : 3][DN == 0 ? 4 : 6];
};

R2L_HD void r2l_chain_cswap(double& a, double& b) {
  const double lo = fmin(a, b), hi = fmax(a, b);
  a = lo;
  b = hi;
}
R2L_HD double r2l_chain_med3(double a, double b, double c) { return fmax(fmin(a, b), fmin(fmax(a, b), c)); }

// One step of the pipeline: raw row q+1 has just entered the window (slot (K+1)%3 of rw).  K = q mod 6, a
// compile-time constant of the 6-fold unrolled loop, makes every register-array index a constant.
// The chain's 45 float64 constants are kernel arguments.  Kept live across the row loop they overflow the scalar
// registers and get parked in VGPR lanes (35 v_readlane per row); read through a laundered pointer to the kernarg
// segment they are re-loaded (s_load, scalar cache) in front of the section that uses them.
typedef const R2L_CONSTAS R2LStaticArgs* R2LStaticArgsK;
R2L_HD R2LStaticArgsK r2l_chain_consts() {
#if defined(__HIP_DEVICE_COMPILE__)
  R2LStaticArgsK p = (R2LStaticArgsK)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
#elif defined(R2L_LOCKSTEP)
  return (R2LStaticArgsK)r2l_ls::kernarg();  // (R2LStaticChainArgs starts with its R2LStaticArgs)
#else
  return nullptr;
#endif
}

template <int DEB, int SH, int DN, int K0>
R2L_HD void r2l_chain_step(const R2LStaticArgs& a_, R2LChainState<DEB, SH, DN>& st, int q_, int y0, int y1, bool le,
                           bool re, int NW, int wave, int lane, double* ex, r2l_d2* fifo, float* outb, size_t plane,
                           int x0, bool store_ok) {
  constexpr int PY = K0 & 1;
  constexpr int FR = R2L_CHAIN_FIFO_ROWS(SH);
  const int H = a_.H;
#ifdef R2L_CHAIN_ARGS_LIVE
  const R2LStaticArgs& a = a_;
#else
  const R2L_CONSTAS R2LStaticArgs& a = *r2l_chain_consts();
#endif
  double yq[4];  // Y(q) (new)
  double yp[4];  // the sharpened row it completes
  {
  constexpr int K = K0;
  const int q = q_;
  const bool qin = (unsigned)q < (unsigned)H;
  // ---- demosaic + colour of row q: Y(q) to the window, (U, V)(q) to the LDS ring ---------------------------
  {
    double d[4][3];
    if constexpr (DEB == 0) {
      const auto* u = st.rw[(K + 2) % 3];    // raw row q-1
      const auto* m = st.rw[K % 3];          // raw row q
      const auto* l = st.rw[(K + 1) % 3];    // raw row q+1
      if (q > 0 && q < H - 1) {
        r2l_stream_bilinear_row_interior<PY>(u, m, l, le, re, d);
      } else {
        // first / last image row: a mirrored row carries the sites of the row it came from
        const int tpy[3] = {r2l_symmetric(q - 1, H) & 1, r2l_symmetric(q, H) & 1, r2l_symmetric(q + 1, H) & 1};
        r2l_stream_bilinear_row(u, m, l, tpy, le, re, d);
      }
    } else {
      // Malvar2004 convolves the UNMASKED mosaic (mirrored rows / columns are plain values) and selects by the
      // output pixel's site: rows q-2 .. q+2 of the ring of 6
      constexpr int R = DEB ? 6 : 3;
      r2l_stream_malvar_row_shared<PY>(st.rw[(K + 4) % R], st.rw[(K + 5) % R], st.rw[K % R], st.rw[(K + 1) % R],
                                       st.rw[(K + 2) % R], d);
    }
    r2l_d2 uv[4];
    R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c) {
      const double y = fma(a.T[0], d[c][0], fma(a.T[1], d[c][1], a.T[2] * d[c][2]));
      uv[c].x = fma(a.T[3], d[c][0], fma(a.T[4], d[c][1], a.T[5] * d[c][2]));
      uv[c].y = fma(a.T[6], d[c][0], fma(a.T[7], d[c][1], a.T[8] * d[c][2]));
      yq[c] = y;
    }
    if (!qin) {                                // convolve2d(..., fillvalue=0): no luma outside the image
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 4; ++c) yq[c] = 0.0;
    }
    r2l_d2* f = fifo + (size_t)((q + 2 * FR) % FR) * 4 * 64 + lane;  // [row slot][c][lane]: 16-byte lane stride, conflict-free
    R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c) f[c * 64] = uv[c];
  }
  if (SH == 1) {
    // ---- Y'(q-4) = unsharp_mask(Y, radius 1, amount 1) as the reference calls it (multichannel=True on the 2-D
    // plane: every column is a channel, so the sigma-1 Gaussian runs down the columns only; pipeline_numpy.py:170-177):
    // Y + (Y - g (*)_y Y) * amount, 9 taps, scipy 'reflect' = symmetric rows ------------------------------------
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 8; ++i)
      R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c) st.yr[i][c] = st.yr[i + 1][c];
    R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c) st.yr[SH ? 8 : 2][c] = yq[c];
    const int r = q - 4;
    double wv[9];
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 9; ++k) wv[k] = a.uk[k < 4 ? 4 - k : k - 4];
    if (r < 4 || r > H - 5) {  // a row outside the image gives its weight to its mirror image
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < 9; ++k) wv[k] = ((unsigned)(r + k - 4) < (unsigned)H) ? a.uk[k < 4 ? 4 - k : k - 4] : 0.0;
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < 9; ++k) {
        const int rr = r + k - 4;
        if ((unsigned)rr >= (unsigned)H) {
          const int t = r2l_symmetric(rr, H) - (r - 4);  // window slot of the mirror image
          R2L_PRAGMA_UNROLL
          for (int j = 0; j < 9; ++j) wv[j] += (j == t) ? a.uk[k < 4 ? 4 - k : k - 4] : 0.0;
        }
      }
    }
    R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c) {
      // scipy correlate1d sums w0 * c + sum_k (up_k + down_k) * w_k; the mirrored form keeps the same terms
      double bl = wv[4] * st.yr[SH ? 4 : 0][c];
      R2L_PRAGMA_UNROLL
      for (int k = 1; k <= 4; ++k)
        bl += st.yr[SH ? 4 - k : 0][c] * wv[4 - k] + st.yr[SH ? 4 + k : 0][c] * wv[4 + k];
      const double cc = st.yr[SH ? 4 : 0][c];
      yp[c] = cc + (cc - bl) * a.amount;
    }
  } else {
    // ---- Y'(q-1) = sharpen: centre cross of a.ksharp (the corners of both K and the identity are zero) --------
    double* ynew = st.yr[K % 3];
    const double* ym = st.yr[(K + 2) % 3];     // Y(q-1)
    const double* yu = st.yr[(K + 1) % 3];     // Y(q-2)
    R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c) ynew[c] = yq[c];
    const double kc = a.ksharp[4], kl = a.ksharp[3], kr = a.ksharp[5], ku = a.ksharp[1], kd = a.ksharp[7];
    // convolve2d flips the kernel: out(p) = sum K[i][j] * Y(p - (i-1, j-1))  ->  K[1][0] weighs the RIGHT
    // neighbour, K[0][1] the row BELOW
    R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c) {
      const double left = (c > 0) ? ym[c > 0 ? c - 1 : 0] : st.yl, right = (c < 3) ? ym[c < 3 ? c + 1 : 3] : st.yrr;
      yp[c] = fma(kc, ym[c], fma(kl, right, fma(kr, left, fma(ku, yq[c], kd * yu[c]))));
    }
  }
  }
  // From here on in terms of q = q_ - 3 behind unsharp_masking (its Y' row is q_ - 4 = q - 1, as behind the 3x3).
  constexpr int K = SH ? (K0 + 3) % 6 : K0;
  const int q = SH ? q_ - 3 : q_;
  // ---- strip edges: this wavefront's edge columns to LDS, the neighbours' back ----------------------------
  double rl_y = 0.0, rl_p2 = 0.0, rl_p3 = 0.0;  // from the left wavefront: its Y(q)[col 3], Y'(q-1)[cols 2, 3]
  double rr_y = 0.0, rr_p0 = 0.0, rr_p1 = 0.0;  // from the right wavefront: its Y(q)[col 0], Y'(q-1)[cols 0, 1]
  if (NW > 1) {
    double* mine = ex + ((q & 1) * NW + wave) * R2L_CHAIN_EX;
    if (lane == 0) {
      mine[0] = yq[0];
      mine[1] = yp[0];
      mine[2] = yp[1];
    }
    if (lane == 63) {
      mine[3] = yq[3];
      mine[4] = yp[2];
      mine[5] = yp[3];
    }
    R2L_LDS_BARRIER();
    if (lane == 0 && wave > 0) {
      const double* o = mine - R2L_CHAIN_EX;
      rl_y = o[3];
      rl_p2 = o[4];
      rl_p3 = o[5];
    }
    if (lane == 63 && wave < NW - 1) {
      const double* o = mine + R2L_CHAIN_EX;
      rr_y = o[0];
      rr_p0 = o[1];
      rr_p1 = o[2];
    }
  }
  (void)rl_p2;
  (void)rr_p1;
  // neighbours of Y(q) for the next step's sharpen (zero outside the image)
  {
    const double l = r2l_wave_shr1_d(yq[3], rl_y), r = r2l_wave_shl1_d(yq[0], rr_y);
    st.yl = le ? 0.0 : l;
    st.yrr = re ? 0.0 : r;
  }
  if (DN == 1) {
    // ---- median_filter(Y', 3), scipy 'reflect' = symmetric rows and columns: row q-1 joins the window ----------
    if ((unsigned)(q - 1) < (unsigned)H) {
      const double l1 = r2l_wave_shr1_d(yp[3], rl_p3), r1 = r2l_wave_shl1_d(yp[0], rr_p0);
      double* m = st.hb[(K + 2) % 3];  // slot of row q-1
      m[0] = le ? yp[0] : l1;
      m[1] = yp[0];
      m[2] = yp[1];
      m[3] = yp[2];
      m[4] = yp[3];
      m[5] = re ? yp[3] : r1;
    }
    const int y = q - 2;  // rows y-1, y, y+1 sit in slots K % 3, (K + 1) % 3, (K + 2) % 3
    if (y >= y0 && y < y1) {
      const double* r0 = st.hb[K % 3];
      const double* r1 = st.hb[(K + 1) % 3];
      const double* r2 = st.hb[(K + 2) % 3];
      // columns sorted once (lo <= mid <= hi), then per pixel: median(max of the los, median of the mids, min of
      // the his) -- the median of 9 as a selection, bit-identical to sorting
      double lo[6], mi[6], hi[6];
      const bool top = y == 0, bot = y == H - 1;  // the row mirrored in from outside is the edge row itself
      R2L_PRAGMA_UNROLL
      for (int j = 0; j < 6; ++j) {
        double p = top ? r1[j] : r0[j], c = r1[j], n = bot ? r1[j] : r2[j];
        r2l_chain_cswap(p, c);
        r2l_chain_cswap(c, n);
        r2l_chain_cswap(p, c);
        lo[j] = p;
        mi[j] = c;
        hi[j] = n;
      }
      const r2l_d2* f = fifo + (size_t)((y + 2 * FR) % FR) * 4 * 64 + lane;
      float x[3][4];
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 4; ++c) {
        const double a0 = fmax(fmax(lo[c], lo[c + 1]), lo[c + 2]);
        const double b0 = r2l_chain_med3(mi[c], mi[c + 1], mi[c + 2]);
        const double c0 = fmin(fmin(hi[c], hi[c + 1]), hi[c + 2]);
        const double yy = r2l_chain_med3(a0, b0, c0);
        const r2l_d2 uv = f[c * 64];
        R2L_PRAGMA_UNROLL
        for (int k = 0; k < 3; ++k) {
          const double rgb = fma(a.M2[k * 3], yy, fma(a.M2[k * 3 + 1], uv.x, a.M2[k * 3 + 2] * uv.y));
          x[k][c] = r2l_clip_gamma(rgb, a.inv_gamma);
        }
      }
      r2l_static_normalize<4>(a, x);
      if (store_ok) {
        const size_t off = (size_t)y * a_.W + x0;
        R2L_PRAGMA_UNROLL
        for (int k = 0; k < 3; ++k) {
          r2l_f4 s4;
          s4.x = x[k][0];
          s4.y = x[k][1];
          s4.z = x[k][2];
          s4.w = x[k][3];
          r2l_stream_store_f4(outb + (size_t)k * plane + off, s4);
        }
      }
    }
    return;
  }
  // ---- Hb(q-1): horizontal pass of gaussian_filter over Y'(q-1), scipy 'reflect' = symmetric columns --------
  if ((unsigned)(q - 1) < (unsigned)H) {
    double e[8];
    const double l2 = r2l_wave_shr1_d(yp[2], rl_p2), l1 = r2l_wave_shr1_d(yp[3], rl_p3);
    const double r1 = r2l_wave_shl1_d(yp[0], rr_p0), r2 = r2l_wave_shl1_d(yp[1], rr_p1);
    e[0] = le ? yp[1] : l2;
    e[1] = le ? yp[0] : l1;
    e[2] = yp[0];
    e[3] = yp[1];
    e[4] = yp[2];
    e[5] = yp[3];
    e[6] = re ? yp[3] : r1;
    e[7] = re ? yp[2] : r2;
    double* h = st.hb[(K + 5) % (DN == 0 ? 6 : 3)];  // slot of row q-1
    R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c)
      h[c] = fma(a.gk[2], e[c + 2], fma(a.gk[1], e[c + 1] + e[c + 3], a.gk[0] * (e[c] + e[c + 4])));
  }
  // ---- output row y = q-3: vertical pass over Hb(y-2 .. y+2), symmetric rows; chroma from the ring ---------
  const int y = q - 3;
  if (y >= y0 && y < y1) {
    // weights of the 5 window rows: a row outside the image gives its weight to its mirror image
    double wv[5];
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 5; ++k) wv[k] = a.gk[k];
    if (y < 2 || y > H - 3) {
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < 5; ++k) wv[k] = ((unsigned)(y + k - 2) < (unsigned)H) ? a.gk[k] : 0.0;
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < 5; ++k) {
        const int r = y + k - 2;
        if ((unsigned)r >= (unsigned)H) {
          const int t = r2l_symmetric(r, H) - (y - 2);  // window slot of the mirror image
          R2L_PRAGMA_UNROLL
          for (int j = 0; j < 5; ++j) wv[j] += (j == t) ? a.gk[k] : 0.0;
        }
      }
    }
    // rows y-2..y+2 sit in slots (K + 1 .. K + 5) % 6  (row q-1 = y+2 is slot (K+5)%6)
    const double* h0 = st.hb[(K + 1) % (DN == 0 ? 6 : 3)];
    const double* h1 = st.hb[(K + 2) % (DN == 0 ? 6 : 3)];
    const double* h2 = st.hb[(K + 3) % (DN == 0 ? 6 : 3)];
    const double* h3 = st.hb[(K + 4) % (DN == 0 ? 6 : 3)];
    const double* h4 = st.hb[(K + 5) % (DN == 0 ? 6 : 3)];
    const r2l_d2* f = fifo + (size_t)((y + 2 * FR) % FR) * 4 * 64 + lane;
    float x[3][4];
    R2L_PRAGMA_UNROLL
    for (int c = 0; c < 4; ++c) {
      const double yy = fma(wv[0], h0[c], fma(wv[1], h1[c], fma(wv[2], h2[c], fma(wv[3], h3[c], wv[4] * h4[c]))));
      const r2l_d2 uv = f[c * 64];
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < 3; ++k) {
        const double rgb = fma(a.M2[k * 3], yy, fma(a.M2[k * 3 + 1], uv.x, a.M2[k * 3 + 2] * uv.y));
        x[k][c] = r2l_clip_gamma(rgb, a.inv_gamma);
      }
    }
    r2l_static_normalize<4>(a, x);
    if (store_ok) {
      const size_t off = (size_t)y * a_.W + x0;
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < 3; ++k) {
        r2l_f4 s4;
        s4.x = x[k][0];
        s4.y = x[k][1];
        s4.z = x[k][2];
        s4.w = x[k][3];
        r2l_stream_store_f4(outb + (size_t)k * plane + off, s4);
      }
    }
  }
}

template <int RAWK, int DEB, int SH, int DN>
R2L_BLOCKFN void r2l_static_chain_block(const R2LStaticChainArgs& ca, int bid, int nblk, float* lds_f) {
  (void)nblk;
  const R2LStaticArgs& a = ca.s;
  const int NW = ca.nw;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double* ex = (double*)lds_f;
  r2l_d2* fifo = (r2l_d2*)(ex + 2 * NW * R2L_CHAIN_EX) + (size_t)wave * (R2L_CHAIN_FIFO_DOUBLES(SH) / 2);
  const int band = bid % ca.nband, b = bid / ca.nband;
  const int y0 = band * ca.band_h;
  const int y1 = (y0 + ca.band_h < a.H) ? y0 + ca.band_h : a.H;
  const int xs = wave * 256 + 4 * lane;
  const bool store_ok = xs < a.W;
  const int x0 = store_ok ? xs : a.W - 4;  // lanes beyond the image edge shadow the last column group
  const bool le = x0 == 0, re = x0 + 4 >= a.W;
  const size_t plane = (size_t)a.H * a.W;
  const size_t img = (size_t)b * plane;
  float* outb = a.out + (size_t)b * 3 * plane;
  R2LChainState<DEB, SH, DN> st;
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < (DN == 0 ? 6 : 3); ++i)
    R2L_PRAGMA_UNROLL
  for (int c = 0; c < (DN == 0 ? 4 : 6); ++c) st.hb[i][c] = 0.0;  // rows outside the image are never written: they stay finite
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < (SH ? 9 : 3); ++i)
    R2L_PRAGMA_UNROLL
  for (int c = 0; c < 4; ++c) st.yr[i][c] = 0.0;
  st.yl = st.yrr = 0.0;
  // Output row y = q - LAG.  First luma row computed: q0 <= y0 - LAG (Gaussian: Hb(y0-2) needs Y'(y0-2) needs
  // Y(y0-3); median: Y'(y0-1) needs Y(y0-2); unsharp_masking: 3 rows more on either side), rounded down to a multiple
  // of 6 so that q mod 6 is the unroll position; last: y1 + LAG - 1
  constexpr int LAG = (DN == 0 ? 3 : 2) + (SH ? 3 : 0);
  constexpr int LA = DEB ? 2 : 1;  // raw rows the demosaic looks ahead
  constexpr int NR = DEB ? 6 : 3;
  int q0 = y0 - LAG;
  q0 = (q0 >= 0) ? q0 - q0 % 6 : -(((-q0) + 5) / 6) * 6;
  const int q1 = y1 + LAG;  // exclusive
  constexpr bool LANES = R2L_HAVE_LANE_SHIFTS;
  constexpr int PF = DEB ? 2 : R2L_CHAIN_PF;
  R2LRowStageT<RAWK> stage;
  // BRANCH-FREE row loop on float32 / 16-bit frames (R2L_CHAIN_BF; r2l_stream_fetch_row_bf): every fetch unconditional
  // (rows past the band's last needed one re-fetch that row), every group of 6 steps in full (a step past q1 only finishes
  // rows >= y1, which are not stored) -- hipcc then counts its vmcnt waits instead of waiting for the row just requested
  constexpr bool BF = R2L_CHAIN_BF && RAWK != R2L_RAW_F64;
  constexpr unsigned ESZ = (RAWK == R2L_RAW_U16) ? 2u : 4u;
  const unsigned xo = ESZ * (unsigned)x0, xleft = ESZ * (unsigned)(x0 + (le ? 0 : -2)), xright = ESZ * (unsigned)(x0 + (re ? 2 : 4));
  const unsigned xl = LANES ? (lane < 32 ? xleft : xright) : xleft, xr = xright;
  const int rlast = q1 - 1 + LA;  // last raw row (before the symmetric extension) the band's steps consume
  (void)xo; (void)xl; (void)xr; (void)rlast;
#define R2L_CHAIN_FETCH(row_, st_)                                                                                \
  {                                                                                                               \
    if constexpr (BF) {                                                                                           \
      const int rr_ = ((row_) < rlast) ? (row_) : rlast;                                                          \
      r2l_stream_fetch_row_bf<RAWK, LANES>(a, img, r2l_symmetric(rr_, a.H), xo, xl, xr, le, re, (st_));           \
    } else {                                                                                                      \
      r2l_stream_fetch_row<RAWK, LANES>(a, img, r2l_symmetric((row_), a.H), x0, le, re, (st_));                   \
    }                                                                                                             \
  }
  // warm-up: raw rows q0-LA .. q0+LA-1 into their slots (q0 is a multiple of 6, hence of NR)
  R2L_PRAGMA_UNROLL
  for (int i = -LA; i < LA; ++i) {
    R2L_CHAIN_FETCH(q0 + i, stage)
    r2l_stream_convert_row<RAWK, LANES>(a, stage, le, re, st.rw[(i + NR) % NR]);
  }
  static_assert(6 % PF == 0, "the prefetch ring is indexed by the unroll position");
  R2LRowStageT<RAWK> pf[PF];  // ring: step K consumes pf[K % PF] (raw row q + LA) and refills it with row q + LA + PF
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < PF; ++i) R2L_CHAIN_FETCH(q0 + LA + i, pf[i])
  for (int qb = q0; qb < q1; qb += 6) {
    R2L_CHAIN_PRIO(qb - q0, q1 - q0);
#define R2L_CHAIN_STEP(K)                                                                                         \
  if (BF || qb + K < q1) {                                                                                        \
    const int q = qb + K;                                                                                         \
    r2l_stream_convert_row<RAWK, LANES>(a, pf[K % PF], le, re, st.rw[(K + LA) % NR]);                             \
    if (BF || q + PF < q1) R2L_CHAIN_FETCH(q + LA + PF, pf[K % PF])                                               \
    r2l_chain_step<DEB, SH, DN, K>(a, st, q, y0, y1, le, re, NW, wave, lane, ex, fifo, outb, plane, x0, store_ok); \
  }
    R2L_CHAIN_STEP(0)
    R2L_CHAIN_STEP(1)
    R2L_CHAIN_STEP(2)
    R2L_CHAIN_STEP(3)
    R2L_CHAIN_STEP(4)
    R2L_CHAIN_STEP(5)
#undef R2L_CHAIN_STEP
  }
#undef R2L_CHAIN_FETCH
}

#endif  // !R2L_SERIAL
