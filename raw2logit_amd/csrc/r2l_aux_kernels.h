// r2l_aux_kernels.h -- adversarial auxiliary losses between the outputs of two processors.
//
//   SSIM(window_size=11)(img1, img2), size_average=True          utils/ssim.py:9-39, train.py:261-262
//   l2_regularization(x, y) = ((x - y) ** 2).sum()               utils/base.py:342-343
//   (AuxLoss: img1 = default processor's output under no_grad, img2 = the adversarial processor's output,
//    utils/base.py:346-358)
//
// SSIM needs five 11x11 Gaussian-windowed moments per pixel (zero padding, depthwise):
//   mu1 = w*x, mu2 = w*y, E11 = w*x^2, E22 = w*y^2, E12 = w*xy
//   S = (2 mu1 mu2 + C1)(2 (E12 - mu1 mu2) + C2) / ((mu1^2 + mu2^2 + C1)(E11 - mu1^2 + E22 - mu2^2 + C2))
// The window is the outer product of a normalised 1-D Gaussian (sigma 1.5), so each moment is a horizontal
// 11-tap pass followed by a vertical one.  One workgroup owns a 64x64 tile of one (image, channel) plane and a
// 74x74 frame (halo 5): x and y frames -> LDS, horizontal pass of the 5 products -> LDS, vertical pass + the
// SSIM formula in registers.  8 B per element in, nothing out (forward) -- the ~230 FMA per element bound it.
// Backward w.r.t. img2: dS/d(mu2, E22, E12) at every pixel (same two passes), stored as three planes D, and
//   grad(q) = gscale * [ (w*D_mu)(q) + 2 y(q) (w*D_22)(q) + x(q) (w*D_12)(q) ]        (w symmetric)
// by a second kernel of the same shape.
#pragma once
#include "r2l_param_kernels.h"

#define R2L_SSIM_K 11
#define R2L_SSIM_R 5
#define R2L_SSIM_T 64                                  // tile edge
#define R2L_SSIM_F (R2L_SSIM_T + 2 * R2L_SSIM_R)       // frame edge 74
#define R2L_SSIM_FS 76                                 // frame row stride (pairs / floats)
#define R2L_SSIM_NLOAD ((R2L_SSIM_F * R2L_SSIM_F + R2L_NT - 1) / R2L_NT)  // frame elements per lane
// LDS (floats): XY frame of (x,y) pairs; horizontal sums as pairs (h1,h2), (h11,h22) and the plane h12
#define R2L_SSIM_LDS_FLOATS (2 * R2L_SSIM_F * R2L_SSIM_FS + 5 * R2L_SSIM_F * R2L_SSIM_T)
#define R2L_SSIM_BWD_LDS_FLOATS (3 * R2L_SSIM_F * R2L_SSIM_FS + 3 * R2L_SSIM_F * R2L_SSIM_T)

struct R2LSsimArgs {
  const float* img1;  // (B,C,H,W) reference output (no gradient)
  const float* img2;  // (B,C,H,W) adversarial output
  float g[R2L_SSIM_K + 1];  // normalised 1-D Gaussian, float32 (utils/ssim.py:9-11)
  float* partial;     // [1][nblk] partial sums of the SSIM map
  float* dmaps;       // B*C*H*W pairs (D_mu, D_22), then the plane D_12
  int nplanes, H, W, mode;  // mode bit 0: sum the SSIM map into `partial`; bit 1: write the D maps
};

// frame elements of one lane, fetched one tile ahead (written to LDS at the start of the tile's turn)
struct R2LSsimPre {
  r2l_f2 v[R2L_SSIM_NLOAD];
  float dx[R2L_SSIM_NLOAD];
};
struct R2LSsimTile {
  int pl, oy, ox;
};
R2L_HD R2LSsimTile r2l_ssim_tile(int t, int ntx, int nty) {
  R2LSsimTile q;
  q.pl = t / (ntx * nty);
  const int r = t - q.pl * (ntx * nty);
  q.oy = (r / ntx) * R2L_SSIM_T;
  q.ox = (r % ntx) * R2L_SSIM_T;
  return q;
}
// MODE 0: pairs (img1, img2); MODE 1: pairs from p2 (D_mu, D_22) + scalars from p1 (D_12); zero outside the image
template <int MODE>
R2L_HD void r2l_ssim_fetch(int tid, const float* p1, const float* p2, const R2LSsimTile& q, int H, int W,
                           R2LSsimPre& pre) {
  const size_t hw = (size_t)H * W;
  R2L_PRAGMA_UNROLL
  for (int it = 0; it < R2L_SSIM_NLOAD; ++it) {
    const int i = tid + it * R2L_NT;
    const int fy = i / R2L_SSIM_F, fx = i - fy * R2L_SSIM_F;
    const int gy = q.oy - R2L_SSIM_R + fy, gx = q.ox - R2L_SSIM_R + fx;
    pre.v[it].x = pre.v[it].y = 0.f;
    if (MODE == 1) pre.dx[it] = 0.f;
    if (i < R2L_SSIM_F * R2L_SSIM_F && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
      const size_t e = (size_t)q.pl * hw + (size_t)gy * W + gx;
      if (MODE == 0) {
        pre.v[it].x = p1[e];
        pre.v[it].y = p2[e];
      } else {
        pre.v[it] = ((const r2l_f2*)p2)[e];
        pre.dx[it] = p1[e];
      }
    }
  }
}
template <int MODE>
R2L_HD void r2l_ssim_park(int tid, const R2LSsimPre& pre, r2l_f2* P, float* X) {
  R2L_PRAGMA_UNROLL
  for (int it = 0; it < R2L_SSIM_NLOAD; ++it) {
    const int i = tid + it * R2L_NT;
    const int fy = i / R2L_SSIM_F, fx = i - fy * R2L_SSIM_F;
    if (i < R2L_SSIM_F * R2L_SSIM_F) {
      P[fy * R2L_SSIM_FS + fx] = pre.v[it];
      if (MODE == 1) X[fy * R2L_SSIM_FS + fx] = pre.dx[it];
    }
  }
}

// Two values that share their window weights travel as one pair through v_pk_fma_f32: (x, y) -> (mu1, mu2),
// (x^2, y^2) -> (E11, E22); x*y -> E12 stays scalar.  3 instructions per tap and output instead of 5.
R2L_BLOCKFN void r2l_ssim_block(const R2LSsimArgs& a, int bid, int nblk, float* lds) {
  r2l_f2* XY = (r2l_f2*)lds;                                    // [F][FS] pairs (x, y)
  r2l_f2* H12 = XY + R2L_SSIM_F * R2L_SSIM_FS;                  // [F][T] pairs (h1, h2)
  r2l_f2* HSQ = H12 + R2L_SSIM_F * R2L_SSIM_T;                  // [F][T] pairs (h11, h22)
  float* HX = (float*)(HSQ + R2L_SSIM_F * R2L_SSIM_T);          // [F][T] h12
  const int ntx = (a.W + R2L_SSIM_T - 1) / R2L_SSIM_T, nty = (a.H + R2L_SSIM_T - 1) / R2L_SSIM_T;
  const int ntiles = a.nplanes * ntx * nty;
  const size_t hw = (size_t)a.H * a.W, np = (size_t)a.nplanes * hw;
  R2L_TREG_DECL(R2LAcc6, regs);
  R2L_TREG_DECL(R2LSsimPre, pre);
  R2L_PHASE_BEGIN
  R2L_TREG(regs).acc[0] = 0.f;
  if (bid < ntiles) r2l_ssim_fetch<0>(tid, a.img1, a.img2, r2l_ssim_tile(bid, ntx, nty), a.H, a.W, R2L_TREG(pre));
  R2L_PHASE_END
  for (int t = bid; t < ntiles; t += nblk) {
    const R2LSsimTile q = r2l_ssim_tile(t, ntx, nty);
    const int pl = q.pl, oy = q.oy, ox = q.ox;
    R2L_PHASE_BEGIN
    r2l_ssim_park<0>(tid, R2L_TREG(pre), XY, nullptr);
    R2L_PHASE_END
    R2L_PHASE_BEGIN  // next tile's frame: in flight during both passes
    if (t + nblk < ntiles)
      r2l_ssim_fetch<0>(tid, a.img1, a.img2, r2l_ssim_tile(t + nblk, ntx, nty), a.H, a.W, R2L_TREG(pre));
    for (int i = tid; i < R2L_SSIM_F * (R2L_SSIM_T / 4); i += R2L_NT) {
      const int fy = i / (R2L_SSIM_T / 4), c0 = 4 * (i - fy * (R2L_SSIM_T / 4));
      r2l_p2 xy[R2L_SSIM_K + 3], sq[R2L_SSIM_K + 3];
      float xm[R2L_SSIM_K + 3];
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < R2L_SSIM_K + 3; k += 2) {  // 128-bit reads of two pairs (64-bit ones: 8-way bank conflict)
        const r2l_f4 v = r2l_lds_f4((const float*)(XY + fy * R2L_SSIM_FS + c0 + k));
        xy[k] = r2l_mk2(v.x, v.y);
        xy[k + 1] = r2l_mk2(v.z, v.w);
        sq[k] = r2l_pmul(xy[k], xy[k]);
        sq[k + 1] = r2l_pmul(xy[k + 1], xy[k + 1]);
        xm[k] = v.x * v.y;
        xm[k + 1] = v.z * v.w;
      }
      r2l_p2 h[4], hs[4];
      float hx[4];
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 4; ++c) {
        h[c] = hs[c] = r2l_splat2(0.f);
        hx[c] = 0.f;
        R2L_PRAGMA_UNROLL
        for (int k = 0; k < R2L_SSIM_K; ++k) {
          const r2l_p2 w = r2l_splat2(a.g[k]);
          h[c] = r2l_pfma(w, xy[c + k], h[c]);
          hs[c] = r2l_pfma(w, sq[c + k], hs[c]);
          hx[c] = fmaf(a.g[k], xm[c + k], hx[c]);
        }
      }
      const int o = fy * R2L_SSIM_T + c0;
      r2l_f4 st;
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 4; c += 2) {
        st.x = h[c][0], st.y = h[c][1], st.z = h[c + 1][0], st.w = h[c + 1][1];
        *(r2l_f4*)(H12 + o + c) = st;
        st.x = hs[c][0], st.y = hs[c][1], st.z = hs[c + 1][0], st.w = hs[c + 1][1];
        *(r2l_f4*)(HSQ + o + c) = st;
      }
      st.x = hx[0], st.y = hx[1], st.z = hx[2], st.w = hx[3];
      *(r2l_f4*)(HX + o) = st;
    }
    R2L_PHASE_END
    R2L_PHASE_BEGIN  // vertical pass + SSIM: one column x 8 rows per lane
    {
      const int c = tid & 63, r0 = 8 * (tid >> 6);
      r2l_p2 mu[8], ee[8];
      float exy[8];
      {
        r2l_p2 col[R2L_SSIM_K + 7];
        R2L_PRAGMA_UNROLL
        for (int k = 0; k < R2L_SSIM_K + 7; ++k) {
          const r2l_f2 v = H12[(r0 + k) * R2L_SSIM_T + c];
          col[k] = r2l_mk2(v.x, v.y);
        }
        R2L_PRAGMA_UNROLL
        for (int rr = 0; rr < 8; ++rr) {
          r2l_p2 sacc = r2l_splat2(0.f);
          R2L_PRAGMA_UNROLL
          for (int k = 0; k < R2L_SSIM_K; ++k) sacc = r2l_pfma(r2l_splat2(a.g[k]), col[rr + k], sacc);
          mu[rr] = sacc;
        }
        R2L_PRAGMA_UNROLL
        for (int k = 0; k < R2L_SSIM_K + 7; ++k) {
          const r2l_f2 v = HSQ[(r0 + k) * R2L_SSIM_T + c];
          col[k] = r2l_mk2(v.x, v.y);
        }
        R2L_PRAGMA_UNROLL
        for (int rr = 0; rr < 8; ++rr) {
          r2l_p2 sacc = r2l_splat2(0.f);
          R2L_PRAGMA_UNROLL
          for (int k = 0; k < R2L_SSIM_K; ++k) sacc = r2l_pfma(r2l_splat2(a.g[k]), col[rr + k], sacc);
          ee[rr] = sacc;
        }
      }
      {
        float col[R2L_SSIM_K + 7];
        R2L_PRAGMA_UNROLL
        for (int k = 0; k < R2L_SSIM_K + 7; ++k) col[k] = HX[(r0 + k) * R2L_SSIM_T + c];
        R2L_PRAGMA_UNROLL
        for (int rr = 0; rr < 8; ++rr) {
          float sacc = 0.f;
          R2L_PRAGMA_UNROLL
          for (int k = 0; k < R2L_SSIM_K; ++k) sacc = fmaf(a.g[k], col[rr + k], sacc);
          exy[rr] = sacc;
        }
      }
      const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
      R2L_PRAGMA_UNROLL
      for (int rr = 0; rr < 8; ++rr) {
        const int gy = oy + r0 + rr, gx = ox + c;
        if (gy < a.H && gx < a.W) {
          const float mu1 = mu[rr][0], mu2 = mu[rr][1];
          const float s11 = ee[rr][0] - mu1 * mu1, s22 = ee[rr][1] - mu2 * mu2, s12 = exy[rr] - mu1 * mu2;
          const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2;
          const float B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
          const float ib1 = r2l_rcp(B1), ib2 = r2l_rcp(B2);  // B1 >= C1, B2 >= C2 - rounding: well away from 0
          const float ib = ib1 * ib2;
          const float S = A1 * A2 * ib;
          if (a.mode & 1) R2L_TREG(regs).acc[0] += S;
          if (a.mode & 2) {
            const float dA1 = A2 * ib, dA2 = A1 * ib, dB1 = -S * ib1, dB2 = -S * ib2;
            const size_t e = (size_t)pl * hw + (size_t)gy * a.W + gx;
            r2l_f2 d;
            d.x = 2.f * mu1 * (dA1 - dA2) + 2.f * mu2 * (dB1 - dB2);  // dS/d mu2
            d.y = dB2;                                                // dS/d E[y^2]
            ((r2l_f2*)a.dmaps)[e] = d;
            a.dmaps[2 * np + e] = 2.f * dA2;                          // dS/d E[xy]
          }
        }
      }
    }
    R2L_PHASE_END
  }
  if (a.mode & 1) {
    R2L_BLOCK_REDUCE(1, regs, lds, a.partial, bid, nblk)
  }
}

// grad(q) = gscale * [ (w*D_mu)(q) + 2 y(q) (w*D_22)(q) + x(q) (w*D_12)(q) ]
struct R2LSsimBwdArgs {
  const float* img1;
  const float* img2;
  const float* dmaps;   // pairs (D_mu, D_22), then the plane D_12
  const float* gup;     // device scalar: upstream gradient d loss / d mean-SSIM
  float scale;          // 1 / number of elements
  float* grad;          // (B,C,H,W)
  float g[R2L_SSIM_K + 1];
  int nplanes, H, W;
};
R2L_BLOCKFN void r2l_ssim_bwd_block(const R2LSsimBwdArgs& a, int bid, int nblk, float* lds) {
  r2l_f2* DP = (r2l_f2*)lds;                                    // [F][FS] pairs (D_mu, D_22)
  float* DX = (float*)(DP + R2L_SSIM_F * R2L_SSIM_FS);          // [F][FS] D_12
  r2l_f2* HP = (r2l_f2*)(DX + R2L_SSIM_F * R2L_SSIM_FS);        // [F][T] horizontal sums, pairs
  float* HX = (float*)(HP + R2L_SSIM_F * R2L_SSIM_T);           // [F][T]
  const int ntx = (a.W + R2L_SSIM_T - 1) / R2L_SSIM_T, nty = (a.H + R2L_SSIM_T - 1) / R2L_SSIM_T;
  const int ntiles = a.nplanes * ntx * nty;
  const size_t hw = (size_t)a.H * a.W, np = (size_t)a.nplanes * hw;
  R2L_TREG_DECL(R2LSsimPre, pre);
  R2L_PHASE_BEGIN
  if (bid < ntiles)
    r2l_ssim_fetch<1>(tid, a.dmaps + 2 * np, a.dmaps, r2l_ssim_tile(bid, ntx, nty), a.H, a.W, R2L_TREG(pre));
  R2L_PHASE_END
  for (int t = bid; t < ntiles; t += nblk) {
    const R2LSsimTile q = r2l_ssim_tile(t, ntx, nty);
    const int pl = q.pl, oy = q.oy, ox = q.ox;
    R2L_PHASE_BEGIN
    r2l_ssim_park<1>(tid, R2L_TREG(pre), DP, DX);
    R2L_PHASE_END
    R2L_PHASE_BEGIN
    if (t + nblk < ntiles)
      r2l_ssim_fetch<1>(tid, a.dmaps + 2 * np, a.dmaps, r2l_ssim_tile(t + nblk, ntx, nty), a.H, a.W, R2L_TREG(pre));
    for (int i = tid; i < R2L_SSIM_F * (R2L_SSIM_T / 4); i += R2L_NT) {
      const int fy = i / (R2L_SSIM_T / 4), c0 = 4 * (i - fy * (R2L_SSIM_T / 4));
      r2l_p2 dp[R2L_SSIM_K + 3];
      float dx[R2L_SSIM_K + 5];
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < R2L_SSIM_K + 3; k += 2) {
        const r2l_f4 v = r2l_lds_f4((const float*)(DP + fy * R2L_SSIM_FS + c0 + k));
        dp[k] = r2l_mk2(v.x, v.y);
        dp[k + 1] = r2l_mk2(v.z, v.w);
      }
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < R2L_SSIM_K + 5; k += 4) {
        const r2l_f4 v = r2l_lds_f4(DX + fy * R2L_SSIM_FS + c0 + k);
        dx[k] = v.x, dx[k + 1] = v.y, dx[k + 2] = v.z, dx[k + 3] = v.w;
      }
      r2l_p2 h[4];
      float hx[4];
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 4; ++c) {
        h[c] = r2l_splat2(0.f);
        hx[c] = 0.f;
        R2L_PRAGMA_UNROLL
        for (int k = 0; k < R2L_SSIM_K; ++k) {
          h[c] = r2l_pfma(r2l_splat2(a.g[k]), dp[c + k], h[c]);
          hx[c] = fmaf(a.g[k], dx[c + k], hx[c]);
        }
      }
      const int o = fy * R2L_SSIM_T + c0;
      r2l_f4 st;
      st.x = h[0][0], st.y = h[0][1], st.z = h[1][0], st.w = h[1][1];
      *(r2l_f4*)(HP + o) = st;
      st.x = h[2][0], st.y = h[2][1], st.z = h[3][0], st.w = h[3][1];
      *(r2l_f4*)(HP + o + 2) = st;
      st.x = hx[0], st.y = hx[1], st.z = hx[2], st.w = hx[3];
      *(r2l_f4*)(HX + o) = st;
    }
    R2L_PHASE_END
    R2L_PHASE_BEGIN
    {
      const int c = tid & 63, r0 = 8 * (tid >> 6);
      float xv[8], yv[8];  // the images at this lane's 8 outputs: loaded up front (the stores below may alias)
      R2L_PRAGMA_UNROLL
      for (int rr = 0; rr < 8; ++rr) {
        const int gy = oy + r0 + rr, gx = ox + c;
        xv[rr] = yv[rr] = 0.f;
        if (gy < a.H && gx < a.W) {
          const size_t e = (size_t)pl * hw + (size_t)gy * a.W + gx;
          xv[rr] = a.img1[e];
          yv[rr] = a.img2[e];
        }
      }
      r2l_p2 colp[R2L_SSIM_K + 7];
      float colx[R2L_SSIM_K + 7];
      R2L_PRAGMA_UNROLL
      for (int k = 0; k < R2L_SSIM_K + 7; ++k) {
        const r2l_f2 v = HP[(r0 + k) * R2L_SSIM_T + c];
        colp[k] = r2l_mk2(v.x, v.y);
        colx[k] = HX[(r0 + k) * R2L_SSIM_T + c];
      }
      const float gs = *a.gup * a.scale;
      R2L_PRAGMA_UNROLL
      for (int rr = 0; rr < 8; ++rr) {
        r2l_p2 sp = r2l_splat2(0.f);
        float sx = 0.f;
        R2L_PRAGMA_UNROLL
        for (int k = 0; k < R2L_SSIM_K; ++k) {
          sp = r2l_pfma(r2l_splat2(a.g[k]), colp[rr + k], sp);
          sx = fmaf(a.g[k], colx[rr + k], sx);
        }
        const int gy = oy + r0 + rr, gx = ox + c;
        if (gy < a.H && gx < a.W) {
          const size_t e = (size_t)pl * hw + (size_t)gy * a.W + gx;
          a.grad[e] = gs * fmaf(xv[rr], sx, fmaf(2.f * yv[rr], sp[1], sp[0]));
        }
      }
    }
    R2L_PHASE_END
  }
}

// ---- l2_regularization: sum (x - y)^2 and its gradient 2 (y - x) * gscale ---------------------------------
struct R2LL2Args {
  const float* x;
  const float* y;
  const float* gup;  // backward: device scalar upstream gradient (null: forward)
  float* grad;
  float* partial;
  size_t n;  // multiple of 4
};
R2L_BLOCKFN void r2l_l2_block(const R2LL2Args& a, int bid, int nblk, float* lds) {
  R2L_TREG_DECL(R2LAcc6, regs);
  R2L_PHASE_BEGIN
  R2L_TREG(regs).acc[0] = 0.f;
  const float gs = a.gup ? *a.gup : 0.f;
  for (size_t i4 = (size_t)bid * R2L_NT + tid; i4 < a.n / 4; i4 += (size_t)nblk * R2L_NT) {
    const r2l_f4 xv = *(const r2l_f4*)(a.x + 4 * i4);
    const r2l_f4 yv = *(const r2l_f4*)(a.y + 4 * i4);
    const float d0 = xv.x - yv.x, d1 = xv.y - yv.y, d2 = xv.z - yv.z, d3 = xv.w - yv.w;
    if (a.grad) {
      r2l_f4 o;
      o.x = -2.f * d0 * gs;
      o.y = -2.f * d1 * gs;
      o.z = -2.f * d2 * gs;
      o.w = -2.f * d3 * gs;
      *(r2l_f4*)(a.grad + 4 * i4) = o;
    } else {
      R2L_TREG(regs).acc[0] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  R2L_PHASE_END
  if (a.partial) {
    R2L_BLOCK_REDUCE(1, regs, lds, a.partial, bid, nblk)
  }
}
