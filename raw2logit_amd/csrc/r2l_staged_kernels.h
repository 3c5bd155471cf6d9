// r2l_staged_kernels.h -- one kernel per stage of ParametrizedProcessing.forward, for track_stages=True.
//
// The reference materialises every stage tensor and keeps it in the autograd graph so that
// `stage.retain_grad()` / `stage.grad` work (pipeline_torch.py:197-221, model.py:249-254).  This path
// therefore cannot be fused; it trades speed for the per-stage tensors and also provides d/d raw.
// Kernels are plain one-lane-per-pixel streaming kernels (no LDS tiles); weight / statistic gradients use
// the same fixed-order workgroup reduction as the fused kernels.
//
//   conv33   Debayer: 3->3 channel 3x3 cross-correlation, mirror ('reflect') padding        :187, :228-237
//   mix3     einsum('bchw,kc->bkhw') with a 3x3 matrix (white balance as a diagonal one)     :190-194, :198-203
//   pconv    yuv[:, [0]] = conv(yuv[:, [0]]): KxK on channel 0, zero (K=3) or mirror (K=5) padding   :195, :202
//   clip / gamma / add / BatchNorm                                                            :206-217
#pragma once
#include "r2l_param_kernels.h"

struct R2LStageArgs {
  const float* x;    // forward input (B,3,H,W)
  const float* g;    // backward: gradient w.r.t. the stage output
  const float* w;    // weights / matrix / coefficients (device)
  const float* aux;  // op-specific extra input
  float* y;          // forward output / backward gradient w.r.t. the stage input
  float* partial;    // [nacc][nblk] reduction partials
  int B, H, W, K, pad_mirror;
};

R2L_HD void r2l_px_coords(size_t idx, int H, int W, int& b, int& yy, int& xx) {
  xx = (int)(idx % W);
  const size_t r = idx / W;
  yy = (int)(r % H);
  b = (int)(r / H);
}
// pre-images of in-image index q under the mirror extension with pad P: q itself, -q (1 <= q <= P) and
// 2(n-1)-q (n-1-P <= q <= n-2); -1 marks "none" (kept distinct from every valid coordinate by the caller)
template <int P>
R2L_HD void r2l_mirror_preimages(int q, int n, int e[3]) {
  e[0] = q;
  e[1] = (q >= 1 && q <= P) ? -q : -1000000;
  e[2] = (q >= n - 1 - P && q <= n - 2) ? 2 * (n - 1) - q : -1000000;
}

template <int N>
struct R2LAccN {
  float acc[N];
};

// ---- conv33 ------------------------------------------------------------------------------------------
R2L_BLOCKFN void r2l_conv33_fwd_block(const R2LStageArgs& a, int bid, int nblk, float* lds) {
  (void)lds;
  const size_t hw = (size_t)a.H * a.W, n = (size_t)a.B * hw;
  R2L_PHASE_BEGIN
  for (size_t idx = (size_t)bid * R2L_NT + tid; idx < n; idx += (size_t)nblk * R2L_NT) {
    int b, yy, xx;
    r2l_px_coords(idx, a.H, a.W, b, yy, xx);
    const float* xb = a.x + (size_t)b * 3 * hw;
    float o[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < 3; ++i) {
      const int sy = r2l_mirror(yy + i - 1, a.H);
      for (int j = 0; j < 3; ++j) {
        const int sx = r2l_mirror(xx + j - 1, a.W);
        for (int c = 0; c < 3; ++c) {
          const float v = xb[(size_t)c * hw + (size_t)sy * a.W + sx];
          for (int k = 0; k < 3; ++k) o[k] = fmaf(a.w[((k * 3 + c) * 3 + i) * 3 + j], v, o[k]);
        }
      }
    }
    for (int k = 0; k < 3; ++k) a.y[((size_t)b * 3 + k) * hw + (size_t)yy * a.W + xx] = o[k];
  }
  R2L_PHASE_END
}
// gradient w.r.t. the input (a.y) and the 81 weights (a.partial)
R2L_BLOCKFN void r2l_conv33_bwd_block(const R2LStageArgs& a, int bid, int nblk, float* lds) {
  const size_t hw = (size_t)a.H * a.W, n = (size_t)a.B * hw;
  R2L_TREG_DECL(R2LAccN<81>, regs);
  R2L_PHASE_BEGIN
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 81; ++i) R2L_TREG(regs).acc[i] = 0.f;
  for (size_t idx = (size_t)bid * R2L_NT + tid; idx < n; idx += (size_t)nblk * R2L_NT) {
    int b, yy, xx;
    r2l_px_coords(idx, a.H, a.W, b, yy, xx);
    const float* xb = a.x + (size_t)b * 3 * hw;
    const float* gb = a.g + (size_t)b * 3 * hw;
    // weight gradient: gW[k][c][t] += g[k](p) * x[c](mirror(p + t))
    float gp[3];
    for (int k = 0; k < 3; ++k) gp[k] = gb[(size_t)k * hw + (size_t)yy * a.W + xx];
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i) {
      const int sy = r2l_mirror(yy + i - 1, a.H);
      R2L_PRAGMA_UNROLL
      for (int j = 0; j < 3; ++j) {
        const int sx = r2l_mirror(xx + j - 1, a.W);
        R2L_PRAGMA_UNROLL
        for (int c = 0; c < 3; ++c) {
          const float v = xb[(size_t)c * hw + (size_t)sy * a.W + sx];
          R2L_PRAGMA_UNROLL
          for (int k = 0; k < 3; ++k)
            R2L_TREG(regs).acc[((k * 3 + c) * 3 + i) * 3 + j] =
                fmaf(gp[k], v, R2L_TREG(regs).acc[((k * 3 + c) * 3 + i) * 3 + j]);
        }
      }
    }
    // input gradient at q = (yy, xx): sum over taps t and pre-images q' of q of W[k][c][t] * g[k](q' - t)
    if (a.y) {
      int ey[3], ex[3];
      r2l_mirror_preimages<1>(yy, a.H, ey);
      r2l_mirror_preimages<1>(xx, a.W, ex);
      float o[3] = {0.f, 0.f, 0.f};
      for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) {
          if (ey[p] < -1 || ex[q] < -1) continue;
          for (int i = 0; i < 3; ++i) {
            const int py = ey[p] - (i - 1);
            if ((unsigned)py >= (unsigned)a.H) continue;
            for (int j = 0; j < 3; ++j) {
              const int px = ex[q] - (j - 1);
              if ((unsigned)px >= (unsigned)a.W) continue;
              for (int k = 0; k < 3; ++k) {
                const float gv = gb[(size_t)k * hw + (size_t)py * a.W + px];
                for (int c = 0; c < 3; ++c) o[c] = fmaf(a.w[((k * 3 + c) * 3 + i) * 3 + j], gv, o[c]);
              }
            }
          }
        }
      for (int c = 0; c < 3; ++c) a.y[((size_t)b * 3 + c) * hw + (size_t)yy * a.W + xx] = o[c];
    }
  }
  R2L_PHASE_END
  R2L_BLOCK_REDUCE(81, regs, lds, a.partial, bid, nblk)
}

// ---- mix3: y[k] = sum_c M[k][c] x[c] ------------------------------------------------------------------
R2L_BLOCKFN void r2l_mix3_fwd_block(const R2LStageArgs& a, int bid, int nblk, float* lds) {
  (void)lds;
  const size_t hw = (size_t)a.H * a.W, n = (size_t)a.B * hw;
  R2L_PHASE_BEGIN
  for (size_t idx = (size_t)bid * R2L_NT + tid; idx < n; idx += (size_t)nblk * R2L_NT) {
    const size_t b = idx / hw, p = idx - b * hw;
    const float* xb = a.x + b * 3 * hw + p;
    const float x0 = xb[0], x1 = xb[hw], x2 = xb[2 * hw];
    for (int k = 0; k < 3; ++k)
      a.y[(b * 3 + k) * hw + p] = a.w[k * 3] * x0 + a.w[k * 3 + 1] * x1 + a.w[k * 3 + 2] * x2;
  }
  R2L_PHASE_END
}
R2L_BLOCKFN void r2l_mix3_bwd_block(const R2LStageArgs& a, int bid, int nblk, float* lds) {
  const size_t hw = (size_t)a.H * a.W, n = (size_t)a.B * hw;
  R2L_TREG_DECL(R2LAccN<9>, regs);
  R2L_PHASE_BEGIN
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 9; ++i) R2L_TREG(regs).acc[i] = 0.f;
  for (size_t idx = (size_t)bid * R2L_NT + tid; idx < n; idx += (size_t)nblk * R2L_NT) {
    const size_t b = idx / hw, p = idx - b * hw;
    const float* gb = a.g + b * 3 * hw + p;
    const float g0 = gb[0], g1 = gb[hw], g2 = gb[2 * hw];
    if (a.y)
      for (int c = 0; c < 3; ++c) a.y[(b * 3 + c) * hw + p] = a.w[c] * g0 + a.w[3 + c] * g1 + a.w[6 + c] * g2;
    if (a.x) {
      const float* xb = a.x + b * 3 * hw + p;
      const float gg[3] = {g0, g1, g2};
      R2L_PRAGMA_UNROLL
      for (int c = 0; c < 3; ++c) {
        const float xv = xb[(size_t)c * hw];
        R2L_PRAGMA_UNROLL
        for (int k = 0; k < 3; ++k) R2L_TREG(regs).acc[k * 3 + c] = fmaf(gg[k], xv, R2L_TREG(regs).acc[k * 3 + c]);
      }
    }
  }
  R2L_PHASE_END
  R2L_BLOCK_REDUCE(9, regs, lds, a.partial, bid, nblk)
}

// ---- pconv: channel 0 <- KxK cross-correlation of channel 0 (zero or mirror padding), 1 and 2 copied ----
R2L_HD float r2l_pconv_sample(const float* pl, int yy, int xx, int H, int W, int mirror) {
  if (mirror) return pl[(size_t)r2l_mirror(yy, H) * W + r2l_mirror(xx, W)];
  return ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? pl[(size_t)yy * W + xx] : 0.f;
}
R2L_BLOCKFN void r2l_pconv_fwd_block(const R2LStageArgs& a, int bid, int nblk, float* lds) {
  (void)lds;
  const size_t hw = (size_t)a.H * a.W, n = (size_t)a.B * hw;
  const int K = a.K, R = K / 2;
  R2L_PHASE_BEGIN
  for (size_t idx = (size_t)bid * R2L_NT + tid; idx < n; idx += (size_t)nblk * R2L_NT) {
    int b, yy, xx;
    r2l_px_coords(idx, a.H, a.W, b, yy, xx);
    const float* xb = a.x + (size_t)b * 3 * hw;
    float s = 0.f;
    for (int i = 0; i < K; ++i)
      for (int j = 0; j < K; ++j)
        s = fmaf(a.w[i * K + j], r2l_pconv_sample(xb, yy + i - R, xx + j - R, a.H, a.W, a.pad_mirror), s);
    const size_t p = (size_t)yy * a.W + xx;
    float* yb = a.y + (size_t)b * 3 * hw;
    yb[p] = s;
    yb[hw + p] = xb[hw + p];
    yb[2 * hw + p] = xb[2 * hw + p];
  }
  R2L_PHASE_END
}
R2L_BLOCKFN void r2l_pconv_bwd_block(const R2LStageArgs& a, int bid, int nblk, float* lds) {
  const size_t hw = (size_t)a.H * a.W, n = (size_t)a.B * hw;
  const int K = a.K, R = K / 2;
  R2L_TREG_DECL(R2LAccN<25>, regs);
  R2L_PHASE_BEGIN
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 25; ++i) R2L_TREG(regs).acc[i] = 0.f;
  for (size_t idx = (size_t)bid * R2L_NT + tid; idx < n; idx += (size_t)nblk * R2L_NT) {
    int b, yy, xx;
    r2l_px_coords(idx, a.H, a.W, b, yy, xx);
    const float* xb = a.x + (size_t)b * 3 * hw;
    const float* gb = a.g + (size_t)b * 3 * hw;
    const size_t p = (size_t)yy * a.W + xx;
    const float g0 = gb[p];
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 5; ++i)
      R2L_PRAGMA_UNROLL
    for (int j = 0; j < 5; ++j)
      if (i < K && j < K)
        R2L_TREG(regs).acc[i * 5 + j] =
            fmaf(g0, r2l_pconv_sample(xb, yy + i - R, xx + j - R, a.H, a.W, a.pad_mirror),
                 R2L_TREG(regs).acc[i * 5 + j]);
    if (a.y) {
      int ey[3], ex[3];
      if (a.pad_mirror) {
        r2l_mirror_preimages<2>(yy, a.H, ey);
        r2l_mirror_preimages<2>(xx, a.W, ex);
      } else {
        ey[0] = yy;
        ex[0] = xx;
        ey[1] = ey[2] = ex[1] = ex[2] = -1000000;
      }
      float s = 0.f;
      for (int pp = 0; pp < 3; ++pp)
        for (int q = 0; q < 3; ++q) {
          if (ey[pp] < -2 || ex[q] < -2) continue;
          for (int i = 0; i < K; ++i) {
            const int py = ey[pp] - (i - R);
            if ((unsigned)py >= (unsigned)a.H) continue;
            for (int j = 0; j < K; ++j) {
              const int px = ex[q] - (j - R);
              if ((unsigned)px >= (unsigned)a.W) continue;
              s = fmaf(a.w[i * K + j], gb[(size_t)py * a.W + px], s);
            }
          }
        }
      float* yb = a.y + (size_t)b * 3 * hw;
      yb[p] = s;
      yb[hw + p] = gb[hw + p];
      yb[2 * hw + p] = gb[2 * hw + p];
    }
  }
  R2L_PHASE_END
  R2L_BLOCK_REDUCE(25, regs, lds, a.partial, bid, nblk)
}

// ---- pointwise stages over all B*3*H*W elements --------------------------------------------------------
// op 0 clip fwd: y = clamp(x, 1e-5, 1)                 op 1 clip bwd: y = g * [1e-5 <= x <= 1]
// op 2 gamma fwd: y = exp(log(x) / w[0])               op 3 gamma bwd: y = g * out / (gamma * x), acc: g*out*ln x
// op 4 add fwd: y = x + w[(c,h,w)] (broadcast over B)
// op 5 bn apply: y = (x - w[c]) * w[3+c]                op 6 bn bwd: y = w[3+c] * g - w[3+c] * aux2[c] - out * w[3+c] * aux2[3+c]
// op 7 bn stats: acc[c] += x - .5, acc[3+c] += (x - .5)^2
// op 8 normalize: y = (x - w[c]) / w[3+c]  (torchvision T.Normalize(mean, std), train.py:157-171)
struct R2LPointArgs {
  const float* x;
  const float* g;
  const float* w;
  const float* aux;   // gamma bwd: forward output; bn bwd: forward output (xhat)
  const float* aux2;  // bn bwd: mean_g[3], mean_gxhat[3] or null (eval mode)
  float* y;
  float* partial;
  int B, H, W, op;
};
R2L_BLOCKFN void r2l_point_block(const R2LPointArgs& a, int bid, int nblk, float* lds) {
  const size_t hw = (size_t)a.H * a.W, n4 = (size_t)a.B * 3 * hw / 4;  // hw is a multiple of 4
  R2L_TREG_DECL(R2LAccN<6>, regs);
  R2L_PHASE_BEGIN
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 6; ++i) R2L_TREG(regs).acc[i] = 0.f;
  for (size_t i4 = (size_t)bid * R2L_NT + tid; i4 < n4; i4 += (size_t)nblk * R2L_NT) {
    const size_t e = i4 * 4;
    const int c = (int)((e / hw) % 3);
    float xv[4] = {0, 0, 0, 0}, gv[4] = {0, 0, 0, 0}, av[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};
    if (a.x) {
      const r2l_f4 t = *(const r2l_f4*)(a.x + e);
      xv[0] = t.x, xv[1] = t.y, xv[2] = t.z, xv[3] = t.w;
    }
    if (a.g) {
      const r2l_f4 t = *(const r2l_f4*)(a.g + e);
      gv[0] = t.x, gv[1] = t.y, gv[2] = t.z, gv[3] = t.w;
    }
    if (a.aux) {
      const r2l_f4 t = *(const r2l_f4*)(a.aux + e);
      av[0] = t.x, av[1] = t.y, av[2] = t.z, av[3] = t.w;
    }
    for (int q = 0; q < 4; ++q) {
      switch (a.op) {
        case 0: o[q] = fminf(fmaxf(xv[q], 1e-5f), 1.0f); break;
        case 1: o[q] = (xv[q] >= 1e-5f && xv[q] <= 1.0f) ? gv[q] : 0.f; break;
        case 2: o[q] = r2l_exp2(r2l_log2(xv[q]) * (1.0f / a.w[0])); break;
        case 3: {
          o[q] = gv[q] * av[q] * (1.0f / a.w[0]) * r2l_rcp(xv[q]);
          R2L_TREG(regs).acc[0] = fmaf(gv[q] * av[q], r2l_log2(xv[q]), R2L_TREG(regs).acc[0]);
        } break;
        case 4: o[q] = xv[q] + a.w[(e + q) % (3 * hw)]; break;
        case 5: o[q] = (xv[q] - a.w[c]) * a.w[3 + c]; break;
        case 8: o[q] = (xv[q] - a.w[c]) / a.w[3 + c]; break;
        case 6:  // (the bias-free form of r2l_bn_bwd_pair: istd * g first, then the two small terms)
          o[q] = fmaf(av[q], -(a.aux2 ? a.w[3 + c] * a.aux2[3 + c] : 0.f), fmaf(a.w[3 + c], gv[q], -(a.aux2 ? a.w[3 + c] * a.aux2[c] : 0.f)));
          break;
        default: {
          const float d = xv[q] - 0.5f;
          R2L_PRAGMA_UNROLL
          for (int k = 0; k < 3; ++k) {
            R2L_TREG(regs).acc[k] += (c == k) ? d : 0.f;
            R2L_TREG(regs).acc[3 + k] += (c == k) ? d * d : 0.f;
          }
        } break;
      }
    }
    if (a.y) {
      r2l_f4 st;
      st.x = o[0], st.y = o[1], st.z = o[2], st.w = o[3];
      *(r2l_f4*)(a.y + e) = st;
    }
  }
  R2L_PHASE_END
  if (a.partial) {
    R2L_BLOCK_REDUCE(6, regs, lds, a.partial, bid, nblk)
  }
}

// ---- weak augmentation (utils/augmentation.py:8-14, :70-74): horizontal flip, vertical flip, rot90 --------
// y = rot90^k(vflip^v(hflip^h(x))) over the last two axes of (N, H, W) planes, k counted the way
// `x.rot90(k, dims=(-1, -2))` counts (the reference's RandomRotate90); inverse != 0 applies the inverse map
// (the VJP of a permutation is its inverse).  Out is (N, H, W) for even k and (N, W, H) for odd k.
struct R2LAugArgs {
  const float* x;
  float* y;
  int N, H, W;  // input plane size
  int hflip, vflip, k, inverse;
};
// forward map of one input coordinate: (r, c) in H x W -> (r2, c2) in Ho x Wo
R2L_HOSTDEV void r2l_aug_map(int H, int W, int hflip, int vflip, int k, int r, int c, int& r2, int& c2) {
  if (hflip) c = W - 1 - c;
  if (vflip) r = H - 1 - r;
  int h = H, w = W;
  for (int i = 0; i < (k & 3); ++i) {  // one rot90 with dims=(-1,-2): out[j][h-1-i'] ... (i, j) -> (j, h - 1 - i)
    const int nr = c, nc = h - 1 - r;
    r = nr;
    c = nc;
    const int t = h;
    h = w;
    w = t;
  }
  r2 = r;
  c2 = c;
}
R2L_BLOCKFN void r2l_aug_block(const R2LAugArgs& a, int bid, int nblk, float* lds) {
  (void)lds;
  const size_t hw = (size_t)a.H * a.W, n = (size_t)a.N * hw;
  const int Wo = (a.k & 1) ? a.H : a.W;
  R2L_PHASE_BEGIN
  for (size_t e = (size_t)bid * R2L_NT + tid; e < n; e += (size_t)nblk * R2L_NT) {
    const size_t pl = e / hw, rem = e - pl * hw;
    const int r = (int)(rem / a.W), c = (int)(rem - (size_t)r * a.W);
    int r2, c2;
    r2l_aug_map(a.H, a.W, a.hflip, a.vflip, a.k, r, c, r2, c2);
    const size_t o = pl * hw + (size_t)r2 * Wo + c2;
    if (a.inverse)
      a.y[e] = a.x[o];   // x is the augmented-shape gradient, y the input-shape gradient
    else
      a.y[o] = a.x[e];
  }
  R2L_PHASE_END
}
// The same permutation 16 bytes at a time (H % 4 == 0 and W % 4 == 0): the map is affine, element (r, c) of an input plane
// goes to element s0 + sr r + sc c of the output plane.  Even k: sc = +-1, a thread moves 4 consecutive columns as one
// vector (reversed for sc = -1).  Odd k: sr = +-1 -- a TRANSPOSE, through a 64 x 64 tile in LDS: rows of the source are
// read as vectors, columns of the tile leave as vectors along the destination's rows.  (The element-wise kernel above
// reads or writes 4-byte elements at a stride of a whole row there: 0.63 ms per pass on 64 x 3 x 512 x 512 against 0.1.)
struct R2LAugTiledArgs {
  const float* x;
  float* y;
  int N, H, W;     // input-shape plane size
  int s0, sr, sc;  // forward map (r2l_aug_map)
  int odd, inverse;
  int ntr, ntc;    // tiles per plane
};
#define R2L_AUG_TS 64
#define R2L_AUG_LDS_FLOATS (R2L_AUG_TS * (R2L_AUG_TS + 1))
R2L_HD r2l_f4 r2l_rev4(const r2l_f4& v) {
  r2l_f4 o;
  o.x = v.w;
  o.y = v.z;
  o.z = v.y;
  o.w = v.x;
  return o;
}
R2L_BLOCKFN void r2l_aug_tiled_block(const R2LAugTiledArgs& a, int bid, int nblk, float* lds) {
  constexpr int TS = R2L_AUG_TS, LS = TS + 1, RPP = R2L_NT / (TS / 4);  // tile rows per pass of the workgroup
  const size_t hw = (size_t)a.H * a.W;
  const int ntiles = a.N * a.ntr * a.ntc;
  for (int tile = bid; tile < ntiles; tile += nblk) {
    const int pl = tile / (a.ntr * a.ntc), tr = (tile / a.ntc) % a.ntr, tc = tile % a.ntc;
    const int r0 = tr * TS, c0 = tc * TS;
    const float* xb = a.x + (size_t)pl * hw;
    float* yb = a.y + (size_t)pl * hw;
    if (!a.odd) {
      // a row goes to a row: no transpose, no LDS
      R2L_PHASE_BEGIN
      for (int i = 0; i < TS / RPP; ++i) {
        const int r = r0 + tid / (TS / 4) + RPP * i, c = c0 + 4 * (tid % (TS / 4));
        if (r < a.H && c < a.W) {
          const long o = (long)a.s0 + (long)a.sr * r + (long)a.sc * c;  // augmented position of (r, c)
          const long ov = a.sc == 1 ? o : o - 3;                         // ... of the vector's first element
          if (!a.inverse) {
            const r2l_f4 v = *(const r2l_f4*)(xb + (size_t)r * a.W + c);
            *(r2l_f4*)(yb + ov) = a.sc == 1 ? v : r2l_rev4(v);
          } else {
            const r2l_f4 v = *(const r2l_f4*)(xb + ov);
            *(r2l_f4*)(yb + (size_t)r * a.W + c) = a.sc == 1 ? v : r2l_rev4(v);
          }
        }
      }
      R2L_PHASE_END
      continue;
    }
    // odd k: input-shape element (r, c) <-> augmented element s0 + sr r + sc c with sr = +-1: along an augmented row the
    // input ROW index runs.  lds[(r - r0) * LS + (c - c0)] holds input-shape element (r, c).
    R2L_PHASE_BEGIN
    for (int i = 0; i < TS / RPP; ++i) {
      if (!a.inverse) {  // input rows in as vectors
        const int rl = tid / (TS / 4) + RPP * i, cl = 4 * (tid % (TS / 4));
        const int r = r0 + rl, c = c0 + cl;
        if (r < a.H && c < a.W) {
          const r2l_f4 v = *(const r2l_f4*)(xb + (size_t)r * a.W + c);
          float* d = lds + rl * LS + cl;
          d[0] = v.x;
          d[1] = v.y;
          d[2] = v.z;
          d[3] = v.w;
        }
      } else {  // augmented rows in as vectors: fixed c, 4 consecutive r
        const int cl = tid / (TS / 4) + RPP * i, rl = 4 * (tid % (TS / 4));
        const int r = r0 + rl, c = c0 + cl;
        if (r < a.H && c < a.W) {
          const long o = (long)a.s0 + (long)a.sr * r + (long)a.sc * c;
          r2l_f4 v = *(const r2l_f4*)(xb + (a.sr == 1 ? o : o - 3));
          if (a.sr != 1) v = r2l_rev4(v);
          lds[(rl + 0) * LS + cl] = v.x;
          lds[(rl + 1) * LS + cl] = v.y;
          lds[(rl + 2) * LS + cl] = v.z;
          lds[(rl + 3) * LS + cl] = v.w;
        }
      }
    }
    R2L_PHASE_END
    R2L_PHASE_BEGIN
    for (int i = 0; i < TS / RPP; ++i) {
      if (!a.inverse) {  // tile columns out as vectors along the augmented rows
        const int cl = tid / (TS / 4) + RPP * i, rl = 4 * (tid % (TS / 4));
        const int r = r0 + rl, c = c0 + cl;
        if (r < a.H && c < a.W) {
          r2l_f4 v;
          v.x = lds[(rl + 0) * LS + cl];
          v.y = lds[(rl + 1) * LS + cl];
          v.z = lds[(rl + 2) * LS + cl];
          v.w = lds[(rl + 3) * LS + cl];
          const long o = (long)a.s0 + (long)a.sr * r + (long)a.sc * c;
          *(r2l_f4*)(yb + (a.sr == 1 ? o : o - 3)) = a.sr == 1 ? v : r2l_rev4(v);
        }
      } else {  // input rows out as vectors
        const int rl = tid / (TS / 4) + RPP * i, cl = 4 * (tid % (TS / 4));
        const int r = r0 + rl, c = c0 + cl;
        if (r < a.H && c < a.W) {
          const float* d = lds + rl * LS + cl;
          r2l_f4 v;
          v.x = d[0];
          v.y = d[1];
          v.z = d[2];
          v.w = d[3];
          *(r2l_f4*)(yb + (size_t)r * a.W + c) = v;
        }
      }
    }
    R2L_PHASE_END
  }
}
// ---- AddGaussianNoise (utils/augmentation.py:17-31): y = x + noise * std, noise drawn by the caller ------
struct R2LAxpyArgs {
  const float* x;
  const float* noise;
  float* y;
  float std;
  size_t n;
};
R2L_BLOCKFN void r2l_axpy_block(const R2LAxpyArgs& a, int bid, int nblk, float* lds) {
  (void)lds;
  R2L_PHASE_BEGIN
  for (size_t e = (size_t)bid * R2L_NT + tid; e < a.n; e += (size_t)nblk * R2L_NT) a.y[e] = a.x[e] + a.noise[e] * a.std;
  R2L_PHASE_END
}

// ---- AddGaussianNoise with the normal deviates generated in the kernel -------------
// y[i] = x[i] + std * n[i], n ~ N(0,1) from Philox4x32-10 (Salmon et al. 2011; the counter-based generator torch
// uses on GPUs) + Box-Muller.  Counter = (group index i/4 as 64 bits, offset as 64 bits), key = seed: the four
// 32-bit outputs of one counter give the four deviates of elements 4g .. 4g+3, so the result is a pure function
// of (seed, offset, i) -- independent of grid and launch shape, reproducible, and restated bit for bit (up to the
// float32 log / sin / cos) by the oracle.  No noise tensor is read or written: 4 B in + 4 B out per element instead
// of the 16 B/element of torch.randn_like + axpy.
struct R2LPhiloxArgs {
  const float* x;
  float* y;
  float std;
  unsigned long long seed, offset;
  size_t n;
};
R2L_HD void r2l_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                              unsigned o[4]) {
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o[0] = c0;
  o[1] = c1;
  o[2] = c2;
  o[3] = c3;
}
R2L_HD void r2l_box_muller(unsigned a, unsigned b, float& n0, float& n1) {
  const float u1 = ((float)a + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]: (a + 1) / 2^32 (float32 rounding may reach 1)
  const float u2 = (float)b * 2.3283064365386963e-10f;           // [0, 1]
  const float r = sqrtf(-2.0f * logf(u1));
  const float t = 6.2831853071795865f * u2;
  n0 = r * cosf(t);
  n1 = r * sinf(t);
}
R2L_BLOCKFN void r2l_philox_noise_block(const R2LPhiloxArgs& a, int bid, int nblk, float* lds) {
  (void)lds;
  const size_t ngroups = (a.n + 3) / 4;
  R2L_PHASE_BEGIN
  for (size_t g = (size_t)bid * R2L_NT + tid; g < ngroups; g += (size_t)nblk * R2L_NT) {
    unsigned o[4];
    r2l_philox4x32_10((unsigned)g, (unsigned)(g >> 32), (unsigned)a.offset, (unsigned)(a.offset >> 32),
                      (unsigned)a.seed, (unsigned)(a.seed >> 32), o);
    float nz[4];
    r2l_box_muller(o[0], o[1], nz[0], nz[1]);
    r2l_box_muller(o[2], o[3], nz[2], nz[3]);
    const size_t e = 4 * g;
    if (e + 3 < a.n && (((uintptr_t)(a.x + e) | (uintptr_t)(a.y + e)) & 15) == 0) {
      const r2l_f4 v = *(const r2l_f4*)(a.x + e);
      r2l_f4 w;
      w.x = fmaf(nz[0], a.std, v.x);
      w.y = fmaf(nz[1], a.std, v.y);
      w.z = fmaf(nz[2], a.std, v.z);
      w.w = fmaf(nz[3], a.std, v.w);
      *(r2l_f4*)(a.y + e) = w;
    } else {
      for (int k = 0; k < 4; ++k)
        if (e + k < a.n) a.y[e + k] = fmaf(nz[k], a.std, a.x[e + k]);
    }
  }
  R2L_PHASE_END
}
