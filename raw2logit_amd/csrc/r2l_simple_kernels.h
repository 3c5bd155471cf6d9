// r2l_simple_kernels.h -- small kernels around the fused ISP: parameter folding, fixed-order
// reductions, BatchNorm backward sums, additive-layer gradient, raw2rgb forward / backward.
#pragma once
#include "r2l_param_kernels.h"

// ---- fold / unfold (one lane per element, float64) ------------------------------------------------
struct R2LFoldArgs {
  const float* params;
  R2LFolded* F;
  unsigned* counters;  // arrival counters of the in-kernel reductions: a fresh workspace starts at zero
};
R2L_BLOCKFN void r2l_fold_block(const R2LFoldArgs& a, int bid, int nblk, float* lds) {
  (void)bid;
  (void)nblk;
  (void)lds;
  R2L_PHASE_BEGIN
  if (tid < R2L_FOLDED_NFLOATS) r2l_fold_one(a.params, a.F, tid);
  if (a.counters && tid < 1 + R2L_MAX_GROUPS) a.counters[tid] = 0;
  R2L_PHASE_END
}

// Step prologue (r2l_isp_step_fwd): the nine parameter tensors of the module -> the packed block (kept in the
// workspace: the backward reads the values the forward saw), the folded block, the arrival counters, and in eval
// mode BatchNorm's (mean, 1/sqrt(var + eps)) from the running statistics.  One launch instead of torch.cat +
// fold (+ two ATen kernels for the eval statistics).
struct R2LPackFoldArgs {
  const float* src[9];  // black_level, white_balance, colour_correction, gamma_correct, debayer.weight,
                        // sharpening_filter.weight, gaussian_blur.weight, M_RGB_2_YUV, M_YUV_2_RGB
  float* packed;        // [R2L_P_COUNT]
  R2LFolded* F;
  unsigned* counters;
  const float* running_mean;  // eval mode: -> bn; else null
  const float* running_var;
  float* bn;
  double eps;
};
R2L_HD void r2l_pack_slot(int i, int& t, int& o) {
  const int start[10] = {R2L_P_BLACK_LEVEL, R2L_P_WHITE_BALANCE, R2L_P_CCM, R2L_P_GAMMA, R2L_P_DEBAYER,
                         R2L_P_SHARPEN, R2L_P_BLUR, R2L_P_M_RGB2YUV, R2L_P_M_YUV2RGB, R2L_P_COUNT};
  t = 0;
  for (int k = 1; k < 9; ++k)
    if (i >= start[k]) t = k;
  o = i - start[t];
}
R2L_BLOCKFN void r2l_pack_fold_block(const R2LPackFoldArgs& a, int bid, int nblk, float* lds) {
  (void)bid;
  (void)nblk;
  float* pl = lds;  // [R2L_P_COUNT]
  R2L_PHASE_BEGIN
  if (tid < R2L_P_COUNT) {
    int t, o;
    r2l_pack_slot(tid, t, o);
    const float* src = a.src[0];  // (a select chain, not a dynamically indexed kernel-argument array)
    R2L_PRAGMA_UNROLL
    for (int k = 1; k < 9; ++k) src = (t == k) ? a.src[k] : src;
    const float v = src[o];
    pl[tid] = v;
    a.packed[tid] = v;
  }
  if (a.counters && tid < 1 + R2L_MAX_GROUPS) a.counters[tid] = 0;
  if (a.running_mean && tid >= 256 && tid < 259) {
    const int k = tid - 256;
    a.bn[k] = a.running_mean[k];
    a.bn[3 + k] = (float)(1.0 / sqrt((double)a.running_var[k] + a.eps));
  }
  R2L_PHASE_END
  R2L_PHASE_BEGIN
  if (tid < R2L_FOLDED_NFLOATS) r2l_fold_one(pl, a.F, tid);
  R2L_PHASE_END
}

struct R2LUnfoldArgs {
  const float* params;
  const double* sums;  // [R2L_NSUMS]
  float* grad_params;  // [R2L_P_NTRAIN]
  float scale;
};
R2L_BLOCKFN void r2l_unfold_block(const R2LUnfoldArgs& a, int bid, int nblk, float* lds) {
  (void)bid;
  (void)nblk;
  double* sums = (double*)(lds + 4);
  double* tg = sums + R2L_NSUMS;
  float* pl = (float*)(tg + R2L_UNFOLD_TG);
  R2L_PHASE_BEGIN
  if (tid < R2L_NSUMS) sums[tid] = a.sums[tid] * (double)a.scale;
  R2L_PHASE_END
  r2l_unfold_phases(a.params, sums, tg, pl, a.grad_params);
}

R2L_BLOCKFN void r2l_bn_finalize_block(const R2LBnFinalizeArgs& a, int bid, int nblk, float* lds) {
  (void)bid;
  (void)nblk;
  (void)lds;
  r2l_bn_finalize_phases(a);
}

// ---- BatchNorm backward means of the global batch: gathered[nranks][6] summed in rank order, / n -----
struct R2LBnBwdMeansArgs {
  const double* gathered;
  int nranks;
  const double* n;  // pixel count of the global batch
  float* bn_bwd;    // mean_c(g)[3], mean_c(g*xhat)[3]
};
R2L_BLOCKFN void r2l_bn_bwd_means_block(const R2LBnBwdMeansArgs& a, int bid, int nblk, float* lds) {
  (void)bid;
  (void)nblk;
  (void)lds;
  R2L_PHASE_BEGIN
  if (tid < 6) {
    double s = 0.0;
    for (int r = 0; r < a.nranks; ++r) s += a.gathered[r * 6 + tid];
    a.bn_bwd[tid] = (float)(s / *a.n);
  }
  R2L_PHASE_END
}

// ---- rows of per-workgroup partials -> float64 sums, fixed order ---------------------------------
// workgroup s reduces partial[s][0..n): lane t adds the elements t, t+256, ... in float64, then lane 0
// adds the 256 lane sums in order.
struct R2LReduceRowsArgs {
  const float* partial;  // [nslots][n]
  double* sums;          // [nslots]
  int n;
  double scale;  // applied to the result
  float* fsums;  // optional float32 copy of the result
  double* count_out;  // optional: receives `count` (workgroup 0)
  double count;
  const double* divide_by;  // optional: fsums = result / *divide_by instead of a plain copy
};
R2L_BLOCKFN void r2l_reduce_rows_block(const R2LReduceRowsArgs& a, int bid, int nblk, float* lds) {
  (void)nblk;
  double* dl = (double*)lds;
  R2L_PHASE_BEGIN
  double s = 0.0;
  for (int j = tid; j < a.n; j += R2L_NT) s += (double)a.partial[(size_t)bid * a.n + j];
  dl[tid] = s;
  R2L_PHASE_END
  R2L_PHASE_BEGIN
  if (tid < 32) {  // fixed-order tree: 32 lanes add R2L_NT/32 neighbours each, lane 0 adds the 32 results
    double t = 0.0;
    for (int j = 0; j < R2L_NT / 32; ++j) t += dl[tid * (R2L_NT / 32) + j];
    dl[R2L_NT + tid] = t;
  }
  R2L_PHASE_END
  R2L_PHASE_BEGIN
  if (tid == 0) {
    double t = 0.0;
    for (int j = 0; j < 32; ++j) t += dl[R2L_NT + j];
    if (a.sums) a.sums[bid] = t * a.scale;
    if (a.fsums) a.fsums[bid] = (float)(a.divide_by ? t * a.scale / *a.divide_by : t * a.scale);
    if (a.count_out && bid == 0) *a.count_out = a.count;
  }
  R2L_PHASE_END
}

// ---- BatchNorm backward sums: sum_c g, sum_c g*xhat (xhat == saved forward output) ---------------
struct R2LBnReduceArgs {
  const float* gout;
  const float* out;
  float* partial;  // [6][nblk]
  int B, H, W;
  R2LTree tree;          // in-kernel final reduction -> sums[6]
  double* sums;
  const double* totals;  // optional: totals[6] = pixel count n of the global batch
  float* bn_bwd;         // optional (needs totals): mean_c(g)[3], mean_c(g*xhat)[3] = sums / n as float32
  int order;             // 0: items in memory order; 1: from the end of the tensors to their start (diagnostic builds)
};
struct R2LAcc6 {
  float acc[6];
};
#define R2L_SEG (16 * R2L_NT)  // floats per (plane, segment) work item: 4 x float4 per lane
R2L_HD void r2l_bn_reduce_item(int tid, const R2LBnReduceArgs& a, int item, int nsegpp, R2LAcc6& regs) {
  const int plane = item / nsegpp, seg = item - plane * nsegpp;
  const int k = plane % 3;
  const size_t hw = (size_t)a.H * a.W;  // multiple of 4 because H and W are even
  const float* g = a.gout + (size_t)plane * hw;
  const float* o = a.out + (size_t)plane * hw;
  float sg = 0.f, sgx = 0.f;
  R2L_PRAGMA_UNROLL
  for (int q = 0; q < 4; ++q) {
    const size_t e = (size_t)seg * R2L_SEG + (size_t)(q * R2L_NT + tid) * 4;
    if (e < hw) {
      // nontemporal: 400 MB read once; plain loads also evict the raw frames and dL/dY'' that the two backward
      // kernels are about to re-read (82 -> 65 us here, -8 us in bwd1, -5 us in bwd2)
#ifdef R2L_BNR_GOUT_PLAIN  // A/B builds: grad_out allocates in the caches on its way through (kernel B1 reads it again right away)
      const r2l_f4 gv = *(const r2l_f4*)(g + e);
#else
      const r2l_f4 gv = r2l_load_f4_nt(g + e);
#endif
      const r2l_f4 ov = r2l_load_f4_nt(o + e);
      sg += (gv.x + gv.y) + (gv.z + gv.w);
      sgx = fmaf(gv.x, ov.x, sgx);
      sgx = fmaf(gv.y, ov.y, sgx);
      sgx = fmaf(gv.z, ov.z, sgx);
      sgx = fmaf(gv.w, ov.w, sgx);
    }
  }
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 3; ++c) {
    regs.acc[c] += (k == c) ? sg : 0.f;
    regs.acc[3 + c] += (k == c) ? sgx : 0.f;
  }
}
R2L_BLOCKFN void r2l_bn_reduce_block(const R2LBnReduceArgs& a, int bid, int nblk, float* lds) {
  R2L_TREG_DECL(R2LAcc6, regs);
  R2L_PHASE_BEGIN
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 6; ++i) R2L_TREG(regs).acc[i] = 0.f;
  R2L_PHASE_END
  const size_t hw = (size_t)a.H * a.W;
  const int nsegpp = (int)((hw + R2L_SEG - 1) / R2L_SEG);
  const int nitems = 3 * a.B * nsegpp;
  R2L_PHASE_BEGIN  // one phase: the wavefronts stream independently (a barrier per item would cap the loads in flight)
  for (int item = bid; item < nitems; item += nblk)
    r2l_bn_reduce_item(tid, a, a.order ? nitems - 1 - item : item, nsegpp, R2L_TREG(regs));
  R2L_PHASE_END
  R2L_TAILST(10);
  R2L_BLOCK_REDUCE(6, regs, lds, a.partial, bid, nblk)
  R2L_TAILST(11);
  if (a.tree.counters) {
    double* sl = (double*)(lds + 4);
    if (!r2l_tree_finish<6>(a.tree, bid, nblk, lds, sl, (double*)(lds + 512), (R2L_RED_FLOATS_N(6) - 512) / 2))
      return;
    R2L_PHASE_BEGIN
    if (tid < 6) {
      a.sums[tid] = sl[tid];
      if (a.bn_bwd && a.totals) a.bn_bwd[tid] = (float)(sl[tid] / a.totals[6]);
    }
    R2L_PHASE_END
    R2L_TAILST(17);
  }
}

// ---- gradient of the additive layer (pipeline_torch.py:213): sum over the batch ------------------
struct R2LAddBwdArgs {
  const float* gout;
  const float* out;
  const float* bn;      // mean, istd or null
  const float* bn_bwd;  // mean_g, mean_gxhat or null
  float* gadd;          // (3,H,W)
  int B, H, W;
};
R2L_BLOCKFN void r2l_add_bwd_block(const R2LAddBwdArgs& a, int bid, int nblk, float* lds) {
  (void)lds;
  const size_t hw = (size_t)a.H * a.W;
  const size_t nchunk = 3 * hw / 4;
  R2L_PHASE_BEGIN
  for (size_t ch = (size_t)bid * R2L_NT + tid; ch < nchunk; ch += (size_t)nblk * R2L_NT) {
    const size_t e = ch * 4;
    const int k = (int)(e / hw);
    float istd = 1.f, mg = 0.f, mgx = 0.f;
    if (a.bn) istd = a.bn[3 + k];
    if (a.bn_bwd) {  // (istd * mean(g), istd * mean(g * xhat): the bias-free form of r2l_bn_bwd_pair)
      mg = istd * a.bn_bwd[k];
      mgx = istd * a.bn_bwd[3 + k];
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < a.B; ++b) {
      const r2l_f4 g = *(const r2l_f4*)(a.gout + (size_t)b * 3 * hw + e);
      if (a.bn) {
        const r2l_f4 o = *(const r2l_f4*)(a.out + (size_t)b * 3 * hw + e);
        s[0] += fmaf(o.x, -mgx, fmaf(istd, g.x, -mg));
        s[1] += fmaf(o.y, -mgx, fmaf(istd, g.y, -mg));
        s[2] += fmaf(o.z, -mgx, fmaf(istd, g.z, -mg));
        s[3] += fmaf(o.w, -mgx, fmaf(istd, g.w, -mg));
      } else {
        s[0] += g.x;
        s[1] += g.y;
        s[2] += g.z;
        s[3] += g.w;
      }
    }
    r2l_f4 st;
    st.x = s[0];
    st.y = s[1];
    st.z = s[2];
    st.w = s[3];
    *(r2l_f4*)(a.gadd + e) = st;
  }
  R2L_PHASE_END
}

// ---- raw2rgb (pipeline_torch.py:240-283) -----------------------------------------------------------
// one lane per pair of horizontally adjacent Bayer quads (4 columns x 2 rows of raw)
struct R2LRaw2RgbArgs {
  R2LRaw raw;        // fwd input / (bwd: unused)
  const float* bl;   // 4 floats or null
  float* out;        // fwd output
  const float* gout;  // bwd input
  float* graw;        // bwd output (B,H,W) or null
  float* partial;     // bwd: [4][nblk] black-level partials or null
  int B, H, W, reduce_size, C;
};
R2L_HD void r2l_quadpair_coords(size_t idx, int H, int W, int& b, int& qy, int& qx2) {
  const int npx = (W + 3) / 4, nqy = H / 2;
  qx2 = (int)(idx % npx);
  const size_t r = idx / npx;
  qy = (int)(r % nqy);
  b = (int)(r / nqy);
}
R2L_BLOCKFN void r2l_raw2rgb_fwd_block(const R2LRaw2RgbArgs& a, int bid, int nblk, float* lds) {
  (void)lds;
  const int npx = (a.W + 3) / 4;
  const size_t nitems = (size_t)a.B * (a.H / 2) * npx;
  const int C = a.C;
  R2L_PHASE_BEGIN
  float bl[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.bl)
    for (int s = 0; s < 4; ++s) bl[s] = a.bl[s];
  for (size_t idx = (size_t)bid * R2L_NT + tid; idx < nitems; idx += (size_t)nblk * R2L_NT) {
    int b, qy, qx2;
    r2l_quadpair_coords(idx, a.H, a.W, b, qy, qx2);
    const int y = 2 * qy, x = 4 * qx2;
    const int nq = (x + 2 < a.W) ? 2 : 1;  // quads in this pair (W even)
    const size_t r0 = ((size_t)b * a.H + y) * a.W + x, r1 = r0 + a.W;
    for (int q = 0; q < nq; ++q) {
      const float R = r2l_raw_elem(a.raw, r0 + 2 * q) - bl[0], G1 = r2l_raw_elem(a.raw, r0 + 2 * q + 1) - bl[1];
      const float G2 = r2l_raw_elem(a.raw, r1 + 2 * q) - bl[2], Bv = r2l_raw_elem(a.raw, r1 + 2 * q + 1) - bl[3];
      if (a.reduce_size) {
        const int h2 = a.H / 2, w2 = a.W / 2;
        float* o = a.out + (size_t)b * C * h2 * w2 + (size_t)qy * w2 + (x / 2 + q);
        const size_t pl = (size_t)h2 * w2;
        if (C == 3) {
          o[0] = R;
          o[pl] = (G1 + G2) / 2;
          o[2 * pl] = Bv;
        } else {
          o[0] = R;
          o[pl] = G1;
          o[2 * pl] = G2;
          o[3 * pl] = Bv;
        }
      } else {
        const size_t pl = (size_t)a.H * a.W;
        float* o = a.out + (size_t)b * C * pl + (size_t)y * a.W + x + 2 * q;
        // zero-filled mosaic: every channel gets the 2x2 block, one site non-zero (two for G, C == 3)
        for (int c = 0; c < C; ++c) {
          float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
          if (c == 0) v00 = R;
          if (C == 3) {
            if (c == 1) {
              v01 = G1;
              v10 = G2;
            }
            if (c == 2) v11 = Bv;
          } else {
            if (c == 1) v01 = G1;
            if (c == 2) v10 = G2;
            if (c == 3) v11 = Bv;
          }
          float* oc = o + (size_t)c * pl;
          oc[0] = v00;
          oc[1] = v01;
          oc[a.W] = v10;
          oc[a.W + 1] = v11;
        }
      }
    }
  }
  R2L_PHASE_END
}

struct R2LAcc4 {
  float acc[4];
};
R2L_BLOCKFN void r2l_raw2rgb_bwd_block(const R2LRaw2RgbArgs& a, int bid, int nblk, float* lds) {
  const int npx = (a.W + 3) / 4;
  const size_t nitems = (size_t)a.B * (a.H / 2) * npx;
  const int C = a.C;
  R2L_TREG_DECL(R2LAcc4, regs);
  R2L_PHASE_BEGIN
  for (int s = 0; s < 4; ++s) R2L_TREG(regs).acc[s] = 0.f;
  for (size_t idx = (size_t)bid * R2L_NT + tid; idx < nitems; idx += (size_t)nblk * R2L_NT) {
    int b, qy, qx2;
    r2l_quadpair_coords(idx, a.H, a.W, b, qy, qx2);
    const int y = 2 * qy, x = 4 * qx2;
    const int nq = (x + 2 < a.W) ? 2 : 1;
    for (int q = 0; q < nq; ++q) {
      float gR, gG1, gG2, gB;
      if (a.reduce_size) {
        const int h2 = a.H / 2, w2 = a.W / 2;
        const size_t pl = (size_t)h2 * w2;
        const float* g = a.gout + (size_t)b * C * pl + (size_t)qy * w2 + (x / 2 + q);
        if (C == 3) {
          gR = g[0];
          gG1 = gG2 = g[pl] / 2;
          gB = g[2 * pl];
        } else {
          gR = g[0];
          gG1 = g[pl];
          gG2 = g[2 * pl];
          gB = g[3 * pl];
        }
      } else {
        const size_t pl = (size_t)a.H * a.W;
        const float* g = a.gout + (size_t)b * C * pl + (size_t)y * a.W + x + 2 * q;
        gR = g[0];
        if (C == 3) {
          gG1 = g[pl + 1];
          gG2 = g[pl + a.W];
          gB = g[2 * pl + a.W + 1];
        } else {
          gG1 = g[pl + 1];
          gG2 = g[2 * pl + a.W];
          gB = g[3 * pl + a.W + 1];
        }
      }
      if (a.graw) {
        float* r0 = a.graw + ((size_t)b * a.H + y) * a.W + x + 2 * q;
        r0[0] = gR;
        r0[1] = gG1;
        r0[a.W] = gG2;
        r0[a.W + 1] = gB;
      }
      R2L_TREG(regs).acc[0] -= gR;
      R2L_TREG(regs).acc[1] -= gG1;
      R2L_TREG(regs).acc[2] -= gG2;
      R2L_TREG(regs).acc[3] -= gB;
    }
  }
  R2L_PHASE_END
  if (a.partial) {
    R2L_BLOCK_REDUCE(4, regs, lds, a.partial, bid, nblk)
  }
}
