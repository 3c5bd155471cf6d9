// r2l_param_plane_bwd.h -- kernel B1 of the backward (r2l_bwd1_block, r2l_param_kernels.h: BatchNorm / gamma / clip /
// YUV->RGB adjoints, d/d blur weights, d/d chroma stencils, d/d gamma, dL/dY'' out) as TWO passes over planes, organised
// like the apply pass of the forward (r2l_fwd_apply_block, r2l_param_stream.h) instead of as LDS tiles: INDEPENDENT
// wavefronts, a wavefront owns (image, band of rows, 256-column strip) and walks down the band with its windows and
// accumulators in registers.  Everything B1 needs is a gather from planes (raw, the kept Y', grad_out), so there is no
// exchange between wavefronts, no halo to recompute, no phases, no barriers.
//   r2l_bwd1_plane_block  raw + Y' + grad_out -> dL/dY'' (stored), chroma-stencil / black-level / gamma sums
//   r2l_bwd1_blur_block   dL/dY'' + Y'        -> the 25 blur-weight sums
// Two kernels because one does not fit: windows (66 registers) + prefetch (36) + all 46 accumulator pairs (92) +
// the pointwise part's temporaries need more than the 256 registers of two wavefronts per SIMD (125 spilled registers);
// without the 25 blur pairs the first pass takes 211.  The second pass re-reads dL/dY'' (4 B/px) and Y' (4 B/px).
// When kernel B2 runs as plane passes too, the second pass is r2l_bwd1_blur_hp_block instead (below): the blur-weight sums
// AND kernel B2's first pass (the blur's adjoint) in one walk over dL/dY''.
//
// What the tile kernel's threads did not have to handle: a lane sees BOTH row parities.  The parity-indexed sums (folded
// chroma stencils GAU / GAV [row parity][tap] x column-parity pair, SU / SV) therefore exist twice, 20 pairs per row
// parity; the bank of the row at hand is in registers, the other one in a wavefront-private LDS area, swapped in place
// (4 floats at a time) once per row: 10 ds_read_b128 + 10 ds_write_b128 against ~300 vector instructions of a row step.
// Border rows: window rows outside the image are zero; the forward takes the blur's border-row weight sets (as the
// forward kernels do), and the first / last two image rows add the mirror padding's share of the blur-weight sums as a
// few extra products of their own window rows (row -1 IS row 1, ...) behind uniform branches; mirror columns are part
// of the 8-wide rows as in the forward; raw rows are fetched mirrored.
#pragma once
#include "r2l_param_stream.h"

#ifndef R2L_SERIAL

#ifndef R2L_BP_PF
#define R2L_BP_PF 2   // rows of raw / Y' in flight
#endif
#ifndef R2L_BP_PFG
#define R2L_BP_PFG 1  // rows of grad_out in flight (12 registers each)
#endif
#define R2L_BP_NWV 4                                  // wavefronts (work items in flight) per workgroup
#define R2L_BP_BANK 40                                // floats of one parity bank: GAU[9], GAV[9], SU, SV as pairs
#define R2L_BP_NT (64 * R2L_BP_NWV)
#define R2L_BP_RED_FLOATS_R(ROWS, NT) ((ROWS) * ((NT) + 1) + (ROWS) * 16)  // r2l_bp_block_reduce, ROWS slots at a time
#define R2L_BP_RED_FLOATS R2L_BP_RED_FLOATS_R(32, R2L_BP_NT)
#define R2L_BP_SWAP_FLOATS (R2L_BP_NWV * 64 * R2L_BP_BANK)
// B1's plane pass: its 81 slots in two batches of 41 (two workgroups per CU: 45 KB each)
#define R2L_BP_ROWS 41
#define R2L_BP_LDS_FLOATS                                                                    \
  (R2L_BP_SWAP_FLOATS > R2L_BP_RED_FLOATS_R(R2L_BP_ROWS, R2L_BP_NT) ? R2L_BP_SWAP_FLOATS \
                                                                     : R2L_BP_RED_FLOATS_R(R2L_BP_ROWS, R2L_BP_NT))

struct R2LBpStage {  // grad_out of one row in flight: 3 channels x the lane's 4 pixels
  r2l_f4 g[3];
};
// The LAST reader of a kept plane may take it around the caches (nontemporal loads), so that what it leaves in the 256 MiB memory-side
// cache is what the next passes want (profiles/r05_nt_stores.txt, 64x512x512, alternating processes):
//   HP in the sums pass (the step's last kernel): the NEXT step's apply pass 63.5 -> 60.1 us, blur pass 43.3 -> 41.5, statistics -0.9
//     -- 67 MB less dead weight in the cache when the forward starts.  ADOPTED (R2L_B2S_HP_NT 1).
//   dL/dY'' in the blur pass, Y' in the blur pass, the raw frames in the sums pass: nothing, nothing, +1 us.  Switches kept for A/B runs.
#ifndef R2L_B2S_HP_NT
#define R2L_B2S_HP_NT 1
#endif
#ifdef R2L_EXP_HB_GY_NT
#define R2L_HB_GY_LOAD(p) r2l_load_f4_nt(p)
#else
#define R2L_HB_GY_LOAD(p) r2l_stream_load_f4(p)
#endif
#ifndef R2L_HB_YP_NT
#define R2L_HB_YP_NT false   // Y' in the blur pass (its last reader)
#endif
#ifndef R2L_B2S_RAW_NT
#define R2L_B2S_RAW_NT false  // the raw frames in the sums pass (their last reader in the step)
#endif
#if R2L_B2S_HP_NT
#define R2L_B2S_HP_LOAD(p) r2l_load_f4_nt(p)
#else
#define R2L_B2S_HP_LOAD(p) r2l_stream_load_f4(p)
#endif
template <bool EPI>
R2L_HD void r2l_bp_fetch_g(const float* gimg, unsigned plane, int y, int H, int W, int x0, const R2LEpi& ep,
                           R2LBpStage& s) {
  const int yc = y < 0 ? 0 : (y >= H ? H - 1 : y);
  if (EPI) {  // grad_out arrives in the augmented layout the forward's epilogue wrote: element s0 + sr y + sc x (R2LEpi)
    const float* p = gimg + (ep.s0 + ep.sr * yc + ep.sc * x0);
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) s.g[k] = r2l_epi_load4(p + (size_t)k * plane, ep.sc);
    return;
  }
  const float* p = gimg + (size_t)yc * W + x0;
  R2L_PRAGMA_UNROLL
#ifdef R2L_BP_GOUT_PLAIN  // A/B builds
  for (int k = 0; k < 3; ++k) s.g[k] = *(const r2l_f4*)(p + (size_t)k * plane);
#else
  for (int k = 0; k < 3; ++k) s.g[k] = r2l_load_f4_nt(p + (size_t)k * plane);  // read once: nontemporal (as kernel B1)
#endif
}

// ================================================================================================
// BatchNorm's backward sums WITHOUT reading the forward's output back (round 5).  sum_c g and sum_c g * xhat need xhat, the
// normalised output -- 12 B/px that the apply pass wrote around the caches and that, in training, left them long before the
// backward starts.  The raw frame (4 B/px) and the kept plane Y' (4 B/px) determine it: this pass walks them exactly like the
// apply pass (r2l_fa_step: chroma stencils, blur of Y', colour code, the same functions on the same numbers -- xhat is the
// apply pass's stored value bit for bit), reads grad_out once (nontemporal) and keeps six sums per lane.  HBM traffic 20 B/px
// instead of 24, 12 of them -- grad_out -- from memory, the rest out of the memory-side cache when the forward just ran:
// 66 -> 4x us at 64x512x512 (profiles/r05_bnr_recompute.txt).  Kernel B1 recomputes the forward from the same two planes.
// Reduction: per band the lanes' float32 pair sums -> float64, butterfly over the wavefront in a fixed order, added to the
// wavefront's float64 totals in LDS; per workgroup (high, low) float32 halves of the six totals -> the shared tree
// (12 slots, as the statistics pass: r2l_fs_stats_finish).
#ifndef R2L_BNR_PF
#define R2L_BNR_PF 2  // rows in flight per stream (raw, Y', grad_out): 3 takes the kernel past 168 registers
#endif
struct R2LBnrArgs {
  R2LFwdStreamArgs s;    // raw, F, bn (mean, 1/std), yp_in, B, H, W, nband, band_h, nitems, stat_partial, tree, ep, xcdm
  const float* gout;
  double* sums;          // [6]
  const double* totals;  // optional: totals[6] = pixel count n of the global batch
  float* bn_bwd;         // optional (needs totals): sums / n as float32
};
R2L_HD void r2l_bnr_lane_sums(const r2l_p2* acc, bool ok, int lane, double* tots) {
  double part[6];
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 6; ++k) {
    double v = ok ? (double)acc[k][0] + (double)acc[k][1] : 0.0;  // (lanes beyond the frame's last column computed nothing real)
    R2L_PRAGMA_UNROLL
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    part[k] = v;
  }
  if (lane == 0) {
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 6; ++i) tots[i] += part[i];
  }
}
template <bool U16, bool EPI>
R2L_HD void r2l_bnr_item(const R2LBnrArgs& ba, int item, int lane, const float mean[3], const float istd[3], double* tots) {
  const R2LFwdStreamArgs& a = ba.s;
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  const int nstrip = (a.W + 255) >> 8;
  const unsigned plane = (unsigned)a.H * (unsigned)a.W;
  const int strip = item % nstrip, ib = item / nstrip;
  const int band = ib % a.nband, b = ib / a.nband;
  const int xs = strip * 256 + 4 * lane;
  const bool in_w = xs < a.W;
  const int x0 = in_w ? xs : a.W - 4;
  const bool le = x0 == 0, re = x0 + 4 >= a.W;
  const int y0 = band * a.band_h;  // a multiple of 6 (the host rounds band_h)
  const int y1 = (y0 + a.band_h < a.H) ? y0 + a.band_h : a.H;
  const size_t img = (size_t)b * plane;
  const float* ypimg = a.yp_in + img;
  const float* gimg = ba.gout + (size_t)b * 3 * plane;
  R2LFaState st;
  r2l_p2 acc[6];
  float piv[3] = {0.5f, 0.5f, 0.5f};  // (unused by this mode)
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 6; ++i) acc[i] = r2l_splat2(0.f);
  constexpr int PF = R2L_BNR_PF;
  R2LFsStage pf[PF];   // raw row q + 1
  R2LFaStage pfy[PF];  // Y' row q + 2
  R2LBpStage pfg[PF];  // grad_out row q
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < PF; ++i) {
    r2l_fa_fetch_raw<U16>(a, img, r2l_mirror(R2L_NH(y0 - 3 + i), a.H), x0, le, re, lane, pf[(2 + i) % PF]);
    r2l_fa_fetch(ypimg, R2L_NH(y0 - 2 + i), a.H, a.W, x0, le, re, lane, pfy[(2 + i) % PF]);
  }
#define R2L_BNR_LOAD_STEP(K, q)                                                                          \
  {                                                                                                      \
    r2l_fs_convert<U16>(a, F, pf[(K) % PF], le, re, st.v[((K) + 1) % 3]);                                \
    r2l_fa_build(pfy[(K) % PF], (unsigned)((q) + 2) < (unsigned)a.H, le, re, st.yp[((K) + 2) % 6]);      \
    r2l_fa_fetch_raw<U16>(a, img, r2l_mirror(R2L_NH((q) + 1 + PF), a.H), x0, le, re, lane, pf[(K) % PF]); \
    r2l_fa_fetch(ypimg, R2L_NH((q) + 2 + PF), a.H, a.W, x0, le, re, lane, pfy[(K) % PF]);                \
  }
  R2L_BNR_LOAD_STEP(2, y0 - 4)
  R2L_BNR_LOAD_STEP(3, y0 - 3)
  R2L_BNR_LOAD_STEP(4, y0 - 2)
  R2L_BNR_LOAD_STEP(5, y0 - 1)
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < PF; ++i) r2l_bp_fetch_g<EPI>(gimg, plane, y0 + i, a.H, a.W, x0, a.ep, pfg[i % PF]);
  for (int qb = y0; qb < y1; qb += 6) {
    R2L_PROGRESS_PRIO(qb - y0, y1 - y0);
#define R2L_BNR_STEP(K)                                                                                   \
  {                                                                                                       \
    const int q = qb + K;                                                                                 \
    R2L_BNR_LOAD_STEP(K, q)                                                                               \
    r2l_p2 gk[3][2];                                                                                      \
    R2L_PRAGMA_UNROLL                                                                                     \
    for (int k = 0; k < 3; ++k) {                                                                         \
      gk[k][0] = r2l_mk2(pfg[K % PF].g[k].x, pfg[K % PF].g[k].y);                                         \
      gk[k][1] = r2l_mk2(pfg[K % PF].g[k].z, pfg[K % PF].g[k].w);                                         \
    }                                                                                                     \
    r2l_bp_fetch_g<EPI>(gimg, plane, q + PF, a.H, a.W, x0, a.ep, pfg[K % PF]); /* (rows past H: clamped, never counted) */ \
    if (r2l_opaque_true())                                                                                \
      r2l_fa_step<K, false, false, true>(a, st, acc, piv, q, y0, false, nullptr, plane, x0, mean, istd, gk, \
                                          q < y1 ? 1.f : 0.f);                                            \
  }
    R2L_BNR_STEP(0)
    R2L_BNR_STEP(1)
    R2L_BNR_STEP(2)
    R2L_BNR_STEP(3)
    R2L_BNR_STEP(4)
    R2L_BNR_STEP(5)
#undef R2L_BNR_STEP
  }
#undef R2L_BNR_LOAD_STEP
  r2l_bnr_lane_sums(acc, in_w, lane, tots);
}
template <bool U16, bool EPI, int NWV>
R2L_BLOCKFN void r2l_bnr_planes_block(const R2LBnrArgs& ba, int bid, int nblk, float* lds) {
  const R2LFwdStreamArgs& a = ba.s;
  constexpr int NT = NWV * 64;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  float mean[3], istd[3];
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    mean[k] = a.bn[k];
    istd[k] = a.bn[3 + k];
  }
  float* red = lds + 16;
  double* tots = (double*)(lds + 16 + R2L_FS_RED_FLOATS(NWV)) + wave * 6;
  if (lane < 6) tots[lane] = 0.0;
  R2L_PRAGMA_NOUNROLL
  for (int item = r2l_xcd_window(bid, nblk, a.xcdm) * NWV + wave; item < a.nitems; item += nblk * NWV)
    r2l_bnr_item<U16, EPI>(ba, item, lane, mean, istd, tots);
  // ---- the wavefronts' float64 totals -> (high, low) float32 halves per workgroup -> the shared tree (as r2l_fs_stats_finish)
  R2L_LDS_BARRIER();
  if (tid < 6) {
    double acc = 0.0;
    for (int w = 0; w < NWV; ++w) acc += (tots - wave * 6)[w * 6 + tid];
    const float hi = (float)acc;
    r2l_store_coherent(&a.stat_partial[(size_t)tid * nblk + bid], hi);
    r2l_store_coherent(&a.stat_partial[(size_t)(6 + tid) * nblk + bid], (float)(acc - (double)hi));
  }
  R2L_STORES_DONE();
  R2L_LDS_BARRIER();
  double* sl = (double*)(red + 4);
  if (a.tree.counters &&
      r2l_tree_finish<12, NT>(a.tree, bid, nblk, red, sl, (double*)(red + 512), (R2L_FS_RED_FLOATS(NWV) - 512) / 2)) {
    if (tid < 6) sl[tid] += sl[6 + tid];
    R2L_LDS_BARRIER();
    if (tid < 6) {
      ba.sums[tid] = sl[tid];
      if (ba.bn_bwd && ba.totals) ba.bn_bwd[tid] = (float)(sl[tid] / ba.totals[6]);
    }
  }
}

struct R2LBpState {
  float v[3][6];   // V rows (slot = row mod 3)
  float yp[6][8];  // Y' rows (slot = row mod 6)
};
struct R2LBpAcc {
  r2l_p2 gau[9];    // bank of the CURRENT row parity: sum gU(p) * v_ext(p + t), halves = column parity
  r2l_p2 gav[9];
  r2l_p2 su, sv;
  r2l_p2 ggam;
};

// the bank in registers <-> the bank in the wavefront's LDS area, in place, 4 floats at a time
R2L_HD void r2l_bp_swap(R2LBpAcc& A, float* bank /* this lane's 40 floats, [chunk][lane][4] */) {
  float* f[10];
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 10; ++c) f[c] = bank + c * 64 * 4;
#define R2L_BP_SW(c, a, b)                      \
  {                                             \
    const r2l_f4 old = r2l_lds_f4(f[c]);        \
    r2l_f4 nw;                                  \
    nw.x = (a)[0];                              \
    nw.y = (a)[1];                              \
    nw.z = (b)[0];                              \
    nw.w = (b)[1];                              \
    *(r2l_f4*)f[c] = nw;                        \
    (a) = r2l_mk2(old.x, old.y);                \
    (b) = r2l_mk2(old.z, old.w);                \
  }
  R2L_BP_SW(0, A.gau[0], A.gau[1])
  R2L_BP_SW(1, A.gau[2], A.gau[3])
  R2L_BP_SW(2, A.gau[4], A.gau[5])
  R2L_BP_SW(3, A.gau[6], A.gau[7])
  R2L_BP_SW(4, A.gau[8], A.gav[0])
  R2L_BP_SW(5, A.gav[1], A.gav[2])
  R2L_BP_SW(6, A.gav[3], A.gav[4])
  R2L_BP_SW(7, A.gav[5], A.gav[6])
  R2L_BP_SW(8, A.gav[7], A.gav[8])
  R2L_BP_SW(9, A.su, A.sv)
#undef R2L_BP_SW
}

// one output row y (K = y mod 6, row parity K & 1); the windows hold V(y-1 .. y+1) and Y'(y-2 .. y+2)
template <int K>
R2L_HD void r2l_bp_step(const R2LBwd1Args& a, R2LBpState& st, R2LBpAcc& A, const R2LBpStage& gs, int y, bool store_ok,
                        float* gyb, int x0, const R2LBnConsts& bc) {
  constexpr int PY = K & 1;
  const int H = a.H;
  // ---- the blur's window: rows outside the image are zero; the forward takes the weight sets with the mirror padding
  // folded in for the first / last two image rows (as the forward kernels do) -----------------------------------------
  float yw[5][8];
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 5; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 8; ++j) yw[i][j] = st.yp[(K + 4 + i) % 6][j];
  r2l_p2 ypp[2], u[2], v[2];
  {
    R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque_after(a.F, yw[2][2]));
    const int set = (y < 2) ? y : (y - (H - 2)) + 2;
#ifdef R2L_EXP_CONST_WEIGHTS
    const float* w25 = &F.blur[0];  // (timing only: no border sets)
    (void)set;
#else
    const R2L_CONSTAS float* w25 =
        (y >= 2 && y < H - 2) ? &F.blur[0] : &F.blur_edge[0][0] + 25 * set;
#endif
    r2l_blur_row2w(yw, w25, ypp);
  }
  const float* vu = st.v[(K + 2) % 3];  // V(y-1)
  const float* vm = st.v[K % 3];        // V(y)
  const float* vl = st.v[(K + 1) % 3];  // V(y+1)
  {
    R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque_after(a.F, ypp[1][1]));
    r2l_fs_stencil_parity(vu, vm, vl, F.AU2[PY], u);
    r2l_fs_stencil_parity(vu, vm, vl, F.AV2[PY], v);
  }
  // ---- pointwise part: forward values, BatchNorm / gamma / clip adjoints (r2l_bwd1_row) --------------------------------
  r2l_p2 gy2[2], gu[2], gv[2];
  {
    R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque_after(a.F, v[1][1]));
    r2l_p2 grgb[3][2];
    r2l_p2 ggam = r2l_splat2(0.f);
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 3; ++k) {
      const float g[4] = {gs.g[k].x, gs.g[k].y, gs.g[k].z, gs.g[k].w};
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) {
        r2l_p2 rgb = r2l_pmul(r2l_splat2(F.M2[k * 3]), ypp[p]);
        rgb = r2l_pfma(r2l_splat2(F.M2[k * 3 + 1]), u[p], rgb);
        rgb = r2l_pfma(r2l_splat2(F.M2[k * 3 + 2]), v[p], rgb);
        const r2l_p2 xc = r2l_mk2(fminf(fmaxf(rgb[0], 1e-5f), 1.0f), fminf(fmaxf(rgb[1], 1e-5f), 1.0f));
        const r2l_p2 lg = r2l_mk2(r2l_log2(xc[0]), r2l_log2(xc[1]));
        const r2l_p2 e = r2l_pmul(lg, r2l_splat2(F.inv_gamma));
        const r2l_p2 og = r2l_mk2(r2l_exp2(e[0]), r2l_exp2(e[1]));
        const r2l_p2 xhat = r2l_pmul(r2l_padd(og, r2l_splat2(-bc.mean[k])), r2l_splat2(bc.istd[k]));
        r2l_p2 gx = r2l_bn_bwd_pair(r2l_mk2(g[2 * p], g[2 * p + 1]), xhat, bc.istd[k], bc.mg[k], bc.mgx[k]);
        if (!store_ok) gx = r2l_splat2(0.f);  // lanes past the frame's last column, rows past the band's end
        const r2l_p2 gxo = r2l_pmul(gx, og);
        ggam = r2l_pfma(gxo, lg, ggam);
        const r2l_p2 gc = r2l_pmul(r2l_pmul(gxo, r2l_splat2(F.inv_gamma)), r2l_mk2(r2l_rcp(xc[0]), r2l_rcp(xc[1])));
        grgb[k][p] = r2l_mk2((rgb[0] == xc[0]) ? gc[0] : 0.f, (rgb[1] == xc[1]) ? gc[1] : 0.f);  // clip backward
      }
    }
    A.ggam = r2l_padd(A.ggam, ggam);
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) {
      gy2[p] = r2l_pfma(r2l_splat2(F.M2[6]), grgb[2][p],
                        r2l_pfma(r2l_splat2(F.M2[3]), grgb[1][p], r2l_pmul(r2l_splat2(F.M2[0]), grgb[0][p])));
      gu[p] = r2l_pfma(r2l_splat2(F.M2[7]), grgb[2][p],
                       r2l_pfma(r2l_splat2(F.M2[4]), grgb[1][p], r2l_pmul(r2l_splat2(F.M2[1]), grgb[0][p])));
      gv[p] = r2l_pfma(r2l_splat2(F.M2[8]), grgb[2][p],
                       r2l_pfma(r2l_splat2(F.M2[5]), grgb[1][p], r2l_pmul(r2l_splat2(F.M2[2]), grgb[0][p])));
    }
  }
  if (store_ok) {
    r2l_f4 s4;
    s4.x = gy2[0][0];
    s4.y = gy2[0][1];
    s4.z = gy2[1][0];
    s4.w = gy2[1][1];
#ifdef R2L_EXP_GY_NT
    r2l_store_f4_nt(gyb + (unsigned)y * (unsigned)a.W + (unsigned)x0, s4);
#else
    *(r2l_f4*)(gyb + (unsigned)y * (unsigned)a.W + (unsigned)x0) = s4;
#endif
  }
  // ---- folded chroma stencils of this row's parity ----------------------------------------------------------------------
  const float* rows[3] = {vu, vm, vl};
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i) {
    r2l_p2 x[3][2];  // the window row's pairs at column offsets 0, 1, 2 (r2l_row_pairs: straddles as one v_pk_mov_b32)
    r2l_row_pairs(rows[i], x);
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 3; ++j) {
      r2l_p2 su = A.gau[i * 3 + j], sv = A.gav[i * 3 + j];
      R2L_PRAGMA_UNROLL
      for (int p = 0; p < 2; ++p) {
        su = r2l_pfma(gu[p], x[j][p], su);
        sv = r2l_pfma(gv[p], x[j][p], sv);
      }
      A.gau[i * 3 + j] = su;
      A.gav[i * 3 + j] = sv;
    }
  }
  A.su = r2l_padd(A.su, r2l_padd(gu[0], gu[1]));
  A.sv = r2l_padd(A.sv, r2l_padd(gv[0], gv[1]));
}

// value of global slot i (layout R2L_B1_*, r2l_common.h) held by a lane: E = the even-row bank, O = the odd-row bank
R2L_HD float r2l_bp_slot(const R2LBpAcc& A, const float* E, const float* O, int i) {
  if (i < R2L_B1_GAU) return 0.f;  // (the blur-weight sums: r2l_bwd1_blur_block, which runs behind this kernel)
  if (i < R2L_B1_SU) {
    const int tbl = (i - R2L_B1_GAU) / 36, k = (i - R2L_B1_GAU) % 36, par = k / 9, t = k % 9;
    const float* b = (par >> 1) ? O : E;  // bank layout: gau[9] pairs, gav[9] pairs, su, sv
    return b[(tbl * 9 + t) * 2 + (par & 1)];
  }
  if (i < R2L_B1_GGAM) {
    const int tbl = (i - R2L_B1_SU) / 4, par = (i - R2L_B1_SU) % 4;
    const float* b = (par >> 1) ? O : E;
    return b[(18 + tbl) * 2 + (par & 1)];
  }
  return A.ggam[0] + A.ggam[1];
}

// lanes -> one partial per slot and workgroup, in a fixed order (R2L_BLOCK_REDUCE_F for R2L_BP_NT threads): slots
// [0, NSLOTS) of val(i) go to partial[(slot0 + i) * nblk + bid].  The last workgroup's copy of this is part of the launch's
// tail: one wait for the coherent stores, behind the last batch of slots (not one per batch).  (Adding the 64 lanes of a
// wavefront in registers first -- six DPP steps per slot -- was measured and is slower: ~100 cycles per slot and
// wavefront on the vector unit, against LDS traffic that runs beside it; B1 +4 us, profiles/r04_tails.txt.)
// ROWS slots go through LDS at a time (LDS floats: R2L_BP_RED_FLOATS_R(ROWS, NT)): every batch costs three barriers
template <int NSLOTS, int NT = R2L_BP_NT, int ROWS = 32, class VAL>
R2L_BLOCKFN void r2l_bp_block_reduce(float* lds, int tid, float* partial, int slot0, int bid, int nblk, VAL&& val) {
  static_assert(ROWS <= NT, "one lane per slot in the last stage");
  R2L_PRAGMA_UNROLL
  for (int base = 0; base < NSLOTS; base += ROWS) {
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < ROWS; ++i)
      if (base + i < NSLOTS) lds[i * (NT + 1) + tid] = val(base + i);
    R2L_LDS_BARRIER();
    {
      const int slot = tid >> 4, part = tid & 15;  // 16 lanes per slot add NT / 16 values each; NT / 16 slots per pass
      R2L_PRAGMA_UNROLL
      for (int h = 0; h < (ROWS * 16 + NT - 1) / NT; ++h) {
        const int sl = slot + h * (NT / 16);
        if (sl < ROWS) {
          float s = 0.f;
          if (base + sl < NSLOTS) {
            R2L_PRAGMA_UNROLL
            for (int j = 0; j < NT / 16; ++j) s += lds[sl * (NT + 1) + part + 16 * j];
          }
          lds[ROWS * (NT + 1) + sl * 16 + part] = s;
        }
      }
    }
    R2L_LDS_BARRIER();
    if (tid < ROWS && base + tid < NSLOTS) {
      float s = 0.f;
      R2L_PRAGMA_UNROLL
      for (int j = 0; j < 16; ++j) s += lds[ROWS * (NT + 1) + tid * 16 + j];
      r2l_store_coherent(&partial[(size_t)(slot0 + base + tid) * nblk + bid], s);
    }
    if (base + ROWS >= NSLOTS) R2L_STORES_DONE();  // B2's last workgroups finish the reduction
    R2L_LDS_BARRIER();
  }
}

template <bool U16, bool EPI>
R2L_BLOCKFN void r2l_bwd1_plane_block(const R2LBwd1Args& a, int bid, int nblk, float* lds) {
  constexpr int NWV = R2L_BP_NWV, NT = R2L_BP_NT;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  float* bank = lds + (size_t)wave * 64 * R2L_BP_BANK + lane * 4;  // [chunk][lane][4]
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 10; ++c) {
    r2l_f4 z;
    z.x = z.y = z.z = z.w = 0.f;
    *(r2l_f4*)(bank + c * 64 * 4) = z;
  }
  R2LBpAcc A;
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 9; ++i) A.gau[i] = A.gav[i] = r2l_splat2(0.f);
  A.su = A.sv = A.ggam = r2l_splat2(0.f);
  R2LBnConsts bc;
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) {
    bc.mean[k] = a.bn ? a.bn[k] : 0.f;
    bc.istd[k] = a.bn ? a.bn[3 + k] : 1.f;
    bc.mg[k] = a.bn_bwd ? bc.istd[k] * a.bn_bwd[k] : 0.f;
    bc.mgx[k] = a.bn_bwd ? bc.istd[k] * a.bn_bwd[3 + k] : 0.f;
  }
  // the streaming kernels' argument block, for their raw-row fetch / convert (r2l_fa_fetch_raw, r2l_fs_convert)
  R2LFwdStreamArgs sa;
  sa.raw = a.raw;
  sa.W = a.W;
  sa.H = a.H;
  const int nstrip = (a.W + 255) >> 8;
  const unsigned plane = (unsigned)a.H * (unsigned)a.W;
  const int band_h = a.band_h, nband = (a.H + band_h - 1) / band_h, nitems = a.B * nband * nstrip;
  constexpr int PF = R2L_BP_PF, PFG = R2L_BP_PFG;
  static_assert(6 % PF == 0 && 6 % PFG == 0, "the prefetch rings are indexed by the unroll position");
  // the registers hold the bank of EVEN rows between items (every band starts on an even row, at K = 0)
  R2L_PRAGMA_NOUNROLL
  for (int item = r2l_xcd_window(bid, nblk, a.xcdm) * NWV + wave; item < nitems; item += nblk * NWV) {
    const int strip = item % nstrip, ib = item / nstrip;
    const int band = ib % nband, b = ib / nband;
    const int xs = strip * 256 + 4 * lane;
    const bool in_w = xs < a.W;
    const int x0 = in_w ? xs : a.W - 4;
    const bool le = x0 == 0, re = x0 + 4 >= a.W;
    const int y0 = band * band_h;  // a multiple of 6
    const int y1 = (y0 + band_h < a.H) ? y0 + band_h : a.H;
    const size_t img = (size_t)b * plane;
    const float* ypimg = a.yp + img;
    const float* gimg = a.gout + (size_t)b * 3 * plane;
    float* gyb = a.gypp + img;
    R2LBpState st;
    R2LFsStage pf[PF];   // raw row q + 1
    R2LFaStage pfy[PF];  // Y' row q + 2
    R2LBpStage pfg[PFG];  // grad_out row q
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < PF; ++i) {
      r2l_fa_fetch_raw<U16>(sa, img, r2l_mirror(R2L_NH(y0 - 3 + i), a.H), x0, le, re, lane, pf[(2 + i) % PF]);
      r2l_fa_fetch(ypimg, R2L_NH(y0 - 2 + i), a.H, a.W, x0, le, re, lane, pfy[(2 + i) % PF]);
    }
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < PFG; ++i) r2l_bp_fetch_g<EPI>(gimg, plane, y0 + i, a.H, a.W, x0, a.ep, pfg[i % PFG]);  // (first used at K = 0)
#define R2L_BP_LOAD_STEP(K, q)                                                                          \
  r2l_fs_convert<U16>(sa, F, pf[(K) % PF], le, re, st.v[((K) + 1) % 3]);                                \
  r2l_fa_build(pfy[(K) % PF], (unsigned)((q) + 2) < (unsigned)a.H, le, re, st.yp[((K) + 2) % 6]);       \
  r2l_fa_fetch_raw<U16>(sa, img, r2l_mirror(R2L_NH((q) + 1 + PF), a.H), x0, le, re, lane, pf[(K) % PF]);        \
  r2l_fa_fetch(ypimg, R2L_NH((q) + 2 + PF), a.H, a.W, x0, le, re, lane, pfy[(K) % PF]);
    R2L_BP_LOAD_STEP(2, y0 - 4)
    R2L_BP_LOAD_STEP(3, y0 - 3)
    R2L_BP_LOAD_STEP(4, y0 - 2)
    R2L_BP_LOAD_STEP(5, y0 - 1)
    // every group of 6 steps runs in full (rows past the band's end: clamped fetches, zero cotangent, nothing stored);
    // 6 is even, so the banks are back in place at the end of a group
    for (int qb = y0; qb < y1; qb += 6) {
      R2L_PROGRESS_PRIO(qb - y0, y1 - y0);
#define R2L_BP_STEP(K)                                                                                  \
  {                                                                                                     \
    const int q = qb + K;                                                                               \
    R2L_PROGRESS_PRIO_STEP(q - y0, y1 - y0);                                                            \
    R2L_BP_LOAD_STEP(K, q)                                                                              \
    const R2LBpStage g_ = pfg[(K) % PFG];                                                               \
    r2l_bp_fetch_g<EPI>(gimg, plane, q + PFG, a.H, a.W, x0, a.ep, pfg[(K) % PFG]);                      \
    if (K) r2l_bp_swap(A, bank); /* the bank of this row's parity into the registers */                 \
    if (r2l_opaque_true()) r2l_bp_step<K>(a, st, A, g_, q, in_w && q < y1, gyb, x0, bc);                \
  }
      R2L_BP_STEP(0)
      R2L_BP_STEP(1)
      R2L_BP_STEP(2)
      R2L_BP_STEP(3)
      R2L_BP_STEP(4)
      R2L_BP_STEP(5)
      r2l_bp_swap(A, bank);  // (K = 5 was an odd row)
#undef R2L_BP_STEP
    }
#undef R2L_BP_LOAD_STEP
  }
  // ---- lanes -> one partial per slot and workgroup, fixed order (R2L_BLOCK_REDUCE_F for NT threads) ----------------------
  float E[R2L_BP_BANK], O[R2L_BP_BANK];
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 9; ++i) {
    E[2 * i] = A.gau[i][0];
    E[2 * i + 1] = A.gau[i][1];
    E[18 + 2 * i] = A.gav[i][0];
    E[18 + 2 * i + 1] = A.gav[i][1];
  }
  E[36] = A.su[0];
  E[37] = A.su[1];
  E[38] = A.sv[0];
  E[39] = A.sv[1];
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 10; ++c) {
    const r2l_f4 o4 = r2l_lds_f4(bank + c * 64 * 4);
    O[4 * c] = o4.x;
    O[4 * c + 1] = o4.y;
    O[4 * c + 2] = o4.z;
    O[4 * c + 3] = o4.w;
  }
  R2L_LDS_BARRIER();  // the bank areas become reduction scratch
  // (not the blur-weight sums, slots < R2L_B1_GAU: the blur pass that runs behind this launch on the same grid writes them)
  r2l_bp_block_reduce<R2L_B1_NACC - R2L_B1_GAU, R2L_BP_NT, R2L_BP_ROWS>(
      lds, tid, a.partial, R2L_B1_GAU, bid, nblk, [&](int i) { return r2l_bp_slot(A, E, O, R2L_B1_GAU + i); });
}

// ---- second pass: the 25 blur-weight sums  d/d gaussian_blur.weight[i][j] = sum_p gY''(p) * Y'_ext(p + (i-2, j-2)) from
// the dL/dY'' plane the first pass wrote and the kept Y' plane --------------------------------------------------------
struct R2LBbStage {
  r2l_f4 g;  // dL/dY'' of the lane's 4 pixels
};
template <int K>
R2L_HD void r2l_bb_step(const R2LBwd1Args& a, const float yp[6][8], r2l_p2 blur[25], const R2LBbStage& gs, int y,
                        bool ok) {
  const int H = a.H;
  float yw[5][8];
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 5; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 8; ++j) yw[i][j] = yp[(K + 4 + i) % 6][j];
  r2l_p2 gy2[2];
  gy2[0] = r2l_mk2(ok ? gs.g.x : 0.f, ok ? gs.g.y : 0.f);  // (lanes past the last column, rows past the band's end)
  gy2[1] = r2l_mk2(ok ? gs.g.z : 0.f, ok ? gs.g.w : 0.f);
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 5; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 5; ++j) {
    r2l_p2 s = blur[i * 5 + j];
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) s = r2l_pfma(gy2[p], r2l_mk2(yw[i][2 * p + j], yw[i][2 * p + j + 1]), s);
    blur[i * 5 + j] = s;
  }
  // ... and the mirror padding's share on the first / last two image rows (window rows outside the image are zero): row
  // -1 IS row 1 (window row 2 of y = 1, 3 of y = 0), row -2 is row 2 (window row 4 of y = 0); row H is row H-2, row H+1
  // is row H-3 likewise.  src = the window row that holds the mirrored row, dst = the window row it stands in for.
#define R2L_BP_MIRROR(dst, src)                                                                          \
  R2L_PRAGMA_UNROLL                                                                                      \
  for (int j = 0; j < 5; ++j) {                                                                          \
    r2l_p2 s = blur[(dst) * 5 + j];                                                                      \
    R2L_PRAGMA_UNROLL                                                                                    \
    for (int p = 0; p < 2; ++p) s = r2l_pfma(gy2[p], r2l_mk2(yw[src][2 * p + j], yw[src][2 * p + j + 1]), s); \
    blur[(dst) * 5 + j] = s;                                                                             \
  }
  if (y < 2 || y >= H - 2) {  // uniform
    if (y == 0) {
      R2L_BP_MIRROR(0, 4)
      R2L_BP_MIRROR(1, 3)
    }
    if (y == 1) R2L_BP_MIRROR(0, 2)
    if (y == H - 2) R2L_BP_MIRROR(4, 2)
    if (y == H - 1) {
      R2L_BP_MIRROR(3, 1)
      R2L_BP_MIRROR(4, 0)
    }
  }
#undef R2L_BP_MIRROR
}
#ifndef R2L_BB_PF
#define R2L_BB_PF 2
#endif
R2L_BLOCKFN void r2l_bwd1_blur_block(const R2LBwd1Args& a, int bid, int nblk, float* lds) {
  constexpr int NWV = R2L_BP_NWV;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  r2l_p2 blur[25];
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 25; ++i) blur[i] = r2l_splat2(0.f);
  const int nstrip = (a.W + 255) >> 8;
  const unsigned plane = (unsigned)a.H * (unsigned)a.W;
  const int band_h = a.band_h, nband = (a.H + band_h - 1) / band_h, nitems = a.B * nband * nstrip;
  constexpr int PF = R2L_BB_PF;
  static_assert(6 % PF == 0, "the prefetch ring is indexed by the unroll position");
  R2L_PRAGMA_NOUNROLL
  for (int item = r2l_xcd_window(bid, nblk, a.xcdm_hb) * NWV + wave; item < nitems; item += nblk * NWV) {
    const int strip = item % nstrip, ib = item / nstrip;
    const int band = ib % nband, b = ib / nband;
    const int xs = strip * 256 + 4 * lane;
    const bool in_w = xs < a.W;
    const int x0 = in_w ? xs : a.W - 4;
    const bool le = x0 == 0, re = x0 + 4 >= a.W;
    const int y0 = band * band_h;  // a multiple of 6
    const int y1 = (y0 + band_h < a.H) ? y0 + band_h : a.H;
    const size_t img = (size_t)b * plane;
    const float* ypimg = a.yp + img;
    const float* gimg = a.gypp + img;
    float yp[6][8];
    R2LFaStage pfy[PF];  // Y' row q + 2
    R2LBbStage pfg[PF];  // dL/dY'' row q
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < PF; ++i) {
      r2l_fa_fetch<R2L_HB_YP_NT>(ypimg, R2L_NH(y0 - 2 + i), a.H, a.W, x0, le, re, lane, pfy[(2 + i) % PF]);
      const int yc = (y0 + i < a.H) ? y0 + i : a.H - 1;
      pfg[i % PF].g = R2L_HB_GY_LOAD(gimg + (size_t)yc * a.W + x0);
    }
#define R2L_BB_LOAD_STEP(K, q)                                                                          \
  r2l_fa_build(pfy[(K) % PF], (unsigned)((q) + 2) < (unsigned)a.H, le, re, yp[((K) + 2) % 6]);          \
  r2l_fa_fetch<R2L_HB_YP_NT>(ypimg, R2L_NH((q) + 2 + PF), a.H, a.W, x0, le, re, lane, pfy[(K) % PF]);
    R2L_BB_LOAD_STEP(2, y0 - 4)
    R2L_BB_LOAD_STEP(3, y0 - 3)
    R2L_BB_LOAD_STEP(4, y0 - 2)
    R2L_BB_LOAD_STEP(5, y0 - 1)
    for (int qb = y0; qb < y1; qb += 6) {
      R2L_PROGRESS_PRIO(qb - y0, y1 - y0);
#define R2L_BB_STEP(K)                                                                                  \
  {                                                                                                     \
    const int q = qb + K;                                                                               \
    R2L_PROGRESS_PRIO_STEP(q - y0, y1 - y0);                                                            \
    R2L_BB_LOAD_STEP(K, q)                                                                              \
    const R2LBbStage g_ = pfg[(K) % PF];                                                                \
    {                                                                                                   \
      const int yc = (q + PF < a.H) ? q + PF : a.H - 1;                                                 \
      pfg[(K) % PF].g = R2L_HB_GY_LOAD(gimg + (size_t)yc * a.W + x0);                               \
    }                                                                                                   \
    if (r2l_opaque_true()) r2l_bb_step<K>(a, yp, blur, g_, q, in_w && q < y1);                          \
  }
      R2L_BB_STEP(0)
      R2L_BB_STEP(1)
      R2L_BB_STEP(2)
      R2L_BB_STEP(3)
      R2L_BB_STEP(4)
      R2L_BB_STEP(5)
#undef R2L_BB_STEP
    }
#undef R2L_BB_LOAD_STEP
  }
  r2l_bp_block_reduce<R2L_B1_GAU, R2L_BP_NT>(lds, tid, a.partial, 0, bid, nblk, [&](int i) { return blur[i][0] + blur[i][1]; });
}

// ================================================================================================
// Kernel B2 (r2l_bwd2_block) as two passes over planes, the same way:
//   r2l_bwd2_hp_block    dL/dY''  -> HP = adjoint of the 5x5 blur with mirror padding (a plane in the workspace)
//   r2l_bwd2_sums_block  HP + raw -> the sharpen's adjoint (zero padding), d/d sharpening_filter.weight, d/d luma stencil
//                                    sums; its last workgroups finish the reduction of B1's and B2's partials and unfold
// The adjoint of a mirror-padded correlation: HP(q) = sum_t blur[t] g0(q - t) (g0 = dL/dY'' extended with zeros) for every
// position, and an in-image position adds the values of its out-of-image mirror images -- rows {1, 2, H-3, H-2} those
// of rows {-1, -2, H+1, H}, columns likewise, corners both (r2l_fold_mirror of the tile kernel).  Here an out-of-image
// row's value is a few more products of the window rows at hand (row -1 only sees g rows 0 and 1, ...), added behind a
// uniform branch on those four rows; an out-of-image column's value involves the lane's own first / last two columns
// only and is carried along as one extra pair per side, used by the lanes at the image's edges.
struct R2LHpAcc {
  r2l_p2 h[2];  // HP of the lane's 4 columns
  r2l_p2 el;    // (HP(-2), HP(-1)): out-of-image columns left of the image (lanes with x0 == 0)
  r2l_p2 er;    // (HP(W), HP(W+1))
};
// one window row (8 wide: columns x0-2 .. x0+5, zero outside the image) against blur row i, taps in adjoint order
template <class WT>
R2L_HD void r2l_hp_row(R2LHpAcc& A, const float g[8], WT blur, int i) {
  float bf[5];  // bf[s] = blur[i][4 - s]: HP(x) += bf[s] * g(x + s - 2)
  R2L_PRAGMA_UNROLL
  for (int s_ = 0; s_ < 5; ++s_) bf[s_] = blur[i * 5 + 4 - s_];
  r2l_p2 P[4], O[3];
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 4; ++k) P[k] = r2l_mk2(g[2 * k], g[2 * k + 1]);
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) O[k] = r2l_straddle(P[k], P[k + 1]);
  R2L_PRAGMA_UNROLL
  for (int s_ = 0; s_ < 5; ++s_) {
    const r2l_p2 w = r2l_splat2(bf[s_]);
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) A.h[p] = r2l_pfma(w, (s_ & 1) ? O[p + s_ / 2] : P[p + s_ / 2], A.h[p]);
  }
  // columns -2, -1 see the image's columns 0, 1 only; columns W, W+1 its last two (the lane's c0 .. c3 = g[2 .. 5])
  A.el = r2l_pfma(r2l_mk2(bf[4], bf[3]), r2l_splat2(g[2]), A.el);
  A.el = r2l_pfma(r2l_mk2(0.f, bf[4]), r2l_splat2(g[3]), A.el);
  A.er = r2l_pfma(r2l_splat2(bf[0]), r2l_mk2(g[4], g[5]), A.er);
  A.er = r2l_pfma(r2l_mk2(bf[1], 0.f), r2l_splat2(g[5]), A.er);
}
// staged row -> 8 values, columns x0-2 .. x0+5, ZERO outside the image (r2l_fa_build mirrors)
R2L_HD void r2l_hp_build(const R2LFaStage& s, bool rin, bool le, bool re, float o[8]) {
  const float c0 = rin ? s.c.x : 0.f, c1 = rin ? s.c.y : 0.f, c2 = rin ? s.c.z : 0.f, c3 = rin ? s.c.w : 0.f;
  const float e0 = rin ? s.e.x : 0.f, e1 = rin ? s.e.y : 0.f;
  const float l2 = r2l_wshr(c2, e0), l1 = r2l_wshr(c3, e1);
  const float r1 = r2l_wshl(c0, e0), r2 = r2l_wshl(c1, e1);
  o[0] = le ? 0.f : l2;
  o[1] = le ? 0.f : l1;
  o[2] = c0;
  o[3] = c1;
  o[4] = c2;
  o[5] = c3;
  o[6] = re ? 0.f : r1;
  o[7] = re ? 0.f : r2;
}
template <int K, class ArgsT>
R2L_HD void r2l_hp_step(const ArgsT& a, const float gw[6][8], int q, bool le, bool re, bool store_ok, float* hpb,
                        int x0) {
  const int H = a.H;
  R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque_after(a.F, gw[(K + 2) % 6][2]));
  R2LHpAcc A;
  A.h[0] = A.h[1] = A.el = A.er = r2l_splat2(0.f);
  // window rows q-2 .. q+2 sit in ring slots K+4 .. K+8; window row r holds g(q - 2 + r) and meets blur row 4 - r
  R2L_PRAGMA_UNROLL
  for (int r = 0; r < 5; ++r) r2l_hp_row(A, gw[(K + 4 + r) % 6], F.blur, 4 - r);
  if (q < 3 || q >= H - 3) {  // uniform: the rows whose mirror images lie outside the image
    // row -1 (mirror image of row 1) sees g rows 1 and 0 = window rows 2 and 1 of q = 1 through blur rows 0 and 1; ...
    if (q == 1) {
      r2l_hp_row(A, gw[(K + 4 + 2) % 6], F.blur, 0);
      r2l_hp_row(A, gw[(K + 4 + 1) % 6], F.blur, 1);
    }
    if (q == 2) r2l_hp_row(A, gw[(K + 4 + 0) % 6], F.blur, 0);  // row -2: g row 0 = window row 0 through blur row 0
    if (q == H - 2) {  // row H: g rows H-1, H-2 = window rows 3, 2 through blur rows 3, 4
      r2l_hp_row(A, gw[(K + 4 + 3) % 6], F.blur, 3);
      r2l_hp_row(A, gw[(K + 4 + 2) % 6], F.blur, 4);
    }
    if (q == H - 3) r2l_hp_row(A, gw[(K + 4 + 4) % 6], F.blur, 4);  // row H+1: g row H-1 = window row 4, blur row 4
  }
  float h0 = A.h[0][0], h1 = A.h[0][1], h2 = A.h[1][0], h3 = A.h[1][1];
  // columns 1, 2 add columns -1, -2; columns W-2, W-3 add columns W, W+1 (a lane at the right edge holds W-4 .. W-1)
  h1 += le ? A.el[1] : 0.f;
  h2 += le ? A.el[0] : 0.f;
  h2 += re ? A.er[0] : 0.f;
  h1 += re ? A.er[1] : 0.f;
  if (store_ok) {
    r2l_f4 s4;
    s4.x = h0;
    s4.y = h1;
    s4.z = h2;
    s4.w = h3;
#ifdef R2L_EXP_HP_NT
    r2l_store_f4_nt(hpb + (unsigned)q * (unsigned)a.W + (unsigned)x0, s4);
#else
    *(r2l_f4*)(hpb + (unsigned)q * (unsigned)a.W + (unsigned)x0) = s4;
#endif
  }
}
#ifndef R2L_HP_PF
#define R2L_HP_PF 2
#endif
R2L_BLOCKFN void r2l_bwd2_hp_block(const R2LBwd2Args& a, int bid, int nblk, float* lds) {
  (void)lds;
  (void)nblk;
  constexpr int NWV = R2L_BP_NWV;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nstrip = (a.W + 255) >> 8;
  const unsigned plane = (unsigned)a.H * (unsigned)a.W;
  const int band_h = a.band_h, nband = (a.H + band_h - 1) / band_h, nitems = a.B * nband * nstrip;
  const int item = bid * NWV + wave;  // one item per wavefront
  if (item >= nitems) return;
  const int strip = item % nstrip, ib = item / nstrip;
  const int band = ib % nband, b = ib / nband;
  const int xs = strip * 256 + 4 * lane;
  const bool in_w = xs < a.W;
  const int x0 = in_w ? xs : a.W - 4;
  const bool le = x0 == 0, re = x0 + 4 >= a.W;
  const int y0 = band * band_h;  // a multiple of 6
  const int y1 = (y0 + band_h < a.H) ? y0 + band_h : a.H;
  const float* gimg = a.gypp + (size_t)b * plane;
  float* hpb = a.hp + (size_t)b * plane;
  float gw[6][8];
  constexpr int PF = R2L_HP_PF;
  static_assert(6 % PF == 0, "the prefetch ring is indexed by the unroll position");
  R2LFaStage pf[PF];  // g row q + 2
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < PF; ++i) r2l_fa_fetch(gimg, R2L_NH(y0 - 2 + i), a.H, a.W, x0, le, re, lane, pf[(2 + i) % PF]);
#define R2L_HP_LOAD_STEP(K, q)                                                                          \
  r2l_hp_build(pf[(K) % PF], (unsigned)((q) + 2) < (unsigned)a.H, le, re, gw[((K) + 2) % 6]);           \
  r2l_fa_fetch(gimg, R2L_NH((q) + 2 + PF), a.H, a.W, x0, le, re, lane, pf[(K) % PF]);
  R2L_HP_LOAD_STEP(2, y0 - 4)
  R2L_HP_LOAD_STEP(3, y0 - 3)
  R2L_HP_LOAD_STEP(4, y0 - 2)
  R2L_HP_LOAD_STEP(5, y0 - 1)
  for (int qb = y0; qb < y1; qb += 6) {
    R2L_PROGRESS_PRIO(qb - y0, y1 - y0);
#define R2L_HP_STEP(K)                                                                                  \
  {                                                                                                     \
    const int q = qb + K;                                                                               \
    R2L_PROGRESS_PRIO_STEP(q - y0, y1 - y0);                                                            \
    R2L_HP_LOAD_STEP(K, q)                                                                              \
    r2l_hp_step<K>(a, gw, q, le, re, in_w && q < y1, hpb, x0);                                          \
  }
    R2L_HP_STEP(0)
    R2L_HP_STEP(1)
    R2L_HP_STEP(2)
    R2L_HP_STEP(3)
    R2L_HP_STEP(4)
    R2L_HP_STEP(5)
#undef R2L_HP_STEP
  }
#undef R2L_HP_LOAD_STEP
}

// ---- B1's second pass and B2's first pass as ONE pass.  Both walk the dL/dY'' plane with a 5-row window of it: the fused
// pass reads dL/dY'' once -- 12 B/px (4 dL/dY'' + 4 Y' in, 4 HP out) instead of 8 + 8 -- and needs no Y' window: the
// blur-weight sums are taken the adjoint's way round,
//   d/d gaussian_blur.weight[i][j] = sum_p g(p) Y'_ext(p + t) = sum_q Y'_ext(q) g0(q - t),   t = (i-2, j-2),
// over the positions q of the MIRROR-EXTENDED image (g0 = dL/dY'' extended with zeros): every in-image q is a pixel of the
// row at hand against the window the adjoint uses anyway (window row r meets weight row 4 - r), and the out-of-image
// rows / columns are the virtual rows / columns of the adjoint -- the same extra (window row, weight row) calls on rows
// {1, 2, H-3, H-2}, and for the lanes at the image's left / right edge the two virtual columns' few products, whose Y'
// values are the lane's own columns 1, 2 / W-2, W-3.  (The sums group differently from r2l_bwd1_blur_block's: equal to
// round-off, not bit for bit; HP is r2l_hp_row's arithmetic unchanged.)
struct R2LBsY {      // Y' of the row at hand: the lane's 4 pixels as pairs, and the virtual columns' values
  r2l_p2 y01, y23;
  float yl1, yl2;    // Y'(1), Y'(2) in the lanes at the left edge (virtual columns -1, -2), else 0
  float yr2, yr1;    // Y'(W-2), Y'(W-3) in the lanes at the right edge (virtual columns W, W+1), else 0
};
// one window row of dL/dY'' (columns x0-2 .. x0+5, zero outside the image) against weight row i: HP as r2l_hp_row, and
// the 5 blur-weight sums of that row
template <class WT>
R2L_HD void r2l_hb_row(R2LHpAcc& A, r2l_p2 blur[25], const R2LBsY& y, const float g[8], WT bw, int i) {
  float bf[5];  // bf[s] = blur[i][4 - s]: HP(x) += bf[s] * g(x + s - 2)
  R2L_PRAGMA_UNROLL
  for (int s_ = 0; s_ < 5; ++s_) bf[s_] = bw[i * 5 + 4 - s_];
  r2l_p2 P[4], O[3];
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 4; ++k) P[k] = r2l_mk2(g[2 * k], g[2 * k + 1]);
  R2L_PRAGMA_UNROLL
  for (int k = 0; k < 3; ++k) O[k] = r2l_straddle(P[k], P[k + 1]);
  R2L_PRAGMA_UNROLL
  for (int s_ = 0; s_ < 5; ++s_) {
    const r2l_p2 w = r2l_splat2(bf[s_]);
    R2L_PRAGMA_UNROLL
    for (int p = 0; p < 2; ++p) A.h[p] = r2l_pfma(w, (s_ & 1) ? O[p + s_ / 2] : P[p + s_ / 2], A.h[p]);
  }
  A.el = r2l_pfma(r2l_mk2(bf[4], bf[3]), r2l_splat2(g[2]), A.el);
  A.el = r2l_pfma(r2l_mk2(0.f, bf[4]), r2l_splat2(g[3]), A.el);
  A.er = r2l_pfma(r2l_splat2(bf[0]), r2l_mk2(g[4], g[5]), A.er);
  A.er = r2l_pfma(r2l_mk2(bf[1], 0.f), r2l_splat2(g[5]), A.er);
  // weight column j meets g(x + 2 - j): the pairs (g[4-j], g[5-j]) and (g[6-j], g[7-j]) -- P for even j, O for odd j
  R2L_PRAGMA_UNROLL
  for (int j = 0; j < 5; ++j) {
    const int k = (4 - j) >> 1;
    r2l_p2 s_ = blur[i * 5 + j];
    s_ = r2l_pfma(y.y01, (j & 1) ? O[k] : P[k], s_);
    s_ = r2l_pfma(y.y23, (j & 1) ? O[k + 1] : P[k + 1], s_);
    blur[i * 5 + j] = s_;
  }
  // virtual columns: -1 (= column 1) sees g columns 0, 1 through weight columns 1, 0; -2 (= column 2) sees column 0 through
  // weight column 0; W (= W-2) sees columns W-1, W-2 through weight columns 3, 4; W+1 (= W-3) sees W-1 through 4
  blur[i * 5 + 1][0] = fmaf(y.yl1, g[2], blur[i * 5 + 1][0]);
  blur[i * 5 + 0][0] = fmaf(y.yl1, g[3], fmaf(y.yl2, g[2], blur[i * 5 + 0][0]));
  blur[i * 5 + 3][1] = fmaf(y.yr2, g[5], blur[i * 5 + 3][1]);
  blur[i * 5 + 4][1] = fmaf(y.yr2, g[4], fmaf(y.yr1, g[5], blur[i * 5 + 4][1]));
}
template <int K>
R2L_HD void r2l_hb_step(const R2LBwd1Args& a, const float gw[6][8], r2l_p2 blur[25], const r2l_f4& yrow, int q, bool le,
                        bool re, bool ok, float* hpb, int x0) {
  const int H = a.H;
  R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque_after(a.F, gw[(K + 2) % 6][2]));
  R2LHpAcc A;
  A.h[0] = A.h[1] = A.el = A.er = r2l_splat2(0.f);
  R2LBsY y;  // (lanes past the last column and rows past the band's end contribute nothing)
  const float y0 = ok ? yrow.x : 0.f, y1 = ok ? yrow.y : 0.f, y2 = ok ? yrow.z : 0.f, y3 = ok ? yrow.w : 0.f;
  y.y01 = r2l_mk2(y0, y1);
  y.y23 = r2l_mk2(y2, y3);
  y.yl1 = le ? y1 : 0.f;
  y.yl2 = le ? y2 : 0.f;
  y.yr2 = re ? y2 : 0.f;
  y.yr1 = re ? y1 : 0.f;
  // window rows q-2 .. q+2 sit in ring slots K+4 .. K+8; window row r holds g(q - 2 + r) and meets weight row 4 - r
  R2L_PRAGMA_UNROLL
  // (one window row at a time: left alone the scheduler interleaves the five rows' independent chains and holds all their
  // pairs at once -- 256 registers and scratch)
  for (int r = 0; r < 5; ++r) {
    r2l_hb_row(A, blur, y, gw[(K + 4 + r) % 6], F.blur, 4 - r);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (q < 3 || q >= H - 3) {  // uniform: the rows whose mirror images lie outside the image (as r2l_hp_step)
    if (q == 1) {
      r2l_hb_row(A, blur, y, gw[(K + 4 + 2) % 6], F.blur, 0);
      r2l_hb_row(A, blur, y, gw[(K + 4 + 1) % 6], F.blur, 1);
    }
    if (q == 2) r2l_hb_row(A, blur, y, gw[(K + 4 + 0) % 6], F.blur, 0);
    if (q == H - 2) {
      r2l_hb_row(A, blur, y, gw[(K + 4 + 3) % 6], F.blur, 3);
      r2l_hb_row(A, blur, y, gw[(K + 4 + 2) % 6], F.blur, 4);
    }
    if (q == H - 3) r2l_hb_row(A, blur, y, gw[(K + 4 + 4) % 6], F.blur, 4);
  }
  float h0 = A.h[0][0], h1 = A.h[0][1], h2 = A.h[1][0], h3 = A.h[1][1];
  h1 += le ? A.el[1] : 0.f;
  h2 += le ? A.el[0] : 0.f;
  h2 += re ? A.er[0] : 0.f;
  h1 += re ? A.er[1] : 0.f;
  if (ok) {
    r2l_f4 s4;
    s4.x = h0;
    s4.y = h1;
    s4.z = h2;
    s4.w = h3;
#ifdef R2L_EXP_HP_NT
    r2l_store_f4_nt(hpb + (unsigned)q * (unsigned)a.W + (unsigned)x0, s4);
#else
    *(r2l_f4*)(hpb + (unsigned)q * (unsigned)a.W + (unsigned)x0) = s4;
#endif
  }
}
#ifndef R2L_HB_OCC
#define R2L_HB_OCC 2
#endif
R2L_BLOCKFN void r2l_bwd1_blur_hp_block(const R2LBwd1Args& a, int bid, int nblk, float* lds) {
  constexpr int NWV = R2L_BP_NWV;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  r2l_p2 blur[25];
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 25; ++i) blur[i] = r2l_splat2(0.f);
  const int nstrip = (a.W + 255) >> 8;
  const unsigned plane = (unsigned)a.H * (unsigned)a.W;
  const int band_h = a.band_hb, nband = (a.H + band_h - 1) / band_h, nitems = a.B * nband * nstrip;
  constexpr int PF = R2L_BB_PF;
  static_assert(6 % PF == 0, "the prefetch ring is indexed by the unroll position");
  R2L_PRAGMA_NOUNROLL
  for (int item = r2l_xcd_window(bid, nblk, a.xcdm_hb) * NWV + wave; item < nitems; item += nblk * NWV) {
    const int strip = item % nstrip, ib = item / nstrip;
    const int band = ib % nband, b = ib / nband;
    const int xs = strip * 256 + 4 * lane;
    const bool in_w = xs < a.W;
    const int x0 = in_w ? xs : a.W - 4;
    const bool le = x0 == 0, re = x0 + 4 >= a.W;
    const int y0 = band * band_h;  // a multiple of 6
    const int y1 = (y0 + band_h < a.H) ? y0 + band_h : a.H;
    const size_t img = (size_t)b * plane;
    const float* ypimg = a.yp + img;
    const float* gimg = a.gypp + img;
    float* hpb = a.hp + img;
    float gw[6][8];      // dL/dY'' rows, zero outside the image (slot = row mod 6)
    R2LFaStage pfg[PF];  // dL/dY'' row q + 2
    r2l_f4 pfy[PF];      // Y' row q
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < PF; ++i) {
      r2l_fa_fetch(gimg, R2L_NH(y0 - 2 + i), a.H, a.W, x0, le, re, lane, pfg[(2 + i) % PF]);
      const int yc = (y0 + i < a.H) ? y0 + i : a.H - 1;
      pfy[i % PF] = r2l_stream_load_f4(ypimg + (size_t)yc * a.W + x0);
    }
#define R2L_BH_LOAD_STEP(K, q)                                                                          \
  r2l_hp_build(pfg[(K) % PF], (unsigned)((q) + 2) < (unsigned)a.H, le, re, gw[((K) + 2) % 6]);          \
  r2l_fa_fetch(gimg, R2L_NH((q) + 2 + PF), a.H, a.W, x0, le, re, lane, pfg[(K) % PF]);
    R2L_BH_LOAD_STEP(2, y0 - 4)
    R2L_BH_LOAD_STEP(3, y0 - 3)
    R2L_BH_LOAD_STEP(4, y0 - 2)
    R2L_BH_LOAD_STEP(5, y0 - 1)
    for (int qb = y0; qb < y1; qb += 6) {
      R2L_PROGRESS_PRIO(qb - y0, y1 - y0);
#define R2L_BH_STEP(K)                                                                                  \
  {                                                                                                     \
    const int q = qb + K;                                                                               \
    R2L_PROGRESS_PRIO_STEP(q - y0, y1 - y0);                                                            \
    R2L_BH_LOAD_STEP(K, q)                                                                              \
    const r2l_f4 y_ = pfy[(K) % PF];                                                                    \
    {                                                                                                   \
      const int yc = (q + PF < a.H) ? q + PF : a.H - 1;                                                 \
      pfy[(K) % PF] = r2l_stream_load_f4(ypimg + (size_t)yc * a.W + x0);                                \
    }                                                                                                   \
    if (r2l_opaque_true()) r2l_hb_step<K>(a, gw, blur, y_, q, le, re, in_w && q < y1, hpb, x0);         \
  }
      R2L_BH_STEP(0)
      R2L_BH_STEP(1)
      R2L_BH_STEP(2)
      R2L_BH_STEP(3)
      R2L_BH_STEP(4)
      R2L_BH_STEP(5)
#undef R2L_BH_STEP
    }
#undef R2L_BH_LOAD_STEP
  }
  r2l_bp_block_reduce<R2L_B1_GAU, R2L_BP_NT>(lds, tid, a.partial, 0, bid, nblk, [&](int i) { return blur[i][0] + blur[i][1]; });
}

// ---- second pass of B2: HP + raw -> gY = sharpen^T(HP) (zero padding), the sums  d/d sharpening_filter.weight[t] =
// sum gY'(p) Y0(p + t)  (gY' = HP, Y0 = luma extended with zeros) and  GAY[parity][t] = sum gY(p) v(p + t), SY ---------
#define R2L_B2S_NWV 4                      // one wavefront per SIMD and workgroup, 3 workgroups per CU (6-wavefront workgroups
                                           // spread 2-2-1-1 over the SIMDs: a second one only fits if it lands 1-1-2-2)
#define R2L_B2S_NT (64 * R2L_B2S_NWV)
#define R2L_B2S_BANK 20                    // floats of one row-parity bank: GAY[9] pairs, SY pair
// reduction scratch (>= the bank areas and the tree's scratch): all 49 slots in one batch -- 53.5 KB, three workgroups per CU
// still fit the 160 KB
#define R2L_B2S_LDS_FLOATS R2L_BP_RED_FLOATS_R(R2L_B2_NACC, R2L_B2S_NT)
struct R2LSumStage {  // one HP row in flight: the lane's 4 values + the neighbour beyond the strip edge
  r2l_f4 c;
  float e;
};
R2L_HD void r2l_b2s_fetch_hp(const float* hpimg, int r, int H, int W, int x0, bool le, bool re, int lane, R2LSumStage& s) {
  const int rc = r < 0 ? 0 : (r >= H ? H - 1 : r);
  const float* p = hpimg + (size_t)rc * W + x0;
  const int eo = (lane < 32) ? (le ? 0 : -1) : (re ? 3 : 4);
  s.c = R2L_B2S_HP_LOAD(p);
  s.e = p[eo];
}
// staged row -> 6 values, columns x0-1 .. x0+4, zero outside the image (the sharpen's zero padding has no adjoint there)
R2L_HD void r2l_b2s_build_hp(const R2LSumStage& s, bool rin, bool le, bool re, float o[6]) {
  const float c0 = rin ? s.c.x : 0.f, c1 = rin ? s.c.y : 0.f, c2 = rin ? s.c.z : 0.f, c3 = rin ? s.c.w : 0.f;
  const float e = rin ? s.e : 0.f;
  const float l = r2l_wshr(c3, e), r = r2l_wshl(c0, e);
  o[0] = le ? 0.f : l;
  o[1] = c0;
  o[2] = c1;
  o[3] = c2;
  o[4] = c3;
  o[5] = re ? 0.f : r;
}
struct R2LSumState {
  float v[3][6];     // V rows (slot = row mod 3), columns x0-1 .. x0+4
  r2l_p2 xp[3][3];   // the inputs of the luma stencil's extra pair (r2l_fl_convert)
  float y[3][6];     // Y rows, zero outside the image
  float hp[3][6];    // HP rows, zero outside the image
};
struct R2LSumAcc {
  r2l_p2 gsh[9];  // sum gY'(p) * Y0(p + t); halves = column parity, added at the end
  r2l_p2 gay[9];  // bank of the CURRENT row parity: sum gY(p) * v(p + t), halves = column parity
  r2l_p2 sy;
};
R2L_HD void r2l_b2s_swap(R2LSumAcc& A, float* bank /* [chunk][lane][4], 5 chunks */) {
#define R2L_B2S_SW(c, a, b)                     \
  {                                             \
    const r2l_f4 old = r2l_lds_f4(bank + (c) * 64 * 4); \
    r2l_f4 nw;                                  \
    nw.x = (a)[0];                              \
    nw.y = (a)[1];                              \
    nw.z = (b)[0];                              \
    nw.w = (b)[1];                              \
    *(r2l_f4*)(bank + (c) * 64 * 4) = nw;       \
    (a) = r2l_mk2(old.x, old.y);                \
    (b) = r2l_mk2(old.z, old.w);                \
  }
  R2L_B2S_SW(0, A.gay[0], A.gay[1])
  R2L_B2S_SW(1, A.gay[2], A.gay[3])
  R2L_B2S_SW(2, A.gay[4], A.gay[5])
  R2L_B2S_SW(3, A.gay[6], A.gay[7])
  R2L_B2S_SW(4, A.gay[8], A.sy)
#undef R2L_B2S_SW
}
// Y(t) from V(t-1 .. t+1) (the luma stencil of the forward, r2l_fl_step), zero outside the image; K = t mod 6
template <int K>
R2L_HD void r2l_b2s_luma(const R2LBwd2Args& a, R2LSumState& st, int t, bool le, bool re) {
  R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque_after(a.F, st.v[(K + 1) % 3][2]));
  constexpr int PY = K & 1;
  float* yq = st.y[K % 3];
  r2l_p2 o[2];
  r2l_fs_stencil_parity(st.v[(K + 2) % 3], st.v[K % 3], st.v[(K + 1) % 3], F.AY2[PY], o);  // V(t-1), V(t), V(t+1)
  r2l_p2 e = r2l_splat2(0.f);  // (Y(x0+4), Y(x0-1)): the columns beyond the lane's four
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 3; ++i)
    R2L_PRAGMA_UNROLL
  for (int j = 0; j < 3; ++j)
    e = r2l_pfma(r2l_mk2(F.AY2[PY][i * 3 + j][0], F.AY2[PY][i * 3 + j][1]), st.xp[(K + 2 + i) % 3][j], e);
  const bool qin = (unsigned)t < (unsigned)a.H;
  yq[0] = (qin && !le) ? e[1] : 0.f;
  yq[1] = qin ? o[0][0] : 0.f;
  yq[2] = qin ? o[0][1] : 0.f;
  yq[3] = qin ? o[1][0] : 0.f;
  yq[4] = qin ? o[1][1] : 0.f;
  yq[5] = (qin && !re) ? e[0] : 0.f;
}
// the sums of step t: GAY / SY of row t (bank of parity K & 1 in the registers), sharpen-weight sums of row t - 1
template <int K, bool ROW_T = true>
R2L_HD void r2l_b2s_sums(const R2LBwd2Args& a, R2LSumState& st, R2LSumAcc& A, bool ok_t, bool ok_tm1) {
  if (ROW_T) {
    R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque_after(a.F, st.hp[(K + 1) % 3][2]));
    // gY(c) = sum_{i,j} sharp[i][j] * HP(t - (i-1), c - (j-1)): window rows HP(t-1), HP(t), HP(t+1) = slots K+2, K, K+1
    r2l_p2 gy[2];
    gy[0] = gy[1] = r2l_splat2(0.f);
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i) {
      r2l_p2 x[3][2];
      r2l_row_pairs(st.hp[(K + 2 + (2 - i)) % 3], x);  // window row 2 - i
      R2L_PRAGMA_UNROLL
      for (int j = 0; j < 3; ++j) {
        const r2l_p2 w = r2l_splat2(F.sharp[i * 3 + j]);
        R2L_PRAGMA_UNROLL
        for (int p = 0; p < 2; ++p) gy[p] = r2l_pfma(w, x[2 - j][p], gy[p]);
      }
    }
    if (!ok_t) gy[0] = gy[1] = r2l_splat2(0.f);
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i) {
      r2l_p2 x[3][2];
      r2l_row_pairs(st.v[(K + 2 + i) % 3], x);  // V(t - 1 + i)
      R2L_PRAGMA_UNROLL
      for (int j = 0; j < 3; ++j) {
        r2l_p2 sacc = A.gay[i * 3 + j];
        R2L_PRAGMA_UNROLL
        for (int p = 0; p < 2; ++p) sacc = r2l_pfma(gy[p], x[j][p], sacc);
        A.gay[i * 3 + j] = sacc;
      }
    }
    A.sy = r2l_padd(A.sy, r2l_padd(gy[0], gy[1]));
  }
  {
    // d/d sharpening_filter.weight[i][j] += gY'(p) * Y0(p + (i-1, j-1)), row t - 1: gY' = HP(t-1) = slot K+2;
    // Y rows t-2, t-1, t = slots K+1, K+2, K
    const float* hc = st.hp[(K + 2) % 3];
    r2l_p2 gp[2];
    gp[0] = r2l_mk2(ok_tm1 ? hc[1] : 0.f, ok_tm1 ? hc[2] : 0.f);
    gp[1] = r2l_mk2(ok_tm1 ? hc[3] : 0.f, ok_tm1 ? hc[4] : 0.f);
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 3; ++i) {
      r2l_p2 x[3][2];
      r2l_row_pairs(st.y[(K + 1 + i) % 3], x);
      R2L_PRAGMA_UNROLL
      for (int j = 0; j < 3; ++j) {
        r2l_p2 sacc = A.gsh[i * 3 + j];
        R2L_PRAGMA_UNROLL
        for (int p = 0; p < 2; ++p) sacc = r2l_pfma(gp[p], x[j][p], sacc);
        A.gsh[i * 3 + j] = sacc;
      }
    }
  }
}
// value of B2's global slot i (layout R2L_B2_*) held by a lane: E = even-row bank, O = odd-row bank (gay[9] pairs, sy)
R2L_HD float r2l_b2s_slot(const R2LSumAcc& A, const float* E, const float* O, int i) {
  if (i < R2L_B2_GAY) return A.gsh[i][0] + A.gsh[i][1];
  if (i < R2L_B2_SY) {
    const int k = i - R2L_B2_GAY, par = k / 9, t = k % 9;
    return ((par >> 1) ? O : E)[t * 2 + (par & 1)];
  }
  const int par = i - R2L_B2_SY;
  return ((par >> 1) ? O : E)[18 + (par & 1)];
}
#ifndef R2L_B2S_PF
#define R2L_B2S_PF 2
#endif
// Helper workgroups of the sums pass (bid >= a.nmain): B1's partials -- complete before this launch starts -- are added
// while the main workgroups walk their items, so that the launch's tail reduces B2's 49 slots, not 155.  Helper h:
// slots 8 h ... 8 h + 7, 32 lanes per slot (lane j: workgroups j, j + 32, ... in order; the 32 lane sums in lane order).
#define R2L_B2S_HELPERS ((R2L_B1_NACC + 7) / 8)
struct R2LB1Totals {  // (r2l_tree_level2's `also`: the last workgroup picks up the helpers' totals with its own loads)
  const double* tot;
  double* sums;
  R2L_MEMBER double fetch(int tid) const { return tid < R2L_B1_NACC ? r2l_load_coherent(tot + tid) : 0.0; }
  R2L_MEMBER void put(int tid, double v) const {
    if (tid < R2L_B1_NACC) sums[tid] = v;
  }
};
R2L_BLOCKFN void r2l_b2s_helper(const R2LBwd2Args& a, int h, float* lds, int tid) {
  static_assert(R2L_B2S_NT == 256, "8 slots x 32 lanes");
  double* hd = (double*)(lds + 2048);
  const int sl = h * 8 + (tid >> 5), j = tid & 31;
  const int n1 = a.b1_n;
  const float* row = a.b1_partial + (size_t)(sl < R2L_B1_NACC ? sl : R2L_B1_NACC - 1) * n1;
  double acc = 0.0;
  for (int m0 = j; m0 < n1; m0 += 32 * 16) {
    float v[16];
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 16; ++k) v[k] = row[(m0 + 32 * k < n1) ? m0 + 32 * k : n1 - 1];
    R2L_PRAGMA_UNROLL
    for (int k = 0; k < 16; ++k) acc += (m0 + 32 * k < n1) ? (double)v[k] : 0.0;
  }
  hd[tid] = acc;
  R2L_LDS_BARRIER();
  if (tid < 8 && h * 8 + tid < R2L_B1_NACC) {
    double t = 0.0;
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < 32; ++i) t += hd[tid * 32 + i];
    r2l_store_coherent(a.b1_tot + h * 8 + tid, t);
  }
  R2L_STORES_DONE();
  R2L_LDS_BARRIER();
}
template <bool U16>
R2L_BLOCKFN void r2l_bwd2_sums_block(const R2LBwd2Args& a, int bid, int nblk_launch, float* lds) {
  constexpr int NWV = R2L_B2S_NWV, NT = R2L_B2S_NT;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nblk = a.nmain > 0 ? a.nmain : nblk_launch;  // the workgroups that walk the work items
  if (bid >= nblk) {
    if (!a.tree.counters) return;
    double* sums = (double*)(lds + 4);
    double* tg = sums + R2L_NSUMS;
    float* pl = (float*)(tg + R2L_UNFOLD_TG);
    if (tid < R2L_P_COUNT) pl[tid] = a.params[tid];
    r2l_b2s_helper(a, bid - nblk, lds, tid);  // (its barriers also publish pl)
    r2l_unfold_tables_lane<NT>(tid, tg, pl);  // in case this workgroup turns out to be the launch's last
    R2L_LDS_BARRIER();
    if (!r2l_tree_level2<R2L_B2_NACC, NT>(a.tree, nblk, R2L_B2S_HELPERS, lds, sums + R2L_B1_NACC, (double*)(lds + 1024),
                                          R2LB1Totals{a.b1_tot, sums}))
      return;
    r2l_unfold_from_lds<NT, true>(sums, tg, pl, a.grad_params);
    return;
  }
#ifdef R2L_EXP_STAMPS
  const unsigned long long tl0_ = __builtin_amdgcn_s_memrealtime();
#endif
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  float* bank = lds + (size_t)wave * 64 * R2L_B2S_BANK + lane * 4;  // [chunk][lane][4]
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 5; ++c) {
    r2l_f4 z;
    z.x = z.y = z.z = z.w = 0.f;
    *(r2l_f4*)(bank + c * 64 * 4) = z;
  }
  R2LSumAcc A;
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 9; ++i) A.gsh[i] = A.gay[i] = r2l_splat2(0.f);
  A.sy = r2l_splat2(0.f);
  R2LFwdStreamArgs sa;  // for the streaming kernels' raw-row fetch / convert
  sa.raw = a.raw;
  sa.W = a.W;
  sa.H = a.H;
  const int nstrip = (a.W + 255) >> 8;
  const unsigned plane = (unsigned)a.H * (unsigned)a.W;
  const int band_h = a.band_h, nband = (a.H + band_h - 1) / band_h, nitems = a.B * nband * nstrip;
  constexpr int PF = R2L_B2S_PF;
  static_assert(6 % PF == 0, "the prefetch ring is indexed by the unroll position");
  // the registers hold the bank of EVEN rows between items (bands start on even rows, at K = 0)
  R2L_PRAGMA_NOUNROLL
  for (int item = r2l_xcd_window(bid, nblk, a.xcdm) * NWV + wave; item < nitems; item += nblk * NWV) {
    const int strip = item % nstrip, ib = item / nstrip;
    const int band = ib % nband, b = ib / nband;
    const int xs = strip * 256 + 4 * lane;
    const bool in_w = xs < a.W;
    const int x0 = in_w ? xs : a.W - 4;
    const bool le = x0 == 0, re = x0 + 4 >= a.W;
    const int y0 = band * band_h;  // a multiple of 6
    const int y1 = (y0 + band_h < a.H) ? y0 + band_h : a.H;
    const size_t img = (size_t)b * plane;
    const float* hpimg = a.hp + img;
    R2LSumState st;
    R2L_PRAGMA_UNROLL
    for (int j = 0; j < 6; ++j) st.y[1][j] = 0.f;  // Y(y0 - 2): only ever meets a zero cotangent, but must be finite
    R2LFlStage pf[PF];    // raw row t + 1
    R2LSumStage pfh[PF];  // HP row t + 1
    // step t: V(t+1) and HP(t+1) arrive; Y(t); GAY / SY of row t; sharpen-weight sums of row t - 1.  Warm-up: t = y0-3
    // (V(y0-2)), y0-2 (V(y0-1), HP(y0-1)), y0-1 (V(y0), HP(y0), Y(y0-1)) = K 3 .. 5; one more step behind the band's last
    // row finishes the sharpen-weight sums of that row.
    R2L_PRAGMA_UNROLL
    for (int i = 0; i < PF; ++i) {
      r2l_fl_fetch<U16, R2L_B2S_RAW_NT>(sa, img, r2l_mirror(R2L_NH(y0 - 2 + i), a.H), x0, le, re, lane, pf[(3 + i) % PF]);
      r2l_b2s_fetch_hp(hpimg, R2L_NH(y0 - 2 + i), a.H, a.W, x0, le, re, lane, pfh[(3 + i) % PF]);
    }
#define R2L_B2S_LOAD_STEP(K, t)                                                                               \
  r2l_fl_convert<U16>(sa, F, pf[(K) % PF], le, re, st.v[((K) + 1) % 3], st.xp[((K) + 1) % 3]);                \
  r2l_b2s_build_hp(pfh[(K) % PF], (unsigned)((t) + 1) < (unsigned)a.H, le, re, st.hp[((K) + 1) % 3]);         \
  r2l_fl_fetch<U16, R2L_B2S_RAW_NT>(sa, img, r2l_mirror(R2L_NH((t) + 1 + PF), a.H), x0, le, re, lane, pf[(K) % PF]);                  \
  r2l_b2s_fetch_hp(hpimg, R2L_NH((t) + 1 + PF), a.H, a.W, x0, le, re, lane, pfh[(K) % PF]);
    R2L_B2S_LOAD_STEP(3, y0 - 3)
    R2L_B2S_LOAD_STEP(4, y0 - 2)
    R2L_B2S_LOAD_STEP(5, y0 - 1)
    if (r2l_opaque_true()) r2l_b2s_luma<5>(a, st, y0 - 1, le, re);
    for (int qb = y0; qb < y1; qb += 6) {  // (a last group that runs past y1 also takes the step t = y1, see below)
      R2L_PROGRESS_PRIO(qb - y0, y1 - y0);
#define R2L_B2S_STEP(K)                                                                                       \
  {                                                                                                           \
    const int t = qb + K;                                                                                     \
    R2L_PROGRESS_PRIO_STEP(t - y0, y1 - y0);                                                                  \
    R2L_B2S_LOAD_STEP(K, t)                                                                                   \
    if (K) r2l_b2s_swap(A, bank); /* the bank of this row's parity into the registers */                      \
    if (r2l_opaque_true()) {                                                                                  \
      r2l_b2s_luma<K>(a, st, t, le, re);                                                                      \
      r2l_b2s_sums<K>(a, st, A, in_w && t < y1, in_w && t - 1 >= y0 && t - 1 < y1);                           \
    }                                                                                                         \
  }
      R2L_B2S_STEP(0)
      R2L_B2S_STEP(1)
      R2L_B2S_STEP(2)
      R2L_B2S_STEP(3)
      R2L_B2S_STEP(4)
      R2L_B2S_STEP(5)
      r2l_b2s_swap(A, bank);  // (K = 5 was an odd row)
#undef R2L_B2S_STEP
    }
    // the sharpen-weight sums of the band's last row want Y(y1): one more step, t = y1, unless the last group ran past it
    if ((y1 - y0) % 6 == 0) {
      r2l_fl_convert<U16>(sa, F, pf[0], le, re, st.v[1], st.xp[1]);  // V(y1 + 1): K = 0
      if (r2l_opaque_true()) {
        r2l_b2s_luma<0>(a, st, y1, le, re);
        r2l_b2s_sums<0, false>(a, st, A, false, in_w);
      }
    }
#undef R2L_B2S_LOAD_STEP
  }
  // ---- lanes -> one partial per slot and workgroup; then the tree over B1's and B2's partials and the unfold --------------
  float E[R2L_B2S_BANK], O[R2L_B2S_BANK];
  R2L_PRAGMA_UNROLL
  for (int i = 0; i < 9; ++i) {
    E[2 * i] = A.gay[i][0];
    E[2 * i + 1] = A.gay[i][1];
  }
  E[18] = A.sy[0];
  E[19] = A.sy[1];
  R2L_PRAGMA_UNROLL
  for (int c = 0; c < 5; ++c) {
    const r2l_f4 o4 = r2l_lds_f4(bank + c * 64 * 4);
    O[4 * c] = o4.x;
    O[4 * c + 1] = o4.y;
    O[4 * c + 2] = o4.z;
    O[4 * c + 3] = o4.w;
  }
#ifdef R2L_EXP_STAMPS
  if (a.debug && bid < 2048 && tid == 0) {  // (start, end of the item loop) of every workgroup: tests/timeline_fwd.py
    ((unsigned long long*)a.debug)[2 * bid] = tl0_;
    ((unsigned long long*)a.debug)[2 * bid + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  R2L_TAILST(20);
  // the packed parameters of the unfold, asked for now: they arrive behind the partials' store wait, not as a round trip of
  // their own in the launch's last workgroup
  static_assert(R2L_P_COUNT <= NT, "one parameter per lane");
  const float pv = (a.tree.counters && tid < R2L_P_COUNT) ? a.params[tid] : 0.f;
  R2L_LDS_BARRIER();  // the bank areas become reduction scratch
  r2l_bp_block_reduce<R2L_B2_NACC, NT, R2L_B2_NACC>(lds, tid, a.partial, 0, bid, nblk,
                                                    [&](int i) { return r2l_b2s_slot(A, E, O, i); });
  R2L_TAILST(21);
  if (a.tree.counters) {
    double* sums = (double*)(lds + 4);
    double* tg = sums + R2L_NSUMS;
    float* pl = (float*)(tg + R2L_UNFOLD_TG);
    // the packed parameters go to LDS, and what the unfold derives from them alone is computed while the first ticket
    // travels (by every workgroup: the vector unit has nothing else to do then; only the last one uses it)
    if (tid < R2L_P_COUNT) pl[tid] = pv;
    R2L_LDS_BARRIER();
    if (a.nmain > 0) {  // B2's slots here, B1's by the helper workgroups
      if (!r2l_tree_level1<R2L_B2_NACC, NT>(a.tree, bid, nblk, lds, [&](int t) { r2l_unfold_tables_lane<NT>(t, tg, pl); })) return;
      if (!r2l_tree_level2<R2L_B2_NACC, NT>(a.tree, nblk, R2L_B2S_HELPERS, lds, sums + R2L_B1_NACC, (double*)(lds + 1024),
                                            R2LB1Totals{a.b1_tot, sums}))
        return;
    } else if (!r2l_tree_finish<R2L_NSUMS, NT>(a.tree, bid, nblk, lds, sums, (double*)(lds + 1024),
                                               (R2L_B2S_LDS_FLOATS - 1024) / 2,
                                               [&](int t) { r2l_unfold_tables_lane<NT>(t, tg, pl); }))
      return;
#ifdef R2L_TEST_HOOKS
    if (a.debug && tid < R2L_NSUMS) ((double*)a.debug)[tid] = sums[tid];  // (tests: the 155 totals, slot by slot)
#endif
    r2l_unfold_from_lds<NT, true>(sums, tg, pl, a.grad_params);
    R2L_TAILST(27);
  }
}

#endif  // !R2L_SERIAL
