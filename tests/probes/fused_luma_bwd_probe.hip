// fused_luma_bwd_probe.hip -- OCCUPANCY / TIMING PROBE, NOT PRODUCT CODE (round 6, review item 3).
//
// "bwd1_blur_hp + bwd2_sums as ONE launch, HP through registers instead of a plane": this file glues the two row loops of
// r2l_param_plane_bwd.h together the cheapest possible way -- one wavefront per (image, band, strip) walks the band once; the row
// step computes HP(q) with r2l_hb_row (blur adjoint + the 25 blur-weight sums, exactly as r2l_hb_step does) and hands it to the
// sums pass's step for row t = q - 1 (r2l_b2s_luma / r2l_b2s_sums, unchanged) in registers.  What it leaves out makes it
// FASTER than a correct kernel could be: the strip-edge HP columns are zero instead of recomputed (2 more HP columns per lane
// and row), the row-mirror additions of the image's first / last rows are skipped, nothing is reduced at the end (the sums go
// to a dummy store).  So its register count is a LOWER bound of the fused kernel's and its time a lower bound of the fused
// kernel's time: if THIS does not beat 68 us for 64x512x512, nothing built on it will.
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 -I raw2logit_amd/csrc -I include \
//         tests/probes/fused_luma_bwd_probe.hip -o tests/probes/fused_luma_bwd_probe -Rpass-analysis=kernel-resource-usage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>
static int r2l_fail(int c, const std::string&) { return c; }
#include "r2l_param_plane_bwd.h"

struct ProbeArgs {
  R2LBwd1Args b1;   // F, yp (Y'), gypp (dL/dY''), B, H, W
  R2LRaw raw;
  float* sink;
  int band_h;
};

#ifndef PROBE_OCC
#define PROBE_OCC 2
#endif
#ifndef PROBE_BLUR_IN_LDS
#define PROBE_BLUR_IN_LDS 0   // 1: the 25 blur-weight pair sums live in a wave-private LDS area (read-modify-write per window row)
#endif

template <int K>
__device__ __forceinline__ void probe_hp_row(const ProbeArgs& pa, const float gw[6][8], r2l_p2 blur[25], const r2l_f4& yrow,
                                             bool le, bool re, bool ok, float hp4[4]) {
  const R2LBwd1Args& a = pa.b1;
  R2LFoldedRef F = R2L_FOLDED_REF(r2l_opaque_after(a.F, gw[(K + 2) % 6][2]));
  R2LHpAcc A;
  A.h[0] = A.h[1] = A.el = A.er = r2l_splat2(0.f);
  R2LBsY y;
  const float y0 = ok ? yrow.x : 0.f, y1 = ok ? yrow.y : 0.f, y2 = ok ? yrow.z : 0.f, y3 = ok ? yrow.w : 0.f;
  y.y01 = r2l_mk2(y0, y1);
  y.y23 = r2l_mk2(y2, y3);
  y.yl1 = le ? y1 : 0.f;
  y.yl2 = le ? y2 : 0.f;
  y.yr2 = re ? y2 : 0.f;
  y.yr1 = re ? y1 : 0.f;
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    r2l_hb_row(A, blur, y, gw[(K + 4 + r) % 6], F.blur, 4 - r);
    __builtin_amdgcn_sched_barrier(0);
  }
  hp4[0] = A.h[0][0];
  hp4[1] = A.h[0][1] + (le ? A.el[1] : 0.f) + (re ? A.er[1] : 0.f);
  hp4[2] = A.h[1][0] + (le ? A.el[0] : 0.f) + (re ? A.er[0] : 0.f);
  hp4[3] = A.h[1][1];
}

__global__ __launch_bounds__(256, PROBE_OCC) void fused_luma_bwd_probe(const ProbeArgs pa) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 64 * R2L_B2S_BANK + 16];
  const R2LBwd1Args& a = pa.b1;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  R2LFoldedRef F = R2L_FOLDED_REF(a.F);
  float* bank = lds + (size_t)wave * 64 * R2L_B2S_BANK + lane * 4;
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    r2l_f4 z;
    z.x = z.y = z.z = z.w = 0.f;
    *(r2l_f4*)(bank + c * 64 * 4) = z;
  }
  r2l_p2 blur[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) blur[i] = r2l_splat2(0.f);
  R2LSumAcc A;
#pragma unroll
  for (int i = 0; i < 9; ++i) A.gsh[i] = A.gay[i] = r2l_splat2(0.f);
  A.sy = r2l_splat2(0.f);
  R2LBwd2Args a2;   // what the sums pass's step functions read
  a2.F = a.F;
  a2.H = a.H;
  a2.W = a.W;
  R2LFwdStreamArgs sa;
  sa.raw = pa.raw;
  sa.W = a.W;
  sa.H = a.H;
  const int nstrip = (a.W + 255) >> 8;
  const unsigned plane = (unsigned)a.H * (unsigned)a.W;
  const int band_h = pa.band_h, nband = (a.H + band_h - 1) / band_h, nitems = a.B * nband * nstrip;
  constexpr int PF = 2;
#pragma unroll 1
  for (int item = blockIdx.x * 4 + wave; item < nitems; item += gridDim.x * 4) {
    const int strip = item % nstrip, ib = item / nstrip;
    const int band = ib % nband, b = ib / nband;
    const int xs = strip * 256 + 4 * lane;
    const bool in_w = xs < a.W;
    const int x0 = in_w ? xs : a.W - 4;
    const bool le = x0 == 0, re = x0 + 4 >= a.W;
    const int y0 = band * band_h;
    const int y1 = (y0 + band_h < a.H) ? y0 + band_h : a.H;
    const size_t img = (size_t)b * plane;
    const float* ypimg = a.yp + img;
    const float* gimg = a.gypp + img;
    float gw[6][8];
    R2LFaStage pfg[PF];
    r2l_f4 pfy[PF];
    R2LFlStage pfr[PF];
    R2LSumState st;
#pragma unroll
    for (int j = 0; j < 6; ++j) st.y[1][j] = st.y[0][j] = st.y[2][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) st.hp[i][j] = st.v[i][j] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) st.xp[i][j] = r2l_splat2(0.f);
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      r2l_fa_fetch(gimg, R2L_NH(y0 - 2 + i), a.H, a.W, x0, le, re, lane, pfg[(2 + i) % PF]);
      const int yc = (y0 + i < a.H) ? y0 + i : a.H - 1;
      pfy[i % PF] = r2l_stream_load_f4(ypimg + (size_t)yc * a.W + x0);
      r2l_fl_fetch<false, false>(sa, img, r2l_mirror(R2L_NH(y0 - 1 + i), a.H), x0, le, re, lane, pfr[(3 + i) % PF]);
    }
#define PROBE_LOAD(K, q)                                                                                 \
  r2l_hp_build(pfg[(K) % PF], (unsigned)((q) + 2) < (unsigned)a.H, le, re, gw[((K) + 2) % 6]);           \
  r2l_fa_fetch(gimg, R2L_NH((q) + 2 + PF), a.H, a.W, x0, le, re, lane, pfg[(K) % PF]);
    PROBE_LOAD(2, y0 - 4)
    PROBE_LOAD(3, y0 - 3)
    PROBE_LOAD(4, y0 - 2)
    PROBE_LOAD(5, y0 - 1)
    for (int qb = y0; qb < y1; qb += 6) {
#define PROBE_STEP(K)                                                                                     \
  {                                                                                                       \
    const int q = qb + K;                                                                                 \
    PROBE_LOAD(K, q)                                                                                      \
    const r2l_f4 y_ = pfy[(K) % PF];                                                                      \
    {                                                                                                     \
      const int yc = (q + PF < a.H) ? q + PF : a.H - 1;                                                   \
      pfy[(K) % PF] = r2l_stream_load_f4(ypimg + (size_t)yc * a.W + x0);                                  \
    }                                                                                                     \
    float hp4[4];                                                                                         \
    probe_hp_row<K>(pa, gw, blur, y_, le, re, in_w && q < y1, hp4);                                       \
    /* HP(q) -> the sums pass's window (strip-edge columns: zero, see the header), V(q) from the raw row */ \
    {                                                                                                     \
      float* h = st.hp[(K) % 3];                                                                          \
      h[0] = le ? 0.f : r2l_wshr(hp4[3], 0.f);                                                            \
      h[1] = hp4[0];                                                                                      \
      h[2] = hp4[1];                                                                                      \
      h[3] = hp4[2];                                                                                      \
      h[4] = hp4[3];                                                                                      \
      h[5] = re ? 0.f : r2l_wshl(hp4[0], 0.f);                                                            \
    }                                                                                                     \
    r2l_fl_convert<false>(sa, F, pfr[(K) % PF], le, re, st.v[(K) % 3], st.xp[(K) % 3]);                   \
    r2l_fl_fetch<false, false>(sa, img, r2l_mirror(R2L_NH(q + PF), a.H), x0, le, re, lane, pfr[(K) % PF]); \
    constexpr int KT = ((K) + 5) % 6; /* the sums pass's step for row t = q - 1 */                        \
    if (KT) r2l_b2s_swap(A, bank);                                                                        \
    if (r2l_opaque_true()) {                                                                              \
      r2l_b2s_luma<KT>(a2, st, q - 1, le, re);                                                            \
      r2l_b2s_sums<KT>(a2, st, A, in_w && q - 1 < y1 && q - 1 >= y0, in_w && q - 2 >= y0 && q - 2 < y1);  \
    }                                                                                                     \
    if (KT == 5) r2l_b2s_swap(A, bank);                                                                   \
  }
      PROBE_STEP(0)
      PROBE_STEP(1)
      PROBE_STEP(2)
      PROBE_STEP(3)
      PROBE_STEP(4)
      PROBE_STEP(5)
    }
  }
  // dummy sink: keep every accumulator alive
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 25; ++i) s += blur[i][0] + blur[i][1];
#pragma unroll
  for (int i = 0; i < 9; ++i) s += A.gsh[i][0] + A.gsh[i][1] + A.gay[i][0] + A.gay[i][1];
  s += A.sy[0] + A.sy[1];
#pragma unroll
  for (int c = 0; c < 5; ++c) s += bank[c * 64 * 4];
  if (s == 123.456f) pa.sink[tid] = s;
}

int main(int argc, char** argv) {
  const int B = 64, H = 512, W = 512;
  const size_t px = (size_t)B * H * W;
  float *raw, *yp, *g, *sink;
  R2LFolded* F;
  hipMalloc(&raw, px * 4);
  hipMalloc(&yp, px * 4);
  hipMalloc(&g, px * 4);
  hipMalloc(&sink, 4096);
  hipMalloc(&F, sizeof(R2LFolded));
  hipMemset(raw, 0, px * 4);
  hipMemset(yp, 0, px * 4);
  hipMemset(g, 0, px * 4);
  std::vector<float> hf(sizeof(R2LFolded) / 4, 0.01f);
  hipMemcpy(F, hf.data(), sizeof(R2LFolded), hipMemcpyHostToDevice);
  ProbeArgs pa;
  pa.b1.F = F;
  pa.b1.yp = yp;
  pa.b1.gypp = g;
  pa.b1.B = B;
  pa.b1.H = H;
  pa.b1.W = W;
  pa.raw = r2l_raw_f32(raw);
  pa.sink = sink;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int band_h : {24, 36, 48, 72}) {
    pa.band_h = band_h;
    const int nitems = B * ((H + band_h - 1) / band_h) * ((W + 255) / 256);
    for (int grid : {512, 768, 1024}) {
      const int g_ = std::min(grid, (nitems + 3) / 4);
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fused_luma_bwd_probe, dim3(g_), dim3(256), 0, 0, pa);
      hipDeviceSynchronize();
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fused_luma_bwd_probe, dim3(g_), dim3(256), 0, 0, pa);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms / 20 * 1e3f);
      }
      printf("fused luma backward probe (lower bound), 64x512x512, band %2d rows, %4d workgroups: %.1f us per launch\n", band_h, g_, best);
    }
  }
  return 0;
}
