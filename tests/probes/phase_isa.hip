// phase_isa.hip -- every hot PHASE of the tile kernels as its own tiny kernel (interior-tile instantiations only), so that
// `hipcc -S` + tests/isa_stats.py count the instructions of one phase in isolation:
//   hipcc -O3 -std=c++17 -fno-slp-vectorize --offload-arch=gfx950 --offload-device-only -S tests/probes/phase_isa.hip -o /tmp/phase.s
//   python tests/isa_stats.py /tmp/phase.s
// Diagnostic only: nothing here is launched.
#include "../../raw2logit_amd/csrc/r2l_param_kernels.h"

typedef R2LGeom<64, 64> G;
#define LDSF (2 * G::PAD + 3 * G::PLANE)

#define PROBE(name, VGPRS_OCC, ...)                                                        \
  __global__ __launch_bounds__(R2L_NT, VGPRS_OCC) void name(R2LBwd2Args a2, R2LBwd1Args a1, int sel) { \
    __shared__ __attribute__((aligned(16))) float lds[LDSF];                                 \
    float* V = lds + G::PAD;                                                                 \
    float* P1 = V + G::PLANE;                                                                \
    float* P2 = P1 + G::PLANE;                                                               \
    const int tid = threadIdx.x;                                                             \
    (void)V; (void)P1; (void)P2; (void)tid; (void)sel;                                       \
    __VA_ARGS__                                                                              \
  }

PROBE(p_b2_adjoint_blur, 4, {
  R2LFoldedRef F = R2L_FOLDED_REF(a2.F);
  r2l_adjoint_blur<G>(tid, P1, P2, F);
})
PROBE(p_compute_y_interior, 4, {
  R2LFoldedRef F = R2L_FOLDED_REF(a2.F);
  r2l_compute_y<G, false>(tid, V, P1, F, 0, 0, a2.H, a2.W);
})
PROBE(p_compute_yp, 4, {
  R2LFoldedRef F = R2L_FOLDED_REF(a2.F);
  r2l_compute_yp<G>(tid, P1, P2, F);
})
PROBE(p_b2_pixels_interior, 4, {
  R2LBwd2Regs regs;
  for (int i = 0; i < R2L_L2_NACC; ++i) regs.acc[i] = a2.partial[i];
  int tx_, row_;
  G::thread_tile(tid, tx_, row_, regs.py);
  R2LTile t;
  t.b = 0;
  t.oy = sel * 64;
  t.ox = 64;
  t.border = false;
  t.ragged = false;
  r2l_bwd2_pixels<G, false>(tid, V, P1, P2, a2, t, regs);
  for (int i = 0; i < R2L_L2_NACC; ++i) a2.partial[i * 512 + tid] = regs.acc[i];
})
PROBE(p_b2_fetch_store, 4, {
  R2LFoldedRef F = R2L_FOLDED_REF(a2.F);
  R2LPrefetch<G> pv;
  R2LPrefetch<G> pg;
  R2LTile t;
  t.b = 0;
  t.oy = sel * 64;
  t.ox = 64;
  t.border = false;
  t.ragged = false;
  r2l_fetch_raw_tile<G, false>(tid, a2.raw, t, a2.H, a2.W, pv);
  r2l_fetch_tile<G, 1>(tid, a2.gypp, t, a2.H, a2.W, pg);
  r2l_store_v<G, false>(tid, V, F, pv, a2.raw);
  r2l_store_plane_s2<G>(tid, P1, pg);
})
PROBE(p_b1_rows2, 2, {
  R2LBwd1Regs regs;
  for (int i = 0; i < R2L_L1_NACC; ++i) regs.acc[i] = r2l_splat2(a1.partial[i]);
  int tx, row0, py;
  G::thread_tile(tid, tx, row0, py);
  regs.py = py;
  R2LGoutPre gp;
  R2LTile t;
  t.b = 0;
  t.oy = sel * 64;
  t.ox = 64;
  t.border = false;
  t.ragged = false;
  r2l_bwd1_fetch_gout<G>(tid, a1, t, gp, 0);
  r2l_bwd1_fetch_gout<G>(tid, a1, t, gp, 1);
  r2l_bwd1_pixels<G, false, false, true>(tid, V, P2, a1, t, gp, regs, [](int) {});
  for (int i = 0; i < R2L_L1_NACC; ++i) a1.partial[i * 512 + tid] = regs.acc[i][0] + regs.acc[i][1];
})
