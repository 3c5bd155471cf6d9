// Streaming-pattern probe for the static short chain (C3): 4 B read : 12 B written per pixel into the real
// (B,3,H,W) layout.  Which access shape / cache policy reaches the highest rate?  No arithmetic.
//   lin  : grid-stride over float4 items (consecutive lanes -> consecutive 16 B)
//   row  : the kernel's shape -- a wavefront owns 256 columns x a band of rows, 4 waves per workgroup = one
//          4-KiB row of a 1024-wide frame; rows walked top to bottom, PF rows of loads in flight
// policies: plain / nontemporal stores / nontemporal loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NTL>
__device__ __forceinline__ f4 ld(const f4* p) {
  return NTL ? __builtin_nontemporal_load(p) : *p;
}
template <bool NTS>
__device__ __forceinline__ void st(f4* p, f4 v) {
  if (NTS) __builtin_nontemporal_store(v, p); else *p = v;
}

template <bool NTL, bool NTS, int U>
__global__ void lin(const f4* __restrict__ a, f4* __restrict__ o, size_t hw4, size_t n4) {
  const size_t T = (size_t)gridDim.x * blockDim.x;
  for (size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i0 < n4; i0 += T * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = i0 + u * T; if (i < n4) v[u] = ld<NTL>(a + i); }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * T;
      if (i < n4) {
        const size_t img = i / hw4, p = i - img * hw4;
        f4* d = o + img * 3 * hw4 + p;
        f4 x = v[u];
        st<NTS>(d, x); x.x += 1.f; st<NTS>(d + hw4, x); x.y += 1.f; st<NTS>(d + 2 * hw4, x);
      }
    }
  }
}

// WPB waves per workgroup; item = (image, band, 256-col strip); items dealt so that a workgroup's waves take
// adjacent strips.  W4 = W/4 float4 per row.
template <bool NTL, bool NTS, int PF, int MODE>  // MODE 0: r1w3, 1: write only, 2: read only
__global__ void row(const f4* __restrict__ a, f4* __restrict__ o, int B, int H, int W4, int nband, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int nseg = W4 / 64, band_h = H / nband;
  const long nitems = (long)B * nseg * nband;
  const size_t hw4 = (size_t)H * W4;
  float acc = 0.f;
  for (long item = (long)blockIdx.x * wpb + wave; item < nitems; item += (long)gridDim.x * wpb) {
    const int seg = item % nseg; const long r = item / nseg;
    const int band = r % nband, b = r / nband;
    const f4* src = a + (size_t)b * hw4 + (size_t)band * band_h * W4 + seg * 64 + lane;
    f4* dst = o + (size_t)b * 3 * hw4 + (size_t)band * band_h * W4 + seg * 64 + lane;
    f4 v[PF];
    if (MODE != 1) {
#pragma unroll
      for (int k = 0; k < PF; ++k) v[k] = ld<NTL>(src + (size_t)k * W4);
    } else {
#pragma unroll
      for (int k = 0; k < PF; ++k) v[k] = (f4){1.f, 2.f, 3.f, (float)k};
    }
    for (int y = 0; y < band_h; y += PF) {
#pragma unroll
      for (int k = 0; k < PF; ++k) {
        f4 x = v[k];
        if (MODE != 1 && y + k + PF < band_h) v[k] = ld<NTL>(src + (size_t)(y + k + PF) * W4);
        if (MODE != 2) {
          f4* d = dst + (size_t)(y + k) * W4;
          st<NTS>(d, x); x.x += 1.f; st<NTS>(d + hw4, x); x.y += 1.f; st<NTS>(d + 2 * hw4, x);
        } else {
          acc += x.x + x.y + x.z + x.w;
        }
      }
    }
  }
  if (MODE == 2 && acc == 12345.678f) sink[0] = acc;
}

static hipEvent_t e0, e1;
template <class F>
static float timeit(F f) {
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const int B = 256, H = 1024, W = 1024, W4 = W / 4;
  const size_t hw4 = (size_t)H * W4, n4 = (size_t)B * hw4;
  f4 *a, *o; float* sink;
  hipMalloc(&a, n4 * 16); hipMalloc(&o, 3 * n4 * 16); hipMalloc(&sink, 4);
  hipMemset(a, 0, n4 * 16);
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double gb = 4.0 * n4 * 16 / 1e9;
#define LIN(NTL, NTS, U, G, BS) { float ms = timeit([&] { lin<NTL, NTS, U><<<G, BS>>>(a, o, hw4, n4); }); \
  printf("lin  ntl=%d nts=%d U=%d grid=%6d bs=%4d  %8.1f us %7.1f GB/s\n", NTL, NTS, U, G, BS, ms * 1e3, gb / ms * 1e3); }
  for (int g : {2048, 8192, 32768}) {
    LIN(false, false, 1, g, 256) LIN(false, true, 1, g, 256) LIN(true, true, 1, g, 256) LIN(true, false, 1, g, 256)
    LIN(false, false, 4, g, 256) LIN(false, true, 4, g, 256) LIN(true, true, 4, g, 256)
  }
  LIN(false, true, 2, 4096, 512) LIN(false, true, 2, 2048, 1024) LIN(true, true, 2, 4096, 512)
#define ROW(NTL, NTS, PF, MODE, NB, G, BS) { float ms = timeit([&] { row<NTL, NTS, PF, MODE><<<G, BS>>>(a, o, B, H, W4, NB, sink); }); \
  const double bytes = (MODE == 0 ? 4.0 : MODE == 1 ? 3.0 : 1.0) * n4 * 16 / 1e9; \
  printf("row  ntl=%d nts=%d PF=%d mode=%d bands=%3d grid=%6d bs=%4d  %8.1f us %7.1f GB/s\n", NTL, NTS, PF, MODE, NB, G, BS, ms * 1e3, bytes / ms * 1e3); }
  for (int nb : {8, 16, 32, 64}) {
    const int items = B * 4 * nb;
    ROW(false, false, 2, 0, nb, items / 4, 256) ROW(false, true, 2, 0, nb, items / 4, 256) ROW(true, true, 2, 0, nb, items / 4, 256)
    ROW(false, true, 4, 0, nb, items / 4, 256) ROW(true, true, 4, 0, nb, items / 4, 256)
  }
  // persistent grids (items dealt round-robin), and mode 1 / 2 ceilings of the row shape
  for (int g : {512, 1024, 2048, 4096}) { ROW(false, true, 2, 0, 32, g, 256) ROW(true, true, 4, 0, 64, g, 256) }
  ROW(false, false, 2, 1, 16, 4096, 256) ROW(false, true, 2, 1, 16, 4096, 256) ROW(false, false, 2, 2, 16, 4096, 256) ROW(true, false, 4, 2, 16, 4096, 256)
  return 0;
}
