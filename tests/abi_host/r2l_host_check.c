/* r2l_host_check.c -- a consumer of include/r2l_isp.h in plain C: no Python, no torch, no C++.
 *
 * TEST INFRASTRUCTURE (tests/test_gpu_abi_host.py builds and runs it; __graft_entry__.build() compiles it too, so that the header
 * is known to be valid C and the library to link from C).  What it shows is SURVEY.md section 8b's boundary: everything a host
 * needs is device pointers, sizes and a stream -- the two calls of one training step of ParametrizedProcessing
 * (processing/pipeline_torch.py:175-225 forward, autograd's backward) and the static pipeline (processing/pipeline_numpy.py:70-141)
 * are driven here with hipMalloc / hipMemcpy only, on golden cases the reference itself generated (tests/golden/, exported to a flat
 * binary file by the test), and judged against the reference's outputs.
 *
 *   r2l_host_check <case.bin>        exit 0: every comparison inside its limit; 1: a comparison failed; 2: a call failed
 *
 * case file (little endian):
 *   int32  magic 0x52324c31, kind (0 = parametrized step, 1 = static chain), B, H, W, a, b, c
 *          kind 0: a = bn_mode (R2L_BN_*), b, c unused      kind 1: a = debayer, b = sharpening, c = denoising
 *   kind 0: float32 raw[B*H*W], packed[R2L_P_COUNT], cot[B*3*H*W], out[B*3*H*W], grad[R2L_P_NTRAIN], running_mean[3], running_var[3],
 *           out_limit[B*3*H*W], grad_limit[R2L_P_NTRAIN]   (the limits of tests/parity_checks.py: check_param_case, per element)
 *   kind 1: float32 raw[B*H*W]; float64 camera[16], gamma; float32 out[B*3*H*W]; float64 out_tol
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "r2l_isp.h"

#define HIP_OK(x)                                                                           \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) {                                                                 \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));  \
      exit(2);                                                                              \
    }                                                                                       \
  } while (0)
#define R2L_OK(x)                                                                           \
  do {                                                                                      \
    int e_ = (x);                                                                           \
    if (e_ != 0) {                                                                          \
      fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, e_, r2l_last_error()); \
      exit(2);                                                                              \
    }                                                                                       \
  } while (0)

static void *read_exact(FILE *f, size_t bytes) {
  void *p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "short case file\n");
    exit(2);
  }
  return p;
}
static void *to_device(const void *host, size_t bytes) {
  void *d = NULL;
  HIP_OK(hipMalloc(&d, bytes ? bytes : 4));
  if (host) HIP_OK(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
  return d;
}
static double max_abs_diff(const float *a, const float *b, size_t n) {
  double m = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const double d = fabs((double)a[i] - (double)b[i]);
    if (!(d <= m)) m = d; /* (a NaN sticks) */
  }
  return m;
}
/* largest |a - b| / limit over the elements (> 1: a failure); *worst_err, *worst_lim: error and limit of that element */
static double worst_ratio(const float *a, const float *b, const float *lim, size_t n, double *worst_err, double *worst_lim) {
  double m = -1.0;
  for (size_t i = 0; i < n; ++i) {
    const double d = fabs((double)a[i] - (double)b[i]), r = d / (double)lim[i];
    if (!(r <= m)) {
      m = r;
      *worst_err = d;
      *worst_lim = lim[i];
    }
  }
  return m;
}

/* one training step: r2l_isp_step_fwd + r2l_isp_step_bwd on one rank (include/r2l_isp.h) */
static int run_step(FILE *f, int B, int H, int W, int bn_mode) {
  const size_t npx = (size_t)B * H * W;
  float *raw = read_exact(f, 4 * npx), *packed = read_exact(f, 4 * R2L_P_COUNT), *cot = read_exact(f, 12 * npx);
  float *out_ref = read_exact(f, 12 * npx), *grad_ref = read_exact(f, 4 * R2L_P_NTRAIN);
  float *rm_ref = read_exact(f, 12), *rv_ref = read_exact(f, 12);
  float *out_lim = read_exact(f, 12 * npx), *grad_lim = read_exact(f, 4 * R2L_P_NTRAIN);
  float *d_raw = to_device(raw, 4 * npx), *d_packed = to_device(packed, 4 * R2L_P_COUNT), *d_cot = to_device(cot, 12 * npx);
  float *d_out = to_device(NULL, 12 * npx), *d_grad = to_device(NULL, 4 * R2L_P_NTRAIN);
  const float rm0[3] = {0.f, 0.f, 0.f}, rv0[3] = {1.f, 1.f, 1.f}; /* a fresh nn.BatchNorm2d */
  const long long nbt0 = 0;
  float *d_rm = to_device(rm0, 12), *d_rv = to_device(rv0, 12);
  long long *d_nbt = to_device(&nbt0, 8);
  /* the nine parameter tensors as nine device pointers, in R2L_P_* order (here: slices of one packed block) */
  static const int offs[9] = {R2L_P_BLACK_LEVEL, R2L_P_WHITE_BALANCE, R2L_P_CCM,       R2L_P_GAMMA,    R2L_P_DEBAYER,
                              R2L_P_SHARPEN,     R2L_P_BLUR,          R2L_P_M_RGB2YUV, R2L_P_M_YUV2RGB};
  const float *table[9];
  for (int i = 0; i < 9; ++i) table[i] = d_packed + offs[i];
  const size_t nws = r2l_isp_workspace_bytes(B, H, W);
  void *ws = to_device(NULL, nws);
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  const int phase = R2L_STEP_ALL | R2L_STEP_KEEP_LUMA; /* a backward follows */
  const int train = bn_mode == R2L_BN_TRAIN;
  R2L_OK(r2l_isp_step_fwd(d_raw, 0, 1.0f, table, NULL, bn_mode, train ? d_rm : NULL, train ? d_rv : NULL, train ? d_nbt : NULL,
                          1e-5, 0.1, d_out, ws, nws, B, H, W, 1, phase, NULL, stream));
  R2L_OK(r2l_isp_step_bwd(d_raw, 0, 1.0f, NULL, d_cot, d_out, d_grad, NULL, bn_mode, ws, nws, B, H, W, 1, phase, NULL, stream));
  HIP_OK(hipStreamSynchronize(stream));
  float *out = malloc(12 * npx), grad[R2L_P_NTRAIN], rm[3], rv[3];
  long long nbt = -1;
  HIP_OK(hipMemcpy(out, d_out, 12 * npx, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(grad, d_grad, sizeof grad, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(rm, d_rm, 12, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(rv, d_rv, 12, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(&nbt, d_nbt, 8, hipMemcpyDeviceToHost));
  int bad = 0;
  double we = 0.0, wl = 0.0;
  const double ro = worst_ratio(out, out_ref, out_lim, 3 * npx, &we, &wl);
  printf("out: worst |library - reference| / limit = %.3f (error %.3e, limit %.3e there)\n", ro, we, wl);
  bad |= !(ro <= 1.0);
  /* gradients, per parameter tensor */
  static const char *names[7] = {"black_level", "white_balance", "colour_correction", "gamma_correct",
                                 "debayer.weight", "sharpening_filter.weight", "gaussian_blur.weight"};
  for (int i = 0; i < 7; ++i) {
    const int lo = offs[i], hi = (i < 6) ? offs[i + 1] : R2L_P_NTRAIN;
    const double r = worst_ratio(grad + lo, grad_ref + lo, grad_lim + lo, (size_t)(hi - lo), &we, &wl);
    printf("grad %-26s worst error / limit = %.3f (error %.3e, limit %.3e)\n", names[i], r, we, wl);
    bad |= !(r <= 1.0);
  }
  if (train) {
    const double em = max_abs_diff(rm, rm_ref, 3), ev = max_abs_diff(rv, rv_ref, 3);
    printf("running_mean error %.3e, running_var error %.3e, num_batches_tracked %lld\n", em, ev, nbt);
    for (int k = 0; k < 3; ++k) /* (np.testing.assert_allclose(rtol=1e-5, atol=1e-6) of the suite) */
      bad |= !(fabs((double)rm[k] - rm_ref[k]) <= 1e-6 + 1e-5 * fabs((double)rm_ref[k])) ||
             !(fabs((double)rv[k] - rv_ref[k]) <= 1e-6 + 1e-5 * fabs((double)rv_ref[k]));
    bad |= nbt != 1;
  }
  return bad;
}

/* the static pipeline: r2l_static_fwd */
static int run_static(FILE *f, int B, int H, int W, int debayer, int sharpening, int denoising) {
  const size_t npx = (size_t)B * H * W;
  float *raw = read_exact(f, 4 * npx);
  double *cam = read_exact(f, 8 * 17); /* camera[16], gamma */
  float *out_ref = read_exact(f, 12 * npx);
  double *tol = read_exact(f, 8);
  float *d_raw = to_device(raw, 4 * npx), *d_out = to_device(NULL, 12 * npx);
  const size_t nws = r2l_static_workspace_bytes(B, H, W, debayer, sharpening, denoising);
  void *ws = nws ? to_device(NULL, nws) : NULL;
  R2L_OK(r2l_static_fwd(d_raw, d_out, B, H, W, cam, debayer, sharpening, denoising, cam[16], ws, nws, NULL));
  HIP_OK(hipDeviceSynchronize());
  float *out = malloc(12 * npx);
  HIP_OK(hipMemcpy(out, d_out, 12 * npx, hipMemcpyDeviceToHost));
  const double eo = max_abs_diff(out, out_ref, 3 * npx);
  printf("static chain (%d, %d, %d): max |library - reference| = %.3e (limit %.1e)\n", debayer, sharpening, denoising, eo, tol[0]);
  return !(eo <= tol[0]);
}

int main(int argc, char **argv) {
  if (argc == 2 && !strcmp(argv[1], "--abi")) { /* no GPU needed: the CPU suite's link check */
    printf("abi %d device_build %d\n", r2l_abi_version(), r2l_is_device_build());
    return r2l_abi_version() == R2L_ABI_VERSION ? 0 : 1;
  }
  if (argc != 2) {
    fprintf(stderr, "usage: %s <case.bin> | --abi\n", argv[0]);
    return 2;
  }
  FILE *f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  int32_t *h = read_exact(f, 32);
  if (h[0] != 0x52324c31) {
    fprintf(stderr, "not a case file\n");
    return 2;
  }
  if (!r2l_is_device_build()) {
    fprintf(stderr, "not the device library\n");
    return 2;
  }
  const int bad = h[1] == 0 ? run_step(f, h[2], h[3], h[4], h[5]) : run_static(f, h[2], h[3], h[4], h[5], h[6], h[7]);
  fclose(f);
  printf(bad ? "FAILED\n" : "ok\n");
  return bad;
}
