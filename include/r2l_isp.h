/* r2l_isp.h -- C ABI of libr2l_isp.so: the MI355X (gfx950) ISP hot path of raw2logit.
 *
 * The reference (aiaudit-org/raw2logit) is pure Python and has no FFI for this path: its boundary is
 * the nn.Module API of processing/pipeline_torch.py and the callable API of
 * processing/pipeline_numpy.py.  This header is the C boundary a maintainer binds underneath those
 * modules (INTEGRATION.md shows the ctypes stub).  Each entry point names the reference code it
 * replaces (path:line relative to the reference tree).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in _host;
 *  - the caller owns every buffer; the library allocates no persistent device memory;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*) and returns;
 *  - return value 0 = success, negative = error (r2l_last_error() gives the thread-local text);
 *  - images are float32, raw is (B,H,W), RGB is (B,3,H,W) contiguous NCHW; H and W even, >= 4;
 *  - no global mutable state: calls on different streams may run concurrently.
 */
#ifndef R2L_ISP_H
#define R2L_ISP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R2L_ABI_VERSION 1

/* ---- packed parameter block (device, float32[R2L_P_COUNT]) -----------------------------------
 * The trainable parameters of ParametrizedProcessing (processing/pipeline_torch.py:152-166) in
 * state_dict order, followed by the two registered buffers (:170-171).                           */
enum {
  R2L_P_BLACK_LEVEL = 0,   /* black_level            (4,)      R,G1,G2,B   :154 */
  R2L_P_WHITE_BALANCE = 4, /* white_balance          (1,3)                 :155 */
  R2L_P_CCM = 7,           /* colour_correction      (3,3) [k][c]          :156 */
  R2L_P_GAMMA = 16,        /* gamma_correct          (1,)                  :158 */
  R2L_P_DEBAYER = 17,      /* debayer.weight         (3,3,3,3) [k][c][i][j] :228-237 */
  R2L_P_SHARPEN = 98,      /* sharpening_filter.weight (1,1,3,3)           :162-163 */
  R2L_P_BLUR = 107,        /* gaussian_blur.weight   (1,1,5,5)             :165-166 */
  R2L_P_NTRAIN = 132,      /* number of trainable scalars == length of grad_params */
  R2L_P_M_RGB2YUV = 132,   /* buffer M_RGB_2_YUV (3,3)                     :170 */
  R2L_P_M_YUV2RGB = 141,   /* buffer M_YUV_2_RGB (3,3)                     :171 */
  R2L_P_COUNT = 150
};

/* flags for r2l_isp_fwd / r2l_isp_bwd */
enum {
  R2L_F_STATS_ONLY = 1, /* fwd: do not write `out`, only the BatchNorm partial sums */
  R2L_F_FOLDED_VALID = 2, /* fwd/bwd: `workspace` was last used by a call with these same `params`;
                             skip re-deriving the folded weights (saves one small launch) */
  R2L_F_KEEP_LUMA = 4, /* fwd (with `out`): keep the sharpened luma plane Y' in the workspace (+4 B/px written);
                          bwd: the forward of this step ran with it -- read Y' there instead of recomputing
                          raw -> Y -> Y' (-25 % of the first gradient kernel for 4.5 B/px read).  Both calls of
                          a step must agree; honoured where the row-streaming forward runs (no additive layer,
                          W % 4 == 0, W <= 2048), ignored elsewhere */
};

/* static-pipeline selectors (processing/pipeline_numpy.py:92-122) */
enum { R2L_DEBAYER_BILINEAR = 0, R2L_DEBAYER_MALVAR2004 = 1,
       R2L_DEBAYER_MENON2007 = 2 /* DDFAPD with the refining step (pipeline_numpy.py:96-97); float64 plane passes, W % 4 == 0,
                                    workspace from r2l_static_workspace_bytes(); third-party algorithm, parity unpinned */ };
enum { R2L_SHARPEN_NONE = 0, R2L_SHARPEN_FILTER = 1, R2L_SHARPEN_UNSHARP = 2 };
enum { R2L_DENOISE_NONE = 0, R2L_DENOISE_GAUSSIAN = 1, R2L_DENOISE_MEDIAN = 2, R2L_DENOISE_FFT = 3 };

int r2l_abi_version(void);
const char *r2l_last_error(void);

/* 1 when the library runs on a GPU (libr2l_isp.so); 0 for the test-only host emulation of the same
 * kernels (tests/_build/libr2l_emul.so), which the product loader refuses. */
int r2l_is_device_build(void);

/* Profiling aid for bench.py: with timing enabled every kernel launch is bracketed by hipEvents on the
 * launch stream; r2l_timing_report() waits for them, writes "kernel_name count total_ms\n" lines into
 * buf (NUL-terminated, truncated to n) and clears the log.  Returns the text length.               */
void r2l_timing_enable(int on);
int r2l_timing_report(char *buf, size_t n);

/* ---- raw2rgb (processing/pipeline_torch.py:240-283; RawToRGB :43-80, NNProcessing front end :111)
 * black_level: 4 floats (R,G1,G2,B) or NULL.  out: (B,out_channels,H,W) if !reduce_size (zero-filled
 * mosaic) else (B,out_channels,H/2,W/2); out_channels in {3,4}.                                  */
int r2l_raw2rgb_fwd(const float *raw, const float *black_level, float *out, int B, int H, int W,
                    int reduce_size, int out_channels, void *stream);
/* VJP: grad_raw (B,H,W) (or NULL) and grad_black_level double[4] (or NULL; needs workspace of
 * r2l_raw2rgb_bwd_workspace_bytes()).                                                            */
size_t r2l_raw2rgb_bwd_workspace_bytes(int B, int H, int W);
int r2l_raw2rgb_bwd(const float *grad_out, float *grad_raw, double *grad_black_level, void *workspace,
                    size_t workspace_bytes, int B, int H, int W, int reduce_size, int out_channels,
                    void *stream);

/* ---- fused parametrized ISP, torch semantics (ParametrizedProcessing.forward,
 * processing/pipeline_torch.py:175-225 with track_stages=False):
 *   black level + mosaic (:183) -> Debayer 3x3 mirror-pad conv (:187) -> white balance (:190) ->
 *   CCM (:191) -> RGB->YUV (:194) -> sharpen Y 3x3 zero-pad (:195) -> blur Y 5x5 mirror-pad (:202) ->
 *   YUV->RGB (:203) -> clip[1e-5,1] (:206) -> exp(log(x)/gamma) (:209) -> [+ additive_layer (:213)] ->
 *   [BatchNorm2d(3, affine=False) normalisation (:217)]
 * in ONE kernel: 4 B/px read, 12 B/px written.
 *
 *  params    float[R2L_P_COUNT]
 *  additive  float[3*256*256] or NULL (requires H == W == 256, as in the reference :130)
 *  bn_mean_istd  float[6] = mean[3], 1/sqrt(var+eps)[3] to apply, or NULL for no normalisation
 *  out       (B,3,H,W), may be NULL with R2L_F_STATS_ONLY
 *  stats     double[7] or NULL: receives sum_c(x-0.5)[3], sum_c((x-0.5)^2)[3] of the PRE-normalisation
 *            output over this call's B*H*W pixels, then that pixel count (r2l_bn_finalize turns them into
 *            batch mean / biased variance; with several GPUs the vectors of all ranks are summed first).
 *            The per-workgroup partial sums are added in a fixed order by the last workgroups of the same
 *            launch (no separate reduction launch, bitwise reproducible).
 *  workspace r2l_isp_workspace_bytes() of device memory, uninitialised on first use; a call WITHOUT
 *            R2L_F_FOLDED_VALID initialises it (folded weights, arrival counters).  A workspace must not be
 *            used by two streams at once.
 */
size_t r2l_isp_workspace_bytes(int B, int H, int W);
int r2l_isp_fwd(const float *raw, const float *params, const float *additive,
                const float *bn_mean_istd, float *out, double *stats, void *workspace,
                size_t workspace_bytes, int B, int H, int W, int flags, void *stream);

/* BatchNorm2d(3, affine=False) bookkeeping of train mode (:216-217) without a host round trip.
 * stats double[nranks][7] = the `stats` vectors of r2l_isp_fwd of all ranks (nranks = 1: this rank's own;
 * several GPUs: the all-gathered vectors), added here in rank order so that every rank derives bit-identical
 * statistics, equal to the single-GPU statistics of the global batch.  Writes bn_mean_istd float[6] (for the
 * apply pass), optionally moments double[7] = batch mean[3], biased variance[3], global pixel count, and
 * optionally updates running_mean / running_var float[3] the way nn.BatchNorm2d does (momentum, unbiased
 * variance).  num_batches_tracked (device int64, optional) is incremented; momentum < 0 selects the cumulative
 * moving average 1/num_batches_tracked (nn.BatchNorm2d(momentum=None)).                                  */
int r2l_bn_finalize(const double *stats, int nranks, float *bn_mean_istd, double *moments,
                    float *running_mean, float *running_var, long long *num_batches_tracked, double eps,
                    double momentum, void *stream);

/* One rank: the statistics pass of r2l_isp_fwd (R2L_F_STATS_ONLY) and r2l_bn_finalize in ONE launch -- the
 * workgroup that finishes the reduction also derives mean / 1/std and updates the running statistics.  The
 * workspace is initialised by this call; the apply pass follows with R2L_F_FOLDED_VALID.                 */
int r2l_isp_fwd_stats_bn(const float *raw, const float *params, const float *additive, double *stats,
                         float *bn_mean_istd, double *moments, float *running_mean, float *running_var,
                         long long *num_batches_tracked, double eps, double momentum, void *workspace,
                         size_t workspace_bytes, int B, int H, int W, void *stream);

/* Several GPUs: gathered_sums double[nranks][6] = the `sums` vectors of r2l_bn_bwd_reduce of all ranks, added in
 * rank order and divided by *n (global pixel count, moments[6]) -> bn_bwd float[6] for r2l_isp_bwd.      */
int r2l_bn_bwd_means(const double *gathered_sums, int nranks, const double *n, float *bn_bwd, void *stream);

/* BatchNorm backward reduction (nn.BatchNorm2d backward in train mode, :216-217):
 * sums double[6] = sum_c(g)[3], sum_c(g*xhat)[3] with xhat == the saved forward output; with `totals`
 * (double[7], totals[6] = pixel count of the global batch: the moments of r2l_bn_finalize) and bn_bwd (float[6]) also the means
 * bn_bwd = sums / n that r2l_isp_bwd consumes (single GPU; with several GPUs the caller all-reduces
 * `sums` and divides itself).  flags: R2L_F_FOLDED_VALID = the workspace went through r2l_isp_fwd /
 * r2l_isp_bwd before (the reduction then finishes inside the same launch).                        */
int r2l_bn_bwd_reduce(const float *grad_out, const float *out, const double *totals, double *sums,
                      float *bn_bwd, void *workspace, size_t workspace_bytes, int B, int H, int W,
                      int flags, void *stream);

/* Backward of r2l_isp_fwd (what autograd computes for pipeline_torch.py:183-217), recomputing the
 * forward from `raw`:
 *  bn_mean_istd  float[6] as given to the forward (NULL: no BatchNorm)
 *  bn_bwd        float[6] = mean_c(g)[3], mean_c(g*xhat)[3] over the GLOBAL batch (train mode), or NULL
 *                (eval mode: only the 1/std scaling applies)
 *  grad_params   float[R2L_P_NTRAIN], layout of the packed block
 *  grad_raw      (B,H,W) or NULL
 */
int r2l_isp_bwd(const float *raw, const float *params, const float *additive,
                const float *bn_mean_istd, const float *bn_bwd, const float *grad_out,
                float *grad_params, float *grad_raw, void *workspace, size_t workspace_bytes, int B,
                int H, int W, int flags, void *stream);

/* gradient of the additive layer (:213): grad_additive[c,y,x] = sum_b d loss / d x[b,c,y,x], where the
 * BatchNorm backward (if any) is applied on the fly from grad_out and the saved output.          */
int r2l_additive_bwd(const float *grad_out, const float *out, const float *bn_mean_istd,
                     const float *bn_bwd, float *grad_additive, int B, int H, int W, void *stream);

/* ---- one training step of ParametrizedProcessing in two calls ------------------------
 * r2l_isp_step_fwd = everything ParametrizedProcessing.forward (:175-225) enqueues for one batch:
 *   the nine parameter tensors (device pointers, in R2L_P_* order: black_level, white_balance,
 *   colour_correction, gamma_correct, debayer.weight, sharpening_filter.weight, gaussian_blur.weight,
 *   M_RGB_2_YUV, M_YUV_2_RGB; the table params_host itself is HOST memory) are gathered into the packed block
 *   inside the workspace and folded by one small launch; train-mode BatchNorm: statistics pass + bookkeeping
 *   (running statistics, num_batches_tracked: nn.BatchNorm2d semantics, momentum < 0 = cumulative average);
 *   eval mode: (mean, 1/sqrt(var+eps)) from the running statistics; then the apply pass -> out.
 * r2l_isp_step_bwd = the whole backward: BatchNorm backward sums, both gradient kernels -> grad_params
 *   (float32[R2L_P_NTRAIN], may be NULL), the additive layer's gradient (may be NULL).
 * The workspace (r2l_isp_workspace_bytes) carries the step's state from the forward to the backward: packed
 * parameters as the forward saw them, folded weights, BatchNorm constants.  It must not be used by another
 * step in between.
 * raw: float32 frames (raw_u16 = 0) or 16-bit containers (raw_u16 = 1, denom = 2**bits - 1).
 * Several ranks (train-mode BatchNorm only): the statistics / backward sums cross ranks between phase A and
 * phase B --  A: this rank's vector lands in the workspace at r2l_isp_step_offset(R2L_STEP_STATS |
 * R2L_STEP_BN_SUMS) (7 resp. 6 doubles); the caller all-gathers them (RCCL) and hands the rank-major result to
 * phase B, which adds the rows in rank order.  One rank: phase = R2L_STEP_ALL.                             */
enum { R2L_BN_NONE = 0, R2L_BN_TRAIN = 1, R2L_BN_EVAL = 2 };
enum { R2L_STEP_ALL = 0, R2L_STEP_A = 1, R2L_STEP_B = 2 };
/* or-ed into `phase` of r2l_isp_step_fwd AND r2l_isp_step_bwd of a step whose backward will run: R2L_F_KEEP_LUMA */
enum { R2L_STEP_KEEP_LUMA = 8 };
/* Output epilogue: the weak augmentation of utils/augmentation.py:70-74, applied to the
 * processor's output at model.py:79-81 -- rot90^k(vflip(hflip(out))) over the last two axes, k as in
 * out.rot90(k, dims=(-1, -2)) -- written by the forward's own stores instead of a separate permutation pass.  Or these
 * into `phase` of r2l_isp_step_fwd AND of the r2l_isp_step_bwd of the same step: `out` / `grad_out` are then in the
 * augmented layout ((B,3,W,H) for odd k, which needs H == W).  Not with an additive layer (error -3: use r2l_augment
 * on the plain output).  Bit-identical to r2l_augment(r2l_isp_step_fwd(...)).                                        */
enum { R2L_STEP_EPI_HFLIP = 16, R2L_STEP_EPI_VFLIP = 32, R2L_STEP_EPI_ROT_SHIFT = 6 /* k << 6: bits 64, 128 */ };
enum { R2L_STEP_STATS = 0, R2L_STEP_MOMENTS = 1, R2L_STEP_BN_SUMS = 2, R2L_STEP_PACKED = 3, R2L_STEP_BN = 4,
       R2L_STEP_LUMA = 5 /* (B,H,W) float32: the sharpened luma plane Y' a train-mode / R2L_STEP_KEEP_LUMA forward leaves */ };
size_t r2l_isp_step_offset(int which, int B, int H, int W); /* byte offset inside the workspace */
int r2l_isp_step_fwd(const void *raw, int raw_u16, float denom, const float *const *params_host,
                     const float *additive, int bn_mode, float *running_mean, float *running_var,
                     long long *num_batches_tracked, double eps, double momentum, float *out, void *workspace,
                     size_t workspace_bytes, int B, int H, int W, int nranks, int phase,
                     const double *gathered_stats, void *stream);
int r2l_isp_step_bwd(const void *raw, int raw_u16, float denom, const float *additive, const float *grad_out,
                     const float *out, float *grad_params, float *grad_additive, int bn_mode, void *workspace,
                     size_t workspace_bytes, int B, int H, int W, int nranks, int phase,
                     const double *gathered_sums, void *stream);

/* ---- static pipeline, numpy semantics (processing(), processing/pipeline_numpy.py:70-141, batched):
 * remove_blacklv (:152-158) -> demosaicing_CFA_Bayer_{bilinear,Malvar2004} (:92-95) -> wb (:161-162) ->
 * CCM (:165-167) -> [sharpening_filter (:180-191) | unsharp_masking (:170-177)] -> [gaussian_denoising (:203-209) | median_denoising
 * (:194-200)] -> clip[0,1] (:138) -> x**(1/gamma) (:241-244).  Linear part in float64 like the reference (black level: see below),
 * output (B,3,H,W) float32 (RawProcessingPipeline.__call__, :55-67).  camera_host: double[16] =
 * black_level[4], white_balance[3], colour_matrix[9] (host memory).
 * R2L_DENOISE_FFT = fft_denoising (:212-238 as :121-122 calls it: keep_fraction 0.3, columns only) -- per image row
 * and colour channel an ideal low-pass along the columns of the sharpened RGB image: luma-plane passes + rocFFT
 * (real-to-complex, Hermitian-symmetrised mask, complex-to-real) + a clip / gamma pass; W % 4 == 0; workspace
 * from r2l_static_workspace_bytes().
 * On frames with W % 4 == 0 up to 2048 wide (1024 behind unsharp_masking) every other combination is ONE launch
 * (row-streaming kernels); otherwise the short chain and the train.py defaults (bilinear + sharpening_filter
 * + gaussian_denoising, train.py:96-101) run as tile kernels and the other combinations as float64
 * luma-plane passes, which need r2l_static_workspace_bytes() of device memory (0 for the single-launch
 * ones; workspace may then be NULL) and W % 4 == 0.                                                     */
size_t r2l_static_workspace_bytes(int B, int H, int W, int debayer, int sharpening, int denoising);
int r2l_static_fwd(const float *raw, float *out, int B, int H, int W, const double *camera_host,
                   int debayer, int sharpening, int denoising, double gamma, void *workspace,
                   size_t workspace_bytes, void *stream);
/* remove_blacklv (:152-158) subtracts IN PLACE, i.e. in the dtype of the frame it is handed.  The reference's
 * datasets hand processing() float32 frames for every tif / png tile (utils/dataset_utils.py:18-26,
 * dataset.py:86-87; docstring :57): r2l_static_fwd (and the 16-bit entry point below, whose division is the
 * datasets' float32 division) therefore rounds the black level to float32 and subtracts in float32 before
 * widening -- near-black pixels differ by up to 1e-4 after the gamma from a float64 subtraction.  A float64
 * ndarray (a DNG: uint16 raw_image_visible / (2**bits - 1) is float64) keeps float64 arithmetic throughout:
 * r2l_static_fwd_f64 reads float64 frames (8 B/px; W % 4 == 0; workspace: r2l_static_workspace_bytes_f64()). */
size_t r2l_static_workspace_bytes_f64(int B, int H, int W, int debayer, int sharpening, int denoising);
int r2l_static_fwd_f64(const double *raw, float *out, int B, int H, int W, const double *camera_host,
                       int debayer, int sharpening, int denoising, double gamma, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ---- 16-bit ingest.  The reference's datasets deliver the sensor's 16-bit
 * containers and normalise them on the host: img = load_image(path) / (2**bits - 1), float32
 * (dataset.py:86-87, :140-143; utils/dataset_utils.py:18-26).  The *_u16 variants read the containers
 * themselves (2 B/px instead of 4) and apply that float32 division inside the kernel (denom = 2**bits - 1;
 * the result equals the correctly rounded float32 quotient, so every output is bit-identical to the float32
 * entry point fed with the host-normalised frame).  W % 4 == 0 required.  No grad_raw (integer input).   */
int r2l_isp_fwd_u16(const unsigned short *raw, float denom, const float *params, const float *additive,
                    const float *bn_mean_istd, float *out, double *stats, void *workspace,
                    size_t workspace_bytes, int B, int H, int W, int flags, void *stream);
int r2l_isp_bwd_u16(const unsigned short *raw, float denom, const float *params, const float *additive,
                    const float *bn_mean_istd, const float *bn_bwd, const float *grad_out,
                    float *grad_params, void *workspace, size_t workspace_bytes, int B, int H, int W,
                    int flags, void *stream);
int r2l_isp_fwd_stats_bn_u16(const unsigned short *raw, float denom, const float *params, const float *additive,
                             double *stats, float *bn_mean_istd, double *moments, float *running_mean,
                             float *running_var, long long *num_batches_tracked, double eps, double momentum,
                             void *workspace, size_t workspace_bytes, int B, int H, int W, void *stream);
int r2l_raw2rgb_fwd_u16(const unsigned short *raw, float denom, const float *black_level, float *out, int B,
                        int H, int W, int reduce_size, int out_channels, void *stream);
int r2l_static_fwd_u16(const unsigned short *raw, float denom, float *out, int B, int H, int W,
                       const double *camera_host, int debayer, int sharpening, int denoising, double gamma,
                       void *workspace, size_t workspace_bytes, void *stream);
/* The static pipeline with the T.Normalize(mean, std) that train.py:157-171 composes behind RawProcessingPipeline
 * fused into the kernels' stores: out = (processing(raw) - mean[c]) / std[c], float32 subtraction and division
 * like torchvision's.  mean_std_host: float[6] = mean[3], std[3] in host memory, or NULL (then identical to the
 * entry points above).  frames: which container `raw` points to (denom is read for R2L_FRAMES_U16 only);
 * workspace as r2l_static_workspace_bytes() / _f64().                                                       */
enum { R2L_FRAMES_F32 = 0, R2L_FRAMES_U16 = 1, R2L_FRAMES_F64 = 2 };
int r2l_static_fwd_norm(const void *raw, int frames, float denom, float *out, int B, int H, int W,
                        const double *camera_host, int debayer, int sharpening, int denoising, double gamma,
                        const float *mean_std_host, void *workspace, size_t workspace_bytes, void *stream);

/* processing()'s numeric arguments (pipeline_numpy.py:70-73: sharp_radius, sharp_amount, median_kernel_size, gaussian_sigma,
 * fft_fraction; used at :117-122) as launch arguments: options_host = double[R2L_SOPT_COUNT] in host memory, or NULL for the
 * reference's defaults (then identical to r2l_static_fwd_norm).  What the kernels' windows hold bounds the values:
 *   gaussian_sigma   (0, 0.625)   scipy's window radius int(4 sigma + 0.5) <= 2 (the 5-tap window of gaussian_denoising);
 *   sharp_radius     (0, 1.125)   radius int(4 sigma + 0.5) <= 4 (the 9-tap window behind unsharp_masking); sharp_amount any;
 *   fft_fraction     [0, 0.5];    median_kernel_size 3 (the fused kernels' network) or 5 (then the chain runs as luma-plane
 * passes: W % 4 == 0, workspace from r2l_static_workspace_bytes_opts) -- anything else returns -4 with the reason in
 * r2l_last_error().  An option of a stage the chain does not run is ignored, like the reference's if-chains.             */
enum { R2L_SOPT_SHARP_RADIUS = 0, R2L_SOPT_SHARP_AMOUNT = 1, R2L_SOPT_GAUSSIAN_SIGMA = 2, R2L_SOPT_FFT_FRACTION = 3,
       R2L_SOPT_MEDIAN_SIZE = 4, R2L_SOPT_COUNT = 5 };
size_t r2l_static_workspace_bytes_opts(int frames, int B, int H, int W, int debayer, int sharpening, int denoising,
                                       const double *options_host);
int r2l_static_fwd_opts(const void *raw, int frames, float denom, float *out, int B, int H, int W,
                        const double *camera_host, int debayer, int sharpening, int denoising, double gamma,
                        const double *options_host, const float *mean_std_host, void *workspace,
                        size_t workspace_bytes, void *stream);

/* ---- staged execution (track_stages=True, pipeline_torch.py:197-221): one entry point per materialised
 * stage, each with its VJP, so that autograd can hold every stage tensor (retain_grad) and d/d raw exists.
 * Tensors are (B,3,H,W) float32.  Weight gradients are float32 arrays; workspace from
 * r2l_stage_workspace_bytes().
 *   conv33   Debayer, 3->3 3x3 cross-correlation, mirror padding                          :187, :228-237
 *   mix3     einsum('bchw,kc->bkhw', x, M) (white balance = diagonal M)                    :190-194, :198-203
 *   pconv    channel 0 <- KxK conv of channel 0 (K=3 zero pad :195 | K=5 mirror pad :202), 1-2 copied;
 *            gk25 is 5x5-strided (entry [i*5+j])
 *   point    op 0/1 clip fwd/bwd (:206)  2/3 gamma fwd/bwd (:209; sums6[0] = sum g*out*log2(x))
 *            4 add (:213)  5 BatchNorm apply  6 BatchNorm backward  7 BatchNorm statistics (sums6)
 *            8 (x - w[c]) / w[3+c]: the T.Normalize(mean, std) that follows the static pipeline (train.py:157-171) */
size_t r2l_stage_workspace_bytes(void);
int r2l_stage_conv33_fwd(const float *x, const float *w, float *y, int B, int H, int W, void *stream);
int r2l_stage_conv33_bwd(const float *x, const float *w, const float *g, float *gx, float *gw,
                         void *workspace, size_t workspace_bytes, int B, int H, int W, void *stream);
int r2l_stage_mix3_fwd(const float *x, const float *m, float *y, int B, int H, int W, void *stream);
int r2l_stage_mix3_bwd(const float *x, const float *m, const float *g, float *gx, float *gm, void *workspace,
                       size_t workspace_bytes, int B, int H, int W, void *stream);
int r2l_stage_pconv_fwd(const float *x, const float *k, float *y, int K, int mirror, int B, int H, int W,
                        void *stream);
int r2l_stage_pconv_bwd(const float *x, const float *k, const float *g, float *gx, float *gk25, int K,
                        int mirror, void *workspace, size_t workspace_bytes, int B, int H, int W,
                        void *stream);
int r2l_stage_point(int op, const float *x, const float *g, const float *w, const float *aux,
                    const float *aux2, float *y, float *sums6, void *workspace, size_t workspace_bytes,
                    int B, int H, int W, void *stream);

/* ---- augmentation of the ISP output (utils/augmentation.py:8-31, :70-74;
 * applied between processor and classifier, model.py:79-81).  x is N planes of H x W float32.
 *   r2l_augment    y = rot90^k(vflip(hflip(x))), k counted like x.rot90(k, dims=(-1, -2)); y is N planes of
 *                  H x W (even k) or W x H (odd k).  inverse != 0: x has y's shape and the inverse map is
 *                  applied (the VJP).  The random draws stay on the host, in the reference's order.
 *   r2l_add_noise  y = x + noise * std (AddGaussianNoise; the caller draws `noise`)
 *   r2l_add_noise_philox  y[i] = x[i] + std * n(seed, offset, i): the N(0,1) deviates are generated in the kernel
 *                  (Philox4x32-10, counter = (i / 4, offset), key = seed, Box-Muller on the four outputs), a pure
 *                  function of its arguments -- no noise tensor exists.  Same distribution as the reference's
 *                  torch.randn_like (utils/augmentation.py:17-30), not the same stream.                       */
int r2l_augment(const float *x, float *y, int N, int H, int W, int hflip, int vflip, int k, int inverse,
                void *stream);
int r2l_add_noise(const float *x, const float *noise, float std, float *y, size_t n, void *stream);
int r2l_add_noise_philox(const float *x, float *y, float std, unsigned long long seed, unsigned long long offset,
                         size_t n, void *stream);

/* ---- adversarial auxiliary losses between the outputs of two processors (SURVEY.md section 8f rank 2;
 * AuxLoss, utils/base.py:346-358: img1 = the default processor's output, img2 = the adversarial processor's).
 *   r2l_ssim_fwd   mean of the SSIM map, window_size 11, sigma 1.5, zero padding, per channel
 *                  (utils/ssim.py:9-39, SSIM(window_size=11), size_average=True) -> ssim_mean double[1]
 *                  keep_for_backward: also leave dS/d(mu2, E[y^2], E[xy]) of every pixel in the workspace
 *   r2l_ssim_bwd   grad_img2 = grad_ssim[0] * d mean-SSIM / d img2   (grad_ssim: device float scalar);
 *                  workspace_has_dmaps: the workspace is the one a keep_for_backward forward filled (else the
 *                  maps are recomputed first)
 *   r2l_l2_fwd     ((x - y) ** 2).sum() (utils/base.py:342-343) -> sum double[1]; n elements, n % 4 == 0
 *   r2l_l2_bwd     grad_y = grad_sum[0] * 2 (y - x)
 * Images are (B,C,H,W) float32; workspace from r2l_aux_workspace_bytes().                              */
size_t r2l_aux_workspace_bytes(int B, int C, int H, int W);
int r2l_ssim_fwd(const float *img1, const float *img2, double *ssim_mean, void *workspace,
                 size_t workspace_bytes, int keep_for_backward, int B, int C, int H, int W, void *stream);
int r2l_ssim_bwd(const float *img1, const float *img2, const float *grad_ssim, float *grad_img2,
                 void *workspace, size_t workspace_bytes, int workspace_has_dmaps, int B, int C, int H, int W,
                 void *stream);
int r2l_l2_fwd(const float *x, const float *y, double *sum, void *workspace, size_t workspace_bytes, size_t n,
               void *stream);
int r2l_l2_bwd(const float *x, const float *y, const float *grad_sum, float *grad_y, size_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* R2L_ISP_H */
