"""CPU oracle for the raw2logit ISP hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``raw2logit_amd``) never does; it fails loudly when the HIP library is
missing instead of falling back to anything in here.

What is restated (all citations relative to /root/reference):

* torch ("parametrized") semantics -- ``processing/pipeline_torch.py``
    raw2rgb                    :240-283
    Debayer (3->3 3x3 conv, mirror pad)             :228-237
    ParametrizedProcessing.forward                  :175-225
  plus a hand-written reverse pass (one VJP per forward op, the chain autograd walks), so that
  parameter / input / per-stage gradients can be checked where the reference cannot run.
* numpy ("static") semantics -- ``processing/pipeline_numpy.py``
    processing                 :70-141
    remove_blacklv             :152-158
    wb_correction              :161-162
    colour_correction          :165-167
    sharpening_filter          :180-191
    gaussian_denoising         :203-209
    median_denoising           :194-200
    fft_denoising              :212-238
    unsharp_masking            :170-177
    adjust_gamma               :241-244

Pinning status
--------------
* torch semantics: PINNED.  ``oracle/gen_golden.py`` imports the reference's own
  ``processing.pipeline_torch`` (unmodified, via sys.modules stubs for absent third-party
  packages) in the build container and stores outputs, every stage, all parameter gradients,
  the input gradient and BatchNorm running statistics under ``tests/golden/``;
  ``tests/test_oracle_golden.py`` checks this file against them.
* numpy semantics: the reference's own arithmetic (black level, WB, CCM, convolve2d sharpening,
  ndimage.gaussian_filter, clip, gamma) is PINNED the same way (``processing()`` itself is run).
  Three pieces of arithmetic live in third-party packages that are NOT vendored in the reference
  tree and are NOT installed in this image -- for those the parity is UNPINNED and rests on the
  published algorithm, restated below:
    - colour-demosaicing==0.1.6 (environment.yml:296): demosaicing_CFA_Bayer_bilinear,
      demosaicing_CFA_Bayer_Malvar2004 and demosaicing_CFA_Bayer_Menon2007 (call sites pipeline_numpy.py:92-97).  Bilinear is partially
      pinned by the reference's own K_G / K_RB restatement (pipeline_torch.py:13-19).
    - scikit-image==0.18.1 (environment.yml:343): rgb2yuv / yuv2rgb (call sites :184-189, :205-207).
      Pinned by M_RGB_2_YUV / M_YUV_2_RGB in pipeline_torch.py:21-26 (inverse agrees to 3.8e-8).
    - scikit-image==0.18.1 unsharp_mask (call site :174, reached with multichannel=True from :117): restated
      in unsharp_mask() below from the package's published source; nothing in the reference tree pins it.
* adversarial auxiliary losses (utils/ssim.py SSIM / ssim, utils/base.py l2_regularization): PINNED --
  both are pure torch and run unmodified in gen_golden.py (tests/golden/aux_losses.npz).
  The reference has no tests and no golden vectors of its own.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------------
# constants (pipeline_torch.py:13-40)
# ----------------------------------------------------------------------------------------------
K_G = np.array([[0, 1, 0], [1, 4, 1], [0, 1, 0]], dtype=np.float32) / 4
K_RB = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], dtype=np.float32) / 4
M_RGB_2_YUV = np.array([[0.299, 0.587, 0.114],
                        [-0.14714119, -0.28886916, 0.43601035],
                        [0.61497538, -0.51496512, -0.10001026]], dtype=np.float32)
M_YUV_2_RGB = np.array([[1.0000000000e+00, -4.1827794561e-09, 1.1398830414e+00],
                        [1.0000000000e+00, -3.9464232326e-01, -5.8062183857e-01],
                        [1.0000000000e+00, 2.0320618153e+00, -1.2232658220e-09]], dtype=np.float32)
K_BLUR = np.array([[6.9625e-08, 2.8089e-05, 2.0755e-04, 2.8089e-05, 6.9625e-08],
                   [2.8089e-05, 1.1332e-02, 8.3731e-02, 1.1332e-02, 2.8089e-05],
                   [2.0755e-04, 8.3731e-02, 6.1869e-01, 8.3731e-02, 2.0755e-04],
                   [2.8089e-05, 1.1332e-02, 8.3731e-02, 1.1332e-02, 2.8089e-05],
                   [6.9625e-08, 2.8089e-05, 2.0755e-04, 2.8089e-05, 6.9625e-08]], dtype=np.float32)
K_SHARP = np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]], dtype=np.float32)
DEFAULT_CAMERA_PARAMS = ([0., 0., 0., 0.], [1., 1., 1.], [1., 0., 0., 0., 1., 0., 0., 0., 1.])

# camera parameters the path is run with (dataset.py:209-213 Drone, :290-294 Microscopy)
DRONE_CAMERA_PARAMS = (
    [0.0625, 0.0626, 0.0625, 0.0626],
    [2.86653646, 1., 1.73079425],
    [1.50768983, -0.33571374, -0.17197604, -0.23048614, 1.70698738, -0.47650126,
     -0.03119153, -0.32803956, 1.35923111],
)
MICROSCOPY_CAMERA_PARAMS = (
    [9.834368023181512e-06] * 4,
    [-0.6567, 1.9673, 3.5304],
    [-2.0338, 0.0933, 0.4157, -0.0286, 2.6464, -0.0574, -0.5516, -0.0947, 2.9308],
)
CAMERAS = {'drone': DRONE_CAMERA_PARAMS, 'microscopy': MICROSCOPY_CAMERA_PARAMS,
           'identity': DEFAULT_CAMERA_PARAMS}

STAGE_ORDER = ('demosaic', 'color_correct', 'sharpening', 'gaussian', 'clipped', 'gamma_correct',
               'noise')


# ----------------------------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------------------------
def synth_raw(B, H, W, seed=0, kind='uniform'):
    """12-bit RGGB frames scaled by 1/4095 (dataset.py:87 scales by 1/(2**bits-1)).

    kind='uniform' is the perf distribution, kind='scene' the smooth parity distribution."""
    rng = np.random.default_rng(seed)
    if kind == 'uniform':
        u16 = rng.integers(0, 4096, (B, H, W))
    elif kind == 'scene':
        y, x = np.mgrid[0:H, 0:W]
        base = 0.25 + 0.2 * np.sin(x / 9.0) + 0.15 * np.cos(y / 7.0)
        u16 = np.clip(np.round(4095 * base[None] + rng.normal(0, 8, (B, H, W))), 0, 4095)
    elif kind == 'dark':
        # engineered so that pre-gamma values land below 0, inside (0, 1e-4) and above 1
        u16 = rng.integers(0, 4096, (B, H, W))
        u16[:, : H // 2] = rng.integers(250, 262, (B, H // 2, W))   # ~ black level of Drone
        u16[:, :2, :] = 4095
    elif kind == 'at_black':
        # float32 frames sitting ON the black level: every value is float32(black_level[site]) + k float32 ulps,
        # k in -3..3 (top quarter: the uniform distribution).  remove_blacklv in float32 (float32 frames, the
        # reference's datasets) gives exact multiples of the ulp, in float64 it also keeps black_level -
        # float32(black_level) ~ 3e-9 -- after x ** (1/2.2) that is 0 against ~2e-4 on the flat-zero pixels.
        bl = np.asarray(DRONE_CAMERA_PARAMS[0], dtype=np.float32)
        site = np.empty((H, W), dtype=np.float32)
        site[0::2, 0::2], site[0::2, 1::2], site[1::2, 0::2], site[1::2, 1::2] = bl
        k = rng.integers(-3, 4, (B, H, W)).astype(np.float32)
        k[:, H // 2:, : W // 2] = 0                      # a flat patch exactly at the black level
        raw = (site[None] + k * np.spacing(site)[None]).astype(np.float32)
        top = rng.integers(0, 4096, (B, H // 4, W)).astype(np.float32) / np.float32(4095)
        raw[:, : H // 4] = top
        return raw
    return (u16.astype(np.float32) / np.float32(4095)).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# torch semantics
# ----------------------------------------------------------------------------------------------
def raw2rgb(raw, black_level=None, reduce_size=True, out_channels=3):
    """pipeline_torch.py:240-283.  raw (B,H,W) -> (B,C,H|H/2,W|W/2), float32 like torch.zeros."""
    assert out_channels in [3, 4]
    if black_level is None:
        black_level = [0, 0, 0, 0]
    raw = np.asarray(raw)
    Bch, H, W = raw.shape
    dt = raw.dtype
    bl = np.asarray(black_level, dtype=dt)
    R = raw[:, 0::2, 0::2] - bl[0]
    G1 = raw[:, 0::2, 1::2] - bl[1]
    G2 = raw[:, 1::2, 0::2] - bl[2]
    Bl = raw[:, 1::2, 1::2] - bl[3]
    if reduce_size:
        rgb = np.zeros((Bch, out_channels, H // 2, W // 2), dtype=dt)
        if out_channels == 3:
            rgb[:, 0] = R
            rgb[:, 1] = (G1 + G2) / 2
            rgb[:, 2] = Bl
        else:
            rgb[:, 0], rgb[:, 1], rgb[:, 2], rgb[:, 3] = R, G1, G2, Bl
    else:
        rgb = np.zeros((Bch, out_channels, H, W), dtype=dt)
        if out_channels == 3:
            rgb[:, 0, 0::2, 0::2] = R
            rgb[:, 1, 0::2, 1::2] = G1
            rgb[:, 1, 1::2, 0::2] = G2
            rgb[:, 2, 1::2, 1::2] = Bl
        else:
            rgb[:, 0, 0::2, 0::2] = R
            rgb[:, 1, 0::2, 1::2] = G1
            rgb[:, 2, 1::2, 0::2] = G2
            rgb[:, 3, 1::2, 1::2] = Bl
    return rgb


def raw2rgb_vjp(g, H, W, reduce_size=True, out_channels=3):
    """VJP of raw2rgb: returns (grad_raw (B,H,W), grad_black_level (4,))."""
    Bch = g.shape[0]
    gr = np.zeros((Bch, H, W), dtype=g.dtype)
    if reduce_size:
        if out_channels == 3:
            gr[:, 0::2, 0::2] = g[:, 0]
            gr[:, 0::2, 1::2] = g[:, 1] / 2
            gr[:, 1::2, 0::2] = g[:, 1] / 2
            gr[:, 1::2, 1::2] = g[:, 2]
        else:
            gr[:, 0::2, 0::2], gr[:, 0::2, 1::2] = g[:, 0], g[:, 1]
            gr[:, 1::2, 0::2], gr[:, 1::2, 1::2] = g[:, 2], g[:, 3]
    else:
        if out_channels == 3:
            gr[:, 0::2, 0::2] = g[:, 0, 0::2, 0::2]
            gr[:, 0::2, 1::2] = g[:, 1, 0::2, 1::2]
            gr[:, 1::2, 0::2] = g[:, 1, 1::2, 0::2]
            gr[:, 1::2, 1::2] = g[:, 2, 1::2, 1::2]
        else:
            gr[:, 0::2, 0::2] = g[:, 0, 0::2, 0::2]
            gr[:, 0::2, 1::2] = g[:, 1, 0::2, 1::2]
            gr[:, 1::2, 0::2] = g[:, 2, 1::2, 0::2]
            gr[:, 1::2, 1::2] = g[:, 3, 1::2, 1::2]
    gbl = -np.array([gr[:, 0::2, 0::2].sum(dtype=np.float64), gr[:, 0::2, 1::2].sum(dtype=np.float64),
                     gr[:, 1::2, 0::2].sum(dtype=np.float64), gr[:, 1::2, 1::2].sum(dtype=np.float64)])
    return gr, gbl


def _pad(x, p, mode):
    """x (...,H,W).  mode 'mirror' = torch 'reflect' (c b | a b c); 'zero'; 'symmetric' = scipy
    'reflect' (b a | a b)."""
    if p == 0:
        return x
    width = [(0, 0)] * (x.ndim - 2) + [(p, p), (p, p)]
    if mode == 'mirror':
        return np.pad(x, width, mode='reflect')
    if mode == 'symmetric':
        return np.pad(x, width, mode='symmetric')
    if mode == 'zero':
        return np.pad(x, width, mode='constant')
    raise ValueError(mode)


def _pad_vjp(gp, p, mode):
    """adjoint of _pad: fold the padded border of gp back onto the interior."""
    if p == 0:
        return gp
    H = gp.shape[-2] - 2 * p
    W = gp.shape[-1] - 2 * p
    g = gp[..., p:p + H, :].copy()
    if mode != 'zero':
        for i in range(1, p + 1):
            if mode == 'mirror':      # padded row p-i  <- interior row i ; row p+H-1+i <- H-1-i
                g[..., i, :] += gp[..., p - i, :]
                g[..., H - 1 - i, :] += gp[..., p + H - 1 + i, :]
            else:                     # symmetric: padded row p-i <- interior row i-1
                g[..., i - 1, :] += gp[..., p - i, :]
                g[..., H - i, :] += gp[..., p + H - 1 + i, :]
    gp2 = g
    g = gp2[..., :, p:p + W].copy()
    if mode != 'zero':
        for i in range(1, p + 1):
            if mode == 'mirror':
                g[..., :, i] += gp2[..., :, p - i]
                g[..., :, W - 1 - i] += gp2[..., :, p + W - 1 + i]
            else:
                g[..., :, i - 1] += gp2[..., :, p - i]
                g[..., :, W - i] += gp2[..., :, p + W - 1 + i]
    return g


# test aid: accumulate the taps of conv2d in reverse order -- the same sums, rounded differently in float32 (a second,
# independent float32 evaluation of the reference's formulas: the control sample of tests/fuzz_gpu.py)
REVERSE_TAP_ORDER = False


def conv2d(x, w, pad_mode):
    """cross-correlation like nn.Conv2d(bias=False): x (B,Ci,H,W), w (Co,Ci,K,K), 'same' size."""
    K = w.shape[-1]
    p = K // 2
    xp = _pad(x, p, pad_mode)
    B, Ci, H, W = x.shape
    out = np.zeros((B, w.shape[0], H, W), dtype=x.dtype)
    taps = range(K - 1, -1, -1) if REVERSE_TAP_ORDER else range(K)
    for i in taps:
        for j in taps:
            patch = xp[:, :, i:i + H, j:j + W]                       # (B,Ci,H,W)
            out += np.einsum('bchw,kc->bkhw', patch, w[:, :, i, j])
    return out


def conv2d_vjp(x, w, pad_mode, g):
    """returns (grad_x, grad_w) for conv2d above."""
    K = w.shape[-1]
    p = K // 2
    xp = _pad(x, p, pad_mode)
    B, Ci, H, W = x.shape
    gxp = np.zeros_like(xp)
    gw = np.zeros(w.shape, dtype=np.float64)
    for i in range(K):
        for j in range(K):
            patch = xp[:, :, i:i + H, j:j + W]
            gw[:, :, i, j] = np.einsum('bkhw,bchw->kc', g.astype(np.float64), patch.astype(np.float64))
            gxp[:, :, i:i + H, j:j + W] += np.einsum('bkhw,kc->bchw', g, w[:, :, i, j])
    return _pad_vjp(gxp, p, pad_mode), gw.astype(x.dtype)


class IspParams:
    """The parameters and buffers of ParametrizedProcessing (pipeline_torch.py:152-173), by the
    reference's state_dict names."""

    NAMES = ('black_level', 'white_balance', 'colour_correction', 'gamma_correct',
             'debayer.weight', 'sharpening_filter.weight', 'gaussian_blur.weight')

    def __init__(self, camera_parameters=None, dtype=np.float32):
        if camera_parameters is None:
            camera_parameters = DEFAULT_CAMERA_PARAMS
        bl, wb, ccm = camera_parameters
        self.dtype = dtype
        self.black_level = np.asarray(bl, dtype=np.float32).astype(dtype)
        self.white_balance = np.asarray(wb, dtype=np.float32).reshape(1, 3).astype(dtype)
        self.colour_correction = np.asarray(ccm, dtype=np.float32).reshape(3, 3).astype(dtype)
        self.gamma_correct = np.asarray([2.2], dtype=np.float32).astype(dtype)
        deb = np.zeros((3, 3, 3, 3), dtype=np.float32)
        deb[0, 0], deb[1, 1], deb[2, 2] = K_RB, K_G, K_RB
        self.debayer = deb.astype(dtype)
        self.sharpening = K_SHARP.reshape(1, 1, 3, 3).astype(dtype)
        self.blur = K_BLUR.reshape(1, 1, 5, 5).astype(dtype)
        self.M_RGB_2_YUV = M_RGB_2_YUV.astype(dtype)
        self.M_YUV_2_RGB = M_YUV_2_RGB.astype(dtype)
        self.additive_layer = None

    def astype(self, dtype):
        o = IspParams.__new__(IspParams)
        o.dtype = dtype
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                setattr(o, k, v.astype(dtype))
            elif k != 'dtype':
                setattr(o, k, v)
        return o

    def by_name(self):
        d = {'black_level': self.black_level, 'white_balance': self.white_balance,
             'colour_correction': self.colour_correction, 'gamma_correct': self.gamma_correct,
             'debayer.weight': self.debayer, 'sharpening_filter.weight': self.sharpening,
             'gaussian_blur.weight': self.blur}
        if self.additive_layer is not None:
            d['additive_layer'] = self.additive_layer
        return d

    def perturb(self, seed, scale=0.05):
        """non-default weights so that zeros/symmetry cannot hide indexing bugs."""
        rng = np.random.default_rng(seed)
        for k, v in self.by_name().items():
            if k == 'additive_layer':
                v += (0.01 * rng.standard_normal(v.shape)).astype(v.dtype)
            elif k == 'gamma_correct':
                v += np.asarray(0.3 * rng.uniform(-1, 1), dtype=v.dtype)
            elif k == 'black_level':
                v += (0.01 * rng.uniform(-1, 1, v.shape)).astype(v.dtype)
            else:
                v += (scale * rng.standard_normal(v.shape)).astype(v.dtype)
        return self

    def pack(self):
        """the 150-float device block the C-ABI takes (include/r2l_isp.h: R2L_P_*)."""
        return np.concatenate([self.black_level.ravel(), self.white_balance.ravel(),
                               self.colour_correction.ravel(), self.gamma_correct.ravel(),
                               self.debayer.ravel(), self.sharpening.ravel(), self.blur.ravel(),
                               self.M_RGB_2_YUV.ravel(), self.M_YUV_2_RGB.ravel()]).astype(np.float32)


def _mix(x, M):
    """einsum('bchw,kc->bkhw') -- pipeline_torch.py:191,194,203."""
    return np.einsum('bchw,kc->bkhw', x, M).astype(x.dtype)


def parametrized_forward(raw, P, track_stages=False, bn=None):
    """ParametrizedProcessing.forward (pipeline_torch.py:175-225).

    bn: None (batch_norm_output=False) or dict(training=bool, running_mean, running_var,
        eps=1e-5, momentum=0.1); in training mode the dict's running stats are updated in place the
        way nn.BatchNorm2d does (biased var to normalise, unbiased into running_var).
    Returns (out, stages, cache)."""
    raw = np.asarray(raw)
    assert raw.ndim == 3, f"needs dims (B, H, W), got {raw.shape}"
    dt = P.dtype
    raw = raw.astype(dt)
    stages = {}
    c = {'raw': raw, 'track': track_stages}
    mosaic = raw2rgb(raw, P.black_level, reduce_size=False)                  # :183
    stages['demosaic'] = mosaic
    deb = conv2d(mosaic, P.debayer, 'mirror')                                # :187
    wbd = deb * P.white_balance.reshape(1, 3, 1, 1)                          # :190
    cc = _mix(wbd, P.colour_correction)                                      # :191
    stages['color_correct'] = cc
    yuv0 = _mix(cc, P.M_RGB_2_YUV)                                           # :194
    ysh = conv2d(yuv0[:, :1], P.sharpening, 'zero')                          # :195
    yuv1 = np.concatenate([ysh, yuv0[:, 1:]], axis=1)
    c.update(mosaic=mosaic, deb=deb, wbd=wbd, cc=cc, yuv0=yuv0, yuv1=yuv1)
    if track_stages:                                                         # :197-200
        rgb_sh = _mix(yuv1, P.M_YUV_2_RGB)
        stages['sharpening'] = rgb_sh
        yuv1b = _mix(rgb_sh, P.M_RGB_2_YUV)
        c.update(rgb_sh=rgb_sh, yuv1b=yuv1b)
    else:
        yuv1b = yuv1
    ybl = conv2d(yuv1b[:, :1], P.blur, 'mirror')                             # :202
    yuv2 = np.concatenate([ybl, yuv1b[:, 1:]], axis=1)
    rgb = _mix(yuv2, P.M_YUV_2_RGB)                                          # :203
    stages['gaussian'] = rgb
    clipped = np.clip(rgb, dt(1e-5), dt(1))                                  # :206
    stages['clipped'] = clipped
    gam = np.exp((dt(1) / P.gamma_correct) * np.log(clipped)).astype(dt)     # :209
    stages['gamma_correct'] = gam
    c.update(yuv1b=yuv1b, yuv2=yuv2, rgb=rgb, clipped=clipped, gam=gam)
    x = gam
    if P.additive_layer is not None:                                         # :212-214
        x = x + P.additive_layer
        stages['noise'] = x
    c['pre_bn'] = x
    if bn is not None:                                                       # :216-217
        eps = bn.get('eps', 1e-5)
        if bn['training']:
            n = x.shape[0] * x.shape[2] * x.shape[3]
            mean = x.mean(axis=(0, 2, 3), dtype=np.float64)
            var = x.astype(np.float64).var(axis=(0, 2, 3))
            m = bn.get('momentum', 0.1)
            bn['running_mean'][...] = (1 - m) * bn['running_mean'] + m * mean
            bn['running_var'][...] = (1 - m) * bn['running_var'] + m * var * n / max(n - 1, 1)
            bn['num_batches_tracked'] = bn.get('num_batches_tracked', 0) + 1
        else:
            mean = np.asarray(bn['running_mean'], dtype=np.float64)
            var = np.asarray(bn['running_var'], dtype=np.float64)
        istd = 1.0 / np.sqrt(var + eps)
        xhat = ((x - mean.reshape(1, 3, 1, 1)) * istd.reshape(1, 3, 1, 1)).astype(dt)
        c.update(bn_training=bn['training'], istd=istd, xhat=xhat)
        x = xhat
    c['has_bn'] = bn is not None
    return x, stages, c


def parametrized_backward(P, c, grad_out, stage_grads=False, clip_shift=0.0):
    """Reverse pass of parametrized_forward: one VJP per forward op, in reverse order (what autograd
    does for pipeline_torch.py:183-217).  Returns (param_grads by state_dict name, grad_raw,
    stage_grads dict or None).  Stage gradients are d loss / d stages[name] as retain_grad() would
    leave them in ``.grad`` (model.py:249-254).

    clip_shift: test aid.  torch.clip's gradient is a step function of the pre-clip value, so a pixel
    whose pre-clip value is within float32 round-off of 1e-5 or 1 may fall on either side.  clip_shift=d
    narrows (d>0) or widens (d<0) the pass band to [1e-5+d, 1-d]; the spread between the two bounds what
    such pixels can contribute to any gradient."""
    dt = P.dtype
    g = np.asarray(grad_out).astype(dt)
    sg = {}
    grads = {}
    if c['has_bn']:
        if c['bn_training']:
            xhat = c['xhat'].astype(np.float64)
            g64 = g.astype(np.float64)
            mg = g64.mean(axis=(0, 2, 3)).reshape(1, 3, 1, 1)
            mgx = (g64 * xhat).mean(axis=(0, 2, 3)).reshape(1, 3, 1, 1)
            g = ((g64 - mg - xhat * mgx) * c['istd'].reshape(1, 3, 1, 1)).astype(dt)
        else:
            g = (g * c['istd'].reshape(1, 3, 1, 1)).astype(dt)
    if P.additive_layer is not None:
        sg['noise'] = g
        grads['additive_layer'] = g.sum(axis=0, keepdims=True, dtype=np.float64).astype(dt)
    sg['gamma_correct'] = g
    # gam = exp(log(clipped)/gamma)
    inv = dt(1) / P.gamma_correct
    lg = np.log(c['clipped'])
    grads['gamma_correct'] = np.asarray(
        [-(g.astype(np.float64) * c['gam'] * lg).sum() / float(P.gamma_correct[0]) ** 2], dtype=dt)
    g = g * c['gam'] * inv / c['clipped']
    sg['clipped'] = g
    # torch.clip passes the gradient where min <= x <= max
    g = g * ((c['rgb'] >= dt(1e-5) + dt(clip_shift)) & (c['rgb'] <= dt(1) - dt(clip_shift))).astype(dt)
    sg['gaussian'] = g
    g_yuv2 = _mix(g, P.M_YUV_2_RGB.T)
    gy, gw = conv2d_vjp(c['yuv1b'][:, :1], P.blur, 'mirror', g_yuv2[:, :1])
    grads['gaussian_blur.weight'] = gw
    g_yuv1b = np.concatenate([gy, g_yuv2[:, 1:]], axis=1)
    if c['track']:
        g_rgb_sh = _mix(g_yuv1b, P.M_RGB_2_YUV.T)
        sg['sharpening'] = g_rgb_sh
        g_yuv1 = _mix(g_rgb_sh, P.M_YUV_2_RGB.T)
    else:
        g_yuv1 = g_yuv1b
    gy, gw = conv2d_vjp(c['yuv0'][:, :1], P.sharpening, 'zero', g_yuv1[:, :1])
    grads['sharpening_filter.weight'] = gw
    g_yuv0 = np.concatenate([gy, g_yuv1[:, 1:]], axis=1)
    g_cc = _mix(g_yuv0, P.M_RGB_2_YUV.T)
    sg['color_correct'] = g_cc
    grads['colour_correction'] = np.einsum('bkhw,bchw->kc', g_cc.astype(np.float64),
                                           c['wbd'].astype(np.float64)).astype(dt)
    g_wbd = _mix(g_cc, P.colour_correction.T)
    grads['white_balance'] = (g_wbd.astype(np.float64) * c['deb']).sum(axis=(0, 2, 3)).reshape(1, 3).astype(dt)
    g_deb = g_wbd * P.white_balance.reshape(1, 3, 1, 1)
    g_mosaic, gw = conv2d_vjp(c['mosaic'], P.debayer, 'mirror', g_deb)
    grads['debayer.weight'] = gw
    sg['demosaic'] = g_mosaic
    H, W = c['raw'].shape[1:]
    g_raw, gbl = raw2rgb_vjp(g_mosaic, H, W, reduce_size=False, out_channels=3)
    grads['black_level'] = gbl.astype(dt)
    return grads, g_raw, (sg if stage_grads else None)


# ----------------------------------------------------------------------------------------------
# numpy ("static") semantics
# ----------------------------------------------------------------------------------------------
def masks_CFA_Bayer(shape):
    """RGGB masks.  colour-demosaicing 0.1.6 `masks_CFA_Bayer(shape, 'RGGB')` [published algorithm]."""
    H, W = shape
    R = np.zeros(shape, dtype=bool)
    G = np.zeros(shape, dtype=bool)
    B = np.zeros(shape, dtype=bool)
    R[0::2, 0::2] = True
    G[0::2, 1::2] = True
    G[1::2, 0::2] = True
    B[1::2, 1::2] = True
    return R, G, B


def demosaicing_CFA_Bayer_bilinear(CFA):
    """colour-demosaicing 0.1.6 [published algorithm; UNPINNED, see module header]:
    R,G,B = scipy.ndimage.convolve(CFA*mask_c, H_c) with default mode='reflect', stacked HWC f64."""
    from scipy.ndimage import convolve
    CFA = np.asarray(CFA, dtype=np.float64)
    R_m, G_m, B_m = masks_CFA_Bayer(CFA.shape)
    H_G = np.array([[0, 1, 0], [1, 4, 1], [0, 1, 0]], dtype=np.float64) / 4
    H_RB = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], dtype=np.float64) / 4
    R = convolve(CFA * R_m, H_RB)
    G = convolve(CFA * G_m, H_G)
    B = convolve(CFA * B_m, H_RB)
    return np.stack([R, G, B], axis=-1)


MALVAR_GR_GB = np.array([[0, 0, -1, 0, 0],
                         [0, 0, 2, 0, 0],
                         [-1, 2, 4, 2, -1],
                         [0, 0, 2, 0, 0],
                         [0, 0, -1, 0, 0]], dtype=np.float64) / 8
MALVAR_Rg_RB_Bg_BR = np.array([[0, 0, 0.5, 0, 0],
                               [0, -1, 0, -1, 0],
                               [-1, 4, 5, 4, -1],
                               [0, -1, 0, -1, 0],
                               [0, 0, 0.5, 0, 0]], dtype=np.float64) / 8
MALVAR_Rg_BR_Bg_RB = MALVAR_Rg_RB_Bg_BR.T.copy()
MALVAR_Rb_BB_Br_RR = np.array([[0, 0, -1.5, 0, 0],
                               [0, 2, 0, 2, 0],
                               [-1.5, 0, 6, 0, -1.5],
                               [0, 2, 0, 2, 0],
                               [0, 0, -1.5, 0, 0]], dtype=np.float64) / 8


def demosaicing_CFA_Bayer_Malvar2004(CFA):
    """colour-demosaicing 0.1.6 [published algorithm (Malvar, He, Cutler 2004); UNPINNED]:
    four 5x5 kernels applied to the unmasked CFA with scipy 'reflect', selected per site."""
    from scipy.ndimage import convolve
    CFA = np.asarray(CFA, dtype=np.float64)
    R_m, G_m, B_m = masks_CFA_Bayer(CFA.shape)
    R = CFA * R_m
    G = CFA * G_m
    B = CFA * B_m
    G = np.where(np.logical_or(R_m, B_m), convolve(CFA, MALVAR_GR_GB), G)
    RBg_RBBR = convolve(CFA, MALVAR_Rg_RB_Bg_BR)
    RBg_BRRB = convolve(CFA, MALVAR_Rg_BR_Bg_RB)
    RBgr_BBRR = convolve(CFA, MALVAR_Rb_BB_Br_RR)
    R_r = np.any(R_m, axis=1)[:, None] * np.ones(R.shape, dtype=bool)
    R_c = np.any(R_m, axis=0)[None, :] * np.ones(R.shape, dtype=bool)
    B_r = np.any(B_m, axis=1)[:, None] * np.ones(B.shape, dtype=bool)
    B_c = np.any(B_m, axis=0)[None, :] * np.ones(B.shape, dtype=bool)
    R = np.where(np.logical_and(R_r, B_c), RBg_RBBR, R)
    R = np.where(np.logical_and(B_r, R_c), RBg_BRRB, R)
    B = np.where(np.logical_and(B_r, R_c), RBg_RBBR, B)
    B = np.where(np.logical_and(R_r, B_c), RBg_BRRB, B)
    R = np.where(np.logical_and(B_r, B_c), RBgr_BBRR, R)
    B = np.where(np.logical_and(R_r, R_c), RBgr_BBRR, B)
    return np.stack([R, G, B], axis=-1)


def _cnv_h(x, y):
    """colour-demosaicing 0.1.6 `_cnv_h`: scipy.ndimage convolve1d along the columns, mode='mirror' (d c b | a b c d | c b a)"""
    from scipy.ndimage import convolve1d
    return convolve1d(x, y, mode='mirror')


def _cnv_v(x, y):
    from scipy.ndimage import convolve1d
    return convolve1d(x, y, mode='mirror', axis=0)


def demosaicing_CFA_Bayer_Menon2007(CFA, refining_step=True):
    """colour-demosaicing 0.1.6 `demosaicing_CFA_Bayer_Menon2007(CFA, 'RGGB', refining_step=True)` [published algorithm:
    Menon, Andriani, Calvagno, "Demosaicing With Directional Filtering and a posteriori Decision", IEEE TIP 2007 (DDFAPD);
    restated from the package's published source; UNPINNED like the other two demosaics -- the package is neither vendored in
    the reference tree nor installed here, and nothing in the reference pins its output].  Call site: pipeline_numpy.py:96-97.

    1. green by directional 5-tap filters, horizontally (G_H) and vertically (G_V), at the red / blue sites;
    2. chrominance differences C_H / C_V at those sites, their gradients D_H / D_V two samples ahead, summed over the 5x5
       kernel k (and its transpose) into the classifiers d_H / d_V; the direction with the SMALLER classifier wins (M = 1:
       horizontal) -- the a-posteriori decision;
    3. red / blue at the green sites from the two neighbours of their row resp. column (bilinear on the colour difference),
       red at the blue sites (and vice versa) along the decided direction;
    4. the refining step: green at the red / blue sites from the 3-tap mean of R - G (B - G) along the decided direction, then
       red / blue at the green sites and at the opposite sites again from the refined differences."""
    from scipy.ndimage import convolve
    CFA = np.asarray(CFA, dtype=np.float64)
    Rm, Gm, Bm = masks_CFA_Bayer(CFA.shape)
    h_0 = np.array([0, 0.5, 0, 0.5, 0])
    h_1 = np.array([-0.25, 0, 0.5, 0, -0.25])
    R, G, B = CFA * Rm, CFA * Gm, CFA * Bm
    G_H = np.where(Gm == 0, _cnv_h(CFA, h_0) + _cnv_h(CFA, h_1), G)
    G_V = np.where(Gm == 0, _cnv_v(CFA, h_0) + _cnv_v(CFA, h_1), G)
    C_H = np.where(Rm == 1, R - G_H, 0)
    C_H = np.where(Bm == 1, B - G_H, C_H)
    C_V = np.where(Rm == 1, R - G_V, 0)
    C_V = np.where(Bm == 1, B - G_V, C_V)
    D_H = np.abs(C_H - np.pad(C_H, ((0, 0), (0, 2)), mode='reflect')[:, 2:])
    D_V = np.abs(C_V - np.pad(C_V, ((0, 2), (0, 0)), mode='reflect')[2:, :])
    k = np.array([[0, 0, 1, 0, 1],
                  [0, 0, 0, 1, 0],
                  [0, 0, 3, 0, 3],
                  [0, 0, 0, 1, 0],
                  [0, 0, 1, 0, 1]], dtype=np.float64)
    d_H = convolve(D_H, k, mode='constant')
    d_V = convolve(D_V, np.transpose(k), mode='constant')
    mask = d_V >= d_H
    G = np.where(mask, G_H, G_V)
    M = np.where(mask, 1, 0)
    R_r = np.transpose(np.any(Rm == 1, axis=1)[np.newaxis]) * np.ones(R.shape)       # red rows
    B_r = np.transpose(np.any(Bm == 1, axis=1)[np.newaxis]) * np.ones(B.shape)       # blue rows
    k_b = np.array([0.5, 0, 0.5])
    R = np.where(np.logical_and(Gm == 1, R_r == 1), G + _cnv_h(R, k_b) - _cnv_h(G, k_b), R)
    R = np.where(np.logical_and(Gm == 1, B_r == 1) == 1, G + _cnv_v(R, k_b) - _cnv_v(G, k_b), R)
    B = np.where(np.logical_and(Gm == 1, B_r == 1), G + _cnv_h(B, k_b) - _cnv_h(G, k_b), B)
    B = np.where(np.logical_and(Gm == 1, R_r == 1) == 1, G + _cnv_v(B, k_b) - _cnv_v(G, k_b), B)
    R = np.where(np.logical_and(B_r == 1, Bm == 1),
                 np.where(M == 1, B + _cnv_h(R, k_b) - _cnv_h(B, k_b), B + _cnv_v(R, k_b) - _cnv_v(B, k_b)), R)
    B = np.where(np.logical_and(R_r == 1, Rm == 1),
                 np.where(M == 1, R + _cnv_h(B, k_b) - _cnv_h(R, k_b), R + _cnv_v(B, k_b) - _cnv_v(R, k_b)), B)
    if refining_step:
        R, G, B = _refining_step_Menon2007(R, G, B, Rm, Gm, Bm, M)
    return np.stack([R, G, B], axis=-1)


def _refining_step_Menon2007(R, G, B, Rm, Gm, Bm, M):
    """colour-demosaicing 0.1.6 `refining_step_Menon2007` [published algorithm; UNPINNED]"""
    M = np.asarray(M, dtype=np.float64)
    # green at the red / blue sites
    R_G, B_G = R - G, B - G
    FIR = np.ones(3) / 3
    B_G_m = np.where(Bm == 1, np.where(M == 1, _cnv_h(B_G, FIR), _cnv_v(B_G, FIR)), 0)
    R_G_m = np.where(Rm == 1, np.where(M == 1, _cnv_h(R_G, FIR), _cnv_v(R_G, FIR)), 0)
    G = np.where(Rm == 1, R - R_G_m, G)
    G = np.where(Bm == 1, B - B_G_m, G)
    # red / blue at the green sites
    R_r = np.transpose(np.any(Rm == 1, axis=1)[np.newaxis]) * np.ones(R.shape)
    R_c = np.any(Rm == 1, axis=0)[np.newaxis] * np.ones(R.shape)
    B_r = np.transpose(np.any(Bm == 1, axis=1)[np.newaxis]) * np.ones(B.shape)
    B_c = np.any(Bm == 1, axis=0)[np.newaxis] * np.ones(B.shape)
    R_G, B_G = R - G, B - G
    k_b = np.array([0.5, 0, 0.5])
    R_G_m = np.where(np.logical_and(Gm == 1, B_r == 1), _cnv_v(R_G, k_b), R_G_m)
    R = np.where(np.logical_and(Gm == 1, B_r == 1), G + R_G_m, R)
    R_G_m = np.where(np.logical_and(Gm == 1, B_c == 1), _cnv_h(R_G, k_b), R_G_m)
    R = np.where(np.logical_and(Gm == 1, B_c == 1), G + R_G_m, R)
    B_G_m = np.where(np.logical_and(Gm == 1, R_r == 1), _cnv_v(B_G, k_b), B_G_m)
    B = np.where(np.logical_and(Gm == 1, R_r == 1), G + B_G_m, B)
    B_G_m = np.where(np.logical_and(Gm == 1, R_c == 1), _cnv_h(B_G, k_b), B_G_m)
    B = np.where(np.logical_and(Gm == 1, R_c == 1), G + B_G_m, B)
    # red at the blue sites, blue at the red sites
    R_B = R - B
    R_B_m = np.where(Bm == 1, np.where(M == 1, _cnv_h(R_B, FIR), _cnv_v(R_B, FIR)), 0)
    R = np.where(Bm == 1, B + R_B_m, R)
    R_B_m = np.where(Rm == 1, np.where(M == 1, _cnv_h(R_B, FIR), _cnv_v(R_B, FIR)), 0)
    B = np.where(Rm == 1, R - R_B_m, B)
    return R, G, B


YUV_FROM_RGB = np.array([[0.299, 0.587, 0.114],
                         [-0.14714119, -0.28886916, 0.43601035],
                         [0.61497538, -0.51496512, -0.10001026]], dtype=np.float64)
RGB_FROM_YUV = np.linalg.inv(YUV_FROM_RGB)


def rgb2yuv(img):
    """scikit-image 0.18.1 `rgb2yuv` [published algorithm]: arr @ yuv_from_rgb.T."""
    return np.asarray(img, dtype=np.float64) @ YUV_FROM_RGB.T


def yuv2rgb(img):
    """scikit-image 0.18.1 `yuv2rgb`: arr @ inv(yuv_from_rgb).T (no clipping)."""
    return np.asarray(img, dtype=np.float64) @ RGB_FROM_YUV.T


def remove_blacklv(rawImg, black_level):
    """pipeline_numpy.py:152-158 -- in place, in the input's dtype."""
    rawImg[0::2, 0::2] -= black_level[0]
    rawImg[0::2, 1::2] -= black_level[1]
    rawImg[1::2, 0::2] -= black_level[2]
    rawImg[1::2, 1::2] -= black_level[3]
    return rawImg


def sharpening_filter(image, kernel=np.array([[0, -1, 0], [-1, 5, -1], [0, -1, 0]])):
    """pipeline_numpy.py:180-191."""
    from scipy.signal import convolve2d
    img_yuv = rgb2yuv(image)
    img_yuv[:, :, 0] = convolve2d(img_yuv[:, :, 0], kernel, 'same', boundary='fill', fillvalue=0)
    return yuv2rgb(img_yuv)


def unsharp_mask(image, radius=1.0, amount=1.0, multichannel=False, preserve_range=False):
    """scikit-image 0.18.1 `skimage.filters.unsharp_mask` restated (the package is absent from the image and
    from the reference tree: PARITY UNPINNED): result = image + (image - gaussian(image, sigma=radius,
    mode='reflect')) * amount, the Gaussian being scipy.ndimage.gaussian_filter with truncate=4.0; with
    multichannel=True the filter runs separately on every slice along the LAST axis; preserve_range=False clips
    to [0,1] / [-1,1].  The reference calls it with multichannel=True on the 2-D luma plane
    (pipeline_numpy.py:117, :170-177), so each image COLUMN is a "channel" and is blurred along the rows only:
    a vertical 9-tap unsharp mask."""
    from scipy import ndimage
    fimg = np.asarray(image, dtype=float)
    vrange = None
    if not preserve_range:
        vrange = (-1.0, 1.0) if np.any(fimg < 0) else (0.0, 1.0)

    def single(ch):
        blurred = ndimage.gaussian_filter(ch, radius, mode='reflect', truncate=4.0)
        res = ch + (ch - blurred) * amount
        return np.clip(res, *vrange) if vrange is not None else res
    if multichannel:
        out = np.empty_like(fimg)
        for c in range(fimg.shape[-1]):
            out[..., c] = single(fimg[..., c])
        return out
    return single(fimg)


def unsharp_masking(img, radius=1.0, amount=1.0, multichannel=True, preserve_range=True):
    """pipeline_numpy.py:170-177 as processing() calls it (:117: multichannel=True)."""
    img = rgb2yuv(img)
    img[:, :, 0] = unsharp_mask(img[:, :, 0], radius=radius, amount=amount, multichannel=multichannel,
                                preserve_range=preserve_range)
    return yuv2rgb(img)


def gaussian_denoising(img, sigma=0.5):
    """pipeline_numpy.py:203-209."""
    from scipy import ndimage
    img = rgb2yuv(img)
    img[:, :, 0] = ndimage.gaussian_filter(img[:, :, 0], sigma)
    return yuv2rgb(img)


def median_denoising(img, size=3):
    """pipeline_numpy.py:194-200."""
    from scipy import ndimage
    img = rgb2yuv(img)
    img[:, :, 0] = ndimage.median_filter(img[:, :, 0], size)
    return yuv2rgb(img)


def fft_denoising(img, keep_fraction=0.3):
    """pipeline_numpy.py:212-238 as processing() calls it (:121-122: row_cut=False, column_cut=True).
    scipy.fftpack.fft2 transforms the LAST two axes of the (H,W,3) array, i.e. (W, channel); zeroing the
    slice [:, int(c*keep):int(c*(1-keep))] (c = W) along axis 1 for every channel frequency and inverting leaves
    the channel transform undone: per image row and colour channel a 1-D ideal low-pass along the columns,
    real part kept.  Restated on axis 1 only (equal to the 2-D form up to float64 round-off)."""
    r, c, _ = img.shape
    f = np.fft.fft(img, axis=1)
    f[:, int(c * keep_fraction):int(c * (1 - keep_fraction))] = 0
    return np.fft.ifft(f, axis=1).real


def gaussian_kernel1d(sigma=0.5, truncate=4.0):
    """the taps scipy.ndimage.gaussian_filter uses: radius=int(truncate*sigma+0.5)."""
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1, dtype=np.float64)
    w = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return w / w.sum()


def processing(img, black_level, white_balance, colour_matrix, debayer="bilinear",
               sharpening="unsharp_masking", denoising="median_filter", gamma=2.2, sharp_radius=1.0, sharp_amount=1.0,
               median_kernel_size=3, gaussian_sigma=0.5, fft_fraction=0.3):
    """pipeline_numpy.py:70-141, for the branches in scope (unknown strings are silently ignored,
    exactly like the reference's if-chains).  img (H,W) float -- MODIFIED IN PLACE by the black
    level step like the reference.  Returns (H,W,3) float64."""
    img = remove_blacklv(img, black_level)
    if debayer == "bilinear":
        img = demosaicing_CFA_Bayer_bilinear(img)
    if debayer == "malvar2004":
        img = demosaicing_CFA_Bayer_Malvar2004(img)
    if debayer == "menon2007":
        img = demosaicing_CFA_Bayer_Menon2007(img)                                                    # :96-97
    img = img * white_balance
    img = np.einsum('ijk,lk->ijl', img, np.array(colour_matrix).reshape(3, 3))
    if sharpening == "sharpening_filter":
        img = sharpening_filter(img)
    if sharpening == "unsharp_masking":
        img = unsharp_masking(img, radius=sharp_radius, amount=sharp_amount, multichannel=True)       # :117
    if denoising == "median_denoising":
        img = median_denoising(img, size=median_kernel_size)                                          # :119
    if denoising == "gaussian_denoising":
        img = gaussian_denoising(img, sigma=gaussian_sigma)                                           # :120
    if denoising == "fft_denoising":
        img = fft_denoising(img, keep_fraction=fft_fraction)                                          # :121-122
    img = np.clip(img, 0, 1)
    img = img ** (1.0 / gamma)
    return img


def static_batch(raw, camera_parameters, debayer='bilinear', sharpening='sharpening_filter',
                 denoising='gaussian_denoising', gamma=2.2, **options):
    """RawProcessingPipeline.__call__ (pipeline_numpy.py:55-67) over a batch: (B,H,W) -> (B,3,H,W)
    float32.  Frames are processed in the dtype they come in, like the reference: remove_blacklv works in
    place, so float32 frames (load_image's np.float32 array / (2**bits-1), utils/dataset_utils.py:18-26,
    dataset.py:86-87) get a float32 subtraction of the float32-rounded black level, float64 frames (a DNG's
    uint16 / (2**bits-1)) a float64 one; everything from the demosaic on is float64."""
    bl, wb, ccm = camera_parameters
    out = []
    raw = np.asarray(raw)
    assert raw.dtype in (np.float32, np.float64), raw.dtype
    for img in raw:
        o = processing(img.copy(), bl, wb, ccm, debayer=debayer,
                       sharpening=sharpening, denoising=denoising, gamma=gamma, **options)
        out.append(o.transpose(2, 0, 1).astype(np.float32))
    return np.stack(out)


# --------------------------------------------------------------------------------------------------------
# adversarial auxiliary losses: utils/ssim.py:9-39, utils/base.py:342-358
# --------------------------------------------------------------------------------------------------------
def ssim_window(window_size=11, sigma=1.5, dtype=np.float32):
    """utils/ssim.py:9-17: normalised 1-D Gaussian (float32) and its outer product."""
    g = np.array([np.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)],
                 dtype=np.float32)
    g = (g / g.sum(dtype=np.float32)).astype(np.float32)
    return g.astype(dtype), np.outer(g, g).astype(np.float32).astype(dtype)


def _ssim_blur(x, w2):
    """depthwise F.conv2d(x, window, padding=window_size//2) (zero padding), x (B,C,H,W)."""
    K = w2.shape[0]
    p = K // 2
    xp = np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)))
    H, W = x.shape[2:]
    out = np.zeros_like(x)
    for i in range(K):
        for j in range(K):
            out += w2[i, j] * xp[:, :, i:i + H, j:j + W]
    return out


def ssim(img1, img2, window_size=11, dtype=np.float64):
    """utils/ssim.py:19-39 with size_average=True: returns (mean SSIM, d mean / d img2)."""
    x = np.asarray(img1, dtype=dtype)
    y = np.asarray(img2, dtype=dtype)
    _, w2 = ssim_window(window_size, dtype=dtype)
    C1, C2 = dtype(0.01 ** 2), dtype(0.03 ** 2)
    mu1, mu2 = _ssim_blur(x, w2), _ssim_blur(y, w2)
    s11 = _ssim_blur(x * x, w2) - mu1 * mu1
    s22 = _ssim_blur(y * y, w2) - mu2 * mu2
    s12 = _ssim_blur(x * y, w2) - mu1 * mu2
    A1, A2 = 2 * mu1 * mu2 + C1, 2 * s12 + C2
    B1, B2 = mu1 * mu1 + mu2 * mu2 + C1, s11 + s22 + C2
    S = A1 * A2 / (B1 * B2)
    n = S.size
    # reverse pass w.r.t. img2 through (mu2, E[y^2], E[xy]); the window is symmetric, so the adjoint of the
    # zero-padded correlation is the same correlation
    dA1, dA2, dB1, dB2 = A2 / (B1 * B2), A1 / (B1 * B2), -S / B1, -S / B2
    d_mu2 = dA1 * 2 * mu1 + dA2 * (-2 * mu1) + dB1 * 2 * mu2 + dB2 * (-2 * mu2)
    d_eyy = dB2                      # s22 = E[y^2] - mu2^2
    d_exy = dA2 * 2                  # s12 = E[xy] - mu1 mu2
    grad = (_ssim_blur(d_mu2, w2) + 2 * y * _ssim_blur(d_eyy, w2) + x * _ssim_blur(d_exy, w2)) / n
    return S.mean(), grad


def l2_regularization(x, y):
    """utils/base.py:342-343: ((x - y) ** 2).sum(); returns (value, d value / d y)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    return ((x - y) ** 2).sum(), 2 * (y - x)


# --------------------------------------------------------------------------------------------------------
# additive Gaussian noise with a counter-based generator
# --------------------------------------------------------------------------------------------------------
def philox4x32_10(c, k):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011: "Parallel random numbers: as easy as 1, 2, 3"), vectorised:
    c (n,4) uint32 counters, k (2,) uint32 key -> (n,4) uint32.  Published algorithm; the known-answer vectors of
    the Random123 distribution are checked in tests/test_oracle_golden.py."""
    c = np.array(c, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    k0, k1 = np.uint64(int(k[0])), np.uint64(int(k[1]))
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c[:, 0]
        p1 = M1 * c[:, 2]
        c = np.stack([((p1 >> np.uint64(32)) ^ c[:, 1] ^ k0) & mask, p1 & mask,
                      ((p0 >> np.uint64(32)) ^ c[:, 3] ^ k1) & mask, p0 & mask], axis=1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c.astype(np.uint32)


def philox_normal(n, seed, offset=0):
    """the N(0,1) deviates of r2l_add_noise_philox for elements 0..n-1: counter = (i // 4 as 64 bits, offset as 64
    bits), key = seed; Box-Muller in float32 on output pairs (0,1) and (2,3)."""
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    c = np.stack([g & np.uint64(0xFFFFFFFF), g >> np.uint64(32),
                  np.full_like(g, int(offset) & 0xFFFFFFFF), np.full_like(g, (int(offset) >> 32) & 0xFFFFFFFF)], axis=1)
    o = philox4x32_10(c, (int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF))
    s = np.float32(2.3283064365386963e-10)
    out = np.empty((len(g), 4), dtype=np.float32)
    for h in (0, 1):
        u1 = (o[:, 2 * h].astype(np.float32) + np.float32(1)) * s
        u2 = o[:, 2 * h + 1].astype(np.float32) * s
        r = np.sqrt(np.float32(-2) * np.log(u1))
        t = np.float32(6.2831853071795865) * u2
        out[:, 2 * h] = r * np.cos(t)
        out[:, 2 * h + 1] = r * np.sin(t)
    return out.reshape(-1)[:n]
