"""Drop-in for the reference's ``processing/pipeline_numpy.py`` (static pipeline) on MI355X.

``processing(img, black_level, white_balance, colour_matrix, debayer=..., sharpening=..., denoising=...)``
and ``RawProcessingPipeline`` keep the reference's signatures and return types
(/root/reference/processing/pipeline_numpy.py:36-141) but run the batched static kernel of libr2l_isp.so
(float64 linear part, numpy semantics: symmetric borders, clip to [0,1], gamma 2.2).

The reference applies this per image inside DataLoader worker processes (train.py:164-171, :318); a GPU
context cannot live in forked workers, so the batched form ``StaticProcessing`` (an nn.Module to use as
the ``processor`` on device batches) is the fast path, and the per-image callables below are the
API-compatible wrappers for the main process (figures/ABtesting.py:135-173, app.py:13-35)."""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as F_


def _device():
    if torch.cuda.is_available():
        return torch.device('cuda')
    return torch.device('cpu')      # only usable under the test emulation; _lib raises otherwise


def processing(img, black_level, white_balance, colour_matrix, debayer="bilinear",
               sharpening="unsharp_masking", sharp_radius=1.0, sharp_amount=1.0,
               denoising="median_filter", median_kernel_size=3, gaussian_sigma=0.5, fft_fraction=0.3,
               weight_chambolle=0.01, weight_bregman=100, sigma_bilateral=0.6, gamma=2.2, bits=16):
    """reference :70-141.  img (H,W) float ndarray -> (H,W,3) float64 ndarray.

    As in the reference, `img` has its black level removed IN PLACE (:152-158), in the array's own dtype:
    a float32 frame (what the reference's datasets hand over, dataset.py:86-87; docstring :57) gets the
    float32 subtraction, a float64 frame the float64 one -- on the device exactly as in the caller's array.
    Option strings that name no algorithm skip their stage (the signature default denoising="median_filter"
    is one).  sharp_radius / sharp_amount / median_kernel_size / gaussian_sigma / fft_fraction (:117-122) are launch
    arguments of the kernels, within what their windows hold (functional.static_pipeline: gaussian_sigma < 0.625,
    sharp_radius < 1.125, median_kernel_size 3 or 5, fft_fraction in [0, 0.5]; anything else raises R2LError with the reason --
    INTEGRATION.md); weight_chambolle / weight_bregman / sigma_bilateral belong to denoisers this library does not build."""
    if not (isinstance(img, np.ndarray) and img.dtype in (np.float32, np.float64)):
        raise TypeError('processing() takes a float32 or float64 ndarray (dataset.py:86-87 delivers float32)')
    raw = torch.from_numpy(np.ascontiguousarray(img))[None].to(_device())
    out = F_.static_pipeline(raw, (black_level, white_balance, colour_matrix), debayer=debayer,
                             sharpening=sharpening, denoising=denoising, gamma=gamma, sharp_radius=sharp_radius,
                             sharp_amount=sharp_amount, median_kernel_size=median_kernel_size,
                             gaussian_sigma=gaussian_sigma, fft_fraction=fft_fraction)
    img[0::2, 0::2] -= black_level[0]      # side effect of remove_blacklv on the caller's array
    img[0::2, 1::2] -= black_level[1]
    img[1::2, 0::2] -= black_level[2]
    img[1::2, 1::2] -= black_level[3]
    return out[0].permute(1, 2, 0).cpu().numpy().astype(np.float64)


class RawProcessingPipeline(object):
    """reference :36-67: callable transform, (H,W) ndarray -> (3,H,W) float32 tensor."""

    def __init__(self, camera_parameters, debayer='bilinear', sharpening='unsharp_masking', denoising='gaussian'):
        self.camera_parameters = camera_parameters

        self.debayer = debayer
        self.sharpening = sharpening
        self.denoising = denoising

    def __call__(self, img):
        black_level, white_balance, colour_matrix = self.camera_parameters
        img = processing(img, black_level, white_balance, colour_matrix,
                         debayer=self.debayer, sharpening=self.sharpening, denoising=self.denoising)
        img = img.transpose(2, 0, 1)

        return torch.Tensor(img)


class StaticProcessing(nn.Module):
    """Batched static pipeline as a ``processor`` module: (B,H,W) raw on the GPU -> (B,3,H,W).

    Equals RawProcessingPipeline applied to every frame followed by the optional T.Normalize(mean, std)
    of train.py:157-171 (fused into the pipeline's kernels: r2l_static_fwd_norm).  No trainable parameters, no gradient.  Frames may be float32 in [0,1] or the
    sensor's 16-bit containers (uint16 / int16 tensors, divided by 2**raw_bits - 1 inside the kernel)."""

    raw_bits = 16

    def __init__(self, camera_parameters, debayer='bilinear', sharpening='sharpening_filter',
                 denoising='gaussian_denoising', gamma=2.2, mean=None, std=None, **options):
        super().__init__()
        unknown = set(options) - set(F_.STATIC_OPTION_DEFAULTS)
        if unknown:
            raise TypeError(f'unknown static options {sorted(unknown)} (have: {sorted(F_.STATIC_OPTION_DEFAULTS)})')
        self.options = dict(options)          # processing()'s numeric arguments (pipeline_numpy.py:70-73), see static_pipeline
        self.camera_parameters = tuple(list(map(float, p)) for p in camera_parameters)
        self.debayer = debayer
        self.sharpening = sharpening
        self.denoising = denoising
        self.gamma = gamma
        self.stages = None
        self.buffer = None
        if mean is not None:
            self.register_buffer('mean_std', torch.cat([torch.as_tensor(mean, dtype=torch.float32).reshape(3),
                                                        torch.as_tensor(std, dtype=torch.float32).reshape(3)]))
        else:
            self.mean_std = None

    @torch.no_grad()
    def forward(self, raw):
        assert raw.ndim == 3, f"needs dims (B, H, W), got {raw.shape}"
        self.stages = {}
        self.buffer = {}
        rgb = F_.static_pipeline(raw, self.camera_parameters, self.debayer, self.sharpening,
                                 self.denoising, self.gamma, bits=self.raw_bits, mean_std=self._mean_std_host(),
                                 **getattr(self, 'options', {}))
        self.buffer['processed_rgb'] = rgb
        return rgb

    def _mean_std_host(self):
        """the six floats of the mean_std buffer on the host (the kernels take them as launch arguments), read
        back once per value of the buffer"""
        ms = self.mean_std
        if ms is None:
            return None
        key = (ms.data_ptr(), ms._version)
        cached = self.__dict__.get('_ms_host')
        if cached is None or cached[0] != key:
            cached = (key, [float(v) for v in ms.detach().cpu().tolist()])
            self.__dict__['_ms_host'] = cached
        return cached[1]
