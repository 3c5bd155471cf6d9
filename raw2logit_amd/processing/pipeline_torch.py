"""Drop-in for the reference's ``processing/pipeline_torch.py`` on MI355X.

Same names, constructor arguments, parameter / buffer names (state_dict keys), attributes read by the
callers (``stages``, ``buffer``, ``track_stages``, ``additive_layer``) and error behaviour as
/root/reference/processing/pipeline_torch.py, so that ``train.py`` (:194-203) and ``model.py`` (:77-83)
use it unchanged -- but ``forward`` enqueues hand-written gfx950 kernels through libr2l_isp.so instead
of ~20 ATen ops.  Modules stay deep-copyable and picklable (train.py:248, utils/base.py:249-264): they
hold tensors only; the shared library is loaded lazily at module level."""
import torch
import torch.nn as nn

from .. import functional as F_
from ..functional import raw2rgb  # noqa: F401  (re-exported: reference :240-283)

# constants, reference :13-40
K_G = torch.Tensor([[0, 1, 0],
                    [1, 4, 1],
                    [0, 1, 0]]) / 4

K_RB = torch.Tensor([[1, 2, 1],
                     [2, 4, 2],
                     [1, 2, 1]]) / 4

M_RGB_2_YUV = torch.Tensor([[0.299, 0.587, 0.114],
                            [-0.14714119, -0.28886916, 0.43601035],
                            [0.61497538, -0.51496512, -0.10001026]])
M_YUV_2_RGB = torch.Tensor([[1.0000000000e+00, -4.1827794561e-09, 1.1398830414e+00],
                            [1.0000000000e+00, -3.9464232326e-01, -5.8062183857e-01],
                            [1.0000000000e+00, 2.0320618153e+00, -1.2232658220e-09]])

K_BLUR = torch.Tensor([[6.9625e-08, 2.8089e-05, 2.0755e-04, 2.8089e-05, 6.9625e-08],
                       [2.8089e-05, 1.1332e-02, 8.3731e-02, 1.1332e-02, 2.8089e-05],
                       [2.0755e-04, 8.3731e-02, 6.1869e-01, 8.3731e-02, 2.0755e-04],
                       [2.8089e-05, 1.1332e-02, 8.3731e-02, 1.1332e-02, 2.8089e-05],
                       [6.9625e-08, 2.8089e-05, 2.0755e-04, 2.8089e-05, 6.9625e-08]])
K_SHARP = torch.Tensor([[0, -1, 0],
                        [-1, 5, -1],
                        [0, -1, 0]])
DEFAULT_CAMERA_PARAMS = (
    [0., 0., 0., 0.],
    [1., 1., 1.],
    [1., 0., 0., 0., 1., 0., 0., 0., 1.],
)


class RawToRGB(nn.Module):
    """reference :43-80 -- raw (B,H,W) -> packed / zero-filled RGB via the raw2rgb kernel.

    Frames may also be the sensor's 16-bit containers (uint16 / int16 tensors): they are divided by
    2**raw_bits - 1 inside the kernel, which is what the reference's datasets do on the host (dataset.py:86-87)."""

    raw_bits = 16

    def __init__(self, reduce_size=True, out_channels=3, track_stages=False, normalize_mosaic=None):
        super().__init__()
        self.stages = None
        self.buffer = None
        self.reduce_size = reduce_size
        self.out_channels = out_channels
        self.track_stages = track_stages
        self.normalize_mosaic = normalize_mosaic

    def forward(self, raw):
        self.stages = {}
        self.buffer = {}

        rgb = F_.raw2rgb_bits(raw, None, self.reduce_size, self.out_channels, self.raw_bits)
        self.stages['demosaic'] = rgb
        if self.normalize_mosaic:
            rgb = self.normalize_mosaic(rgb)

        if self.track_stages and raw.requires_grad:
            for stage in self.stages.values():
                stage.retain_grad()

        self.buffer['processed_rgb'] = rgb

        return rgb


class NNProcessing(nn.Module):
    """reference :83-126 (mode `neural_network`).  Only its `raw2rgb` front end is on this library's hot path
    (SURVEY.md section 8a, a12); the body is segmentation_models_pytorch's UnetPlusPlus, a third-party model that is
    out of scope and absent from the image, so constructing this class without it raises."""

    raw_bits = 16

    def __init__(self, track_stages=False, normalize_mosaic=None, batch_norm_output=True):
        super().__init__()
        try:
            import segmentation_models_pytorch as smp
        except ImportError as e:  # pragma: no cover - smp is absent from the build image
            raise ImportError('NNProcessing needs segmentation_models_pytorch (reference :11, :97)') from e
        self.stages, self.buffer, self.track_stages = None, None, track_stages
        self.normalize_mosaic = normalize_mosaic
        self.model = smp.UnetPlusPlus(encoder_name='resnet34', encoder_depth=3, decoder_channels=[256, 128, 64],
                                      in_channels=3, classes=3)          # reference :97-103
        self.batch_norm = nn.BatchNorm2d(3, affine=False) if batch_norm_output else None

    def front_end(self, raw):
        """the part of forward() this library serves: packed 3-channel mosaic (+ normalize_mosaic), reference :111-114"""
        rgb = F_.raw2rgb_bits(raw, bits=self.raw_bits)
        return self.normalize_mosaic(rgb) if self.normalize_mosaic else rgb

    def forward(self, raw):
        self.stages = {'demosaic': self.front_end(raw)}
        rgb = self.model(self.stages['demosaic'])
        if self.batch_norm is not None:
            rgb = self.batch_norm(rgb)
        self.stages['rgb'] = rgb
        if self.track_stages and raw.requires_grad:
            for stage in self.stages.values():
                stage.retain_grad()
        self.buffer = {'processed_rgb': rgb}
        return rgb


def append_additive_layer(processor):
    """reference :129-131."""
    device = processor.gamma_correct.device if hasattr(processor, 'gamma_correct') else None
    processor.additive_layer = nn.Parameter(torch.zeros((1, 3, 256, 256), device=device))


class Debayer(nn.Conv2d):
    """reference :228-237: trainable 3->3 3x3 conv, mirror ('reflect') padding, bilinear initial weights.

    Inside ParametrizedProcessing only its ``weight`` is read: the fused kernel folds the 81 weights with
    the white balance, colour matrix and RGB->YUV matrix into per-Bayer-parity 3x3 stencils."""

    def __init__(self):
        super().__init__(3, 3, kernel_size=3, padding=1, padding_mode='reflect', bias=False)
        self.weight.data.fill_(0)
        self.weight.data[0, 0] = K_RB.clone()
        self.weight.data[1, 1] = K_G.clone()
        self.weight.data[2, 2] = K_RB.clone()


class _LazyStages(dict):
    """``stages`` after a call on the fused kernels.  The fused path keeps every intermediate in LDS / registers,
    so unlike the reference (:183-214) it has no stage tensors to put into the dict; they are computed by the
    stage-by-stage kernels (no grad) the first time anybody looks into the dict, from the frames of that call --
    callers that never read ``stages`` pay nothing, callers that do see what the reference shows them.  (The
    frames of the last call stay referenced until the next call.  If a parameter was modified in between, the
    stages of that call can no longer be reproduced and the access raises; track_stages=True materialises them
    during the call, gradients included.)"""

    def __init__(self, module, raw):
        super().__init__()
        self._pending = (module, raw, tuple(p._version for p in module.parameters()))

    def _fill(self):
        if self._pending is None:
            return
        module, raw, versions = self._pending
        self._pending = None
        if versions != tuple(p._version for p in module.parameters()):
            raise RuntimeError('processor.stages of a fused forward call were read after the parameters changed; '
                               'use track_stages=True to materialise the stages during the call')
        from ..staged import staged_forward
        with torch.no_grad():
            staged_forward(module, raw.detach(), stages=self, with_batch_norm=False)

    def __getitem__(self, k):
        self._fill()
        return super().__getitem__(k)

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __len__(self):
        self._fill()
        return super().__len__()

    def __contains__(self, k):
        self._fill()
        return super().__contains__(k)

    def keys(self):
        self._fill()
        return super().keys()

    def values(self):
        self._fill()
        return super().values()

    def items(self):
        self._fill()
        return super().items()

    def get(self, k, default=None):
        self._fill()
        return super().get(k, default)

    def __repr__(self):
        self._fill()
        return super().__repr__()

    def __reduce__(self):
        self._fill()
        return (dict, (dict(self),))


class ParametrizedProcessing(nn.Module):
    """Differentiable processing pipeline, reference :134-225, as fused gfx950 kernels.

    Args:
        camera_parameters (tuple(list), optional): (black_level, white_balance, colour_matrix)
        track_stages (bool, optional): whether or not to retain intermediary steps in processing
        batch_norm_output (bool, optional): adds a BatchNorm layer to the end of the processing

    Extra attribute (not in the reference): ``process_group`` -- when set to a torch.distributed group of
    more than one rank, BatchNorm batch statistics (forward) and their backward sums are exchanged over
    RCCL so that every rank normalises with the statistics of the GLOBAL batch, which is what the
    single-GPU reference computes for that batch.  ``raw_bits`` -- frames given as
    uint16 / int16 tensors (the sensor's 16-bit containers) are divided by 2**raw_bits - 1 inside the kernels,
    bit-identically to the host-side normalisation of the reference's datasets (dataset.py:86-87).
    ``supports_output_epilogue`` -- this class pops the one-shot `_epilogue` an augmentation armed (ComposeState.arm)
    in its forward; processors without the attribute are never armed."""

    raw_bits = 16
    supports_output_epilogue = True

    def __init__(self, camera_parameters=None, track_stages=False, batch_norm_output=True):
        super().__init__()
        self.stages = None
        self.buffer = None
        self.track_stages = track_stages

        if camera_parameters is None:
            camera_parameters = DEFAULT_CAMERA_PARAMS

        black_level, white_balance, colour_matrix = camera_parameters

        self.black_level = nn.Parameter(torch.as_tensor(black_level))
        self.white_balance = nn.Parameter(torch.as_tensor(white_balance).reshape(1, 3))
        self.colour_correction = nn.Parameter(torch.as_tensor(colour_matrix).reshape(3, 3))

        self.gamma_correct = nn.Parameter(torch.Tensor([2.2]))

        self.debayer = Debayer()

        self.sharpening_filter = nn.Conv2d(1, 1, kernel_size=3, padding=1, bias=False)
        self.sharpening_filter.weight.data[0][0] = K_SHARP.clone()

        self.gaussian_blur = nn.Conv2d(1, 1, kernel_size=5, padding=2, padding_mode='reflect', bias=False)
        self.gaussian_blur.weight.data[0][0] = K_BLUR.clone()

        self.batch_norm = nn.BatchNorm2d(3, affine=False) if batch_norm_output else None

        self.register_buffer('M_RGB_2_YUV', M_RGB_2_YUV.clone())
        self.register_buffer('M_YUV_2_RGB', M_YUV_2_RGB.clone())

        self.additive_layer = None  # this can be added in later

        self.process_group = None

    # -- packed parameter block of the C ABI (include/r2l_isp.h, R2L_P_*) -----------------------------
    def packed_parameters(self):
        return torch.cat([
            self.black_level.reshape(-1), self.white_balance.reshape(-1),
            self.colour_correction.reshape(-1), self.gamma_correct.reshape(-1),
            self.debayer.weight.reshape(-1), self.sharpening_filter.weight.reshape(-1),
            self.gaussian_blur.weight.reshape(-1),
            self.M_RGB_2_YUV.reshape(-1), self.M_YUV_2_RGB.reshape(-1)]).to(torch.float32)

    def __getstate__(self):
        state = self.__dict__.copy()
        state['process_group'] = None      # communicators are neither picklable nor deep-copyable
        state['stages'] = None
        state['buffer'] = None
        return state

    def forward(self, raw):
        assert raw.ndim == 3, f"needs dims (B, H, W), got {raw.shape}"

        d = self.__dict__          # (plain attributes: nn.Module.__setattr__ costs microseconds per step)
        d['stages'] = {}
        d['buffer'] = {}

        # The fused kernels keep every intermediate in LDS / registers: they neither materialise the stage
        # tensors nor produce d/d raw.  Whenever a caller can observe either (track_stages=True, or frames
        # that require grad as in model.py:228), the stage-by-stage kernels run instead and fill
        # ``self.stages`` exactly like the reference (:183-214).
        # a one-shot output epilogue armed by the augmentation drop-in (ComposeState.arm): the flips / rot90 of
        # utils/augmentation.py:70-74 (applied to this module's output at model.py:79-81) leave the fused kernels as part
        # of their output stores; on the staged path, or where the kernels cannot take it, the moves run as the separate
        # permutation kernel right here -- either way the caller gets the augmented batch
        epilogue = d.pop('_epilogue', None)
        if self.track_stages or (raw.requires_grad and torch.is_grad_enabled()):
            from ..staged import staged_forward
            rgb = staged_forward(self, raw)
        else:
            # (90-degree rotations: the stores of a row-walking kernel land one element per row of the rotated output --
            # 1.16 ms per step at 64x512x512 -- while the permutation kernel transposes through LDS tiles: 0.40 + 2 x 0.1 ms;
            # `fuse_rot90 = True` on the module takes the kernels' own path, square frames only)
            odd = bool(epilogue is not None and (epilogue[2] & 1))
            fuse = epilogue is not None and F_.epilogue_supported(raw, self) and \
                (not odd or (getattr(self, 'fuse_rot90', False) and raw.shape[-1] == raw.shape[-2]))
            rgb = self._fused_forward(raw, epilogue if fuse else None)
            d['stages'] = _LazyStages(self, raw)
            if fuse:
                epilogue = None
        if epilogue is not None:
            from ..augmentation import flip_rot
            rgb = flip_rot(rgb, *epilogue)

        if self.track_stages and raw.requires_grad:
            for stage in self.stages.values():
                stage.retain_grad()

        self.buffer['processed_rgb'] = rgb

        return rgb

    def _fused_forward(self, raw, epilogue=None):
        bn = self.batch_norm
        if bn is None:
            mode = F_.BN_NONE
        elif bn.training or (bn.running_mean is None):
            mode = F_.BN_TRAIN      # batch statistics; running statistics are updated on the device
        else:
            mode = F_.BN_EVAL
        return F_.isp_fused(raw, self, mode, self.process_group, epilogue)
