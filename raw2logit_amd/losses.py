"""Adversarial auxiliary losses of the reference (train.py:255-275) on the GPU: drop-ins for ``utils.ssim.SSIM``,
``utils.ssim.ssim``, ``utils.base.l2_regularization`` and ``utils.base.AuxLoss``.

In the adversarial setup the ISP runs twice per step (default processor under no_grad, adversarial processor
with grad) and this loss ties the two outputs together; the gradient flows to the second argument only when the
first does not require grad (which is how AuxLoss calls it, utils/base.py:354-358)."""
import torch
import torch.nn as nn

from . import _lib
from ._lib import ptr
from .functional import _f32c


def _aux_ws(lib, x):
    B, C, H, W = x.shape
    n = lib.r2l_aux_workspace_bytes(B, C, H, W)
    return torch.empty(n, dtype=torch.uint8, device=x.device), n


class _Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        img1 = _f32c(img1, 'img1')
        img2 = _f32c(img2, 'img2')
        if img1.shape != img2.shape or img1.ndim != 4:
            raise ValueError(f'SSIM needs two (B,C,H,W) images of one shape, got {tuple(img1.shape)} and '
                             f'{tuple(img2.shape)}')
        B, C, H, W = img1.shape
        lib, stream = _lib.library_for(img1)
        ws, n = _aux_ws(lib, img1)
        out = torch.empty(1, dtype=torch.float64, device=img1.device)
        keep = bool(ctx.needs_input_grad[1])     # the same launch then leaves dS/d(moments) for the backward
        lib.check(lib.r2l_ssim_fwd(ptr(img1), ptr(img2), ptr(out), ptr(ws), n, int(keep), B, C, H, W, stream),
                  'r2l_ssim_fwd')
        ctx.save_for_backward(img1, img2)
        ctx.ws = ws if keep else None
        return out.to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g):
        img1, img2 = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise _lib.R2LError('SSIM: the gradient is built for the second argument only (AuxLoss computes the '
                                'first under torch.no_grad(), utils/base.py:354-356)')
        B, C, H, W = img1.shape
        lib, stream = _lib.library_for(img1)
        have = ctx.ws is not None
        ws, n = (ctx.ws, ctx.ws.numel()) if have else _aux_ws(lib, img1)
        grad = torch.empty_like(img2)
        gs = _f32c(g.reshape(1).to(torch.float32), 'grad')
        lib.check(lib.r2l_ssim_bwd(ptr(img1), ptr(img2), ptr(gs), ptr(grad), ptr(ws), n, int(have), B, C, H, W,
                                   stream), 'r2l_ssim_bwd')
        return None, grad


class _L2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x = _f32c(x, 'x')
        y = _f32c(y, 'y')
        if x.shape != y.shape:
            raise ValueError('l2_regularization needs two tensors of one shape')
        if x.numel() % 4:
            raise ValueError('l2_regularization: the element count must be a multiple of 4')
        lib, stream = _lib.library_for(x)
        ws = torch.empty(lib.r2l_aux_workspace_bytes(1, 1, 2, 2), dtype=torch.uint8, device=x.device)
        out = torch.empty(1, dtype=torch.float64, device=x.device)
        lib.check(lib.r2l_l2_fwd(ptr(x), ptr(y), ptr(out), ptr(ws), ws.numel(), x.numel(), stream), 'r2l_l2_fwd')
        ctx.save_for_backward(x, y)
        return out.to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise _lib.R2LError('l2_regularization: the gradient is built for the second argument only')
        lib, stream = _lib.library_for(x)
        grad = torch.empty_like(y)
        gs = _f32c(g.reshape(1).to(torch.float32), 'grad')
        lib.check(lib.r2l_l2_bwd(ptr(x), ptr(y), ptr(gs), ptr(grad), x.numel(), stream), 'r2l_l2_bwd')
        return None, grad


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/ssim.py:66-74."""
    if window_size != 11 or not size_average:
        raise NotImplementedError('only SSIM(window_size=11, size_average=True) -- what train.py:262 uses -- is '
                                  'built for the GPU')
    return _Ssim.apply(img1, img2)


class SSIM(nn.Module):
    """utils/ssim.py:41-64 (same constructor); forward(img1, img2) -> scalar mean SSIM."""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        if window_size != 11 or not size_average:
            raise NotImplementedError('only SSIM(window_size=11, size_average=True) is built for the GPU')
        self.window_size = window_size
        self.size_average = size_average

    def forward(self, img1, img2):
        return _Ssim.apply(img1, img2)


def l2_regularization(x, y):
    """utils/base.py:342-343."""
    return _L2.apply(x, y)


class AuxLoss(nn.Module):
    """utils/base.py:346-358 (``self.processor`` there is a typo for ``processor_adv``: it only works because
    LitModel assigns the attribute from outside; here both names refer to the adversarial processor)."""

    def __init__(self, loss_aux, processor_adv, processor_default, weight=1):
        super().__init__()
        self.loss_aux = loss_aux
        self.weight = weight
        self.processor_adv = processor_adv
        self.processor_default = processor_default

    @property
    def processor(self):
        return self.processor_adv

    def forward(self, x):
        with torch.no_grad():
            x_reference = self.processor_default(x)
        x_processed = self.processor.buffer['processed_rgb']
        return self.weight * self.loss_aux(x_reference, x_processed)
