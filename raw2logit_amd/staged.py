"""Staged execution of ParametrizedProcessing (track_stages=True), reference pipeline_torch.py:175-225.

Every stage the reference materialises is a tensor produced by its own HIP kernel and its own
torch.autograd.Function, so `stage.retain_grad()` / `stage.grad` (model.py:249-254) and d/d raw work exactly
as with the reference's ATen graph.  Slower than the fused path by construction (about twenty passes over the
frames); used for visualisation / gradient tracking, not for training throughput."""
import torch

from . import _lib
from ._lib import ptr
from .functional import raw2rgb_bits, _f32c, gather_ranks, bn_finalize, bn_bwd_means


def _ws(lib, like):
    n = lib.r2l_stage_workspace_bytes()
    return torch.empty(n, dtype=torch.uint8, device=like.device), n


def _dims(x):
    B, C, H, W = x.shape
    assert C == 3
    return B, H, W


class _Conv33(torch.autograd.Function):
    """Debayer: nn.Conv2d(3, 3, 3, padding=1, padding_mode='reflect', bias=False) (:228-237)."""

    @staticmethod
    def forward(ctx, x, w):
        x = _f32c(x, 'x')
        w = _f32c(w, 'weight')
        B, H, W = _dims(x)
        lib, s = _lib.library_for(x)
        y = torch.empty_like(x)
        lib.check(lib.r2l_stage_conv33_fwd(ptr(x), ptr(w), ptr(y), B, H, W, s), 'conv33_fwd')
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _f32c(g, 'g')
        B, H, W = _dims(x)
        lib, s = _lib.library_for(x)
        ws, n = _ws(lib, x)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty(81, dtype=torch.float32, device=x.device)
        lib.check(lib.r2l_stage_conv33_bwd(ptr(x), ptr(w), ptr(g), ptr(gx), ptr(gw), ptr(ws), n, B, H, W, s),
                  'conv33_bwd')
        return gx, gw.view(3, 3, 3, 3)


class _Mix3(torch.autograd.Function):
    """torch.einsum('bchw,kc->bkhw', x, M) with a 3x3 matrix (:191, :194, :198-203)."""

    @staticmethod
    def forward(ctx, x, m):
        x = _f32c(x, 'x')
        m = _f32c(m, 'M')
        B, H, W = _dims(x)
        lib, s = _lib.library_for(x)
        y = torch.empty_like(x)
        lib.check(lib.r2l_stage_mix3_fwd(ptr(x), ptr(m), ptr(y), B, H, W, s), 'mix3_fwd')
        ctx.save_for_backward(x, m)
        return y

    @staticmethod
    def backward(ctx, g):
        x, m = ctx.saved_tensors
        g = _f32c(g, 'g')
        B, H, W = _dims(x)
        lib, s = _lib.library_for(x)
        ws, n = _ws(lib, x)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gm = torch.empty(9, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[1] else None
        lib.check(lib.r2l_stage_mix3_bwd(ptr(x), ptr(m), ptr(g), ptr(gx), ptr(gm), ptr(ws), n, B, H, W, s),
                  'mix3_bwd')
        return gx, (gm.view(3, 3) if gm is not None else None)


class _PlaneConv(torch.autograd.Function):
    """yuv[:, [0]] = conv(yuv[:, [0]]): K=3 zero padding (:162-163, :195) or K=5 mirror padding (:165, :202)."""

    @staticmethod
    def forward(ctx, x, k, mirror):
        x = _f32c(x, 'x')
        K = k.shape[-1]
        kk = _f32c(k.reshape(K, K), 'kernel')
        B, H, W = _dims(x)
        lib, s = _lib.library_for(x)
        y = torch.empty_like(x)
        lib.check(lib.r2l_stage_pconv_fwd(ptr(x), ptr(kk), ptr(y), K, int(mirror), B, H, W, s), 'pconv_fwd')
        ctx.save_for_backward(x, kk)
        ctx.meta = (K, int(mirror), tuple(k.shape))
        return y

    @staticmethod
    def backward(ctx, g):
        x, kk = ctx.saved_tensors
        K, mirror, kshape = ctx.meta
        g = _f32c(g, 'g')
        B, H, W = _dims(x)
        lib, s = _lib.library_for(x)
        ws, n = _ws(lib, x)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gk = torch.empty(25, dtype=torch.float32, device=x.device)
        lib.check(lib.r2l_stage_pconv_bwd(ptr(x), ptr(kk), ptr(g), ptr(gx), ptr(gk), K, mirror, ptr(ws), n,
                                          B, H, W, s), 'pconv_bwd')
        return gx, gk.view(5, 5)[:K, :K].reshape(kshape), None


def _point(lib, s, op, like, x=None, g=None, w=None, aux=None, aux2=None, out=True, sums=False):
    B, H, W = _dims(like)
    y = torch.empty_like(like) if out else None
    sm = ws = None
    n = 0
    if sums:
        sm = torch.empty(6, dtype=torch.float32, device=like.device)
        ws, n = _ws(lib, like)
    lib.check(lib.r2l_stage_point(op, ptr(x), ptr(g), ptr(w), ptr(aux), ptr(aux2), ptr(y), ptr(sm), ptr(ws), n,
                                  B, H, W, s), f'point[{op}]')
    return y, sm


class _Clip(torch.autograd.Function):
    """torch.clip(rgb, 1e-5, 1) (:206)."""

    @staticmethod
    def forward(ctx, x):
        x = _f32c(x, 'x')
        lib, s = _lib.library_for(x)
        y, _ = _point(lib, s, 0, x, x=x)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        lib, s = _lib.library_for(x)
        gx, _ = _point(lib, s, 1, x, x=x, g=_f32c(g, 'g'))
        return gx


class _Gamma(torch.autograd.Function):
    """torch.exp((1 / gamma) * torch.log(rgb)) (:209)."""

    @staticmethod
    def forward(ctx, x, gamma):
        x = _f32c(x, 'x')
        gm = _f32c(gamma.reshape(1), 'gamma')
        lib, s = _lib.library_for(x)
        y, _ = _point(lib, s, 2, x, x=x, w=gm)
        ctx.save_for_backward(x, gm, y)
        ctx.gshape = tuple(gamma.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gm, y = ctx.saved_tensors
        lib, s = _lib.library_for(x)
        gx, sm = _point(lib, s, 3, x, x=x, g=_f32c(g, 'g'), w=gm, aux=y, sums=True)
        # d/d gamma exp(ln(x)/gamma) = out * ln(x) * (-1/gamma^2); the kernel summed g*out*log2(x)
        ggamma = (sm[0] * (-0.6931471805599453) / (gm * gm)).reshape(ctx.gshape)
        return gx, ggamma


class _Add(torch.autograd.Function):
    """rgb + additive_layer, (1,3,H,W) broadcast over the batch (:213)."""

    @staticmethod
    def forward(ctx, x, p):
        x = _f32c(x, 'x')
        p = _f32c(p, 'additive_layer')
        if tuple(p.shape) != (1,) + tuple(x.shape[1:]):
            raise RuntimeError(f'additive_layer {tuple(p.shape)} does not broadcast to {tuple(x.shape)}')
        lib, s = _lib.library_for(x)
        y, _ = _point(lib, s, 4, x, x=x, w=p)
        ctx.pshape = tuple(p.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        g = _f32c(g, 'g')
        B, H, W = _dims(g)
        lib, s = _lib.library_for(g)
        gp = torch.empty(ctx.pshape, dtype=torch.float32, device=g.device)
        lib.check(lib.r2l_additive_bwd(ptr(g), None, None, None, ptr(gp), B, H, W, s), 'additive_bwd')
        return g, gp


class _BatchNorm(torch.autograd.Function):
    """nn.BatchNorm2d(3, affine=False) (:168, :216-217): batch statistics in train mode (running statistics
    updated on the device), running statistics in eval mode."""

    @staticmethod
    def forward(ctx, x, bn_module, training, group):
        x = _f32c(x, 'x')
        B, H, W = _dims(x)
        lib, s = _lib.library_for(x)
        dev = x.device
        ctx.training = training
        ctx.group = group
        if training:
            _, sm = _point(lib, s, 7, x, x=x, out=False, sums=True)
            stats = torch.cat([sm.to(torch.float64),
                               torch.full((1,), float(B * H * W), dtype=torch.float64, device=dev)])
            gathered, nranks = gather_ranks(stats, group, 'bn statistics all-gather')
            bn, moments = bn_finalize(lib, s, gathered, nranks, bn_module, bn_module.eps, bn_module.momentum)
            ctx.moments = moments
        else:
            mean = bn_module.running_mean.detach().to(device=dev, dtype=torch.float64)
            var = bn_module.running_var.detach().to(device=dev, dtype=torch.float64)
            bn = torch.cat([mean, torch.rsqrt(var + bn_module.eps)]).to(torch.float32)
        y, _ = _point(lib, s, 5, x, x=x, w=bn)
        ctx.save_for_backward(bn, y)
        return y

    @staticmethod
    def backward(ctx, g):
        bn, y = ctx.saved_tensors
        g = _f32c(g, 'g')
        B, H, W = _dims(g)
        lib, s = _lib.library_for(g)
        coef = None
        if ctx.training:
            sums = torch.empty(6, dtype=torch.float64, device=g.device)
            n = lib.r2l_isp_workspace_bytes(B, H, W)
            ws = torch.empty(n, dtype=torch.uint8, device=g.device)
            lib.check(lib.r2l_bn_bwd_reduce(ptr(g), ptr(y), None, ptr(sums), None, ptr(ws), n, B, H, W, 0, s),
                      'bn_bwd_reduce')
            coef = bn_bwd_means(lib, s, sums, ctx.moments, ctx.group)   # (one rank: just sums / n)
        gx, _ = _point(lib, s, 6, g, g=g, w=bn, aux=y, aux2=coef)
        return gx, None, None, None


def staged_forward(module, raw, stages=None, with_batch_norm=True):
    """The body of ParametrizedProcessing.forward (:183-217), stage by stage; fills module.stages (or `stages`)."""
    m = module
    if stages is None:
        stages = m.stages
    rgb = raw2rgb_bits(raw, m.black_level, False, 3, getattr(m, 'raw_bits', 16))          # :183
    stages['demosaic'] = rgb
    rgb = _Conv33.apply(rgb, m.debayer.weight)                                           # :187
    rgb = _Mix3.apply(rgb, torch.diag(m.white_balance.reshape(3)))                       # :190
    rgb = _Mix3.apply(rgb, m.colour_correction)                                          # :191
    stages['color_correct'] = rgb
    yuv = _Mix3.apply(rgb, m.M_RGB_2_YUV)                                                # :194
    yuv = _PlaneConv.apply(yuv, m.sharpening_filter.weight, False)                       # :195
    if m.track_stages:  # the reference takes the YUV->RGB->YUV round trip only then (:197-200)
        rgb = _Mix3.apply(yuv, m.M_YUV_2_RGB)                                            # :198
        stages['sharpening'] = rgb
        yuv = _Mix3.apply(rgb, m.M_RGB_2_YUV)                                            # :200
    yuv = _PlaneConv.apply(yuv, m.gaussian_blur.weight, True)                            # :202
    rgb = _Mix3.apply(yuv, m.M_YUV_2_RGB)                                                # :203
    stages['gaussian'] = rgb
    rgb = _Clip.apply(rgb)                                                               # :206
    stages['clipped'] = rgb
    rgb = _Gamma.apply(rgb, m.gamma_correct)                                             # :209
    stages['gamma_correct'] = rgb
    if m.additive_layer is not None:                                                     # :212-214
        rgb = _Add.apply(rgb, m.additive_layer)
        stages['noise'] = rgb
    if m.batch_norm is not None and with_batch_norm:                                     # :216-217
        bn = m.batch_norm
        training = bn.training or bn.running_mean is None
        rgb = _BatchNorm.apply(rgb, bn, training, m.process_group)
    return rgb
