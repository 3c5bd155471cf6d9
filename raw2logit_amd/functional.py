"""autograd-facing ops over the C ABI (include/r2l_isp.h).  PyTorch is plumbing here: it owns device
memory, streams and torch.distributed; all image arithmetic happens in the HIP kernels."""
import ctypes
import weakref

import torch
import torch.distributed as dist

from . import _lib
from ._lib import ptr


def _f32c(t, name):
    if t.dtype != torch.float32:
        raise TypeError(f'{name} must be float32, got {t.dtype}')
    return t if t.is_contiguous() else t.contiguous()


U16_DTYPES = (torch.uint16, torch.int16)   # int16 = the same 16 bits (torch's uint16 support is recent)


def _raw_arg(raw, bits, name='raw'):
    """(tensor, denom | None): float32 frames in [0,1], or the sensor's 16-bit containers with
    denom = 2**bits - 1 (dataset.py:86-87: the normalisation then happens inside the kernel)."""
    if raw.dtype in U16_DTYPES:
        if raw.shape[-1] % 4:
            raise ValueError('16-bit frames need W % 4 == 0')
        return (raw if raw.is_contiguous() else raw.contiguous()), float(2 ** int(bits) - 1)
    return _f32c(raw, name), None


def _workspace(lib, like, B, H, W):
    n = lib.r2l_isp_workspace_bytes(B, H, W)
    return torch.empty(n, dtype=torch.uint8, device=like.device), n


def _group_size(group):
    if group is None or not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size(group)


# --------------------------------------------------------------------------------------------------
# raw2rgb (pipeline_torch.py:240-283)
# --------------------------------------------------------------------------------------------------
class _Raw2Rgb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, black_level, reduce_size, out_channels, bits=16):
        raw, denom = _raw_arg(raw, bits)
        B, H, W = raw.shape
        lib, stream = _lib.library_for(raw)
        bl = None
        if black_level is not None:
            bl = _f32c(black_level.to(device=raw.device, dtype=torch.float32).reshape(-1), 'black_level')
            assert bl.numel() == 4
        if reduce_size:
            out = torch.empty((B, out_channels, H // 2, W // 2), dtype=torch.float32, device=raw.device)
        else:
            out = torch.empty((B, out_channels, H, W), dtype=torch.float32, device=raw.device)
        if denom is None:
            lib.check(lib.r2l_raw2rgb_fwd(ptr(raw), ptr(bl), ptr(out), B, H, W, int(reduce_size),
                                          int(out_channels), stream), 'r2l_raw2rgb_fwd')
        else:
            lib.check(lib.r2l_raw2rgb_fwd_u16(ptr(raw), denom, ptr(bl), ptr(out), B, H, W, int(reduce_size),
                                              int(out_channels), stream), 'r2l_raw2rgb_fwd_u16')
        ctx.dims = (B, H, W, bool(reduce_size), int(out_channels))
        ctx.has_bl = black_level is not None
        ctx.bl_shape = None if black_level is None else tuple(black_level.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        B, H, W, reduce_size, oc = ctx.dims
        gout = _f32c(gout, 'grad_out')
        lib, stream = _lib.library_for(gout)
        need_raw = ctx.needs_input_grad[0]
        need_bl = ctx.has_bl and ctx.needs_input_grad[1]
        graw = torch.empty((B, H, W), dtype=torch.float32, device=gout.device) if need_raw else None
        gbl = ws = None
        nws = 0
        if need_bl:
            gbl = torch.empty(4, dtype=torch.float64, device=gout.device)
            nws = lib.r2l_raw2rgb_bwd_workspace_bytes(B, H, W)
            ws = torch.empty(nws, dtype=torch.uint8, device=gout.device)
        if need_raw or need_bl:
            lib.check(lib.r2l_raw2rgb_bwd(ptr(gout), ptr(graw), ptr(gbl), ptr(ws), nws, B, H, W,
                                          int(reduce_size), oc, stream), 'r2l_raw2rgb_bwd')
        if gbl is not None:
            gbl = gbl.to(torch.float32).reshape(ctx.bl_shape)
        return graw, gbl, None, None, None


def raw2rgb(raw, black_level=None, reduce_size=True, out_channels=3):
    """drop-in for processing.pipeline_torch.raw2rgb (:240-283), same signature.  `raw` may also hold 16-bit
    containers (uint16 / int16 tensor): they are divided by 2**16 - 1 inside the kernel (dataset.py:86-87);
    raw2rgb_bits() takes other bit depths."""
    return raw2rgb_bits(raw, black_level, reduce_size, out_channels, 16)


def raw2rgb_bits(raw, black_level=None, reduce_size=True, out_channels=3, bits=16):
    assert out_channels in [3, 4]
    if black_level is not None and not torch.is_tensor(black_level):
        black_level = torch.as_tensor(black_level, dtype=torch.float32, device=raw.device)
    return _Raw2Rgb.apply(raw, black_level, reduce_size, out_channels, bits)


# --------------------------------------------------------------------------------------------------
# fused parametrized ISP (ParametrizedProcessing.forward, pipeline_torch.py:175-225)
# --------------------------------------------------------------------------------------------------
BN_NONE, BN_TRAIN, BN_EVAL = 0, 1, 2

# (name, offset, numel) of the trainable tensors inside the packed block (include/r2l_isp.h, R2L_P_*)
PARAM_LAYOUT = (('black_level', 0, 4), ('white_balance', 4, 3), ('colour_correction', 7, 9),
                ('gamma_correct', 16, 1), ('debayer.weight', 17, 81), ('sharpening_filter.weight', 98, 9),
                ('gaussian_blur.weight', 107, 25))


class CommTimer:
    """wall time of every ISP collective (bench.py's `comm_us`): device events on the current stream around each
    exchange (host clock for CPU tensors).  Off unless a bench pass switches it on; costs one branch per collective."""
    on = False
    _open = []

    @classmethod
    def enable(cls, on):
        cls.on = bool(on)
        cls._open = []

    @classmethod
    def begin(cls, name, t):
        if not cls.on:
            return None
        if t.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            return (name, e0, e1)
        import time
        return (name, time.perf_counter(), None)

    @classmethod
    def end(cls, tok):
        if tok is None:
            return
        name, e0, e1 = tok
        if e1 is None:
            import time
            cls._open.append((name, 1e6 * (time.perf_counter() - e0)))
        else:
            e1.record()
            cls._open.append((name, e0, e1))

    @classmethod
    def report(cls):
        """{name: {'calls': n, 'avg_us': t}} of everything recorded since enable(True)"""
        acc = {}
        for rec in cls._open:
            if len(rec) == 3:
                rec[2].synchronize()
                us = 1e3 * rec[1].elapsed_time(rec[2])
            else:
                us = rec[1]
            a = acc.setdefault(rec[0], [0, 0.0])
            a[0] += 1
            a[1] += us
        cls._open = []
        return {k: {'calls': n, 'avg_us': round(t / n, 1)} for k, (n, t) in acc.items()}


_HOST_STAGED = weakref.WeakKeyDictionary()     # process group object -> (backend string, staged); dies with the group


def _backend_for_device_tensors(group):
    name = str(dist.get_backend(group))
    if ',' in name or ':' in name:                   # "cpu:gloo,cuda:nccl": the entry of the cuda device type
        per = dict(part.split(':', 1) for part in name.split(',') if ':' in part)
        name = per.get('cuda', name)
    return name.strip().lower()


def _host_staged(group, t):
    """gloo moves host memory: device tensors of a gloo group (several ranks sharing one GPU -- the functional
    multi-rank check on a one-GPU box) cross through a host copy.  RCCL ("nccl") takes device pointers as they are.
    The backend that serves the tensor's DEVICE decides ("cpu:gloo,cuda:nccl" groups hand device tensors to RCCL).
    Cached on the group OBJECT (weakly: an id() can be reused by a later group after destroy_process_group() and a
    re-init with another backend in the same process); group=None resolves to the default group's object first."""
    if not t.is_cuda:
        return False
    if group is None:
        # the default group: key the cache on the default ProcessGroup OBJECT (a re-init after destroy_process_group() makes a
        # new one), so that the eager data-parallel step pays no backend-string lookup per collective
        try:
            group = dist.distributed_c10d._get_default_group()
        except Exception:                            # noqa: BLE001  (private accessor gone: the per-call lookup still works)
            return _backend_for_device_tensors(None) == 'gloo'
    try:
        ent = _HOST_STAGED.get(group)
    except TypeError:                                # (a group object that cannot be weakly referenced: no cache)
        return _backend_for_device_tensors(group) == 'gloo'
    if ent is None:
        name = _backend_for_device_tensors(group)
        ent = (name, name == 'gloo')
        try:
            _HOST_STAGED[group] = ent
        except TypeError:
            pass
    return ent[1]


def split_single_rank(group):
    """test hook (R2L_SPLIT_SINGLE_RANK=1): let a group of ONE rank take the N > 1 code path -- both step calls split
    around real all-gathers, the gradient all-reduce issued -- so that a one-GPU box runs (and captures into a HIP
    graph) the RCCL collectives of the multi-GPU step.  Off unless set; has no effect without an initialised group."""
    import os
    return group is not None and dist.is_available() and dist.is_initialized() and \
        os.environ.get('R2L_SPLIT_SINGLE_RANK') == '1'


def gather_ranks(vec, group=None, what='gather'):
    """this rank's small float64 vector -> (all ranks' vectors, rank-major in one flat tensor, nranks), over
    RCCL/xGMI.  The kernels that consume it add the rows in rank order, so every rank computes bit-identical
    results, equal to the single-GPU result for the global batch."""
    n = _group_size(group)
    if n == 1 and not split_single_rank(group):
        return vec, 1
    tok = CommTimer.begin(what, vec)
    if _host_staged(group, vec):
        h = vec.cpu()
        oh = torch.empty(n * h.numel(), dtype=h.dtype)
        dist.all_gather_into_tensor(oh, h, group=group)
        out = oh.to(vec.device)
    else:
        out = torch.empty(n * vec.numel(), dtype=vec.dtype, device=vec.device)   # rank-major, flat (gloo wants 1-D)
        dist.all_gather_into_tensor(out, vec, group=group)
    CommTimer.end(tok)
    return out, n


class GradAllReduce:
    """data-parallel sum of the ISP's parameter gradients (132 floats; + 196 608 with an additive layer): ONE flat
    all-reduce, issued asynchronously right after backward -- nothing needs the result before the optimiser (or the
    next forward), so it overlaps with whatever the caller does in between (the task model's own DDP buckets).

        h = GradAllReduce(params, group)     # after loss.backward()
        ...
        h.wait()                             # before optimiser.step(): p.grad now holds the sum over ranks
    `average=True` divides by the number of ranks (DistributedDataParallel's convention)."""

    def __init__(self, params, group=None, average=False):
        self.params = [p for p in params if p.grad is not None]
        self.n = _group_size(group)
        self.work = self.flat = self.host = None
        self.average = average
        if (self.n == 1 and not split_single_rank(group)) or not self.params:
            return
        self.tok = CommTimer.begin('grad all-reduce', self.params[0].grad)
        self.flat = torch.cat([p.grad.reshape(-1) for p in self.params])
        if _host_staged(group, self.flat):
            self.host = self.flat.cpu()
            self.work = dist.all_reduce(self.host, group=group, async_op=True)
        else:
            self.work = dist.all_reduce(self.flat, group=group, async_op=True)

    def wait(self):
        if self.work is None:
            return
        self.work.wait()
        if self.host is not None:
            self.flat.copy_(self.host)
        if self.average:
            self.flat /= self.n
        torch._foreach_copy_([p.grad for p in self.params],
                             [c.view_as(p.grad) for c, p in
                              zip(self.flat.split([p.numel() for p in self.params]), self.params)])
        CommTimer.end(self.tok)
        self.work = None


def _bn_buffers(bn_module, dev):
    """(running_mean, running_var, num_batches_tracked) of a BatchNorm module that tracks them, else Nones"""
    if bn_module is not None and bn_module.track_running_stats and bn_module.running_mean is not None:
        rm, rv, nbt = bn_module.running_mean, bn_module.running_var, bn_module.num_batches_tracked
        if rm.device != dev or nbt.dtype != torch.int64:
            raise RuntimeError('BatchNorm buffers must live on the device of the frames')
        return rm, rv, nbt
    return None, None, None


def bn_finalize(lib, stream, stats, nranks, bn_module, eps, momentum):
    """statistics vectors of all ranks -> (mean, istd) float32[6], moments float64[7] (mean, biased var, global
    pixel count); updates the module's running statistics and num_batches_tracked on the device the way
    nn.BatchNorm2d does in train mode."""
    dev = stats.device
    bn = torch.empty(6, dtype=torch.float32, device=dev)
    moments = torch.empty(7, dtype=torch.float64, device=dev)
    rm, rv, nbt = _bn_buffers(bn_module, dev)
    lib.check(lib.r2l_bn_finalize(ptr(stats), nranks, ptr(bn), ptr(moments), ptr(rm), ptr(rv), ptr(nbt),
                                  float(eps), float(momentum) if momentum is not None else -1.0, stream),
              'r2l_bn_finalize')
    return bn, moments


def bn_bwd_means(lib, stream, sums, moments, group):
    """several ranks: the BatchNorm backward sums cross ranks (all-gather, added in rank order in the kernel)"""
    gathered, n = gather_ranks(sums, group, 'bn-bwd sums all-gather')
    bn_bwd = torch.empty(6, dtype=torch.float32, device=sums.device)
    lib.check(lib.r2l_bn_bwd_means(ptr(gathered), n, ptr(moments[6:]), ptr(bn_bwd), stream), 'r2l_bn_bwd_means')
    return bn_bwd


_STEP_ALL, _STEP_A, _STEP_B = 0, 1, 2
_STEP_KEEP_LUMA = 8      # or-ed into `phase` of both calls of a step whose backward will run (R2L_STEP_KEEP_LUMA)
_STEP_EPI_HFLIP, _STEP_EPI_VFLIP, _STEP_EPI_ROT_SHIFT = 16, 32, 6       # output epilogue (R2L_STEP_EPI_*)


def epilogue_bits(epilogue):
    """(hflip, vflip, k) -> the bits r2l_isp_step_fwd / _bwd take in `phase` (include/r2l_isp.h: R2L_STEP_EPI_*)"""
    if not epilogue:
        return 0
    hflip, vflip, k = epilogue
    return (_STEP_EPI_HFLIP if hflip else 0) | (_STEP_EPI_VFLIP if vflip else 0) | ((int(k) & 3) << _STEP_EPI_ROT_SHIFT)


def epilogue_supported(raw, module):
    """can the fused kernels write this module's output through an epilogue?  (no additive layer: its gradient is
    summed in the ISP's own layout; rotations by 90 degrees need square frames -- checked per draw)"""
    return module.additive_layer is None
_STEP_STATS, _STEP_MOMENTS, _STEP_BN_SUMS = 0, 1, 2
_STEP_LAYOUT = {}


def _step_layout(lib, B, H, W):
    """(workspace bytes, byte offsets of the statistics / BatchNorm backward sums inside it) for a frame shape"""
    key = (lib.path, B, H, W)
    lay = _STEP_LAYOUT.get(key)
    if lay is None:
        lay = (lib.r2l_isp_workspace_bytes(B, H, W), lib.r2l_isp_step_offset(_STEP_STATS, B, H, W),
               lib.r2l_isp_step_offset(_STEP_BN_SUMS, B, H, W))
        _STEP_LAYOUT[key] = lay
    return lay


class _IspFused(torch.autograd.Function):
    """out = f(raw, 7 parameter tensors, M_RGB_2_YUV, M_YUV_2_RGB, additive | None, ...): one C-ABI call for the
    forward (r2l_isp_step_fwd: parameter gather + fold, BatchNorm statistics pass + bookkeeping, apply pass) and one
    for the backward (r2l_isp_step_bwd: BatchNorm backward sums, both gradient kernels, additive-layer gradient).
    The workspace tensor carries the step's state (packed parameters as the forward saw them, folded weights,
    BatchNorm constants) from one to the other; backward hands each parameter a view of the single 132-float
    gradient the kernels produce.  With several ranks and train-mode BatchNorm each call splits in two around an
    all-gather of 7 resp. 6 doubles (RCCL)."""

    @staticmethod
    def forward(ctx, raw, bl, wb, ccm, gamma, deb, sharp, blur, m1, m2, additive, bn_mode, bn_module, eps,
                momentum, group, bits=16, grad_mode=True, epilogue=None):
        raw, denom = _raw_arg(raw, bits)
        params = (bl, wb, ccm, gamma, deb, sharp, blur, m1, m2)
        sizes = (4, 3, 9, 1, 81, 9, 25, 9, 9)
        table = (ctypes.c_void_p * 9)()
        alive = []          # contiguous copies of strided parameters: referenced until the launches are enqueued
        f32, rdev = torch.float32, raw.device
        for i in range(9):
            p = params[i]
            if p.dtype is not f32 or p.numel() != sizes[i] or p.device != rdev:
                raise TypeError(f'parameter {i} of the ISP must be {sizes[i]} float32 values on {rdev} '
                                f'(got {tuple(p.shape)} {p.dtype} on {p.device}): the kernels compute in float32 '
                                f'like the reference; .double() / .half() modules are not supported')
            if not p.is_contiguous():
                p = p.detach().contiguous()
                alive.append(p)
            table[i] = p.data_ptr()
        B, H, W = raw.shape
        lib, stream = _lib.library_for(raw)
        if additive is not None:
            additive = _f32c(additive, 'additive_layer')
            if tuple(additive.shape) != (1, 3, H, W):
                raise RuntimeError(f'additive_layer {tuple(additive.shape)} does not broadcast to '
                                   f'frames of {H}x{W}')      # same failure the reference has (:213)
        nws, off_stats, off_sums = _step_layout(lib, B, H, W)
        dev = raw.device
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        epi = epilogue_bits(epilogue)
        if epi and (additive is not None or ((epilogue[2] & 1) and H != W)):
            raise _lib.R2LError('this call cannot take an output epilogue (no additive layer; square frames for a rotation '
                                'by 90 degrees): apply the augmentation to the output instead')
        out = torch.empty((B, 3, W, H) if (epi and (epilogue[2] & 1)) else (B, 3, H, W), dtype=torch.float32, device=dev)
        rm = rv = nbt = None
        if bn_mode == BN_TRAIN:
            rm, rv, nbt = _bn_buffers(bn_module, dev)
        elif bn_mode == BN_EVAL:
            rm, rv = bn_module.running_mean, bn_module.running_var
            if rm.device != dev or rm.dtype != torch.float32:
                raise RuntimeError('BatchNorm buffers must be float32 on the device of the frames')
        nranks = _group_size(group) if bn_mode == BN_TRAIN else 1
        mom = float(momentum) if momentum is not None else -1.0
        # a backward will follow: the forward keeps the sharpened luma plane for its first gradient kernel
        # (needs_input_grad is also set under torch.no_grad(); grad_mode is the caller's torch.is_grad_enabled())
        keep = _STEP_KEEP_LUMA if (grad_mode and any(ctx.needs_input_grad[1:8])) else 0

        def call(phase, gathered):
            lib.check(lib.r2l_isp_step_fwd(ptr(raw), int(denom is not None), denom or 1.0, table, ptr(additive),
                                           bn_mode, ptr(rm), ptr(rv), ptr(nbt), float(eps), mom, ptr(out), ptr(ws),
                                           nws, B, H, W, nranks, phase | keep | epi, ptr(gathered), stream),
                      'r2l_isp_step_fwd')
        split = nranks > 1 or (bn_mode == BN_TRAIN and split_single_rank(group))
        if not split:
            call(_STEP_ALL, None)
        else:
            call(_STEP_A, None)
            gathered, _ = gather_ranks(ws[off_stats:off_stats + 56].view(torch.float64), group,
                                       'bn statistics all-gather')
            call(_STEP_B, gathered)
        del alive
        ctx.bn_mode = bn_mode
        ctx.keep = keep | epi
        ctx.group = group
        ctx.nranks = nranks
        ctx.split = split
        ctx.denom = denom
        ctx.off_sums = off_sums
        ctx.has_additive = additive is not None
        ctx.shapes = [tuple(p.shape) for p in params[:7]]
        ctx.save_for_backward(raw, additive, out)
        ctx.ws = ws
        return out

    @staticmethod
    def backward(ctx, gout):
        raw, additive, out = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise _lib.R2LError(
                'the fused ISP kernels do not produce d/d raw; gradients w.r.t. the raw frames are '
                'only defined on the staged path (track_stages=True)')
        gout = _f32c(gout, 'grad_out')
        B, H, W = raw.shape
        lib, stream = _lib.library_for(raw)
        ws, nws = ctx.ws, ctx.ws.numel()
        need_p = any(ctx.needs_input_grad[1:8])
        need_a = ctx.has_additive and ctx.needs_input_grad[10]
        gp = torch.empty(_lib.R2L_P_NTRAIN, dtype=torch.float32, device=raw.device) if need_p else None
        gadd = torch.empty_like(additive) if need_a else None
        denom = ctx.denom

        def call(phase, gathered):
            lib.check(lib.r2l_isp_step_bwd(ptr(raw), int(denom is not None), denom or 1.0, ptr(additive), ptr(gout),
                                           ptr(out), ptr(gp), ptr(gadd), ctx.bn_mode, ptr(ws), nws, B, H, W,
                                           ctx.nranks, phase | ctx.keep, ptr(gathered), stream), 'r2l_isp_step_bwd')
        if need_p or need_a:
            if not ctx.split:
                call(_STEP_ALL, None)
            else:
                call(_STEP_A, None)
                gathered, _ = gather_ranks(ws[ctx.off_sums:ctx.off_sums + 48].view(torch.float64), ctx.group,
                                           'bn-bwd sums all-gather')
                call(_STEP_B, gathered)
        grads = [None] * 7
        if need_p:
            for i, ((_, off, n), shape) in enumerate(zip(PARAM_LAYOUT, ctx.shapes)):
                if ctx.needs_input_grad[1 + i]:
                    grads[i] = gp[off:off + n].view(shape)
        return (None, *grads, None, None, gadd, None, None, None, None, None, None, None, None)


def isp_fused(raw, module, bn_mode=BN_NONE, group=None, epilogue=None):
    """fused forward of a ParametrizedProcessing-shaped module (parameters by the reference's names).  epilogue =
    (hflip, vflip, k): the output leaves the kernels as rot90^k(vflip(hflip(out)))."""
    bn = module.batch_norm
    return _IspFused.apply(raw, module.black_level, module.white_balance, module.colour_correction,
                           module.gamma_correct, module.debayer.weight, module.sharpening_filter.weight,
                           module.gaussian_blur.weight, module.M_RGB_2_YUV, module.M_YUV_2_RGB,
                           module.additive_layer, bn_mode, bn, bn.eps if bn is not None else 1e-5,
                           bn.momentum if bn is not None else None, group, getattr(module, 'raw_bits', 16),
                           torch.is_grad_enabled(), epilogue)


# --------------------------------------------------------------------------------------------------
# static pipeline (processing(), pipeline_numpy.py:70-141), batched
# --------------------------------------------------------------------------------------------------
_DEBAYER = {'bilinear': 0, 'malvar2004': 1, 'menon2007': 2}
_SHARPEN = {'sharpening_filter': 1, 'unsharp_masking': 2}
_DENOISE = {'gaussian_denoising': 1, 'median_denoising': 2, 'fft_denoising': 3}


STATIC_OPTION_DEFAULTS = dict(sharp_radius=1.0, sharp_amount=1.0, gaussian_sigma=0.5, fft_fraction=0.3, median_kernel_size=3)


def static_pipeline(raw, camera_parameters, debayer='bilinear', sharpening='sharpening_filter',
                    denoising='gaussian_denoising', gamma=2.2, bits=16, mean_std=None, sharp_radius=1.0, sharp_amount=1.0,
                    median_kernel_size=3, gaussian_sigma=0.5, fft_fraction=0.3):
    """(B,H,W) raw on the GPU -> (B,3,H,W) float32, numpy semantics of the reference.

    sharp_radius, sharp_amount, median_kernel_size, gaussian_sigma, fft_fraction: processing()'s numeric arguments
    (pipeline_numpy.py:70-73, used at :117-122), launch arguments of the kernels (r2l_static_fwd_opts).  What the kernels'
    windows hold bounds them: gaussian_sigma in (0, 0.625), sharp_radius in (0, 1.125), fft_fraction in [0, 0.5],
    median_kernel_size 3 or 5 (5: the chain runs as luma-plane passes, W % 4 == 0) -- other values of an option the chain USES raise R2LError with the reason; an option of a stage the
    chain does not run is ignored like the reference's if-chains ignore it.

    mean_std: six host floats (mean[3], std[3]) -- the T.Normalize(mean, std) that train.py:157-171 composes
    behind RawProcessingPipeline, applied inside the kernels' stores (float32 subtraction and division).

    The black level is removed in the arithmetic of the frames' dtype, as the reference's in-place
    remove_blacklv does (pipeline_numpy.py:152-158): float32 frames -- what its datasets deliver
    (utils/dataset_utils.py:18-26, dataset.py:86-87) -- and 16-bit containers (divided by 2**bits - 1 in
    float32 like dataset.py:87) subtract the float32-rounded black level in float32; float64 frames (a DNG's
    uint16 / (2**bits - 1)) stay float64.  Everything after the demosaic is float64 in both cases.

    Like the reference's if-chains (pipeline_numpy.py:110-122) a sharpening / denoising string that
    names no algorithm means "skip that stage"; algorithms the reference has but this library does not
    build (tv_chambolle / tv_bregman / bilateral denoising) raise instead of silently differing.  debayer='menon2007'
    (pipeline_numpy.py:96-97) runs as float64 plane passes (W % 4 == 0): a third-party algorithm restated from its published
    source, parity unpinned like Malvar2004's."""
    assert raw.ndim == 3, f"needs dims (B, H, W), got {raw.shape}"
    f64 = raw.dtype == torch.float64
    if f64:
        if raw.shape[-1] % 4:
            raise ValueError('float64 frames need W % 4 == 0')
        raw, denom = (raw if raw.is_contiguous() else raw.contiguous()), None
    else:
        raw, denom = _raw_arg(raw, bits)
    known_sharp = {'sharpening_filter', 'unsharp_masking'}
    known_den = {'median_denoising', 'gaussian_denoising', 'fft_denoising', 'tv_chambolle', 'tv_bregman',
                 'bilateral'}
    if debayer not in _DEBAYER:
        raise NotImplementedError(f"debayer '{debayer}' is not built for the GPU (have: {list(_DEBAYER)})")
    if sharpening in known_sharp and sharpening not in _SHARPEN:
        raise NotImplementedError(f"sharpening '{sharpening}' is not built for the GPU")
    if denoising in known_den and denoising not in _DENOISE:
        raise NotImplementedError(f"denoising '{denoising}' is not built for the GPU")
    bl, wb, ccm = camera_parameters
    cam = (ctypes.c_double * 16)(*[float(v) for v in list(bl) + list(wb) + list(ccm)])
    B, H, W = raw.shape
    lib, stream = _lib.library_for(raw)
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=raw.device)
    codes = (_DEBAYER[debayer], _SHARPEN.get(sharpening, 0), _DENOISE.get(denoising, 0))
    opts = dict(sharp_radius=sharp_radius, sharp_amount=sharp_amount, gaussian_sigma=gaussian_sigma,
                fft_fraction=fft_fraction, median_kernel_size=median_kernel_size)
    custom = opts != STATIC_OPTION_DEFAULTS
    frames = 2 if f64 else (0 if denom is None else 1)
    if custom:
        ov = (ctypes.c_double * 5)(float(sharp_radius), float(sharp_amount), float(gaussian_sigma), float(fft_fraction),
                                   float(median_kernel_size))         # R2L_SOPT_* order (include/r2l_isp.h)
        nws = lib.r2l_static_workspace_bytes_opts(frames, B, H, W, *codes, ov)     # (a 5x5 median runs as plane passes)
    else:
        nws = (lib.r2l_static_workspace_bytes_f64 if f64 else lib.r2l_static_workspace_bytes)(B, H, W, *codes)
    ws = torch.empty(nws, dtype=torch.uint8, device=raw.device) if nws else None      # 0: single-launch chains
    tail = (ptr(out), B, H, W, cam, *codes, float(gamma), ptr(ws), nws, stream)
    if custom:
        ms = (ctypes.c_float * 6)(*[float(v) for v in mean_std]) if mean_std is not None else None
        lib.check(lib.r2l_static_fwd_opts(ptr(raw), frames, float(denom or 1.0), ptr(out), B, H, W, cam, *codes,
                                          float(gamma), ov, ms, ptr(ws), nws, stream), 'r2l_static_fwd_opts')
    elif mean_std is not None:
        ms = (ctypes.c_float * 6)(*[float(v) for v in mean_std])
        lib.check(lib.r2l_static_fwd_norm(ptr(raw), frames, float(denom or 1.0), ptr(out), B, H, W, cam, *codes,
                                          float(gamma), ms, ptr(ws), nws, stream), 'r2l_static_fwd_norm')
    elif f64:
        lib.check(lib.r2l_static_fwd_f64(ptr(raw), *tail), 'r2l_static_fwd_f64')
    elif denom is None:
        lib.check(lib.r2l_static_fwd(ptr(raw), *tail), 'r2l_static_fwd')
    else:
        lib.check(lib.r2l_static_fwd_u16(ptr(raw), denom, *tail), 'r2l_static_fwd_u16')
    return out


def normalize(rgb, mean_std):
    """(x - mean[c]) / std[c] on (B,3,H,W): torchvision's T.Normalize as applied after the static pipeline
    (train.py:157-171); mean_std = float32[6] on the device of rgb."""
    rgb = _f32c(rgb, 'rgb')
    B, C, H, W = rgb.shape
    assert C == 3
    lib, stream = _lib.library_for(rgb)
    ms = _f32c(mean_std.to(rgb.device), 'mean_std')
    out = torch.empty_like(rgb)
    lib.check(lib.r2l_stage_point(8, ptr(rgb), None, ptr(ms), None, None, ptr(out), None, None, 0, B, H, W,
                                  stream), 'r2l_stage_point(normalize)')
    return out
