"""ctypes binding of libr2l_isp.so (C ABI: include/r2l_isp.h).

The product has exactly one compute path and one dispatch target: the HIP kernels for gfx950.  If the
shared library is missing, is not the gfx950 build, or a tensor is not on the GPU, every op raises --
there is no PyTorch / numpy / CPU fallback and no hook for one in this package.  (The CPU-only test
suite runs the kernel SOURCE through a host emulation; the code that serves CPU tensors with it lives
in tests/emul_hook.py and patches this module from the outside.)"""
import ctypes
import os
import subprocess
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get('R2L_LIB_PATH') or os.path.join(_HERE, 'libr2l_isp.so')   # override: kernel A/B builds
CSRC = os.path.join(_HERE, 'csrc')

R2L_P_COUNT = 150
R2L_P_NTRAIN = 132
R2L_F_STATS_ONLY = 1
R2L_F_FOLDED_VALID = 2

_c_float_p = ctypes.c_void_p   # raw addresses from tensor.data_ptr()
_SIGNATURES = {
    'r2l_abi_version': (ctypes.c_int, []),
    'r2l_last_error': (ctypes.c_char_p, []),
    'r2l_is_device_build': (ctypes.c_int, []),
    'r2l_timing_enable': (None, [ctypes.c_int]),
    'r2l_timing_report': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t]),
    'r2l_raw2rgb_fwd': (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'r2l_raw2rgb_bwd_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'r2l_raw2rgb_bwd': (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'r2l_isp_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'r2l_isp_fwd': (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'r2l_bn_finalize': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_float_p, ctypes.c_void_p, _c_float_p,
                                       _c_float_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_double,
                                       ctypes.c_void_p]),
    'r2l_isp_fwd_stats_bn': (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p, _c_float_p,
                                            ctypes.c_void_p, _c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_double,
                                            ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'r2l_isp_fwd_stats_bn_u16': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, _c_float_p, _c_float_p,
                                                ctypes.c_void_p, _c_float_p, ctypes.c_void_p, _c_float_p, _c_float_p,
                                                ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                                ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_void_p]),
    'r2l_bn_bwd_means': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, _c_float_p, ctypes.c_void_p]),
    'r2l_bn_bwd_reduce': (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p, _c_float_p,
                                         ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'r2l_isp_bwd': (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                   _c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_size_t,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p]),
    'r2l_isp_step_offset': (ctypes.c_size_t, [ctypes.c_int] * 4),
    'r2l_isp_step_fwd': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.POINTER(ctypes.c_void_p),
                                        _c_float_p, ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_double,
                                        ctypes.c_double, _c_float_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_void_p]),
    'r2l_isp_step_bwd': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, _c_float_p, _c_float_p, _c_float_p,
                                        _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p]),
    'r2l_additive_bwd': (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'r2l_isp_fwd_u16': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, _c_float_p, _c_float_p, _c_float_p,
                                       _c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'r2l_isp_bwd_u16': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, _c_float_p, _c_float_p, _c_float_p,
                                       _c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'r2l_raw2rgb_fwd_u16': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, _c_float_p, _c_float_p] +
                            [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    'r2l_static_fwd_u16': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_float, _c_float_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t,
                                          ctypes.c_void_p]),
    'r2l_augment': (ctypes.c_int, [_c_float_p, _c_float_p] + [ctypes.c_int] * 7 + [ctypes.c_void_p]),
    'r2l_add_noise': (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_float, _c_float_p, ctypes.c_size_t,
                                     ctypes.c_void_p]),
    'r2l_add_noise_philox': (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64,
                                            ctypes.c_size_t, ctypes.c_void_p]),
    'r2l_aux_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 4),
    'r2l_ssim_fwd': (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t] +
                     [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    'r2l_ssim_bwd': (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p,
                                    ctypes.c_size_t] + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    'r2l_l2_fwd': (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                  ctypes.c_size_t, ctypes.c_void_p]),
    'r2l_l2_bwd': (ctypes.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_size_t, ctypes.c_void_p]),
    'r2l_stage_workspace_bytes': (ctypes.c_size_t, []),
    'r2l_stage_conv33_fwd': (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    'r2l_stage_conv33_bwd': (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_void_p, ctypes.c_size_t] +
                             [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    'r2l_stage_mix3_fwd': (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    'r2l_stage_mix3_bwd': (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_void_p, ctypes.c_size_t] +
                           [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    'r2l_stage_pconv_fwd': (ctypes.c_int, [_c_float_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    'r2l_stage_pconv_bwd': (ctypes.c_int, [_c_float_p] * 5 + [ctypes.c_int] * 2 +
                            [ctypes.c_void_p, ctypes.c_size_t] + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    'r2l_stage_point': (ctypes.c_int, [ctypes.c_int] + [_c_float_p] * 7 + [ctypes.c_void_p, ctypes.c_size_t] +
                        [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    'r2l_static_workspace_bytes': (ctypes.c_size_t, [ctypes.c_int] * 6),
    'r2l_static_workspace_bytes_f64': (ctypes.c_size_t, [ctypes.c_int] * 6),
    'r2l_static_fwd_f64': (ctypes.c_int, [ctypes.c_void_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t,
                                          ctypes.c_void_p]),
    'r2l_static_fwd_norm': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, _c_float_p, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                           ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_size_t,
                                           ctypes.c_void_p]),
    'r2l_static_workspace_bytes_opts': (ctypes.c_size_t, [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_double)]),
    'r2l_static_fwd_opts': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, _c_float_p, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                           ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_size_t,
                                           ctypes.c_void_p]),
    'r2l_static_fwd': (ctypes.c_int, [_c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t,
                                      ctypes.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class R2LError(RuntimeError):
    pass


class Library:
    def __init__(self, path):
        if not os.path.exists(path):
            raise R2LError(
                f'{path} not found: the HIP extension is not built. Run '
                f'`python -c "import __graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950). '
                f'raw2logit_amd has no CPU or PyTorch fallback.')
        self.path = path
        self.cdll = ctypes.CDLL(path)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(self.cdll, name)          # AttributeError if a declared symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
        self.is_device = bool(self.cdll.r2l_is_device_build())
        self._check_build()

    def _check_build(self):
        if not self.is_device:
            raise R2LError(f'{self.path} is the test-only host emulation, not the gfx950 build')

    def check(self, code, what):
        if code != 0:
            msg = self.cdll.r2l_last_error()
            raise R2LError(f'{what} failed ({code}): {msg.decode() if msg else "?"}')

    def __getattr__(self, name):
        return getattr(self.cdll, name)


_DEVICE_LIB = None


def device_library():
    global _DEVICE_LIB
    if _DEVICE_LIB is None:
        _DEVICE_LIB = Library(LIB_PATH)
    return _DEVICE_LIB


def library_for(t):
    """the library that may touch tensor `t`, plus the stream handle to enqueue on."""
    if t.is_cuda:
        lib = device_library()
        if t.device.index is not None and t.device.index != torch.cuda.current_device():
            # kernels are launched on the calling thread's current HIP device (one process per GPU is the
            # supported layout: torch.cuda.set_device(local_rank) at start-up)
            raise R2LError(f'the tensor lives on {t.device} but the current device is cuda:'
                           f'{torch.cuda.current_device()}; use torch.cuda.set_device / torch.cuda.device(...)')
        return lib, ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    raise R2LError('raw2logit_amd runs on MI355X only: the tensor is on the CPU and there is no CPU '
                   'path (move it to the GPU: tensor.cuda())')


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def hipcc_command(out_path=LIB_PATH, extra=()):
    # -fno-slp-vectorize: packed f32 is written out by hand where it pays (r2l_p2 pairs of adjacent pixels); the
    # SLP vectoriser's own pairing adds register shuffles and ~100 live VGPRs to the backward kernels
    return ['hipcc', '-O3', '-std=c++17', '-fno-slp-vectorize', '--offload-arch=gfx950', '-shared', '-fPIC',
            *extra, os.path.join(CSRC, 'r2l_api.hip'), '-o', out_path, '-ldl']    # (rocFFT: opened at first fft_denoising call)


def source_digest():
    """12 hex digits over the kernel sources and the C-ABI header: the build tag profiling scripts write into
    profiles/*pmc_traffic*.json (`_meta.library_digest`) and bench.py reports beside the traffic it reads from them"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + [os.path.join(REPO_ROOT, 'include', 'r2l_isp.h')]:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode())
            with open(path, 'rb') as fh:
                h.update(fh.read())
    return h.hexdigest()[:12]


def build_device_library(verbose=True, out_path=LIB_PATH, extra=()):
    """compile libr2l_isp.so in-tree for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(REPO_ROOT, 'include', 'r2l_isp.h')]
    if os.path.exists(out_path) and all(os.path.getmtime(out_path) >= os.path.getmtime(s) for s in srcs):
        return out_path
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    tmp = out_path + f'.{os.getpid()}.tmp'
    cmd = hipcc_command(tmp, extra)
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(tmp, out_path)
    return out_path
