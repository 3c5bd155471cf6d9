"""Band height of the two-wavefronts-per-SIMD plane passes (B1's plane pass, the blur pass) at several batch sizes, one process:
R2L_BP_BAND / R2L_HB_BAND of the diagnostic build are read at every launch.  R2L_LIB_PATH=tests/_build/libr2l_isp_hooks.so"""
import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault('R2L_LIB_PATH', os.path.join(HERE, '_build', 'libr2l_isp_hooks.so'))
sys.path.insert(0, os.path.dirname(HERE))
from raw2logit_amd import _lib, cameras
from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
lib = _lib.device_library()
dev = 'cuda'


def kernels(step, n=20):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    lib.r2l_timing_enable(1)
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14)
    lib.r2l_timing_report(buf, len(buf))
    lib.r2l_timing_enable(0)
    out = {}
    for ln in buf.value.decode().splitlines():
        name, c, ms = ln.split()
        out[name.replace('r2l_launch_', '').replace('_kernel', '')] = 1e3 * float(ms) / int(c)
    return out


for B, S in ((16, 512), (32, 512), (48, 512), (64, 256), (128, 256), (256, 512), (64, 512)):
    raw = torch.rand(B, S, S, device=dev)
    cot = torch.randn(B, 3, S, S, device=dev)
    m = ParametrizedProcessing(cameras.DRONE, track_stages=False, batch_norm_output=True).to(dev).train()

    def step():
        for p in m.parameters():
            p.grad = None
        m(raw).backward(cot)
    for c in (0, 6, 12, 18, 24, 30, 36, 48):
        for k in ('R2L_BP_BAND', 'R2L_HB_BAND'):
            if c:
                os.environ[k] = str(c)
            else:
                os.environ.pop(k, None)
        k = kernels(step)
        print(f'{B:4d}x{S}^2  band {c or "default":>7}  bwd1_plane {k.get("bwd1_plane", 0):7.1f}  bwd1_blur_hp {k.get("bwd1_blur_hp", 0):7.1f}  '
              f'sum {k.get("bwd1_plane", 0) + k.get("bwd1_blur_hp", 0):7.1f} us', flush=True)
    del raw, cot, m
    torch.cuda.empty_cache()
