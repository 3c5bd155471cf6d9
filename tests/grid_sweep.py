"""Grid caps of the persistent kernels of the step at 64 x 512 x 512 (diagnostic build, one process): R2L_GRID_BNR (bn_reduce),
R2L_GRID_BWD1 (B1's plane pass and the blur pass), R2L_GRID_BWD2 (the sums pass)."""
import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault('R2L_LIB_PATH', os.path.join(HERE, '_build', 'libr2l_isp_hooks.so'))
sys.path.insert(0, os.path.dirname(HERE))
from raw2logit_amd import _lib, cameras
from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
lib = _lib.device_library()
dev = 'cuda'
B, S = 64, 512
raw = torch.rand(B, S, S, device=dev)
cot = torch.randn(B, 3, S, S, device=dev)
m = ParametrizedProcessing(cameras.DRONE, track_stages=False, batch_norm_output=True).to(dev).train()


def step():
    for p in m.parameters():
        p.grad = None
    m(raw).backward(cot)


def kernels(n=30):
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    lib.r2l_timing_enable(1)
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14)
    lib.r2l_timing_report(buf, len(buf))
    lib.r2l_timing_enable(0)
    return {ln.split()[0].replace('r2l_launch_', '').replace('_kernel', ''): 1e3 * float(ln.split()[2]) / int(ln.split()[1])
            for ln in buf.value.decode().splitlines()}


for _ in range(200):
    step()
for var, vals, show in (('R2L_GRID_BNR', (0, 256, 384, 512, 768, 1024, 0), 'bn_reduce'),
                        ('R2L_GRID_BWD1', (0, 256, 384, 448, 512, 0), 'bwd1_plane'),
                        ('R2L_GRID_BWD2', (0, 512, 640, 704, 768, 0), 'bwd2_sums')):
    for v in vals:
        if v:
            os.environ[var] = str(v)
        else:
            os.environ.pop(var, None)
        k = kernels()
        print(f'{var}={v or "default":>7}  {show} {k[show]:6.1f} us   step kernels {sum(k.values()):6.1f} us', flush=True)
    os.environ.pop(var, None)
