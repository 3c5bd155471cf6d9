"""What the passes of the 64 x 512 x 512 step inherit from their predecessors, and XCD windows of the band passes -- one process,
diagnostic build (tests/_build/libr2l_isp_hooks.so reads its overrides at every launch):
  R2L_EXP_FLUSH=<kernel>   a 768 MB read-modify-write pass in front of that kernel (L2s and the 256 MB memory-side cache evicted)
  R2L_EXP_TWICE=<kernel>   an untimed launch of the same kernel in front of the timed one (everything it touches is recent)
  R2L_XCD_FS / _FA / _BP / _HB / _B2S = M   M neighbouring workgroups per XCD (r2l_xcd_window)
HIP events per kernel, us."""
import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
os.environ.setdefault('R2L_LIB_PATH', os.path.join(TESTS, '_build', 'libr2l_isp_hooks.so'))
sys.path.insert(0, os.path.dirname(TESTS))
from raw2logit_amd import _lib, cameras
from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
lib = _lib.device_library()
dev = 'cuda'
B, S = (int(x) for x in os.environ.get('SHAPE', '64x512').split('x'))
raw = torch.rand(B, S, S, device=dev)
cot = torch.randn(B, 3, S, S, device=dev)
m = ParametrizedProcessing(cameras.DRONE, track_stages=False, batch_norm_output=True).to(dev).train()
ORDER = ['pack_fold', 'fwd_stream_stats_w2', 'fwd_luma', 'fwd_stats', 'fwd_apply', 'bn_reduce', 'bwd1_plane', 'bwd1_blur_hp', 'bwd2_sums']


def step():
    for p in m.parameters():
        p.grad = None
    m(raw).backward(cot)


def kernels(n=24):
    for _ in range(6):
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


def show(tag, k):
    names = [n for n in ORDER if n in k]
    print(f'{tag:34s} ' + ' '.join(f'{n}={k[n]:6.1f}' for n in names) + f'  sum {sum(k.values()):6.1f}', flush=True)


for _ in range(300):
    step()
base = kernels()
show('baseline', base)
names = [n for n in ORDER if n in base and n != 'pack_fold']
print('--- a flush in front of ONE kernel (its own column is what it costs cold)')
for n in names:
    os.environ['R2L_EXP_FLUSH'] = 'r2l_launch_' + n
    show('flush before ' + n, kernels())
os.environ['R2L_EXP_FLUSH'] = 'all'
show('flush before every kernel', kernels())
os.environ.pop('R2L_EXP_FLUSH')
show('baseline', kernels())
print('--- the same kernel launched twice, second launch timed (its own column is what it costs warm)')
for n in names:
    os.environ['R2L_EXP_TWICE'] = 'r2l_launch_' + n
    show('twice ' + n, kernels())
os.environ.pop('R2L_EXP_TWICE')
show('baseline', kernels())
print('--- XCD windows: M neighbouring workgroups per XCD')
for var in ('R2L_XCD_FS', 'R2L_XCD_FA', 'R2L_XCD_BP', 'R2L_XCD_HB', 'R2L_XCD_B2S'):
    for mval in (2, 4, 8, 16):
        os.environ[var] = str(mval)
        show(f'{var}={mval}', kernels())
    os.environ.pop(var)
    show('baseline', kernels())
for mval in (2, 4, 8):
    for var in ('R2L_XCD_FS', 'R2L_XCD_FA', 'R2L_XCD_BP', 'R2L_XCD_HB', 'R2L_XCD_B2S'):
        os.environ[var] = str(mval)
    show(f'all windows = {mval}', kernels())
for var in ('R2L_XCD_FS', 'R2L_XCD_FA', 'R2L_XCD_BP', 'R2L_XCD_HB', 'R2L_XCD_B2S'):
    os.environ.pop(var)
show('baseline', kernels())
