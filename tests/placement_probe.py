"""Is the apply pass's bimodal time (66-67 / 70-72 us between processes on one box) a matter of WHERE its buffers sit?  One process, the same step
with a dummy allocation of varying size made first (shifts the addresses of workspace and output), HIP events per kernel; prints the addresses."""
import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault('R2L_LIB_PATH', os.path.join(HERE, '_build', 'libr2l_isp_hooks.so'))
sys.path.insert(0, os.path.dirname(HERE))
from raw2logit_amd import _lib, cameras
from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
lib = _lib.device_library()
B, S = 64, 512
m = ParametrizedProcessing(cameras.DRONE, track_stages=False, batch_norm_output=True).to('cuda').train()
for pad_mb in (0, 2, 6, 14, 30, 62, 126, 1, 3, 7, 34, 66, 0):
    torch.cuda.empty_cache()
    dummy = torch.empty(max(pad_mb, 0) << 20, dtype=torch.uint8, device='cuda') if pad_mb else None
    raw = torch.rand(B, S, S, device='cuda')
    cot = torch.randn(B, 3, S, S, device='cuda')
    y = None

    def step():
        global y
        for p in m.parameters():
            p.grad = None
        y = m(raw)
        y.backward(cot)
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    lib.r2l_timing_enable(1)
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14)
    lib.r2l_timing_report(buf, len(buf))
    lib.r2l_timing_enable(0)
    k = {l.split()[0].replace('r2l_launch_', '').replace('_kernel', ''): 1e3 * float(l.split()[2]) / int(l.split()[1]) for l in buf.value.decode().splitlines()}
    ws = y.grad_fn.ws if y.grad_fn is not None and hasattr(y.grad_fn, 'ws') else None
    print(f'pad {pad_mb:4d} MB  raw {raw.data_ptr():#x} out {y.data_ptr():#x} cot {cot.data_ptr():#x} ws {ws.data_ptr() if ws is not None else 0:#x}  '
          + ' '.join(f'{a}={v:.1f}' for a, v in sorted(k.items())) + f'  sum {sum(k.values()):.1f}', flush=True)
    del raw, cot, y, dummy
