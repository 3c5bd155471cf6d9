"""Host-side cost of one training step of the fused module (tiny frames: the GPU work is negligible)."""
import sys, os, time, cProfile, pstats, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2logit_amd import cameras
from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
dev = 'cuda'
raw = torch.rand((2, 64, 64), device=dev)
cot = torch.randn((2, 3, 64, 64), device=dev)
m = ParametrizedProcessing(cameras.DRONE, batch_norm_output=True).to(dev).train()
params = list(m.parameters())


def step():
    for p in params:
        p.grad = None
    m(raw).backward(cot)


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print('us per step (launch-bound):', (time.perf_counter() - t0) / 200 * 1e6)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
print(s.getvalue()[:3500])

# the same two C-ABI calls without autograd / module / allocation: what the launches themselves cost the host
import ctypes
from raw2logit_amd import _lib, functional as F_
from raw2logit_amd._lib import ptr
lib, stream = _lib.library_for(raw)
B, H, W = raw.shape
nws, _, _ = F_._step_layout(lib, B, H, W)
ws = torch.empty(nws, dtype=torch.uint8, device=dev)
out = torch.empty((B, 3, H, W), device=dev)
gp = torch.empty(132, device=dev)
ps = [m.black_level, m.white_balance, m.colour_correction, m.gamma_correct, m.debayer.weight,
      m.sharpening_filter.weight, m.gaussian_blur.weight, m.M_RGB_2_YUV, m.M_YUV_2_RGB]
table = (ctypes.c_void_p * 9)(*[p.data_ptr() for p in ps])
bn = m.batch_norm


def raw_step():
    lib.r2l_isp_step_fwd(ptr(raw), 0, 1.0, table, None, 1, ptr(bn.running_mean), ptr(bn.running_var),
                         ptr(bn.num_batches_tracked), 1e-5, 0.1, ptr(out), ptr(ws), nws, B, H, W, 1, 0, None, stream)
    lib.r2l_isp_step_bwd(ptr(raw), 0, 1.0, None, ptr(cot), ptr(out), ptr(gp), None, 1, ptr(ws), nws, B, H, W, 1, 0,
                         None, stream)


for _ in range(20):
    raw_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    raw_step()
torch.cuda.synchronize()
print('us per step, the two C-ABI calls alone (6 launches):', (time.perf_counter() - t0) / 500 * 1e6)
