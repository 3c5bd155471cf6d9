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
