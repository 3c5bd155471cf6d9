"""A/B of library builds at 64 x 512 x 512, alternating, each in its own process: python tests/lib_ab.py libA.so libB.so ...  (paths under tests/_build)"""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CODE = r'''
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(%r))
from raw2logit_amd import _lib, cameras
from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
lib = _lib.device_library()
B, S = 64, 512
raw = torch.rand(B, S, S, device='cuda'); cot = torch.randn(B, 3, S, S, device='cuda')
m = ParametrizedProcessing(cameras.DRONE, track_stages=False, batch_norm_output=True).to('cuda').train()
def step():
    for p in m.parameters(): p.grad = None
    m(raw).backward(cot)
for _ in range(300): step()
torch.cuda.synchronize()
lib.r2l_timing_enable(1)
for _ in range(40): step()
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 14); lib.r2l_timing_report(buf, len(buf))
print(' '.join('%%s=%%.1f' %% (l.split()[0].replace('r2l_launch_','').replace('_kernel',''), 1e3*float(l.split()[2])/int(l.split()[1])) for l in buf.value.decode().splitlines()))
''' % HERE
for rep in range(int(os.environ.get('REPS', '2'))):
    for name in sys.argv[1:]:
        e = dict(os.environ, R2L_LIB_PATH=os.path.join(HERE, '_build', name))
        r = subprocess.run([sys.executable, '-c', CODE], env=e, capture_output=True, text=True)
        print('%-28s' % name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
