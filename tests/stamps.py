import os, torch
os.environ['R2L_LIB_PATH'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_build', 'lib_stamps.so')
import sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import isp_oracle as orc
from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
raw = torch.from_numpy(orc.synth_raw(64, 512, 512, seed=0)).cuda()
m = ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=False).cuda()
with torch.no_grad():
    for _ in range(3): y = m(raw)
    torch.cuda.synchronize()
st = y.flatten()[:512 * 8].view(512, 8).double().cpu()
names = ['load', 'Y', 'YP', 'fill', 'pixels']
print('per-workgroup s_memtime ticks per phase, mean over 512 WGs (8 tiles each); 100 MHz ticks? ->', st[:, :5].sum(1).mean().item())
for i, n in enumerate(names): print(f'  {n:8s} mean {st[:, i].mean().item():10.0f}  min {st[:, i].min().item():10.0f}  max {st[:, i].max().item():10.0f}')
