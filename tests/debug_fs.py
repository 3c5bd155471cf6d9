import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import isp_oracle as orc
from raw2logit_amd.processing import pipeline_torch as ppt
np.set_printoptions(linewidth=250, precision=2, suppress=False)
for (B, H, W, bn) in ((1, 16, 16, False), (1, 16, 16, True), (1, 40, 264, False), (2, 70, 520, False)):
    raw = orc.synth_raw(B, H, W, seed=3, kind='scene')
    m = ppt.ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=bn).cuda().train()
    with torch.no_grad():
        y = m(torch.from_numpy(raw).cuda()).cpu().numpy()
    P = orc.IspParams(orc.DRONE_CAMERA_PARAMS, dtype=np.float64)
    o, _, c = orc.parametrized_forward(raw, P, bn=dict(training=True, running_mean=np.zeros(3), running_var=np.ones(3)) if bn else None)
    e = np.abs(y - o)
    print('shape', (B, H, W), 'bn', bn, 'max err', e.max(), 'at', np.unravel_index(e.argmax(), e.shape))
    bad = e.max(axis=(0, 1)) > 1e-4
    print(' bad rows', np.where(bad.any(axis=1))[0][:40], ' bad cols', np.where(bad.any(axis=0))[0][:40])
