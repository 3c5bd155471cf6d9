import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ['R2L_LIB_PATH'] = '/root/repo/tests/_build/libr2l_isp_hooks.so'
from oracle import isp_oracle as orc
import parity_checks as pc
from raw2logit_amd import _lib
dev = torch.device('cuda')
np.set_printoptions(linewidth=220, precision=4, suppress=False)
for shape in ((1, 24, 256),):
    B, H, W = shape
    P = orc.IspParams(orc.DRONE_CAMERA_PARAMS); P.perturb(19)
    raw = torch.from_numpy(orc.synth_raw(B, H, W, seed=3, kind='scene')).to(dev)
    cot = torch.from_numpy(np.random.default_rng(5).standard_normal((B, 3, H, W)).astype(np.float32)).to(dev)
    case = dict(camera='drone', track=False, additive=False, training=True, bn=False)
    lib = _lib.device_library()
    lib.cdll.r2l_test_debug_offset.restype = ctypes.c_size_t
    off = lib.cdll.r2l_test_debug_offset(B, H, W) + 4 * 16 * 2048
    def run(env):
        m = pc.make_module(case, P, dev)
        os.environ.update(env)
        try:
            y = m(raw)
            ws = y.grad_fn.ws
            (y * cot).sum().backward()
        finally:
            for k in env: del os.environ[k]
        torch.cuda.synchronize()
        return ws[off:off + 8 * 155].view(torch.float64).cpu().numpy().copy()
    ref = run({'R2L_BWD2_TILED': '1'})
    g = run({})
    b2r, b2g = ref[106:], g[106:]
    print('GSHARP ref', b2r[:9]); print('GSHARP got', b2g[:9])
    for par in range(4):
        print('GAY par', par, 'ref', b2r[9 + par * 9:18 + par * 9]); print('GAY par', par, 'got', b2g[9 + par * 9:18 + par * 9])
    print('SY ref', b2r[45:49]); print('SY got', b2g[45:49])
    print('B1 max rel diff', np.abs(ref[:106] - g[:106]).max() / np.abs(ref[:106]).max())
