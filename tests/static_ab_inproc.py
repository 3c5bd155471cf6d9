"""In-process A/B of static short-chain builds: the rate depends on where a process's buffers land in HBM
(+-4 % between processes), so variants are compared on the SAME buffers, interleaved.
usage: python tests/static_ab_inproc.py name=path.so ...   (the product build is always included as 'default')"""
import os, sys, ctypes, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from raw2logit_amd import _lib, cameras
libs = {'default': _lib.LIB_PATH}
for a in sys.argv[1:]:
    k, v = a.split('=')
    libs[k] = os.path.join(HERE, '_build', v) if not os.path.isabs(v) else v
deb = int(os.environ.get('DEB', '0'))
sharp, den = int(os.environ.get('SHARP', '0')), int(os.environ.get('DENOISE', '0'))   # (1, 1: the train.py default chain)
B, S = int(os.environ.get('B', '256')), int(os.environ.get('S', '1024'))
dev = torch.device('cuda', 0)
n = B * S * S
raw = torch.randint(0, 4096, (n,), device=dev, dtype=torch.int32).to(torch.float32) / 4095.0
out = torch.empty(3 * n, dtype=torch.float32, device=dev)
bl, wb, ccm = cameras.DRONE
cam = (ctypes.c_double * 16)(*[float(v) for v in list(bl) + list(wb) + list(ccm)])
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
fns = {}
for k, path in libs.items():
    h = ctypes.CDLL(path)
    h.r2l_static_fwd.restype = ctypes.c_int
    h.r2l_static_fwd.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    fns[k] = h.r2l_static_fwd
def run(f, reps=10):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for r in range(reps):
        e0.record()
        rc = f(raw.data_ptr(), out.data_ptr(), B, S, S, cam, deb, sharp, den, 2.2, None, 0, stream)
        e1.record(); torch.cuda.synchronize()
        assert rc == 0
        if r >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
bands = [b for b in os.environ.get('BANDS', '0').split()]       # R2L_STREAM_BANDS settings to sweep (0 = default)
res = {(k, b): [] for k in fns for b in bands}
for rnd in range(3):
    for b in bands:
        if b == '0':
            os.environ.pop('R2L_STREAM_BANDS', None)
        else:
            os.environ['R2L_STREAM_BANDS'] = b
        for k, f in fns.items():
            res[(k, b)].append(run(f))
for (k, b), v in res.items():
    print('%-10s bands %-3s median us per round: %s   %6.1f GB/s' % (k, b, ' '.join('%7.1f' % x for x in v), 16.0 * n / min(v) / 1e3))
# the builds must agree bit for bit (same arithmetic, different loops): compare every output with the first build's
ref = None
for k, f in fns.items():
    out.fill_(-7.0)
    assert f(raw.data_ptr(), out.data_ptr(), B, S, S, cam, deb, sharp, den, 2.2, None, 0, stream) == 0
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    else:
        print('%-10s max |out - %s| = %.3e' % (k, next(iter(fns)), float((out - ref).abs().max())))
