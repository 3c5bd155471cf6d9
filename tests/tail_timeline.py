"""Diagnostic: the stations of each launch's tail (end of the item loop -> partials stored -> tickets -> tree levels -> unfold /
BatchNorm bookkeeping) as the launch's LAST workgroup passes them, s_memrealtime (100 MHz), from a -DR2L_EXP_STAMPS
-DR2L_TEST_HOOKS build:  R2L_STAMPS_LIB=lib_tl.so python tests/tail_timeline.py"""
import os, sys, ctypes, torch
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ['R2L_LIB_PATH'] = os.path.join(HERE, '_build', os.environ.get('R2L_STAMPS_LIB', 'lib_tl.so'))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from oracle import isp_oracle as orc
from raw2logit_amd import _lib
import parity_checks as pc
B, H, W = [int(x) for x in os.environ.get('SHAPE', '64x512x512').split('x')]
dev = torch.device('cuda')
raw = torch.from_numpy(orc.synth_raw(B, H, W, seed=0, kind='uniform')).to(dev)
m = pc.make_module(dict(camera='drone', track=False, additive=False, training=True, bn=True), orc.IspParams(orc.DRONE_CAMERA_PARAMS), dev)
lib = _lib.device_library()
cot = None
names = {0: 'statistics pass', 10: 'bn_reduce', 20: "B2's sums pass"}
# station -> slot offset (r2l_common.h: R2L_TAILST), in the order the last workgroup passes them
STATIONS = [('partials stored', 1), ('ticket 1', 2), ('level 1: group partial stored', 4), ('ticket 2', 5), ('level 2 summed', 6),
            ('unfold: parameters in LDS', 3), ('unfold: folded tables', 8), ('unfold: black-level parts', 9),
            ('unfold / bookkeeping done', 7)]
acc = {}
for rep in range(int(os.environ.get('REPS', '12'))):
    y = m(raw)
    if cot is None:
        cot = torch.randn_like(y)
    (y * cot).sum().backward()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    assert lib.cdll.r2l_test_tail_stamps(buf) == 0
    if rep < 4:
        continue
    for base in names:
        acc.setdefault(base, []).append([(buf[base + k] - buf[base]) / 100.0 for k in range(10)])
for base, name in names.items():
    rows = torch.tensor(acc[base], dtype=torch.float64)
    med = rows.median(dim=0).values.tolist()
    print(f'{name} ({B}x{H}x{W}): us after the last workgroup left its item loop (median of {len(rows)} steps)')
    prev = 0.0
    for label, k in STATIONS:
        if k in (3, 8, 9) and base != 20:
            continue
        print(f'   {label:32s} {med[k]:7.2f}   (+{med[k] - prev:5.2f})')
        prev = med[k]
