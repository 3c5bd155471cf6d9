"""BASELINE config 3 (fused static demosaic->WB->CCM->clip->gamma, 256 x 1024 x 1024): per-LAUNCH durations from a cold
start, and what the slow first launches are.

Round 3 left two numbers for the same kernel on the same shape: 876 us after 2 warm-up launches (this script's
predecessor) and 742 us after 12 (bench.py).  This probe times every launch by itself (HIP events on the launch stream) in
four situations that separate the three candidate causes:

  A  fresh process, GPU idle for seconds:        launches 1..N on new buffers            (clock ramp + first touch + TLB)
  B  same buffers after 3 s of host sleep:       launches 1..N                           (clock ramp only: nothing is new)
  C  right after B, NEW buffers:                 launches 1..N                           (first touch / TLB only: clocks are up)
  D  after 3 s of sleep, 0.3 s of ANOTHER kernel (a device-to-device copy) first:        (clocks brought up by other work)

plus the shader / memory clocks the driver reports before and after each series (rocm-smi, if the box lets us).  The
warm figures (launch 13 on) are the ones bench.py's static_c3 record reports; all chains of that record are printed the
same way at the end (one launch per call for every fused chain).
"""
import ctypes
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2logit_amd import _lib, cameras, functional as F_   # noqa: E402

lib = _lib.device_library()
B, S = int(os.environ.get('B', '256')), int(os.environ.get('S', '1024'))
N = int(os.environ.get('N', '24'))
dev = 'cuda'


def clocks():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.strip() for ln in out.splitlines() if 'sclk' in ln or 'mclk' in ln or 'fclk' in ln]
        return ' | '.join(k.split(':', 1)[-1].strip() if ':' in k else k for k in keep[:3]) or 'n/a'
    except Exception as e:                                  # noqa: BLE001
        return f'n/a ({type(e).__name__})'


def frames():
    u = torch.randint(0, 4096, (B, S, S), device=dev, dtype=torch.int32, generator=torch.Generator(dev).manual_seed(0))
    raw = u.to(torch.float32) / 4095.0
    del u
    return raw


def series(raw, chain, n, fresh_out):
    """n launches, each timed by itself; fresh_out: a NEW output tensor per launch from a cold allocator"""
    us = []
    keep = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = F_.static_pipeline(raw, cameras.DRONE, *chain)
        e1.record()
        e1.synchronize()
        us.append(1e3 * e0.elapsed_time(e1))
        if fresh_out:
            keep.append(out)
    return us


def show(tag, us):
    print(f'{tag:58s} launches 1-4: ' + ' '.join(f'{u:6.0f}' for u in us[:4]) + f' | 5-12 mean {sum(us[4:12]) / 8:6.0f}'
          f' | 13-{len(us)} mean {sum(us[12:]) / max(len(us) - 12, 1):6.0f} us', flush=True)


short = ('bilinear', 'none', 'none')
print(f'static short chain (bilinear), {B}x{S}x{S}; per-launch HIP-event times; clocks before: {clocks()}')
time.sleep(3.0)
raw = frames()
torch.cuda.synchronize()
time.sleep(3.0)
show('A  fresh process, idle GPU, new buffers', series(raw, short, N, False))
print('   clocks after A:', clocks())
time.sleep(3.0)
print('   clocks after 3 s of sleep:', clocks())
show('B  same buffers after 3 s of sleep (clock ramp only)', series(raw, short, N, False))
del raw
torch.cuda.empty_cache()
raw2 = frames()                                      # new addresses; the chip is still at speed
show('C  right after B, NEW buffers (first touch / TLB only)', series(raw2, short, N, False))
time.sleep(3.0)
a = torch.empty(1 << 28, device=dev, dtype=torch.float32)
b = torch.empty_like(a)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    b.copy_(a)
    torch.cuda.synchronize()
show('D  after 3 s of sleep + 0.3 s of device copies first', series(raw2, short, N, False))
del a, b
print('   clocks after D:', clocks())

# ---- the warm figures of every chain bench.py's static_c3 reports (12 warm-up launches, then 20 timed ones) ----
print()
for what, chain in (('short chain, bilinear', short), ('short chain, Malvar2004', ('malvar2004', 'none', 'none')),
                    ('train.py default chain (bilinear + sharpening_filter + gaussian)',
                     ('bilinear', 'sharpening_filter', 'gaussian_denoising')),
                    ('Malvar2004 + sharpening_filter + gaussian', ('malvar2004', 'sharpening_filter', 'gaussian_denoising')),
                    ('bilinear + sharpening_filter + median', ('bilinear', 'sharpening_filter', 'median_denoising'))):
    for _ in range(12):
        F_.static_pipeline(raw2, cameras.DRONE, *chain)
    torch.cuda.synchronize()
    lib.r2l_timing_enable(1)
    for _ in range(20):
        F_.static_pipeline(raw2, cameras.DRONE, *chain)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14)
    lib.r2l_timing_report(buf, len(buf))
    lib.r2l_timing_enable(0)
    tot, names = 0.0, []
    for ln in buf.value.decode().splitlines():
        n, c, ms = ln.split()
        tot += 1e3 * float(ms)
        names.append(f'{n.replace("r2l_launch_", "").replace("_kernel", "")} x{int(c) // 20}')
    us = tot / 20
    print(f'{what:70s} {us:8.1f} us per call  {B * S * S * 16 / us / 1e3:7.1f} GB/s = {100 * B * S * S * 16 / us / 1e3 / 8000:5.1f} % '
          f'of 8 TB/s   [{", ".join(names)}]', flush=True)
