"""SSIM / L2 auxiliary losses on the headline batch (two processor outputs of 64x3x512x512): time and bytes."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raw2logit_amd import _lib, losses
lib = _lib.device_library()
B, C, S = 64, 3, 512
x = torch.rand((B, C, S, S), device='cuda')
y = (x + 0.05 * torch.randn_like(x)).clamp(0, 1).requires_grad_(True)
n = B * C * S * S
for name, fn, bytes_fwd, bytes_bwd in (('ssim', losses.SSIM(11), 8.0, 8.0 + 12.0 + 12.0 + 8.0 + 4.0),
                                       ('l2', losses.l2_regularization, 8.0, 12.0)):
    for _ in range(2):
        y.grad = None
        fn(x, y).backward()
    torch.cuda.synchronize(); lib.r2l_timing_enable(1)
    for _ in range(5):
        y.grad = None
        fn(x, y).backward()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14); lib.r2l_timing_report(buf, len(buf)); lib.r2l_timing_enable(0)
    tot = 0.0
    for l in buf.value.decode().splitlines():
        k, c, ms = l.split(); us = 1e3 * float(ms) / 5; tot += us
        print(f'{name:5s} {k:34s} {int(c)//5} launch(es)/step {us:9.1f} us/step')
    print(f'{name:5s} fwd+bwd {tot:9.1f} us  = {n / tot / 1e3:7.1f} Gelem/s, {n * (bytes_fwd + bytes_bwd) / tot / 1e3:7.1f} GB/s algorithmic')
