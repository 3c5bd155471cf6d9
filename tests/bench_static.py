"""BASELINE config 3: fused static demosaic->WB->CCM->clip->gamma, batch x 1024 x 1024, GB/s."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import isp_oracle as orc
from raw2logit_amd import _lib, functional as F_
lib = _lib.device_library()
B = int(os.environ.get('B', '256')); S = 1024
raw = torch.rand((B, S, S), device='cuda')
for deb, sh, dn in (('bilinear', 'none', 'none'), ('malvar2004', 'none', 'none'),
                    ('bilinear', 'sharpening_filter', 'gaussian_denoising')):
    for _ in range(2): F_.static_pipeline(raw, orc.DRONE_CAMERA_PARAMS, deb, sh, dn)
    torch.cuda.synchronize(); lib.r2l_timing_enable(1)
    for _ in range(5): F_.static_pipeline(raw, orc.DRONE_CAMERA_PARAMS, deb, sh, dn)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14); lib.r2l_timing_report(buf, len(buf)); lib.r2l_timing_enable(0)
    for l in buf.value.decode().splitlines():
        n, c, ms = l.split(); us = 1e3 * float(ms) / int(c)
        print(f'{deb:10s} {sh[:6]:6s} {dn[:6]:6s} {n:36s} {us:9.1f} us  {B*S*S*16/us/1e3:8.1f} GB/s  {B*S*S/us/1e3:7.1f} Gpix/s  ({100*B*S*S*16/us/1e3/8000:.1f} % of 8 TB/s)')
# the multi-pass chains (luma-plane passes): whole-call time
import time
for deb, sh, dn in (('malvar2004', 'sharpening_filter', 'gaussian_denoising'), ('bilinear', 'sharpening_filter', 'median_denoising')):
    for _ in range(2): F_.static_pipeline(raw, orc.DRONE_CAMERA_PARAMS, deb, sh, dn)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): F_.static_pipeline(raw, orc.DRONE_CAMERA_PARAMS, deb, sh, dn)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 5 * 1e6
    print(f'{deb:10s} {sh[:6]:6s} {dn[:6]:6s} {"4 launches (luma-plane passes)":36s} {us:9.1f} us  {B*S*S/us/1e3:7.1f} Gpix/s')
