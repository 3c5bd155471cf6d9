"""Does the memory-side cache (256 MiB) serve a re-read?  Read bandwidth of repeated passes over buffers of 32 MB .. 1 GB
(torch reductions and copies: nothing of this library) -- above the HBM rate for footprints that fit = it does."""
import time, torch
dev = 'cuda'
for mb in (32, 64, 96, 128, 192, 256, 384, 512, 1024):
    n = mb * (1 << 20) // 4
    x = torch.rand(n, device=dev)
    y = torch.empty_like(x)
    for name, fn, bytes_ in (('sum (read)', lambda: x.sum(), 4 * n), ('copy (read+write)', lambda: y.copy_(x), 8 * n)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        e1.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 20
        print(f'{mb:5d} MB  {name:18s} {us:8.1f} us  {bytes_ / us / 1e3:8.1f} GB/s', flush=True)
    del x, y
