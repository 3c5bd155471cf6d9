"""Does a tensor that was just written stay in the 256 MB Infinity Cache (MALL) for the next kernel?  Times an in-place
elementwise pass over N MB right after a producer wrote it, against the same pass on cold data."""
import torch, sys
dev = 'cuda'
def t(fn, n=20, pre=None):
    ts = []
    for _ in range(n):
        if pre: pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
for mb in (50, 100, 150, 201, 268, 402):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device=dev); src = torch.randn(n, device=dev)
    big = torch.empty(600 * 1000 * 1000 // 4, device=dev)
    flush = lambda: big.fill_(1.0)
    prod = lambda: x.copy_(src)                       # producer: writes x (reads src)
    prod_only = lambda: x.fill_(0.5)                  # producer: writes only
    us_cold = t(lambda: x.mul_(1.0001), pre=flush)
    us_after_copy = t(lambda: x.mul_(1.0001), pre=prod)
    us_after_fill = t(lambda: x.mul_(1.0001), pre=prod_only)
    us_read_cold = t(lambda: x.sum(), pre=flush)
    us_read_warm = t(lambda: x.sum(), pre=prod_only)
    print(f'{mb:4d} MB: in-place pass cold {us_cold:7.1f} us ({2*mb/us_cold:5.2f} TB/s)  after copy {us_after_copy:7.1f} ({2*mb/us_after_copy:5.2f})  '
          f'after fill {us_after_fill:7.1f} ({2*mb/us_after_fill:5.2f})   read cold {us_read_cold:7.1f} ({mb/us_read_cold:5.2f})  read after fill {us_read_warm:7.1f} ({mb/us_read_warm:5.2f})')
