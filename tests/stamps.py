"""Diagnostic: per-phase cycle totals (s_memtime) of the tile kernels, from a -DR2L_EXP_STAMPS build."""
import os, sys, ctypes, torch
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ['R2L_LIB_PATH'] = os.path.join(HERE, '_build', os.environ.get('R2L_STAMPS_LIB', 'lib_stamps.so'))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import isp_oracle as orc
from raw2logit_amd import _lib
from raw2logit_amd._lib import ptr
lib = _lib.device_library()
B, H, W = 64, 512, 512
dev = 'cuda'
raw = torch.from_numpy(orc.synth_raw(B, H, W, seed=0)).to(dev)
P = torch.from_numpy(orc.IspParams(orc.DRONE_CAMERA_PARAMS).pack()).to(dev)
n = lib.r2l_isp_workspace_bytes(B, H, W)
ws = torch.zeros(n, dtype=torch.uint8, device=dev)
out = torch.empty((B, 3, H, W), device=dev)
gout = torch.randn((B, 3, H, W), device=dev)
gp = torch.empty(132, device=dev)
bn = torch.tensor([0.4, 0.4, 0.4, 5., 5., 5.], device=dev)
bnb = torch.tensor([0.01, 0.01, 0.01, 0.02, 0.02, 0.02], device=dev)
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
KEEP = 4 if os.environ.get('R2L_STAMPS_KEEP_LUMA') else 0     # R2L_F_KEEP_LUMA: kernel B1 reads Y' from the workspace
for _ in range(3):
    lib.check(lib.r2l_isp_fwd(ptr(raw), ptr(P), None, ptr(bn), ptr(out), None, ptr(ws), n, B, H, W, KEEP, s), 'fwd')
    lib.check(lib.r2l_isp_bwd(ptr(raw), ptr(P), None, ptr(bn), ptr(bnb), ptr(gout), ptr(gp), None, ptr(ws), n, B, H, W, 2 | KEEP, s), 'bwd')
torch.cuda.synchronize()
MAXB = 2048                                  # R2L_MAX_BLOCKS
lib.cdll.r2l_test_debug_offset.restype = ctypes.c_size_t
off = lib.cdll.r2l_test_debug_offset(B, H, W)
dbg = ws[off:off + 3 * 8 * MAXB * 4].view(torch.float32).view(3, MAXB, 8).double().cpu()
names = {0: ['store', 'Y', 'YP', 'fill', 'pixels'], 1: ['store', 'Y', 'YP', 'fill', 'pixels'],
         2: ['store', 'adjblur', 'Y+fold', 'pixels', 'blockred', 'tree', 'unfold']}
def report(k, kn, dbg):
    pass


if os.environ.get('R2L_STAMPS_STATS'):      # the statistics-only forward pass instead of the apply pass
    st7 = torch.empty(7, dtype=torch.float64, device=dev)
    for _ in range(3):
        lib.check(lib.r2l_isp_fwd(ptr(raw), ptr(P), None, None, None, ptr(st7), ptr(ws), n, B, H, W, 1, s), 'fwd stats')
    torch.cuda.synchronize()
    dbg = ws[off:off + 3 * 8 * MAXB * 4].view(torch.float32).view(3, MAXB, 8).double().cpu()
for k, kn in enumerate(['fwd', 'bwd1', 'bwd2']):
    d = dbg[k]
    nb = int((d.sum(1) > 0).sum())
    if nb == 0:          # (the row-streaming forward carries no stamps)
        continue
    d = d[:nb]
    rt = d[:, 7].mean().item()          # s_memrealtime ticks (100 MHz) over the same span
    d = d[:, :7].clone()
    sub = None
    if kn == 'bwd1' and KEEP:            # slots 1, 2 are stages INSIDE the pixel phase (slot 4) of the kept-luma kernel
        sub = d[:, 1:3].clone()
        d[:, 1:3] = 0
    tot = d.sum(1).mean().item()
    print(f'{kn}: {nb} workgroups, mean total {tot:,.0f} cycles over {rt/100:.1f} us: in-kernel clock {tot/rt*0.1:.2f} GHz')
    tt = d.sum(1)
    print(f'   per-workgroup total: min {tt.min().item():,.0f}  median {tt.median().item():,.0f}  p90 {tt.quantile(0.9).item():,.0f}  max {tt.max().item():,.0f}')
    for i, nm in enumerate(names[k]):
        print(f'   {nm:8s} mean {d[:, i].mean().item():10,.0f}  ({100*d[:, i].mean().item()/tot:4.1f} %)  max {d[:, i].max().item():10,.0f}')
    if sub is not None:
        a, b, c = sub[:, 0].mean().item(), sub[:, 1].mean().item(), (d[:, 4] - sub.sum(1)).mean().item()
        print(f'   inside pixels: blur+chroma {a:,.0f}   grad_out + pointwise + next-tile loads {b:,.0f}   '
              f'gradient correlations {c:,.0f}')
if os.environ.get('R2L_STAMPS_DETAIL'):
    for k, kn in enumerate(['fwd', 'bwd1', 'bwd2']):
        d = dbg[k][:512, :7].sum(1)
        print(kn, 'by XCD (bid % 8):', [int(d[x::8].mean().item() / 1e3) for x in range(8)])
        print(kn, 'by w = bid // 8 (kcycles):', [int(d[8 * w:8 * w + 8].mean().item() / 1e3) for w in range(64)])
        print(kn, 'fill-phase cycles by w:', [int(dbg[k][8 * w:8 * w + 8, 3].mean().item() / 1e3) for w in range(64)])
