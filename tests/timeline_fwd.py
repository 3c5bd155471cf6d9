"""Diagnostic: start / end times (s_memrealtime, 100 MHz) of every workgroup of the three train-mode forward kernels
(luma, statistics, apply), from a -DR2L_EXP_STAMPS -DR2L_TEST_HOOKS build:  R2L_STAMPS_LIB=... python tests/timeline_fwd.py"""
import os, sys, ctypes, torch
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ['R2L_LIB_PATH'] = os.path.join(HERE, '_build', os.environ.get('R2L_STAMPS_LIB', 'lib_stamps.so'))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from oracle import isp_oracle as orc
from raw2logit_amd import _lib
import parity_checks as pc
B, H, W = [int(x) for x in os.environ.get('SHAPE', '64x512x512').split('x')]
dev = torch.device('cuda')
raw = torch.from_numpy(orc.synth_raw(B, H, W, seed=0, kind='uniform')).to(dev)
m = pc.make_module(dict(camera='drone', track=False, additive=False, training=True, bn=True), orc.IspParams(orc.DRONE_CAMERA_PARAMS), dev)
for _ in range(4):
    y = m(raw)
torch.cuda.synchronize()
lib = _lib.device_library()
lib.cdll.r2l_test_debug_offset.restype = ctypes.c_size_t
off = lib.cdll.r2l_test_debug_offset(B, H, W)
tl = y.grad_fn.ws[off:off + 8 * 24576].view(torch.int64).cpu()
hw = y.grad_fn.ws[off + 8 * 24576:off + 8 * (24576 + 4096)].view(torch.int64).cpu()
if os.environ.get('R2L_TL_STREAM'):
    # the row-streaming statistics pass: entry / item loop begins / item loop ends / tail done, per workgroup (4 records at 4 * bid)
    d = tl[:8192].view(-1, 4)
    d = d[(d[:, 3] > 0) & (d[:, 3] - d[:, 0] < 100000000)]
    d = d[d[:, 0] > d[:, 0].max() - 100000]
    t00 = d[:, 0].min().item()
    r = (d - t00).double() / 100
    qq = torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64)
    print(f'streaming statistics pass: {len(d)} workgroups (us after the first workgroup\'s entry; min / 10 % / median / 90 % / max)')
    for j, nm in enumerate(('entry', 'item loop begins', 'item loop ends', 'tail done')):
        print(f'   {nm:18s}', [round(x, 2) for x in r[:, j].quantile(qq).tolist()])
    print('   item loop duration', [round(x, 2) for x in (r[:, 2] - r[:, 1]).quantile(qq).tolist()],
          ' tail duration', [round(x, 2) for x in (r[:, 3] - r[:, 2]).quantile(qq).tolist()])
    ps = tl[8192:8192 + 64 * 16].view(16, 64)
    for w in range(16):
        row = ps[w][ps[w] > 0]
        if len(row) > 3:
            dt = ((row[1:] - row[:-1]).double() / 100).tolist()
            print(f'   sampled workgroup {5 + 128 * w}: us per step', ' '.join(f'{x:.2f}' for x in dt))
    tl[:8192] = 0
t0 = None
for k, name in enumerate(('luma', 'stats', 'apply')):
    d = tl[8192 * k:8192 * (k + 1)].view(-1, 2)
    d = d[d[:, 1] > 0]
    if len(d) == 0:
        continue
    if t0 is None:
        t0 = d[:, 0].min().item()
    st, en = (d[:, 0] - t0).double() / 100, (d[:, 1] - t0).double() / 100
    dur = en - st
    print(f'{name}: {len(d)} workgroups; first start {st.min():.1f} us, last start {st.max():.1f}; first end {en.min():.1f}, last end '
          f'{en.max():.1f}; duration min {dur.min():.1f} median {dur.median():.1f} max {dur.max():.1f} us')
    q = torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9], dtype=torch.float64)
    print('   start quantiles', [round(x, 1) for x in st.quantile(q).tolist()], ' end quantiles', [round(x, 1) for x in en.quantile(q).tolist()])

# ---- where the wavefronts of the statistics kernel ran: wavefronts per SIMD, and how long the workgroups of a CU took
hw = hw[hw != 0]
if len(hw):
    import collections
    hid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
    simd, cu, sh, se = (hid >> 4) & 3, (hid >> 8) & 0xf, (hid >> 12) & 1, (hid >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    per_simd = collections.Counter((cuid * 4 + simd).tolist())
    per_cu = collections.Counter(cuid.tolist())
    print(f'statistics kernel: {len(hw)} wavefronts on {len(per_cu)} CUs / {len(per_simd)} SIMDs;',
          'wavefronts per SIMD:', sorted(collections.Counter(per_simd.values()).items()),
          ' per CU:', sorted(collections.Counter(per_cu.values()).items()))
    d = tl[8192:16384].view(-1, 2)
    nb = int((d[:, 1] > 0).sum())
    dur = ((d[:nb, 1] - d[:nb, 0]).double() / 100)
    wg_cu = cuid.view(-1)[::1]
    # duration of a workgroup against the wavefronts its CU hosts (wavefront 0 of workgroup b is record 4 b)
    hw_all = y.grad_fn.ws[off + 8 * 24576:off + 8 * (24576 + 4096)].view(torch.int64).cpu().view(-1, 4)[:nb]
    c0 = hw_all[:, 0]
    hid0, xcc0 = c0 & 0xffffffff, (c0 >> 32) & 0xf
    cu0 = ((xcc0 * 8 + ((hid0 >> 13) & 7)) * 2 + ((hid0 >> 12) & 1)) * 16 + ((hid0 >> 8) & 0xf)
    for n in sorted(set(per_cu.values())):
        sel = torch.tensor([per_cu[int(c)] == n for c in cu0.tolist()])
        if sel.any():
            print(f'   workgroups on CUs with {n} wavefronts: {int(sel.sum())}, duration median {dur[sel].median():.1f} max {dur[sel].max():.1f} us')

# ---- does the slow start of the first kernel come from the tiny prologue kernel in front of it?  The statistics pass alone
# (C ABI, folded weights valid: no prologue launch), directly behind a chip-filling kernel
if os.environ.get('R2L_FORCE_SPLIT'):
    from raw2logit_amd._lib import ptr
    P = torch.from_numpy(orc.IspParams(orc.DRONE_CAMERA_PARAMS).pack()).to(dev)
    ws = y.grad_fn.ws
    n = ws.numel()
    st7 = torch.empty(7, dtype=torch.float64, device=dev)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    big = torch.empty(64 << 20, device=dev)
    for mode in ('behind a fill kernel', 'behind an idle stream'):
        for _ in range(3):
            if mode == 'behind a fill kernel':
                big.fill_(1.0)
            else:
                torch.cuda.synchronize()
            lib.check(lib.r2l_isp_fwd(ptr(raw), ptr(P), None, None, None, ptr(st7), ptr(ws), n, B, H, W, 1 | 2, s), 'stats')
        torch.cuda.synchronize()
        tl = ws[off:off + 8 * 24576].view(torch.int64).cpu()
        d = tl[:8192].view(-1, 2)
        d = d[d[:, 1] > 0]
        st = (d[:, 0] - d[:, 0].min()).double() / 100
        print(f'luma pass {mode}, no prologue kernel: {len(d)} workgroups, last start {st.max():.1f} us; start quantiles',
              [round(x, 1) for x in st.quantile(q).tolist()])

# ---- the sums pass of kernel B2 (r2l_bwd2_sums_block): start / end of every workgroup's item loop
if os.environ.get('R2L_TL_BWD'):
    cot = torch.randn_like(y)
    for _ in range(3):
        y = m(raw)
        (y * cot).sum().backward()
    torch.cuda.synchronize()
    d = y.grad_fn.ws if y.grad_fn is not None else None
    ws = m(raw).grad_fn.ws if d is None else d
    tlb = ws[off + 4 * 16 * 2048:off + 4 * 16 * 2048 + 8 * 4096].view(torch.int64).cpu().view(-1, 2)
    tlb = tlb[(tlb[:, 1] > tlb[:, 0]) & (tlb[:, 1] - tlb[:, 0] < 100000000)]   # (the area also holds older float records)
    tlb = tlb[(tlb[:, 0] - tlb[:, 0].median()).abs() < 100000000]
    tlb = tlb[tlb[:, 0] > tlb[:, 0].max() - 10000]          # the last launch (records of earlier, larger grids may linger)
    st, en = (tlb[:, 0] - tlb[:, 0].min()).double() / 100, (tlb[:, 1] - tlb[:, 0].min()).double() / 100
    dur = en - st
    print(f'bwd2 sums: {len(tlb)} workgroups; last start {st.max():.1f} us; first end {en.min():.1f}, last end {en.max():.1f}; '
          f'duration min {dur.min():.1f} median {dur.median():.1f} max {dur.max():.1f} us')
    print('   end quantiles', [round(x, 1) for x in en.quantile(q).tolist()])
