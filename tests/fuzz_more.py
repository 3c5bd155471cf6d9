"""Randomised parity sweep of the paths tests/fuzz_gpu.py does not touch (not part of the pytest suite):
static chains, raw2rgb (+VJP), the stage-by-stage path (stages, d/d raw, parameter gradients), SSIM / L2.
Random frame shapes including the awkward ones (W % 4 == 2, a few rows, edges 2 px past a tile boundary).
Runs on cuda:0, or on the host emulation of the same kernel source when there is no GPU (SECONDS small)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import isp_oracle as orc
from raw2logit_amd.processing import pipeline_torch as ppt
from raw2logit_amd import functional as F_, losses as L_
import parity_checks as pc

dev = 'cuda' if torch.cuda.is_available() else 'cpu'
if dev == 'cpu':
    import conftest
    import emul_hook
    emul_hook.enable(conftest.build_emulation())
rng = np.random.default_rng(int(os.environ.get('SEED', '0')))
budget = float(os.environ.get('SECONDS', '60'))
which = os.environ.get('WHICH', 'static,raw2rgb,staged,aux').split(',')
t0 = time.time()
counts = {k: 0 for k in which}
worst = {k: 0.0 for k in which}


def rand_shape(wmul=2, hmax=150, wmax=300):
    H = 2 * int(rng.integers(2, hmax // 2))
    W = wmul * int(rng.integers(max(2, 4 // wmul), wmax // wmul))
    if rng.integers(0, 4) == 0:
        H = int(rng.choice([4, 6, 62, 64, 66, 68, 126, 128, 130]))
    if rng.integers(0, 4) == 0:
        W = int(rng.choice([4, 8, 60, 64, 68, 124, 128, 132, 256, 260, 264])) if wmul == 4 else \
            int(rng.choice([4, 6, 62, 64, 66, 126, 128, 130, 254, 256, 258]))
    return int(rng.integers(1, 4)), H, W


def fuzz_static():
    B, H, W = rand_shape(wmul=int(rng.choice([2, 4])))
    deb = ['bilinear', 'malvar2004', 'menon2007'][int(rng.integers(0, 3))]
    sh = ['none', 'sharpening_filter', 'unsharp_masking'][int(rng.integers(0, 3))]
    dn = ['none', 'gaussian_denoising', 'median_denoising', 'fft_denoising'][int(rng.integers(0, 4))]
    cam = [orc.DRONE_CAMERA_PARAMS, orc.MICROSCOPY_CAMERA_PARAMS][int(rng.integers(0, 2))]
    fused = (sh == 'none' and dn == 'none') or (deb == 'bilinear' and sh == 'sharpening_filter' and dn == 'gaussian_denoising')
    if (not fused or deb == 'menon2007') and W % 4:
        W += 2            # the plane passes need W % 4 == 0 (documented; the call raises otherwise)
    # processing()'s numeric arguments (pipeline_numpy.py:70-73), inside what the kernels' windows hold, one case in three
    opts = {}
    if rng.integers(0, 3) == 0:
        opts = dict(gaussian_sigma=float(rng.uniform(0.2, 0.62)), sharp_radius=float(rng.uniform(0.2, 1.12)),
                    sharp_amount=float(rng.uniform(0.2, 2.5)), fft_fraction=float(rng.uniform(0.05, 0.5)),
                    median_kernel_size=int(rng.choice([3, 5])))
        if opts['median_kernel_size'] == 5 and dn == 'median_denoising' and W % 4:
            W += 2            # (the 5x5 median runs as plane passes)
    if deb == 'malvar2004' and (H < 6 or W < 6):
        H, W = max(H, 6), max(W, 8)
    if rng.integers(0, 5) == 0:      # 2, 4 or 8 wavefronts side by side in the row-streaming chain kernels
        W, H = int(rng.choice([260, 516, 772, 1024, 1028, 1540, 2048])), 2 * int(rng.integers(2, 40))
    u = rng.integers(0, 4096, (B, H, W)).astype(np.uint16)
    if rng.integers(0, 2):
        u[:, : H // 2] = rng.integers(240, 270, (B, H // 2, W))
    raw_np = u.astype(np.float32) / np.float32(4095)
    ref = orc.static_batch(raw_np, cam, deb, sh, dn, **opts)
    out = F_.static_pipeline(torch.from_numpy(raw_np).to(dev), cam, deb, sh, dn, **opts).cpu().numpy()
    e = np.abs(out - ref).max()
    assert e <= 1e-5, ('static', (B, H, W), deb, sh, dn, opts, e)
    if W % 4 == 0:
        out16 = F_.static_pipeline(torch.from_numpy(u).to(dev), cam, deb, sh, dn, bits=12, **opts).cpu().numpy()
        dd = np.abs(out16.astype(np.float64) - out)
        assert np.array_equal(out16, out), ('static u16', (B, H, W), deb, sh, dn, 'max diff', float(dd.max()), 'pixels', int((dd > 0).sum()),
                                            'camera', 'drone' if cam is orc.DRONE_CAMERA_PARAMS else 'microscopy', np.argwhere(dd > 0)[:4].tolist())
    if W % 4 == 0 and rng.integers(0, 3) == 0:      # float64 frames, and the T.Normalize epilogue on float32 ones
        ref64 = orc.static_batch(raw_np.astype(np.float64), cam, deb, sh, dn, **opts)
        out64 = F_.static_pipeline(torch.from_numpy(raw_np.astype(np.float64)).to(dev), cam, deb, sh, dn, **opts).cpu().numpy()
        e64 = np.abs(out64 - ref64).max()
        assert e64 <= 1e-5, ('static f64', (B, H, W), deb, sh, dn, e64)
        ms = [float(v) for v in rng.uniform(0.2, 0.9, 3)] + [float(v) for v in rng.uniform(0.05, 0.3, 3)]
        outn = F_.static_pipeline(torch.from_numpy(raw_np).to(dev), cam, deb, sh, dn, mean_std=ms, **opts).cpu()
        refn = (torch.from_numpy(out) - torch.tensor(ms[:3]).view(1, 3, 1, 1)) / torch.tensor(ms[3:]).view(1, 3, 1, 1)
        assert torch.equal(outn, refn), ('static normalize', (B, H, W), deb, sh, dn)
        e = max(e, e64)
    return e / 1e-5


def fuzz_raw2rgb():
    B, H, W = rand_shape()
    rs, oc = bool(rng.integers(0, 2)), int(rng.choice([3, 4]))
    bl = [0.0625, 0.0626, 0.0625, 0.0626] if rng.integers(0, 2) else None
    raw_np = orc.synth_raw(B, H, W, seed=int(rng.integers(0, 1 << 30)), kind='uniform')
    ref = orc.raw2rgb(raw_np, bl, rs, oc)
    raw = torch.from_numpy(raw_np).to(dev).requires_grad_(True)
    blt = None if bl is None else torch.tensor(bl, device=dev, requires_grad=True)
    out = F_.raw2rgb(raw, blt, rs, oc)
    assert np.array_equal(out.detach().cpu().numpy(), ref.astype(np.float32)), ('raw2rgb', (B, H, W), rs, oc)
    cot = rng.standard_normal(ref.shape).astype(np.float32)
    (out * torch.from_numpy(cot).to(dev)).sum().backward()
    gref = orc.raw2rgb_vjp(cot, H, W, rs, oc)
    gr = gref[0] if isinstance(gref, tuple) else gref
    e = np.abs(raw.grad.cpu().numpy() - gr).max()
    assert e <= 1e-6, ('raw2rgb vjp', (B, H, W), rs, oc, e)
    return e / 1e-6


def fuzz_staged():
    B, H, W = rand_shape(hmax=100, wmax=160)
    bn = ['none', 'train', 'eval'][int(rng.integers(0, 3))]
    if bn == 'train' and B * H * W < 512:
        bn = 'none'
    raw_np = pc.midtone_frames(B, H, W, seed=int(rng.integers(0, 1 << 30)))
    P = orc.IspParams(orc.DRONE_CAMERA_PARAMS, dtype=np.float32)
    P.perturb(int(rng.integers(0, 1 << 30)), 0.01)
    m = ppt.ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, track_stages=True, batch_norm_output=(bn != 'none'))
    with torch.no_grad():
        for k, v in P.by_name().items():
            if k != 'additive_layer':
                pc.NAME2ATTR[k](m).copy_(torch.from_numpy(np.asarray(v)))
    m = m.to(dev)
    obn = None
    if bn == 'eval':
        m.eval()
        with torch.no_grad():
            m.batch_norm.running_mean.copy_(torch.tensor([0.3, 0.35, 0.4]))
            m.batch_norm.running_var.copy_(torch.tensor([0.02, 0.03, 0.025]))
        obn = dict(training=False, running_mean=np.array([0.3, 0.35, 0.4]), running_var=np.array([0.02, 0.03, 0.025]))
    elif bn == 'train':
        m.train()
        obn = dict(training=True, running_mean=np.zeros(3), running_var=np.ones(3))
    cot = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    raw = torch.from_numpy(raw_np).to(dev).requires_grad_(True)
    y = m(raw)
    (y * torch.from_numpy(cot).to(dev)).sum().backward()
    Pm = P.astype(np.float64)
    o, stages, cache = orc.parametrized_forward(raw_np, Pm, track_stages=True, bn=obn)
    g, graw, sg = orc.parametrized_backward(Pm, cache, cot, stage_grads=True)
    r = 0.0
    e = np.abs(y.detach().cpu().numpy() - o).max()
    lim = 2e-5 * (float(np.max(cache['istd'])) if bn != 'none' else 1.0)
    assert e <= lim, ('staged out', (B, H, W), bn, e, lim)
    r = max(r, e / lim)
    for k, st in m.stages.items():
        e = np.abs(st.detach().cpu().numpy() - stages[k]).max()
        assert e <= 2e-5, ('staged stage', k, (B, H, W), bn, e)
        ge = np.abs(st.grad.cpu().numpy() - sg[k]).max()
        glim = 1e-4 * (np.abs(sg[k]).max() + 1e-6)
        assert ge <= glim, ('staged stage grad', k, (B, H, W), bn, ge, glim)
        r = max(r, ge / glim)
    ge = np.abs(raw.grad.cpu().numpy() - graw).max()
    glim = 1e-4 * (np.abs(graw).max() + 1e-6)
    assert ge <= glim, ('staged d/d raw', (B, H, W), bn, ge, glim)
    r = max(r, ge / glim)
    for k in g:
        got = pc.NAME2ATTR[k](m).grad.detach().cpu().numpy().reshape(np.asarray(g[k]).shape)
        lim = (3e-5 if bn == 'none' else 2e-4) * (np.abs(g[k]).max() + 1e-6) + 2e-7 * np.sqrt(cot.size) + \
            (1e-7 * cot.size if bn != 'none' else 0.0)
        e = np.abs(got - g[k]).max()
        assert e <= 3 * lim, ('staged param grad', k, (B, H, W), bn, e, lim)
        r = max(r, e / lim)
    return r


def fuzz_aux():
    B = int(rng.integers(1, 3))
    H = int(rng.integers(11, 140))
    W = int(rng.integers(11, 200))
    if rng.integers(0, 3) == 0:
        H, W = int(rng.choice([11, 12, 64, 65, 66, 128, 130])), int(rng.choice([11, 16, 63, 64, 66, 127, 128, 132]))
    x = rng.uniform(0, 1, (B, 3, H, W)).astype(np.float32)
    y = np.clip(x + rng.normal(0, 0.1, x.shape), 0, 1).astype(np.float32)
    xt = torch.from_numpy(x).to(dev)
    yt = torch.from_numpy(y).to(dev).requires_grad_(True)
    s = L_.ssim(xt, yt)
    s.backward()
    ref, gref = orc.ssim(x, y)
    e = abs(float(s.detach()) - float(ref))
    assert e <= 2e-5, ('ssim', (B, H, W), e)
    ge = np.abs(yt.grad.cpu().numpy() - gref).max()
    glim = 2e-4 * (np.abs(gref).max() + 1e-12)
    assert ge <= glim, ('ssim grad', (B, H, W), ge, glim)
    yt.grad = None
    if x.size % 4:       # ISP outputs have even H and W; the L2 kernel asks for a multiple of 4 elements
        return max(e / 2e-5, ge / glim)
    l2 = L_.l2_regularization(xt, yt)
    l2.backward()
    lref, lgref = orc.l2_regularization(x, y)
    e2 = abs(float(l2.detach()) - float(lref)) / (abs(float(lref)) + 1e-9)
    assert e2 <= 1e-5, ('l2', (B, H, W), e2)
    assert np.abs(yt.grad.cpu().numpy() - lgref).max() <= 1e-6, ('l2 grad', (B, H, W))
    return max(e / 2e-5, ge / glim)


fns = {'static': fuzz_static, 'raw2rgb': fuzz_raw2rgb, 'staged': fuzz_staged, 'aux': fuzz_aux}
while time.time() - t0 < budget:
    k = which[int(rng.integers(0, len(which)))]
    worst[k] = max(worst[k], float(fns[k]()))
    counts[k] += 1
print('ok:', ', '.join(f'{k} {counts[k]} cases (worst error / limit {worst[k]:.2f})' for k in which), f'in {time.time() - t0:.0f} s')
