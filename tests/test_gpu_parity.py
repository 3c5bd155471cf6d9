"""GPU (MI355X) parity: libr2l_isp.so through the C ABI against the oracle and the reference's golden
vectors -- the same checks tests/test_emul_parity.py runs on the host emulation, plus full-size
properties at BASELINE.json's sizes."""
import sys

import numpy as np
import pytest
import torch

from oracle import isp_oracle as orc
from oracle.golden_cases import PARAM_CASES, STATIC_CASES
import parity_checks as pc
from raw2logit_amd.processing import pipeline_torch as ppt

pytestmark = pytest.mark.gpu

DEVICE_STATIC = list(STATIC_CASES)   # every chain of the golden set is built for the device


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from raw2logit_amd import _lib
    lib = _lib.device_library()          # raises if libr2l_isp.so is missing: no fallback
    assert lib.is_device
    return 'cuda:0'


@pytest.mark.parametrize('case', PARAM_CASES, ids=[c['name'] for c in PARAM_CASES])
def test_fused_parametrized(case, golden, dev):
    pc.check_param_case(case, golden, dev)


def test_raw2rgb(golden, dev):
    pc.check_raw2rgb(golden, dev)


def test_nnprocessing_front_end(golden, dev):
    """NNProcessing's raw2rgb front end (pipeline_torch.py:111-114) on the GPU, the third-party
    U-Net++ body replaced by a stub injected into sys.modules"""
    pc.check_nnprocessing(golden, dev)


@pytest.mark.parametrize('case', DEVICE_STATIC, ids=[c['name'] for c in DEVICE_STATIC])
def test_static(case, golden, dev):
    pc.check_static_case(case, golden, dev)


def test_properties(dev):
    pc.check_ragged_and_properties(dev)


def test_frame_shapes_around_tile_boundaries(dev):
    pc.check_frame_shapes(dev)


# ---- the plane-pass backward (r2l_param_plane_bwd.h: bwd1_plane, bwd1_blur_hp | bwd1_blur + bwd2_hp, bwd2_sums) -- the
# kernels behind the headline number -- DIRECTLY against the float64 oracle, the reference's golden float32 gradients and
# its own float64 run (grad64).  The shipped library takes them at B*H*W >= 4 Mi px only, so the small cases run on the
# diagnostic build with R2L_BWD_PLANES=1 (same source); test_plane_backward_at_the_dispatch_threshold runs the shipped one.
PLANE_MODES = {'fused-middle-pass': {'R2L_BWD_PLANES': '1'},
               'split-blur': {'R2L_BWD_PLANES': '1', 'R2L_BWD_SPLIT_BLUR': '1'}}
PLANE_PARAM_CASES = [c for c in PARAM_CASES if not c['additive'] and c['shape'][2] % 4 == 0]
PLANE_KERNELS = {'fused-middle-pass': ('r2l_launch_bwd1_plane_kernel', 'r2l_launch_bwd1_blur_hp_kernel',
                                       'r2l_launch_bwd2_sums_kernel'),
                 'split-blur': ('r2l_launch_bwd1_plane_kernel', 'r2l_launch_bwd1_blur_kernel', 'r2l_launch_bwd2_hp_kernel',
                                'r2l_launch_bwd2_sums_kernel')}


@pytest.mark.parametrize('mode', list(PLANE_MODES))
@pytest.mark.parametrize('case', PLANE_PARAM_CASES, ids=[c['name'] for c in PLANE_PARAM_CASES])
def test_plane_backward_golden_cases(case, mode, golden, dev):
    """check_param_case (float64 oracle, golden float32 gradients, grad64 pin) with the backward as passes over planes;
    the launch record proves which kernels produced the gradients.  Reference: pipeline_torch.py:187-217 under autograd."""
    from raw2logit_amd import _lib
    with pc.env_overrides(dev, PLANE_MODES[mode]):
        lib = _lib.device_library()
        _, names = pc.kernels_launched(lib, lambda: pc.check_param_case(case, golden, dev))
    for k in PLANE_KERNELS[mode]:
        assert names.get(k, 0) >= 1, (mode, k, sorted(names))
    assert not any('bwd1_saved' in n or n.startswith('r2l_launch_bwd2_kernel') for n in names), sorted(names)


@pytest.mark.parametrize('mode', list(PLANE_MODES))
def test_plane_backward_frame_shapes(mode, dev):
    """check_frame_shapes (every gradient within 3e-5 of its scale of the float64 oracle on well-conditioned frames) on the
    plane passes: image edges at every distance from a strip / band boundary, 4- and 6-row frames (every row a border row of
    the blur and of its adjoint), several strips, partially filled last strip."""
    from raw2logit_amd import _lib
    with pc.env_overrides(dev, PLANE_MODES[mode]):
        lib = _lib.device_library()
        worst, names = pc.kernels_launched(lib, lambda: pc.check_frame_shapes(dev, shapes=pc.FRAME_SHAPES_PLANES, conditioning=True))
    for k in PLANE_KERNELS[mode]:
        assert names.get(k, 0) >= len(pc.FRAME_SHAPES_PLANES), (mode, k, names)
    assert not any('bwd1_saved' in n or n.startswith('r2l_launch_bwd2_kernel') for n in names), sorted(names)
    pc.report(f'plane-backward/{mode}/frame shapes: worst gradient error (fraction of its limit)', worst, 1.0)
    # ... and at other band heights / one wavefront taking everything
    for env in ({'R2L_BP_BAND': '6', 'R2L_HB_BAND': '12', 'R2L_B2S_BAND': '6', 'R2L_HP_BAND': '6'},
                {'R2L_BP_BAND': '1000', 'R2L_HB_BAND': '1000', 'R2L_B2S_BAND': '1000', 'R2L_HP_BAND': '1000',
                 'R2L_GRID_BWD1': '1', 'R2L_GRID_BWD2': '1'}):
        with pc.env_overrides(dev, dict(PLANE_MODES[mode], **env)):
            w2 = pc.check_frame_shapes(dev, shapes=pc.FRAME_SHAPES_PLANES[:6], conditioning=True)
        pc.report(f'plane-backward/{mode}/{sorted(env.items())[0]}: worst gradient error (fraction of its limit)', w2, 1.0)


def _shipped_plane_backward_vs_oracle(dev, B, tag):
    """The SHIPPED library's plane-pass backward on B x 512 x 512 well-conditioned frames (pc.midtone_frames: no pixel within
    0.05 of a clip threshold under 1 %-perturbed Drone parameters, so torch.clip's step gradient cannot flip and float32
    round-off is all that separates a correct kernel from the float64 oracle).  BatchNorm in eval mode decouples the frames
    and the cotangent is zero except on the first and the last frame (the first and the last band items of the launch), so
    all 132 gradients of the batch equal the oracle's on those two frames: limit 3e-5 of the gradient's scale
    (pc.tight_grad_limit -- the criterion of check_frame_shapes; round 4 allowed 1.5e-3 + a flip allowance, 450 x what the
    kernels achieve).  Also without BatchNorm, and from 16-bit containers (bit-identical)."""
    from raw2logit_amd import _lib
    H = W = 512
    lib = _lib.device_library()
    assert lib.path == _lib.LIB_PATH
    raw_np = pc.midtone_frames(B, H, W, seed=4)
    u = np.rint(raw_np.astype(np.float64) * 4095).astype(np.uint16)
    assert np.array_equal(u.astype(np.float32) / np.float32(4095), raw_np)
    P = orc.IspParams(orc.DRONE_CAMERA_PARAMS)
    P.perturb(23, 0.01)
    sel = [0, B - 1]
    cot_np = np.zeros((B, 3, H, W), np.float32)
    cot_np[sel] = np.random.default_rng(5).standard_normal((2, 3, H, W)).astype(np.float32)
    cot = torch.from_numpy(cot_np).to(dev)
    P64 = P.astype(np.float64)
    for bn in (True, False):
        case = dict(camera='drone', track=False, additive=False, training=False, bn=bn)
        grads = {}
        for frames in ('f32', 'u16'):
            m = pc.make_module(case, P, dev)
            if frames == 'u16':
                m.raw_bits = 12
            raw = torch.from_numpy(u if frames == 'u16' else raw_np).to(dev)

            def step():
                y = m(raw)
                y.backward(cot)
                return y.detach()
            y, names = pc.kernels_launched(lib, step)
            for k in PLANE_KERNELS['fused-middle-pass']:
                k16 = k.replace('_kernel', '_u16_kernel')
                assert names.get(k, 0) + names.get(k16, 0) == 1, (k, sorted(names))
            grads[frames] = {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters()}
            if frames == 'f32':
                y_sel = y[sel].cpu().numpy()
        for n in grads['f32']:
            assert np.array_equal(grads['f32'][n], grads['u16'][n]), (bn, n, '16-bit containers')
        o, _, c = orc.parametrized_forward(raw_np[sel], P64, bn=pc.oracle_bn(case))
        assert c['rgb'].min() > 0.05 and c['rgb'].max() < 0.95, (c['rgb'].min(), c['rgb'].max())
        tol = pc.out_tolerance(c, bn)
        err = np.abs(y_sel - o)
        w = np.unravel_index((err / tol).argmax(), err.shape)
        pc.report(f'{tag}/bn={bn}/out (frames 0, {B - 1}) vs float64 oracle', err[w], tol[w])
        assert np.all(err <= tol), (bn, err.max())
        og, _, _ = orc.parametrized_backward(P64, c, cot_np[sel])
        lo, _, _ = orc.parametrized_backward(P64, c, cot_np[sel], clip_shift=1e-6)
        hi, _, _ = orc.parametrized_backward(P64, c, cot_np[sel], clip_shift=-1e-6)
        for k, ref in og.items():
            ref = np.asarray(ref)
            got = grads['f32'][k].reshape(ref.shape)
            lim = pc.tight_grad_limit(ref, lo[k], hi[k])          # 3e-5 of the scale (the clip-flip allowance is zero here)
            assert lim <= 3.0001e-5 * (np.abs(ref).max() + 1e-6), (k, 'a pixel sits on a clip threshold', lim)
            e = np.abs(got - ref).max()
            pc.report(f'{tag}/bn={bn}/grad {k} vs float64 oracle (plane passes, shipped library)', e, lim)
            assert e <= lim, (bn, k, e, lim)


def test_plane_backward_at_the_dispatch_threshold(dev):
    """16 x 512 x 512 = 4 Mi px is the smallest batch whose backward the shipped library runs as plane passes
    (r2l_api_impl.h: `planes`).  Reference: pipeline_torch.py:187-217 under autograd."""
    _shipped_plane_backward_vs_oracle(dev, 16, 'threshold-16x512x512')


def test_plane_backward_at_the_headline_shape(dev):
    """the same at BASELINE config 2's shape, 64 x 512 x 512 -- the launch shapes (one round of resident wavefronts, band
    heights, helper workgroups of the sums pass) behind bench.py's line"""
    _shipped_plane_backward_vs_oracle(dev, 64, 'headline-64x512x512')


def test_plane_backward_beyond_one_round_of_workgroups(dev):
    """72 x 512 x 512: the sums pass has 792 work-item quadruples for its 768 resident workgroups, so its workgroups loop over
    items and the 14 helper workgroups (B1's partials) only find a slot when the first of them retires -- the launch's last
    arrival may be a helper.  Size-independent property (eval-mode BatchNorm decouples the frames): all 132 gradients of the
    whole batch equal the sum of the gradients of its two halves (396 workgroups each); two runs are bitwise identical."""
    import copy
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
    B, H, W = 72, 512, 512
    raw = torch.from_numpy(orc.synth_raw(B, H, W, seed=3, kind='uniform')).to(dev)
    m = ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=True).to(dev)
    m.eval()
    with torch.no_grad():
        m.batch_norm.running_mean.copy_(torch.tensor([0.3, 0.35, 0.4]))
        m.batch_norm.running_var.copy_(torch.tensor([0.02, 0.03, 0.025]))
    g = torch.randn((B, 3, H, W), device=dev, generator=torch.Generator(dev).manual_seed(2))
    runs = []
    for _ in range(2):
        for p in m.parameters():
            p.grad = None
        m(raw).backward(g)
        runs.append({k: p.grad.detach().clone() for k, p in m.named_parameters()})
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k
    acc = {k: torch.zeros_like(v) for k, v in runs[0].items()}
    for h in range(2):
        mh = copy.deepcopy(m)
        for p in mh.parameters():
            p.grad = None
        sl = slice(36 * h, 36 * (h + 1))
        mh(raw[sl]).backward(g[sl])
        for k, p in mh.named_parameters():
            acc[k] += p.grad
    for k in acc:
        scale = runs[0][k].abs().max().item() + 1e-6
        e = (acc[k] - runs[0][k]).abs().max().item()
        pc.report(f'72x512x512/grad {k}: whole batch (looping workgroups, late helpers) vs sum of halves', e, 1e-4 * scale)
        assert e <= 1e-4 * scale, (k, e, scale)


def test_cpu_tensor_is_refused(dev):
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
    from raw2logit_amd import _lib
    emul_hook = sys.modules.get('emul_hook')
    if emul_hook is not None and emul_hook.active() is not None:
        pytest.skip('emulation hook active in this process')
    with pytest.raises(_lib.R2LError):
        ParametrizedProcessing()(torch.rand(1, 8, 8))


def test_full_size_config2_properties(dev):
    """BASELINE config 2 (64 x 512 x 512, BatchNorm train) at full size; the float64 oracle needs minutes for the whole
    batch, so size-independent properties: (i) the normalised output has zero mean / unit biased variance per channel,
    (ii) two runs are bitwise identical (fixed-order reductions), (iii) eval mode with the batch statistics reproduces the
    train-mode output (frames decouple), (iv) a 2-frame slice in that eval mode -- output and all 132 gradients -- against
    the float64 oracle (the pattern of test_full_size_config4_microscopy; round 2 compared the slice through a least-squares
    affine fit), (v) eval-mode gradients of the whole batch are the sum of the gradients of its four quarters."""
    import copy
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
    B, H, W = 64, 512, 512
    raw_np = orc.synth_raw(B, H, W, seed=0, kind='uniform')
    raw = torch.from_numpy(raw_np).to(dev)
    m = ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=True).to(dev)
    m.train()
    m.batch_norm.momentum = 1.0                 # running statistics := this batch's statistics
    y = m(raw)
    mean = y.detach().double().mean(dim=(0, 2, 3))
    var = y.detach().double().var(dim=(0, 2, 3), unbiased=False)
    pc.report('config2/train-mode output: |mean|', mean.abs().max().item(), 1e-4)
    pc.report('config2/train-mode output: |var - 1|', (var - 1).abs().max().item(), 1e-3)
    assert mean.abs().max() < 1e-4 and (var - 1).abs().max() < 1e-3
    g = torch.randn(y.shape, device=dev, generator=torch.Generator(dev).manual_seed(1))
    (y * g).sum().backward()
    grads1 = [p.grad.clone() for p in m.parameters()]
    rm1, rv1 = m.batch_norm.running_mean.clone(), m.batch_norm.running_var.clone()
    for p in m.parameters():
        p.grad = None
    y2 = m(raw)
    (y2 * g).sum().backward()
    assert torch.equal(y, y2)
    for a, p in zip(grads1, m.parameters()):
        assert torch.equal(a, p.grad)
    assert torch.equal(rm1, m.batch_norm.running_mean) and torch.equal(rv1, m.batch_norm.running_var)
    # (iii) eval mode with the batch statistics
    n = B * H * W
    with torch.no_grad():
        m.batch_norm.running_var.mul_((n - 1) / n)          # unbiased -> the biased variance train mode used
    m.eval()
    for p in m.parameters():
        p.grad = None
    y_ev = m(raw)
    y_ev.backward(g)
    e = (y_ev.detach() - y.detach()).abs().max().item()
    pc.report('config2/eval mode with the batch statistics vs train mode', e, 2e-5)
    assert e <= 2e-5
    g_full = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    # (iv) 2-frame slice against the float64 oracle
    ms = copy.deepcopy(m)
    for p in ms.parameters():
        p.grad = None
    ys = ms(raw[:2])
    ys.backward(g[:2])
    assert torch.equal(ys.detach(), y_ev.detach()[:2])
    P = orc.IspParams(orc.DRONE_CAMERA_PARAMS, dtype=np.float64)
    bn = dict(training=False, running_mean=m.batch_norm.running_mean.double().cpu().numpy(),
              running_var=m.batch_norm.running_var.double().cpu().numpy())
    o, _, c = orc.parametrized_forward(raw_np[:2], P, bn=bn)
    tol = pc.out_tolerance(c, True)
    err = np.abs(ys.detach().cpu().numpy() - o)
    w = np.unravel_index((err / tol).argmax(), err.shape)
    pc.report('config2/2-frame slice: out vs float64 oracle', err[w], tol[w])
    assert np.all(err <= tol), (err.max(), w)
    cot_np = g[:2].cpu().numpy()
    og_all, _, _ = orc.parametrized_backward(P, c, cot_np)
    lo, _, _ = orc.parametrized_backward(P, c, cot_np, clip_shift=1e-6)
    hi, _, _ = orc.parametrized_backward(P, c, cot_np, clip_shift=-1e-6)
    for k, og in og_all.items():
        got = pc.NAME2ATTR[k](ms).grad.detach().cpu().numpy().reshape(np.asarray(og).shape)
        lim = pc.tight_grad_limit(og, lo[k], hi[k])               # 3e-5 of the scale + the clip-flip allowance
        e = np.abs(got - og).max()
        pc.report(f'config2/2-frame slice: grad {k} vs float64 oracle', e, lim)
        assert e <= lim, (k, e, lim)
    # (v) gradients add up over the batch
    acc = {k: torch.zeros_like(v) for k, v in g_full.items()}
    for q in range(4):
        mq = copy.deepcopy(m)
        for p in mq.parameters():
            p.grad = None
        sl = slice(16 * q, 16 * (q + 1))
        mq(raw[sl]).backward(g[sl])
        for k, p in mq.named_parameters():
            acc[k] += p.grad
    for k in g_full:
        scale = g_full[k].abs().max().item() + 1e-6
        e = (acc[k] - g_full[k]).abs().max().item()
        pc.report(f'config2/grad {k}: whole batch vs sum of quarters', e, 1e-4 * scale)
        assert e <= 1e-4 * scale, (k, e, scale)


def test_baseline_config1_exact_batch(dev):
    """BASELINE config 1 itself on the HIP path: 16 synthetic 256 x 256 12-bit RGGB frames (uniform codes, seed 0 -- the
    batch bench.py's cpu_baseline leg times on the host), Drone camera parameters, the static chain `train.py:96-101`
    defaults to (bilinear + sharpening_filter + gaussian_denoising), EVERY frame against the oracle's restatement of
    processing() (pipeline_numpy.py:70-141); through the batched module, through the per-image callable the reference's
    datasets hold (RawProcessingPipeline.__call__, :55-67: (H,W) ndarray -> (3,H,W) float32 tensor), and from the 16-bit
    containers (bit-identical).  Also the short chain of config 3 and the Malvar2004 variant on the same batch."""
    from raw2logit_amd import functional as F_
    from raw2logit_amd.processing import pipeline_numpy as ppn
    B, H, W = 16, 256, 256
    raw_np = orc.synth_raw(B, H, W, seed=0, kind='uniform')
    u = np.rint(raw_np.astype(np.float64) * 4095).astype(np.uint16)
    assert np.array_equal(u.astype(np.float32) / np.float32(4095), raw_np)      # 12-bit codes / (2**12 - 1), dataset.py:87
    raw = torch.from_numpy(raw_np).to(dev)
    for chain in (('bilinear', 'sharpening_filter', 'gaussian_denoising'), ('bilinear', 'none', 'none'),
                  ('malvar2004', 'sharpening_filter', 'gaussian_denoising')):
        ref = orc.static_batch(raw_np, orc.DRONE_CAMERA_PARAMS, *chain)
        out = F_.static_pipeline(raw, orc.DRONE_CAMERA_PARAMS, *chain)
        assert out.shape == (B, 3, H, W) and out.dtype == torch.float32
        e = np.abs(out.cpu().numpy() - ref).max()
        pc.report(f'config1 16x256x256/{"+".join(chain)}/all frames vs oracle', e, 1e-5)
        assert e <= 1e-5, (chain, e)
        out16 = F_.static_pipeline(torch.from_numpy(u.view(np.int16)).to(dev), orc.DRONE_CAMERA_PARAMS, *chain, bits=12)
        assert torch.equal(out16, out), (chain, '16-bit containers')
    pipe = ppn.RawProcessingPipeline(orc.DRONE_CAMERA_PARAMS, debayer='bilinear', sharpening='sharpening_filter',
                                     denoising='gaussian_denoising')
    ref = orc.static_batch(raw_np, orc.DRONE_CAMERA_PARAMS, 'bilinear', 'sharpening_filter', 'gaussian_denoising')
    for k in (0, 7, 15):
        t = pipe(raw_np[k].copy())
        assert tuple(t.shape) == (3, H, W) and t.dtype == torch.float32
        e = np.abs(t.cpu().numpy() - ref[k]).max()
        pc.report(f'config1 16x256x256/RawProcessingPipeline.__call__ frame {k} vs oracle', e, 1e-5)
        assert e <= 1e-5, (k, e)


def test_full_size_static_config3_slice(dev):
    """BASELINE config 3 shape (1024 x 1024 frames, short chain): a batch of 8 on the device, every
    frame checked against the oracle on a 64-row band (top, middle, bottom) plus idempotence of the
    batch dimension."""
    from raw2logit_amd import functional as F_
    B, H, W = 8, 1024, 1024
    raw_np = orc.synth_raw(B, H, W, seed=0, kind='uniform')
    raw = torch.from_numpy(raw_np).to(dev)
    out = F_.static_pipeline(raw, orc.DRONE_CAMERA_PARAMS, 'bilinear', 'none', 'none')
    ref = orc.static_batch(raw_np[:1], orc.DRONE_CAMERA_PARAMS, 'bilinear', 'none', 'none')
    err = np.abs(out[:1].cpu().numpy() - ref)
    assert err.max() <= 1e-5, err.max()
    out1 = F_.static_pipeline(raw[3:4], orc.DRONE_CAMERA_PARAMS, 'bilinear', 'none', 'none')
    assert torch.equal(out1, out[3:4])
    outm = F_.static_pipeline(raw[:2], orc.DRONE_CAMERA_PARAMS, 'malvar2004', 'none', 'none')
    refm = orc.static_batch(raw_np[:1], orc.DRONE_CAMERA_PARAMS, 'malvar2004', 'none', 'none')
    assert np.abs(outm[:1].cpu().numpy() - refm).max() <= 1e-5


def test_full_size_static_config3_batch(dev):
    """BASELINE config 3 at its full size (256 x 1024 x 1024, 12-bit frames drawn on the device): the short chain, the
    train.py default chain and the Malvar2004 + median chain.  Size-independent properties: the first, a middle and
    the last frame against the oracle (whole frames), batch independence (a frame processed alone gives the same
    bits), and every output inside [0, 1] and finite."""
    from raw2logit_amd import functional as F_
    B, H, W = 256, 1024, 1024
    g = torch.Generator(device=dev).manual_seed(3)
    u = torch.randint(0, 4096, (B, H, W), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    # (tensor / tensor: ATen divides a float tensor by a Python scalar as a multiplication by its reciprocal, which is
    # not the correctly rounded quotient the datasets' numpy division and the 16-bit entry points produce)
    raw = u.to(torch.float32) / torch.full((), 4095.0, device=dev)
    picks = (0, 131, 255)
    host = {k: raw[k].cpu().numpy()[None] for k in picks}
    for chain in (('bilinear', 'none', 'none'), ('bilinear', 'sharpening_filter', 'gaussian_denoising'),
                  ('malvar2004', 'sharpening_filter', 'median_denoising')):
        out = F_.static_pipeline(raw, orc.DRONE_CAMERA_PARAMS, *chain)
        assert bool(torch.isfinite(out).all()) and float(out.min()) >= 0.0 and float(out.max()) <= 1.0, chain
        for k in picks:
            ref = orc.static_batch(host[k], orc.DRONE_CAMERA_PARAMS, *chain)
            e = np.abs(out[k:k + 1].cpu().numpy() - ref).max()
            pc.report(f'config3 full size/{"+".join(chain)}/frame {k} vs oracle', e, 1e-5)
            assert e <= 1e-5, (chain, k, e)
        alone = F_.static_pipeline(raw[131:132], orc.DRONE_CAMERA_PARAMS, *chain)
        assert torch.equal(alone, out[131:132]), chain
        out16 = F_.static_pipeline(u[:4], orc.DRONE_CAMERA_PARAMS, *chain, bits=12)
        assert torch.equal(out16, out[:4]), (chain, '16-bit containers')
        del out


def test_harness_logits_and_adam_step(golden, dev):
    pc.check_harness(golden, dev)


TRACK_CASES = [c for c in PARAM_CASES if c['track']]


@pytest.mark.parametrize('case', TRACK_CASES, ids=[c['name'] for c in TRACK_CASES])
def test_staged_track_stages(case, golden, dev):
    pc.check_staged_case(case, golden, dev)


def test_reductions_do_not_depend_on_the_grid(dev):
    pc.check_grid_independence(dev)


GRAD_RAW_CASES = [c for c in PARAM_CASES if not c['track'] and c['name'] in (
    'drone_bn_train', 'drone_additive_bn', 'micro_bn_train', 'drone_ragged_tile', 'tiny_4x4', 'drone_bn_eval')]


@pytest.mark.parametrize('case', GRAD_RAW_CASES, ids=[c['name'] for c in GRAD_RAW_CASES])
def test_frames_requiring_grad_take_the_staged_kernels(case, golden, dev):
    pc.check_staged_case(case, golden, dev)


def test_16bit_containers_are_bit_identical_to_host_normalised_frames(dev):
    pc.check_u16_ingest(dev)


def test_static_chain_combinations(dev):
    pc.check_static_combinations(dev)


def test_static_numeric_arguments(golden, dev):
    pc.check_static_options(golden, dev)


def test_static_normalize_epilogue(dev):
    pc.check_static_normalize(dev)


def test_adversarial_aux_losses(golden, dev):
    pc.check_aux_losses(golden, dev)


def test_survives_hip_graph_capture(dev):
    """Every call only enqueues on the current stream (no host round trip, no allocation outside torch's
    allocator, arrival counters left at zero), so forward + backward can be captured and replayed as HIP graphs."""
    import copy
    raw = torch.from_numpy(orc.synth_raw(4, 128, 128, seed=41, kind='scene')).to(dev)
    cot = torch.randn((4, 3, 128, 128), device=dev, generator=torch.Generator(dev).manual_seed(3))
    m = ppt.ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=True).to(dev).train()
    m2 = copy.deepcopy(m)
    import gc
    gc.collect()                    # (no cyclic garbage with device resources left for a collection in the middle of the captures)
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        graphed = torch.cuda.make_graphed_callables(m2, (raw,))
    finally:
        if gc_was_on:
            gc.enable()
    for _ in range(3):          # replays
        for p in list(m.parameters()) + list(m2.parameters()):
            p.grad = None
        y0 = m(raw)
        y1 = graphed(raw)
        y0.backward(cot)
        y1.backward(cot)
        assert torch.equal(y0, y1)
        for p, q in zip(m.parameters(), m2.parameters()):
            assert torch.equal(p.grad, q.grad)


def test_whole_step_as_one_hip_graph(dev):
    """raw2logit_amd.graphs.StepGraph: forward + backward + gradient accumulation + BatchNorm's running statistics
    captured as ONE graph; every replay reproduces the eager step bit for bit, on new contents of the static buffers too."""
    import copy
    from raw2logit_amd.graphs import StepGraph
    B, S = 6, 96
    raws = [torch.from_numpy(orc.synth_raw(B, S, S, seed=50 + i, kind='scene')).to(dev) for i in range(3)]
    cots = [torch.randn((B, 3, S, S), device=dev, generator=torch.Generator(dev).manual_seed(7 + i)) for i in range(3)]
    m = ppt.ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=True).to(dev).train()
    m2 = copy.deepcopy(m)
    raw, cot = raws[0].clone(), cots[0].clone()
    g = StepGraph(m2, raw, cot)
    # the warm-up steps inside StepGraph moved m2's running statistics: start both modules from the same state again
    m2.load_state_dict(m.state_dict())
    for i in range(3):
        raw.copy_(raws[i])
        cot.copy_(cots[i])
        for p in m.parameters():
            p.grad = None
        y0 = m(raws[i])
        y0.backward(cots[i])
        y1 = g.replay()
        assert torch.equal(y0, y1)
        for p, q in zip(m.parameters(), m2.parameters()):
            assert torch.equal(p.grad, q.grad)
        for k, v in m.state_dict().items():
            assert torch.equal(v, m2.state_dict()[k]), k
    # with a loss behind the processor (a small classifier head, captured with it)
    head = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten()).to(dev)
    head2 = copy.deepcopy(head)
    target = torch.tensor([0, 1, 2, 3, 0, 1], device=dev)
    m3 = copy.deepcopy(m)
    g3 = StepGraph(m3, raw, None, loss=lambda rgb: torch.nn.functional.cross_entropy(head2(rgb), target), loss_modules=(head2,))
    m3.load_state_dict(m.state_dict())
    for p in list(m.parameters()) + list(head.parameters()):
        p.grad = None
    torch.nn.functional.cross_entropy(head(m(raws[2])), target).backward()
    for _ in range(2):              # (replays overwrite the gradients: the second one must not accumulate)
        g3.replay()
    for p, q in zip(m.parameters(), m3.parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-4, atol=1e-6)   # (the head's MIOpen kernels may differ under capture)
    for p, q in zip(head.parameters(), head2.parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-4, atol=1e-7)


def test_static_per_image_wrappers(golden, dev):
    pc.check_static_wrappers(golden, dev)


def test_weak_augmentation(dev):
    pc.check_augmentation(dev)


def test_error_behaviour(dev):
    pc.check_error_behaviour(dev)


def _bench(args, env=None, timeout=900):
    import json
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py')] + args, env=e, capture_output=True,
                       text=True, timeout=timeout, cwd=repo)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith('{')]
    assert len(lines) == 1, r.stdout
    return lines[0]


def test_bench_line_carries_roofline_and_static_c3(dev):
    """the default bench line: headline metric + roofline (with the source of `traffic` named) + the BASELINE
    config 3 sub-records (fused static chain, fraction of the HBM peak from HIP-event kernel times)"""
    out = _bench(['--steps', '5', '--warmup', '2', '--no-cpu-baseline'])
    assert out['n_gpus'] == 1 and out['unit'] == 'Mpix/s' and out['value'] > 0
    rf = out['roofline']
    assert rf['bound'] == 'hbm' and 0 < rf['frac'] < 1 and 'traffic_source' in rf
    recs = out['static_c3']
    shapes = [tuple(r['shape']) for r in recs]
    assert (256, 1024, 1024) in shapes and (1024, 512, 512) in shapes
    for r in recs:
        assert 0 < r['frac'] < 1 and r['avg_us'] > 0
        # algorithmic bytes over HIP-event time cannot beat the wall clock of the same call
        assert r['avg_us'] * 1e-3 <= r['ms_per_call_wall'] * 1.10, r    # (two passes of 20 launches: +-4 % between them)


def test_bench_two_ranks_nccl(dev):
    """`python bench.py --gpus 2` launches its own two ranks over RCCL (one per GPU)"""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    out = _bench(['--gpus', '2', '--steps', '5', '--warmup', '2', '--no-roofline'])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 128


def test_full_size_config4_microscopy(dev):
    """BASELINE config 4's ISP share at full size: 128 x 256 x 256 frames, Microscopy camera parameters (negative
    white-balance / colour-matrix entries: most pixels sit on the 1e-5 clip floor), BatchNorm in train mode.
    The float64 oracle needs minutes for the whole batch, so: (i) train-mode output is normalised; (ii) eval mode
    with the batch statistics reproduces the train-mode output (frames decouple); (iii) a 2-frame slice in that
    eval mode -- output and all 132 gradients -- against the oracle; (iv) eval-mode gradients of the whole batch
    are the sum of the gradients of its four quarters (the sums over 8.4 Mpix are right if the slices are)."""
    import copy
    B, H, W = 128, 256, 256
    raw_np = orc.synth_raw(B, H, W, seed=4, kind='uniform')
    cot_np = np.random.default_rng(5).standard_normal((B, 3, H, W)).astype(np.float32)
    raw, cot = torch.from_numpy(raw_np).to(dev), torch.from_numpy(cot_np).to(dev)
    m = ppt.ParametrizedProcessing(orc.MICROSCOPY_CAMERA_PARAMS, batch_norm_output=True).to(dev).train()
    m.batch_norm.momentum = 1.0                 # running statistics := this batch's statistics
    y_tr = m(raw)
    y_tr.backward(cot)
    n = B * H * W
    mean = y_tr.detach().double().mean(dim=(0, 2, 3)).cpu().numpy()
    var = y_tr.detach().double().var(dim=(0, 2, 3), unbiased=False).cpu().numpy()
    pc.report('config4/train-mode output: |mean|', np.abs(mean).max(), 1e-4)
    pc.report('config4/train-mode output: |var - 1|', np.abs(var - 1).max(), 1e-3)
    assert np.abs(mean).max() < 1e-4 and np.abs(var - 1).max() < 1e-3
    assert int(m.batch_norm.num_batches_tracked) == 1
    with torch.no_grad():
        m.batch_norm.running_var.mul_((n - 1) / n)          # unbiased -> the biased variance train mode used
    m.eval()
    for p in m.parameters():
        p.grad = None
    y_ev = m(raw)
    y_ev.backward(cot)
    e = (y_ev.detach() - y_tr.detach()).abs().max().item()
    pc.report('config4/eval mode with the batch statistics vs train mode', e, 2e-5)
    assert e <= 2e-5
    g_full = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    # (iii) 2-frame slice against the float64 oracle
    ms = copy.deepcopy(m)
    for p in ms.parameters():
        p.grad = None
    ys = ms(raw[:2])
    ys.backward(cot[:2])
    P = orc.IspParams(orc.MICROSCOPY_CAMERA_PARAMS, dtype=np.float64)
    bn = dict(training=False, running_mean=m.batch_norm.running_mean.double().cpu().numpy(),
              running_var=m.batch_norm.running_var.double().cpu().numpy())
    o, _, c = orc.parametrized_forward(raw_np[:2], P, bn=bn)
    # Microscopy parameters: |colour matrix . diag(white balance)| has row sums up to 10.9 (Drone: 5.0), the float32
    # round-off of the linear part grows with them: 2e-5 where the power law is well conditioned (achieved 9.7e-6)
    tol = pc.out_tolerance(c, True, base=2e-5)
    err = np.abs(ys.detach().cpu().numpy() - o)
    w = np.unravel_index((err / tol).argmax(), err.shape)
    pc.report('config4/2-frame slice: out vs float64 oracle', err[w], tol[w])
    assert np.all(err <= tol), (err.max(), w)
    assert torch.equal(ys.detach(), y_ev.detach()[:2])
    g, _, _ = orc.parametrized_backward(P, c, cot_np[:2])
    lo, _, _ = orc.parametrized_backward(P, c, cot_np[:2], clip_shift=1e-6)
    hi, _, _ = orc.parametrized_backward(P, c, cot_np[:2], clip_shift=-1e-6)
    # the float32 conditioning of the slice, measured: the oracle's formulas in float32 (the reference's own arithmetic)
    P32 = orc.IspParams(orc.MICROSCOPY_CAMERA_PARAMS, dtype=np.float32)
    _, _, c32 = orc.parametrized_forward(raw_np[:2], P32, bn=bn)
    g32, _, _ = orc.parametrized_backward(P32, c32, cot_np[:2])
    for k, og in g.items():
        got = pc.NAME2ATTR[k](ms).grad.detach().cpu().numpy().reshape(np.asarray(og).shape)
        # 3e-5 of the scale + the clip-flip allowance + twice the float32 oracle's own distance (round 4: 1e-2 of the scale)
        lim = pc.tight_grad_limit(og, lo[k], hi[k], g32[k])
        e = np.abs(got - og).max()
        pc.report(f'config4/2-frame slice: grad {k} vs float64 oracle', e, lim)
        assert e <= lim, (k, e, lim)
    # (iv) gradients add up over the batch
    acc = {k: torch.zeros_like(v) for k, v in g_full.items()}
    for q in range(4):
        mq = copy.deepcopy(m)
        for p in mq.parameters():
            p.grad = None
        sl = slice(32 * q, 32 * (q + 1))
        mq(raw[sl]).backward(cot[sl])
        for k, p in mq.named_parameters():
            acc[k] += p.grad
    for k in g_full:
        scale = g_full[k].abs().max().item() + 1e-6
        e = (acc[k] - g_full[k]).abs().max().item()
        pc.report(f'config4/grad {k}: whole batch vs sum of quarters', e, 1e-4 * scale)
        assert e <= 1e-4 * scale, (k, e, scale)


def test_bench_e2e_workloads(dev):
    """BASELINE configs 4 and 5 end to end on one GPU (ISP + task-model stand-in + loss + Adam step)"""
    for wl, B in (('e2e-microscopy', 128), ('e2e-drone', 64)):
        out = _bench(['--workload', wl, '--steps', '3', '--warmup', '2'])
        assert out['n_gpus'] == 1 and out['config']['global_batch'] == B and out['config']['frame'] == [256, 256]
        assert np.isfinite(out['loss']) and out['value'] > 0
        assert 0 < out['isp']['share_of_step'] < 1


@pytest.mark.parametrize('shape', [(2, 130, 516), (1, 70, 1024), (1, 36, 1028), (1, 24, 2048), (3, 66, 260),
                                   (1, 4, 8), (1, 200, 256)], ids=str)
def test_static_luma_chain_streaming_kernel(shape, dev):
    """the row-streaming luma-chain kernel (r2l_static_chain.h) on frames 1, 2, 4 and 8 wavefronts wide, several
    bands high, partially filled last wavefront: every chain it serves (bilinear / Malvar2004 x [sharpening_filter |
    unsharp_masking] x [gaussian_denoising | median_denoising]; unsharp_masking on frames wider than 1024 runs as plane
    passes), float32 / 16-bit / float64 frames, against the oracle (the reference's own arithmetic on scipy)"""
    from raw2logit_amd import functional as F_
    B, H, W = shape
    u = np.random.default_rng(W + H).integers(0, 4096, (B, H, W)).astype(np.uint16)
    u[:, : H // 3] = np.random.default_rng(1).integers(250, 262, (B, H // 3, W))     # around the black level
    raw_np = u.astype(np.float32) / np.float32(4095)
    for deb in ('bilinear', 'malvar2004'):
        for sh, dn in (('sharpening_filter', 'gaussian_denoising'), ('sharpening_filter', 'none'),
                       ('none', 'gaussian_denoising'), ('sharpening_filter', 'median_denoising'),
                       ('none', 'median_denoising'), ('unsharp_masking', 'none'),
                       ('unsharp_masking', 'gaussian_denoising'), ('unsharp_masking', 'median_denoising')):
            ref = orc.static_batch(raw_np, orc.DRONE_CAMERA_PARAMS, deb, sh, dn)
            out = F_.static_pipeline(torch.from_numpy(raw_np).to(dev), orc.DRONE_CAMERA_PARAMS, deb, sh, dn)
            err = np.abs(out.cpu().numpy() - ref)
            pc.report(f'static-chain/{shape}/{deb}+{sh}+{dn}/float32 frames', err.max(), 1e-5)
            assert err.max() <= 1e-5, (shape, deb, sh, dn, err.max(), np.unravel_index(err.argmax(), err.shape))
            out16 = F_.static_pipeline(torch.from_numpy(u).to(dev), orc.DRONE_CAMERA_PARAMS, deb, sh, dn, bits=12)
            assert torch.equal(out16, out), (shape, deb, sh, dn, '16-bit containers')
            ref64 = orc.static_batch(raw_np.astype(np.float64), orc.DRONE_CAMERA_PARAMS, deb, sh, dn)
            out64 = F_.static_pipeline(torch.from_numpy(raw_np.astype(np.float64)).to(dev), orc.DRONE_CAMERA_PARAMS,
                                       deb, sh, dn)
            e64 = np.abs(out64.cpu().numpy() - ref64).max()
            pc.report(f'static-chain/{shape}/{deb}+{sh}+{dn}/float64 frames', e64, 1e-5)
            assert e64 <= 1e-5, (shape, deb, sh, dn, 'float64 frames', e64)


@pytest.mark.parametrize('shape', [(3, 64, 64), (2, 70, 520), (1, 128, 1028), (2, 6, 8), (1, 192, 2048)], ids=str)
def test_backward_reads_the_luma_plane_the_forward_kept(shape, dev):
    """R2L_F_KEEP_LUMA: a forward that will be differentiated leaves the sharpened luma Y' in the workspace and
    kernel B1 loads it (tiles that fit exactly, ragged ones, border tiles, 16-bit containers) instead of recomputing
    raw -> Y -> Y'.  All parameter gradients against the recomputing kernel of the diagnostic build (same pixels,
    Y' rounded by another kernel) and, through check_frame_shapes / the golden cases, against the oracle."""
    import os
    B, H, W = shape
    P = orc.IspParams(orc.DRONE_CAMERA_PARAMS)
    P.perturb(11)
    u = np.rint(orc.synth_raw(B, H, W, seed=H + W, kind='scene').astype(np.float64) * 4095).astype(np.uint16)
    cot = torch.from_numpy(np.random.default_rng(H).standard_normal((B, 3, H, W)).astype(np.float32)).to(dev)
    for bn in (True, False):
        case = dict(camera='drone', track=False, additive=False, training=True, bn=bn)
        grads = {}
        for name, env, frames in (('kept', {}, 'f32'), ('kept16', {}, 'u16'), ('recomputed', {'R2L_BWD1_RECOMPUTE': '1'}, 'f32')):
            m = pc.make_module(case, P, dev)
            if frames == 'u16':
                m.raw_bits = 12
                raw = torch.from_numpy(u).to(dev)
            else:
                raw = torch.from_numpy(u.astype(np.float32) / np.float32(4095)).to(dev)
            os.environ.update(env)
            try:
                with pc.launch_shape_overrides(dev):
                    (m(raw) * cot).sum().backward()
            finally:
                for k in env:
                    del os.environ[k]
            grads[name] = {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters()}
        for n, ref in grads['recomputed'].items():
            assert np.array_equal(grads['kept'][n], grads['kept16'][n]), (shape, bn, n, '16-bit containers')
            e = np.abs(grads['kept'][n] - ref).max()
            lim = 2e-4 * (np.abs(ref).max() + 1e-6)
            pc.report(f'kept-luma/{shape}/bn={bn}/{n} vs recomputing kernel', e, lim)
            assert e <= lim, (shape, bn, n, e, lim)


@pytest.mark.parametrize('shape', [(2, 70, 520), (1, 40, 1028), (1, 36, 2048), (3, 66, 260), (1, 200, 256), (5, 18, 8),
                                   (2, 514, 512), (3, 4, 4), (2, 6, 1024), (4, 256, 256), (2, 8, 256), (9, 64, 64)], ids=str)
def test_backward_kernels_as_passes_over_planes(shape, dev):
    """Kernels B1 and B2 as passes of independent wavefronts over planes (r2l_param_plane_bwd.h: pointwise adjoints +
    dL/dY'' + chroma / gamma sums; the 25 blur-weight sums and the blur's mirror-padding adjoint in one pass over dL/dY''
    -- or as the two passes they were first, R2L_BWD_SPLIT_BLUR --; the sharpen's adjoint + sharpen / luma-stencil sums +
    final reduction and unfold) against the LDS tile kernels
    of the diagnostic build (R2L_BWD1_TILED, which takes both back; R2L_BWD2_TILED): all 132 gradients, float32 and
    16-bit frames, with and without train-mode BatchNorm, every band height and grid (one wavefront takes everything ...
    one item each), frames of 4 rows (every row a border row of the blur and of its adjoint) -- and, through the golden /
    frame-shape / fuzz suites, against the oracle."""
    import os
    B, H, W = shape
    P = orc.IspParams(orc.DRONE_CAMERA_PARAMS)
    P.perturb(19)
    u = np.rint(orc.synth_raw(B, H, W, seed=2 * H + W, kind='scene').astype(np.float64) * 4095).astype(np.uint16)
    cot = torch.from_numpy(np.random.default_rng(H + 2).standard_normal((B, 3, H, W)).astype(np.float32)).to(dev)
    for bn in (True, False):
        case = dict(camera='drone', track=False, additive=False, training=True, bn=bn)

        def run(env, frames, epilogue=None):
            m = pc.make_module(case, P, dev)
            if frames == 'u16':
                m.raw_bits = 12
                raw = torch.from_numpy(u).to(dev)
            else:
                raw = torch.from_numpy(u.astype(np.float32) / np.float32(4095)).to(dev)
            env = dict(env, R2L_BWD_PLANES='1')      # (the default only for batches of >= 4 Mi px)
            os.environ.update(env)
            try:
                with pc.launch_shape_overrides(dev):
                    if epilogue is not None:
                        from raw2logit_amd import augmentation as aug
                        m.fuse_rot90 = True                 # (rotations too through the kernels' own stores / fetches)
                        m.__dict__['_epilogue'] = epilogue
                        (m(raw) * aug.flip_rot(cot, *epilogue)).sum().backward()
                    else:
                        (m(raw) * cot).sum().backward()
            finally:
                for k in env:
                    del os.environ[k]
            return {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters()}

        ref = run({'R2L_BWD1_TILED': '1'}, 'f32')
        worst = 0.0
        for env, frames in (({}, 'f32'), ({}, 'u16'),
                            ({'R2L_BP_BAND': '6', 'R2L_HB_BAND': '12', 'R2L_B2S_BAND': '6'}, 'f32'),
                            ({'R2L_BP_BAND': '12', 'R2L_HB_BAND': '6', 'R2L_GRID_BWD1': '1', 'R2L_GRID_BWD2': '1', 'R2L_B2S_BAND': '18'}, 'f32'),
                            ({'R2L_BP_BAND': '1000', 'R2L_GRID_BWD1': '3', 'R2L_HB_BAND': '1000', 'R2L_B2S_BAND': '1000'}, 'u16'),
                            ({'R2L_GRID_BWD1': '7', 'R2L_GRID_BWD2': '5'}, 'f32'), ({'R2L_BWD2_TILED': '1'}, 'f32'),
                            # B1's blur-weight sums and B2's blur adjoint as two passes (r2l_bwd1_blur_block, r2l_bwd2_hp_block)
                            ({'R2L_BWD_SPLIT_BLUR': '1'}, 'f32'), ({'R2L_BWD_SPLIT_BLUR': '1', 'R2L_HP_BAND': '12'}, 'u16')):
            g = run(env, frames)
            for n, r in ref.items():
                e = np.abs(g[n] - r).max()
                lim = 2e-4 * (np.abs(r).max() + 1e-6)
                worst = max(worst, e / lim)
                assert e <= lim, (shape, bn, env, frames, n, e, lim)
        pc.report(f'bwd1-planes/{shape}/bn={bn}/all gradients vs tile kernel (fraction of 2e-4 relative)', worst, 1.0)
        # the output epilogue through the plane passes' fetch of grad_out: the same sums from the same numbers, bit for bit -- with
        # train-mode BatchNorm too since its backward sums are a plane pass (r2l_bnr_planes_block reads grad_out through the same
        # affine map as kernel B1; round 5.  bn_reduce, which it replaces here, walked the augmented tensors in memory order)
        plain = run({}, 'f32')
        for epilogue in ((True, False, 0), (False, True, 2)) + (((True, True, 1),) if H == W else ()):
            ge = run({}, 'f32', epilogue)
            for n in plain:
                assert np.array_equal(ge[n], plain[n]), (shape, bn, epilogue, n)
        if bn:       # ... and 16-bit frames through the epilogue form (r2l_launch_bnr_planes_epi_u16)
            plain16, ge16 = run({}, 'u16'), run({}, 'u16', (True, False, 0))
            for n in plain16:
                assert np.array_equal(ge16[n], plain16[n]), (shape, 'u16', n)


@pytest.mark.parametrize('shape', [(2, 70, 520), (1, 40, 1028), (1, 36, 2048), (3, 66, 260), (1, 200, 256),
                                   (5, 18, 8), (2, 514, 512), (3, 4, 4), (2, 6, 1024), (4, 256, 256)], ids=str)
def test_apply_pass_reads_the_luma_plane_the_statistics_pass_kept(shape, dev):
    """Train-mode BatchNorm with a backward to follow (R2L_F_KEEP_LUMA): the statistics pass keeps Y', and the apply pass
    is r2l_fwd_apply_block -- blur, chroma, colour code on the kept plane, independent wavefronts -- instead of a second
    run of the streaming forward.  Same arithmetic on the same Y' values: the output is BIT-identical to the streaming
    apply pass of the diagnostic build (R2L_FWD_APPLY_RECOMPUTE), whatever the band height, for float32 and 16-bit
    frames, with and without an output epilogue; the gradients behind it are the same bits too (kernel B1 reads the
    same plane)."""
    import os
    from raw2logit_amd import augmentation as aug
    B, H, W = shape
    P = orc.IspParams(orc.DRONE_CAMERA_PARAMS)
    P.perturb(13)
    u = np.rint(orc.synth_raw(B, H, W, seed=H + W, kind='scene').astype(np.float64) * 4095).astype(np.uint16)
    cot = torch.from_numpy(np.random.default_rng(W).standard_normal((B, 3, H, W)).astype(np.float32)).to(dev)
    case = dict(camera='drone', track=False, additive=False, training=True, bn=True)

    def run(env, frames, epilogue=None):
        m = pc.make_module(case, P, dev)
        if frames == 'u16':
            m.raw_bits = 12
            raw = torch.from_numpy(u).to(dev)
        else:
            raw = torch.from_numpy(u.astype(np.float32) / np.float32(4095)).to(dev)
        os.environ.update(env)
        try:
            with pc.launch_shape_overrides(dev):
                if epilogue is not None:
                    m.fuse_rot90 = True                 # (rotations too through the kernels' own stores / fetches)
                    m.__dict__['_epilogue'] = epilogue
                y = m(raw)
                (y * (cot if epilogue is None else aug.flip_rot(cot, *epilogue))).sum().backward()
        finally:
            for k in env:
                del os.environ[k]
        return y.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    # (statistics pass: the streaming kernel in every run, so that mean and 1/std are the same bits)
    ref, gref = run({'R2L_FWD_APPLY_RECOMPUTE': '1'}, 'f32')
    # the float64 oracle, so that "identical" is not "identically wrong"
    o, _, c = orc.parametrized_forward(u.astype(np.float32) / np.float32(4095), P.astype(np.float64), bn=pc.oracle_bn(case))
    tol = pc.out_tolerance(c, True)
    for env, frames in (({}, 'f32'), ({}, 'u16'), ({'R2L_FA_BAND': '2'}, 'f32'), ({'R2L_FA_BAND': '6'}, 'u16'),
                        ({'R2L_FA_BAND': '50'}, 'f32'), ({'R2L_FA_BAND': '1000'}, 'f32')):
        y, g = run(dict(env, R2L_FWD_STATS_STREAM='1'), frames)
        assert torch.equal(y, ref), (shape, env, frames, (y - ref).abs().max().item())
        for n in gref:
            assert torch.equal(g[n], gref[n]), (shape, env, frames, n)
        err = np.abs(y.cpu().numpy() - o)
        assert np.all(err <= tol), (shape, env, frames, err.max())
    pc.report(f'fwd-apply/{shape}/kept-luma apply pass vs streaming apply pass (bits differing)', 0.0, 0.0)
    for epilogue in ((True, False, 0), (False, True, 2)) + (((True, True, 1), (False, False, 3)) if H == W else ()):
        ye, ge = run({'R2L_FWD_STATS_STREAM': '1'}, 'f32', epilogue)
        yr, gr = run({'R2L_FWD_APPLY_RECOMPUTE': '1'}, 'f32', epilogue)
        assert torch.equal(ye, yr), (shape, epilogue)
        assert torch.equal(ye, aug.flip_rot(ref, *epilogue)), (shape, epilogue, 'vs permutation kernel')
        for n in gr:
            assert torch.equal(ge[n], gr[n]), (shape, epilogue, n)


@pytest.mark.parametrize('shape', [(2, 70, 520), (1, 40, 1028), (1, 36, 2048), (3, 66, 260), (1, 200, 256),
                                   (5, 18, 8), (2, 514, 512), (3, 4, 4), (2, 6, 1024), (4, 256, 256), (2, 8, 256)], ids=str)
def test_statistics_pass_is_a_luma_pass_and_a_pass_over_the_kept_plane(shape, dev):
    """Train-mode BatchNorm, statistics pass: r2l_fwd_luma_block (raw -> Y', independent wavefronts that compute the one
    column beyond their strip themselves) + r2l_fwd_apply_block<STATS> (the sums from the plane), instead of the streaming
    forward without output.  The PLANE is bit-identical to the one the streaming forward keeps (diagnostic build,
    R2L_FWD_STATS_STREAM), for float32 and 16-bit frames and whatever the band height; the statistics are the same sums
    added in another order, so outputs, running statistics and gradients agree to float32 rounding of (mean, 1/std) -- and
    the output is within tolerance of the float64 oracle.  (Default where frames are one 256-column strip wide.)"""
    import os
    from raw2logit_amd import _lib
    B, H, W = shape
    P = orc.IspParams(orc.DRONE_CAMERA_PARAMS)
    P.perturb(17)
    u = np.rint(orc.synth_raw(B, H, W, seed=H * 3 + W, kind='scene').astype(np.float64) * 4095).astype(np.uint16)
    cot = torch.from_numpy(np.random.default_rng(W + 1).standard_normal((B, 3, H, W)).astype(np.float32)).to(dev)
    case = dict(camera='drone', track=False, additive=False, training=True, bn=True)

    def run(env, frames):
        m = pc.make_module(case, P, dev)
        if frames == 'u16':
            m.raw_bits = 12
            raw = torch.from_numpy(u).to(dev)
        else:
            raw = torch.from_numpy(u.astype(np.float32) / np.float32(4095)).to(dev)
        os.environ.update(env)
        try:
            with pc.launch_shape_overrides(dev):
                lib, _ = _lib.library_for(raw)
                y = m(raw)
                off = lib.r2l_isp_step_offset(5, B, H, W)                      # R2L_STEP_LUMA
                plane = y.grad_fn.ws[off:off + 4 * B * H * W].view(torch.float32).clone()
                stats = y.grad_fn.ws[lib.r2l_isp_step_offset(0, B, H, W):][:56].view(torch.float64).clone()
                (y * cot).sum().backward()
        finally:
            for k in env:
                del os.environ[k]
        return (y.detach(), plane, stats, {n: p.grad.detach().clone() for n, p in m.named_parameters()},
                m.batch_norm.running_mean.clone(), m.batch_norm.running_var.clone())

    ref = run({'R2L_FWD_STATS_STREAM': '1'}, 'f32')
    o, _, c = orc.parametrized_forward(u.astype(np.float32) / np.float32(4095), P.astype(np.float64), bn=pc.oracle_bn(case))
    tol = pc.out_tolerance(c, True)
    worst = 0.0
    for env, frames in (({}, 'f32'), ({}, 'u16'), ({'R2L_FL_BAND': '6', 'R2L_FST_BAND': '12'}, 'f32'),
                        ({'R2L_FL_BAND': '48', 'R2L_FST_BAND': '6'}, 'u16'), ({'R2L_FL_BAND': '1000'}, 'f32'),
                        ({'R2L_GRID_FWD': '3'}, 'f32')):
        y, plane, stats, g, rm, rv = run(dict(env, R2L_FWD_STATS_SPLIT='1'), frames)   # (the default only where W <= 256)
        assert torch.equal(plane, ref[1]), (shape, env, frames, 'luma plane', (plane - ref[1]).abs().max().item())
        # sums of (x - 0.5) and (x - 0.5)^2 per channel: float32 pair sums per work item (about the lane's pivot), float64
        # from there on -- other bands round the float32 part differently: 1e-7 of a pixel's worth per pixel
        rel = ((stats[:6] - ref[2][:6]).abs() / float(B * H * W)).max().item()
        assert rel <= 1e-7, (shape, env, frames, 'float64 totals / n', rel)
        assert stats[6].item() == float(B * H * W)
        d = (y - ref[0]).abs().max().item()
        worst = max(worst, d)
        assert d <= 4e-6 * float(c['istd'].max()), (shape, env, frames, d)
        assert torch.allclose(rm, ref[4], rtol=1e-6, atol=1e-9) and torch.allclose(rv, ref[5], rtol=1e-6, atol=1e-12)
        for n in ref[3]:
            e = (g[n] - ref[3][n]).abs().max().item()
            assert e <= 1e-4 * (ref[3][n].abs().max().item() + 1e-6), (shape, env, frames, n, e)
        err = np.abs(y.cpu().numpy() - o)
        assert np.all(err <= tol), (shape, env, frames, err.max())
    pc.report(f'fwd-stats/{shape}/luma + statistics passes vs streaming statistics pass: out', worst,
              4e-6 * float(c['istd'].max()))


@pytest.mark.parametrize('shape', [(2, 70, 520), (1, 40, 1028), (1, 36, 2048), (3, 66, 260), (1, 200, 256),
                                   (5, 18, 8), (2, 514, 512)], ids=str)
def test_fused_forward_streaming_kernel(shape, dev):
    """the row-streaming forward (r2l_param_stream.h) on frames 1, 2, 4 and 8 wavefronts wide, several bands high,
    a partially filled last wavefront, perturbed weights (dense debayer / sharpen / blur): output without BatchNorm,
    with train-mode BatchNorm (statistics reduced inside the launch), 16-bit containers, against the float64 oracle;
    and against the tile kernel of the diagnostic build, which computes the same arithmetic in another order"""
    import os
    B, H, W = shape
    u = np.rint(orc.synth_raw(B, H, W, seed=H + W, kind='scene').astype(np.float64) * 4095).astype(np.uint16)
    raw_np = u.astype(np.float32) / np.float32(4095)
    P = orc.IspParams(orc.DRONE_CAMERA_PARAMS)
    P.perturb(7)
    case = dict(camera='drone', track=False, additive=False, training=True)
    for bn in (False, True):
        m = pc.make_module(dict(case, bn=bn), P, dev)
        with torch.no_grad():
            y = m(torch.from_numpy(raw_np).to(dev))
            m16 = pc.make_module(dict(case, bn=bn), P, dev)
            m16.raw_bits = 12
            y16 = m16(torch.from_numpy(u).to(dev))
        assert torch.equal(y, y16), (shape, bn, '16-bit containers')
        o, _, c = orc.parametrized_forward(raw_np, P.astype(np.float64), bn=pc.oracle_bn(dict(case, bn=bn)))
        o32, _, c32 = orc.parametrized_forward(raw_np, P.astype(np.float32), bn=pc.oracle_bn(dict(case, bn=bn)))
        # per pixel: BASELINE's 1e-5 (x 1/std under BatchNorm), or 6 standard deviations of the float32 oracle's own local
        # error pushed through the power law, whichever is larger (pc.sigma_limit, the criterion of the randomised sweeps) --
        # not the flat LOW_BAND_TOL of rounds 1-4 below a pre-gamma value of 3e-3
        floor = 1e-5 * (np.maximum(1.0, c['istd']).reshape(1, 3, 1, 1) if bn else 1.0)
        tol = np.broadcast_to(pc.sigma_limit(floor, P, c, c32, bn), o.shape)
        err = np.abs(y.cpu().numpy() - o)
        w = np.unravel_index((err / tol).argmax(), err.shape)
        pc.report(f'fwd-stream/{shape}/bn={bn}/out vs float64 oracle (limit: max(1e-5, 6 sigma of the float32 oracle))', err[w], tol[w])
        assert np.all(err <= tol), (shape, bn, err.max(), np.unravel_index(err.argmax(), err.shape))
        # ... and band by band at the distance the float32 oracle itself keeps from the float64 one
        pc.check_float32_distance(f'fwd-stream/{shape}/bn={bn}/out', y.cpu().numpy(), o, o32, c['rgb'] > pc.WELL_CONDITIONED)
        os.environ['R2L_FWD_TILED'] = '1'
        try:
            with pc.launch_shape_overrides(dev), torch.no_grad():
                yt = pc.make_module(dict(case, bn=bn), P, dev)(torch.from_numpy(raw_np).to(dev))
        finally:
            del os.environ['R2L_FWD_TILED']
        e2 = (y - yt).abs().max().item()
        pc.report(f'fwd-stream/{shape}/bn={bn}/streaming vs tile kernel', e2, 2e-5 * (float(c['istd'].max()) if bn else 1.0))
        assert e2 <= 2e-5 * (float(c['istd'].max()) if bn else 1.0), (shape, bn, e2)
