#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (read-only, from /root/reference).

Runs only in the build container (the reference does not exist on the GPU box).  The reference's
modules are imported UNMODIFIED; absent third-party packages are replaced by empty sys.modules
stubs.  For the static path the three third-party functions the reference
calls (colour_demosaicing bilinear / Malvar2004, skimage rgb2yuv / yuv2rgb) are not installed, so
the published algorithms restated in oracle/isp_oracle.py are plugged into the stubbed names; all
remaining arithmetic of `processing()` is the reference's own code on real scipy.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Only data (inputs + expected outputs) is written; no reference source is copied."""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import isp_oracle as orc  # noqa: E402
from oracle.golden_cases import (PARAM_CASES, RAW2RGB_CASES, STATIC_CASES, STATIC_OPT_CASES, SAMPLE_STRIDE,  # noqa: E402
                                 AUX_CASES, aux_inputs, static_case_frames)
from oracle import harness  # noqa: E402


def import_reference():
    names = ['rawpy', 'skimage', 'skimage.filters', 'skimage.color', 'skimage.restoration',
             'colour_demosaicing', 'matplotlib', 'matplotlib.pyplot', 'dataset',
             'segmentation_models_pytorch', 'utils', 'utils.base', 'numpy.lib.function_base']
    for n in names:
        m = types.ModuleType(n)
        m.__all__ = []
        sys.modules[n] = m
    sys.modules['numpy.lib.function_base'].interp = np.interp
    sys.modules['skimage.filters'].unsharp_mask = orc.unsharp_mask     # restated, see its docstring
    for n in ['rgb2hsv', 'hsv2rgb']:
        setattr(sys.modules['skimage.color'], n, None)
    # third-party arithmetic: published algorithms restated in the oracle
    sys.modules['skimage.color'].rgb2yuv = orc.rgb2yuv
    sys.modules['skimage.color'].yuv2rgb = orc.yuv2rgb
    for n in ['denoise_tv_chambolle', 'denoise_tv_bregman', 'denoise_nl_means', 'denoise_bilateral',
              'denoise_wavelet', 'estimate_sigma']:
        setattr(sys.modules['skimage.restoration'], n, None)
    cd = sys.modules['colour_demosaicing']
    cd.demosaicing_CFA_Bayer_bilinear = orc.demosaicing_CFA_Bayer_bilinear
    cd.demosaicing_CFA_Bayer_Malvar2004 = orc.demosaicing_CFA_Bayer_Malvar2004
    cd.demosaicing_CFA_Bayer_Menon2007 = orc.demosaicing_CFA_Bayer_Menon2007
    sys.modules['dataset'].Subset = object
    sys.modules['utils.base'].np2torch = None
    sys.modules['utils.base'].torch2np = None
    # this repository ships its own drop-in `processing` package (a regular package, which would win over the
    # reference's namespace package whatever the path order): keep the repository off sys.path while importing
    saved_path = list(sys.path)
    sys.path = [REF] + [p for p in sys.path if os.path.abspath(p or os.getcwd()) != REPO]
    for name in [n for n in sys.modules if n == 'processing' or n.startswith('processing.')]:
        del sys.modules[name]
    cwd = os.getcwd()
    os.chdir(REF)          # pipeline_torch.py:5-6 chdir's if README.md is not in cwd
    import processing.pipeline_numpy as ppn
    import processing.pipeline_torch as ppt
    os.chdir(cwd)
    sys.path = saved_path
    assert ppn.__file__.startswith(REF) and ppt.__file__.startswith(REF), (ppn.__file__, ppt.__file__)
    return ppn, ppt


def set_reference_params(proc, P):
    with torch.no_grad():
        proc.black_level.copy_(torch.from_numpy(P.black_level))
        proc.white_balance.copy_(torch.from_numpy(P.white_balance))
        proc.colour_correction.copy_(torch.from_numpy(P.colour_correction))
        proc.gamma_correct.copy_(torch.from_numpy(P.gamma_correct))
        proc.debayer.weight.copy_(torch.from_numpy(P.debayer))
        proc.sharpening_filter.weight.copy_(torch.from_numpy(P.sharpening))
        proc.gaussian_blur.weight.copy_(torch.from_numpy(P.blur))
        if P.additive_layer is not None:
            proc.additive_layer.copy_(torch.from_numpy(P.additive_layer))


def build_params(case):
    """the oracle-side parameter set of a case (also used by the tests)."""
    P = orc.IspParams(orc.CAMERAS[case['camera']])
    if case['additive']:
        P.additive_layer = np.zeros((1, 3, 256, 256), dtype=np.float32)
    if case['perturb'] is not None:
        P.perturb(case['perturb'])
    return P


def sample(a, full):
    a = np.asarray(a)
    if full or a.ndim < 2:
        return a
    return np.ascontiguousarray(a[..., ::SAMPLE_STRIDE, ::SAMPLE_STRIDE])


def gen_param_cases(ppt, out):
    for case in PARAM_CASES:
        name = case['name']
        full = case.get('full', True)
        B, H, W = case['shape']
        raw_np = orc.synth_raw(B, H, W, seed=case['seed'], kind=case['kind'])
        P = build_params(case)
        proc = ppt.ParametrizedProcessing(camera_parameters=orc.CAMERAS[case['camera']],
                                          track_stages=case['track'], batch_norm_output=case['bn'])
        if case['additive']:
            ppt.append_additive_layer(proc)
        set_reference_params(proc, P)
        # packed parameters as the REFERENCE module holds them (checks build_params too)
        sd = proc.state_dict()
        packed = np.concatenate([sd[k].numpy().ravel() for k in
                                 ['black_level', 'white_balance', 'colour_correction', 'gamma_correct',
                                  'debayer.weight', 'sharpening_filter.weight', 'gaussian_blur.weight',
                                  'M_RGB_2_YUV', 'M_YUV_2_RGB']]).astype(np.float32)
        assert np.array_equal(packed, P.pack()), name
        if case['bn'] and not case['training']:
            # non-trivial running statistics for eval mode
            with torch.no_grad():
                proc.batch_norm.running_mean.copy_(torch.tensor([0.4, 0.45, 0.35]))
                proc.batch_norm.running_var.copy_(torch.tensor([0.03, 0.05, 0.04]))
        proc.train(case['training'])
        import copy
        proc64 = copy.deepcopy(proc)            # before the float32 run touches the running statistics
        raw = torch.from_numpy(raw_np).requires_grad_(True)
        y = proc(raw)
        rng = np.random.default_rng(1000 + case['seed'])
        cot = rng.standard_normal(tuple(y.shape)).astype(np.float32)
        (y * torch.from_numpy(cot)).sum().backward()
        pre = f'{name}/'
        if full:
            out[pre + 'raw'] = raw_np
            out[pre + 'cot'] = cot
        out[pre + 'packed'] = packed
        if P.additive_layer is not None:
            out[pre + 'additive'] = sample(P.additive_layer, full)
        out[pre + 'out'] = sample(y.detach().numpy(), full)
        out[pre + 'out_sum'] = np.float64(y.detach().double().sum().item())
        out[pre + 'grad_raw'] = sample(raw.grad.numpy(), full)
        out[pre + 'grad_raw_sum'] = np.float64(raw.grad.double().sum().item())
        for k, p in proc.named_parameters():
            g = p.grad.numpy()
            out[pre + 'grad/' + k] = sample(g, full) if k == 'additive_layer' else g
        for k, s in proc.stages.items():
            out[pre + 'stage/' + k] = sample(s.detach().numpy(), full)
            if case['track']:
                out[pre + 'stage_grad/' + k] = sample(s.grad.numpy(), full)
        out[pre + 'stage_keys'] = np.array(list(proc.stages.keys()))
        if case['bn']:
            bnm = proc.batch_norm
            out[pre + 'bn_running_mean_1'] = bnm.running_mean.numpy().copy()
            out[pre + 'bn_running_var_1'] = bnm.running_var.numpy().copy()
            out[pre + 'bn_nbt_1'] = np.int64(bnm.num_batches_tracked.item())
            if case['training'] and full:
                raw2_np = orc.synth_raw(B, H, W, seed=case['seed'] + 100, kind=case['kind'])
                with torch.no_grad():
                    y2 = proc(torch.from_numpy(raw2_np))
                out[pre + 'out_step2'] = y2.numpy()
                out[pre + 'bn_running_mean_2'] = bnm.running_mean.numpy().copy()
                out[pre + 'bn_running_var_2'] = bnm.running_var.numpy().copy()
                out[pre + 'bn_nbt_2'] = np.int64(bnm.num_batches_tracked.item())
        # The reference itself in float64 (same module, same inputs: a deep copy made before the float32 run, converted
        # with .double() and run under a float64 default dtype -- raw2rgb allocates with torch.zeros(), :261 / :272):
        # |ref32 - ref64| is how far the reference's OWN float32 arithmetic is from exact, per pixel; the tests bound
        # this library's distance from ref64 by a multiple of it.
        torch.set_default_dtype(torch.float64)
        try:
            p64 = proc64.double()
            p64.train(case['training'])
            raw64 = torch.from_numpy(raw_np).double().requires_grad_(True)
            y64 = p64(raw64)
            (y64 * torch.from_numpy(cot).double()).sum().backward()
        finally:
            torch.set_default_dtype(torch.float32)
        assert y64.dtype == torch.float64
        out[pre + 'out64'] = sample(y64.detach().numpy(), full)
        for k, p in p64.named_parameters():
            g = p.grad.numpy()
            out[pre + 'grad64/' + k] = sample(g, full) if k == 'additive_layer' else g
        # cross-check the oracle right here so a bad restatement is caught at generation time
        bn = None
        if case['bn']:
            bn = dict(training=case['training'],
                      running_mean=np.array([0.4, 0.45, 0.35]) if not case['training'] else np.zeros(3),
                      running_var=np.array([0.03, 0.05, 0.04]) if not case['training'] else np.ones(3))
        yo, so, cache = orc.parametrized_forward(raw_np, P, track_stages=case['track'], bn=bn)
        go, gro, _ = orc.parametrized_backward(P, cache, cot)
        err = np.abs(yo - y.detach().numpy()).max()
        gerr = max(np.abs(go[k] - p.grad.numpy()).max() / (np.abs(p.grad.numpy()).max() + 1e-12)
                   for k, p in proc.named_parameters())
        print(f'  {name:32s} oracle-vs-reference: out {err:.2e}  grads(rel) {gerr:.2e}  '
              f'grad_raw {np.abs(gro - raw.grad.numpy()).max():.2e}')


def gen_raw2rgb(ppt, out):
    raw_np = orc.synth_raw(2, 8, 12, seed=42, kind='uniform')
    out['raw2rgb/raw'] = raw_np
    bl = [0.0625, 0.0626, 0.0627, 0.0628]
    out['raw2rgb/black_level'] = np.asarray(bl, dtype=np.float32)
    for reduce_size, oc in RAW2RGB_CASES:
        for use_bl in (False, True):
            raw = torch.from_numpy(raw_np).requires_grad_(True)
            blt = torch.tensor(bl, requires_grad=True) if use_bl else None
            y = ppt.raw2rgb(raw, black_level=blt, reduce_size=reduce_size, out_channels=oc)
            rng = np.random.default_rng(7)
            cot = rng.standard_normal(tuple(y.shape)).astype(np.float32)
            (y * torch.from_numpy(cot)).sum().backward()
            key = f'raw2rgb/r{int(reduce_size)}_c{oc}_bl{int(use_bl)}/'
            out[key + 'out'] = y.detach().numpy()
            out[key + 'cot'] = cot
            out[key + 'grad_raw'] = raw.grad.numpy()
            if use_bl:
                out[key + 'grad_bl'] = blt.grad.numpy()
    # RawToRGB module (pipeline_torch.py:43-80), mode 'none' of train.py:200-203
    m = ppt.RawToRGB(reduce_size=True, out_channels=3, track_stages=False, normalize_mosaic=None)
    out['raw2rgb/module_default'] = m(torch.from_numpy(raw_np)).numpy()


def gen_static(ppn, out):
    for case in STATIC_CASES:
        name = case['name']
        B, H, W = case['shape']
        # frames in the dtype the case names: float32 is what the reference's datasets hand to processing()
        # (utils/dataset_utils.py:18-26 -> dataset.py:86-87), float64 what a DNG's uint16 / (2**bits-1) gives
        raw_np, u16 = static_case_frames(case)
        cam = orc.CAMERAS[case['camera']]
        res = []
        for img in raw_np:
            frame = img.copy()
            o = ppn.processing(frame, *cam, debayer=case['debayer'],
                               sharpening=case['sharpening'], denoising=case['denoising'])
            assert frame.dtype == raw_np.dtype        # remove_blacklv worked in place, in the frame's dtype
            res.append(o)
        res = np.stack(res)                                  # (B,H,W,3) float64
        out[f'{name}/raw'] = raw_np
        if u16 is not None:
            out[f'{name}/u16'] = u16
        out[f'{name}/out_hwc_f64'] = res
        # RawProcessingPipeline wrapper (pipeline_numpy.py:36-67): (3,H,W) float32 tensor
        pipe = ppn.RawProcessingPipeline(cam, debayer=case['debayer'], sharpening=case['sharpening'],
                                         denoising=case['denoising'])
        t = pipe(raw_np[0].copy())
        assert t.dtype == torch.float32 and tuple(t.shape) == (3, H, W)
        out[f'{name}/pipeline_chw_f32'] = t.numpy()
        mine = orc.static_batch(raw_np, cam, case['debayer'], case['sharpening'], case['denoising'])
        other = np.float64 if raw_np.dtype == np.float32 else np.float32
        cross = orc.static_batch(raw_np.astype(other), cam, case['debayer'], case['sharpening'], case['denoising'])
        print(f'  {name:32s} {str(raw_np.dtype):8s} oracle-vs-reference: '
              f'{np.abs(mine - res.transpose(0, 3, 1, 2)).max():.2e}   (frames cast to {other.__name__}: '
              f'{np.abs(cross - res.transpose(0, 3, 1, 2)).max():.2e})')


def gen_static_opts(ppn, out):
    """processing() with its numeric arguments away from the defaults (pipeline_numpy.py:70-73, :117-122): the reference's own
    function decides what they mean (scipy's gaussian_filter / median_filter and its fft_denoising are its own calls; the
    unsharp mask behind sharp_radius / sharp_amount is the restated skimage function, unpinned like every unsharp case)"""
    for case in STATIC_OPT_CASES:
        name = case['name']
        raw_np, _ = static_case_frames(case)
        cam = orc.CAMERAS[case['camera']]
        res = np.stack([ppn.processing(img.copy(), *cam, debayer=case['debayer'], sharpening=case['sharpening'],
                                       denoising=case['denoising'], **case['opts']) for img in raw_np])
        out[f'{name}/raw'] = raw_np
        out[f'{name}/out_hwc_f64'] = res
        mine = orc.static_batch(raw_np, cam, case['debayer'], case['sharpening'], case['denoising'], **case['opts'])
        print(f'  {name:32s} {str(raw_np.dtype):8s} oracle-vs-reference: {np.abs(mine - res.transpose(0, 3, 1, 2)).max():.2e}')


def gen_harness(ppt, out):
    torch.manual_seed(0)
    raw_np = orc.synth_raw(4, 32, 32, seed=21, kind='scene')
    labels = np.array([0, 3, 1, 2], dtype=np.int64)
    proc = ppt.ParametrizedProcessing(camera_parameters=orc.DRONE_CAMERA_PARAMS, track_stages=False,
                                      batch_norm_output=True)
    proc.train()
    clf = harness.make_classifier()
    logits, loss = harness.train_step(proc, clf, torch.from_numpy(raw_np), torch.from_numpy(labels))
    out['harness/raw'] = raw_np
    out['harness/labels'] = labels
    out['harness/logits'] = logits.numpy()
    out['harness/loss'] = np.float32(loss.item())
    for k, p in proc.named_parameters():
        out['harness/after_step/' + k] = p.detach().numpy().copy()
    # eval-mode logits after the step (BN running statistics in use: model.py:136-142)
    proc.eval()
    with torch.no_grad():
        out['harness/logits_eval_after'] = clf(proc(torch.from_numpy(raw_np))).numpy()


def gen_aux(_mod, out):
    """utils/ssim.py SSIM(window_size=11) and utils/base.py l2_regularization, run as they are (pure torch;
    loaded by file path because the `utils` package itself needs MLflow/Lightning)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_ssim', os.path.join(REF, 'utils', 'ssim.py'))
    ref_ssim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_ssim)
    for case in AUX_CASES:
        x, y = aux_inputs(case)
        ty = torch.from_numpy(y).requires_grad_(True)
        v = ref_ssim.SSIM(window_size=11)(torch.from_numpy(x), ty)
        v.backward()
        name = case['name']
        out[f'{name}/ssim'] = np.float32(v.item())
        out[f'{name}/ssim_grad'] = ty.grad.numpy().copy()
        ty2 = torch.from_numpy(y).requires_grad_(True)
        l2 = ((torch.from_numpy(x) - ty2) ** 2).sum()          # utils/base.py:342-343
        l2.backward()
        out[f'{name}/l2'] = np.float32(l2.item())
        out[f'{name}/l2_grad'] = ty2.grad.numpy().copy()
        ov, og = orc.ssim(x, y)
        print(f'  {name:24s} oracle-vs-reference: ssim {abs(ov - v.item()):.2e}  grad {np.abs(og - ty.grad.numpy()).max():.2e}')


def main():
    ppn, ppt = import_reference()
    gold = os.path.join(REPO, 'tests', 'golden')
    os.makedirs(gold, exist_ok=True)
    for fname, fn, mod in [('param_cases.npz', gen_param_cases, ppt), ('raw2rgb.npz', gen_raw2rgb, ppt),
                           ('static_cases.npz', gen_static, ppn), ('static_opts.npz', gen_static_opts, ppn), ('harness.npz', gen_harness, ppt),
                           ('aux_losses.npz', gen_aux, None)]:
        if len(sys.argv) > 1 and fname.split('.')[0] not in sys.argv[1:]:
            continue          # `python oracle/gen_golden.py static_cases` regenerates one file
        out = {}
        print(fname)
        fn(mod, out)
        path = os.path.join(gold, fname)
        np.savez_compressed(path, **out)
        print(f'  -> {path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(out)} arrays')


if __name__ == '__main__':
    main()
