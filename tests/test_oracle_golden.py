"""The oracle (oracle/isp_oracle.py) against the golden vectors generated from the REFERENCE
(oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import isp_oracle as orc
from oracle.golden_cases import PARAM_CASES, RAW2RGB_CASES, STATIC_CASES, STATIC_OPT_CASES, SAMPLE_STRIDE
from oracle.gen_golden import build_params


def _sample(a, full):
    return a if full else a[..., ::SAMPLE_STRIDE, ::SAMPLE_STRIDE]


def _bn(case):
    if not case['bn']:
        return None
    if case['training']:
        return dict(training=True, running_mean=np.zeros(3), running_var=np.ones(3))
    return dict(training=False, running_mean=np.array([0.4, 0.45, 0.35]),
                running_var=np.array([0.03, 0.05, 0.04]))


@pytest.mark.parametrize('case', PARAM_CASES, ids=[c['name'] for c in PARAM_CASES])
def test_parametrized_oracle_matches_reference(case, golden):
    g = golden['param_cases']
    pre = case['name'] + '/'
    full = case.get('full', True)
    B, H, W = case['shape']
    raw = orc.synth_raw(B, H, W, seed=case['seed'], kind=case['kind'])
    cot = np.random.default_rng(1000 + case['seed']).standard_normal((B, 3, H, W)).astype(np.float32)
    if full:
        assert np.array_equal(raw, g[pre + 'raw'])
        assert np.array_equal(cot, g[pre + 'cot'])
    P = build_params(case)
    assert np.array_equal(P.pack(), g[pre + 'packed'])
    bn = _bn(case)
    # float32 oracle, same op order as the reference: what remains is summation-order round-off.  (With
    # most pixels within a decade of the 1e-5 clip floor -- Microscopy parameters -- the gradient is
    # ill-conditioned: d/dx x^(1/gamma) ~ x^-0.55, so 1e-7 of float32 noise before the clip moves the
    # parameter gradients by ~1e-3 relative; a float64 oracle differs from the reference by that much.)
    P64 = P
    out, stages, cache = orc.parametrized_forward(raw, P64, track_stages=case['track'], bn=bn)
    grads, graw, sgr = orc.parametrized_backward(P64, cache, cot, stage_grads=True)
    # the reference clips at 1e-5 before x^(1/gamma): d/dx <= 241 there, so float32 noise of a few 1e-7
    # before the clip is up to ~1e-4 after it (and x 1/std after BatchNorm).  Pixels away from the
    # clip floor must agree to 1e-5.
    pre_gamma = cache['rgb']
    tol = np.where(pre_gamma > 1e-3, 1e-5, 2e-4)
    if case['bn']:
        tol = tol * np.maximum(1.0, cache['istd'].reshape(1, 3, 1, 1))
    assert np.all(np.abs(_sample(out, full) - g[pre + 'out']) <= _sample(tol, full)), \
        np.abs(_sample(out, full) - g[pre + 'out']).max()
    assert list(g[pre + 'stage_keys']) == [k for k in orc.STAGE_ORDER if k in stages]
    for k in stages:
        ref = g[pre + 'stage/' + k]
        got = _sample(stages[k], full)
        if k in ('gamma_correct', 'noise'):
            assert np.all(np.abs(got - ref) <= _sample(np.where(pre_gamma > 1e-3, 1e-5, 2e-4), full)), k
        else:
            assert np.abs(got - ref).max() <= 1e-5, (k, np.abs(got - ref).max())
    for k, ref in ((k[len(pre) + 5:], g[k]) for k in g.files if k.startswith(pre + 'grad/')):
        got = grads[k]
        if k == 'additive_layer':
            got = _sample(got, full)
        scale = np.abs(ref).max() + 1e-6
        assert np.abs(got - ref).max() <= 2e-3 * scale, (k, np.abs(got - ref).max(), scale)
    gr_ref = g[pre + 'grad_raw']
    assert np.abs(_sample(graw, full) - gr_ref).max() <= 5e-3 * (np.abs(gr_ref).max() + 1e-6)
    if case['track']:
        for k in stages:
            ref = g[pre + 'stage_grad/' + k]
            got = _sample(sgr[k], full)
            assert np.abs(got - ref).max() <= 5e-3 * (np.abs(ref).max() + 1e-6), k
    if case['bn'] and case['training']:
        np.testing.assert_allclose(bn['running_mean'], g[pre + 'bn_running_mean_1'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(bn['running_var'], g[pre + 'bn_running_var_1'], rtol=1e-5, atol=1e-6)
        assert bn['num_batches_tracked'] == int(g[pre + 'bn_nbt_1'])
        if full:
            raw2 = orc.synth_raw(B, H, W, seed=case['seed'] + 100, kind=case['kind'])
            out2, _, c2 = orc.parametrized_forward(raw2, P64, track_stages=case['track'], bn=bn)
            tol2 = np.where(c2['rgb'] > 1e-3, 1e-5, 2e-4) * np.maximum(1.0, c2['istd'].reshape(1, 3, 1, 1))
            assert np.all(np.abs(out2 - g[pre + 'out_step2']) <= tol2)
            np.testing.assert_allclose(bn['running_mean'], g[pre + 'bn_running_mean_2'], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(bn['running_var'], g[pre + 'bn_running_var_2'], rtol=1e-5, atol=1e-6)


def test_raw2rgb_oracle_matches_reference(golden):
    g = golden['raw2rgb']
    raw = g['raw2rgb/raw']
    bl = g['raw2rgb/black_level']
    for reduce_size, oc in RAW2RGB_CASES:
        for use_bl in (False, True):
            key = f'raw2rgb/r{int(reduce_size)}_c{oc}_bl{int(use_bl)}/'
            out = orc.raw2rgb(raw, bl if use_bl else None, reduce_size, oc)
            assert np.array_equal(out, g[key + 'out'])          # pure re-indexing: bit exact
            graw, gbl = orc.raw2rgb_vjp(g[key + 'cot'], raw.shape[1], raw.shape[2], reduce_size, oc)
            np.testing.assert_allclose(graw, g[key + 'grad_raw'], rtol=0, atol=1e-7)
            if use_bl:
                np.testing.assert_allclose(gbl, g[key + 'grad_bl'], rtol=1e-5)
    assert np.array_equal(orc.raw2rgb(raw), g['raw2rgb/module_default'])


@pytest.mark.parametrize('case', STATIC_CASES, ids=[c['name'] for c in STATIC_CASES])
def test_static_oracle_matches_reference(case, golden):
    from oracle.golden_cases import static_case_frames
    g = golden['static_cases']
    raw, u16 = static_case_frames(case)          # in the dtype the reference was handed (float32 | float64)
    assert raw.dtype == g[case['name'] + '/raw'].dtype == np.dtype(case.get('dtype', 'float64'))
    assert np.array_equal(raw, g[case['name'] + '/raw'])
    if u16 is not None:
        assert np.array_equal(u16, g[case['name'] + '/u16'])
    out = np.stack([orc.processing(img.copy(), *orc.CAMERAS[case['camera']],
                                   debayer=case['debayer'], sharpening=case['sharpening'],
                                   denoising=case['denoising']) for img in raw])
    np.testing.assert_allclose(out, g[case['name'] + '/out_hwc_f64'], rtol=0, atol=1e-12)
    if case['kind'] == 'at_black':
        # the fixture can tell the two black-level arithmetics apart (frames sitting on the black level)
        other = np.float64 if raw.dtype == np.float32 else np.float32
        wrong = np.stack([orc.processing(img.astype(other), *orc.CAMERAS[case['camera']], debayer=case['debayer'],
                                         sharpening=case['sharpening'], denoising=case['denoising']) for img in raw])
        assert np.abs(wrong - g[case['name'] + '/out_hwc_f64']).max() > 1e-4
    chw = orc.static_batch(raw[:1], orc.CAMERAS[case['camera']], case['debayer'], case['sharpening'],
                           case['denoising'])[0]
    np.testing.assert_allclose(chw, g[case['name'] + '/pipeline_chw_f32'], rtol=0, atol=1e-7)


@pytest.mark.parametrize('case', STATIC_OPT_CASES, ids=[c['name'] for c in STATIC_OPT_CASES])
def test_static_oracle_numeric_arguments_match_reference(case, golden):
    """processing()'s numeric arguments (pipeline_numpy.py:70-73, :117-122) away from their defaults: the oracle against what
    the reference's own processing() returned for them (tests/golden/static_opts.npz)"""
    from oracle.golden_cases import static_case_frames
    g = golden['static_opts']
    raw, _ = static_case_frames(case)
    assert np.array_equal(raw, g[case['name'] + '/raw'])
    out = np.stack([orc.processing(img.copy(), *orc.CAMERAS[case['camera']], debayer=case['debayer'],
                                   sharpening=case['sharpening'], denoising=case['denoising'], **case['opts']) for img in raw])
    np.testing.assert_allclose(out, g[case['name'] + '/out_hwc_f64'], rtol=0, atol=1e-12)
    if case['name'] != 'opt_ignored_on_short_chain':     # the fixture can tell the options from the defaults
        dflt = np.stack([orc.processing(img.copy(), *orc.CAMERAS[case['camera']], debayer=case['debayer'],
                                        sharpening=case['sharpening'], denoising=case['denoising']) for img in raw])
        assert np.abs(dflt - g[case['name'] + '/out_hwc_f64']).max() > 1e-4


def test_reference_constants_pin_the_third_party_restatements():
    """pipeline_torch.py restates bilinear (K_G, K_RB :13-19), the YUV matrices (:21-26) and the
    sigma=0.5 blur (:28-32); those are the only pins the reference holds on the third-party code."""
    assert np.abs(np.linalg.inv(orc.M_RGB_2_YUV.astype(np.float64)) - orc.M_YUV_2_RGB).max() < 1e-7
    assert np.abs(orc.RGB_FROM_YUV - orc.M_YUV_2_RGB).max() < 1e-7
    w = orc.gaussian_kernel1d(0.5)
    assert np.abs(np.outer(w, w) - orc.K_BLUR).max() < 5e-6
    # bilinear as masked-plane convolution == the torch Debayer on the zero-filled mosaic (interior)
    rng = np.random.default_rng(0)
    cfa = rng.random((12, 12))
    d = orc.demosaicing_CFA_Bayer_bilinear(cfa)
    P = orc.IspParams(dtype=np.float64)
    mosaic = orc.raw2rgb(cfa[None], None, reduce_size=False)
    t = orc.conv2d(mosaic, P.debayer, 'mirror')[0].transpose(1, 2, 0)
    assert np.abs(d[1:-1, 1:-1] - t[1:-1, 1:-1]).max() < 1e-12


def test_aux_loss_oracle_matches_reference(golden):
    """oracle SSIM / l2 (value and gradient) against the reference's utils/ssim.py / utils/base.py outputs"""
    from oracle.golden_cases import AUX_CASES, aux_inputs
    g = golden['aux_losses']
    for case in AUX_CASES:
        x, y = aux_inputs(case)
        v, grad = orc.ssim(x, y)
        assert abs(v - float(g[case['name'] + '/ssim'])) < 5e-7
        assert np.abs(grad - g[case['name'] + '/ssim_grad']).max() < 1e-7
        l2, l2g = orc.l2_regularization(x, y)
        assert abs(l2 - float(g[case['name'] + '/l2'])) <= 2e-6 * l2
        assert np.abs(l2g - g[case['name'] + '/l2_grad']).max() < 1e-6


def test_philox_restatement_against_the_published_known_answers():
    """Philox4x32-10 known-answer vectors of the Random123 distribution (kat_vectors: philox4x32 10) pin the
    oracle's generator; the Box-Muller deviates built on it have the moments of N(0,1)."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        assert tuple(int(v) for v in orc.philox4x32_10(np.array([c]), k)[0]) == want
    z = orc.philox_normal(1 << 20, 1234).astype(np.float64)
    assert abs(z.mean()) < 4e-3 and abs(z.std() - 1) < 4e-3 and abs((z ** 4).mean() - 3) < 5e-2
    assert np.array_equal(orc.philox_normal(10, 5, 3)[:6], orc.philox_normal(6, 5, 3))


def test_normalize_division_sequence_is_correctly_rounded():
    """r2l_static_out (csrc/r2l_static_kernels.h) divides by std as q0 = RN(t r), RN(q0 + (t - q0 s) r) with
    r = RN(1/s): restated here in numpy (the remainder is exact in float64), 3 * 10^6 quotients per std of the
    reference's Normalize constants (train.py:157-162, :185-187) and random ones, against float32 division"""
    rng = np.random.default_rng(0)
    stds = [0.12, 0.11, 0.08, 0.05, 0.097, 0.0423, 0.008] + list(rng.uniform(0.01, 2.0, 5))
    n = 1_000_000
    for s in stds:
        s = np.float32(s)
        r = np.float32(1) / s
        t = np.concatenate([rng.uniform(-1, 1, n).astype(np.float32),
                            rng.uniform(0, 1, n).astype(np.float32) - np.float32(0.35),
                            (10.0 ** rng.uniform(-9, 0, n)).astype(np.float32)])
        q0 = (t * r).astype(np.float32)
        rem = (t.astype(np.float64) - q0.astype(np.float64) * np.float64(s)).astype(np.float32)
        q1 = (q0.astype(np.float64) + rem.astype(np.float64) * np.float64(r)).astype(np.float32)
        assert np.array_equal(q1, (t / s).astype(np.float32)), float(s)
