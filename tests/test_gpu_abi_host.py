"""SURVEY.md section 8b from the other side: a host with NO Python and NO torch.  tests/abi_host/r2l_host_check.c (plain C11: the
header compiles as C) links libr2l_isp.so, allocates with hipMalloc, and drives one training step of ParametrizedProcessing
(r2l_isp_step_fwd + r2l_isp_step_bwd) or one static chain (r2l_static_fwd) on golden cases the REFERENCE generated
(tests/golden/*.npz, exported here to flat binary files); it judges outputs, all 132 parameter gradients and BatchNorm's running
statistics itself and exits non-zero when one is off."""
import os
import struct
import subprocess

import numpy as np
import pytest

import conftest
import parity_checks as pc
from oracle import isp_oracle as orc
from oracle.gen_golden import build_params
from oracle.golden_cases import PARAM_CASES, STATIC_CASES

MAGIC = 0x52324c31
BN = {'none': 0, 'train': 1, 'eval': 2}
DEBAYER = {'bilinear': 0, 'malvar2004': 1}
SHARPEN = {'sharpening_filter': 1, 'unsharp_masking': 2}
DENOISE = {'gaussian_denoising': 1, 'median_denoising': 2}
GRAD_KEYS = ('black_level', 'white_balance', 'colour_correction', 'gamma_correct', 'debayer.weight',
             'sharpening_filter.weight', 'gaussian_blur.weight')


def _run(binary, path):
    r = subprocess.run([binary, path], capture_output=True, text=True, timeout=300)
    print(r.stdout + r.stderr)
    assert r.returncode == 0, (r.returncode, (r.stdout + r.stderr)[-2000:])
    assert r.stdout.strip().endswith('ok')


def _param_case(name):
    return next(c for c in PARAM_CASES if c['name'] == name)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['drone_bn_train', 'drone_nobn', 'micro_bn_train', 'identity_bn_train', 'drone_perturbed_bn_train',
                                  'drone_ragged_tile'])
def test_training_step_from_plain_c(name, golden, tmp_path):
    case = _param_case(name)
    assert not case.get('additive') and not case.get('track')
    g = golden['param_cases']
    raw, cot, packed, out = (g[f'{name}/{k}'] for k in ('raw', 'cot', 'packed', 'out'))
    B, H, W = raw.shape
    grad = np.concatenate([g[f'{name}/grad/{k}'].reshape(-1) for k in GRAD_KEYS]).astype(np.float32)
    assert grad.size == 132 and packed.size == 150
    mode = BN['train' if f'{name}/bn_running_mean_1' in g else 'none']
    rm = g[f'{name}/bn_running_mean_1'] if mode else np.zeros(3, np.float32)
    rv = g[f'{name}/bn_running_var_1'] if mode else np.ones(3, np.float32)
    # the limits of the golden suite, element by element (tests/parity_checks.py: check_param_case "vs reference (golden, float32)"):
    # outputs 2 x out_tolerance (1e-5, 7.5e-5 next to the clip floor, x 1/std behind BatchNorm); gradients 2 x grad_rtol x the
    # tensor's largest entry + 2 x what pixels within 1e-6 of a clip threshold can contribute.  The float64 oracle is the CHECKER here.
    P = build_params(case)
    assert np.array_equal(P.pack(), packed)                # the parameters the reference module held
    P64 = P.astype(np.float64)
    bn = pc.oracle_bn(case)
    _, _, cache = orc.parametrized_forward(raw, P64, track_stages=False, bn=bn)
    o_grads, _, _ = orc.parametrized_backward(P64, cache, cot)
    o_lo, _, _ = orc.parametrized_backward(P64, cache, cot, clip_shift=1e-6)
    o_hi, _, _ = orc.parametrized_backward(P64, cache, cot, clip_shift=-1e-6)
    out_lim = (2 * pc.out_tolerance(cache, case['bn'])).astype(np.float32)
    rtol = case.get('grad_rtol', pc.DEFAULT_GRAD_RTOL)
    grad_lim = []
    for k in GRAD_KEYS:
        ref = g[f'{name}/grad/{k}']
        flip = max(np.abs(np.asarray(o_lo[k]) - np.asarray(o_grads[k])).max(), np.abs(np.asarray(o_hi[k]) - np.asarray(o_grads[k])).max())
        grad_lim.append(np.full(ref.size, 2 * rtol * (np.abs(ref).max() + 1e-6) + 2 * flip, np.float32))
    grad_lim = np.concatenate(grad_lim)
    path = str(tmp_path / f'{name}.bin')
    with open(path, 'wb') as f:
        f.write(struct.pack('<8i', MAGIC, 0, B, H, W, mode, 0, 0))
        for a in (raw, packed, cot, out, grad, rm, rv, np.broadcast_to(out_lim, out.shape), grad_lim):
            f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())
    _run(conftest.build_host_check(), path)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['f32_drone_short_dark', 'f32_drone_short_scene', 'f32_drone_default_dark', 'f32_drone_default_uniform',
                                  'f32_micro_default_scene', 'f32_drone_malvar_short_dark', 'f32_drone_malvar_median_dark',
                                  'f32_drone_default_at_black'])
def test_static_chain_from_plain_c(name, golden, tmp_path):
    case = next(c for c in STATIC_CASES if c['name'] == name)
    g = golden['static_cases']
    raw = g[f'{name}/raw']
    assert raw.dtype == np.float32
    B, H, W = raw.shape
    ref = g[f'{name}/out_hwc_f64'].transpose(0, 3, 1, 2).astype(np.float32)
    bl, wb, ccm = orc.CAMERAS[case['camera']]
    cam = [float(v) for v in list(bl) + list(wb) + list(np.asarray(ccm).reshape(-1))]
    assert len(cam) == 16
    path = str(tmp_path / f'{name}.bin')
    with open(path, 'wb') as f:
        f.write(struct.pack('<8i', MAGIC, 1, B, H, W, DEBAYER[case['debayer']], SHARPEN.get(case['sharpening'], 0),
                            DENOISE.get(case['denoising'], 0)))
        f.write(np.ascontiguousarray(raw).tobytes())
        f.write(struct.pack('<17d', *cam, 2.2))
        f.write(np.ascontiguousarray(ref).tobytes())
        f.write(struct.pack('<d', 1e-5))
    _run(conftest.build_host_check(), path)
