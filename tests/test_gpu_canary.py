"""Guard zones around every buffer the kernels are handed (tests/guarded_arena.py): no out-of-bounds write within 64 KiB of
`raw`, `grad_out`, `out`, the 132-float gradient or the workspace (folded weights, partial sums, arrival counters, the three
(B,H,W) planes), and no result that depends on what lies outside -- or in a not-yet-written part of -- those buffers.
Runs the product library on the headline shape and ragged / minimal shapes through the same Python -> ctypes -> C-ABI path
every parity check takes.  Reference behaviour being matched: ATen's bounds-checked kernels under
`/root/reference/processing/pipeline_torch.py:187-217` and numpy's under `processing/pipeline_numpy.py:70-141`."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import guarded_arena as ga  # noqa: E402
import parity_checks as pc  # noqa: E402
from oracle import isp_oracle as orc  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return 'cuda:0'


def _step_bytes(B, H, W):
    px = B * H * W
    return int(px * (4 + 12 + 12 + 12 + 3 * 4) * 1.05) + (64 << 20)     # raw, out, grad_out, workspace planes + partials + guards


def _param_step(dev, B, H, W, bn, u16, seed, cot_seed=1, kind='uniform', train=True):
    """one fused forward + backward with every device buffer inside the arena -> {'out', 7 gradients, running statistics}"""
    raw_np = orc.synth_raw(B, H, W, seed=seed, kind=kind)
    u = np.rint(raw_np.astype(np.float64) * 4095).astype(np.uint16)
    cot_np = np.random.default_rng(cot_seed).standard_normal((B, 3, H, W)).astype(np.float32)
    P = orc.IspParams(orc.DRONE_CAMERA_PARAMS)
    P.perturb(11)
    proto = pc.make_module(dict(camera='drone', track=False, additive=False, training=train, bn=bn), P, dev)
    if u16:
        proto.raw_bits = 12

    def fn(arena):
        m = copy.deepcopy(proto)
        raw = arena.place(u.view(np.int16) if u16 else raw_np, 'raw')
        cot = arena.place(cot_np, 'grad_out')
        y = m(raw)
        y.backward(cot)
        res = {'out': y}
        for n, p in m.named_parameters():
            res['grad ' + n] = p.grad
        if bn:
            res['running_mean'], res['running_var'] = m.batch_norm.running_mean, m.batch_norm.running_var
        return res
    return fn


def test_headline_step_inside_guard_zones(dev):
    """BASELINE config 2 (64 x 512 x 512, BatchNorm train): streaming statistics pass, apply pass, the BatchNorm backward sums (recomputed from the planes), the three plane
    passes of the backward -- the kernels behind bench.py's line -- on the SHIPPED library, float32 frames and 16-bit containers."""
    from raw2logit_amd import _lib
    B, H, W = 64, 512, 512
    lib = _lib.device_library()
    assert lib.path == _lib.LIB_PATH
    for u16 in (False, True):
        fn = _param_step(dev, B, H, W, True, u16, seed=0)
        res, names = pc.kernels_launched(lib, lambda: ga.run_both(dev, _step_bytes(B, H, W), fn, f'64x512x512 u16={u16}'))
        sfx = '_u16_kernel' if u16 else '_kernel'
        for k in ('r2l_launch_fwd_stream_stats_w2', 'r2l_launch_fwd_apply', 'r2l_launch_bnr_planes', 'r2l_launch_bwd1_plane', 'r2l_launch_bwd2_sums'):
            assert names.get(k + sfx, 0) == 2, (k, sorted(names))
        assert names.get('r2l_launch_bwd1_blur_hp_kernel', 0) == 2 and 'r2l_launch_bn_reduce_kernel' not in names, sorted(names)
        assert bool(torch.isfinite(res['out']).all())


def test_config5_shard_inside_guard_zones(dev):
    """64 x 256 x 256 (config 5's share per GPU; the reference's tile size, dataset.py:92): frames one strip wide take the luma
    + statistics-from-the-plane kernels, and the plane-pass backward at its dispatch threshold"""
    B, H, W = 64, 256, 256
    for u16 in (False, True):
        ga.run_both(dev, _step_bytes(B, H, W), _param_step(dev, B, H, W, True, u16, seed=3), f'64x256x256 u16={u16}')


PLANE_ENV = {'R2L_BWD_PLANES': '1'}


@pytest.mark.parametrize('mode', ['fused-middle-pass', 'split-blur'])
def test_plane_passes_ragged_shapes_inside_guard_zones(mode, dev):
    """every shape of FRAME_SHAPES_PLANES (image edges at every distance from a strip / band boundary, 4- and 6-row frames,
    a last strip of 4 columns) through the streaming forward and the plane-pass backward (diagnostic build, same source;
    the launch record proves the kernels), with and without train-mode BatchNorm, float32 and 16-bit frames"""
    from raw2logit_amd import _lib
    env = dict(PLANE_ENV, **({'R2L_BWD_SPLIT_BLUR': '1'} if mode == 'split-blur' else {}))
    with pc.env_overrides(dev, env):
        lib = _lib.device_library()
        for n, (H, W) in enumerate(pc.FRAME_SHAPES_PLANES):
            for bn in (False, True):
                for u16 in (False, True):
                    fn = _param_step(dev, 2, H, W, bn, u16, seed=50 + n, kind='scene')
                    _, names = pc.kernels_launched(lib, lambda: ga.run_both(dev, _step_bytes(2, H, W), fn,
                                                                             f'planes {mode} {H}x{W} bn={bn} u16={u16}'))
                    assert any(k.startswith('r2l_launch_bwd1_plane') for k in names), sorted(names)
                    assert any(k.startswith('r2l_launch_bwd2_sums') for k in names), sorted(names)
                    assert not any('bwd1_saved' in k or k.startswith('r2l_launch_bwd2_kernel') for k in names), sorted(names)
        # other band heights (bands of 6 rows: every band boundary next to an image edge somewhere; one band for everything)
        for extra in ({'R2L_BP_BAND': '6', 'R2L_HB_BAND': '6', 'R2L_B2S_BAND': '6', 'R2L_HP_BAND': '6', 'R2L_FA_BAND': '6',
                       'R2L_FS_BAND': '8'},
                      {'R2L_BP_BAND': '1000', 'R2L_HB_BAND': '1000', 'R2L_B2S_BAND': '1000', 'R2L_HP_BAND': '1000',
                       'R2L_FA_BAND': '1000', 'R2L_FS_BAND': '1000'}):
            with pc.env_overrides(dev, dict(env, **extra)):
                for n, (H, W) in enumerate(pc.FRAME_SHAPES_PLANES[:8]):
                    ga.run_both(dev, _step_bytes(2, H, W), _param_step(dev, 2, H, W, True, False, seed=70 + n, kind='scene'),
                                f'planes {mode} {H}x{W} {sorted(extra.items())[0]}')


STREAM_SHAPES = [(2, 70, 520), (1, 40, 1028), (1, 36, 2048), (3, 66, 260), (1, 200, 256), (5, 18, 8), (2, 514, 512)]


@pytest.mark.parametrize('shape', STREAM_SHAPES, ids=str)
def test_streaming_forward_shapes_inside_guard_zones(shape, dev):
    """the 7 shapes of test_fused_forward_streaming_kernel (1, 2, 4 and 8 wavefronts per row, partially filled last wavefront,
    several bands) on the shipped library: statistics pass + apply pass (train), single pass (eval / no BatchNorm), and the
    backward the library picks at that size"""
    B, H, W = shape
    for bn, train in ((True, True), (True, False), (False, True)):
        for u16 in (False, True):
            ga.run_both(dev, _step_bytes(B, H, W), _param_step(dev, B, H, W, bn, u16, seed=H + W, kind='scene', train=train),
                        f'fwd-stream {shape} bn={bn} train={train} u16={u16}')


@pytest.mark.parametrize('W', [4, 8, 260, 1028, 2048])
def test_static_chains_inside_guard_zones(W, dev):
    """static kernels (branch-free row loops of r2l_static_stream.h / r2l_static_chain.h) at the widths where a strip is a
    few columns or ends 4 columns into a new strip: short chain with bilinear and Malvar2004, the train.py default chain,
    Malvar2004 + median; float32 frames and 16-bit containers; frames of 4, 10 and 50 rows"""
    from raw2logit_amd import functional as F_
    for H in (4, 10, 50):
        B = 3
        raw_np = orc.synth_raw(B, H, W, seed=W + H, kind='uniform')
        u = np.rint(raw_np.astype(np.float64) * 4095).astype(np.uint16)
        for chain in (('bilinear', 'none', 'none'), ('malvar2004', 'none', 'none'),
                      ('bilinear', 'sharpening_filter', 'gaussian_denoising'), ('malvar2004', 'sharpening_filter', 'median_denoising')):
            outs = {}
            for u16 in (False, True):
                def fn(arena):
                    raw = arena.place(u.view(np.int16) if u16 else raw_np, 'raw')
                    return {'out': F_.static_pipeline(raw, orc.DRONE_CAMERA_PARAMS, *chain, bits=12)}
                outs[u16] = ga.run_both(dev, (B * H * W * 40) + (32 << 20), fn, f'static {"+".join(chain)} {B}x{H}x{W} u16={u16}')['out']
            assert torch.equal(outs[False], outs[True]), (chain, H, W, '16-bit containers')
            ref = orc.static_batch(raw_np, orc.DRONE_CAMERA_PARAMS, *chain)
            e = np.abs(outs[False].numpy() - ref).max()
            assert e <= 1e-5, (chain, H, W, e)


def test_static_config3_shape_inside_guard_zones(dev):
    """config 3's frame shape (1024 x 1024; 16 frames) and 512 x 512 (64 frames): the bilinear and Malvar2004 short chains"""
    from raw2logit_amd import functional as F_
    for B, S in ((16, 1024), (64, 512)):
        raw_np = orc.synth_raw(B, S, S, seed=1, kind='uniform')
        for deb in ('bilinear', 'malvar2004'):
            def fn(arena):
                return {'out': F_.static_pipeline(arena.place(raw_np, 'raw'), orc.DRONE_CAMERA_PARAMS, deb, 'none', 'none')}
            ga.run_both(dev, B * S * S * 20 + (32 << 20), fn, f'static short chain {deb} {B}x{S}x{S}')
