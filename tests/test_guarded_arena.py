"""the guard-zone arena itself (tests/guarded_arena.py), on host memory: payloads are poisoned, an in-bounds write passes,
a write one element past a payload or before it is caught, a result that depends on unwritten memory is caught"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import guarded_arena as ga  # noqa: E402


def test_arena_serves_torch_empty_and_catches_out_of_bounds_writes():
    with ga.guarded('cpu', 1 << 20, 'nan') as arena:
        a = torch.empty(100, dtype=torch.float32, device='cpu')
        b = torch.empty((3, 5), dtype=torch.float64, device=torch.device('cpu'))
        z = torch.zeros(7, dtype=torch.int64, device='cpu')
        c = arena.place(np.arange(10, dtype=np.uint16), 'raw')
        assert len(arena.blocks) == 4 and bool(a.isnan().all()) and int(z.sum()) == 0
        assert bool((b.abs() > 1e300).all())       # (two float32 NaN patterns read as one float64: 2.2e307 -- poison all the same)
        assert c.dtype == torch.uint16 or c.dtype == torch.int16 or c.numel() == 10
        a.fill_(1.0)
        b.fill_(2.0)
        arena.check_guards('in-bounds writes')
        s, e, _ = arena.blocks[0]
        arena.buf[e] = 7                     # one byte past the first payload
        with pytest.raises(AssertionError, match='guard bytes overwritten'):
            arena.check_guards('one past the end')
        arena.buf[e] = 0x00                  # (restore: byte 0 of the little-endian 0x7fc00000 pattern)
        arena.check_guards('restored')
        arena.buf[s - 1] = 1                 # one byte before it
        with pytest.raises(AssertionError, match='before its start'):
            arena.check_guards('one before the start')
    assert torch.empty is not None and torch.empty.__module__ != ga.__name__       # the patch is undone
    t = torch.empty(4)
    assert t.shape == (4,)


def test_run_both_catches_results_that_depend_on_unwritten_memory():
    def good(arena):
        x = arena.place(np.ones(16, np.float32))
        y = torch.empty(16, dtype=torch.float32, device='cpu')
        y.copy_(x * 2)
        return {'y': y}
    ga.run_both('cpu', 1 << 20, good, 'good')

    def reads_unwritten(arena):
        y = torch.empty(16, dtype=torch.float32, device='cpu')     # never written: NaN in one run, 0 in the other
        return {'y': y + 1}
    with pytest.raises(AssertionError, match='never given|contains NaN'):
        ga.run_both('cpu', 1 << 20, reads_unwritten, 'bad')

    def reads_past_the_end(arena):
        x = arena.place(np.ones(16, np.float32))
        s, e, _ = arena.blocks[-1]
        beyond = arena.buf[s:e + 4].view(torch.float32)             # 17 elements: the last one is guard memory
        y = torch.empty(1, dtype=torch.float32, device='cpu')
        y.copy_(torch.nan_to_num(beyond, nan=5.0).sum().reshape(1))
        return {'y': y}
    with pytest.raises(AssertionError, match='never given'):
        ga.run_both('cpu', 1 << 20, reads_past_the_end, 'oob read')


def test_tile_kernels_of_the_host_emulation_inside_guard_zones():
    """the harness of tests/test_gpu_canary.py on host memory: the emulation (-DR2L_EMUL, the tile forms of the kernels) walks
    the same C ABI with every buffer between poisoned guard zones -- fused forward + backward on ragged shapes, a static chain"""
    import conftest
    import emul_hook
    prev = emul_hook.active()
    emul_hook.enable(conftest.build_emulation())
    try:
        import test_gpu_canary as tc
        from oracle import isp_oracle as orc
        from raw2logit_amd import functional as F_
        for (B, H, W, bn, u16) in ((2, 66, 132, True, False), (2, 6, 80, False, True), (1, 4, 4, False, False), (2, 70, 68, True, True)):
            r = ga.run_both('cpu', tc._step_bytes(B, H, W), tc._param_step('cpu', B, H, W, bn, u16, seed=5, kind='scene'),
                            f'emulation {B}x{H}x{W} bn={bn} u16={u16}')
            assert tuple(r['out'].shape) == (B, 3, H, W)
        raw_np = orc.synth_raw(3, 10, 8, seed=1, kind='uniform')
        for chain in (('malvar2004', 'none', 'none'), ('bilinear', 'sharpening_filter', 'gaussian_denoising')):
            def fn(arena):
                return {'out': F_.static_pipeline(arena.place(raw_np, 'raw'), orc.DRONE_CAMERA_PARAMS, *chain, bits=12)}
            out = ga.run_both('cpu', 3 * 10 * 8 * 40 + (32 << 20), fn, 'emulation static ' + '+'.join(chain))['out']
            assert np.abs(out.numpy() - orc.static_batch(raw_np, orc.DRONE_CAMERA_PARAMS, *chain)).max() <= 1e-5
    finally:
        emul_hook.enable(prev.path if prev is not None else None)
