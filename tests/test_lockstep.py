"""The kernels behind the headline numbers under AddressSanitizer + UBSan on the CPU.

The row-streaming forward, the passes over planes and the branch-free static loops fetch unconditionally from clamped addresses
and talk between lanes (DPP shifts, readfirstlane, shuffles, one barrier per row); the serial emulation cannot run them and the
GPU has no sanitizer on this pool.  tests/emul/r2l_lockstep.cpp compiles EVERY kernel in its device form with one fiber per lane
(cooperatively scheduled on a single host thread: tests/emul/r2l_lockstep_rt.h -- addresses and undefined behaviour are checked,
races between lanes are not modelled), built with -fsanitize=address,undefined; tests/lockstep_checks.py runs the GPU suite's
parity checks on it in a subprocess that preloads libasan, so that the numpy / torch buffers the kernels are handed carry
malloc redzones: a lane that reads or writes one element past `raw`, `out`, the workspace planes or the kernel's LDS block aborts
the run.  Reference behaviour being matched: ATen's bounds-checked kernels under `processing/pipeline_torch.py:187-217`."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import conftest  # noqa: E402

# the default CPU suite runs a slice of every group (~ 40 s after a 40 s build); R2L_LOCKSTEP_FULL=1 runs all 57 checks (~ 5 minutes)
GROUPS = [('planes',), ('shapes',), ('stream', 'passes'), ('static', 'canary')]


def _asan_runtime():
    p = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.fixture(scope='module')
def lockstep_lib():
    if _asan_runtime() is None:
        pytest.skip('no libasan.so next to gcc')
    return conftest.build_lockstep()


@pytest.mark.parametrize('groups', GROUPS, ids=['+'.join(g) for g in GROUPS])
def test_kernels_in_lock_step_under_address_sanitizer(groups, lockstep_lib):
    env = dict(os.environ, LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS='detect_leaks=0:abort_on_error=0',
               UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1', OMP_NUM_THREADS='1')
    if os.environ.get('R2L_LOCKSTEP_FULL') != '1':
        env['R2L_LOCKSTEP_QUICK'] = '1'
    r = subprocess.run([sys.executable, os.path.join(HERE, 'lockstep_checks.py'), lockstep_lib, *groups],
                       env=env, capture_output=True, text=True, timeout=3000)
    tail = '\n'.join(ln for ln in (r.stdout + r.stderr).splitlines() if not ln.startswith('[parity]'))[-6000:]
    assert 'AddressSanitizer' not in r.stdout + r.stderr and 'runtime error' not in r.stdout + r.stderr, tail
    assert r.returncode == 0, tail
    last = [ln for ln in r.stdout.splitlines() if 'lock-step checks passed' in ln]
    assert last and 'FAILED' not in last[-1], tail
    print(last[-1])


def test_two_gloo_ranks_on_the_lock_step_emulation(lockstep_lib, emulation, tmp_path):
    """SURVEY.md section 8e on the kernels that carry it on the GPU: two gloo ranks, each serving its half of the batch with the
    LOCK-STEP emulation under ASan -- the row-streaming statistics pass without its in-kernel finalisation, r2l_bn_finalize /
    r2l_bn_bwd_means over the gathered rows of both ranks, the apply pass, bn_reduce and the plane-pass backward (R2L_BWD_PLANES=1)
    in their phase-A / phase-B split -- against the single-process run of the whole batch (tests/test_distributed.py's worker and
    limits; there the serial emulation runs the tile kernels)."""
    import numpy as np
    import torch
    import torch.multiprocessing as mp
    import test_distributed as td
    from oracle import isp_oracle as orc
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
    world = 2
    add = dict(LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS='detect_leaks=0', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1',
               R2L_BWD_PLANES='1', OMP_NUM_THREADS='1')
    old = {k: os.environ.get(k) for k in add}
    os.environ.update(add)
    try:
        mp.spawn(td._worker, args=(world, td._free_port(), lockstep_lib, str(tmp_path)), nprocs=world, join=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    B, H, W = 4, 24, 40
    raw = torch.from_numpy(orc.synth_raw(B, H, W, seed=3, kind='scene'))
    cot = torch.from_numpy(np.random.default_rng(7).standard_normal((B, 3, H, W)).astype(np.float32))
    m = ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=True).train()     # (single process: the serial emulation)
    y = m(raw)
    (y * cot).sum().backward()
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).numpy()
    r = [np.load(os.path.join(str(tmp_path), f'rank{k}.npz')) for k in range(world)]
    y_sharded = np.concatenate([r[0]['y'], r[1]['y']])
    assert np.abs(y_sharded - y.detach().numpy()).max() < 2e-5
    assert np.array_equal(r[0]['g'], r[1]['g'])
    assert np.abs(r[0]['g'] - g).max() <= 2e-4 * (np.abs(g).max() + 1e-6)
    for k in range(world):
        np.testing.assert_allclose(r[k]['rm'], m.batch_norm.running_mean.numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(r[k]['rv'], m.batch_norm.running_var.numpy(), rtol=1e-5, atol=1e-7)
