"""Short runs of the randomised sweeps (tests/fuzz_gpu.py, tests/fuzz_more.py) as part of the suites: on the host
emulation of the kernel source for the CPU suite, on the gfx950 build for the GPU suite.  The sweeps draw random
frame shapes (including the awkward ones around tile boundaries), cameras, BatchNorm modes and containers and
compare with the oracle; longer runs: `SECONDS=120 python tests/fuzz_gpu.py`."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(script, seconds, seed, hide_gpu):
    env = dict(os.environ, SECONDS=str(seconds), SEED=str(seed))
    if hide_gpu:
        env['HIP_VISIBLE_DEVICES'] = ''
        env['CUDA_VISIBLE_DEVICES'] = ''
    p = subprocess.run([sys.executable, os.path.join(HERE, script)], env=env, capture_output=True, text=True,
                       timeout=600)
    tail = (p.stdout + p.stderr)[-1500:]
    assert p.returncode == 0, tail
    assert ' ok' in p.stdout or 'ok:' in p.stdout, tail


@pytest.mark.parametrize('script', ['fuzz_gpu.py', 'fuzz_more.py'])
def test_random_sweep_on_the_emulation(script, emulation):
    _run(script, 8, 101, hide_gpu=True)


@pytest.mark.gpu
@pytest.mark.parametrize('script', ['fuzz_gpu.py', 'fuzz_more.py'])
def test_random_sweep_on_the_gpu(script):
    import torch
    assert torch.cuda.is_available()
    _run(script, 10, 202, hide_gpu=False)
