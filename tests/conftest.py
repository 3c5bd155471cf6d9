import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

EMUL_SRC = os.path.join(REPO, 'tests', 'emul', 'r2l_emul.cpp')
EMUL_LIB = os.path.join(REPO, 'tests', '_build', 'libr2l_emul.so')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def build_emulation():
    """g++ build of the HOST EMULATION of the kernel source (test infrastructure, see r2l_emul.cpp)."""
    csrc = os.path.join(REPO, 'raw2logit_amd', 'csrc')
    deps = [EMUL_SRC, os.path.join(REPO, 'include', 'r2l_isp.h')] + \
           [os.path.join(csrc, f) for f in os.listdir(csrc)]
    if os.path.exists(EMUL_LIB) and all(os.path.getmtime(EMUL_LIB) >= os.path.getmtime(d) for d in deps):
        return EMUL_LIB
    os.makedirs(os.path.dirname(EMUL_LIB), exist_ok=True)
    tmp = EMUL_LIB + f'.{os.getpid()}.tmp'
    subprocess.run(['g++', '-std=c++17', '-O2', '-shared', '-fPIC', EMUL_SRC, '-o', tmp], check=True)
    os.replace(tmp, EMUL_LIB)
    return EMUL_LIB


@pytest.fixture(scope='session')
def emulation():
    """CPU tensors are served by the host emulation of the HIP kernels for the duration of the tests."""
    from raw2logit_amd import _lib
    lib = _lib.enable_test_emulation(build_emulation())
    yield lib
    _lib.enable_test_emulation(None)


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    d = os.path.join(REPO, 'tests', 'golden')
    return {n: np.load(os.path.join(d, n + '.npz'), allow_pickle=False)
            for n in ('param_cases', 'raw2rgb', 'static_cases', 'harness', 'aux_losses')}
