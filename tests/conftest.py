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
    subprocess.run(['g++', '-std=c++17', '-O2', '-DR2L_TEST_HOOKS', '-shared', '-fPIC', EMUL_SRC, '-o', tmp],
                   check=True)
    os.replace(tmp, EMUL_LIB)
    return EMUL_LIB


LOCKSTEP_SRC = os.path.join(REPO, 'tests', 'emul', 'r2l_lockstep.cpp')
LOCKSTEP_LIB = os.path.join(REPO, 'tests', '_build', 'libr2l_lockstep_asan.so')


def build_lockstep():
    """g++ -fsanitize=address,undefined build of the LOCK-STEP emulation (tests/emul/r2l_lockstep.cpp + r2l_lockstep_rt.h: every
    kernel in its device form, one FIBER per lane, cooperatively scheduled on a single host thread: address / UB checks, no model of
    inter-lane races).  Test infrastructure; loaded only by tests/lockstep_checks.py in a
    subprocess that preloads libasan.  -O0: the run time is thread rendezvous, not arithmetic, and the build takes 40 s
    instead of 6 min.  (No -pthread: nothing runs concurrently.)"""
    csrc = os.path.join(REPO, 'raw2logit_amd', 'csrc')
    deps = [LOCKSTEP_SRC, os.path.join(REPO, 'tests', 'emul', 'r2l_lockstep_rt.h'), os.path.join(REPO, 'include', 'r2l_isp.h')] + \
           [os.path.join(csrc, f) for f in os.listdir(csrc)]
    if os.path.exists(LOCKSTEP_LIB) and all(os.path.getmtime(LOCKSTEP_LIB) >= os.path.getmtime(d) for d in deps):
        return LOCKSTEP_LIB
    os.makedirs(os.path.dirname(LOCKSTEP_LIB), exist_ok=True)
    tmp = LOCKSTEP_LIB + f'.{os.getpid()}.tmp'
    subprocess.run(['g++', '-std=c++17', '-O0', '-g', '-fno-omit-frame-pointer', '-fsanitize=address,undefined',
                    '-DR2L_TEST_HOOKS', '-shared', '-fPIC', LOCKSTEP_SRC, '-o', tmp], check=True)
    os.replace(tmp, LOCKSTEP_LIB)
    return LOCKSTEP_LIB


HOST_CHECK_SRC = os.path.join(REPO, 'tests', 'abi_host', 'r2l_host_check.c')
HOST_CHECK = os.path.join(REPO, 'tests', '_build', 'r2l_host_check')


def build_host_check():
    """gcc -std=c11 build of the plain-C consumer of include/r2l_isp.h (tests/abi_host/r2l_host_check.c): the header compiles as C,
    the gfx950 library links from C, and -- on the GPU box -- a host with no Python and no torch drives a training step and a
    static chain (tests/test_gpu_abi_host.py).  The library is found relative to the executable ($ORIGIN): the tree travels."""
    from raw2logit_amd import _lib
    lib = _lib.build_device_library()
    deps = [HOST_CHECK_SRC, os.path.join(REPO, 'include', 'r2l_isp.h'), lib]
    if os.path.exists(HOST_CHECK) and all(os.path.getmtime(HOST_CHECK) >= os.path.getmtime(d) for d in deps):
        return HOST_CHECK
    os.makedirs(os.path.dirname(HOST_CHECK), exist_ok=True)
    tmp = HOST_CHECK + f'.{os.getpid()}.tmp'
    subprocess.run(['gcc', '-std=c11', '-O1', '-Wall', '-Wextra', '-Werror', '-I' + os.path.join(REPO, 'include'),
                    '-isystem', '/opt/rocm/include', '-D__HIP_PLATFORM_AMD__', HOST_CHECK_SRC, '-o', tmp,
                    '-L' + os.path.dirname(lib), '-lr2l_isp', '-L/opt/rocm/lib', '-lamdhip64', '-lm',
                    '-Wl,-rpath,$ORIGIN/../../raw2logit_amd', '-Wl,-rpath,/opt/rocm/lib'], check=True)
    os.replace(tmp, HOST_CHECK)
    return HOST_CHECK


HOOKS_LIB = os.path.join(REPO, 'tests', '_build', 'libr2l_isp_hooks.so')


def build_hooks_library():
    """the gfx950 library once more with -DR2L_TEST_HOOKS: launch shapes can be overridden through R2L_GRID_* /
    R2L_STREAM_BANDS (the shipped libr2l_isp.so ignores them).  Built by __graft_entry__.build() so that it travels
    to the GPU box; tests/parity_checks.py loads it for the grid-independence checks only."""
    from raw2logit_amd import _lib
    return _lib.build_device_library(out_path=HOOKS_LIB, extra=('-DR2L_TEST_HOOKS',))


@pytest.fixture(scope='session')
def emulation():
    """CPU tensors are served by the host emulation of the HIP kernels for the duration of the tests."""
    import emul_hook
    lib = emul_hook.enable(build_emulation())
    yield lib
    emul_hook.enable(None)


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    d = os.path.join(REPO, 'tests', 'golden')
    return {n: np.load(os.path.join(d, n + '.npz'), allow_pickle=False)
            for n in ('param_cases', 'raw2rgb', 'static_cases', 'static_opts', 'harness', 'aux_losses')}


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """achieved error of every parity check next to its limit (tests/parity_checks.py: report())"""
    try:
        import torch
        if not torch.cuda.is_available():
            terminalreporter.write_line(
                'note: this run has no GPU -- the host emulation covers the tile forms of the kernels; the row-streaming '
                'forward, the passes over planes and the static stream / chain kernels only run under `-m gpu`')
    except Exception:
        pass
    pc = sys.modules.get('parity_checks')
    if pc is None:
        return
    log = {}
    for what, err, tol in pc.ERROR_LOG:
        if what not in log or err / max(tol, 1e-300) > log[what][0] / max(log[what][1], 1e-300):
            log[what] = (err, tol)
    if not log:
        return
    tr = terminalreporter
    tr.section('parity: achieved max|err| vs limit')
    for what, (err, tol) in sorted(log.items(), key=lambda kv: -kv[1][0] / max(kv[1][1], 1e-300)):
        tr.write_line(f'{what:72s} {err:10.3e} / {tol:9.3e}  {100.0 * err / tol if tol > 0 else 0.0:6.1f} %')
    out = os.environ.get('R2L_PARITY_LOG')
    if out:
        with open(out, 'w') as f:
            for what, (err, tol) in sorted(log.items()):
                f.write(f'{what}\t{err:.6e}\t{tol:.6e}\n')
