"""The kernels behind the headline numbers under AddressSanitizer + UBSan on the CPU.

The row-streaming forward, the passes over planes and the branch-free static loops fetch unconditionally from clamped addresses
and talk between lanes (DPP shifts, readfirstlane, shuffles, one barrier per row); the serial emulation cannot run them and the
GPU has no sanitizer on this pool.  tests/emul/r2l_lockstep.cpp compiles EVERY kernel in its device form with one host thread
per lane (tests/emul/r2l_lockstep_rt.h), built with -fsanitize=address,undefined; tests/lockstep_checks.py runs the GPU suite's
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
