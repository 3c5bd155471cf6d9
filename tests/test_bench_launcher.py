"""bench.py's own multi-rank launcher: `python bench.py --gpus N` with no launcher around it
starts N ranks as child processes, they rendezvous on 127.0.0.1, run the barrier / max-over-ranks timing
protocol and rank 0 prints ONE JSON line with n_gpus == N.  Here: 2 ranks over gloo, kernels served by the host
emulation (R2L_BENCH_DEVICE=emulation) -- a functional check of the launcher, not a measurement.  The nccl twin
runs on the GPU box (tests/test_gpu_parity.py::test_bench_two_ranks_nccl) when it has two GPUs."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_args, extra_env, timeout=600):
    env = dict(os.environ, R2L_BENCH_BACKEND='gloo', R2L_BENCH_DEVICE='emulation')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '2', '--warmup', '1',
                           '--batch', '2', '--size', '32', '--no-roofline', '--no-cpu-baseline'] + extra_args,
                          env=env, capture_output=True, text=True, timeout=timeout, cwd=REPO)


def _json_lines(stdout):
    return [json.loads(line) for line in stdout.splitlines() if line.startswith('{')]


def test_gpus_2_launches_two_ranks(emulation):
    r = _run(['--gpus', '2'], {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                  # rank 0 only
    out = lines[0]
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['warmup'] == 1
    assert out['config']['global_batch'] == 4 and out['scaling'] == 'weak'
    assert out['value'] > 0 and out['ms_per_step'] > 0
    assert 'grad all-reduce' in out['config']['step']


def test_gpus_8_launches_eight_ranks(emulation):
    """the world size the driver's scaling run ends on: `python bench.py --gpus 8` starts eight ranks (gloo, kernels served by the
    host emulation), the barrier / max-over-ranks timing protocol, the split step calls around the two all-gathers and the
    asynchronous gradient all-reduce complete on every rank, rank 0 prints ONE line with n_gpus == 8 and the weak-scaling batch"""
    r = _run(['--gpus', '8'], {}, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out['n_gpus'] == 8 and out['config']['global_batch'] == 16 and out['scaling'] == 'weak'
    assert out['value'] > 0 and out['ms_per_step'] > 0 and 'ms_per_step_graph' not in out      # no graph trial unless asked for
    assert 'batch shard x8' in out['config']['parallelism']


def test_single_rank_line(emulation):
    r = _run(['--gpus', '1'], {})
    assert r.returncode == 0, r.stderr[-2000:]
    out, = _json_lines(r.stdout)
    assert out['n_gpus'] == 1 and out['config']['global_batch'] == 2


def test_world_size_mismatch_is_an_error(emulation):
    r = _run(['--gpus', '2'], {'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and not _json_lines(r.stdout)
