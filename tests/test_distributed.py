"""Batch sharding over ranks on CPU: world_size 2, gloo, kernels served by the
host emulation.  Each rank processes half of the batch; BatchNorm batch statistics (forward) and their
backward sums cross ranks, the 132-float ISP gradient is summed; the result must equal the single-process
result on the whole batch (which is what the single-GPU reference computes)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn_ranks(fn, world, emul_path, out_dir):
    """mp.spawn over a free port; started over ONCE if the rendezvous itself failed (the port probed free was taken before rank 0
    bound it -- seen once in this round's CPU runs while another process was opening sockets -- or a peer could not connect):
    never on an assertion of a worker"""
    for attempt in (0, 1):
        try:
            return mp.spawn(fn, args=(world, _free_port(), emul_path, out_dir), nprocs=world, join=True)
        except Exception as e:                       # noqa: BLE001
            text = str(e)
            rendezvous = any(k in text for k in ('Address already in use', 'EADDRINUSE', 'Connection refused', 'Connection reset',
                                                 'connect() timed out', 'failed to connect', 'The server socket has failed',
                                                 'DistNetworkError', 'DistStoreError', 'Broken pipe'))
            if attempt == 1 or not rendezvous or 'AssertionError' in text:
                raise
            print(f'test_distributed: rendezvous failed ({text[-300:]!r}); starting the {world} ranks over on another port')


def _worker(rank, world, port, emul_path, out_dir):
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import emul_hook
    emul_hook.enable(emul_path)
    from oracle import isp_oracle as orc
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
    B, H, W = 4, 24, 40
    raw = torch.from_numpy(orc.synth_raw(B, H, W, seed=3, kind='scene'))
    cot = torch.from_numpy(np.random.default_rng(7).standard_normal((B, 3, H, W)).astype(np.float32))
    lo, hi = rank * B // world, (rank + 1) * B // world
    m = ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=True).train()
    m.process_group = dist.group.WORLD
    y = m(raw[lo:hi])
    (y * cot[lo:hi]).sum().backward()
    from raw2logit_amd import functional as F_
    h = F_.GradAllReduce(m.parameters(), dist.group.WORLD)     # data-parallel sum of the ISP gradient, asynchronous
    h.wait()
    flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    # the staged kernels (track_stages=True) exchange the same statistics
    mt = ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, track_stages=True, batch_norm_output=True).train()
    mt.process_group = dist.group.WORLD
    yt = mt(raw[lo:hi])
    (yt * cot[lo:hi]).sum().backward()
    flat_t = torch.cat([p.grad.reshape(-1) for p in mt.parameters()])
    dist.all_reduce(flat_t)
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), y=y.detach().numpy(), g=flat.numpy(),
             rm=m.batch_norm.running_mean.numpy(), rv=m.batch_norm.running_var.numpy(),
             yt=yt.detach().numpy(), gt=flat_t.numpy())
    dist.barrier()                 # (a rank that tears its gloo context down while a peer is still inside its last collective
    dist.destroy_process_group()   #  aborts that peer: 'terminate called without an active exception', seen with 8 ranks)


def test_two_rank_shard_equals_single_process(emulation, tmp_path):
    from oracle import isp_oracle as orc
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
    import conftest
    world = 2
    _spawn_ranks(_worker, world, conftest.EMUL_LIB, str(tmp_path))
    B, H, W = 4, 24, 40
    raw = torch.from_numpy(orc.synth_raw(B, H, W, seed=3, kind='scene'))
    cot = torch.from_numpy(np.random.default_rng(7).standard_normal((B, 3, H, W)).astype(np.float32))
    m = ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=True).train()
    y = m(raw)
    (y * cot).sum().backward()
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).numpy()
    r = [np.load(os.path.join(str(tmp_path), f'rank{k}.npz')) for k in range(world)]
    y_sharded = np.concatenate([r[0]['y'], r[1]['y']])
    assert np.abs(y_sharded - y.detach().numpy()).max() < 2e-5
    assert np.array_equal(r[0]['g'], r[1]['g'])
    assert np.abs(r[0]['g'] - g).max() <= 2e-4 * (np.abs(g).max() + 1e-6)
    yt_sharded = np.concatenate([r[0]['yt'], r[1]['yt']])
    assert np.abs(yt_sharded - y.detach().numpy()).max() < 5e-5       # staged == fused == single process
    assert np.abs(r[0]['gt'] - g).max() <= 3e-3 * (np.abs(g).max() + 1e-6)
    for k in range(world):
        np.testing.assert_allclose(r[k]['rm'], m.batch_norm.running_mean.numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(r[k]['rv'], m.batch_norm.running_var.numpy(), rtol=1e-5, atol=1e-7)


def _single_rank_worker(rank, world, port, emul_path, out_dir):
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['R2L_SPLIT_SINGLE_RANK'] = '1'
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import emul_hook
    emul_hook.enable(emul_path)
    from oracle import isp_oracle as orc
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
    from raw2logit_amd import functional as F_
    B, H, W = 3, 24, 40
    raw = torch.from_numpy(orc.synth_raw(B, H, W, seed=3, kind='scene'))
    cot = torch.from_numpy(np.random.default_rng(7).standard_normal((B, 3, H, W)).astype(np.float32))
    m = ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=True).train()
    m.process_group = dist.group.WORLD
    F_.CommTimer.enable(True)
    y = m(raw)
    (y * cot).sum().backward()
    F_.GradAllReduce(m.parameters(), dist.group.WORLD).wait()
    comm = F_.CommTimer.report()
    np.savez(os.path.join(out_dir, 'single.npz'), y=y.detach().numpy(),
             g=torch.cat([p.grad.reshape(-1) for p in m.parameters()]).numpy(), comm=np.asarray(sorted(comm)),
             rm=m.batch_norm.running_mean.numpy(), rv=m.batch_norm.running_var.numpy())
    dist.barrier()                 # (a rank that tears its gloo context down while a peer is still inside its last collective
    dist.destroy_process_group()   #  aborts that peer: 'terminate called without an active exception', seen with 8 ranks)


def test_single_rank_takes_the_split_path_when_asked(emulation, tmp_path):
    """R2L_SPLIT_SINGLE_RANK=1 (functional.split_single_rank): a world of ONE rank runs the N > 1 code -- phase A / B calls
    around real all-gathers of one row, r2l_bn_finalize / r2l_bn_bwd_means as launches, the gradient all-reduce -- which
    is how tests/test_gpu_multirank.py runs (and graph-captures) the RCCL collectives on a one-GPU box."""
    from oracle import isp_oracle as orc
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
    import conftest
    _spawn_ranks(_single_rank_worker, 1, conftest.EMUL_LIB, str(tmp_path))
    r = np.load(os.path.join(str(tmp_path), 'single.npz'))
    assert list(r['comm']) == ['bn statistics all-gather', 'bn-bwd sums all-gather', 'grad all-reduce']
    B, H, W = 3, 24, 40
    raw = torch.from_numpy(orc.synth_raw(B, H, W, seed=3, kind='scene'))
    cot = torch.from_numpy(np.random.default_rng(7).standard_normal((B, 3, H, W)).astype(np.float32))
    m = ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=True).train()
    y = m(raw)
    (y * cot).sum().backward()
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).numpy()
    assert np.abs(r['y'] - y.detach().numpy()).max() < 2e-5
    assert np.abs(r['g'] - g).max() <= 2e-4 * (np.abs(g).max() + 1e-6)
    np.testing.assert_allclose(r['rm'], m.batch_norm.running_mean.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r['rv'], m.batch_norm.running_var.numpy(), rtol=1e-5, atol=1e-7)


# ---- BASELINE config 5's partition: 512 frames -> 8 ranks x 64 frames (train.py:361-368 runs the same batch on ONE GPU: the
# single-process batch statistics are the yardstick).  H, W scaled down (8 x 12 instead of 256 x 256); the batch and the rank
# count are config 5's own.
C5_WORLD, C5_BATCH, C5_HW = 8, 512, (8, 12)


def _c5_inputs():
    from oracle import isp_oracle as orc
    B, (H, W) = C5_BATCH, C5_HW
    raw = orc.synth_raw(B, H, W, seed=11, kind='uniform')
    cot = np.random.default_rng(12).standard_normal((B, 3, H, W)).astype(np.float32)
    return torch.from_numpy(raw), torch.from_numpy(cot)


def _c5_worker(rank, world, port, emul_path, out_dir):
    sys.path.insert(0, REPO)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import emul_hook
    emul_hook.enable(emul_path)
    from oracle import isp_oracle as orc
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
    from raw2logit_amd import functional as F_
    raw, cot = _c5_inputs()
    per = C5_BATCH // world
    lo, hi = rank * per, (rank + 1) * per
    m = ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=True).train()
    m.process_group = dist.group.WORLD
    out = {}
    for step in range(2):                      # two steps: running statistics after two updates, num_batches_tracked
        for p in m.parameters():
            p.grad = None
        y = m(raw[lo:hi])
        y.backward(cot[lo:hi])
        F_.GradAllReduce(list(m.parameters()), dist.group.WORLD).wait()
        out[f'y{step}'] = y.detach().numpy()
        out[f'g{step}'] = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).numpy()
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), rm=m.batch_norm.running_mean.numpy(),
             rv=m.batch_norm.running_var.numpy(), nbt=m.batch_norm.num_batches_tracked.numpy(), **out)
    dist.barrier()                 # (a rank that tears its gloo context down while a peer is still inside its last collective
    dist.destroy_process_group()   #  aborts that peer: 'terminate called without an active exception', seen with 8 ranks)


def test_config5_partition_eight_ranks(emulation, tmp_path):
    """BASELINE config 5 as it is sharded: 512 frames over EIGHT ranks, 64 each (gloo, kernels served by the host emulation).
    Every rank must hold bit-identical BatchNorm running statistics and bit-identical summed gradients (the kernels add the
    gathered rows in rank order), and the sharded run must equal ONE process on the whole batch -- what the reference's single
    GPU computes (train.py:361-368)."""
    from oracle import isp_oracle as orc
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
    import conftest
    world = C5_WORLD
    _spawn_ranks(_c5_worker, world, conftest.EMUL_LIB, str(tmp_path))
    r = [np.load(os.path.join(str(tmp_path), f'rank{k}.npz')) for k in range(world)]
    raw, cot = _c5_inputs()
    m = ParametrizedProcessing(orc.DRONE_CAMERA_PARAMS, batch_norm_output=True).train()
    for step in range(2):
        for p in m.parameters():
            p.grad = None
        y = m(raw)
        y.backward(cot)
        g = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).numpy()
        for k in range(1, world):
            assert np.array_equal(r[0][f'g{step}'], r[k][f'g{step}']), (step, k)      # same sum on every rank, bit for bit
        ys = np.concatenate([r[k][f'y{step}'] for k in range(world)])
        assert ys.shape == (C5_BATCH, 3) + C5_HW
        assert np.abs(ys - y.detach().numpy()).max() < 2e-5, step
        assert np.abs(r[0][f'g{step}'] - g).max() <= 2e-4 * (np.abs(g).max() + 1e-6), step
    for k in range(world):
        assert np.array_equal(r[k]['rm'], r[0]['rm']) and np.array_equal(r[k]['rv'], r[0]['rv'])      # statistics: bit-identical
        assert int(r[k]['nbt']) == 2
    np.testing.assert_allclose(r[0]['rm'], m.batch_norm.running_mean.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(r[0]['rv'], m.batch_norm.running_var.numpy(), rtol=1e-5, atol=1e-7)
