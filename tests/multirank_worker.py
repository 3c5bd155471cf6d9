"""One rank of the multi-rank device tests (tests/test_gpu_multirank.py): a FRESH process that joins a gloo group and
takes its contiguous share of a batch through ParametrizedProcessing on the GPU -- the N > 1 code of SURVEY.md
section 8e on real hardware: r2l_isp_step_fwd / _bwd phases A and B, r2l_bn_finalize / r2l_bn_bwd_means with nranks > 1
(rank-ordered sums inside the kernels), the row-streaming forward, the kept-luma backward, the staged kernels.

    python tests/multirank_worker.py <rank> <world> <port> <out_dir>

All ranks share cuda:0 on a one-GPU box (gloo moves the 7 / 6 / 132-float vectors through host memory: see
raw2logit_amd.functional._host_staged); with several GPUs each rank takes its own and the backend is RCCL
(R2L_TEST_BACKEND=nccl).  The cases live in CASES below; the parent test runs the same cases in one process on the
whole batch and compares."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# name, (global batch, H, W), frames, path, camera
CASES = (
    ('w1_f32', (4, 40, 256), 'f32', 'fused', 'drone'),            # streaming forward, 1 wavefront per row
    ('w2_u16', (4, 36, 512), 'u16', 'fused', 'drone'),            # 2 wavefronts per row, 16-bit containers
    ('w4_f32', (2, 20, 1024), 'f32', 'fused', 'microscopy'),      # 4 wavefronts per row
    ('ragged_f32', (4, 30, 70), 'f32', 'fused', 'drone'),         # W % 4 != 0: tile forward + recomputing backward
    ('tiles_bands_f32', (6, 130, 260), 'f32', 'fused', 'drone'),  # several bands / tiles, partially filled last strip
    ('staged_f32', (4, 24, 40), 'f32', 'staged', 'drone'),        # track_stages=True: the stage-by-stage kernels
    ('staged_u16', (2, 16, 24), 'u16', 'staged', 'drone'),
    ('config5_shard', (128, 256, 256), 'f32', 'fused', 'drone'),  # BASELINE config 5's per-GPU share (64 x 256 x 256) x 2
    ('planes_shard', (48, 512, 512), 'f32', 'fused', 'drone'),    # 24 x 512 x 512 per rank (>= 4 Mi px): the backward as passes over planes
)


def case_inputs(name, shape, frames):
    """(raw as the kernels take it, cotangent), both numpy, whole batch"""
    from oracle import isp_oracle as orc
    B, H, W = shape
    seed = sum(map(ord, name))
    u = np.rint(orc.synth_raw(B, H, W, seed=seed % 97, kind='scene' if B < 40 else 'uniform').astype(np.float64)
                * 4095).astype(np.uint16)
    cot = np.random.default_rng(seed).standard_normal((B, 3, H, W)).astype(np.float32)
    raw = u.astype(np.int16) if frames == 'u16' else u.astype(np.float32) / np.float32(4095)
    return raw, cot


def case_params(camera, perturb_seed=5):
    """oracle-side parameters: the camera's, with dense non-default weights (zeros hide nothing)"""
    from oracle import isp_oracle as orc
    P = orc.IspParams(orc.CAMERAS[camera])
    P.perturb(perturb_seed, scale=0.02)
    return P


def make_module(path, camera, dev):
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import parity_checks as pc
    m = pc.make_module(dict(camera=camera, track=(path == 'staged'), bn=True, additive=False, training=True),
                       case_params(camera), dev)
    m.raw_bits = 12
    return m


def run_case(case, dev, lo, hi, group):
    """this process's frames [lo, hi) of the case's batch -> dict of numpy results"""
    import torch
    name, shape, frames, path, camera = case
    raw_np, cot_np = case_inputs(name, shape, frames)
    raw = torch.from_numpy(raw_np[lo:hi]).to(dev)
    cot = torch.from_numpy(cot_np[lo:hi]).to(dev)
    m = make_module(path, camera, dev)
    m.process_group = group
    y = m(raw)
    y.backward(cot)
    from raw2logit_amd import functional as F_
    params = list(m.parameters())
    local = torch.cat([p.grad.reshape(-1) for p in params]).cpu().numpy()
    h = F_.GradAllReduce(params, group)          # data-parallel sum, asynchronous; awaited before anybody reads .grad
    h.wait()
    flat = torch.cat([p.grad.reshape(-1) for p in params]).cpu().numpy()
    # second step on the same module: running statistics after two updates, num_batches_tracked
    for p in params:
        p.grad = None
    y2 = m(raw)
    out = {'y': y.detach().cpu().numpy(), 'g': flat, 'g_local': local,
           'rm': m.batch_norm.running_mean.cpu().numpy(), 'rv': m.batch_norm.running_var.cpu().numpy(),
           'nbt': np.asarray(int(m.batch_norm.num_batches_tracked)), 'y2_equal': np.asarray(bool(torch.equal(y2, y)))}
    if os.environ.get('R2L_TEST_GRAPH') == '1' and path == 'fused':
        out.update(graph_step(case, dev, raw, cot, group, y, flat))
    return out


def graph_step(case, dev, raw, cot, group, y_eager, g_eager):
    """the same step -- both calls split around their all-gathers, the gradient all-reduce behind the backward -- captured
    into ONE HIP graph with its RCCL collectives (raw2logit_amd/graphs.py: StepGraph(process_group=...)): output and
    all-reduced gradient of a replay against the eager step's, the collectives the eager step issued, and both timings"""
    import time
    import torch
    from raw2logit_amd import functional as F_
    from raw2logit_amd.graphs import StepGraph
    name, shape, frames, path, camera = case
    me = make_module(path, camera, dev)
    me.process_group = group
    pe = list(me.parameters())

    def eager():
        for p in pe:
            p.grad = None
        me(raw).backward(cot)
        F_.GradAllReduce(pe, group).wait()
    F_.CommTimer.enable(True)
    eager()
    torch.cuda.synchronize()
    comm = F_.CommTimer.report()
    F_.CommTimer.enable(False)
    mg = make_module(path, camera, dev)
    sg = StepGraph(mg, raw, cot, process_group=group)
    sg.replay()
    for p in mg.parameters():
        p.grad = None                       # (what optimizer.zero_grad() does: the next replay re-attaches the tensors)
    yg = sg.replay()
    gg = torch.cat([p.grad.reshape(-1) for p in mg.parameters()]).cpu().numpy()
    res = {'graph_y_equal': np.asarray(bool(np.array_equal(yg.detach().cpu().numpy(), y_eager.detach().cpu().numpy()))),
           'graph_g_equal': np.asarray(bool(np.array_equal(gg, g_eager))),
           'graph_g_maxdiff': np.asarray(float(np.abs(gg - g_eager).max())),
           'comm_names': np.asarray(sorted(comm)), 'comm_calls': np.asarray([comm[k]['calls'] for k in sorted(comm)]),
           'comm_us': np.asarray([comm[k]['avg_us'] for k in sorted(comm)])}
    import torch.distributed as dist
    for tag, fn in (('eager', eager), ('graph', sg.replay)):
        for _ in range(10):
            fn()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        res[f'ms_{tag}'] = np.asarray(1e3 * (time.perf_counter() - t0) / 50)
    return res


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import torch.distributed as dist
    backend = os.environ.get('R2L_TEST_BACKEND', 'gloo')
    ndev = torch.cuda.device_count()
    assert ndev >= 1, 'needs a GPU'
    index = rank % ndev if backend == 'nccl' else 0
    torch.cuda.set_device(index)
    dev = torch.device('cuda', index)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from raw2logit_amd import _lib
    assert _lib.device_library().is_device
    only = os.environ.get('R2L_MULTIRANK_CASES')
    res = {}
    for case in CASES:
        if only and case[0] not in only.split(','):
            continue
        B = case[1][0]
        lo, hi = rank * B // world, (rank + 1) * B // world
        for k, v in run_case(case, dev, lo, hi, dist.group.WORLD).items():
            res[f'{case[0]}/{k}'] = v
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
