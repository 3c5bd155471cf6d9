#!/usr/bin/env python3
"""bench.py -- ISP Mpix/s (fwd+bwd) of the fused parametrized pipeline on 512x512 raw batches.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): ParametrizedProcessing, Drone camera
parameters, batch_norm_output=True in train mode, 64 x 512 x 512 synthetic 12-bit RGGB frames PER GPU
(weak scaling: the batch shards over ranks; BatchNorm batch statistics and their backward sums are
exchanged over RCCL, the 132-float ISP gradient is all-reduced).  One step = forward + backward with a
fixed random cotangent, inputs resident in HBM.  Pixels are raw Bayer pixels (B*H*W).

The JSON line also carries
  roofline      the dominant kernel's algorithmic HBM bytes / its average duration (HIP events on the
                launch stream, collected by the library's timing hooks in a second, instrumented pass of
                the same K steps) against the 8 TB/s HBM3E peak;
  cpu_baseline  the numpy oracle (a port of the reference's pipeline_torch.py forward + backward) timed on
                the host on a bounded sample of the same workload (rank 0, N == 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# algorithmic HBM bytes per raw pixel of each kernel (DESIGN.md section "bytes per pixel")
ALGO_BYTES_PER_PX = {
    'r2l_launch_fwd_kernel': 16.0,          # raw 4 in, RGB 12 out (the stats-only pass reads 4, writes 0)
    'r2l_launch_bwd1_kernel': 20.0,       # raw 4 + grad_out 12 in, dL/dY'' 4 out
    'r2l_launch_bwd2_kernel': 8.0,        # raw 4 + dL/dY'' 4 in
    'r2l_launch_bn_reduce_kernel': 24.0,  # grad_out 12 + saved output 12 in
}


def pmc_traffic(kernel, B, S, name='r01_pmc_traffic.json', shape=(64, 512)):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in
    their own runs, gfx950 correction applied; profiles/r01_pmc_traffic*.json), valid for the workload shape the
    passes were collected on only (64x512x512 parametrized, 256x1024x1024 static); None otherwise."""
    path = os.path.join(REPO, 'profiles', name)
    if (B, S) != shape or not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    return t.get(kernel, {}).get('total_bytes')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='frames per GPU (default: 64 parametrized, 256 static)')
    ap.add_argument('--size', type=int, default=None, help='frame height = width (default: 512 parametrized, 1024 static)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--workload', choices=('parametrized', 'static'), default='parametrized',
                    help='parametrized = the headline metric (BASELINE config 2); static = the fused static chain '
                         'demosaic->WB->CCM->clip->gamma on 256x1024x1024 frames per GPU (BASELINE config 3)')
    ap.add_argument('--debayer', choices=('bilinear', 'malvar2004'), default='bilinear')
    ap.add_argument('--raw-u16', action='store_true',
                    help='feed the 12-bit frames as uint16 containers (2 B/px ingest, normalised in-kernel; '
                         'SURVEY.md section 8f) instead of float32: a separate variant, not the headline config')
    return ap.parse_args()


def cpu_baseline(size):
    """numpy oracle (port of pipeline_torch.py fwd + hand-written bwd), float32, one host thread,
    BatchNorm train mode, on 2 frames of the workload's size; repeated until ~10 s have passed."""
    import numpy as np
    from oracle import isp_oracle as orc
    B = 2
    raw = orc.synth_raw(B, size, size, seed=0, kind='uniform')
    cot = np.random.default_rng(1).standard_normal((B, 3, size, size)).astype(np.float32)
    P = orc.IspParams(orc.DRONE_CAMERA_PARAMS)
    t0 = time.perf_counter()
    n = 0
    while True:
        bn = dict(training=True, running_mean=np.zeros(3), running_var=np.ones(3))
        _, _, c = orc.parametrized_forward(raw, P, bn=bn)
        orc.parametrized_backward(P, c, cot)
        n += 1
        dt = time.perf_counter() - t0
        if dt > 12.0:
            break
    return {'value': round(n * B * size * size / dt / 1e6, 3), 'unit': 'Mpix/s', 'cores': 1,
            'kind': 'port',
            'sample': f'{n} x (fwd+bwd of {B}x{size}x{size} frames, BN train), numpy oracle float32, '
                      f'{dt:.1f} s on 1 of {os.cpu_count()} host cores'}


def cpu_baseline_static(debayer):
    """numpy oracle of processing() (pipeline_numpy.py:70-141, float64 like the reference) on BASELINE config 1
    frames (256x256), short chain, one host thread, repeated for ~10 s."""
    import numpy as np
    from oracle import isp_oracle as orc
    raw = orc.synth_raw(16, 256, 256, seed=0, kind='uniform')
    t0 = time.perf_counter()
    n = 0
    while True:
        orc.static_batch(raw, orc.DRONE_CAMERA_PARAMS, debayer, 'none', 'none')
        n += 1
        dt = time.perf_counter() - t0
        if dt > 10.0:
            break
    return {'value': round(n * raw.size / dt / 1e6, 3), 'unit': 'Mpix/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n} x 16x256x256 frames ({debayer}, short chain), numpy/scipy oracle float64, '
                      f'{dt:.1f} s on 1 of {os.cpu_count()} host cores'}


def _init_distributed(torch, dist, world, local_rank):
    """one process per GPU over RCCL (backend "nccl").  R2L_BENCH_BACKEND=gloo lets the N > 1 code path be
    exercised with several processes on ONE GPU (a functional check, not a measurement)."""
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # before anything initialises HIP
    index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(index)
    dev = torch.device('cuda', index)
    if world > 1:
        backend = os.environ.get('R2L_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    return dev


def main_static(args):
    """BASELINE config 3: one step = the fused static chain over 256x1024x1024 frames per GPU (no exchange between
    ranks: static mode needs no collective)."""
    import torch
    import torch.distributed as dist
    from raw2logit_amd import _lib, cameras, functional as F_
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dev = _init_distributed(torch, dist, world, local_rank)
    lib = _lib.device_library()
    B, S = (args.batch or 256), (args.size or 1024)
    gen = torch.Generator(dev).manual_seed(rank)
    u = torch.randint(0, 4096, (B, S, S), device=dev, generator=gen, dtype=torch.int32)
    raw = u.to(torch.uint16) if args.raw_u16 else u.to(torch.float32) / 4095.0

    def step():
        return F_.static_pipeline(raw, cameras.DRONE, args.debayer, 'none', 'none', bits=12)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    px = world * B * S * S
    lib.r2l_timing_enable(1)
    for _ in range(args.steps):
        step()
    barrier()
    buf = ctypes.create_string_buffer(1 << 14)
    lib.r2l_timing_report(buf, len(buf))
    lib.r2l_timing_enable(0)
    name, cnt, ms = buf.value.decode().split()
    avg_us = 1e3 * float(ms) / int(cnt)
    bpp = 14.0 if args.raw_u16 else 16.0
    ach = B * S * S * bpp / (avg_us * 1e-6) / 1e9
    if rank == 0:
        out = {'metric': 'ISP Mpix/s (static fwd: demosaic->WB->CCM->clip->gamma) on 1024x1024 raw batches',
               'value': round(px * args.steps / dt / 1e6, 1), 'unit': 'Mpix/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 4), 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': f'static short chain ({args.debayer}), {B}x{S}x{S} 12-bit RGGB frames per GPU'
                                      + (' as uint16 containers' if args.raw_u16 else '') + ', Drone camera parameters',
                          'global_batch': world * B, 'frame': [S, S],
                          'parallelism': f'batch shard x{world}, no collective' if world > 1 else 'single GPU'},
               'roofline': {'bound': 'hbm', 'kernel': name, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS,
                            'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                            'traffic': None if args.raw_u16 else
                            pmc_traffic(name, B, S, 'r01_pmc_traffic_static.json', (256, 1024)),
                            'avg_us': round(avg_us, 1), 'algo_bytes_per_px': bpp}}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_static(args.debayer)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.workload == 'static':
        return main_static(args)
    import torch
    import torch.distributed as dist
    import numpy as np
    from raw2logit_amd import _lib, cameras         # (the oracle is only touched by the cpu_baseline leg)
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dev = _init_distributed(torch, dist, world, local_rank)
    lib = _lib.device_library()                     # raises if the HIP extension is missing

    B, S = (args.batch or 64), (args.size or 512)
    # SURVEY.md section 8d "perf" distribution: uniform 12-bit codes, raw = u16 / 4095 (float32)
    u16 = np.random.default_rng(rank).integers(0, 4096, (B, S, S)).astype(np.uint16)
    raw = torch.from_numpy(u16 if args.raw_u16 else u16.astype(np.float32) / np.float32(4095)).to(dev)
    cot = torch.randn((B, 3, S, S), device=dev, generator=torch.Generator(dev).manual_seed(1 + rank))
    model = ParametrizedProcessing(cameras.DRONE, track_stages=False, batch_norm_output=True)
    model = model.to(dev).train()
    model.raw_bits = 12
    if world > 1:
        model.process_group = dist.group.WORLD
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        y = model(raw)
        y.backward(cot)
        if world > 1:                                # data-parallel sum of the 132-float ISP gradient
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            dist.all_reduce(flat)
            torch._foreach_copy_([p.grad for p in params],
                                 [c.view_as(p.grad) for c, p in zip(flat.split([p.numel() for p in params]), params)])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    px_per_step = world * B * S * S
    value = px_per_step * args.steps / dt / 1e6

    roofline = None
    kernels = {}
    if not args.no_roofline:
        # second pass of the same K steps with the library's per-kernel HIP-event hooks switched on
        lib.r2l_timing_enable(1)
        for _ in range(args.steps):
            step()
        barrier()
        buf = ctypes.create_string_buffer(1 << 16)
        lib.r2l_timing_report(buf, len(buf))
        lib.r2l_timing_enable(0)
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            kernels[name] = {'launches': int(cnt), 'avg_us': round(1e3 * float(ms) / int(cnt), 2)}
        # 16-bit container variants (r2l_launch_*_u16_kernel): 2 B/px less raw traffic
        algo = dict(ALGO_BYTES_PER_PX)
        for k, v in ALGO_BYTES_PER_PX.items():
            if 'bn_reduce' not in k:
                algo[k.replace('_kernel', '_u16_kernel')] = v - 2.0
        cand = {k: v for k, v in kernels.items() if k in algo}
        if cand:
            total = {k: v['launches'] * v['avg_us'] for k, v in cand.items()}
            dom = max(total, key=total.get)
            avg_us = cand[dom]['avg_us']
            bpp = algo[dom]
            if dom.startswith('r2l_launch_fwd'):
                # two launches per step: stats-only (raw only) and apply (raw + 12 B/px out): average bytes
                bpp = ((bpp - 12.0) + bpp) / 2
            achieved = B * S * S * bpp / (avg_us * 1e-6) / 1e9
            roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': round(achieved, 1),
                        'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                        'traffic': pmc_traffic(dom, B, S) if not args.raw_u16 else None, 'avg_us': avg_us, 'algo_bytes_per_px': bpp}

    if rank == 0:
        out = {
            'metric': 'ISP Mpix/s (fwd+bwd) on 512x512 raw batches', 'value': round(value, 1),
            'unit': 'Mpix/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * dt / args.steps, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic' + (' (uint16 containers)' if args.raw_u16 else ''),
            'config': {'workload': f'parametrized ISP fwd+bwd, BatchNorm train, {B}x{S}x{S} 12-bit RGGB '
                                   f'frames per GPU, Drone camera parameters',
                       'global_batch': world * B, 'frame': [S, S],
                       'parallelism': f'batch shard x{world}' if world > 1 else 'single GPU',
                       'step': 'forward + backward' + (' + 132-float grad all-reduce' if world > 1 else '')},
            'roofline': roofline,
            # the whole step against SURVEY.md section 8d's 52 B/px (single fused backward; this kernel split
            # moves 72 B/px)
            'step_roofline': {'algo_bytes_per_px': 52.0, 'achieved': round(px_per_step / world * 52.0 * args.steps / dt / 1e9, 1),
                              'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                              'frac': round(px_per_step / world * 52.0 * args.steps / dt / 1e9 / HBM_PEAK_GBS, 4)},
            'kernels': kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(S)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
