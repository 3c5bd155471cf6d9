#!/usr/bin/env python3
"""bench.py -- ISP Mpix/s (fwd+bwd) of the fused parametrized pipeline on 512x512 raw batches.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU over RCCL.  Under torch.distributed.run the ranks are already there (WORLD_SIZE
    set); started plainly, `python bench.py --gpus N` launches the N ranks itself -- as child processes of a
    parent that never touches the GPU -- and relays rank 0's JSON line.)

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): ParametrizedProcessing, Drone camera
parameters, batch_norm_output=True in train mode, 64 x 512 x 512 synthetic 12-bit RGGB frames PER GPU
(weak scaling: the batch shards over ranks; BatchNorm batch statistics and their backward sums are
exchanged over RCCL, the 132-float ISP gradient is all-reduced).  One step = forward + backward with a
fixed random cotangent, inputs resident in HBM.  Pixels are raw Bayer pixels (B*H*W).

The JSON line also carries
  roofline      the dominant kernel's algorithmic HBM bytes / its average duration (HIP events on the
                launch stream, collected by the library's timing hooks in a second, instrumented pass of
                the same K steps) against the 8 TB/s HBM3E peak;
  static_c3     BASELINE config 3 (the north star's >= 70 % target): the fused static chain demosaic -> WB -> CCM
                -> clip -> gamma on 256x1024x1024 and 1024x512x512 frames, per-launch HIP-event time and fraction
                of the HBM peak at 16 algorithmic B/px; plus the train.py default static chain (N == 1 only);
  cpu_baseline  SURVEY.md section 8d: the numpy restatement of the reference's pipeline_numpy.processing() (default
                chain of train.py:96-101) on BASELINE config 1 (16 x 256 x 256 frames) with 1 process and with a
                16-process pool (train.py:318 num_workers=16), and the numpy port of the parametrized fwd+bwd on
                one core; timed on the host BEFORE the process touches the GPU (rank 0, N == 1 only).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# several ranks: capture + agreement + timing of the graph trial, or the eager line goes out (the environment variable: tests)
WATCHDOG_EXIT = 17             # exit code of a process whose graph-trial watchdog fired (never 0 on a hang)
GRAPH_TRIAL_TIMEOUT_S = float(os.environ.get('R2L_BENCH_TRIAL_TIMEOUT_S', '180'))
PREROLL_S = float(os.environ.get('R2L_BENCH_PREROLL_S', '0.3'))   # untimed pre-roll of the step before the W warm-up steps
                               # (GPU clocks, see main(); counter-collection runs of the profiling scripts set 0)
# Bytes per raw pixel of each kernel family, by kernel-name prefix: (prefix, DESIGN bytes, SURVEY.md section 8d pass,
# section 8d ALGORITHMIC bytes).  `roofline.achieved` / `frac` use the ALGORITHMIC figure of the section-8d pass the kernel
# belongs to (what an ideal split would move: stats 4; forward main 4 + 12; reduction pass 12 + 4; backward main pass 12 + 4);
# the DESIGN figure -- what this kernel moves by construction, kept planes included -- goes to
# `design_bytes_per_px` / `design_frac` beside it.  The `_u16` instantiations (16-bit containers) read 2 B/px less raw.
# The luma path's adjoint passes (bwd1_blur*, bwd2_*) exist only because this design splits the backward main pass: section
# 8d gives them no bytes of their own (None) -- they can never be the roofline's kernel, their time is in `step_roofline`.
ALGO_BYTES_PER_PX = (
    ('r2l_launch_fwd_apply', 20.0, 'forward main pass', 16.0),   # apply pass on the kept luma plane: raw 4 + Y' 4 in, RGB 12 out
    ('r2l_launch_fwd', 16.0, 'forward main pass', 16.0),         # raw 4 in, RGB 12 out; the stats-only pass: see pass_bytes()
    ('r2l_launch_bwd1_plane', 24.0, 'backward main pass', 16.0),  # raw 4 + Y' 4 + grad_out 12 in, dL/dY'' 4 out
    ('r2l_launch_bwd_luma', 16.0, 'backward main pass (luma adjoints)', None),  # dL/dY'' 4 + Y' 4 + raw 4 in (+ HP through LDS)
    ('r2l_launch_bwd1_blur_hp', 12.0, 'backward main pass (luma adjoints)', None),  # dL/dY'' 4 + Y' 4 in, the blur's adjoint 4 out
    ('r2l_launch_bwd1_blur', 8.0, 'backward main pass (luma adjoints)', None),      # blur-weight sums: dL/dY'' 4 + Y' 4 in
    ('r2l_launch_bwd2_hp', 8.0, 'backward main pass (luma adjoints)', None),        # dL/dY'' 4 in, the blur's adjoint 4 out ...
    ('r2l_launch_bwd2_sums', 8.0, 'backward main pass (luma adjoints)', None),      # ... that plane 4 + raw 4 in
    ('r2l_launch_bwd1', 20.0, 'backward main pass', 16.0),       # tile kernels: raw 4 + grad_out 12 in, dL/dY'' 4 out
    ('r2l_launch_bwd2', 8.0, 'backward main pass (luma adjoints)', None),           # raw 4 + dL/dY'' 4 in
    ('r2l_launch_bn_reduce', 24.0, 'reduction pass', 16.0),      # grad_out 12 + saved output 12 in
    ('r2l_launch_bnr_planes', 20.0, 'reduction pass', 16.0),     # the same sums with xhat recomputed: grad_out 12 + raw 4 + Y' 4 in
)


def _family(kernel):
    for fam in ALGO_BYTES_PER_PX:
        if kernel.startswith(fam[0]):
            return fam
    return None


def _u16_less(kernel):
    return 2.0 if '_u16' in kernel and 'bn_reduce' not in kernel else 0.0


def design_bytes_per_px(kernel):
    """bytes per raw pixel this kernel moves by its own design (kept planes included), or None for kernels outside the table"""
    fam = _family(kernel)
    return None if fam is None else fam[1] - _u16_less(kernel)


def algo_bytes_per_px(kernel):
    """SURVEY.md section 8d's algorithmic bytes per raw pixel of the pass `kernel` belongs to (None: the pass has none of its own)"""
    fam = _family(kernel)
    return None if fam is None or fam[3] is None else fam[3] - _u16_less(kernel)


def pass_bytes(kernel, launches_per_step, kernels):
    """(section-8d pass name, algorithmic B/px, design B/px) of `kernel` as it runs in THIS step.  The streaming forward kernel
    serves three roles: the train-mode statistics pass (section 8d: 4 B/px -- raw in; by design it also writes the kept luma
    plane, 8), both passes of a step in one kernel name (average), or the single forward pass (16)."""
    fam = _family(kernel)
    algo, design = algo_bytes_per_px(kernel), design_bytes_per_px(kernel)
    name = fam[2]
    if kernel.startswith('r2l_launch_fwd') and not kernel.startswith('r2l_launch_fwd_apply'):
        if launches_per_step == 2:
            # two launches per step: stats-only (raw only) and apply (raw + 12 B/px out): average bytes
            algo, design, name = (4.0 - _u16_less(kernel) + algo) / 2, ((design - 12.0) + design) / 2, 'statistics + forward main pass'
        elif any(k.startswith('r2l_launch_fwd_apply') for k in kernels):
            algo, design, name = 4.0 - _u16_less(kernel), design - 12.0 + 4.0, 'statistics pass'
    return name, algo, design


PMC_PARAM = 'r06_pmc_traffic.json'
PMC_STATIC = 'r06_pmc_traffic_static.json'      # the three static kernels of static_c3


def pmc_traffic(kernel, B, S, name=PMC_PARAM, shape=(64, 512)):
    """(HBM bytes per launch of `kernel`, where the number comes from).  PMC counters cannot be read inside a
    timed run, so this is NOT measured here: it is taken from the committed rocprofv3 --pmc passes (FETCH_SIZE /
    WRITE_SIZE in their own runs, gfx950 correction applied; profiles/<name>) and only for the workload shape
    those passes were collected on (64x512x512 parametrized, 256x1024x1024 static); (None, reason) otherwise.
    The file says which build it was collected on (`_meta.library_digest`, written by the collecting script from
    raw2logit_amd._lib.source_digest()); the source string reports that tag next to the digest of the sources this
    process runs -- it is read, never asserted."""
    path = os.path.join(REPO, 'profiles', name)
    if (B, S) != shape:
        return None, f'no PMC profile for this shape (profiles/{name} covers {shape[0]}x{shape[1]}x{shape[1]})'
    if not os.path.exists(path):
        return None, f'profiles/{name} not found'
    with open(path) as f:
        t = json.load(f)
    v = t.get(kernel, {}).get('total_bytes')
    if v is None:
        return None, f'profiles/{name} has no entry for {kernel}'
    meta = t.get('_meta') or {}
    tag = meta.get('library_digest')
    try:
        from raw2logit_amd import _lib
        now = _lib.source_digest()
    except Exception:                                # noqa: BLE001
        now = None
    if tag is None:
        build = 'the file carries no build tag'
    elif now is None:
        build = f'collected on library sources {tag}'
    elif tag == now:
        build = f'collected on library sources {tag} = the sources of this run'
    else:
        build = f'collected on library sources {tag}; THIS run is built from {now} (kernels changed since: treat as indicative)'
    return v, (f'committed rocprofv3 --pmc profile profiles/{name} ({meta.get("round", "round unknown")}, '
               f'{meta.get("collected", "date unknown")}; {build}; same shape; not measured in this run)')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='frames per GPU (default: 64 parametrized, 256 static)')
    ap.add_argument('--size', type=int, default=None, help='frame height = width (default: 512 parametrized, 1024 static)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--workload', choices=('parametrized', 'static', 'e2e-microscopy', 'e2e-drone'),
                    default='parametrized',
                    help='parametrized = the headline metric (BASELINE config 2); static = the fused static chain '
                         'demosaic->WB->CCM->clip->gamma on 256x1024x1024 frames per GPU (BASELINE config 3); '
                         'e2e-microscopy = BASELINE config 4: ISP (Microscopy parameters) + ResNet-18, CE loss, Adam '
                         'step, 128x256x256 per GPU; e2e-drone = config 5: ISP (Drone parameters) + U-Net, Dice loss, '
                         '64x256x256 per GPU (512 over 8 GPUs)')
    ap.add_argument('--debayer', choices=('bilinear', 'malvar2004', 'menon2007'), default='bilinear')
    ap.add_argument('--sharpening', default='none', help="static workload: 'none' | 'sharpening_filter' | 'unsharp_masking'")
    ap.add_argument('--denoising', default='none', help="static workload: 'none' | 'gaussian_denoising' | 'median_denoising'")
    ap.add_argument('--normalize', action='store_true',
                    help='static workload: with the T.Normalize(mean, std) epilogue of train.py:157-171')
    ap.add_argument('--no-static-c3', action='store_true',
                    help='skip the static_c3 sub-records (BASELINE config 3) appended to the headline line')
    ap.add_argument('--quick', action='store_true',
                    help='the headline workload only: no cpu_baseline leg, no static_c3 / small_shapes sub-records (A/B runs)')
    ap.add_argument('--graph', action='store_true',
                    help='replay the whole step as ONE HIP graph (raw2logit_amd/graphs.py: StepGraph): the step costs the '
                         'host one graph launch instead of two C-ABI calls + autograd; matters below ~8 Mpix per step, '
                         'where the host is the bound (single GPU)')
    ap.add_argument('--graph-also', action='store_true', help='(accepted for compatibility; see --graph-trial)')
    ap.add_argument('--no-graph-trial', action='store_true', help='(accepted for compatibility: the trial is opt-in since round 6)')
    ap.add_argument('--graph-trial', action='store_true',
                    help='several ranks over RCCL: after the eager measurement, capture the same data-parallel step as ONE HIP '
                         'graph with its collectives (raw2logit_amd/graphs.py) and time it the same way; reported BESIDE the eager '
                         "line's value.  OPT-IN: the capture of RCCL collectives has only ever run against a one-rank group on a "
                         'one-GPU box, so the default N > 1 run does nothing that has not run before.  The ranks agree on whether '
                         'every capture succeeded; a watchdog prints the eager line tagged `watchdog_fired` and ends the process '
                         f'with exit code {WATCHDOG_EXIT} (every rank) if the trial does not finish in time')
    ap.add_argument('--no-small-shapes', action='store_true',
                    help="skip the small_shapes sub-records (the datasets' 256x256 tiles: BASELINE configs 4 / 5 per GPU)")
    ap.add_argument('--cold', action='store_true',
                    help='with --quick: still take the `cold` sub-record (the step with an untimed cache scrub between forward and '
                         'backward); the default run always takes it')
    ap.add_argument('--raw-u16', action='store_true',
                    help='feed the 12-bit frames as uint16 containers (2 B/px ingest, normalised in-kernel; '
                         'SURVEY.md section 8f) instead of float32: a separate variant, not the headline config')
    args = ap.parse_args()
    if args.quick:
        args.no_cpu_baseline = args.no_static_c3 = args.no_small_shapes = True
    return args


def cpu_baseline_parametrized(size, seconds=8.0):
    """numpy oracle (port of pipeline_torch.py fwd + hand-written bwd), float32, one host thread,
    BatchNorm train mode, on 2 frames of the workload's size; repeated for ~8 s."""
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
        if dt > seconds:
            break
    return {'value': round(n * B * size * size / dt / 1e6, 3), 'unit': 'Mpix/s', 'cores': 1,
            'kind': 'port',
            'sample': f'{n} x (fwd+bwd of {B}x{size}x{size} frames, BN train), numpy oracle float32, '
                      f'{dt:.1f} s on 1 of {os.cpu_count()} host cores'}


C1_SHAPE = (16, 256, 256)      # BASELINE config 1


def _pool_worker_init():
    """one compute thread per worker process (BLAS pools of 16 workers would oversubscribe the host)"""
    try:
        import threadpoolctl
        _pool_worker_init.limit = threadpoolctl.threadpool_limits(1)
    except Exception:
        pass


def _static_frame(args):
    """one DataLoader-worker unit of the reference: RawProcessingPipeline.__call__ on one float32 frame"""
    from oracle import isp_oracle as orc
    img, chain = args
    return orc.static_batch(img[None], orc.DRONE_CAMERA_PARAMS, *chain)[0].shape


def cpu_baseline_static(chain=('bilinear', 'sharpening_filter', 'gaussian_denoising'), seconds=8.0, workers=16):
    """the numpy/scipy restatement of the reference's processing() (pipeline_numpy.py:70-141)
    on BASELINE config 1 -- 16 float32 frames of 256x256, the default static chain of train.py:96-101 -- timed
    (i) in this process and (ii) over a pool of `workers` processes that each take whole frames, the way the
    reference's DataLoader runs it (train.py:318 num_workers=16).  Must run before this process initialises the
    GPU (the pool forks)."""
    import multiprocessing as mp
    from oracle import isp_oracle as orc
    raw = orc.synth_raw(*C1_SHAPE, seed=0, kind='uniform')
    px = raw.size
    orc.static_batch(raw[:1], orc.DRONE_CAMERA_PARAMS, *chain)          # imports scipy
    t0 = time.perf_counter()
    n = 0
    while True:
        orc.static_batch(raw, orc.DRONE_CAMERA_PARAMS, *chain)
        n += 1
        dt1 = time.perf_counter() - t0
        if dt1 > seconds:
            break
    one = {'value': round(n * px / dt1 / 1e6, 3), 'unit': 'Mpix/s', 'cores': 1,
           'sample': f'{n} x config C1 (16x256x256 float32 frames), {dt1:.1f} s, 1 process'}
    reps = max(2, int(n * min(workers, os.cpu_count() or 1) * 0.75))      # ~seconds of work for the pool
    items = [(raw[i % 16], chain) for i in range(16 * reps)]
    with mp.get_context('fork').Pool(workers, initializer=_pool_worker_init) as pool:
        pool.map(_static_frame, items[:workers])                        # start-up outside the timed region
        t0 = time.perf_counter()
        pool.map(_static_frame, items, chunksize=1)
        dtp = time.perf_counter() - t0
    return {'value': round(reps * px / dtp / 1e6, 3), 'unit': 'Mpix/s', 'cores': workers, 'kind': 'port',
            'sample': f'{reps} x config C1 (16x256x256 float32 frames; {"+".join(chain)}) through a {workers}-process '
                      f'pool, one frame per task (train.py:318 num_workers=16), {dtp:.1f} s; numpy/scipy restatement '
                      f'of pipeline_numpy.processing(); os.cpu_count() = {os.cpu_count()}',
            'one_process': one}


def cpu_baselines(size, parametrized=True):
    out = cpu_baseline_static()
    if parametrized:
        out['parametrized_fwd_bwd_one_core'] = cpu_baseline_parametrized(size)
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process, which has
    not touched (and never touches) the GPU.  Rank 0 prints the JSON line; the exit code is non-zero if any
    rank fails or fewer than N join."""
    # (no device count here: torch.cuda.device_count() may fall back to hipGetDeviceCount, which initialises HIP in
    # this parent; a rank that finds no GPU for its LOCAL_RANK fails fast instead and the exit code says so)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def _world(args):
    """(world, rank, local_rank) from the launcher's environment, checked against --gpus"""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU '
                         f'(torch.distributed.run --nproc-per-node {args.gpus}) or none at all')
    return world, int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))


def _init_distributed(torch, dist, world, local_rank):
    """one process per GPU over RCCL (backend "nccl").  R2L_BENCH_BACKEND=gloo lets the N > 1 code path be
    exercised with several processes on ONE GPU (a functional check, not a measurement);
    R2L_BENCH_DEVICE=emulation (tests/test_bench_launcher.py only) serves CPU tensors with the host emulation of
    the kernels, so that the launcher / rendezvous / timing protocol can be tested without a GPU."""
    if os.environ.get('R2L_BENCH_DEVICE') == 'emulation':
        sys.path.insert(0, os.path.join(REPO, 'tests'))
        import conftest
        import emul_hook
        emul_hook.enable(conftest.build_emulation())
        dev = torch.device('cpu')
        if world > 1:
            dist.init_process_group('gloo')
        return dev
    one_rank = world == 1 and os.environ.get('R2L_BENCH_ONE_RANK_DIST') == '1'   # (tests: the N > 1 code on a 1-GPU box)
    if world > 1 or one_rank:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # before anything initialises HIP
    ndev = torch.cuda.device_count()
    backend = os.environ.get('R2L_BENCH_BACKEND', 'nccl')
    if ndev < 1 or (backend == 'nccl' and world > ndev):
        raise SystemExit(f'bench.py: rank {local_rank} of {world}: {ndev} GPU(s) visible -- one GPU per rank is needed '
                         f'(R2L_BENCH_BACKEND=gloo lets several ranks share a GPU: a functional check, not a measurement)')
    index = local_rank % ndev
    torch.cuda.set_device(index)
    dev = torch.device('cuda', index)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    if one_rank:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ['R2L_SPLIT_SINGLE_RANK'] = '1'        # raw2logit_amd/functional.py: the exchange path with one rank
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    if world > 1 and dist.get_world_size() != world:
        raise SystemExit(f'bench.py: {dist.get_world_size()} ranks joined, expected {world}')
    return dev


class Clock:
    """the contract's timing: barrier + synchronize on both sides of exactly K steps, MAX over ranks"""

    def __init__(self, torch, dist, world, dev):
        # (world: > 1 whenever a process group exists -- also the one-rank group of R2L_BENCH_ONE_RANK_DIST)
        self.torch, self.dist, self.world, self.dev = torch, dist, world, dev

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        if self.dev.type == 'cuda':
            self.torch.cuda.synchronize()

    def preroll(self, step, finish, seconds):
        """run `step` untimed for ~`seconds` of wall time (same count on every rank: rank 0 decides); returns the count"""
        if seconds <= 0:
            return 0
        n, t0 = 0, time.perf_counter()
        while True:
            for _ in range(10):
                step()
            n += 10
            if finish:
                finish()
            go = time.perf_counter() - t0 < seconds
            if self.world > 1:
                t = self.torch.tensor([1.0 if go else 0.0], device=self.dev)
                self.dist.broadcast(t, 0)
                go = bool(t.item() > 0)
            elif self.dev.type == 'cuda':
                self.torch.cuda.synchronize()
            if not go:
                return n

    def time_steps(self, step, steps, warmup, finish=None):
        """`finish` completes whatever the last step left in flight (the asynchronous gradient all-reduce): it runs
        inside the timed region"""
        for _ in range(warmup):
            step()
        if finish:
            finish()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if finish:
            finish()
        self.barrier()
        dt = time.perf_counter() - t0
        if self.world > 1:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt


def kernel_times(lib, clock, step, steps, finish=None):
    """{kernel: {launches, avg_us}} of `steps` more steps, from the library's HIP events around every launch
    (recorded on the stream the kernels are launched on)"""
    lib.r2l_timing_enable(1)
    clock.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if finish:
        finish()
    clock.barrier()
    kernel_times.wall_ms_per_step = 1e3 * (time.perf_counter() - t0) / max(steps, 1)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.r2l_timing_report(buf, len(buf))
    lib.r2l_timing_enable(0)
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = {'launches': int(cnt), 'avg_us': round(1e3 * float(ms) / int(cnt), 2)}
    return out


def static_records(torch, lib, clock, dev):
    """BASELINE config 3 next to the headline number: per-launch duration of the fused static kernels from HIP
    events, as a fraction of the HBM peak at 16 algorithmic B/px (4 in + 12 out).  A process's first launches of
    these kernels run ~12 % slower, hence the 12 warm-up launches."""
    from raw2logit_amd import cameras, functional as F_
    recs = []
    chains = [('short chain (demosaic->WB->CCM->clip->gamma), bilinear', 256, 1024, ('bilinear', 'none', 'none')),
              ('short chain (demosaic->WB->CCM->clip->gamma), bilinear', 1024, 512, ('bilinear', 'none', 'none')),
              ('short chain (demosaic->WB->CCM->clip->gamma), Malvar2004', 256, 1024, ('malvar2004', 'none', 'none')),
              ('train.py default chain (bilinear + sharpening_filter + gaussian_denoising)', 256, 1024,
               ('bilinear', 'sharpening_filter', 'gaussian_denoising'))]
    for what, B, S, chain in chains:
        u = torch.randint(0, 4096, (B, S, S), device=dev, dtype=torch.int32,
                          generator=torch.Generator(dev).manual_seed(0))
        raw = u.to(torch.float32) / 4095.0
        del u

        def step():
            return F_.static_pipeline(raw, cameras.DRONE, *chain)
        for _ in range(12):
            step()
        clock.barrier()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        clock.barrier()
        wall_ms = 1e3 * (time.perf_counter() - t0) / 20
        k = kernel_times(lib, clock, step, 20)
        name = max(k, key=lambda n: k[n]['launches'] * k[n]['avg_us'])
        avg_us = sum(v['launches'] * v['avg_us'] for v in k.values()) / 20      # all launches of one call
        ach = B * S * S * 16.0 / (avg_us * 1e-6) / 1e9
        traffic, source = pmc_traffic(name, B, S, PMC_STATIC, (256, 1024))
        recs.append({'chain': what, 'shape': [B, S, S], 'kernel': name, 'avg_us': round(avg_us, 1), 'traffic': traffic,
                     'traffic_source': source,
                     'ms_per_call_wall': round(wall_ms, 4), 'algo_bytes_per_px': 16.0,
                     'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(ach / HBM_PEAK_GBS, 4),
                     'Mpix_per_s': round(B * S * S / (wall_ms * 1e-3) / 1e6, 1)})
        del raw
        torch.cuda.empty_cache()
    return recs


def small_shape_records(torch, lib, clock, dev, cameras, ParametrizedProcessing):
    """the step at the reference's real tile size (dataset.py:92: 256 x 256): BASELINE config 5's share per GPU
    (64 frames) and config 4's ISP share (128 frames) -- eager (two C-ABI calls through autograd) and the whole step
    replayed as one HIP graph, plus the kernels' own time from the HIP-event pass."""
    from raw2logit_amd.graphs import StepGraph
    recs = []
    for B, S in ((64, 256), (128, 256)):
        gen = torch.Generator(dev).manual_seed(0)
        raw = torch.randint(0, 4096, (B, S, S), device=dev, generator=gen, dtype=torch.int32).to(torch.float32) / 4095.0
        cot = torch.randn((B, 3, S, S), device=dev, generator=gen)
        rec = {'shape': [B, S, S]}
        for mode in ('eager', 'graph'):
            model = ParametrizedProcessing(cameras.DRONE, track_stages=False, batch_norm_output=True).to(dev).train()
            params = list(model.parameters())
            if mode == 'graph':                      # the whole step as ONE HIP graph (raw2logit_amd/graphs.py)
                try:
                    step = StepGraph(model, raw, cot).replay
                except Exception as e:               # noqa: BLE001  (capture refused on this stack: report, go on)
                    rec['graph_error'] = '%s: %s' % (type(e).__name__, e)
                    torch.cuda.synchronize()
                    continue
            else:
                def step():
                    for p in params:
                        p.grad = None
                    model(raw).backward(cot)
            clock.preroll(step, None, 0.05)
            dt = clock.time_steps(step, 50, 10)
            rec['ms_per_step' + ('_graph' if mode == 'graph' else '')] = round(1e3 * dt / 50, 4)
            if mode == 'eager':
                k = kernel_times(lib, clock, step, 50)
                rec['kernels_us_per_step'] = round(sum(v['launches'] * v['avg_us'] for v in k.values()) / 50, 1)
                rec['kernels'] = {n.replace('r2l_launch_', '').replace('_kernel', ''): v['avg_us'] for n, v in k.items()}
        if 'ms_per_step_graph' in rec:
            rec['Mpix_per_s_graph'] = round(B * S * S / (rec['ms_per_step_graph'] * 1e-3) / 1e6, 1)
        recs.append(rec)
        del raw, cot
    return recs


def cold_record(torch, lib, clock, dev, model, raw, cot, steps):
    """the headline step under TRAINING conditions: a classifier runs between the processor's forward and its
    backward, so nothing the forward left in the 256 MiB memory-side cache survives.  Here an untimed scrub (a 768 MiB buffer read
    and rewritten) runs between forward and backward and again before the next forward; the kernels' own time comes from the
    library's HIP-event hooks (the scrub is a torch kernel: not counted).  Reported next to the back-to-back number, which is the
    cache's best case."""
    scrub = torch.empty(768 << 18, dtype=torch.float32, device=dev)      # 768 MiB
    scrub.zero_()
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        y = model(raw)
        scrub.add_(1.0)
        y.backward(cot)
        scrub.add_(1.0)
    for _ in range(5):
        step()
    k = kernel_times(lib, clock, step, steps)
    us = sum(v['launches'] * v['avg_us'] for v in k.values()) / max(steps, 1)
    px = raw.shape[0] * raw.shape[1] * raw.shape[2]
    ach = px * 52.0 / (us * 1e-6) / 1e9
    del scrub
    torch.cuda.empty_cache()
    return {'what': 'the same step with an untimed 768 MiB scrub (read + rewrite) between forward and backward and before the '
                    'next forward: nothing survives in the 256 MiB memory-side cache across the scrubs (the regime of a training '
                    'step, whose task model runs there); kernel times from the library HIP-event hooks, scrub excluded',
            'ms_per_step_kernels': round(us * 1e-3, 4),
            'kernels': {n.replace('r2l_launch_', '').replace('_kernel', ''): v['avg_us'] for n, v in k.items()},
            'step_roofline': {'algo_bytes_per_px': 52.0, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                              'frac': round(ach / HBM_PEAK_GBS, 4)}}


def fallback_records(torch, lib, clock, dev, cameras, ParametrizedProcessing):
    """the frames the row-streaming / plane kernels do not take: the learned additive layer of
    `--adv_noise_layer` (train.py:108,257; pipeline_torch.py:129-131: fixed at 256x256) at BASELINE config 5's per-GPU shard, and
    a frame width with W % 4 != 0.  fwd+bwd, BatchNorm train mode, eager; kernels from the HIP-event hooks."""
    from raw2logit_amd.processing.pipeline_torch import append_additive_layer
    recs = []
    for what, (B, H, W), additive, track in (
            ('additive layer (train.py --adv_noise_layer)', (64, 256, 256), True, False),
            ('W % 4 != 0', (64, 256, 254), False, False),
            ('frame sides not multiples of 64 (tile kernels of the backward below 4 Mi px)', (64, 200, 200), False, False),
            # model.py:228 (track_images: inputs.requires_grad = True) and track_stages=True: the stage-by-stage kernels, every
            # stage a tensor autograd holds, d/d raw produced (the fused kernels do not write it)
            ('track_stages=True with frames requiring grad (model.py:204-254): staged kernels, d/d raw', (64, 256, 256), False, True)):
        gen = torch.Generator(dev).manual_seed(0)
        raw = torch.randint(0, 4096, (B, H, W), device=dev, generator=gen, dtype=torch.int32).to(torch.float32) / 4095.0
        if track:
            raw.requires_grad_(True)
        cot = torch.randn((B, 3, H, W), device=dev, generator=gen)
        model = ParametrizedProcessing(cameras.DRONE, track_stages=track, batch_norm_output=True)
        if additive:
            append_additive_layer(model)
        model = model.to(dev).train()
        params = list(model.parameters())

        def step():
            for p in params:
                p.grad = None
            raw.grad = None
            model(raw).backward(cot)
        clock.preroll(step, None, 0.05)
        dt = clock.time_steps(step, 50, 10)
        k = kernel_times(lib, clock, step, 50)
        us = sum(v['launches'] * v['avg_us'] for v in k.values()) / 50
        recs.append({'what': what, 'shape': [B, H, W], 'ms_per_step': round(1e3 * dt / 50, 4),
                     'kernels_us_per_step': round(us, 1), 'Mpix_per_s': round(B * H * W / dt * 50 / 1e6, 1),
                     'kernels': {n.replace('r2l_launch_', '').replace('_kernel', ''): v['avg_us'] for n, v in k.items()}})
        del raw, cot, model
    return recs


def main_static(args):
    """BASELINE config 3: one step = the fused static chain over 256x1024x1024 frames per GPU (no exchange between
    ranks: static mode needs no collective)."""
    world, rank, local_rank = _world(args)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baselines(0, parametrized=False)          # before the GPU is touched (the pool forks)
    import torch
    import torch.distributed as dist
    from raw2logit_amd import _lib, cameras, functional as F_
    dev = _init_distributed(torch, dist, world, local_rank)
    lib = _lib.library_for(torch.empty(1, device=dev))[0]
    clock = Clock(torch, dist, world, dev)
    B, S = (args.batch or 256), (args.size or 1024)
    gen = torch.Generator(dev).manual_seed(rank)
    u = torch.randint(0, 4096, (B, S, S), device=dev, generator=gen, dtype=torch.int32)
    raw = u.to(torch.uint16) if args.raw_u16 else u.to(torch.float32) / 4095.0
    chain = (args.debayer, args.sharpening, args.denoising)
    norm = [0.35, 0.36, 0.35, 0.12, 0.11, 0.12] if args.normalize else None      # train.py:157-158 (Drone)

    def step():
        return F_.static_pipeline(raw, cameras.DRONE, *chain, bits=12, mean_std=norm)

    dt = clock.time_steps(step, args.steps, args.warmup)
    px = world * B * S * S
    k = kernel_times(lib, clock, step, args.steps) if not args.no_roofline else {}
    roofline = None
    if k:
        name = max(k, key=lambda n: k[n]['launches'] * k[n]['avg_us'])
        avg_us = sum(v['launches'] * v['avg_us'] for v in k.values()) / args.steps
        bpp = 14.0 if args.raw_u16 else 16.0
        ach = B * S * S * bpp / (avg_us * 1e-6) / 1e9
        traffic, source = (None, 'no PMC profile for 16-bit containers') if args.raw_u16 else \
            pmc_traffic(name, B, S, PMC_STATIC, (256, 1024))
        roofline = {'bound': 'hbm', 'kernel': name, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': traffic,
                    'traffic_source': source, 'avg_us': round(avg_us, 1), 'algo_bytes_per_px': bpp}
    if rank == 0:
        short = chain[1:] == ('none', 'none')
        out = {'metric': 'ISP Mpix/s (static fwd: demosaic->WB->CCM->clip->gamma) on 1024x1024 raw batches',
               'value': round(px * args.steps / dt / 1e6, 1), 'unit': 'Mpix/s', 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 4), 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': ('static short chain' if short else 'static chain ' + '+'.join(chain[1:])) +
                                      f' ({args.debayer}), {B}x{S}x{S} 12-bit RGGB frames per GPU'
                                      + (' as uint16 containers' if args.raw_u16 else '') + ', Drone camera parameters'
                                      + (', T.Normalize epilogue' if args.normalize else ''),
                          'global_batch': world * B, 'frame': [S, S],
                          'parallelism': f'batch shard x{world}, no collective' if world > 1 else 'single GPU'},
               'roofline': roofline, 'kernels': k}
        if cpu is not None:
            out['cpu_baseline'] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_e2e(args):
    """BASELINE configs 4 / 5: the ISP inside a whole training step (model.py:77-146, train.py:194-236):
    logits = classifier(processor(raw)); loss; backward; Adam over processor + classifier parameters.  The task
    models are plain-torch stand-ins on ATen / MIOpen (tests/standin_models.py: torchvision / smp are not in the
    image and the task models are outside the hot-path scope); with several ranks the classifier is wrapped in
    DistributedDataParallel (RCCL all-reduce overlapped with its backward), the ISP exchanges its BatchNorm
    statistics in its own two small all-gathers and its 132-float gradient in one all-reduce."""
    world, rank, local_rank = _world(args)
    import torch
    import torch.distributed as dist
    import torch.nn.functional as F
    from raw2logit_amd import _lib, cameras, functional as F_
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import standin_models as sm
    dev = _init_distributed(torch, dist, world, local_rank)
    lib = _lib.library_for(torch.empty(1, device=dev))[0]
    clock = Clock(torch, dist, world, dev)
    micro = args.workload == 'e2e-microscopy'
    B, S = (args.batch or (128 if micro else 64)), (args.size or 256)
    gen = torch.Generator(dev).manual_seed(rank)
    raw = torch.randint(0, 4096, (B, S, S), device=dev, generator=gen, dtype=torch.int32).to(torch.float32) / 4095.0
    torch.manual_seed(0)                      # identical initial weights on every rank
    proc = ParametrizedProcessing(cameras.MICROSCOPY if micro else cameras.DRONE, track_stages=False,
                                  batch_norm_output=True).to(dev).train()
    if micro:
        clf = sm.ResNet18(n_classes=16).to(dev).train()
        target = torch.randint(0, 16, (B,), device=dev, generator=gen)

        def loss_fn(logits):
            return F.cross_entropy(logits, target)
    else:
        clf = sm.SmallUNet().to(dev).train()
        target = torch.rand((B, S, S), device=dev, generator=gen) > 0.7

        def loss_fn(logits):
            return sm.dice_loss(logits, target)
    isp_params = list(proc.parameters())
    net = clf
    if world > 1:
        proc.process_group = dist.group.WORLD
        if dev.type == 'cuda':
            net = torch.nn.parallel.DistributedDataParallel(clf, device_ids=[dev.index])
        else:
            net = torch.nn.parallel.DistributedDataParallel(clf)
    opt = torch.optim.Adam(isp_params + list(clf.parameters()), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(net(proc(raw)))
        loss.backward()
        if world > 1:            # one flat asynchronous all-reduce of the ISP gradient (mean, like DDP), awaited before Adam
            F_.GradAllReduce(isp_params, dist.group.WORLD, average=True).wait()
        opt.step()
        return loss

    dt = clock.time_steps(step, args.steps, args.warmup)
    kernels = kernel_times(lib, clock, step, args.steps) if not args.no_roofline else {}
    isp_us = sum(v['launches'] * v['avg_us'] for v in kernels.values()) / max(args.steps, 1)
    loss = float(step().item())
    if rank == 0:
        ms = 1e3 * dt / args.steps
        out = {'metric': 'training-step Mpix/s (raw pixels): ISP fwd+bwd + task model fwd+bwd + Adam',
               'value': round(world * B * S * S * args.steps / dt / 1e6, 1), 'unit': 'Mpix/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
               'data': 'synthetic' + (f' -- {world} gloo ranks SHARING GPU(s): functional check of the multi-rank device path, not a '
                                      f'measurement' if dev.type == 'cuda' and world > 1 and
                                      os.environ.get('R2L_BENCH_BACKEND', 'nccl') != 'nccl' else ''),
               'config': {'workload': ('BASELINE config 4: ParametrizedProcessing (Microscopy camera parameters, '
                                       'BatchNorm train) + ResNet-18 (16 classes), CE loss, Adam'
                                       if micro else
                                       'BASELINE config 5: ParametrizedProcessing (Drone camera parameters, BatchNorm '
                                       'train) + U-Net (1 class), Dice loss, Adam') +
                                      f', {B}x{S}x{S} 12-bit RGGB frames per GPU; task model = plain-torch stand-in '
                                      f'(ATen/MIOpen, float32)',
                          'global_batch': world * B, 'frame': [S, S],
                          'parallelism': f'data parallel x{world} (DDP on the task model, ISP statistics all-gather + '
                                         f'132-float all-reduce)' if world > 1 else 'single GPU'},
               'isp': {'kernels_us_per_step': round(isp_us, 1), 'share_of_step': round(isp_us * 1e-3 / ms, 4),
                       'kernels': kernels},
               'loss': loss}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))           # the parent never touches the GPU
    if args.workload == 'static':
        return main_static(args)
    if args.workload.startswith('e2e-'):
        return main_e2e(args)
    world, rank, local_rank = _world(args)
    B, S = (args.batch or 64), (args.size or 512)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baselines(S)                 # before the GPU is touched (the pool forks)
    import torch
    import torch.distributed as dist
    import numpy as np
    from raw2logit_amd import _lib, cameras, functional as F_   # (the oracle is only touched by the cpu_baseline leg)
    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing

    dev = _init_distributed(torch, dist, world, local_rank)
    lib = _lib.library_for(torch.empty(1, device=dev))[0]      # raises if the HIP extension is missing
    # a process group exists: several ranks, or the one-rank RCCL group of R2L_BENCH_ONE_RANK_DIST (tests: the N > 1 code path
    # on a one-GPU box)
    group_on = dist.is_available() and dist.is_initialized()
    clock = Clock(torch, dist, 2 if group_on else 1, dev)

    # SURVEY.md section 8d "perf" distribution: uniform 12-bit codes, raw = u16 / 4095 (float32)
    u16 = np.random.default_rng(rank).integers(0, 4096, (B, S, S)).astype(np.uint16)
    raw = torch.from_numpy(u16 if args.raw_u16 else u16.astype(np.float32) / np.float32(4095)).to(dev)
    cot = torch.randn((B, 3, S, S), device=dev, generator=torch.Generator(dev).manual_seed(1 + rank))
    model = ParametrizedProcessing(cameras.DRONE, track_stages=False, batch_norm_output=True)
    model = model.to(dev).train()
    model.raw_bits = 12
    if group_on:
        model.process_group = dist.group.WORLD
    params = list(model.parameters())
    fwd = model
    nccl = os.environ.get('R2L_BENCH_BACKEND', 'nccl') == 'nccl'
    if args.graph:
        if group_on and not nccl:
            raise SystemExit('bench.py: --graph with several ranks needs RCCL (a gloo group moves host memory: not capturable)')
        from raw2logit_amd.graphs import StepGraph
        # several ranks: the two statistics all-gathers and the gradient all-reduce are captured with the kernels
        graph_step = StepGraph(model, raw, cot, process_group=dist.group.WORLD if group_on else None)

    pending = []

    def finish():                                    # the gradient all-reduce of the previous step lands here:
        while pending:                               # nothing needs the sum before the optimiser / the next forward
            pending.pop().wait()

    def step(eager=False):
        finish()
        if args.graph and not eager:                 # the whole step as ONE HIP graph (raw2logit_amd/graphs.py)
            graph_step.replay()
            return
        for p in params:
            p.grad = None
        y = fwd(raw)
        y.backward(cot)
        if group_on:                                 # data-parallel sum of the 132-float ISP gradient, asynchronous
            pending.append(F_.GradAllReduce(params, dist.group.WORLD))

    # GPU clocks: this process has just spent ~30 s on the host (the cpu_baseline leg) and finds the chip at idle clocks;
    # the contract's W = 5 warm-up steps are 2 ms of work and do not bring them up, so 20 timed steps (9 ms) would
    # measure the clock ramp, a few % low and noisy.  The pre-roll runs the same step untimed for a
    # fixed wall time first; the timed region itself is unchanged (W warm-up steps, then exactly K steps).
    preroll_steps = clock.preroll(step, finish, PREROLL_S if dev.type == 'cuda' else 0.0)

    # the contract's measurement: W untimed warm-up steps, then exactly K steps between barrier + synchronize
    dt = clock.time_steps(step, args.steps, args.warmup, finish)
    px_per_step = world * B * S * S
    value = px_per_step * args.steps / dt / 1e6

    roofline = None
    kernels = {}
    comm_us = None
    if not args.no_roofline:
        # instrumented pass of the same K steps with the library's per-kernel HIP-event hooks switched on
        F_.CommTimer.enable(group_on)
        # (a replayed graph makes no library calls: the per-kernel hooks see the same kernels through the eager step)
        kernels = kernel_times(lib, clock, (lambda: step(True)) if args.graph else step, args.steps, finish)
        comm_us = F_.CommTimer.report() if group_on else None
        F_.CommTimer.enable(False)

    passes = None
    if kernels:
        # roofline: the kernel with the most time among those that ARE a section-8d pass, priced at that pass's algorithmic
        # bytes (VERDICT r5 #1b); what the kernel moves by its own design sits beside it as design_*
        cand = {k: v for k, v in kernels.items() if algo_bytes_per_px(k) is not None}
        if cand:
            total = {k: v['launches'] * v['avg_us'] for k, v in cand.items()}
            dom = max(total, key=total.get)
            avg_us = cand[dom]['avg_us']
            pname, bpp, dbpp = pass_bytes(dom, cand[dom]['launches'] // max(args.steps, 1), cand)
            achieved = B * S * S * bpp / (avg_us * 1e-6) / 1e9
            design = B * S * S * dbpp / (avg_us * 1e-6) / 1e9
            traffic, source = pmc_traffic(dom, B, S) if not args.raw_u16 else \
                (None, 'no PMC profile for 16-bit containers')
            roofline = {'bound': 'hbm', 'kernel': dom, 'pass': pname, 'achieved': round(achieved, 1),
                        'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                        'traffic': traffic, 'traffic_source': source, 'avg_us': avg_us, 'algo_bytes_per_px': bpp,
                        'algo_bytes_source': 'SURVEY.md section 8d, per pass',
                        'design_bytes_per_px': dbpp, 'design_achieved': round(design, 1),
                        'design_frac': round(design / HBM_PEAK_GBS, 4)}
        # every section-8d pass of the step: its kernels' summed time against its algorithmic bytes
        acc = {}
        for k, v in kernels.items():
            fam = _family(k)
            if fam is None:
                continue
            name, algo, _ = pass_bytes(k, v['launches'] // max(args.steps, 1), kernels)
            key = name.split(' (')[0]
            a = acc.setdefault(key, {'us': 0.0, 'algo_bytes_per_px': 0.0, 'kernels': []})
            a['us'] += v['launches'] * v['avg_us'] / max(args.steps, 1)
            a['algo_bytes_per_px'] = max(a['algo_bytes_per_px'], algo or 0.0)
            a['kernels'].append(k.replace('r2l_launch_', '').replace('_kernel', ''))
        passes = [{'pass': n, 'kernels': a['kernels'], 'us_per_step': round(a['us'], 1),
                   'algo_bytes_per_px': a['algo_bytes_per_px'],
                   'frac': round(B * S * S * a['algo_bytes_per_px'] / (a['us'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
                  for n, a in acc.items() if a['us'] > 0]

    graph_ms = graph_err = graph_local_ms = graph_local_err = None
    static_c3 = small = None
    # (sub-records: a failure in one of them must not cost the headline line -- it is reported in its place)
    cold = fallback = None
    if world == 1 and dev.type == 'cuda' and (args.cold or not args.no_small_shapes) and not args.raw_u16:
        try:
            cold = cold_record(torch, lib, clock, dev, model, raw, cot, args.steps)
        except Exception as e:                       # noqa: BLE001
            cold = {'error': '%s: %s' % (type(e).__name__, e)}
    if world == 1 and dev.type == 'cuda' and not args.no_small_shapes:
        try:
            small = small_shape_records(torch, lib, clock, dev, cameras, ParametrizedProcessing)
        except Exception as e:                       # noqa: BLE001
            small = {'error': '%s: %s' % (type(e).__name__, e)}
        try:
            fallback = fallback_records(torch, lib, clock, dev, cameras, ParametrizedProcessing)
        except Exception as e:                       # noqa: BLE001
            fallback = {'error': '%s: %s' % (type(e).__name__, e)}
    if world == 1 and dev.type == 'cuda' and not args.no_static_c3:
        del raw, cot
        torch.cuda.empty_cache()
        try:
            static_c3 = static_records(torch, lib, clock, dev)
        except Exception as e:                       # noqa: BLE001
            static_c3 = {'error': '%s: %s' % (type(e).__name__, e)}

    def line(graph_ms, graph_err, graph_local_ms=None, watchdog=False, graph_local_err=None):
        """the JSON line.  Its value is the EAGER step's (what roofline / kernels describe); with several RCCL ranks the step as
        one HIP graph with its collectives captured is timed too and reported BESIDE it (ms_per_step_graph / value_graph) --
        that path has only ever run against a one-rank RCCL group, so it does not carry the headline until a multi-GPU run has
        validated it (`--graph` makes it the timed step explicitly)."""
        ms = 1e3 * dt / args.steps
        val = px_per_step / (ms * 1e-3) / 1e6
        out = {
            'metric': 'ISP Mpix/s (fwd+bwd) on 512x512 raw batches', 'value': round(val, 1),
            'unit': 'Mpix/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic' + (' (uint16 containers)' if args.raw_u16 else '') +
                    (' -- HOST EMULATION: functional check of the launcher, not a measurement'
                     if dev.type != 'cuda' else '') +
                    (f' -- {world} gloo ranks SHARING GPU(s): functional check of the multi-rank device path, not a '
                     f'measurement' if dev.type == 'cuda' and world > 1 and
                     os.environ.get('R2L_BENCH_BACKEND', 'nccl') != 'nccl' else ''),
            'config': {'workload': f'parametrized ISP fwd+bwd, BatchNorm train, {B}x{S}x{S} 12-bit RGGB '
                                   f'frames per GPU, Drone camera parameters',
                       'global_batch': world * B, 'frame': [S, S],
                       'parallelism': f'batch shard x{world}' if group_on else 'single GPU',
                       'step': 'forward + backward' + (' + 132-float grad all-reduce' if group_on else '')},
            'roofline': roofline,
            # the whole step against SURVEY.md section 8d's 52 B/px (single fused backward)
            'step_roofline': {'algo_bytes_per_px': 52.0, 'achieved': round(px_per_step / world * 52.0 / (ms * 1e-3) / 1e9, 1),
                              'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                              'frac': round(px_per_step / world * 52.0 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            # per section-8d pass: the time of ALL kernels this design spends on it against the pass's algorithmic bytes
            'pass_rooflines': passes,
            'kernels': kernels,
            # the kernels' HIP events sit between the launches of the instrumented pass: each event pair costs the queue
            # ~2 us, so that pass runs slower than the timed one -- its own wall clock is the one the kernel sum must fit in
            'instrumented_ms_per_step': round(getattr(kernel_times, 'wall_ms_per_step', 0.0), 4) if kernels else None,
            'warmup_effective': args.warmup + preroll_steps,     # untimed steps in front of the K timed ones
            'preroll': {'steps': preroll_steps, 'seconds': PREROLL_S,
                        'why': 'untimed; brings the GPU clocks up after the host-side cpu_baseline leg, before the W warm-up '
                               'steps and the K timed steps'},
        }
        if graph_ms is not None:
            out['ms_per_step_graph'] = round(graph_ms, 4)
            out['value_graph'] = round(px_per_step / (graph_ms * 1e-3) / 1e6, 1)
            out['ms_per_step_eager'] = round(1e3 * dt / args.steps, 4)
            out['value_eager'] = round(value, 1)
            out['launch'] = ('eager step timed as the value; the same step as one HIP graph with its RCCL collectives captured '
                             '(raw2logit_amd/graphs.py: StepGraph) beside it')
        if graph_local_ms is not None and graph_ms is not None:
            # what the three collectives cost INSIDE the captured step: the graph over the world's group minus the same graph
            # over a group of this rank alone (same kernels, same RCCL calls, no hop) -- the number the ~50 us budget of
            # the >= 6x target at config 5's step is about
            out['ms_per_step_graph_local'] = round(graph_local_ms, 4)
            out['graph_comm_us'] = round(1e3 * (graph_ms - graph_local_ms), 1)
        if graph_err is not None:
            out['graph_error'] = graph_err
        if graph_local_err is not None:
            out['graph_local_error'] = graph_local_err      # (graph_comm_us is then missing; nothing else depends on it)
        if watchdog:
            out['watchdog_fired'] = True
        if comm_us is not None:
            # wall time of each ISP collective per call (device events on the launch stream around the exchange, from the
            # instrumented pass): the two small all-gathers sit inside the step, the gradient all-reduce overlaps
            out['comm_us'] = comm_us
        if args.graph:
            out['config']['step'] += ' (the whole step replayed as one HIP graph)'
        if cold is not None:
            out['cold'] = cold
        if small is not None:
            out['small_shapes'] = small
        if fallback is not None:
            out['fallback_paths'] = fallback
        if static_c3 is not None:
            out['static_c3'] = static_c3
        if cpu is not None:
            out['cpu_baseline'] = cpu
        return out

    # ---- several ranks over RCCL: the same data-parallel step as ONE HIP graph with its collectives captured.  At N > 1 the eager
    # step pays the host three more times (two statistics all-gathers, the gradient all-reduce) and is host-bound; the graph costs
    # one launch.  The eager line is complete at this point: a watchdog prints it and ends the process should the trial not come
    # back (a capture or a replay that hangs on one rank), and the ranks agree on success before anything collective is timed.
    if group_on and nccl and dev.type == 'cuda' and not args.graph and args.graph_trial:
        import threading
        printed = threading.Lock()                   # exactly one JSON line, whoever gets there first

        def give_up():
            # the eager measurement is complete and valid: print it, tagged, and end the process (a capture or replay that hangs
            # on one rank cannot be unwound; destroy_process_group() would hang with it) -- with a NON-ZERO code on every rank:
            # the GPU work of this process is wedged, and a launcher must be able to tell that from a clean run
            if not printed.acquire(blocking=False):
                return
            if rank == 0:
                print(json.dumps(line(None, 'watchdog: the graph trial did not finish in %g s' % GRAPH_TRIAL_TIMEOUT_S,
                                      watchdog=True)), flush=True)
            sys.stdout.flush()
            os._exit(WATCHDOG_EXIT)
        dog = threading.Timer(GRAPH_TRIAL_TIMEOUT_S, give_up)
        dog.daemon = True
        dog.start()
        try:
            from raw2logit_amd.graphs import StepGraph
            finish()
            ok, g2 = 1.0, None
            # a group of this rank alone (every rank creates every group): the same split step with its collectives going nowhere
            solo = None
            for r_ in range(world):
                g_ = dist.new_group([r_])
                if r_ == rank:
                    solo = g_
            try:
                m2 = ParametrizedProcessing(cameras.DRONE, track_stages=False, batch_norm_output=True).to(dev).train()
                m2.raw_bits = 12
                g2 = StepGraph(m2, raw, cot, process_group=dist.group.WORLD)
            except Exception as e:                   # noqa: BLE001
                ok, graph_err = 0.0, '%s: %s' % (type(e).__name__, e)
            t = torch.tensor([ok], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)     # every rank captured, or nobody replays
            if float(t.item()) > 0:
                g2.replay()
                torch.cuda.synchronize()
                clock.preroll(g2.replay, None, 0.05)
                graph_ms = 1e3 * clock.time_steps(g2.replay, args.steps, args.warmup) / args.steps
                try:
                    old_split = os.environ.get('R2L_SPLIT_SINGLE_RANK')
                    os.environ['R2L_SPLIT_SINGLE_RANK'] = '1'      # a one-rank group takes the N > 1 path (functional.py)
                    m3 = ParametrizedProcessing(cameras.DRONE, track_stages=False, batch_norm_output=True).to(dev).train()
                    m3.raw_bits = 12
                    g3 = StepGraph(m3, raw, cot, process_group=solo)
                    g3.replay()
                    torch.cuda.synchronize()
                    graph_local_ms = 1e3 * clock.time_steps(g3.replay, args.steps, args.warmup) / args.steps
                except Exception as e:               # noqa: BLE001   (best effort: the graph's own numbers stand without it)
                    graph_local_ms = None
                    graph_local_err = '%s: %s' % (type(e).__name__, e)
                finally:
                    if old_split is None:
                        os.environ.pop('R2L_SPLIT_SINGLE_RANK', None)
                    else:
                        os.environ['R2L_SPLIT_SINGLE_RANK'] = old_split
            elif graph_err is None:
                graph_err = 'the capture failed on another rank'
        except Exception as e:                       # noqa: BLE001
            graph_ms, graph_err = None, '%s: %s' % (type(e).__name__, e)
        dog.cancel()
        if not printed.acquire(blocking=False):      # the watchdog is printing: it ends the process
            time.sleep(60)
    if rank == 0:
        print(json.dumps(line(graph_ms, graph_err, graph_local_ms, graph_local_err=graph_local_err)), flush=True)
    if group_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
