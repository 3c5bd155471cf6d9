"""Driver of the LOCK-STEP emulation run (tests/test_lockstep.py starts it in a subprocess under LD_PRELOAD=libasan.so):
the parity checks of the GPU suite on host memory, served by tests/_build/libr2l_lockstep.so -- every kernel in its device form,
one fiber per lane on a single host thread (one lane runs at a time: address and UB checks, no inter-lane races), built with -fsanitize=address,undefined (tests/emul/r2l_lockstep_rt.h).

    python tests/lockstep_checks.py <library> [group ...]        groups: planes shapes stream passes static canary [fuzz]

Prints one line per check; exit code 0 only if every check passed (AddressSanitizer aborts the process at the first bad access;
UBSan reports are fatal through UBSAN_OPTIONS=halt_on_error=1)."""
import os
import sys
import time
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import emul_hook  # noqa: E402

PLANE_KERNELS = ('r2l_launch_bwd1_plane', 'r2l_launch_bwd2_sums')


def main():
    lib_path = sys.argv[1]
    groups = set(sys.argv[2:]) or {'planes', 'shapes', 'stream', 'passes', 'static', 'canary'}
    emul_hook.enable(lib_path)
    lib = emul_hook.active()
    assert not lib.is_device
    import parity_checks as pc
    import test_gpu_parity as tg
    from oracle import isp_oracle as orc
    golden = {n: np.load(os.path.join(HERE, 'golden', n + '.npz'), allow_pickle=False)
              for n in ('param_cases', 'raw2rgb', 'static_cases', 'harness', 'aux_losses')}
    torch.set_num_threads(1)
    results = []
    quick = os.environ.get('R2L_LOCKSTEP_QUICK') == '1'     # the default CPU suite: a slice of every group (~ 2 minutes in all)

    def run(name, fn, expect=()):
        t0 = time.time()
        try:
            _, names = pc.kernels_launched(lib, fn)
            for k in expect:
                assert any(n.startswith(k) for n in names), (k, sorted(names))
            results.append((name, True))
            print(f'PASS {name}  [{time.time() - t0:.1f} s]  kernels: ' +
                  ' '.join(sorted(n.replace('r2l_launch_', '').replace('_kernel', '') for n in names)), flush=True)
        except Exception:   # noqa: BLE001
            results.append((name, False))
            print(f'FAIL {name}\n{traceback.format_exc()}', flush=True)

    class env:
        def __init__(self, **kw):
            self.kw = {k: str(v) for k, v in kw.items()}

        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in self.kw}
            os.environ.update(self.kw)

        def __exit__(self, *exc):
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    if 'planes' in groups:
        # golden cases: streaming forward (statistics + apply, or luma + statistics from the plane) + the backward as plane passes
        for mode, extra in (('fused-middle-pass', {}), ('split-blur', {'R2L_BWD_SPLIT_BLUR': 1})):
            for case in tg.PLANE_PARAM_CASES:
                if mode == 'split-blur' and case['name'] not in ('drone_bn_train', 'drone_perturbed_nobn', 'tiny_4x4'):
                    continue
                if quick and case['name'] not in (('tiny_4x4',) if mode == 'split-blur' else
                                                  ('drone_bn_train', 'drone_perturbed_nobn', 'tiny_4x4', 'drone_bn_eval')):
                    continue
                with env(R2L_BWD_PLANES=1, **extra):
                    run(f'golden {case["name"]} / {mode}', lambda: pc.check_param_case(case, golden, 'cpu'), PLANE_KERNELS)
    if 'shapes' in groups:
        plane_shapes = [(6, 80), (4, 8), (10, 260), (8, 516), (14, 1028)] if quick else pc.FRAME_SHAPES_PLANES
        with env(R2L_BWD_PLANES=1):
            run('frame shapes (edges at every distance from a strip / band boundary), plane passes',
                lambda: pc.check_frame_shapes('cpu', shapes=plane_shapes, conditioning=True), PLANE_KERNELS)
        with env(R2L_BWD_PLANES=1, R2L_BP_BAND=6, R2L_HB_BAND=6, R2L_B2S_BAND=6, R2L_HP_BAND=6, R2L_FA_BAND=6, R2L_FS_BAND=8):
            run('frame shapes, bands of 6 rows',
                lambda: pc.check_frame_shapes('cpu', shapes=[(14, 132)] if quick else pc.FRAME_SHAPES_PLANES[:8], conditioning=True),
                PLANE_KERNELS)
        run('frame shapes around tile boundaries (tile kernels in their device form: DPP row shifts, staged copies)',
            lambda: pc.check_frame_shapes('cpu', shapes=[(6, 78), (4, 6)] if quick else pc.FRAME_SHAPES[:6]))
    if 'stream' in groups:
        for shape in (((5, 18, 8), (1, 14, 520), (1, 12, 1028)) if quick else
                      ((2, 70, 520), (1, 40, 1028), (1, 36, 2048), (3, 66, 260), (5, 18, 8))):
            run(f'row-streaming forward {shape}: 1 / 2 / 4 / 8 wavefronts per row',
                lambda: tg.test_fused_forward_streaming_kernel(shape, 'cpu'), ('r2l_launch_fwd_stream',))
    if 'passes' in groups:
        for shape in (((1, 14, 520),) if quick else ((2, 70, 520), (1, 40, 1028), (3, 66, 260), (5, 18, 8))):
            run(f'apply pass on the kept luma plane {shape}',
                lambda: tg.test_apply_pass_reads_the_luma_plane_the_statistics_pass_kept(shape, 'cpu'), ('r2l_launch_fwd_apply',))
            run(f'backward as passes over planes {shape}',
                lambda: tg.test_backward_kernels_as_passes_over_planes(shape, 'cpu'), PLANE_KERNELS)
        for shape in (((3, 4, 4), (2, 8, 256)) if quick else ((1, 200, 256), (3, 4, 4), (2, 8, 256))):
            run(f'statistics pass as luma pass + pass over the plane {shape}',
                lambda: tg.test_statistics_pass_is_a_luma_pass_and_a_pass_over_the_kept_plane(shape, 'cpu'),
                ('r2l_launch_fwd_luma', 'r2l_launch_fwd_stats'))
    if 'static' in groups:
        for shape in (((1, 10, 516), (2, 4, 8)) if quick else ((2, 130, 516), (1, 36, 1028), (3, 66, 260), (2, 4, 8))):
            run(f'static luma chains as row-streaming kernels {shape}',
                lambda: tg.test_static_luma_chain_streaming_kernel(shape, 'cpu'), ('r2l_launch_static_chain',))
        run('static chain combinations (branch-free Malvar2004 / bilinear short chains, median, unsharp)',
            lambda: pc.check_static_combinations('cpu'), ('r2l_launch_static_stream',))
        for case in tg.DEVICE_STATIC[:(2 if quick else 6)]:
            run(f'static golden {case["name"]}', lambda: pc.check_static_case(case, golden, 'cpu'))
    if 'canary' in groups:
        import guarded_arena as ga
        import test_gpu_canary as tc
        with env(R2L_BWD_PLANES=1):
            for n, (H, W) in enumerate([(6, 80), (10, 260)] if quick else pc.FRAME_SHAPES_PLANES[:6]):
                for u16 in (False, True):
                    run(f'guard zones + poison, plane passes {H}x{W} u16={u16}',
                        lambda: ga.run_both('cpu', tc._step_bytes(2, H, W), tc._param_step('cpu', 2, H, W, True, u16, seed=50 + n,
                                                                                           kind='scene'), f'{H}x{W}'))
    if 'fuzz' in groups:
        # random frame shapes and band heights for a time budget (not part of the default run: R2L_LOCKSTEP_FUZZ_S seconds, SEED):
        # the plane passes / streaming forward at shapes nobody listed, every access under ASan, every result against the oracle
        rng = np.random.default_rng(int(os.environ.get('SEED', '1')))
        t_end = time.time() + float(os.environ.get('R2L_LOCKSTEP_FUZZ_S', '120'))
        n = 0
        while time.time() < t_end:
            n += 1
            B = int(rng.integers(1, 4))
            H = 2 * int(rng.integers(2, 24))
            W = 4 * int(rng.integers(1, 70)) if rng.random() < 0.8 else int(rng.choice([260, 508, 512, 516, 772, 1028]))
            bands = {k: 6 * int(rng.integers(1, 5)) for k in ('R2L_BP_BAND', 'R2L_HB_BAND', 'R2L_B2S_BAND', 'R2L_HP_BAND', 'R2L_FA_BAND',
                                                              'R2L_FL_BAND', 'R2L_FST_BAND')}
            bands['R2L_FS_BAND'] = 2 * int(rng.integers(2, 16))
            if rng.random() < 0.3:
                bands = {}
            kind = rng.random()
            with env(R2L_BWD_PLANES=1, **bands):
                if kind < 0.6:
                    run(f'fuzz {n}: plane passes {B}x{H}x{W} {bands}',
                        lambda: pc.check_frame_shapes('cpu', shapes=[(H, W)], B=B, conditioning=True), PLANE_KERNELS)
                elif kind < 0.8:
                    # (the apply pass against the streaming forward, bit for bit.  NOT the plane passes against the tile kernels at
                    # 2e-4 of the scale: on frames of 4-8 rows under BatchNorm two correct float32 evaluations differ by more -- the
                    # float32 oracle is 2.4e-4 from its float64 run on 3x4x772 -- and check_frame_shapes judges against the oracle)
                    run(f'fuzz {n}: apply pass on the kept luma plane {B}x{H}x{W} {bands}',
                        lambda: tg.test_apply_pass_reads_the_luma_plane_the_statistics_pass_kept((B, H, W), 'cpu'), ('r2l_launch_fwd_apply',))
                else:
                    run(f'fuzz {n}: static luma chains {B}x{H}x{W}',
                        lambda: tg.test_static_luma_chain_streaming_kernel((B, H, W), 'cpu'), ('r2l_launch_static_chain',))
    bad = [n for n, ok in results if not ok]
    print(f'{len(results) - len(bad)} of {len(results)} lock-step checks passed' + (': FAILED ' + '; '.join(bad) if bad else ''), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
