"""CPU-only: the REAL kernel source (raw2logit_amd/csrc/*.h), compiled by g++ as a host emulation
(tests/emul/r2l_emul.cpp), driven through the product's Python modules and C ABI, against the oracle and
the reference's golden vectors.  This pins tiling, halo, border, reduction and autograd-glue logic
without a GPU; tests/test_gpu_parity.py repeats the same checks on the gfx950 build."""
import pytest

from oracle.golden_cases import PARAM_CASES, STATIC_CASES
import parity_checks as pc

DEVICE_STATIC = list(STATIC_CASES)   # every chain of the golden set is built for the device


@pytest.mark.parametrize('case', PARAM_CASES, ids=[c['name'] for c in PARAM_CASES])
def test_fused_parametrized(case, golden, emulation):
    pc.check_param_case(case, golden, 'cpu')


def test_raw2rgb(golden, emulation):
    pc.check_raw2rgb(golden, 'cpu')


def test_nnprocessing_front_end(golden, emulation):
    pc.check_nnprocessing(golden, 'cpu')


@pytest.mark.parametrize('case', DEVICE_STATIC, ids=[c['name'] for c in DEVICE_STATIC])
def test_static(case, golden, emulation):
    pc.check_static_case(case, golden, 'cpu')


def test_properties(emulation):
    pc.check_ragged_and_properties('cpu')


def test_frame_shapes_around_tile_boundaries(emulation):
    pc.check_frame_shapes('cpu')


def test_harness_logits_and_adam_step(golden, emulation):
    pc.check_harness(golden, 'cpu')


TRACK_CASES = [c for c in PARAM_CASES if c['track']]


@pytest.mark.parametrize('case', TRACK_CASES, ids=[c['name'] for c in TRACK_CASES])
def test_staged_track_stages(case, golden, emulation):
    pc.check_staged_case(case, golden, 'cpu')


def test_reductions_do_not_depend_on_the_grid(emulation):
    pc.check_grid_independence('cpu')


GRAD_RAW_CASES = [c for c in PARAM_CASES if not c['track'] and c['name'] in (
    'drone_bn_train', 'drone_additive_bn', 'micro_bn_train', 'drone_ragged_tile', 'tiny_4x4', 'drone_bn_eval')]


@pytest.mark.parametrize('case', GRAD_RAW_CASES, ids=[c['name'] for c in GRAD_RAW_CASES])
def test_frames_requiring_grad_take_the_staged_kernels(case, golden, emulation):
    pc.check_staged_case(case, golden, 'cpu')


def test_16bit_containers_are_bit_identical_to_host_normalised_frames(emulation):
    pc.check_u16_ingest('cpu')


def test_static_chain_combinations(emulation):
    pc.check_static_combinations('cpu')


def test_static_numeric_arguments(golden, emulation):
    pc.check_static_options(golden, 'cpu')


def test_static_normalize_epilogue(emulation):
    pc.check_static_normalize('cpu')


def test_adversarial_aux_losses(golden, emulation):
    pc.check_aux_losses(golden, 'cpu')


def test_static_per_image_wrappers(golden, emulation):
    pc.check_static_wrappers(golden, 'cpu')


def test_weak_augmentation(emulation):
    pc.check_augmentation('cpu')


def test_error_behaviour(emulation):
    pc.check_error_behaviour('cpu')


def test_tile_walk_covers_every_tile_once(emulation):
    """the persistent kernels' tile walk (r2l_walk_init / r2l_walk_next), even shares and the uneven shares kernel B2 uses
    when two workgroups share a CU (every asym-th round is served by the older half of the workgroups only): every
    tile exactly once, whatever the grid; with asym = 4 on BASELINE config 2's shape the older workgroups take 9 tiles and
    the younger ones 7"""
    import ctypes
    import numpy as np
    lib = emulation.cdll
    lib.r2l_test_walk.restype = ctypes.c_int
    for (B, H, W) in ((64, 512, 512), (3, 200, 520), (1, 64, 64), (16, 1024, 1024), (128, 256, 256), (5, 70, 70)):
        ntiles = B * ((H + 63) // 64) * ((W + 63) // 64)
        for nblk in (512, 256, 24, 8, 7, 1):
            for asym in (0, 2, 3, 4, 5):
                owner = np.empty(ntiles, dtype=np.int32)
                visits = np.empty(ntiles, dtype=np.int32)
                most = lib.r2l_test_walk(B, H, W, nblk, asym, owner.ctypes.data_as(ctypes.c_void_p),
                                         visits.ctypes.data_as(ctypes.c_void_p))
                assert np.all(visits == 1), (B, H, W, nblk, asym, int((visits != 1).sum()))
                assert owner.min() >= 0 and owner.max() < nblk
                if (B, H, W, nblk) == (64, 512, 512, 512):
                    counts = np.bincount(owner, minlength=nblk)
                    if asym == 0:
                        assert counts.min() == counts.max() == 8
                    if asym == 4:
                        assert set(counts[:256]) == {9} and set(counts[256:]) == {7}, (counts[:4], counts[-4:])
                    assert most == counts.max()
