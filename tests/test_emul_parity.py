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
