"""GPU parity, part 1: the HIP kernels, called through the C ABI, replay the golden fixtures recorded from the
real reference (tests/golden/) bit-exactly — per-call entry points and the fused rollout entry point."""
import pytest

from tests import replay

pytestmark = pytest.mark.gpu

SINGLE = ['single_s9_partial2', 'single_s12_default', 'single_s12_one_channel', 'single_s10_raw',
          'single_s12_positions', 'single_s11_partial3_i32', 'single_s36_default', 'single_s12_lazyreset']
GRID = ['grid_s9_default', 'grid_s7_raw']


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


@pytest.mark.parametrize('name', SINGLE)
def test_single_snake_matches_reference(hip, name):
    replay.replay_single(hip(), replay.load(name))


@pytest.mark.parametrize('name', [n for n in SINGLE if n != 'single_s12_lazyreset'])
def test_single_snake_rollout_matches_reference(hip, name):
    replay.replay_single_rollout(hip(), replay.load(name))


@pytest.mark.parametrize('name', GRID)
def test_gridworld_matches_reference(hip, name):
    replay.replay_grid(hip(), replay.load(name))


@pytest.mark.parametrize('name', GRID)
def test_gridworld_rollout_matches_reference(hip, name):
    replay.replay_grid_rollout(hip(), replay.load(name))


MULTI = ['multi_k2_s12_default', 'multi_k4_s25_default', 'multi_k4_s25_train', 'multi_k3_s14_noboost',
         'multi_k6_s10_crowded']


@pytest.mark.parametrize('name', MULTI)
def test_multi_snake_matches_reference(hip, name):
    replay.replay_multi(hip(), replay.load_multi(name))


@pytest.mark.parametrize('name', MULTI)
def test_multi_snake_rollout_matches_reference(hip, name):
    replay.replay_multi_rollout(hip(), replay.load_multi(name))
