"""The dispatch table of the SingleSnake / SimpleGridworld entry points (wurm_amd/csrc/single_snake.hip: route_of), one
case per row: which kernel serves a call is a function of (kind, size, batch, observation, injection) alone, read off
ONE table, and `wurm_single_last_route()` names the row that was taken.  Results never depend on the row (every route
is compared with the oracle in its own test file); this file pins the table itself.
The reference has one code path for every shape (/root/reference wurm/envs/single_snake.py:197-342)."""
import numpy as np
import pytest

from tests.backends import OracleBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


def _route():
    from wurm_amd import _lib
    return _lib.lib().wurm_single_last_route().decode()


def _fresh(o, N, S):
    envs = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(envs, np.ones(N, np.uint8), 'none')
    return envs


ROLLOUT_ROWS = [
    # (S, N, mode, lane threshold, expected row)
    (9, 70, 'partial_2', 0, 'lane_rollout'),
    (9, 70, 'one_channel', 0, 'lane_rollout'),
    (9, 70, 'default', 0, 'lane_rollout'),
    (9, 70, 'positions', 0, 'lane_rollout'),
    (9, 70, 'partial_2', 1 << 40, 'rollout_s9'),
    (9, 70, 'none', 1 << 40, 'rollout_s9'),
    (9, 70, 'partial_3', 0, 'lane_rollout'),            # (round 5: 7 x 7 crops through bit planes)
    (9, 70, 'raw', 0, 'lane_rollout'),                  # (round 5: the state itself through a byte slab)
    (9, 70, 'partial_3', 1 << 40, 'rollout_s9'),
    (9, 70, 'raw', 1 << 40, 'generic'),
    (9, 70, 'partial_4', 0, 'rollout_generic_partial'), # 9 x 9 crops are outside the lane kernel's domain
    (9, 70, 'one_channel', 1 << 40, 'generic'),
    (9, 70, 'partial_4', 1 << 40, 'rollout_generic_partial'),
    (10, 70, 'partial_2', 0, 'lane_wide'),              # (round 6: 10 x 10 / 11 x 11, one env per lane on 128-bit masks)
    (11, 70, 'none', 0, 'lane_wide'),
    (11, 70, 'default', 0, 'lane_wide'),
    (10, 70, 'one_channel', 0, 'lane_wide'),
    (11, 70, 'partial_3', 0, 'lane_wide'),
    (11, 70, 'positions', 0, 'lane_wide'),
    (10, 70, 'partial_2', 1 << 40, 'rollout_lean'),
    (11, 70, 'none', 1 << 40, 'rollout_lean'),
    (11, 70, 'partial_5', 0, 'rollout_generic_partial'),
    (11, 70, 'default', 1 << 40, 'generic'),
    (10, 70, 'raw', 0, 'generic'),                      # ('raw' and other crops stay with the one-env-per-wave kernels)
    (12, 40, 'partial_2', 0, 'grid_rollout'),           # (the suite runs with WURM_GRID_ROLLOUT_MIN_SIZE = 12: tests/conftest.py; the
    (20, 12, 'default', 0, 'grid_rollout'),             #  shipped thresholds by observation mode: tests/test_grid_rollout_routing.py)
]


@pytest.mark.parametrize('S,N,mode,min_envs,row', ROLLOUT_ROWS)
def test_rollout_rows(hip, S, N, mode, min_envs, row):
    from wurm_amd._lib import knobs
    rng = np.random.RandomState(S * 100 + N)
    o, h = OracleBackend(seed=3), hip(seed=3)
    envs = _fresh(o, N, S)
    o.call = h.call = 1
    actions = rng.randint(0, 4, size=(12, N)).astype(np.int64)
    eo, eh, ao, ah = envs.copy(), envs.copy(), actions.copy(), actions.copy()
    with knobs(WURM_LANE_ROLLOUT_MIN_ENVS=min_envs):
        rh = h.single_rollout(eh, ah, mode)
        assert _route() == row
    ro = o.single_rollout(eo, ao, mode)
    for k in ro:
        if ro[k] is not None:
            a, b = np.asarray(ro[k]), np.asarray(rh[k])
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                                  b.view(np.uint32) if b.dtype == np.float32 else b), k
    assert np.array_equal(eo, eh)


STEP_ROWS = [
    # (S, N, mode, lane-step threshold, grid-step threshold in cells, expected row)
    (9, 70, 'partial_2', 0, 1 << 40, 'lane_step'),
    (10, 70, 'none', 0, 1 << 40, 'lane_step'),
    (9, 70, 'one_channel', 0, 1 << 40, 'generic'),       # the per-call lane kernel writes crops only (the mirror serves the rest)
    (9, 70, 'partial_2', 1 << 40, 1 << 40, 'generic'),
    (12, 40, 'partial_2', 0, 0, 'grid_step'),
    (12, 40, 'partial_2', 0, 1 << 40, 'generic'),
    (36, 8, 'default', 0, 0, 'grid_step'),
]


@pytest.mark.parametrize('S,N,mode,lane_min,grid_min,row', STEP_ROWS)
def test_step_rows(hip, S, N, mode, lane_min, grid_min, row):
    from wurm_amd._lib import knobs
    rng = np.random.RandomState(S + N)
    o, h = OracleBackend(seed=5), hip(seed=5)
    envs = _fresh(o, N, S)
    o.call = h.call = 1
    actions = rng.randint(0, 4, size=N).astype(np.int64)
    eo, eh = envs.copy(), envs.copy()
    with knobs(WURM_LANE_STEP_MIN_ENVS=lane_min, WURM_GRID_STEP_MIN_CELLS=grid_min):
        rh = h.single_step(eh, actions.copy(), mode)
        assert _route() == row
    ro = o.single_step(eo, actions.copy(), mode)
    for a, b in zip(ro, rh):
        if a is not None:
            a, b = np.asarray(a), np.asarray(b)
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                                  b.view(np.uint32) if b.dtype == np.float32 else b)
    assert np.array_equal(eo, eh)


def test_grid_rows_in_an_env_dependent_order(hip):
    """WURM_GRID_ROTATE = 1 (off by default): the clock-grid kernels start every env's observation rows at an env-dependent
    row — the same bytes in another order (tools/placement_probe.py)"""
    from wurm_amd._lib import knobs
    for k in (2, 4, 256):   # (round 5: the coarser skews, start row (env % k) * iters / k — profiles/r05_placement_probe.txt)
        with knobs(WURM_GRID_ROTATE=k):
            test_rollout_rows(hip, 20, 12, 'default', 0, 'grid_rollout')
            test_step_rows(hip, 36, 8, 'default', 0, 0, 'grid_step')
    with knobs(WURM_GRID_ROTATE=1):
        test_rollout_rows(hip, 20, 12, 'default', 0, 'grid_rollout')
        test_rollout_rows(hip, 36, 9, 'raw', 0, 'grid_rollout')
        test_rollout_rows(hip, 17, 11, 'one_channel', 0, 'grid_rollout')
        test_step_rows(hip, 36, 8, 'default', 0, 0, 'grid_step')


def test_reset_and_observe_are_generic(hip):
    o, h = OracleBackend(seed=9), hip(seed=9)
    envs = _fresh(o, 33, 9)
    e = envs.copy()
    h.single_reset(e, np.ones(33, np.uint8), 'partial_2')
    assert _route() == 'generic'
    h.single_observe(e, 'default')
    assert _route() == 'generic'


def test_gridworld_calls_are_generic_and_large_rollouts_take_the_lane_row(hip):
    from wurm_amd._lib import knobs
    h = hip(seed=1)
    envs = np.zeros((16, 2, 9, 9), np.float32)
    h.grid_reset(envs, np.ones(16, np.uint8), (4, 4), 'default')
    h.grid_step(envs, np.zeros(16, np.int64), 'default')
    assert _route() == 'generic'
    for mode, min_envs, row in [('default', 0, 'gridworld_lane'), ('raw', 0, 'gridworld_lane'),
                                ('positions', 0, 'gridworld_lane'), ('none', 0, 'gridworld_lane'),
                                ('default', 17, 'generic'), ('default', 16, 'gridworld_lane')]:
        with knobs(WURM_LANE_ROLLOUT_MIN_ENVS=min_envs):
            h.grid_rollout(envs, np.zeros((3, 16), np.int64), (4, 4), mode)
            assert _route() == row, (mode, min_envs)
    for mode, min_envs, row in [('default', 0, 'gridworld_lane_step'), ('raw', 0, 'gridworld_lane_step'),
                                ('positions', 0, 'gridworld_lane_step'), ('none', 0, 'gridworld_lane_step'),
                                ('default', 17, 'generic'), ('default', 16, 'gridworld_lane_step')]:
        with knobs(WURM_LANE_STEP_MIN_ENVS=min_envs):  # (the per-call step: 12 288 envs and more by default)
            h.grid_step(envs, np.zeros(16, np.int64), mode)
            assert _route() == row, (mode, min_envs)
    with knobs(WURM_LANE_ROLLOUT_MIN_ENVS=0):  # recorded outcomes: the one-env-per-wave kernel consumes them
        h.grid_rollout(envs, np.zeros((3, 16), np.int64), (4, 4), 'default', inject_food=np.zeros((3, 16), np.int32),
                       inject_reset=np.zeros((3, 16), np.int32))
        assert _route() == 'generic'


def test_defaults_of_the_thresholds():
    """the batch sizes from which the lane rows take over (options.hpp) — a change here changes which kernel BASELINE's
    configs run on, so it is pinned"""
    from wurm_amd import _lib
    l = _lib.lib()
    assert l.wurm_get_option(b'WURM_LANE_ROLLOUT_MIN_ENVS') == 6144
    assert l.wurm_get_option(b'WURM_LANE_STEP_MIN_ENVS') == 12288
    assert l.wurm_get_option(b'WURM_GRID_STEP_MIN_CELLS') == 1 << 20
