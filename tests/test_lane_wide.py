"""GPU parity for the one-env-per-LANE rollout of 10 x 10 and 11 x 11 SingleSnake grids (`lane_wide_rollout_kernel`,
wurm_amd/csrc/lane_wide.hpp): a 128-bit occupancy mask over the whole grid, the body as a 192-bit queue of moves, observations
('default', 'one_channel', 'partial_2', 'partial_3', none) as bit planes expanded through a table ('positions' straight
from the pair lanes).  Compared with the CPU oracle
(which knows none of that) on every output of every step: every envs-per-wave setting, ragged batches, tape lengths around the
chunk and the action batch, hostile action values, long snakes (past 64 segments: both words of the mask, all three of the
queue), start states that must go to the generic path inside the launch, chained launches, default routing from 6 144 envs
on, and the tapes recorded from the real reference injected through it.
Loop being matched: /root/reference tests/test_single_snake_env.py:24-31 over wurm/envs/single_snake.py:197-342."""
import numpy as np
import pytest

from tests import replay
from tests.backends import OracleBackend
from tests.test_lane_rollout import _compare_rollout, _fresh, _route, _same, lane_path

pytestmark = pytest.mark.gpu
MODES = ['partial_2', 'partial_3', 'default', 'one_channel', 'positions', 'none']


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


@pytest.mark.parametrize('S', [10, 11])
@pytest.mark.parametrize('epw', [8, 16, 32])
@pytest.mark.parametrize('mode', MODES)
def test_every_envs_per_wave_ragged_batch(hip, S, epw, mode):
    N, T = 3 * epw + 5, 130
    rng = np.random.RandomState(epw + len(mode) + S)
    o, h = OracleBackend(seed=17, env_offset=5), hip(seed=17, env_offset=5)
    envs = _fresh(o, N, S)
    o.call = h.call = 1
    with lane_path(epw):
        out = _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), mode)
        assert _route() == 'lane_wide'
    assert out['done'].sum() > N          # resets happened
    assert out['reward'].sum() > 0        # and food was eaten (respawn path)


@pytest.mark.parametrize('S,mode', [(10, 'partial_2'), (11, 'default'), (10, 'one_channel'), (11, 'partial_3')])
@pytest.mark.parametrize('epw,T', [(8, 1), (8, 7), (8, 8), (8, 9), (8, 127), (8, 128), (8, 129), (16, 63), (16, 64), (16, 65),
                                   (32, 1), (32, 2), (32, 31), (32, 32), (32, 33)])
def test_tape_lengths_around_chunk_and_action_batch(hip, S, mode, epw, T):
    N = 2 * epw
    rng = np.random.RandomState(T + epw)
    o, h = OracleBackend(seed=T), hip(seed=T)
    envs = _fresh(o, N, S)
    o.call = h.call = 7 + T
    with lane_path(epw):
        _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), mode)
        assert _route() == 'lane_wide'


@pytest.mark.parametrize('dtype', [np.int64, np.int32])
@pytest.mark.parametrize('S,epw', [(10, 8), (11, 32)])
def test_action_values_outside_0_to_3(hip, dtype, S, epw):
    """single_snake.py:221-222 only recognises a reversal for actions 0..3; anything else moves by action % 4 and the
    tape keeps the C remainder (this build wraps negative actions instead of faulting)."""
    N, T = 72, 140
    rng = np.random.RandomState(5)
    o, h = OracleBackend(seed=3), hip(seed=3)
    envs = _fresh(o, N, S)
    actions = rng.randint(-9, 13, size=(T, N)).astype(dtype)
    actions[::7] = np.iinfo(dtype).max
    actions[3::11] = np.iinfo(dtype).min + 1
    o.call = h.call = 1
    with lane_path(epw):
        _compare_rollout(o, h, envs, actions, 'partial_2')
        assert _route() == 'lane_wide'


def _serpentine(S, T, N, seed):
    """a scripted sweep of the interior (right along a row, one down, left along the next, ...) with some noise"""
    w = S - 2
    period = 2 * w
    tape = []
    for t in range(T):
        ph = t % period
        tape.append(3 if ph < w - 1 else 0 if ph == w - 1 else 1 if ph < 2 * w - 1 else 0)
    actions = np.repeat(np.asarray(tape, np.int64)[:, None], N, axis=1)
    rng = np.random.RandomState(seed)
    noise = rng.rand(T, N) < 0.1
    actions[noise] = rng.randint(0, 4, size=int(noise.sum()))
    return actions


@pytest.mark.parametrize('S', [10, 11])
@pytest.mark.parametrize('mode', ['partial_2', 'default'])
def test_long_snakes_and_food_respawn(hip, S, mode):
    N, T = 70, 900
    o, h = OracleBackend(seed=11), hip(seed=11)
    envs = _fresh(o, N, S)
    o.call = h.call = 1
    with lane_path(16):
        out = _compare_rollout(o, h, envs, _serpentine(S, T, N, 2), mode)
    assert out['reward'].sum() > 3 * N


@pytest.mark.parametrize('S', [10, 11])
@pytest.mark.parametrize('mode', MODES)
def test_snakes_longer_than_64_segments(hip, S, mode):
    """Hand-built start states: snakes of up to (S - 2)^2 - 1 segments coiled through the interior (cells in both words of the
    occupancy mask, moves in all three words of the queue, every free cell a food candidate)."""
    N, T = 16, 70
    o, h = OracleBackend(seed=4), hip(seed=4)
    envs = _fresh(o, N, S)
    w = S - 2
    path = []
    for r in range(1, w + 1):                    # boustrophedon over the interior
        cols = range(1, w + 1) if r % 2 == 1 else range(w, 0, -1)
        path += [(r, c) for c in cols]
    for i, L in enumerate([40, 33, 17, w * w - 1, w * w - 2] + ([63, 64, 65] if w * w > 66 else [50, 60, 62])):
        e = np.zeros((3, S, S), np.float32)
        for v, (y, x) in enumerate(path[:L], start=1):
            e[2, y, x] = v
        hy, hx = path[L - 1]
        e[1, hy, hx] = 1
        fy, fx = path[L]                         # food right in front of the head: the snake grows at once
        e[0, fy, fx] = 1
        envs[i] = e
    assert (o.single_check(envs) == 0).all()
    rng = np.random.RandomState(9)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    o.call = h.call = 3
    with lane_path(8):
        _compare_rollout(o, h, envs, actions, mode)
        assert _route() == 'lane_wide'


@pytest.mark.parametrize('S', [10, 11])
@pytest.mark.parametrize('epw', [8, 32])
def test_start_states_outside_the_domain(hip, S, epw):
    """Start states the lane kernel must hand to the generic path inside the launch: head on the border ring (a done env
    that was not reset), food on a body cell, no head, two foods, a body whose values are not edge-adjacent, a body value
    missing — mixed with ordinary envs in the same waves."""
    N, T = 150, 90
    rng = np.random.RandomState(8)
    o, h = OracleBackend(seed=21), hip(seed=21)
    envs = _fresh(o, N, S)
    for _ in range(9):  # step without resets: finished envs keep their head on the ring / lose it
        o.single_step(envs, rng.randint(0, 4, size=N).astype(np.int64), 'none')
    fresh = _fresh(OracleBackend(seed=22), 1, S)[0]
    ys, xs = np.nonzero(fresh[2] == 1)
    fresh[0] = 0
    fresh[0, ys[0], xs[0]] = 1          # food on the tail cell of an otherwise regular env
    envs[5] = fresh
    envs[6, 0, 3, 3] = 1                # (possibly) a second food
    envs[7, 1] = 0                      # no head
    gap = _fresh(OracleBackend(seed=23), 1, S)[0]
    ys, xs = np.nonzero(gap[2] == 1)
    gap[2, ys[0], xs[0]] = 0            # tail moved two cells away: values 1..3 present, not adjacent
    free = np.argwhere((gap.sum(0)[1:-1, 1:-1] == 0)) + 1
    far = [c for c in free if abs(c[0] - ys[0]) + abs(c[1] - xs[0]) > 2][0]
    gap[2, far[0], far[1]] = 1
    envs[70] = gap
    hole = _fresh(OracleBackend(seed=24), 1, S)[0]
    hole[2][hole[2] == 2] = 0           # body value 2 missing
    envs[71] = hole
    envs[72, 2, 0, 0] = 200             # a body value beyond the table
    assert (o.single_check(envs) != 0).sum() >= 3
    o.call = h.call = 50
    for mode in ('partial_2', 'default'):
        with lane_path(epw):
            _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), mode, check=False)
            assert _route() == 'lane_wide'


@pytest.mark.parametrize('S', [10, 11])
def test_chained_launches_equal_one_launch(hip, S):
    """Two launches of 96 steps == one launch of 192 steps (state handed over through HBM, call counter continued)."""
    N = 85
    rng = np.random.RandomState(4)
    actions = rng.randint(0, 4, size=(192, N)).astype(np.int64)
    h1, h2 = hip(seed=6), hip(seed=6)
    with lane_path(16):
        e1, e2 = _fresh(h1, N, S), _fresh(h2, N, S)
        a1, a2 = actions.copy(), actions.copy()
        whole = h1.single_rollout(e1, a1, 'default')
        first = h2.single_rollout(e2, a2[:96], 'default')
        second = h2.single_rollout(e2, a2[96:], 'default')
    for k in whole:
        _same(whole[k], np.concatenate([first[k], second[k]]), k)
    _same(e1, e2, 'final state')
    _same(a1, a2, 'actions')


@pytest.mark.parametrize('S,mode', [(10, 'partial_2'), (11, 'default'), (11, 'partial_3'), (10, 'one_channel')])
def test_lane_kernel_equals_one_env_per_wave_kernel(hip, S, mode):
    """the same launch through the one-env-per-wave kernels (lane kernels forbidden) and through the lane kernel"""
    N, T = 700, 130
    rng = np.random.RandomState(12)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    h1, h2 = hip(seed=9, env_offset=123), hip(seed=9, env_offset=123)
    e1, e2 = _fresh(h1, N, S), _fresh(h2, N, S)
    a1, a2 = actions.copy(), actions.copy()
    with lane_path(min_envs=1 << 40):
        r1 = h1.single_rollout(e1, a1, mode)
        assert _route() != 'lane_wide'
    with lane_path(32):
        r2 = h2.single_rollout(e2, a2, mode)
        assert _route() == 'lane_wide'
    for k in r1:
        _same(r1[k], r2[k], k)
    _same(e1, e2, 'final state')
    _same(a1, a2, 'actions')


@pytest.mark.parametrize('S,N,mode', [(10, 6144 + 37, 'partial_2'), (11, 8192, 'default'), (10, 12288 + 3, 'one_channel'),
                                      (11, 40960 + 5, 'partial_2')])
def test_large_batches_default_routing(hip, S, N, mode):
    """From 6 144 envs on the rollout entry point takes the lane kernel by itself; call counters and env ids beyond 32 bits;
    four waves per workgroup; the last wave ragged; 8 / 16 / 32 envs per wave by batch size."""
    T = 34
    rng = np.random.RandomState(N)
    o, h = OracleBackend(seed=31, env_offset=(1 << 33) + 5), hip(seed=31, env_offset=(1 << 33) + 5)
    envs = _fresh(o, N, S)
    o.call = h.call = (1 << 40) + 3
    _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), mode)
    assert _route() == 'lane_wide'


def test_modes_the_lane_kernel_does_not_serve_keep_their_kernels(hip):
    N, T = 6400, 5
    rng = np.random.RandomState(1)
    for S, mode in ((10, 'raw'), (11, 'raw'), (10, 'partial_4'), (11, 'partial_1')):
        o, h = OracleBackend(seed=2), hip(seed=2)
        envs = _fresh(o, N, S)
        o.call = h.call = 9
        _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), mode)
        assert _route() != 'lane_wide'


def test_reference_tapes_injected_through_the_lane_kernel(hip):
    """tests/golden/single_s11_partial3_i32.npz (11 x 11, partial_3, int32 actions) replayed through
    lane_wide_rollout_kernel<16, 11, partial, 7, INJ> against the reference's own record; single_s10_raw.npz (recorded in a
    mode the lane kernel does not write): its random outcomes drive the same launch in the modes it does write, the
    reference's record checks everything but the observation, the oracle (injected alike) the observation."""
    with lane_path():
        replay.replay_single_rollout(hip(), replay.load('single_s11_partial3_i32'))
        assert _route() == 'lane_wide'
        for name in ('single_s10_raw', 'single_s11_partial3_i32'):
            tape = replay.load(name)
            for mode in ('partial_2', 'default', 'one_channel', 'partial_3', 'positions'):
                replay.replay_single_rollout_vs_oracle(hip(), OracleBackend(), tape, mode)
                assert _route() == 'lane_wide'
