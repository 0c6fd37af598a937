"""GPU parity for the one-env-per-LANE rollout (`lane_rollout_kernel`, wurm_amd/csrc/lane_rollout.hpp): SingleSnake 9 x 9,
`partial_2` crop or no observation, the kernel that takes fused rollouts of 6 144 envs and more (BASELINE configs[2]:
65 536 envs, 8 192 per GPU).  It keeps the body as a queue of moves and an occupancy mask per lane, works on chunks of
64 / EPW steps, builds the crops as bit planes and expands them through a table — so it is compared with the CPU oracle
(which knows none of that) on every output of every step: every envs-per-wave setting, ragged batches, tape lengths
around the chunk and action-batch sizes, hostile action values, long snakes, start states that must go to the generic
path inside the launch, chained launches, and the tape recorded from the real reference injected through it.
Loop being matched: /root/reference tests/test_single_snake_env.py:24-31 over wurm/envs/single_snake.py:197-342."""
import contextlib
import os

import numpy as np
import pytest

from tests import replay
from tests.backends import OracleBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


@contextlib.contextmanager
def lane_path(epw=None, min_envs=0):
    """force (min_envs=0) or forbid (min_envs=huge) the lane kernel, optionally with a fixed envs-per-wave"""
    from wurm_amd._lib import knobs
    with knobs(WURM_LANE_ROLLOUT_MIN_ENVS=min_envs, WURM_LANE_ROLLOUT_EPW=epw):
        yield


def _same(a, b, what):
    if a is None and b is None:
        return
    a, b = np.asarray(a), np.asarray(b)
    x, y = (a.view(np.uint32), b.view(np.uint32)) if a.dtype == np.float32 else (a, b)
    assert x.shape == y.shape, f'{what}: shape {x.shape} vs {y.shape}'
    bad = np.argwhere(x != y)
    assert len(bad) == 0, f'{what}: {len(bad)} mismatches, first at {bad[0].tolist()}: {a[tuple(bad[0])]} vs {b[tuple(bad[0])]}'


def _fresh(backend, N, S=9):
    envs = np.zeros((N, 3, S, S), np.float32)
    backend.single_reset(envs, np.ones(N, np.uint8), 'none')
    return envs


def _compare_rollout(o, h, envs, actions, mode, check=True):
    eo, eh = envs.copy(), envs.copy()
    ao, ah = actions.copy(), actions.copy()
    ro, rh = o.single_rollout(eo, ao, mode), h.single_rollout(eh, ah, mode)
    for k in ro:
        _same(ro[k], rh[k], k)
    _same(ao, ah, 'sanitised actions')
    _same(eo, eh, 'final state')
    if check:
        assert (o.single_check(eo) == 0).all()
    return ro


@pytest.mark.parametrize('epw', [4, 8, 16, 32, 64])
@pytest.mark.parametrize('mode', ['partial_2', 'none'])
def test_every_envs_per_wave_ragged_batch(hip, epw, mode):
    N, T = 3 * epw + 5, 150
    rng = np.random.RandomState(epw + len(mode))
    o, h = OracleBackend(seed=17, env_offset=5), hip(seed=17, env_offset=5)
    envs = _fresh(o, N)
    o.call = h.call = 1
    with lane_path(epw):
        out = _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), mode)
    assert out['done'].sum() > N          # resets happened
    assert out['reward'].sum() > 0        # and food was eaten (respawn path)


@pytest.mark.parametrize('epw,T', [(64, 1), (64, 15), (64, 16), (64, 17), (64, 33), (8, 1), (8, 7), (8, 8), (8, 9), (8, 127),
                                   (8, 128), (8, 129), (8, 300), (4, 15), (4, 16), (4, 17), (4, 255), (4, 256), (4, 257),
                                   (16, 63), (16, 64), (16, 65), (32, 31), (32, 32), (32, 33)])
def test_tape_lengths_around_chunk_and_action_batch(hip, epw, T):
    N = 2 * epw
    rng = np.random.RandomState(T + epw)
    o, h = OracleBackend(seed=T), hip(seed=T)
    envs = _fresh(o, N)
    o.call = h.call = 7 + T
    with lane_path(epw):
        _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), 'partial_2')


@pytest.mark.parametrize('dtype', [np.int64, np.int32])
@pytest.mark.parametrize('epw', [8, 64])
def test_action_values_outside_0_to_3(hip, dtype, epw):
    """single_snake.py:221-222 only recognises a reversal for actions 0..3; anything else moves by action % 4 and the
    tape keeps the C remainder (this build wraps negative actions instead of faulting)."""
    N, T = 72, 140
    rng = np.random.RandomState(5)
    o, h = OracleBackend(seed=3), hip(seed=3)
    envs = _fresh(o, N)
    actions = rng.randint(-9, 13, size=(T, N)).astype(dtype)
    actions[::7] = np.iinfo(dtype).max
    actions[3::11] = np.iinfo(dtype).min + 1
    o.call = h.call = 1
    with lane_path(epw):
        _compare_rollout(o, h, envs, actions, 'partial_2')


@pytest.mark.parametrize('epw', [4, 64])
def test_long_snakes_and_food_respawn(hip, epw):
    """A scripted serpentine sweep of the interior makes snakes long: self collisions, many food draws among few free
    cells, move queues far beyond one register."""
    N, T = 70, 900
    o, h = OracleBackend(seed=11), hip(seed=11)
    envs = _fresh(o, N)
    tape = []
    for t in range(T):
        phase = t % 12
        tape.append(3 if phase < 5 else 0 if phase == 5 else 1 if phase < 11 else 0)
    actions = np.repeat(np.asarray(tape, np.int64)[:, None], N, axis=1)
    rng = np.random.RandomState(2)
    noise = rng.rand(T, N) < 0.12
    actions[noise] = rng.randint(0, 4, size=int(noise.sum()))
    o.call = h.call = 1
    with lane_path(epw):
        out = _compare_rollout(o, h, envs, actions, 'partial_2')
    assert out['reward'].sum() > 3 * N
    # a snake that grows past 16 segments needs the second register of the move queue
    lengths = envs[:, 2].max(axis=(1, 2))
    assert lengths.max() >= 3


def test_snake_longer_than_32_segments(hip):
    """Hand-built start state: a 40-segment snake coiled through the interior (moves in all three queue registers);
    the lane kernel must read it, step it and write it back exactly."""
    N, T, S = 8, 60, 9
    o, h = OracleBackend(seed=4), hip(seed=4)
    envs = _fresh(o, N)
    path = []
    for r in range(1, 8):                       # boustrophedon over the 7 x 7 interior
        cols = range(1, 8) if r % 2 == 1 else range(7, 0, -1)
        path += [(r, c) for c in cols]
    for i, L in enumerate([40, 33, 17, 48]):
        e = np.zeros((3, S, S), np.float32)
        for v, (y, x) in enumerate(path[:L], start=1):
            e[2, y, x] = v
        hy, hx = path[L - 1]
        e[1, hy, hx] = 1
        fy, fx = path[L]                        # food right in front of the head: the snake grows at once
        e[0, fy, fx] = 1
        envs[i] = e
    assert (o.single_check(envs) == 0).all()
    rng = np.random.RandomState(9)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    o.call = h.call = 3
    with lane_path(8):
        _compare_rollout(o, h, envs, actions, 'partial_2')


@pytest.mark.parametrize('epw', [4, 16, 64])
def test_start_states_outside_the_domain(hip, epw):
    """Start states the lane kernel must hand to the generic path inside the launch: head on the border ring (a done env
    that was not reset), food on a body cell, no head, two foods, a body whose values are not edge-adjacent, a body value
    missing — mixed with ordinary envs in the same waves."""
    N, T, S = 150, 90, 9
    rng = np.random.RandomState(8)
    o, h = OracleBackend(seed=21), hip(seed=21)
    envs = _fresh(o, N)
    for _ in range(9):  # step without resets: finished envs keep their head on the ring / lose it
        o.single_step(envs, rng.randint(0, 4, size=N).astype(np.int64), 'none')
    fresh = _fresh(OracleBackend(seed=22), 1)[0]
    ys, xs = np.nonzero(fresh[2] == 1)
    fresh[0] = 0
    fresh[0, ys[0], xs[0]] = 1          # food on the tail cell of an otherwise regular env
    envs[5] = fresh
    envs[6, 0, 3, 3] = 1                # (possibly) a second food
    envs[7, 1] = 0                      # no head
    gap = _fresh(OracleBackend(seed=23), 1)[0]
    ys, xs = np.nonzero(gap[2] == 1)
    gap[2, ys[0], xs[0]] = 0            # tail moved two cells away: values 1..3 present, not adjacent
    free = np.argwhere((gap.sum(0)[1:-1, 1:-1] == 0)) + 1
    far = [c for c in free if abs(c[0] - ys[0]) + abs(c[1] - xs[0]) > 2][0]
    gap[2, far[0], far[1]] = 1
    envs[70] = gap
    hole = _fresh(OracleBackend(seed=24), 1)[0]
    hole[2][hole[2] == 2] = 0           # body value 2 missing
    envs[71] = hole
    assert (o.single_check(envs) != 0).sum() >= 3
    o.call = h.call = 50
    with lane_path(epw):
        _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), 'partial_2', check=False)


def test_chained_launches_equal_one_launch(hip):
    """Two launches of 96 steps == one launch of 192 steps (state handed over through HBM, call counter continued)."""
    N = 85
    rng = np.random.RandomState(4)
    actions = rng.randint(0, 4, size=(192, N)).astype(np.int64)
    h1, h2 = hip(seed=6), hip(seed=6)
    with lane_path(16):
        e1, e2 = _fresh(h1, N), _fresh(h2, N)
        a1, a2 = actions.copy(), actions.copy()
        whole = h1.single_rollout(e1, a1, 'partial_2')
        first = h2.single_rollout(e2, a2[:96], 'partial_2')
        second = h2.single_rollout(e2, a2[96:], 'partial_2')
    for k in whole:
        _same(whole[k], np.concatenate([first[k], second[k]]), k)
    _same(e1, e2, 'final state')
    _same(a1, a2, 'actions')


def test_lane_kernel_equals_one_env_per_wave_kernel(hip):
    """the same launch through rollout_s9_kernel (lane kernel forbidden) and through the lane kernel"""
    N, T = 700, 130
    rng = np.random.RandomState(12)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    h1, h2 = hip(seed=9, env_offset=123), hip(seed=9, env_offset=123)
    e1, e2 = _fresh(h1, N), _fresh(h2, N)
    a1, a2 = actions.copy(), actions.copy()
    with lane_path(min_envs=1 << 40):
        r1 = h1.single_rollout(e1, a1, 'partial_2')
    with lane_path(32):
        r2 = h2.single_rollout(e2, a2, 'partial_2')
    for k in r1:
        _same(r1[k], r2[k], k)
    _same(e1, e2, 'final state')
    _same(a1, a2, 'actions')


@pytest.mark.parametrize('N,epw', [(6144 + 37, None), (8192, None), (2048 + 640, 64)])
def test_large_batches_default_routing(hip, N, epw):
    """Above 6 144 envs the rollout entry point takes the lane kernel by itself; call counters and env ids beyond 32 bits;
    four waves per workgroup; the last wave ragged."""
    T = 70
    rng = np.random.RandomState(N)
    o, h = OracleBackend(seed=31, env_offset=(1 << 33) + 5), hip(seed=31, env_offset=(1 << 33) + 5)
    envs = _fresh(o, N)
    o.call = h.call = (1 << 40) + 3
    ctx = lane_path(epw, min_envs=2048) if epw else contextlib.nullcontext()
    with ctx:
        _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), 'partial_2')


def test_reference_tape_injected_through_the_lane_kernel(hip):
    """tests/golden/single_s9_partial2.npz — 48 envs x 150 steps of the REAL reference (random outcomes recorded:
    food cells, reset seed / direction / food) — replayed through lane_rollout_kernel<16, partial, INJ>."""
    with lane_path():
        replay.replay_single_rollout(hip(), replay.load('single_s9_partial2'))


def test_bench_shape_8192x128(hip):
    """exactly the launch `bench.py` times for the cfg3 per-GPU share (8 192 envs x 128 steps, RNG mode) vs the oracle"""
    N, T = 8192, 128
    rng = np.random.RandomState(0)
    o, h = OracleBackend(seed=0), hip(seed=0)
    envs = _fresh(o, N)
    o.call = h.call = 1
    _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), 'partial_2')


# ---- round 4: every other observation mode through the lane kernel (lr_write_generic)
GENERIC_MODES = ['one_channel', 'default', 'positions', 'partial_0', 'partial_1',
                 'partial_3', 'raw',   # (round 5: 7 x 7 crops through bit planes, 'raw' through a byte slab)
                 'partial_4']          # (9 x 9 crops are routed to the one-env-per-wave kernels: still one launch, same results)


@pytest.mark.parametrize('epw', [4, 8, 16, 32, 64])
@pytest.mark.parametrize('mode', GENERIC_MODES)
def test_other_observation_modes_every_envs_per_wave(hip, epw, mode):
    """'one_channel' is the reference's constructor default (single_snake.py:55-65); ragged batch, resets, food respawn"""
    N, T = 2 * epw + 3, 90
    rng = np.random.RandomState(epw + len(mode))
    o, h = OracleBackend(seed=19, env_offset=7), hip(seed=19, env_offset=7)
    envs = _fresh(o, N)
    o.call = h.call = 1
    with lane_path(epw):
        from wurm_amd import _lib
        n0 = _lib.lib().wurm_launch_count()
        out = _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), mode)
        assert _lib.lib().wurm_launch_count() - n0 == 1     # one launch: the lane kernel, no second pass
    assert out['done'].sum() > N and out['reward'].sum() > 0


@pytest.mark.parametrize('mode', ['one_channel', 'default', 'partial_3', 'raw'])
def test_other_modes_start_states_outside_the_domain(hip, mode):
    """envs the lane kernel hands to the one-env-per-wave code inside the launch (head on the ring, no head, two foods),
    mixed with ordinary ones: the generic writer must leave their observations to that code"""
    N, T = 150, 60
    rng = np.random.RandomState(8)
    o, h = OracleBackend(seed=21), hip(seed=21)
    envs = _fresh(o, N)
    for _ in range(9):
        o.single_step(envs, rng.randint(0, 4, size=N).astype(np.int64), 'none')
    envs[6, 0, 3, 3] = 1
    hole = _fresh(OracleBackend(seed=24), 1)[0]
    hole[2][hole[2] == 2] = 0
    envs[71] = hole
    if mode.startswith('partial'):
        # the reference raises for an env without a head (:191); this build writes zeros — not part of this comparison
        keep = envs[:, 1].sum(axis=(1, 2)) > 0
        envs = envs[keep]
        N = envs.shape[0]
    o.call = h.call = 50
    with lane_path(16):
        _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), mode, check=False)


@pytest.mark.parametrize('mode', ['one_channel', 'default', 'partial_3', 'raw'])
def test_other_modes_large_batch_default_routing(hip, mode):
    N, T = 6144 + 37, 40
    rng = np.random.RandomState(N)
    o, h = OracleBackend(seed=31, env_offset=(1 << 33) + 5), hip(seed=31, env_offset=(1 << 33) + 5)
    envs = _fresh(o, N)
    o.call = h.call = (1 << 40) + 3
    _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), mode)


@pytest.mark.parametrize('epw', [4, 32, 64])
@pytest.mark.parametrize('mode', ['raw', 'partial_3'])
def test_long_snakes_in_the_round_5_modes(hip, epw, mode):
    """the serpentine sweep of test_long_snakes_and_food_respawn: long bodies ('raw' shows every body VALUE: the walk down
    the move queue), self collisions ('raw' shows the SUM of the new head and the segment it ran into,
    single_snake.py:258-262), heads on the ring (7 x 7 windows that reach three cells past the grid)"""
    N, T = 70, 600
    o, h = OracleBackend(seed=11), hip(seed=11)
    envs = _fresh(o, N)
    tape = []
    for t in range(T):
        phase = t % 12
        tape.append(3 if phase < 5 else 0 if phase == 5 else 1 if phase < 11 else 0)
    actions = np.repeat(np.asarray(tape, np.int64)[:, None], N, axis=1)
    rng = np.random.RandomState(2)
    noise = rng.rand(T, N) < 0.12
    actions[noise] = rng.randint(0, 4, size=int(noise.sum()))
    o.call = h.call = 1
    with lane_path(epw):
        out = _compare_rollout(o, h, envs, actions, mode)
        assert _route() == 'lane_rollout'
    assert out['reward'].sum() > 3 * N and out['self_collision'].sum() > 0 and out['edge_collision'].sum() > 0
    if mode == 'raw':
        assert out['obs'][:, :, 2].max() >= 8        # bodies of 8 and more segments were observed


@pytest.mark.parametrize('mode', ['raw', 'partial_3'])
def test_snake_longer_than_32_segments_in_the_round_5_modes(hip, mode):
    N, T, S = 8, 60, 9
    o, h = OracleBackend(seed=4), hip(seed=4)
    envs = _fresh(o, N)
    path = []
    for r in range(1, 8):
        cols = range(1, 8) if r % 2 == 1 else range(7, 0, -1)
        path += [(r, c) for c in cols]
    for i, L in enumerate([40, 33, 17, 48]):
        e = np.zeros((3, S, S), np.float32)
        for v, (y, x) in enumerate(path[:L], start=1):
            e[2, y, x] = v
        hy, hx = path[L - 1]
        e[1, hy, hx] = 1
        fy, fx = path[L]
        e[0, fy, fx] = 1
        envs[i] = e
    rng = np.random.RandomState(9)
    o.call = h.call = 3
    with lane_path(8):
        _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), mode)
        assert _route() == 'lane_rollout'


@pytest.mark.parametrize('epw,T', [(64, 1), (64, 17), (8, 7), (8, 9), (16, 130), (32, 3)])
@pytest.mark.parametrize('mode', ['raw', 'partial_3'])
def test_round_5_modes_tape_lengths_around_the_chunk(hip, mode, epw, T):
    N = 2 * epw + 5
    rng = np.random.RandomState(T + epw)
    o, h = OracleBackend(seed=6), hip(seed=6)
    envs = _fresh(o, N)
    o.call = h.call = 9
    with lane_path(epw):
        _compare_rollout(o, h, envs, rng.randint(-3, 9, size=(T, N)).astype(np.int64), mode)


def test_reference_tape_injected_through_the_round_5_modes(hip):
    """the recorded reference tapes whose observation mode is 'raw' or 'partial_3' (tests/golden), injected through
    lane_rollout_kernel<16, ., INJ>; where the fixtures hold neither mode, the 'partial_2' tape's random outcomes drive the
    same launch in those modes and the oracle (injected alike) is the checker"""
    tape = replay.load('single_s9_partial2')
    with lane_path():
        for mode in ('raw', 'partial_3'):
            replay.replay_single_rollout_vs_oracle(hip(), OracleBackend(), tape, mode)
            assert _route() == 'lane_rollout'


def _route():
    from wurm_amd import _lib
    return _lib.lib().wurm_single_last_route().decode()
