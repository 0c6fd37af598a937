"""GPU parity for the lean rollout kernel (`rollout_lean_kernel`, wurm_amd/csrc/single_snake.hip): SingleSnake on
grids of at most 128 cells with a `partial_n` crop of n <= 3 or no observation, RNG mode.  It keeps the body channel
as expiry clocks, takes resets and food draws from per-chunk precomputed Philox blocks and assumes a well-formed
start state — so it is compared with the CPU oracle (which knows none of that) on every output of every step, over
all the shapes it accepts, tape lengths around the 64-step chunk size, hostile action values, and start states that
must send it to the generic path."""
import numpy as np
import pytest

from tests.backends import OracleBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


def _same(a, b, what):
    if a is None and b is None:
        return
    a, b = np.asarray(a), np.asarray(b)
    x, y = (a.view(np.uint32), b.view(np.uint32)) if a.dtype == np.float32 else (a, b)
    assert x.shape == y.shape, f'{what}: shape {x.shape} vs {y.shape}'
    bad = np.argwhere(x != y)
    assert len(bad) == 0, f'{what}: {len(bad)} mismatches, first at {bad[0].tolist()}: {a[tuple(bad[0])]} vs {b[tuple(bad[0])]}'


def _fresh(backend, N, S):
    envs = np.zeros((N, 3, S, S), np.float32)
    backend.single_reset(envs, np.ones(N, np.uint8), 'none')
    return envs


def _compare_rollout(o, h, envs, actions, mode):
    eo, eh = envs.copy(), envs.copy()
    ao, ah = actions.copy(), actions.copy()
    ro, rh = o.single_rollout(eo, ao, mode), h.single_rollout(eh, ah, mode)
    for k in ro:
        _same(ro[k], rh[k], k)
    _same(ao, ah, 'sanitised actions')
    _same(eo, eh, 'final state')
    assert (o.single_check(eo) == 0).all()
    return ro


@pytest.mark.parametrize('S', [9, 10, 11])
@pytest.mark.parametrize('mode', ['partial_0', 'partial_1', 'partial_2', 'partial_3', 'none'])
def test_every_accepted_shape(hip, S, mode):
    N, T = 37, 150
    rng = np.random.RandomState(100 * S + len(mode) + int(mode[-1]) if mode != 'none' else S)
    o, h = OracleBackend(seed=17, env_offset=5), hip(seed=17, env_offset=5)
    envs = _fresh(o, N, S)
    _same(envs, _fresh(h, N, S), 'fresh envs')
    out = _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), mode)
    assert out['done'].sum() > N          # resets happened
    assert out['reward'].sum() > 0        # and food was eaten (respawn path)


@pytest.mark.parametrize('T', [1, 2, 63, 64, 65, 127, 128, 129, 400])
def test_tape_lengths_around_the_chunk_size(hip, T):
    N, S = 19, 9
    rng = np.random.RandomState(T)
    o, h = OracleBackend(seed=T), hip(seed=T)
    envs = _fresh(o, N, S)
    o.call = h.call = 7 + T
    _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), 'partial_2')


@pytest.mark.parametrize('dtype', [np.int64, np.int32])
def test_action_values_outside_0_to_3(hip, dtype):
    """single_snake.py:221-222 only recognises a reversal for actions 0..3; anything else moves by action % 4 and the
    tape keeps the caller's value (this build wraps negative actions instead of faulting)."""
    N, S, T = 24, 9, 140
    rng = np.random.RandomState(5)
    o, h = OracleBackend(seed=3), hip(seed=3)
    envs = _fresh(o, N, S)
    actions = rng.randint(-9, 13, size=(T, N)).astype(dtype)
    actions[::7] = np.iinfo(dtype).max
    actions[3::11] = np.iinfo(dtype).min + 1
    o.call = h.call = 1
    _compare_rollout(o, h, envs, actions, 'partial_2')


def test_long_snakes_and_food_respawn(hip):
    """A scripted serpentine sweep of the interior makes snakes long: self collisions, many food draws among few free
    cells, body values far above the start length."""
    N, S, T = 16, 9, 600
    o, h = OracleBackend(seed=11), hip(seed=11)
    envs = _fresh(o, N, S)
    # boustrophedon over rows: move along x, step down at the walls (actions: 0 down(+y), 1 left, 2 up, 3 right)
    tape = []
    for t in range(T):
        phase = t % 12
        tape.append(3 if phase < 5 else 0 if phase == 5 else 1 if phase < 11 else 0)
    actions = np.repeat(np.asarray(tape, np.int64)[:, None], N, axis=1)
    rng = np.random.RandomState(2)
    noise = rng.rand(T, N) < 0.15
    actions[noise] = rng.randint(0, 4, size=int(noise.sum()))
    o.call = h.call = 1
    out = _compare_rollout(o, h, envs, actions, 'partial_3')
    assert out['reward'].sum() > 3 * N


def test_start_states_outside_the_lean_domain(hip):
    """Start states the lean loop must hand to the generic path: head on the border ring (a done env that was not
    reset), food lying on a body cell, no head, two foods — mixed with ordinary envs in the same launch."""
    N, S, T = 40, 9, 90
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
    assert (o.single_check(envs) != 0).sum() >= 3
    o.call = h.call = 50
    _compare_rollout_allow_irregular(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), 'partial_2')


def _compare_rollout_allow_irregular(o, h, envs, actions, mode):
    eo, eh = envs.copy(), envs.copy()
    ao, ah = actions.copy(), actions.copy()
    ro, rh = o.single_rollout(eo, ao, mode), h.single_rollout(eh, ah, mode)
    for k in ro:
        _same(ro[k], rh[k], k)
    _same(ao, ah, 'sanitised actions')
    _same(eo, eh, 'final state')


def test_chained_launches_equal_one_launch(hip):
    """Two launches of 96 steps == one launch of 192 steps (state handed over through HBM, call counter continued)."""
    N, S = 21, 9
    rng = np.random.RandomState(4)
    actions = rng.randint(0, 4, size=(192, N)).astype(np.int64)
    h1, h2 = hip(seed=6), hip(seed=6)
    e1, e2 = _fresh(h1, N, S), _fresh(h2, N, S)
    a1, a2 = actions.copy(), actions.copy()
    whole = h1.single_rollout(e1, a1, 'partial_2')
    first = h2.single_rollout(e2, a2[:96], 'partial_2')
    second = h2.single_rollout(e2, a2[96:], 'partial_2')
    for k in whole:
        _same(whole[k], np.concatenate([first[k], second[k]]), k)
    _same(e1, e2, 'final state')
    _same(a1, a2, 'actions')


@pytest.mark.parametrize('S,mode', [(9, 'partial_2'), (9, 'none'), (10, 'partial_1'), (12, 'partial_2')])
def test_four_waves_per_workgroup(hip, S, mode):
    """Batches above 4096 envs launch four waves (four envs) per workgroup, the last workgroup partly empty."""
    N, T = 4096 + 37, 70
    rng = np.random.RandomState(S)
    o, h = OracleBackend(seed=31, env_offset=(1 << 33) + 5), hip(seed=31, env_offset=(1 << 33) + 5)
    envs = _fresh(o, N, S)
    o.call = h.call = (1 << 40) + 3      # call counters and env ids beyond 32 bits
    _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), mode)


def test_full_size_rollout_equals_per_call_loop(hip):
    """BASELINE cfg3's per-GPU share (8192 x 9 x 9, partial_2): the fused rollout == the per-call loop, on the GPU."""
    N, S, T = 8192, 9, 64
    rng = np.random.RandomState(0)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    h1, h2 = hip(seed=12), hip(seed=12)
    e1, e2 = _fresh(h1, N, S), _fresh(h2, N, S)
    a1, a2 = actions.copy(), actions.copy()
    out = h1.single_rollout(e1, a1, 'partial_2')
    for t in range(T):
        obs, r, d, sc, ec = h2.single_step(e2, a2[t], 'partial_2')
        _same(obs, out['obs'][t], f'obs t={t}')
        _same(r, out['reward'][t], f'reward t={t}')
        _same(d, out['done'][t], f'done t={t}')
        _same(sc, out['self_collision'][t], f'self collision t={t}')
        _same(ec, out['edge_collision'][t], f'edge collision t={t}')
        h2.single_reset(e2, d, 'none')
    _same(e1, e2, 'final state')
    _same(a1, a2, 'sanitised actions')


def test_the_launch_bench_py_times_512x1024(hip):
    """exactly the launch the driver's `python bench.py` times — rollout_s9_kernel<partial>, 512 envs x 1 024 batch-steps,
    RNG mode, one wave per 64-thread workgroup — against the oracle on every output of every env-step"""
    N, S, T = 512, 9, 1024
    rng = np.random.RandomState(0)
    o, h = OracleBackend(seed=0), hip(seed=0)
    envs = _fresh(o, N, S)
    o.call = h.call = 1
    out = _compare_rollout(o, h, envs, rng.randint(0, 4, size=(T, N)).astype(np.int64), 'partial_2')
    assert out['done'].sum() > 50000
