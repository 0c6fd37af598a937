"""GPU parity, part 2: HIP kernels vs the CPU oracle on the same seeded inputs with the build's own
counter-based RNG (no injection) — every output of every step must be bit-identical — plus size-independent
properties at BASELINE.json's full sizes."""
import numpy as np
import pytest

from tests.backends import OracleBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


def _fresh_single(backend, N, S):
    envs = np.zeros((N, 3, S, S), np.float32)
    backend.single_reset(envs, np.ones(N, np.uint8), 'none')
    return envs


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _same(a, b, what):
    if a is None and b is None:
        return
    x, y = (_bits(a), _bits(b)) if a.dtype.kind == 'f' else (np.asarray(a), np.asarray(b))
    assert x.shape == y.shape, f'{what}: shape {x.shape} vs {y.shape}'
    bad = np.argwhere(x != y)
    assert len(bad) == 0, f'{what}: {len(bad)} mismatches, first at {bad[0].tolist()}: {a[tuple(bad[0])]} vs {b[tuple(bad[0])]}'


@pytest.mark.parametrize('N,S,T,mode,dtype', [
    (64, 9, 120, 'partial_2', np.int64),      # BASELINE cfg2 shape
    (33, 9, 80, 'default', np.int32),         # ragged: not a multiple of the waves per block
    (16, 12, 100, 'one_channel', np.int64),
    (8, 16, 60, 'positions', np.int64),
    (8, 23, 60, 'partial_4', np.int64),
    (6, 36, 60, 'default', np.int64),         # BASELINE cfg5 shape
    (3, 45, 30, 'raw', np.int64),
    (2, 64, 20, 'partial_1', np.int64),       # maximum supported size
])
def test_single_step_reset_loop(hip, N, S, T, mode, dtype):
    rng = np.random.RandomState(1234 + S)
    o, h = OracleBackend(seed=99, env_offset=1000), hip(seed=99, env_offset=1000)
    eo, eh = _fresh_single(o, N, S), _fresh_single(h, N, S)
    _same(eo, eh, 'fresh envs')
    assert (o.single_check(eo) == 0).all()
    for t in range(T):
        a = rng.randint(0, 4, size=N).astype(dtype)
        ao, ah = a.copy(), a.copy()
        ro, rh = o.single_step(eo, ao, mode), h.single_step(eh, ah, mode)
        _same(ao, ah, f'actions t={t}')
        _same(eo, eh, f'state t={t}')
        for x, y, w in zip(ro, rh, ('obs', 'reward', 'done', 'self_collision', 'edge_collision')):
            _same(x, y, f'{w} t={t}')
        if t % 3 != 2:
            bo, bh = o.single_reset(eo, ro[2], mode), h.single_reset(eh, rh[2], mode)
            _same(eo, eh, f'reset state t={t}')
            _same(bo, bh, f'reset obs t={t}')
        else:  # leave the done envs un-reset for one step: irregular states must agree too
            o._next()
            h._next()
    _same(o.single_check(eo), h.single_check(eh), 'consistency masks')


@pytest.mark.parametrize('N,S,T,mode', [(40, 9, 200, 'partial_2'), (5, 36, 70, 'default'),
                                        (12, 12, 130, 'one_channel')])
def test_single_rollout_equals_loop(hip, N, S, T, mode):
    """rollout(actions[T,N]) == T x (step; reset) — on the GPU and against the oracle."""
    rng = np.random.RandomState(7)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    o, h, h2 = OracleBackend(seed=5), hip(seed=5), hip(seed=5)
    eo, eh, eh2 = _fresh_single(o, N, S), _fresh_single(h, N, S), _fresh_single(h2, N, S)
    ao, ah, ah2 = actions.copy(), actions.copy(), actions.copy()
    ro = o.single_rollout(eo, ao, mode)
    rh = h.single_rollout(eh, ah, mode)
    for k in ro:
        _same(ro[k], rh[k], k)
    _same(ao, ah, 'actions')
    _same(eo, eh, 'final state')
    for t in range(T):
        obs, r, d, sc, ec = h2.single_step(eh2, ah2[t], mode)
        _same(obs, rh['obs'][t], f'loop obs t={t}')
        _same(d, rh['done'][t], f'loop done t={t}')
        h2.single_reset(eh2, d, 'none')
    _same(eh2, eh, 'loop final state')


@pytest.mark.parametrize('S,mode', [(9, 'partial_2'), (12, 'default'), (20, 'partial_3')])
def test_rollout_from_irregular_states(hip, S, mode):
    """The rollout kernel carries head / length / orientation / food as scalars only when the start state is a
    well-formed snake; otherwise it must fall back to the generic path.  Start from states that are NOT well formed:
    done envs that were stepped on without a reset (vanished heads, overlapping body values), envs with two foods,
    with no food, with two cells holding the maximum body value.  (More than one head per env is outside the
    kernels' domain)"""
    N, T = 48, 70
    rng = np.random.RandomState(21)
    o, h = OracleBackend(seed=9), hip(seed=9)
    envs = _fresh_single(o, N, S)
    for t in range(12):  # no resets: many envs end up irregular
        o.single_step(envs, rng.randint(0, 4, size=N).astype(np.int64), 'none')
    envs[0, 0, 2, 2] = 1          # a second food
    envs[1, 0] = 0                # no food at all
    envs[3, 2, 5, 5] = envs[3, 2].max()  # two cells hold the maximum body value
    assert (o.single_check(envs) != 0).sum() > 2
    eo, eh = envs.copy(), envs.copy()
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    ao, ah = actions.copy(), actions.copy()
    o.call = h.call = 100
    ro, rh = o.single_rollout(eo, ao, mode), h.single_rollout(eh, ah, mode)
    for k in ro:
        _same(ro[k], rh[k], k)
    _same(ao, ah, 'actions')
    _same(eo, eh, 'final state')


@pytest.mark.parametrize('N,S,T,mode', [(64, 9, 150, 'default'), (20, 7, 60, 'raw'), (8, 30, 50, 'positions')])
def test_gridworld_loop_and_rollout(hip, N, S, T, mode):
    rng = np.random.RandomState(3)
    start = (S // 2, S // 2)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    o, h = OracleBackend(seed=8), hip(seed=8)
    eo, eh = np.zeros((N, 2, S, S), np.float32), np.zeros((N, 2, S, S), np.float32)
    o.grid_reset(eo, np.ones(N), start, 'none')
    h.grid_reset(eh, np.ones(N), start, 'none')
    _same(eo, eh, 'fresh')
    for t in range(T // 2):
        ro, rh = o.grid_step(eo, actions[t].copy(), mode), h.grid_step(eh, actions[t].copy(), mode)
        for x, y in zip(ro, rh):
            _same(x, y, f'step t={t}')
        _same(eo, eh, f'state t={t}')
        _same(o.grid_reset(eo, ro[2], start, mode), h.grid_reset(eh, rh[2], start, mode), f'reset obs t={t}')
        _same(eo, eh, f'reset state t={t}')
    ro = o.grid_rollout(eo, actions[T // 2:].copy(), start, mode)
    rh = h.grid_rollout(eh, actions[T // 2:].copy(), start, mode)
    for k in ro:
        _same(ro[k], rh[k], k)
    _same(eo, eh, 'final')


def test_empty_batch(hip):
    h = hip()
    envs = np.zeros((0, 3, 9, 9), np.float32)
    obs, r, d, sc, ec = h.single_step(envs, np.zeros(0, np.int64), 'partial_2')
    assert obs.shape == (0, 75) and r.shape == (0,)


def test_sharding_invariance(hip):
    """Trajectories are keyed by the global env id: two shards (env_offset 0 / 24) == one batch of 48."""
    N, S, T = 48, 9, 100
    rng = np.random.RandomState(11)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    full = hip(seed=3)
    e = _fresh_single(full, N, S)
    a = actions.copy()
    r = full.single_rollout(e, a, 'partial_2')
    parts = []
    for lo in (0, 24):
        b = hip(seed=3, env_offset=lo)
        es = _fresh_single(b, 24, S)
        as_ = np.ascontiguousarray(actions[:, lo:lo + 24])
        rs = b.single_rollout(es, as_, 'partial_2')
        parts.append((es, rs))
    _same(np.concatenate([p[0] for p in parts]), e, 'sharded final state')
    for k in r:
        _same(np.concatenate([p[1][k] for p in parts], axis=1), r[k], f'sharded {k}')


def test_full_size_properties(hip):
    """BASELINE cfg2 at full size (512 x 9 x 9, partial_2) and one GPU's share of cfg3 (8192): invariants that
    do not need the oracle — every env consistent after every reset, reward == growth, done == OR of causes,
    observations take only the three legal values."""
    import torch
    from wurm_amd import _lib
    for N in (512, 8192):
        h = hip(seed=1)
        S, T = 9, 64
        dev = h.dev
        lib = h.lib
        envs = torch.zeros((N, 3, S, S), device=dev)
        ones = torch.ones(N, dtype=torch.uint8, device=dev)
        assert lib.wurm_single_reset(_lib.ptr(envs), _lib.ptr(ones), None, _lib.OBS_NONE, 0, _lib.i64(N), S,
                                     _lib.u64(1), _lib.u64(0), _lib.i64(0), None, None) == 0
        err = torch.empty(N, dtype=torch.int32, device=dev)
        g = torch.Generator().manual_seed(0)
        actions = torch.randint(4, (T, N), generator=g).to(dev)
        reward = torch.empty(N, device=dev)
        done, sc, ec = (torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(3))
        obs = torch.empty((N, 75), device=dev)
        total_reward = 0.0
        for t in range(T):
            length_before = envs[:, 2].amax(dim=(1, 2))
            assert lib.wurm_single_step(_lib.ptr(envs), _lib.ptr(actions[t]), 0, _lib.ptr(reward), _lib.ptr(done),
                                        _lib.ptr(sc), _lib.ptr(ec), _lib.ptr(obs), _lib.OBS_PARTIAL, 2,
                                        _lib.i64(N), S, _lib.u64(1), _lib.u64(1 + 2 * t), _lib.i64(0), None,
                                        None) == 0
            alive = done == 0
            length_after = envs[:, 2].amax(dim=(1, 2))
            assert torch.equal((length_after - length_before)[alive], reward[alive])
            assert torch.equal(done, sc | ec)
            assert ((obs == 0) | (obs == 1) | (obs == 127.0 / 255.0)).all()
            assert lib.wurm_single_reset(_lib.ptr(envs), _lib.ptr(done), None, _lib.OBS_NONE, 0, _lib.i64(N), S,
                                         _lib.u64(1), _lib.u64(2 + 2 * t), _lib.i64(0), None, None) == 0
            assert lib.wurm_single_check(_lib.ptr(envs), _lib.ptr(err), _lib.i64(N), S, None) == 0
            assert int(err.abs().max()) == 0, f'inconsistent env after reset at t={t}'
            total_reward += float(reward.sum())
        assert total_reward > 0


@pytest.mark.parametrize('S,mode', [(12, 'one_channel'), (14, 'partial_2')])
def test_grid_rollout_long_tape_rebases_clocks(hip, S, mode):
    """The LDS clock-grid rollout (grid_rollout.hip) keeps 16-bit expiry clocks and re-bases them between 64-step chunks
    once they pass 0xC000: a tape long enough to cross that several times must still equal the oracle bit for bit."""
    N, T = 3, 120000
    rng = np.random.RandomState(S)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    o, h = OracleBackend(seed=13), hip(seed=13)
    eo, eh = _fresh_single(o, N, S), _fresh_single(h, N, S)
    ao, ah = actions.copy(), actions.copy()
    ro, rh = o.single_rollout(eo, ao, mode), h.single_rollout(eh, ah, mode)
    _same(ao, ah, 'actions')
    for k in ro:
        _same(ro[k], rh[k], k)
    _same(eo, eh, 'final state')
