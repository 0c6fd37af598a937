"""Statistical parity for the random outcomes no reference test pins: which free cell food
respawns in, where resets place snakes, the Bernoulli draws of MultiSnake.  The reference draws them uniformly
(randperm over the free cells, randint, rand < p); this build draws them from its counter-based Philox streams.
These tests check the DISTRIBUTIONS on the CPU oracle — the HIP kernels are bit-identical to the oracle in RNG mode
(tests/test_hip_vs_oracle.py, tests/test_hip_multi_vs_oracle.py), so the result carries over to the GPU path."""
import numpy as np
import pytest
from scipy import stats

from oracle import oracle as o

ALPHA = 1e-4  # per-test false-alarm rate


def test_single_food_respawn_is_uniform_over_free_interior_cells():
    S, N = 9, 20000
    env = np.zeros((1, 3, S, S), np.float32)
    for v, (y, x) in zip((1, 2, 3, 4), ((3, 3), (3, 4), (4, 4), (5, 4))):
        env[0, 2, y, x] = v
    env[0, 1, 5, 4] = 1
    env[0, 0, 6, 4] = 1  # food straight ahead: action 0 eats it every time
    envs = np.repeat(env, N, axis=0)
    _, reward, done, _, _ = o.single_step(envs, np.zeros(N, np.int64), 'none', seed=123, call=7)
    assert (reward == 1).all() and not done.any()
    cells = envs[:, 0].reshape(N, -1).argmax(axis=1)
    assert (envs[:, 0].reshape(N, -1).sum(axis=1) == 1).all()
    occupied = (envs[0, 1] + envs[0, 2]) > 0
    free = [y * S + x for y in range(1, S - 1) for x in range(1, S - 1) if not occupied[y, x]]
    counts = np.bincount(cells, minlength=S * S)
    assert counts[[c for c in range(S * S) if c not in free]].sum() == 0  # never on the snake or the border ring
    assert stats.chisquare(counts[free]).pvalue > ALPHA


def test_single_reset_positions_and_directions_are_uniform():
    S, N = 14, 30000
    envs = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(envs, np.ones(N, np.uint8), 'none', seed=5, call=3)
    body = envs[:, 2].reshape(N, -1)
    seeds = (body == 2).argmax(axis=1)
    heads = (body == 3).argmax(axis=1)
    sy, sx = seeds // S, seeds % S
    assert sy.min() == 4 and sy.max() == S - 5 and sx.min() == 4 and sx.max() == S - 5  # randint(4, S-4), :358-359
    n = S - 8
    assert stats.chisquare(np.bincount((sy - 4) * n + (sx - 4), minlength=n * n)).pvalue > ALPHA
    d = np.select([heads == seeds - S, heads == seeds + 1, heads == seeds + S, heads == seeds - 1], [0, 1, 2, 3], -1)
    assert (d >= 0).all() and stats.chisquare(np.bincount(d, minlength=4)).pvalue > ALPHA
    # independence of position and direction (contingency test)
    table = np.zeros((n * n, 4))
    np.add.at(table, ((sy - 4) * n + (sx - 4), d), 1)
    assert stats.chi2_contingency(table)[1] > ALPHA
    assert (o.single_check(envs) == 0).all()


def _multi_board(N, K, S):
    st = o.multi_empty_state(N, K, S)
    st['colours'][...] = 100
    return st


def test_multi_bernoulli_rates():
    """food-on-death, boost cost and rate food fire at their configured probabilities"""
    N, K, S = 4000, 2, 12
    # boost cost: one boosting length-5 snake per env, prob 0.3
    st = _multi_board(N, K, S)
    for e in range(N):
        for v, x in zip((1, 2, 3, 4, 5), (2, 3, 4, 5, 6)):
            st['bodies'][e * K, 0, 6, x] = v
        st['heads'][e * K, 0, 6, 6] = 1
        st['dones'][e * K + 1] = 1
    st['orientations'][::K] = 1  # heading +x
    cfg = o.multi_cfg(K, boost=True, food_on_death_prob=0.0, boost_cost_prob=0.3, food_mode='only_one')
    acts = np.zeros((K, N), np.int64)
    acts[0] = 7  # direction 3 (+x) with boost
    r = o.multi_step(st, acts, cfg, 'full', seed=9, call=1)
    paid = (r['rewards'][::K] < 0)
    assert abs(paid.mean() - 0.3) < 4 * np.sqrt(0.3 * 0.7 / N)
    assert stats.binomtest(int(paid.sum()), N, 0.3).pvalue > ALPHA

    # food on death: a length-5 snake runs into the wall; its body cells (not on the ring, not on row 1) spawn food w.p. 0.6
    st = _multi_board(N, K, S)
    for e in range(N):
        for v, x in zip((1, 2, 3, 4, 5), (6, 7, 8, 9, 10)):
            st['bodies'][e * K, 0, 6, x] = v
        st['heads'][e * K, 0, 6, 10] = 1
        st['dones'][e * K + 1] = 1
        st['foods'][e, 0, 2, 2] = 1
    st['orientations'][::K] = 1
    cfg = o.multi_cfg(K, boost=False, food_on_death_prob=0.6, food_mode='only_one')
    acts = np.zeros((K, N), np.int64)
    acts[0] = 3
    r = o.multi_step(st, acts, cfg, 'full', seed=10, call=1)
    assert st['dones'][::K].all()
    spawned = st['foods'][:, 0, 6, 7:11].sum()  # after the step the dead body covered cells x=7..10 (x=11 is the ring)
    trials = N * 4
    assert stats.binomtest(int(spawned), trials, 0.6).pvalue > ALPHA

    # rate food: every free interior cell w.p. 0.01
    st = _multi_board(N, K, S)
    st['dones'][:] = 1
    cfg = o.multi_cfg(K, boost=False, food_on_death_prob=0.0, food_mode='random_rate', food_rate=0.01)
    o.multi_step(st, np.zeros((K, N), np.int64), cfg, 'full', seed=11, call=1)
    trials = N * (S - 2) ** 2
    assert stats.binomtest(int(st['foods'].sum()), trials, 0.01).pvalue > ALPHA
    assert st['foods'][:, 0, 0].sum() == 0 and st['foods'][:, 0, :, 0].sum() == 0  # never on the ring


def test_rate_food_count_and_cells_match_independent_bernoulli_draws():
    """Round 5: rate food is drawn as COUNT ~ Binomial(n free cells, p) from one uniform + a uniformly random subset of that
    size (oracle/multi_snake.c, wurm_amd/csrc/multi_snake.hip).  The reference draws every free cell independently
    (multi_snake.py:401-408): the count must follow Binomial(n, p), every free cell must be hit at rate p, and cells must
    be hit independently of one another (pairs at rate p^2)."""
    N, K, S = 40000, 2, 9
    p = 0.06
    st = _multi_board(N, K, S)
    st['dones'][:] = 1                          # no snakes: every interior cell is free
    st['foods'][:, 0, 3, 3] = 1                 # ... but one, which already holds food
    cfg = o.multi_cfg(K, boost=False, food_on_death_prob=0.0, food_mode='random_rate', food_rate=p)
    o.multi_step(st, np.zeros((K, N), np.int64), cfg, 'full', seed=21, call=5)
    new = st['foods'][:, 0].copy()
    new[:, 3, 3] -= 1
    assert new.min() == 0 and new.max() == 1
    assert new[:, 0].sum() == 0 and new[:, -1].sum() == 0 and new[:, :, 0].sum() == 0 and new[:, :, -1].sum() == 0
    n = (S - 2) ** 2 - 1
    counts = new.reshape(N, -1).sum(axis=1).astype(int)
    # the count: chi-square against Binomial(n, p), tail pooled
    kmax = 9
    obs = np.bincount(np.minimum(counts, kmax), minlength=kmax + 1)
    exp = stats.binom.pmf(np.arange(kmax + 1), n, p)
    exp[kmax] = 1 - exp[:kmax].sum()
    assert stats.chisquare(obs, exp * N).pvalue > ALPHA
    # every free cell at rate p: uniform over the cells, and the total at rate p
    per_cell = new[:, 1:-1, 1:-1].reshape(N, -1).sum(axis=0)
    free = np.ones((S - 2) ** 2, bool)
    free[2 * (S - 2) + 2] = False               # (3, 3) in interior coordinates
    assert per_cell[~free].sum() == 0
    assert stats.chisquare(per_cell[free]).pvalue > ALPHA
    assert stats.binomtest(int(per_cell.sum()), N * n, p).pvalue > ALPHA
    # independence: two fixed cells are hit together at rate p^2
    both = (new[:, 1, 1] * new[:, 5, 6]).sum()
    assert stats.binomtest(int(both), N, p * p).pvalue > ALPHA


def test_rate_food_at_rates_far_above_the_references_draws_cell_by_cell():
    """P(no food) = (1 - p)^n below 1e-6: the recurrence would start from too few bits — cell by cell (the round-2 form)"""
    N, K, S = 3000, 2, 12
    st = _multi_board(N, K, S)
    st['dones'][:] = 1
    cfg = o.multi_cfg(K, boost=False, food_on_death_prob=0.0, food_mode='random_rate', food_rate=0.5)
    o.multi_step(st, np.zeros((K, N), np.int64), cfg, 'full', seed=12, call=1)
    trials = N * (S - 2) ** 2
    assert stats.binomtest(int(st['foods'].sum()), trials, 0.5).pvalue > ALPHA
    per_cell = st['foods'][:, 0, 1:-1, 1:-1].reshape(N, -1).sum(axis=0)
    assert stats.chisquare(per_cell).pvalue > ALPHA
    # p = 1: every free cell; p = 0: none
    for rate, want in ((1.0, (S - 2) ** 2), (0.0, 0)):
        st = _multi_board(8, K, S)
        st['dones'][:] = 1
        o.multi_step(st, np.zeros((K, 8), np.int64), o.multi_cfg(K, boost=False, food_on_death_prob=0.0,
                                                                 food_mode='random_rate', food_rate=rate), 'full', seed=1, call=1)
        assert (st['foods'].reshape(8, -1).sum(axis=1) == want).all()


def test_multi_spawn_cells_are_uniform_over_available_cells():
    N, K, S = 30000, 1, 10
    st = _multi_board(N, K, S)
    cfg = o.multi_cfg(K)
    assert o.multi_reset(st, np.ones(N), cfg, seed=3, call=2) == 0
    body = st['bodies'].reshape(N, -1)
    seeds = (body == 2).argmax(axis=1)
    sy, sx = seeds // S, seeds % S
    assert sy.min() == 2 and sy.max() == S - 3 and sx.min() == 2 and sx.max() == S - 3  # >= 2 from the border (:938-941)
    n = S - 4
    assert stats.chisquare(np.bincount((sy - 2) * n + (sx - 2), minlength=n * n)).pvalue > ALPHA
    assert stats.chisquare(np.bincount(st['orientations'], minlength=4)).pvalue > ALPHA
    assert (o.multi_check(st) == 0).all()


def test_streams_are_independent_of_batch_composition():
    """An env's draws depend only on (seed, global env id, call): stepping it alone or inside a batch is identical."""
    S, N = 9, 64
    envs = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(envs, np.ones(N, np.uint8), 'none', seed=77, call=0)
    solo = np.zeros((1, 3, S, S), np.float32)
    o.single_reset(solo, np.ones(1, np.uint8), 'none', seed=77, call=0, env_offset=41)
    assert np.array_equal(solo[0], envs[41])
