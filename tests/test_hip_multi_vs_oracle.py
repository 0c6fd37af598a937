"""GPU parity for MultiSnake with the build's own RNG (no injection): HIP kernels vs the CPU oracle, bit-exact on
state, per-agent outputs and observations, over step / reset / respawn cycles."""
import numpy as np
import pytest

from oracle import oracle as _o
from tests.backends import OracleBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


def _same(a, b, what):
    if a is None and b is None:
        return
    if a.dtype.kind == 'f':
        a32, b32 = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
        assert np.array_equal(a32.view(np.uint32), b32.view(np.uint32)), what
    else:
        assert np.array_equal(a, b), what


def _same_state(a, b, what):
    for k in a:
        _same(a[k], b[k], f'{what}: {k}')


CFGS = {
    'default': dict(boost=True, food_on_death_prob=0.5, boost_cost_prob=0.5, food_mode='only_one', food_rate=5e-4,
                    reward_on_death=-1, respawn_mode='all', colour_mode='random'),
    'train': dict(boost=True, food_on_death_prob=0.33, boost_cost_prob=0.25, food_mode='random_rate',
                  food_rate=2.5e-4, reward_on_death=-1, respawn_mode='any', colour_mode='random'),
    'dense': dict(boost=True, food_on_death_prob=0.9, boost_cost_prob=0.8, food_mode='random_rate', food_rate=2e-2,
                  reward_on_death=-2, respawn_mode='any', colour_mode='fixed'),
    # rates at which P(no food) = (1 - p)^n drops below 1e-6: rate food is drawn cell by cell, not as a Binomial count
    'flood': dict(boost=True, food_on_death_prob=0.5, boost_cost_prob=0.5, food_mode='random_rate', food_rate=0.3,
                  reward_on_death=-1, respawn_mode='any', colour_mode='random'),
    'noboost': dict(boost=False, food_on_death_prob=0.0, boost_cost_prob=0.5, food_mode='only_one', food_rate=5e-4,
                    reward_on_death=-1, respawn_mode='any', colour_mode='random'),
}


def test_workgroup_kernel_whole_views_per_wave(hip):
    """multi_step_wg_kernel deals the K agents' views to its four waves as (agent, half) items; bit 0 of
    WURM_MULTI_GROUP_VARIANT selects whole views per wave (the A/B switch of tools/multi_speeds_percall.py): same bytes"""
    from wurm_amd._lib import knobs
    with knobs(WURM_MULTI_GROUP_VARIANT=1):
        test_multi_step_reset_loop(hip, 4, 10, 36, 25, 'full', 'train')


@pytest.mark.parametrize('N,K,S,T,mode,cfg', [
    (24, 2, 12, 120, 'full', 'default'),
    (10, 4, 25, 80, 'full', 'default'),        # BASELINE cfg4 shape
    (10, 4, 25, 120, 'partial_5', 'train'),    # tests/test_multi_snake_env.py:100-104 dynamics
    (8, 6, 14, 100, 'partial_2', 'dense'),
    (9, 3, 10, 100, 'full', 'noboost'),
    (3, 10, 36, 40, 'partial_3', 'train'),     # experiments/speeds.py shape (10 agents, 36x36)
    (4, 10, 36, 40, 'full', 'train'),          # ... with its own observation: one env per WORKGROUP (multi_step_wg_kernel)
    (5, 4, 48, 30, 'full', 'default'),         # 18 KB of grids per env: the workgroup kernel as well
    (3, 3, 64, 24, 'full', 'dense'),
    (5, 1, 9, 60, 'full', 'dense'),
    (7, 1, 5, 60, 'full', 'default'),          # the smallest grid the reference can populate (seed rows 2..S-3)
    (7, 1, 6, 60, 'partial_1', 'train'),
    (8, 3, 12, 60, 'partial_2', 'flood'),      # rate food cell by cell (the Binomial recurrence's domain ends at P0 = 1e-6)
    (8, 2, 9, 40, 'full', 'flood'),
])
def test_multi_step_reset_loop(hip, N, K, S, T, mode, cfg):
    cfg = CFGS[cfg]
    rng = np.random.RandomState(100 + K * S)
    o, h = OracleBackend(seed=77, env_offset=500), hip(seed=77, env_offset=500)
    so, sh = _o.multi_empty_state(N, K, S), _o.multi_empty_state(N, K, S)
    fixed = cfg['colour_mode'] == 'fixed'
    so['colours'][...] = o.multi_colours(N, K, fixed, call=0)
    sh['colours'][...] = h.multi_colours(N, K, fixed, call=0)
    _same(so['colours'], sh['colours'], 'initial colours')
    o._next(); h._next()
    assert o.multi_reset(so, np.ones(N), cfg) == 0
    assert h.multi_reset(sh, np.ones(N), cfg) == 0
    _same_state(so, sh, 'fresh envs')
    assert (o.multi_check(so) == 0).all()
    deaths = 0
    for t in range(T):
        a = rng.randint(0, 8, size=(K, N)).astype(np.int64)
        ro, rh = o.multi_step(so, a, cfg, mode), h.multi_step(sh, a, cfg, mode)
        _same_state(so, sh, f'state t={t}')
        for k in ro:
            _same(ro[k], rh[k], f'{k} t={t}')
        deaths += int(so['dones'].sum())
        o.multi_reset(so, ro['all_done'], cfg, mode=mode)
        h.multi_reset(sh, rh['all_done'], cfg, mode=mode)
        _same_state(so, sh, f'reset state t={t}')
        _same(o.last_reset_obs, h.last_reset_obs, f'reset obs t={t}')
        _same(o.multi_check(so), h.multi_check(sh), f'consistency mask t={t}')
        assert (o.multi_check(so) == 0).all(), f'inconsistent after reset at t={t}'
    assert deaths > 0


def test_orientations(hip):
    rng = np.random.RandomState(5)
    o, h = OracleBackend(), hip()
    for S in (9, 12, 25):
        N = 40
        envs = np.zeros((N, 3, S, S), np.float32)
        o.single_reset(envs, np.ones(N), 'none')
        for t in range(10):
            o.single_step(envs, rng.randint(0, 4, size=N).astype(np.int64), 'none')
        _same(o.orientations(envs), h.orientations(envs), f'orientations S={S}')


def test_observe_entry_point(hip):
    o, h = OracleBackend(seed=3), hip(seed=3)
    N, K, S = 6, 3, 16
    cfg = CFGS['train']
    st = _o.multi_empty_state(N, K, S)
    st['colours'][...] = o.multi_colours(N, K, False, call=0)
    o._next()
    o.multi_reset(st, np.ones(N), cfg)
    for mode in ('full', 'partial_4'):
        _same(o.multi_observe(st, mode), h.multi_observe(st, mode), mode)


@pytest.mark.parametrize('N,K,S,T,mode,cfg', [
    (16, 4, 25, 60, 'full', 'default'),
    (12, 4, 25, 90, 'partial_5', 'train'),
    (10, 6, 14, 80, 'partial_2', 'dense'),
    (9, 2, 12, 150, 'full', 'noboost'),
    (4, 10, 36, 30, 'full', 'train'),
    (8, 3, 12, 60, 'partial_2', 'flood'),
])
def test_multi_rollout_equals_loop(hip, N, K, S, T, mode, cfg):
    """wurm_multi_rollout == T x (step; reset(__all__)) of the oracle, bit for bit, with the build's RNG."""
    cfg = CFGS[cfg]
    rng = np.random.RandomState(7 * K + S)
    o, h = OracleBackend(seed=21, env_offset=64), hip(seed=21, env_offset=64)
    so, sh = _o.multi_empty_state(N, K, S), _o.multi_empty_state(N, K, S)
    fixed = cfg['colour_mode'] == 'fixed'
    so['colours'][...] = o.multi_colours(N, K, fixed, call=0)
    sh['colours'][...] = so['colours']
    o._next(); h._next()
    o.multi_reset(so, np.ones(N), cfg)
    h.multi_reset(sh, np.ones(N), cfg)
    _same_state(so, sh, 'fresh envs')
    actions = rng.randint(0, 8, size=(T, K, N)).astype(np.int64)
    ro, rh = o.multi_rollout(so, actions, cfg, mode), h.multi_rollout(sh, actions, cfg, mode)
    for k in ro:
        _same(ro[k], rh[k], k)
    _same_state(so, sh, 'final state')
    assert ro['dones'].sum() > 0


@pytest.mark.parametrize('N,K,S,T,mode,cfg', [
    (6, 32, 36, 12, 'full', 'train'),       # 90 KB of grids per env: the kernels are opted into > 64 KB of LDS
    (5, 24, 36, 12, 'partial_3', 'train'),  # 75 KB with the env image of partial_n
    (4, 12, 64, 10, 'full', 'default'),     # 12 snakes on 64 x 64: 112 KB
])
def test_envs_larger_than_64kb_of_lds(hip, N, K, S, T, mode, cfg):
    """One env's grids must fit a CU's LDS: up to 64 KB as before, up to 160 KB with the kernels' opt-in."""
    test_multi_step_reset_loop(hip, N, K, S, T, mode, cfg)


def test_env_beyond_160kb_is_unsupported(hip):
    from wurm_amd.envs import MultiSnake
    with pytest.raises(NotImplementedError):
        MultiSnake(2, 40, 64, device='cuda:0', seed=1)   # 2 x 40 x 4096 bytes of body clocks alone


def test_food_under_a_dying_snake_counts_twice_towards_max_food(hip):
    """`self.foods += food_on_death` (multi_snake.py:672) leaves a 2 where a dead body lay over food until the clamp at :692,
    and `_add_food` (:680) sums the plane in between: with max_food - 1 cells of food, one of them under the snake that dies,
    the sum is max_food and NO rate food is added (round 6's fuzz, seed 722: the kernel counted cells and added two).
    Hand-made state (the dynamics never leave food under a body), both with and without a snake that survives."""
    cfg = dict(boost=True, food_on_death_prob=1.0, boost_cost_prob=0.0, food_mode='random_rate', food_rate=0.2,
               reward_on_death=-2, respawn_mode='all', colour_mode='random')
    N, K, S = 6, 2, 12
    o, h = OracleBackend(seed=11, env_offset=4), hip(seed=11, env_offset=4)
    st = _o.multi_empty_state(N, K, S)
    st['colours'][...] = o.multi_colours(N, K, False, call=0)
    for e in range(N):
        # snake 0: along row 4, head at (4, 7) facing east; snake 1: row 8, head at (8, 10) — one step east of it is the wall
        for s, (y, x0) in enumerate(((4, 5), (8, 8))):
            a = e * K + s
            for v, x in enumerate(range(x0, x0 + 3), start=1):
                st['bodies'][a, 0, y, x] = v
            st['heads'][a, 0, y, x0 + 2] = 1
            st['orientations'][a] = 1
        # max_food = 8 K = 16: env e holds 15 - (e % 3) cells of food, one of them under the middle of snake 1
        cells = [(1, 2), (1, 9), (2, 1), (2, 6), (3, 3), (5, 1), (5, 9), (6, 2), (6, 6), (6, 10), (9, 1), (9, 4), (10, 2), (10, 7), (8, 9)]
        for (y, x) in cells[e % 3:]:
            st['foods'][e, 0, y, x] = 1
    sh = {k: v.copy() for k, v in st.items()}
    actions = np.zeros((K, N), np.int64)
    actions[0] = 1   # snake 0 keeps going east
    actions[1] = 1   # snake 1 runs into the wall and dies: food on its body cells, (8, 9) among them
    o.call = h.call = 9
    ro, rh = o.multi_step(st, actions, cfg, 'full'), h.multi_step(sh, actions, cfg, 'full')
    assert st['dones'].reshape(N, K)[:, 1].all() and not st['dones'].reshape(N, K)[:, 0].any()
    _same_state(st, sh, 'state')
    for k in ro:
        _same(ro[k], rh[k], k)
    # envs with 15 cells of food: the doubled cell makes the sum 16 -> no rate food (the same count in both of them: the cells
    # they had + the dead body's other cell); the others, one and two cells short of that, got rate food on top
    total = st['foods'].reshape(N, -1).sum(1)
    assert total[0] == total[3] == 16 and (total[[1, 2, 4, 5]] > np.array([15, 14, 15, 14])).any()
