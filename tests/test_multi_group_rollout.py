"""GPU parity of multi_rollout_group_kernel (wurm_amd/csrc/multi_snake.hip, round 4): 'full' observations of at most 5
snakes with G consecutive envs per workgroup, the writer waves turning per-agent class codes into one linear run per
agent.  The kernel is the default from 2 048 envs on; here it is forced for small batches (WURM_MULTI_GROUP_MIN_ENVS = 0)
in every compiled shape (WURM_MULTI_GROUP_SHAPE = 1000 G + 100 W + 10 EPS + waves per SIMD: G envs per workgroup, W writer
waves, EPS envs per stepper wave), against
  (a) the oracle with the build's RNG: every output of every step and the final state, ragged N (not a multiple of G),
      K = 1 .. 5, sizes whose planes start on every 16-byte phase, T across the 64-step action chunks;
  (b) the fixtures recorded from the real reference (outcomes injected), through the rollout entry point;
  (c) hand-made states in which snakes overlap and heads share a cell (the paint order of _observe_agent :268-281)."""
import numpy as np
import pytest

from oracle import oracle as _o
from tests import replay
from tests.backends import OracleBackend
from tests.test_hip_multi_vs_oracle import CFGS, _same, _same_state
from wurm_amd._lib import knobs

pytestmark = pytest.mark.gpu

SHAPES = [8215, 8416, 4414, 8424]


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


def _fresh(o, h, N, K, S, cfg):
    so, sh = _o.multi_empty_state(N, K, S), _o.multi_empty_state(N, K, S)
    so['colours'][...] = o.multi_colours(N, K, cfg['colour_mode'] == 'fixed', call=0)
    sh['colours'][...] = so['colours']
    o._next(); h._next()
    o.multi_reset(so, np.ones(N), cfg)
    h.multi_reset(sh, np.ones(N), cfg)
    _same_state(so, sh, 'fresh envs')
    return so, sh


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('N,K,S,T,cfg', [
    (19, 4, 25, 70, 'default'),     # BASELINE cfg4 shape, ragged against G = 4 and G = 8, T across the 64-step action chunk
    (8, 4, 25, 20, 'train'),        # respawn 'any', random_rate food
    (5, 2, 12, 130, 'noboost'),
    (13, 5, 10, 60, 'dense'),       # the most snakes a 16-bit class word holds
    (9, 1, 9, 40, 'dense'),
    (6, 3, 14, 33, 'default'),
    (3, 4, 27, 24, 'default'),      # 729 floats per plane: planes start on another 16-byte phase than 625
    (3, 4, 26, 24, 'train'),        # 676: every plane on the same phase
])
def test_group_rollout_equals_the_oracle(hip, shape, N, K, S, T, cfg):
    cfg = CFGS[cfg]
    rng = np.random.RandomState(11 * K + S + shape)
    o, h = OracleBackend(seed=31, env_offset=1000), hip(seed=31, env_offset=1000)
    so, sh = _fresh(o, h, N, K, S, cfg)
    actions = rng.randint(0, 8, size=(T, K, N)).astype(np.int64)
    ro = o.multi_rollout(so, actions, cfg, 'full')
    with knobs(WURM_MULTI_GROUP_MIN_ENVS=0, WURM_MULTI_GROUP_SHAPE=shape):
        rh = h.multi_rollout(sh, actions, cfg, 'full')
    for k in ro:
        _same(ro[k], rh[k], k)
    _same_state(so, sh, 'final state')
    assert ro['dones'].sum() > 0


def test_group_rollout_is_taken_and_equals_the_two_wave_kernel(hip):
    """the same launch through the older kernel (threshold out of reach) and through the grouped one: identical outputs;
    the launch counter says one kernel each"""
    from wurm_amd import _lib
    cfg = CFGS['default']
    N, K, S, T = 37, 4, 25, 40
    rng = np.random.RandomState(5)
    actions = rng.randint(0, 8, size=(T, K, N)).astype(np.int64)
    outs = []
    for min_envs in (1 << 40, 0):
        o, h = OracleBackend(seed=9), hip(seed=9)
        so, sh = _fresh(o, h, N, K, S, cfg)
        with knobs(WURM_MULTI_GROUP_MIN_ENVS=min_envs):
            n0 = _lib.lib().wurm_launch_count()
            outs.append((h.multi_rollout(sh, actions, cfg, 'full'), sh))
            assert _lib.lib().wurm_launch_count() - n0 == 1
    (ra, sa), (rb, sb) = outs
    for k in ra:
        _same(ra[k], rb[k], k)
    _same_state(sa, sb, 'final state')


@pytest.mark.parametrize('name', ['multi_k4_s25_default', 'multi_k2_s12_default', 'multi_k3_s14_noboost'])
def test_group_rollout_replays_the_reference(hip, name):
    """outcomes recorded from the real reference, injected (tests/golden/make_golden_multi.py) — 'full' fixtures only"""
    fx = replay.load_multi(name)
    if fx['mode'] != 'full':
        pytest.skip('fixture does not use the full observation')
    for shape in (8416, 4414):
        with knobs(WURM_MULTI_GROUP_MIN_ENVS=0, WURM_MULTI_GROUP_SHAPE=shape):
            replay.replay_multi_rollout(hip(), fx)


def test_paint_order_on_hand_made_states(hip):
    """two bodies on one cell, two heads on one cell, a head on somebody else's body, food under a snake, snakes on the
    border: one step of the rollout on such a state shows the observation of the stepped state — compared with the oracle,
    which paints food, own body, own head, other bodies, other heads, border in the reference's order"""
    cfg = dict(CFGS['noboost'], food_on_death_prob=0.0)
    N, K, S = 6, 3, 10
    o, h = OracleBackend(seed=4), hip(seed=4)
    so, sh = _fresh(o, h, N, K, S, cfg)
    C = S * S
    for st in (so, sh):
        bodies = st['bodies'].reshape(N, K, S, S)
        heads = st['heads'].reshape(N, K, S, S)
        foods = st['foods'].reshape(N, S, S)
        # env 0: snake 1's body laid over snake 0's cells
        bodies[0, 1] = np.maximum(bodies[0, 1], bodies[0, 0])
        # env 1: food under every cell of snake 2
        foods[1][bodies[1, 2] > 0] = 1
        # env 2: snake 1 takes snake 0's shape entirely (heads share a cell)
        bodies[2, 1] = bodies[2, 0]
        heads[2, 1] = heads[2, 0]
    st0 = {k: v.copy() for k, v in so.items()}
    actions = np.zeros((1, K, N), np.int64)
    ro = o.multi_rollout(so, actions, cfg, 'full')
    with knobs(WURM_MULTI_GROUP_MIN_ENVS=0):
        rh = h.multi_rollout(sh, actions, cfg, 'full')
    for k in ro:
        _same(ro[k], rh[k], k)
    _same_state(so, sh, 'final state')
    del st0


@pytest.mark.parametrize('wpb', ['-1', '8'])   # automatic (every wave its own env's views) / eight envs per workgroup, shared runs
def test_per_call_parity_suites_with_the_grouped_writer(wpb):
    """multi_step_kernel with `grp_emit` ('full' observations through class codes and the colour table, per wave or by the
    workgroup's waves together): the MultiSnake per-call suites — oracle loops, fused step / reset, resident mirror,
    reference fixtures, KATs — in a child process with WURM_MULTI_GROUP_MIN_ENVS=0, which turns it on for every batch size"""
    import os
    import subprocess
    import sys
    if os.environ.get('WURM_MULTI_GROUP_MIN_ENVS') == '0':
        pytest.skip('already inside the forced run')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WURM_MULTI_GROUP_MIN_ENVS='0', WURM_MULTI_GROUP_STEP_WPB=wpb)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-m', 'gpu', '-x', 'tests/test_hip_multi_vs_oracle.py',
                        'tests/test_hip_multi_fused.py', 'tests/test_multi_resident.py', 'tests/test_kat_multi_snake.py',
                        'tests/test_hip_golden.py', '-k', 'not larger_than_64kb'],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize('shape', [5014, 4514, 3014])   # 1000 G + 100 W + 10 EPS + waves per SIMD: G 4 W 10, G 4 W 5, G 2 W 10
@pytest.mark.parametrize('N,K,S,T,cfg', [
    (9, 10, 36, 30, 'train'),       # experiments/speeds.py shape: 10 snakes on 36 x 36, respawn 'any'
    (7, 6, 14, 70, 'dense'),
    (5, 8, 25, 40, 'default'),
    (3, 10, 12, 50, 'dense'),       # crowded
])
def test_wide_group_rollout_six_to_ten_snakes(hip, shape, N, K, S, T, cfg):
    """6 .. 10 snakes: 32-bit class words, one code buffer, two barriers per step (multi_rollout_group_kernel<.., WIDE>)"""
    cfg = CFGS[cfg]
    rng = np.random.RandomState(13 * K + S + shape)
    o, h = OracleBackend(seed=37, env_offset=2000), hip(seed=37, env_offset=2000)
    so, sh = _fresh(o, h, N, K, S, cfg)
    actions = rng.randint(0, 8, size=(T, K, N)).astype(np.int64)
    ro = o.multi_rollout(so, actions, cfg, 'full')
    with knobs(WURM_MULTI_GROUP_MIN_ENVS=0, WURM_MULTI_GROUP_SHAPE=shape):
        rh = h.multi_rollout(sh, actions, cfg, 'full')
    for k in ro:
        _same(ro[k], rh[k], k)
    _same_state(so, sh, 'final state')
