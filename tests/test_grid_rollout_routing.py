"""Which SingleSnake rollouts the LDS clock-grid kernel takes BY DEFAULT (wurm_amd/csrc/grid_rollout.hip:
grid_rollout_eligible): 'default' / 'raw' from 14 x 14 on, crops / positions / none from 18 x 18, 'one_channel' from 26 x 26 —
below those sizes the one-env-per-wave kernels it replaced are faster (tools/grid_vs_generic_probe.py,
profiles/r06_grid_vs_generic.txt).  The rest of the suite runs with WURM_GRID_ROLLOUT_MIN_SIZE = 12 (tests/conftest.py: the
kernel stays covered at every size it serves); here the option is -1, the shipped value: the rows of the table, and the
results of whichever kernel takes a row against the oracle.
Loop being matched: /root/reference tests/test_single_snake_env.py:24-31 over wurm/envs/single_snake.py:197-342."""
import numpy as np
import pytest

from tests.backends import OracleBackend
from tests.test_lane_rollout import _compare_rollout, _fresh, _route

pytestmark = pytest.mark.gpu

ROWS = [
    # (S, N, mode, expected row with the shipped thresholds)
    (12, 40, 'default', 'generic'), (13, 40, 'raw', 'generic'), (14, 40, 'default', 'grid_rollout'), (14, 30, 'raw', 'grid_rollout'),
    (12, 40, 'partial_2', 'generic'), (17, 30, 'partial_3', 'generic'), (18, 30, 'partial_2', 'grid_rollout'),
    (17, 30, 'none', 'generic'), (18, 30, 'none', 'grid_rollout'), (17, 30, 'positions', 'generic'), (20, 20, 'positions', 'grid_rollout'),
    (16, 30, 'one_channel', 'generic'), (25, 12, 'one_channel', 'generic'), (26, 12, 'one_channel', 'grid_rollout'),
    (36, 9, 'default', 'grid_rollout'), (36, 9, 'one_channel', 'grid_rollout'),
]


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


@pytest.mark.parametrize('S,N,mode,row', ROWS)
def test_shipped_thresholds(hip, S, N, mode, row):
    from wurm_amd._lib import knobs
    rng = np.random.RandomState(S * 100 + N)
    o, h = OracleBackend(seed=3, env_offset=9), hip(seed=3, env_offset=9)
    envs = _fresh(o, N, S)
    o.call = h.call = 5
    with knobs(WURM_GRID_ROLLOUT_MIN_SIZE=-1):
        _compare_rollout(o, h, envs, rng.randint(0, 4, size=(70, N)).astype(np.int64), mode)
        assert _route() == row


def test_the_option_forces_one_threshold_for_every_mode(hip):
    from wurm_amd._lib import knobs
    rng = np.random.RandomState(4)
    for min_size, S, mode, row in ((12, 12, 'one_channel', 'grid_rollout'), (30, 20, 'default', 'generic'), (20, 20, 'one_channel', 'grid_rollout')):
        o, h = OracleBackend(seed=7), hip(seed=7)
        envs = _fresh(o, 21, S)
        o.call = h.call = 1
        with knobs(WURM_GRID_ROLLOUT_MIN_SIZE=min_size):
            _compare_rollout(o, h, envs, rng.randint(0, 4, size=(40, 21)).astype(np.int64), mode)
            assert _route() == row


def test_class_rollout_below_the_threshold_writes_a_lazy_mirror_out(hip):
    """SingleSnake.rollout on a size the clock-grid rollout does not take by default, alternating with per-call steps that keep
    the clock-grid mirror: the entry point writes the mirror out, runs on the planes and reports it stale — against the same
    object without a mirror"""
    import torch
    from wurm_amd._lib import knobs
    from wurm_amd.envs import SingleSnake
    dev = torch.device('cuda:0')
    N, S, mode = 40, 12, 'one_channel'
    g = torch.Generator().manual_seed(3)
    plan = [torch.randint(4, (5, N), generator=g).to(dev) for _ in range(4)]
    outs = []
    with knobs(WURM_GRID_ROLLOUT_MIN_SIZE=-1, WURM_GRID_STEP_MIN_CELLS=0):
        for policy in (True, False):
            env = SingleSnake(num_envs=N, size=S, observation_mode=mode, device=dev, seed=11, resident_mirror=policy)
            rec = []
            for tape in plan:
                for t in range(2):
                    o, r, d, _ = env.step(tape[t].clone())
                    env.reset(d, return_observations=False)
                    rec += [o.clone(), r.clone(), d.clone()]
                out = env.rollout(tape.clone())
                assert _route() == 'generic'
                rec += [out['observations'].clone(), out['rewards'].clone(), out['dones'].clone()]
            rec.append(env.envs.clone())
            outs.append(rec)
    for i, (x, y) in enumerate(zip(*outs)):
        assert torch.equal(x, y), f'record {i}'
