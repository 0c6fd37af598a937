"""Live cross-check of the CPU oracle against the REAL reference on random configurations — beyond the stored
fixtures.  Runs only where the reference tree is mounted (/root/reference: the build container); skipped everywhere
else (the GPU box never sees the reference).  The recorder of tests/golden/ drives the reference, captures its random
outcomes, and the tape is replayed into the oracle with those outcomes injected: bit-equality on every output."""
import os
import sys

import numpy as np
import pytest

REF = os.environ.get('WURM_REFERENCE_ROOT', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'wurm')), reason='reference tree not mounted')

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def rec():
    os.environ.setdefault('MPLBACKEND', 'Agg')
    sys.dont_write_bytecode = True
    sys.path.insert(0, GOLD)
    import make_golden
    import make_golden_multi
    return make_golden, make_golden_multi


def _multi_fx(out):
    """in-memory record -> the dict replay.load_multi() would build from the .npz"""
    N, K, S, T = (int(v) for v in out['meta'][:4])
    fx = dict(out)
    for k in ('death_a', 'death_b', 'rate'):
        fx['inj_' + k] = np.stack([np.unpackbits(out['inj_' + k][t])[:N * S * S].reshape(N, S, S) for t in range(T)])
    fx['cfg_dict'] = eval(str(out['cfg']), {'__builtins__': {}}, {})
    return fx


@pytest.mark.parametrize('seed', range(6))
def test_single_snake_random_config(rec, seed):
    from tests import replay
    from tests.backends import OracleBackend
    rng = np.random.RandomState(500 + seed)
    S = int(rng.choice([9, 10, 12, 15, 20]))
    mode = ['default', 'raw', 'one_channel', 'positions', f'partial_{rng.randint(1, 5)}'][rng.randint(5)]
    reset_every = 1 if mode.startswith('partial') else int(rng.choice([1, 1, 3]))
    fx = rec[0].record_single(None, N=int(rng.randint(2, 24)), S=S, T=int(rng.randint(20, 70)), mode=mode,
                              seed=1000 + seed, reset_every=reset_every)
    replay.replay_single(OracleBackend(), fx)
    if reset_every == 1:
        replay.replay_single_rollout(OracleBackend(), fx)


@pytest.mark.parametrize('seed', range(2))
def test_gridworld_random_config(rec, seed):
    from tests import replay
    from tests.backends import OracleBackend
    rng = np.random.RandomState(600 + seed)
    S = int(rng.choice([5, 7, 9, 13]))
    fx = rec[0].record_grid(None, N=int(rng.randint(2, 30)), S=S, T=int(rng.randint(20, 60)),
                            mode=['default', 'raw'][rng.randint(2)], seed=2000 + seed,
                            start=(int(rng.randint(1, S - 1)), int(rng.randint(1, S - 1))))
    replay.replay_grid(OracleBackend(), fx)
    replay.replay_grid_rollout(OracleBackend(), fx)


@pytest.mark.parametrize('seed', range(11))
def test_multi_snake_random_config(rec, seed):
    from tests import replay
    from tests.backends import OracleBackend
    rng = np.random.RandomState(700 + seed)
    K = int(rng.choice([1, 2, 3, 4, 6]))
    S = int(rng.choice([10, 12, 14, 18, 25]))
    if seed >= 8:  # the smallest grids the reference can populate
        K, S = 1, int(rng.choice([5, 6, 7]))
    kw = dict(boost=bool(rng.rand() < 0.8), food_on_death_prob=float(rng.choice([0.0, 0.3, 0.5, 1.0])),
              boost_cost_prob=float(rng.choice([0.0, 0.25, 0.5, 1.0])),
              food_mode=['only_one', 'random_rate'][rng.randint(2)], food_rate=float(rng.choice([5e-4, 5e-3, 3e-2])),
              respawn_mode=['all', 'any'][rng.randint(2)], agent_colours=['random', 'fixed'][rng.randint(2)],
              observation_mode=['full', f'partial_{rng.randint(1, 5)}'][rng.randint(2)],
              reward_on_death=int(rng.choice([-1, -2, 0])))
    try:
        out = rec[1].record_multi(None, N=int(rng.randint(2, 10)), K=K, S=S, T=int(rng.randint(15, 50)),
                                  seed=3000 + seed, **kw)
    except RuntimeError as e:
        if 'no available locations' in str(e):
            pytest.skip('the reference itself cannot place the snakes of this random configuration')
        raise
    fx = _multi_fx(out)
    replay.replay_multi(OracleBackend(), fx)
    replay.replay_multi_rollout(OracleBackend(), fx)
