"""Pins the CPU oracle against golden fixtures recorded from the real reference (tests/golden/make_golden.py):
every step's sanitised actions, state, reward, done, info and observation bits, and every reset."""
import pytest

from tests import replay
from tests.backends import OracleBackend

SINGLE = ['single_s9_partial2', 'single_s12_default', 'single_s12_one_channel', 'single_s10_raw',
          'single_s12_positions', 'single_s11_partial3_i32', 'single_s36_default', 'single_s12_lazyreset']
GRID = ['grid_s9_default', 'grid_s7_raw']


@pytest.mark.parametrize('name', SINGLE)
def test_single_snake_matches_reference(name):
    replay.replay_single(OracleBackend(), replay.load(name))


@pytest.mark.parametrize('name', [n for n in SINGLE if n != 'single_s12_lazyreset'])
def test_single_snake_rollout_matches_reference(name):
    replay.replay_single_rollout(OracleBackend(), replay.load(name))


@pytest.mark.parametrize('name', GRID)
def test_gridworld_matches_reference(name):
    replay.replay_grid(OracleBackend(), replay.load(name))


@pytest.mark.parametrize('name', GRID)
def test_gridworld_rollout_matches_reference(name):
    replay.replay_grid_rollout(OracleBackend(), replay.load(name))


MULTI = ['multi_k2_s12_default', 'multi_k4_s25_default', 'multi_k4_s25_train', 'multi_k3_s14_noboost',
         'multi_k6_s10_crowded']


@pytest.mark.parametrize('name', MULTI)
def test_multi_snake_matches_reference(name):
    replay.replay_multi(OracleBackend(), replay.load_multi(name))


RL = ['a2c_nstep_t40_n64', 'a2c_nstep_t5_n512', 'a2c_gae_t40_n64', 'a2c_gae_t20_n128', 'a2c_nstep_norm_t30_n32']


@pytest.mark.parametrize('name', RL)
def test_a2c_returns_match_reference(name):
    """oracle/rl.c vs the returns the real wurm.rl.A2C.loss built (tests/golden/make_golden_rl.py): bit-exact."""
    import numpy as np
    from oracle import oracle
    z = replay.load(name)
    T, N, gae, norm = (int(v) for v in z['meta'])
    got = oracle.a2c_returns(z['bootstrap'], z['rewards'].reshape(T, N), z['values'].reshape(T, N),
                             z['dones'].reshape(T, N), float(z['gamma']), bool(gae), float(z['gae_lambda']))
    want = z['returns'].reshape(T, N)
    if norm:  # the fixture holds the normalised returns (a2c.py:68-69): apply the same two reductions
        import torch
        t = torch.from_numpy(got)
        got = ((t - t.mean()) / (t.std() + 1e-8)).numpy()
        assert np.allclose(got, want, rtol=0, atol=1e-6)
    else:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize('name', MULTI)
def test_multi_snake_rollout_loop_matches_reference(name):
    """the oracle-side definition of a MultiSnake rollout (loop of step + reset) against the reference's tape"""
    replay.replay_multi_rollout(OracleBackend(), replay.load_multi(name))
