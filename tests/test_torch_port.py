"""The torch-op restatement used as bench.py's CPU baseline (oracle/torch_port.py) computes the reference's function:
the random outcomes it draws are replayed into the scalar C oracle (pinned to the reference by tests/golden) and every
state, sanitised action, reward, done flag and observation must be bit-equal on every step."""
import numpy as np
import pytest
import torch

from oracle import oracle
from oracle.torch_port import TorchSingleSnake


@pytest.mark.parametrize('N,S,mode', [(64, 9, 'partial_2'), (33, 12, 'partial_3'), (16, 20, 'partial_1')])
def test_torch_port_equals_oracle(N, S, mode):
    env = TorchSingleSnake(N, S, mode, seed=3)
    ref = np.zeros((N, 3, S, S), np.float32)
    oracle.single_reset(ref, np.ones(N, np.uint8), 'none', inject_reset=env.last_reset.numpy())
    assert np.array_equal(ref, env.envs.numpy())
    g = torch.Generator().manual_seed(4)
    resets = 0
    for t in range(150):
        a = torch.randint(4, (N,), generator=g)
        a_ref = a.numpy().copy()
        obs, r, d, info = env.step(a)
        o2, r2, d2, sc2, ec2 = oracle.single_step(ref, a_ref, mode, inject_food=env.last_food.numpy())
        assert np.array_equal(a.numpy(), a_ref), t
        assert np.array_equal(env.envs.numpy(), ref), t
        assert np.array_equal(obs.numpy().view(np.uint32), o2.view(np.uint32)), t
        assert np.array_equal(r.numpy()[:, 0], r2) and np.array_equal(d.numpy()[:, 0].astype(np.uint8), d2)
        assert np.array_equal(info['self_collision'].numpy().astype(np.uint8), sc2)
        assert np.array_equal(info['edge_collision'].numpy().astype(np.uint8), ec2)
        ob = env.reset(d)
        ob2 = oracle.single_reset(ref, d2, mode, inject_reset=env.last_reset.numpy())
        assert np.array_equal(env.envs.numpy(), ref), t
        assert np.array_equal(ob.numpy().view(np.uint32), ob2.view(np.uint32)), t
        resets += int(d2.sum())
    assert resets > 0
