"""The caller-side harnesses run end to end on the GPU: the A2C loop of the reference's experiments/main.py:194-247
(model -> Categorical -> step -> store -> reset -> update) trains without touching the host in the step path."""
import math
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))


def test_a2c_loop_runs():
    import a2c_loop
    hist = a2c_loop.run(num_envs=256, size=9, observation='partial_2', steps=600, update_steps=5, log_interval=200,
                        lr=3e-3, verbose=False)
    assert len(hist) == 3
    for row in hist:
        assert all(math.isfinite(v) for v in row.values())
        assert 0 <= row['done_rate'] <= 1 and row['mean_length'] >= 3
        # a random policy dies at ~12 % of its steps; early training stays in that region
        assert 0.02 < row['done_rate'] < 0.4 and row['reward_rate'] > 0


def test_a2c_with_the_fused_actor_runs():
    """The same experiment with the acting half inside the env kernel (SingleSnake.policy_rollout): finite losses and
    the statistics of a policy that is at least not worse than random early on."""
    import a2c_fused_actor
    hist = a2c_fused_actor.run(num_envs=256, size=9, observation='partial_2', steps=3000, update_steps=5,
                               log_interval=1000, lr=1e-3, verbose=False)
    assert len(hist) >= 3
    for row in hist:
        assert all(math.isfinite(v) for v in row.values())
        assert 0 < row['done_rate'] < 0.4 and row['reward_rate'] > 0
