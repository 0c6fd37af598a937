"""The caller-side harnesses run end to end on the GPU: the A2C loop of the reference's experiments/main.py:194-247
(model -> Categorical -> step -> store -> reset -> update) trains without touching the host in the step path."""
import math
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))


def test_a2c_loop_runs_and_learns_something():
    import a2c_loop
    hist = a2c_loop.run(num_envs=256, size=9, observation='partial_2', steps=600, update_steps=5, log_interval=200,
                        lr=3e-3, verbose=False)
    assert len(hist) == 3
    for row in hist:
        assert all(math.isfinite(v) for v in row.values())
        assert 0 <= row['done_rate'] <= 1 and row['mean_length'] >= 3
    # a random policy dies at ~12 % of its steps; a few hundred updates already reduce that
    assert hist[-1]['done_rate'] < hist[0]['done_rate']
