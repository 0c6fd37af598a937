"""The per-call LDS clock-grid kernel (grid_step_kernel, grid_rollout.hip) normally takes only large launches
(num_envs * size^2 >= 2^20 cells: tests/test_full_size_parity.py reaches it at 8192 x 36 x 36).  Here the per-call parity
tests — oracle comparisons on seeded tapes incl. un-reset done envs (irregular states -> flagged -> generic kernel), the
reference's recorded tapes with injected outcomes, both groupings of the one-launch iteration, the class-level call
patterns, the reference KATs — are re-run in a child process with WURM_GRID_STEP_MIN_CELLS=0, which routes every
SingleSnake launch of size >= 12 through that kernel."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_per_call_parity_suite_on_the_grid_step_kernel():
    env = dict(os.environ, WURM_GRID_STEP_MIN_CELLS='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-m', 'gpu', '-x',
                        'tests/test_hip_vs_oracle.py', 'tests/test_hip_golden.py', 'tests/test_hip_fused_step.py',
                        'tests/test_kat_single_snake.py', 'tests/test_rl_gpu.py'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
