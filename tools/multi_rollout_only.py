#!/usr/bin/env python3
"""A few MultiSnake rollout launches (cfg4: 4096 x 25 x 25, K=4; obs mode from argv: full | none) — PMC target."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'full'
N, K, S, chunk = 4096, 4, 25, 16
dev = torch.device('cuda:0')
env = MultiSnake(N, K, S, device=dev, seed=0, observation_mode='full')
actions = torch.randint(8, (chunk * 5, K, N), device=dev, dtype=torch.int64)
for r in range(5):
    env.rollout(actions[r * chunk:(r + 1) * chunk], return_observations=(mode != 'none'))
torch.cuda.synchronize()
print('done')
