#!/usr/bin/env python3
"""A few rollout launches only (cfg2: 512 x 9 x 9 partial_2, chunk 256) — target for rocprofv3 --pmc passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 256
with_obs = (sys.argv[3] != 'noobs') if len(sys.argv) > 3 else True
dev = torch.device('cuda:0')
env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device=dev, seed=0)
actions = torch.randint(4, (chunk * 6, N), device=dev, dtype=torch.int64)
for c in range(0, chunk * 6, chunk):
    env.rollout(actions[c:c + chunk], return_observations=with_obs)
torch.cuda.synchronize()
print('done')
