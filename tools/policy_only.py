#!/usr/bin/env python3
"""A few policy_rollout launches only (512 x 9 x 9 partial_2, 256 steps) — target for rocprofv3 --pmc passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.agents import FeedforwardAgent, pack_policy_params  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.manual_seed(0)
env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device='cuda', seed=0)
params = pack_policy_params(FeedforwardAgent(4, 2, 64, 75).to('cuda'))
state = env.reset()
for _ in range(6):
    state = env.policy_rollout(params, state, T, check=False)['state']
torch.cuda.synchronize()
print('done')
