"""cfg4 rollouts of T = 1, 2, 4, 16 steps with and without observations, a few launches each — a rocprofv3 --kernel-trace --stats
target: the kernel's own duration by T (tools/multi_rollout_fixed_cost.py times the same from the host)"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from wurm_amd.envs import MultiSnake
dev = torch.device('cuda:0')
env = MultiSnake(4096, 4, 25, device=dev, seed=0)
for obs in (False, True):
    for T in (1, 2, 4, 16):
        acts = torch.randint(8, (6, T, 4, 4096), device=dev, dtype=torch.int64)
        for i in range(6):
            env.rollout(acts[i], return_observations=obs)
        torch.cuda.synchronize()
        print('T', T, 'obs', obs, flush=True)
