#!/usr/bin/env python3
"""A per-call loop of one shape (PMC / trace target): percall_size_only.py [S] [mode] [N] [reset_obs: 0 | 1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 9
mode = sys.argv[2] if len(sys.argv) > 2 else 'default'
N = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
keep = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
dev = torch.device('cuda:0')
env = SingleSnake(num_envs=N, size=S, observation_mode=mode, device=dev, seed=1)
acts = torch.randint(4, (30, N), device=dev, dtype=torch.int64)
for t in range(30):
    o, r, d, _ = env.step(acts[t])
    env.reset(d, return_observations=keep)
torch.cuda.synchronize()
