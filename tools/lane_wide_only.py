#!/usr/bin/env python3
"""A few SingleSnake rollout launches of one shape (PMC / trace target): lane_wide_only.py [S] [mode] [T] [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 10
mode = sys.argv[2] if len(sys.argv) > 2 else 'partial_2'
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
N = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
dev = torch.device('cuda:0')
env = SingleSnake(num_envs=N, size=S, observation_mode=mode, device=dev, seed=1)
tape = torch.randint(4, (6, T, N), device=dev, dtype=torch.int64)
for r in range(6):
    env.rollout(tape[r])
torch.cuda.synchronize()
