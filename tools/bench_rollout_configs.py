#!/usr/bin/env python3
"""Fused-rollout throughput for SingleSnake shapes other than the headline."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

dev = torch.device('cuda:0')


def run(N, S, mode, chunk, reps, obs=True):
    env = SingleSnake(N, S, observation_mode=mode, device=dev, seed=0)
    actions = torch.randint(4, (chunk * (reps + 1), N), device=dev)
    env.rollout(actions[:chunk], return_observations=obs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(1, reps + 1):
        env.rollout(actions[r * chunk:(r + 1) * chunk], return_observations=obs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(N=N, S=S, mode=mode if obs else 'none', chunk=chunk, env_steps_per_s=N * chunk * reps / dt,
                ms_per_launch=dt / reps * 1e3)


for cfg in [(8192, 36, 'default', 16, 8, True), (8192, 36, 'default', 16, 8, False), (8192, 12, 'default', 64, 8, True),
            (65536, 9, 'partial_2', 64, 8, True), (512, 9, 'default', 256, 16, True), (8192, 9, 'default', 128, 8, True)]:
    print(json.dumps(run(*cfg)))
