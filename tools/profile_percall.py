#!/usr/bin/env python3
"""cProfile of the per-call Python API loop (SingleSnake 512x9x9 partial_2): where do the ~15 us per call go?"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

dev = torch.device('cuda:0')
N, T = 512, 3000
env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device=dev, seed=0)
actions = torch.randint(4, (T + 200, N), device=dev, dtype=torch.int64)
rows = list(actions.unbind(0))


def loop(lo, hi):
    for t in range(lo, hi):
        _, _, d, _ = env.step(rows[t])
        env.reset(d, return_observations=False)


loop(0, 200)
torch.cuda.synchronize()
t0 = time.perf_counter()
loop(200, 200 + T)
torch.cuda.synchronize()
print('us per batch-step (no profiler):', (time.perf_counter() - t0) / T * 1e6)
pr = cProfile.Profile()
pr.enable()
loop(200, 200 + T)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
