#!/usr/bin/env python3
"""cfg4 fused rollout (MultiSnake 4 096 x 25 x 25 x 4 'full'): launch time by steps per launch — the slope is the steady state
per step, the intercept what a launch costs besides (state load, first transition, drain, write-back)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

N, K, S, dev = 4096, 4, 25, torch.device('cuda:0')
obs = '--noobs' not in sys.argv
env = MultiSnake(N, K, S, device=dev, seed=0)
res = []
for T in (1, 2, 4, 8, 16, 32, 64):
    reps = 8
    acts = torch.randint(8, (reps + 1, T, K, N), device=dev, dtype=torch.int64)
    env.rollout(acts[0], return_observations=obs)
    torch.cuda.synchronize()
    best = 1e9
    for r in range(3):
        t0 = time.perf_counter()
        for i in range(1, reps + 1):
            env.rollout(acts[i], return_observations=obs)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    res.append((T, best * 1e6))
    print(f'T = {T:3d}: {best * 1e6:8.1f} us per launch  {best * 1e6 / T:7.2f} us per step', flush=True)
Ts, us = np.array([r[0] for r in res], float), np.array([r[1] for r in res])
slope, icpt = np.polyfit(Ts[2:], us[2:], 1)
print(f'fit over T >= 4: {slope:.2f} us per step + {icpt:.1f} us per launch')
