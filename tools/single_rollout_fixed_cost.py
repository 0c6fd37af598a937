#!/usr/bin/env python3
"""SingleSnake fused rollouts: launch time by steps per launch (slope: steady state per step; intercept: what a launch costs
besides — the fp32 state read, fill, drain, write-back).  cfg5 (8 192 x 36 x 36 'default') and cfg3 (65 536 x 9 x 9 'partial_2')."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

dev = torch.device('cuda:0')
for name, N, S, mode, Ts in (('cfg5', 8192, 36, 'default', (4, 8, 16, 32, 64)), ('cfg3', 65536, 9, 'partial_2', (8, 16, 32, 64, 128))):
    env = SingleSnake(N, S, observation_mode=mode, device=dev, seed=0)
    res = []
    for T in Ts:
        reps = 6
        acts = torch.randint(4, (reps + 1, T, N), device=dev, dtype=torch.int64)
        env.rollout(acts[0])
        torch.cuda.synchronize()
        best = 1e9
        for r in range(3):
            t0 = time.perf_counter()
            for i in range(1, reps + 1):
                env.rollout(acts[i])
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps)
        res.append((T, best * 1e6))
        print(f'{name} T = {T:3d}: {best * 1e6:8.1f} us per launch  {best * 1e6 / T:7.2f} us per step', flush=True)
    Ts_, us = np.array([r[0] for r in res], float), np.array([r[1] for r in res])
    slope, icpt = np.polyfit(Ts_, us, 1)
    print(f'{name}: {slope:.2f} us per step + {icpt:.1f} us per launch')
    del env
