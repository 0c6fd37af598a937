#!/usr/bin/env python3
"""The per-call MultiSnake loop (`step(a); reset(d['__all__'])`) of 4 snakes on 25 x 25 with the reference's training
dynamics and partial_5 crops, by number of envs: how much of an iteration is per-env work and how much is fixed.

    python tools/multi_percall_by_envs.py [--full]     (--full: cfg4's dynamics and the 'full' observation instead)
Run under `rocprofv3 --kernel-trace --stats` for the kernel durations beside the loop times."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--full', action='store_true')
ap.add_argument('--mirror', default='default', help="'default' (None), 'off'")
args = ap.parse_args()
K, S, dev, T = 4, 25, torch.device('cuda:0'), 200
kw = {} if args.full else dict(observation_mode='partial_5', food_mode='random_rate', respawn_mode='any', boost_cost_prob=0.25,
                               food_on_death_prob=0.33, food_rate=2.5e-4)
keys = [f'agent_{i}' for i in range(K)]
for N in (256, 512, 1024, 2048, 4096, 8192):
    acts = torch.randint(8, (T + 20, K, N), device=dev, dtype=torch.int64)
    best = 1e9
    for rnd in range(3):
        env = MultiSnake(N, K, S, device=dev, seed=rnd, resident_mirror=None if args.mirror == 'default' else False, **kw)
        for t in range(T + 20):
            if t == 20:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            o = env.step(dict(zip(keys, acts[t].unbind(0))))
            env.reset(o[2]['__all__'], return_observations=False)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / T)
    print(f'{N:6d} envs: {best * 1e6:7.2f} us per iteration', flush=True)
