#!/usr/bin/env python3
"""Fused-rollout throughput of MultiSnake (BASELINE cfg4 shape and variants); DESIGN.md §7."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

dev = torch.device('cuda:0')


def run(N, K, S, chunk, reps, obs=True, **kw):
    env = MultiSnake(N, K, S, device=dev, seed=0, **kw)
    actions = torch.randint(8, (chunk * (reps + 1), K, N), device=dev)
    env.rollout(actions[:chunk], return_observations=obs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(1, reps + 1):
        env.rollout(actions[r * chunk:(r + 1) * chunk], return_observations=obs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mode = env.observation_mode if obs else 'none'
    obs_bytes = (12 * K * S * S if mode == 'full' else 12 * K * env.observation_size ** 2) if obs else 0
    per = 8 * (1 + 2 * K) * S * S + obs_bytes + 40 * K
    eps = N * chunk * reps / dt
    return dict(N=N, K=K, S=S, mode=mode, chunk=chunk, env_steps_per_s=eps, ms_per_launch=dt / reps * 1e3,
                algorithmic_GBs=per * eps / 1e9, obs_write_GBs=obs_bytes * eps / 1e9)


train = dict(respawn_mode='any', food_mode='random_rate', boost_cost_prob=0.25, observation_mode='partial_5',
             food_on_death_prob=0.33, food_rate=2.5e-4)
print(json.dumps({'cfg4 4096x25 K=4 full': run(4096, 4, 25, 16, 8)}))
print(json.dumps({'cfg4 4096x25 K=4 no obs': run(4096, 4, 25, 16, 8, obs=False)}))
print(json.dumps({'cfg4b 4096x25 K=4 train partial_5': run(4096, 4, 25, 16, 8, **train)}))
print(json.dumps({'16384x25 K=4 full': run(16384, 4, 25, 8, 6)}))
print(json.dumps({'speeds 4096x36 K=10 any full': run(4096, 10, 36, 4, 4, respawn_mode='any')}))
