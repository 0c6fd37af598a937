#!/usr/bin/env python3
"""Timing of the HBM-bound configurations (BASELINE configs[3], configs[4]): fused rollout and per-call loop.
usage: bench_big.py [cfg5] [cfg4] [cfg3]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import SingleSnake, MultiSnake  # noqa: E402


def timed(fn, reps, rounds=5):
    """median over `rounds` of the mean wall time of `reps` back-to-back calls (after 3 warm-up calls)"""
    for _ in range(3):
        fn()
    out = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / reps)
    return sorted(out)[len(out) // 2]


which = sys.argv[1:] or ['cfg5', 'cfg4', 'cfg3']
out = {}
if 'cfg5' in which:
    for T in (16, 64):
        env = SingleSnake(8192, 36, observation_mode='default', device='cuda', seed=0)
        acts = torch.randint(4, (4, T, 8192), device='cuda')
        it = iter(range(10 ** 9))
        dt = timed(lambda: env.rollout(acts[next(it) % 4]), 20)
        out[f'cfg5_rollout_T{T}'] = {'ms': dt * 1e3, 'env_steps_per_s': 8192 * T / dt, 'obs_TBs': 8192 * T * 15552 / dt / 1e12}
        del env, acts
    env = SingleSnake(8192, 36, observation_mode='default', device='cuda', seed=0)
    acts = torch.randint(4, (64, 8192), device='cuda')
    it = iter(range(10 ** 9))

    def pair():
        _, _, d, _ = env.step(acts[next(it) % 64])
        env.reset(d, return_observations=False)
    dt = timed(pair, 50)
    out['cfg5_percall'] = {'us': dt * 1e6, 'env_steps_per_s': 8192 / dt}
    del env, acts
if 'cfg4' in which:
    for T in (16, 64):
        env = MultiSnake(4096, 4, 25, device='cuda', seed=0)
        acts = torch.randint(8, (4, T, 4, 4096), device='cuda')
        it = iter(range(10 ** 9))
        dt = timed(lambda: env.rollout(acts[next(it) % 4]), 20)
        out[f'cfg4_rollout_T{T}'] = {'ms': dt * 1e3, 'env_steps_per_s': 4096 * T / dt, 'obs_TBs': 4096 * T * 30000 / dt / 1e12}
        del env, acts
    env = MultiSnake(4096, 4, 25, device='cuda', seed=0)
    acts = torch.randint(8, (4, 16, 4, 4096), device='cuda')
    it = iter(range(10 ** 9))
    dt = timed(lambda: env.rollout(acts[next(it) % 4], return_observations=False), 20)
    out['cfg4_rollout_T16_noobs'] = {'ms': dt * 1e3}
    del env, acts
if 'cfg3' in which:
    env = SingleSnake(65536, 9, observation_mode='partial_2', device='cuda', seed=0)
    acts = torch.randint(4, (64, 65536), device='cuda')
    it = iter(range(10 ** 9))

    def pair3():
        _, _, d, _ = env.step(acts[next(it) % 64])
        env.reset(d, return_observations=False)
    dt = timed(pair3, 50)
    out['cfg3_percall_65536'] = {'us': dt * 1e6, 'env_steps_per_s': 65536 / dt}
for k, v in out.items():
    print(k, json.dumps(v))
