#!/usr/bin/env python3
"""cfg4' (the reference's multi-agent training defaults, experiments/multiagent.py:79-86 / tests/test_multi_snake_env.py:100-104:
MultiSnake 4096 x 25 x 25 x 4, random_rate food, respawn 'any', partial_5): fused rollout and per-call timings."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--snakes', type=int, default=4)
ap.add_argument('--size', type=int, default=25)
ap.add_argument('--chunk', type=int, default=16)
ap.add_argument('--mode', default='partial_5')
ap.add_argument('--percall', type=int, default=200)
ap.add_argument('--defaults', action='store_true', help="constructor defaults and 'full' observations (BASELINE configs[3]) instead of the training dynamics")
args = ap.parse_args()
N, K, S, T = args.envs, args.snakes, args.size, args.chunk
dev = torch.device('cuda:0')
kw = dict(observation_mode=args.mode, food_mode='random_rate', respawn_mode='any', boost_cost_prob=0.25,
          food_on_death_prob=0.33, food_rate=2.5e-4)
if args.defaults:
    kw = dict(observation_mode='full')
    args.mode = 'full (defaults)'
env = MultiSnake(N, K, S, device=dev, seed=0, **kw)
acts = torch.randint(8, (8, T, K, N), device=dev)
for i in range(3):
    env.rollout(acts[i])
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
best = []
for rep in range(5):
    ev[0].record()
    for i in range(8):
        env.rollout(acts[i])
    ev[1].record()
    torch.cuda.synchronize()
    best.append(ev[0].elapsed_time(ev[1]) / 8)
ms = sorted(best)[len(best) // 2]
print(f'rollout {N}x{S}x{S}x{K} {args.mode} chunk {T}: {ms:.4f} ms per launch, {N * T / ms * 1e3:.4g} env-steps/s')
if args.percall:
    env = MultiSnake(N, K, S, device=dev, seed=0, **kw)
    a = torch.randint(8, (args.percall, K, N), device=dev)
    keys = [f'agent_{i}' for i in range(K)]
    for t in range(20):
        o, r, d, info = env.step(dict(zip(keys, a[t])))
        env.reset(d['__all__'], return_observations=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.percall):
        o, r, d, info = env.step(dict(zip(keys, a[t])))
        env.reset(d['__all__'], return_observations=False)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / args.percall * 1e6
    print(f'per call: {us:.1f} us per iteration, {N / us * 1e6:.4g} env-steps/s')
