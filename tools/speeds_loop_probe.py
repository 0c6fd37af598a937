#!/usr/bin/env python3
"""experiments/speeds.py:30-38 as written — `obs, r, d, info = env.step(actions); env.reset(d['__all__'])` with the returned
dict discarded — at its shape (4096 x 36 x 36, 10 snakes, boost, respawn 'any'): microseconds and launches per iteration,
and whether the env has noticed that the caller drops what reset returns (MultiSnake._lazy_obs_mode, round 6)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--snakes', type=int, default=10)
ap.add_argument('--size', type=int, default=36)
ap.add_argument('--iters', type=int, default=60)
ap.add_argument('--keep', action='store_true', help='keep what reset returns (the A2C loops do)')
ap.add_argument('--check', action='store_true', help='env.check_consistency() after the reset (speeds.py:37)')
args = ap.parse_args()
N, K, S = args.envs, args.snakes, args.size
dev = torch.device('cuda:0')
env = MultiSnake(N, K, S, device=dev, seed=0, boost=True, respawn_mode='any')
tape = torch.randint(8, (args.iters + 12, K, N), device=dev)
keys = [f'agent_{i}' for i in range(K)]
dicts = [dict(zip(keys, tape[t].unbind(0))) for t in range(tape.shape[0])]
count = _lib.lib().wurm_launch_count
kept = None
for rep in range(3):
    for t in range(12):
        o = env.step(dicts[t])
        if args.keep:
            kept = env.reset(o[2]['__all__'])
        else:
            env.reset(o[2]['__all__'])
        if args.check:
            env.check_consistency()
    torch.cuda.synchronize()
    n0, t0 = count(), time.perf_counter()
    for t in range(12, 12 + args.iters):
        o = env.step(dicts[t])
        if args.keep:
            kept = env.reset(o[2]['__all__'])
        else:
            env.reset(o[2]['__all__'])
        if args.check:
            env.check_consistency()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / args.iters * 1e6
    print(f'{N}x{S}x{S}x{K} loop as written{" (kept)" if args.keep else ""}{" + check" if args.check else ""}: {us:.1f} us per '
          f'iteration, {(count() - n0) / args.iters:.2f} launches, lazy reset observations: {env._lazy_obs_mode}', flush=True)
