#!/usr/bin/env python3
"""cfg4' (MultiSnake 4096 x 25 x 25 x 4, random_rate food, respawn 'any', partial_5) fused rollout, A/B within ONE process:
the shape-specialised kernel (WURM_MULTI_SHAPE_KERNELS = 1) against the generic one (0), alternating, same tapes, and a
bit-for-bit comparison of what the two leave behind (state + every output of one launch)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--snakes', type=int, default=4)
ap.add_argument('--size', type=int, default=25)
ap.add_argument('--chunk', type=int, default=16)
ap.add_argument('--mode', default='partial_5')
ap.add_argument('--rounds', type=int, default=6)
ap.add_argument('--option', default='WURM_MULTI_SHAPE_KERNELS')
args = ap.parse_args()
N, K, S, T = args.envs, args.snakes, args.size, args.chunk
dev = torch.device('cuda:0')
kw = dict(observation_mode=args.mode, food_mode='random_rate', respawn_mode='any', boost_cost_prob=0.25,
          food_on_death_prob=0.33, food_rate=2.5e-4)
acts = torch.randint(8, (8, T, K, N), device=dev)


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, dict):
        return [t for k in sorted(o) for t in flat(o[k])]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in flat(x)]
    return []


def run(opt):
    _lib.set_option(args.option, opt)
    env = MultiSnake(N, K, S, device=dev, seed=0, **kw)
    outs = []
    for i in range(3):
        outs.append(flat(env.rollout(acts[i])))
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for rep in range(5):
        ev[0].record()
        for i in range(8):
            env.rollout(acts[i])
        ev[1].record()
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) / 8)
    state = [env.foods.clone(), env.heads.clone(), env.bodies.clone(), env.dones.clone(), env.orientations.clone()]
    return sorted(ts)[len(ts) // 2], min(ts), outs, state


res = {0: [], 1: []}
keep = {}
for r in range(args.rounds):
    for o in (0, 1):
        ms, mn, outs, state = run(o)
        res[o].append(ms)
        keep[o] = (outs, state)
        print(f'round {r} option {o}: {ms:.4f} ms (min {mn:.4f})  {N * T / ms * 1e3:.4g} env-steps/s', flush=True)
same = all(torch.equal(a, b) for x, y in zip(keep[0][0], keep[1][0]) for a, b in zip(x, y)) and \
    all(torch.equal(a, b) for a, b in zip(keep[0][1], keep[1][1]))
for o in (0, 1):
    v = sorted(res[o])
    print(f'option {o}: median {v[len(v) // 2]:.4f} ms, min {v[0]:.4f}, {N * T / v[len(v) // 2] * 1e3:.4g} env-steps/s')
print('outputs and final state identical:', same)
