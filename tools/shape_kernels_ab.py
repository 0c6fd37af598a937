#!/usr/bin/env python3
"""A/B within ONE process of a library option (default WURM_MULTI_SHAPE_KERNELS: kernels with K, S and the crop radius
compiled in against the generic ones) on the MultiSnake shapes the reference itself runs — alternating, same tapes, and a
bit-for-bit comparison of everything the two variants return and leave behind.

cases: cfg4prime (4096 x 25 x 25 x 4, random_rate food, respawn 'any', partial_5), cfg4 (constructor defaults, 'full'),
speeds (4096 x 36 x 36 x 10, boost, respawn 'any', 'full'); each as a fused rollout and per call."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cases', default='cfg4prime,cfg4,speeds')
ap.add_argument('--rounds', type=int, default=4)
ap.add_argument('--option', default='WURM_MULTI_SHAPE_KERNELS')
ap.add_argument('--values', default='0,1')
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--no-percall', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda:0')
N = args.envs
CASES = {
    'cfg4prime': (4, 25, 16, dict(observation_mode='partial_5', food_mode='random_rate', respawn_mode='any', boost_cost_prob=0.25,
                                  food_on_death_prob=0.33, food_rate=2.5e-4)),
    'cfg4': (4, 25, 16, dict()),
    'speeds': (10, 36, 4, dict(boost=True, respawn_mode='any')),
}


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, dict):
        return [t for k in sorted(o) for t in flat(o[k])]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in flat(x)]
    return []


def state_of(env):
    return [getattr(env, n).clone() for n in ('foods', 'heads', 'bodies', 'dones', 'orientations', 'agent_colours')]


def run_rollout(K, S, T, kw, acts):
    env = MultiSnake(N, K, S, device=dev, seed=0, **kw)
    outs = [[t.clone() for t in flat(env.rollout(acts[i]))] for i in range(2)]
    env.rollout(acts[2])
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
    return sorted(ts)[len(ts) // 2], outs, state_of(env)


def run_percall(K, S, kw, tape, iters=120):
    env = MultiSnake(N, K, S, device=dev, seed=0, **kw)
    keys = [f'agent_{i}' for i in range(K)]
    dicts = [dict(zip(keys, tape[t].unbind(0))) for t in range(tape.shape[0])]
    outs = []
    for t in range(20):
        o = env.step(dicts[t])
        if t < 6:
            outs.append([x.clone() for x in flat(o)])
        env.reset(o[2]['__all__'], return_observations=False)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for t in range(20, 20 + iters):
            o = env.step(dicts[t])
            env.reset(o[2]['__all__'], return_observations=False)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / iters * 1e3)
    return sorted(ts)[1], outs, state_of(env)


values = [int(v) for v in args.values.split(',')]
for case in args.cases.split(','):
    K, S, T, kw = CASES[case]
    acts = torch.randint(8, (8, T, K, N), device=dev)
    tape = torch.randint(8, (20 + 120, K, N), device=dev)
    for what in ('rollout',) + (() if args.no_percall else ('percall',)):
        res, keep = {v: [] for v in values}, {}
        for r in range(args.rounds):
            for v in values:
                _lib.set_option(args.option, v)
                ms, outs, state = run_rollout(K, S, T, kw, acts) if what == 'rollout' else run_percall(K, S, kw, tape)
                res[v].append(ms)
                keep[v] = (outs, state)
        _lib.set_option(args.option, None)
        same = all(torch.equal(a, b) for v in values[1:] for x, y in zip(keep[values[0]][0], keep[v][0]) for a, b in zip(x, y)) and \
            all(torch.equal(a, b) for v in values[1:] for a, b in zip(keep[values[0]][1], keep[v][1]))
        steps = T if what == 'rollout' else 1
        for v in values:
            x = sorted(res[v])
            med = x[len(x) // 2]
            print(f'{case:10s} {what:8s} {args.option}={v}: median {med * (1 if what == "rollout" else 1e3):.4f} '
                  f'{"ms" if what == "rollout" else "us"} (min {x[0] * (1 if what == "rollout" else 1e3):.4f}), '
                  f'{N * steps / med * 1e3:.4g} env-steps/s', flush=True)
        print(f'{case:10s} {what:8s} outputs and final state identical: {same}', flush=True)
