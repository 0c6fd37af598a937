#!/usr/bin/env python3
"""cfg4 (MultiSnake 4096 x 25 x 25, 4 agents, 'full') fused rollout per shape of multi_rollout_group_kernel
(WURM_MULTI_GROUP_SHAPE = 1000 G + 100 W + 10 EPS + waves per SIMD) and through the older two-wave kernel.  usage: tools/multi_group_sweep.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device('cuda:0')


def run(chunk, reps=6, **kw):
    env = MultiSnake(N, 4, 25, device=dev, seed=0, **kw)
    acts = torch.randint(8, (reps + 1, chunk, 4, N), device=dev, dtype=torch.int64)
    env.rollout(acts[0])
    torch.cuda.synchronize()
    ts = []
    for r in range(3):
        t0 = time.perf_counter()
        for i in range(1, reps + 1):
            env.rollout(acts[i])
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps)
    return min(ts), sorted(ts)[1]


for name, opts in [('two-wave (r03)', dict(WURM_MULTI_GROUP_MIN_ENVS=1 << 40))] + \
        [(f'group {s}', dict(WURM_MULTI_GROUP_MIN_ENVS=0, WURM_MULTI_GROUP_SHAPE=s)) for s in (8416, 8215)]:
    with _lib.knobs(**opts):
        for chunk in (16, 64):
            best, med = run(chunk)
            gb = 30000.0 * N * chunk / 1e9
            print(f'{name:16s} chunk {chunk:3d}: best {best * 1e3:7.4f} ms  median {med * 1e3:7.4f} ms   obs stream {gb / best / 1e3:5.2f} TB/s '
                  f'{N * chunk / best:.3g} env-steps/s', flush=True)
        best, med = run(16, food_mode='random_rate', respawn_mode='any', boost_cost_prob=0.25, food_on_death_prob=0.33, food_rate=2.5e-4)
        print(f'{name:16s} train dynamics, full obs, chunk 16: best {best * 1e3:7.4f} ms', flush=True)
