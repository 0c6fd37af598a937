#!/usr/bin/env python3
"""A/B of two builds of the library on the MultiSnake paths that read and keep the mirror (round 6's experiment with a
mirror of per-snake cell LISTS instead of the grids of clocks: docs/experiments/multi_list_mirror.patch).

    python tools/list_mirror_ab.py wurm_amd/libwurm_hip.so wurm_amd/libwurm_hip_list.so
Each build runs in its own child process (the library is loaded once per process), alternating, best of three."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
from wurm_amd.envs import MultiSnake
dev = torch.device('cuda:0')
TRAIN = dict(observation_mode='partial_5', food_mode='random_rate', respawn_mode='any', boost_cost_prob=0.25,
             food_on_death_prob=0.33, food_rate=2.5e-4)
def percall(N, kw, T=200):
    keys = [f'agent_{i}' for i in range(4)]
    acts = torch.randint(8, (T + 20, 4, N), device=dev, dtype=torch.int64)
    env = MultiSnake(N, 4, 25, device=dev, seed=0, **kw)
    for t in range(T + 20):
        if t == 20:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        o = env.step(dict(zip(keys, acts[t].unbind(0))))
        env.reset(o[2]['__all__'], return_observations=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / T * 1e6
def rollout(N, kw, T=16, reps=30):
    env = MultiSnake(N, 4, 25, device=dev, seed=0, **kw)
    acts = torch.randint(8, (reps + 5, T, 4, N), device=dev, dtype=torch.int64)
    best = 1e9
    for r in range(reps + 5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.rollout(acts[r])
        torch.cuda.synchronize()
        if r >= 5:
            best = min(best, time.perf_counter() - t0)
    return best * 1e6
print('percall_train_512', percall(512, TRAIN))
print('percall_train_4096', percall(4096, TRAIN))
print('percall_cfg4_full_4096', percall(4096, dict(observation_mode='full')))
print('rollout16_train_4096', rollout(4096, TRAIN))
print('rollout16_cfg4_full_4096', rollout(4096, dict(observation_mode='full')))
''' % ROOT

libs = [os.path.abspath(a) for a in sys.argv[1:]]
best = {}
for rnd in range(3):
    for lib in libs:
        env = dict(os.environ, WURM_HIP_LIBRARY=lib)
        out = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True, timeout=600)
        if out.returncode:
            print(out.stderr[-2000:])
            sys.exit(1)
        for ln in out.stdout.splitlines():
            k, v = ln.split()
            d = best.setdefault(k, {})
            d[lib] = min(d.get(lib, 1e9), float(v))
print(f"{'us (best of 3 processes)':28s} " + ' '.join(f'{os.path.basename(l):>24s}' for l in libs))
for k, d in best.items():
    print(f'{k:28s} ' + ' '.join(f'{d[l]:24.2f}' for l in libs))
