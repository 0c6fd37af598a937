#!/usr/bin/env python3
"""Per-call MultiSnake loop `obs, r, d, info = env.step(a); env.reset(d['__all__'], False)` at small batches (host-bound):
us per iteration and launches per iteration."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

dev = torch.device('cuda:0')
TRAIN = dict(food_mode='random_rate', respawn_mode='any', boost_cost_prob=0.25, food_on_death_prob=0.33, food_rate=2.5e-4)
for name, (N, K, S), kw in (('512x12x12_k2', (512, 2, 12), {}), ('512x25x25_k4_train_partial5', (512, 4, 25), dict(TRAIN, observation_mode='partial_5')),
                            ('4096x25x25_k4_full', (4096, 4, 25), {}), ('4096x25x25_k4_train_partial5', (4096, 4, 25), dict(TRAIN, observation_mode='partial_5'))):
    env = MultiSnake(N, K, S, device=dev, seed=0, **kw)
    T = 400
    a = torch.randint(8, (T, K, N), device=dev)
    keys = [f'agent_{i}' for i in range(K)]
    acts = [dict(zip(keys, a[t].unbind(0))) for t in range(T)]   # (what the policy hands over: made outside the timed loop)
    for form in ('rows', 'obs'):
        for rep in range(2):    # (two passes over the tape first: output slabs and the allocator's blocks exist)
            for t in range(T):
                o, r, d, info = env.step(acts[t])
                env.reset(d['__all__'], return_observations=(form == 'obs'))
        torch.cuda.synchronize()
        n0 = _lib.lib().wurm_launch_count()
        t0 = time.perf_counter()
        for t in range(T):
            o, r, d, info = env.step(acts[t])
            env.reset(d['__all__'], return_observations=(form == 'obs'))
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / T * 1e6
        print(f'{name:32s} reset obs={form == "obs"!s:5s}: {us:7.2f} us per iteration (host issue {t_host / T * 1e6:6.2f} us), '
              f'{(_lib.lib().wurm_launch_count() - n0) / T:.2f} launches, {N / us * 1e6:.4g} env-steps/s, machine {type(env._fs).__name__}')
