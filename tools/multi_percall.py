#!/usr/bin/env python3
"""The per-call MultiSnake loop at a given shape (target for rocprofv3 --kernel-trace).
usage: multi_percall.py N K S [check] [noobs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

N, K, S = (int(v) for v in sys.argv[1:4])
check = 'check' in sys.argv
noobs = 'noobs' in sys.argv
T = 30
dev = torch.device('cuda:0')
kw = dict(boost=True, respawn_mode='any') if 'speeds' in sys.argv else {}
env = MultiSnake(N, K, S, device=dev, seed=0, **kw)
actions = torch.randint(8, (T + 5, K, N), device=dev)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(5 * rep, 5 * rep + (T if rep else 5)):
        _, _, d, _ = env.step({f'agent_{i}': actions[t, i] for i in range(K)})
        if noobs:
            env.reset(d['__all__'], return_observations=False)
        else:
            env.reset(d['__all__'])
        if check:
            env.check_consistency()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f'{N}x{K}x{S}: {dt / T * 1e6:.1f} us per iteration, {N * T / dt:.3e} env-steps/s')
