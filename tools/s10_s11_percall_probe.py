#!/usr/bin/env python3
"""The per-call loop `env.step(a); env.reset(done)` of SingleSnake by grid size around 9 x 9 (65 536 envs): us per iteration,
launches per iteration, the route of the step."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = 100
fn = _lib.lib().wurm_single_last_route
fn.restype = ctypes.c_char_p
for S in (9, 10, 11, 12):
    for mode in ('partial_2', 'one_channel', 'default'):
        best = 1e9
        for rnd in range(3):
            env = SingleSnake(num_envs=N, size=S, observation_mode=mode, device=dev, seed=rnd)
            acts = torch.randint(4, (T + 10, N), device=dev, dtype=torch.int64)
            for t in range(T + 10):
                if t == 10:
                    torch.cuda.synchronize(); t0 = time.perf_counter(); l0 = _lib.lib().wurm_launch_count()
                o, r, d, _ = env.step(acts[t])
                env.reset(d, return_observations=False)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / T)
            launches = (_lib.lib().wurm_launch_count() - l0) / T
            del env
        print(f'S={S:2d} {mode:12s}: {best * 1e6:7.2f} us per iteration  {launches:.1f} launches  route={fn().decode()}', flush=True)
