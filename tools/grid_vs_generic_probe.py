#!/usr/bin/env python3
"""SingleSnake rollouts of grids from 12 x 12 on: the LDS clock-grid kernel (grid_rollout.hip) against the one-env-per-wave
kernels it replaced, by grid size, batch size and observation mode — option WURM_GRID_ROLLOUT_MIN_SIZE switched within one
process.  ms per launch / env-steps/s, both ways."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

dev = torch.device('cuda:0')
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '12,13,14,16,20,24').split(',')]
for N in (65536, 8192):
    for S in sizes:
        if N * S * S * 3 * 4 * 17 > 40e9:
            continue
        for mode in ('partial_2', 'one_channel', 'default'):
            T = 16
            res = {}
            for min_size in (12, 65):
                with _lib.knobs(WURM_GRID_ROLLOUT_MIN_SIZE=min_size):
                    env = SingleSnake(num_envs=N, size=S, observation_mode=mode, device=dev, seed=1)
                    tape = torch.randint(4, (6, T, N), device=dev, dtype=torch.int64)
                    best = 1e9
                    for r in range(6):
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                        out = env.rollout(tape[r])
                        torch.cuda.synchronize()
                        if r >= 2:
                            best = min(best, time.perf_counter() - t0)
                    res[min_size] = best
                    del env, out, tape
            print(f'N={N:6d} S={S:2d} {mode:12s}: clock grid {res[12] * 1e3:8.3f} ms  one env per wave {res[65] * 1e3:8.3f} ms   ratio {res[12] / res[65]:5.2f}', flush=True)
