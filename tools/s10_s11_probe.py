#!/usr/bin/env python3
"""SingleSnake rollouts of 65 536 envs by grid size around the lane kernels' 9 x 9: what the sizes between the lane kernels
(S = 9) and the LDS clock grids (S >= 12) get.  ms per launch, env-steps/s, observation GB/s (the bytes the launch must write)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for S in (9, 10, 11, 12, 13):
    for mode in ('partial_2', 'one_channel', 'default'):
        T = 32 if mode != 'default' else 16
        env = SingleSnake(num_envs=N, size=S, observation_mode=mode, device=dev, seed=1)
        tape = torch.randint(4, (8, T, N), device=dev, dtype=torch.int64)
        best = 1e9
        for r in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = env.rollout(tape[r])
            torch.cuda.synchronize()
            if r >= 2:
                best = min(best, time.perf_counter() - t0)
        obs = out['observations']
        import ctypes
        fn = _lib.lib().wurm_single_last_route
        fn.restype = ctypes.c_char_p
        route = fn().decode()
        nbytes = obs.numel() * 4
        print(f'S={S:2d} {mode:12s} T={T:2d}: {best * 1e3:7.3f} ms  {N * T / best:9.3e} env-steps/s  {nbytes / best / 1e9:7.0f} GB/s of observations'
              f'  route={route}', flush=True)
        del env, out, obs
