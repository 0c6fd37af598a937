#!/usr/bin/env python3
"""cfg5 (SingleSnake 8 192 x 36 x 36 'default') fused rollout by resident waves per CU (WURM_GRID_WAVES_PER_CU) and steps per launch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

dev = torch.device('cuda:0')
N, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 36)
for wpc in (8, 10, 12, 16, 20, 24, 32):
    with _lib.knobs(WURM_GRID_WAVES_PER_CU=wpc):
        env = SingleSnake(N, S, observation_mode='default', device=dev, seed=0)
        line = f'waves per CU {wpc:2d}:'
        for T in (16, 64):
            reps = 6
            acts = torch.randint(4, (reps + 1, T, N), device=dev, dtype=torch.int64)
            env.rollout(acts[0])
            torch.cuda.synchronize()
            best = 1e9
            for r in range(3):
                t0 = time.perf_counter()
                for i in range(1, reps + 1):
                    env.rollout(acts[i])
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / reps)
            line += f'   T={T}: {best * 1e3:7.4f} ms ({N * T * 3 * S * S * 4 / best / 1e12:5.2f} TB/s of observations)'
        print(line, flush=True)
        del env
