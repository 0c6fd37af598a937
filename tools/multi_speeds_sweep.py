#!/usr/bin/env python3
"""experiments/speeds.py shape (MultiSnake 4096 x 36 x 36, 10 agents, respawn 'any', 'full' observations) fused rollout: the
round-3 two-wave kernel against the WIDE shapes of multi_rollout_group_kernel (32-bit class words, one buffer), A/B in one process."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

N, K, dev = 4096, 10, torch.device('cuda:0')
for chunk in (4, 16):
    acts = torch.randint(8, (5, chunk, K, N), device=dev, dtype=torch.int64)
    for name, opts in [('two-wave (r03)', dict(WURM_MULTI_GROUP_MIN_ENVS=1 << 40))] + \
            [(f'wide {s}', dict(WURM_MULTI_GROUP_MIN_ENVS=0, WURM_MULTI_GROUP_SHAPE=s)) for s in (5014, 4514, 3014)]:
        with _lib.knobs(**opts):
            env = MultiSnake(N, K, 36, device=dev, seed=0, boost=True, respawn_mode='any')
            env.rollout(acts[0])
            torch.cuda.synchronize()
            best = 1e9
            for r in range(3):
                t0 = time.perf_counter()
                for i in range(1, 5):
                    env.rollout(acts[i])
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 4)
            print(f'{name:16s} chunk {chunk:2d}: {best * 1e3:7.4f} ms  {best / chunk * 1e6:6.1f} us per step  {N * chunk / best:.3g} env-steps/s  '
                  f'obs {155520.0 * N * chunk / best / 1e12:.2f} TB/s', flush=True)
            del env
