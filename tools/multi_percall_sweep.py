#!/usr/bin/env python3
"""cfg4 per call (`step(a); reset(d['__all__'], return_observations=False)`) with the grouped observation writer of the
per-call step kernel off / 4 / 8 envs per workgroup (WURM_MULTI_GROUP_STEP_WPB), A/B in one process."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

N, K, dev, T = 4096, 4, torch.device('cuda:0'), 60
acts = torch.randint(8, (T + 10, K, N), device=dev, dtype=torch.int64)
keys = [f'agent_{i}' for i in range(K)]
best = {}
for rnd in range(3):
    for wpb in (0, 1, 4, 8):
        for mirror in (None, False):
            with _lib.knobs(WURM_MULTI_GROUP_STEP_WPB=wpb):
                env = MultiSnake(N, K, 25, device=dev, seed=0, resident_mirror=mirror)
                for t in range(T + 10):
                    if t == 10:
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                    o = env.step(dict(zip(keys, acts[t].unbind(0))))
                    env.reset(o[2]['__all__'], return_observations=False)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / T
                k = (wpb, 'mirror' if mirror is None else 'no mirror')
                best[k] = min(best.get(k, 1e9), dt)
for k, v in sorted(best.items()):
    print(f'grouped writer wpb={k[0]} {k[1]:10s}: {v * 1e6:6.2f} us per iteration', flush=True)
