#!/usr/bin/env python3
"""experiments/speeds.py shape per call (4 096 x 36 x 36 x 10, boost, respawn 'any'; multi_step_wg_kernel): the loop as written
(`step; reset(done['__all__'])`, two observations per iteration) and with return_observations=False, A/B over
WURM_MULTI_GROUP_VARIANT bit 0 (whole agent views per wave / (agent, half) items)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

N, K, S, T, dev = 4096, 10, 36, 30, torch.device('cuda:0')
acts = torch.randint(8, (T + 6, K, N), device=dev, dtype=torch.int64)
keys = [f'agent_{i}' for i in range(K)]
best = {}
for rnd in range(3):
    for variant in (0, 1):
        for form in ('no reset obs', 'as written'):
            with _lib.knobs(WURM_MULTI_GROUP_VARIANT=variant):
                env = MultiSnake(N, K, S, device=dev, seed=0, boost=True, respawn_mode='any')
                for t in range(T + 6):
                    if t == 6:
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                    o = env.step(dict(zip(keys, acts[t].unbind(0))))
                    if form == 'as written':
                        env.reset(o[2]['__all__'])
                    else:
                        env.reset(o[2]['__all__'], return_observations=False)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / T
                best[(variant, form)] = min(best.get((variant, form), 1e9), dt)
                del env
for k, v in sorted(best.items()):
    print(f'variant {k[0]} ({"whole views" if k[0] else "(agent, half) items"})  {k[1]:12s}: {v * 1e6:7.1f} us per iteration', flush=True)
