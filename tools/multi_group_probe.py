#!/usr/bin/env python3
"""Which half bounds the cfg4 'full' rollout kernel (multi_rollout_group_kernel)?  A/B on one box in one process, in
alternating order.  Needs the PROBE build of the library (the shipped one has no such switches):
    make -C wurm_amd/csrc probe
    WURM_HIP_LIBRARY=$PWD/wurm_amd/libwurm_hip_probe.so python tools/multi_group_probe.py
WURM_MULTI_GROUP_VARIANT bits: 4 = writers compute but do not store, 8 = steppers skip the transition (results are wrong by
construction with a probe bit; only the time is read)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

N, dev = 4096, torch.device('cuda:0')
assert 'probe' in os.environ.get('WURM_HIP_LIBRARY', ''), 'set WURM_HIP_LIBRARY to the probe build'
variants = [int(v) for v in sys.argv[1:]] or [0, 4, 8, 12]
chunk = 64
env = MultiSnake(N, 4, 25, device=dev, seed=0)
acts = torch.randint(8, (7, chunk, 4, N), device=dev, dtype=torch.int64)
best = {v: 1e9 for v in variants}
for rnd in range(4):
    for v in (variants if rnd % 2 == 0 else variants[::-1]):
        with _lib.knobs(WURM_MULTI_GROUP_VARIANT=v):
            env.rollout(acts[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(1, 7):
                env.rollout(acts[i])
            torch.cuda.synchronize()
            best[v] = min(best[v], (time.perf_counter() - t0) / 6)
for v in variants:
    print(f'variant {v:2d}: best {best[v] * 1e3:.4f} ms per {chunk} steps = {best[v] / chunk * 1e6:.2f} us per step', flush=True)
