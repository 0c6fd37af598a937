"""per call `obs, r, d, info = env.step(a); env.reset(d)` at 65 536 x 9 x 9 by observation mode: resident mirror on / off"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from wurm_amd.envs import SingleSnake
from wurm_amd import _lib
dev = torch.device('cuda:0')
N, T = 65536, 200
acts = torch.randint(4, (T + 10, N), device=dev)
for mode in ('raw', 'partial_3', 'default', 'partial_2'):
    for mirror in (True, False):
        with _lib.knobs(WURM_RESIDENT_MIN_ENVS=None if mirror else 10 ** 9):
            env = SingleSnake(N, 9, observation_mode=mode, device=dev, seed=0)
            for t in range(10):
                _, _, d, _ = env.step(acts[t]); env.reset(d)
            ts = []
            for r in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for t in range(10, 10 + T):
                    _, _, d, _ = env.step(acts[t]); env.reset(d)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / T)
            ts.sort()
            print(f'{mode:10s} mirror {mirror!s:5s} {ts[1] * 1e6:7.2f} us per iteration  {N / ts[1]:.3e} env-steps/s', flush=True)
            del env
