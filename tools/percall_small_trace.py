"""the per-call loop `env.step(a); env.reset(d)` at the small BASELINE shapes (cfg2 SingleSnake 512 x 9 x 9 partial_2, cfg1
SimpleGridworld 64 x 9 x 9 default), 2 000 iterations each: wall time per iteration, and a rocprofv3 --kernel-trace target for
the duration of fused_step_kernel itself"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from wurm_amd.envs import SingleSnake, SimpleGridworld
dev = torch.device('cuda:0')
T = 2000
for name, make, N in (('cfg2 SingleSnake 512x9 partial_2', lambda: SingleSnake(512, 9, observation_mode='partial_2', device=dev, seed=0), 512),
                      ('cfg1 SimpleGridworld 64x9 default', lambda: SimpleGridworld(64, 9, start_location=(4, 4), observation_mode='default', device=dev, seed=0), 64)):
    env = make()
    acts = torch.randint(4, (T + 10, N), device=dev)
    for t in range(10):
        _, _, d, _ = env.step(acts[t]); env.reset(d)
    ts = []
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(10, 10 + T):
            _, _, d, _ = env.step(acts[t]); env.reset(d)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / T)
    ts.sort()
    print(f'{name}: {ts[1] * 1e6:.2f} us per iteration (min {ts[0] * 1e6:.2f})', flush=True)
