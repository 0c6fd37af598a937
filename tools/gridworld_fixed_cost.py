#!/usr/bin/env python3
"""SimpleGridworld 65 536 x 9 x 9 lane rollout: launch time by steps per launch (slope = the steady state per step, intercept =
what a launch costs before and after its steps), observation modes 'default' and 'none'."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import SimpleGridworld  # noqa: E402

dev = torch.device('cuda:0')
N = 65536
for mode in ('default', 'none'):
    rows = []
    for T in (1, 2, 4, 8, 16, 32, 64):
        env = SimpleGridworld(N, 9, start_location=(4, 4), observation_mode='default', device=dev, seed=0)
        acts = torch.randint(4, (5, T, N), device=dev)
        kw = {} if mode == 'default' else {'return_observations': False}
        env.rollout(acts[0], **kw); torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(1, 5):
                env.rollout(acts[i], **kw)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 4)
        ts.sort()
        rows.append((T, ts[2]))
        print(f'{mode:8s} {T:2d} steps per launch: {ts[2] * 1e3:8.1f} us  ({ts[2] * 1e3 / T:6.2f} us per step)', flush=True)
        del env, acts
    (t1, y1), (t2, y2) = rows[-3], rows[-1]
    slope = (y2 - y1) / (t2 - t1)
    print(f'{mode:8s} slope {slope * 1e3:.2f} us per step, intercept {(y2 - slope * t2) * 1e3:.1f} us per launch')
