#!/usr/bin/env python3
"""What kind of box is this?  The HBM-bound rollouts differ by up to 20 % between boxes of the pool (cfg5 16 steps: 0.385 against
0.46-0.47 ms for the same launch): this prints what a user process can see of the box — rocm-smi's partition modes, clocks and
power cap, the fill / copy rate — beside the cfg5 and cfg4 16-step launch times, one line of JSON per run (profiles/r04_box_probe.jsonl)."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import MultiSnake, SingleSnake  # noqa: E402


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=30).stdout.strip()
    except Exception as e:  # noqa: BLE001
        return f'({e})'


dev = torch.device('cuda:0')
rec = {'host': sh('hostname'), 'name': torch.cuda.get_device_name(0),
       'partition': sh('rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -i partition'),
       'clocks': sh('rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -4'),
       'power': sh('rocm-smi --showmaxpower --showpower 2>/dev/null | grep -iE "power" | head -3'),
       'vbios': sh('rocm-smi --showvbios 2>/dev/null | grep -i vbios | head -1')}
x = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); x.fill_(1); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
rec['fill_TBps'] = round((2 << 30) / best / 1e12, 2)
del x


def timed(env, acts):
    env.rollout(acts[0]); torch.cuda.synchronize()
    b = 1e9
    for r in range(3):
        t0 = time.perf_counter()
        for i in range(1, acts.shape[0]):
            env.rollout(acts[i])
        torch.cuda.synchronize()
        b = min(b, (time.perf_counter() - t0) / (acts.shape[0] - 1))
    return round(b * 1e3, 4)


e5 = SingleSnake(8192, 36, observation_mode='default', device=dev, seed=0)
rec['cfg5_16_ms'] = timed(e5, torch.randint(4, (7, 16, 8192), device=dev))
rec['cfg5_64_ms'] = timed(e5, torch.randint(4, (5, 64, 8192), device=dev))
del e5
e4 = MultiSnake(4096, 4, 25, device=dev, seed=0)
rec['cfg4_16_ms'] = timed(e4, torch.randint(8, (7, 16, 4, 4096), device=dev))
del e4
e3 = SingleSnake(65536, 9, observation_mode='partial_2', device=dev, seed=0)
rec['cfg3_64_ms'] = timed(e3, torch.randint(4, (7, 64, 65536), device=dev))
print(json.dumps(rec))
