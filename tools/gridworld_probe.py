#!/usr/bin/env python3
"""SimpleGridworld 65 536 x 9 x 9 fused rollouts by observation mode, steps per launch and envs per wave of the lane kernel
(WURM_GRIDWORLD_LANE_EPW; -1: automatic), against the one-env-per-wave kernels (WURM_LANE_ROLLOUT_MIN_ENVS beyond the batch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import SimpleGridworld  # noqa: E402

dev = torch.device('cuda:0')
N = 65536
for mode, T in (('default', 16), ('default', 64), ('raw', 16), ('positions', 64)):
    for knob, epw in ((0, -1), (0, 8), (0, 16), (0, 32), (0, 64), (1 << 40, -1)):
        if mode == 'positions' and epw not in (64, -1):
            continue
        with _lib.knobs(WURM_LANE_ROLLOUT_MIN_ENVS=knob, WURM_GRIDWORLD_LANE_EPW=epw):
            env = SimpleGridworld(N, 9, start_location=(4, 4), observation_mode=mode, device=dev, seed=0)
            acts = torch.randint(4, (5, T, N), device=dev)
            env.rollout(acts[0]); torch.cuda.synchronize()
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(1, 5):
                    env.rollout(acts[i])
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 4)
            ts.sort()
            elems = {'default': 243, 'raw': 162, 'positions': 4}[mode]
            print(f'{mode:9s} {T:2d} steps  ' + (f'lane, envs per wave {epw:2d}' if knob == 0 else 'one env per wave     ') +
                  f'  {_lib.lib().wurm_single_last_route().decode():16s} {ts[2]:.4f} ms (min {ts[0]:.4f})  {N * T / ts[2] * 1e3:.3e} env-steps/s  '
                  f'{N * T * elems * 4 / ts[2] / 1e9:.2f} TB/s of observations', flush=True)
            del env, acts
