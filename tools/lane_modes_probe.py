"""SingleSnake 65 536 x 9 x 9 fused rollouts by observation mode: the lane kernel against the one-env-per-wave kernels
(WURM_LANE_ROLLOUT_MIN_ENVS), optionally by envs per wave"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from wurm_amd.envs import SingleSnake
from wurm_amd import _lib
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
for mode in ('raw', 'partial_3', 'default', 'one_channel', 'partial_2'):
    for knob, epw in ((0, None), (0, 16), (0, 32), (0, 64), (1 << 40, None)):
        with _lib.knobs(WURM_LANE_ROLLOUT_MIN_ENVS=knob, WURM_LANE_ROLLOUT_EPW=epw):
            env = SingleSnake(N, 9, observation_mode=mode, device=dev, seed=0)
            acts = torch.randint(4, (7, T, N), device=dev)
            env.rollout(acts[0]); torch.cuda.synchronize()
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(1, 7):
                    env.rollout(acts[i])
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 6)
            ts.sort()
            elems = env.rollout(acts[0])['observations'].shape[2:].numel()
            print(f'{mode:10s} epw {epw} {_lib.lib().wurm_single_last_route().decode():24s} ms {ts[2]:.4f}  eps {N * T / ts[2] * 1e3:.3e}'
                  f'  obs {N * T * elems * 4 / ts[2] / 1e9:.2f} TB/s', flush=True)
            del env, acts
