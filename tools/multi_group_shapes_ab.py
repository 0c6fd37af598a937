#!/usr/bin/env python3
"""cfg4 'full' fused rollout (BASELINE configs[3]: MultiSnake 4096 x 25 x 25 x 4, 16 steps per launch) by the shape of
multi_rollout_group_kernel (WURM_MULTI_GROUP_SHAPE = 1000 G + 100 W + 10 EPS + waves per SIMD), shape-specialised and generic
(WURM_MULTI_SHAPE_KERNELS), alternating within one process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

dev = torch.device('cuda:0')
N, K, S = 4096, 4, 25
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
acts = torch.randint(8, (8, T, K, N), device=dev)
res = {}
for rnd in range(4):
    for shape in (8215, 8416, 4414):
        for spec in (1, 0):
            _lib.set_option('WURM_MULTI_GROUP_SHAPE', shape)
            _lib.set_option('WURM_MULTI_SHAPE_KERNELS', spec)
            env = MultiSnake(N, K, S, device=dev, seed=0)
            for i in range(3):
                env.rollout(acts[i])
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ts = []
            for rep in range(5):
                ev[0].record()
                for i in range(8):
                    env.rollout(acts[i])
                ev[1].record()
                torch.cuda.synchronize()
                ts.append(ev[0].elapsed_time(ev[1]) / 8)
            res.setdefault((shape, spec), []).append(sorted(ts)[2])
            del env
for (shape, spec), v in res.items():
    v.sort()
    print(f'shape {shape} specialised {spec}: median {v[len(v) // 2]:.4f} ms (min {v[0]:.4f}, max {v[-1]:.4f}) per {T}-step launch, '
          f'{N * T / v[len(v) // 2] * 1e3:.4g} env-steps/s')
