#!/usr/bin/env python3
"""A few MultiSnake step/reset launches (cfg4: 4096 x 25 x 25, K=4, full obs) — target for rocprofv3 --pmc passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

N, K, S, T = 4096, 4, 25, 12
dev = torch.device('cuda:0')
env = MultiSnake(N, K, S, device=dev, seed=0)
actions = {f'agent_{i}': torch.randint(8, (T, N), device=dev) for i in range(K)}
for t in range(T):
    _, _, d, _ = env.step({k: v[t] for k, v in actions.items()})
    env.reset(d['__all__'])
torch.cuda.synchronize()
print('done')
