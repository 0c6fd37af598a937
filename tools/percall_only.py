#!/usr/bin/env python3
"""A few per-call step / reset launches (SingleSnake N x 9 x 9 partial_2; N from argv, default 65536; second argument
`ref`: reset(done) returns its observation as in the reference, default: it does not) — PMC / kernel-trace target."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device='cuda', seed=0)
a = torch.randint(4, (400, N), device='cuda')
for t in range(400):
    _, _, d, _ = env.step(a[t])
    env.reset(d, return_observations=(len(sys.argv) > 2 and sys.argv[2] == 'ref'))
torch.cuda.synchronize()
print('done')
