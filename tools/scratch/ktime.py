import os, sys
sys.path.insert(0, '/root/repo')
import torch
from wurm_amd.envs import SingleSnake
dev = torch.device('cuda:0')
for N, Ts in [(65536, [1, 16, 64]), (8192, [1, 16, 128])]:
    for T in Ts:
        env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device=dev, seed=0)
        actions = torch.randint(4, (T, N), device=dev, dtype=torch.int64)
        for _ in range(6):
            env.rollout(actions.clone())
        torch.cuda.synchronize()
print('done')
