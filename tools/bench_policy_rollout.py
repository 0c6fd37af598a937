#!/usr/bin/env python3
"""Throughput of the fused acting loop (wurm_single_policy_rollout): env-steps/s with a random-init 75->64->64->{4,1}
policy in the loop, 9x9 partial_2.  usage: bench_policy_rollout.py [num_envs ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.agents import FeedforwardAgent, pack_policy_params  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

for N in [int(a) for a in sys.argv[1:]] or [512, 8192, 65536]:
    torch.manual_seed(0)
    env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device='cuda', seed=0)
    agent = FeedforwardAgent(4, 2, 64, 75).to('cuda')
    params = pack_policy_params(agent)
    state = env.reset()
    T = 256 if N <= 8192 else 32
    for _ in range(2):
        state = env.policy_rollout(params, state, T, check=False)['state']
    torch.cuda.synchronize()
    reps = 8
    t0 = time.perf_counter()
    for _ in range(reps):
        out = env.policy_rollout(params, state, T, check=False)
        state = out['state']
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({'num_envs': N, 'steps_per_launch': T, 'env_steps_per_s': N * T * reps / dt,
                      'ms_per_launch': dt / reps * 1e3, 'done_rate': float(out['dones'].float().mean()),
                      'reward_rate': float(out['rewards'].mean())}))
