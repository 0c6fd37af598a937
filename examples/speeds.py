#!/usr/bin/env python3
"""The reference's only benchmark script (experiments/speeds.py:10-44: MultiSnake, 10 agents, 36x36, num_envs 2^4..2^12,
random actions, step + reset + check_consistency) against wurm_amd, plus the same loop through the fused rollout."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--num-agents', type=int, default=10)
ap.add_argument('--size', type=int, default=36)
ap.add_argument('--num-steps', type=int, default=10)
args = ap.parse_args()

for n in [2 ** k for k in range(4, 13)]:
    env = MultiSnake(num_envs=n, num_snakes=args.num_agents, size=args.size, boost=True, device='cuda',
                     respawn_mode='any', seed=0)
    tape = torch.randint(8, size=(3 * args.num_steps, args.num_agents, n), device='cuda')

    def loop(lo):                                                            # speeds.py:30-38
        for i in range(lo, lo + args.num_steps):
            observations, reward, done, info = env.step({f'agent_{a}': tape[i, a] for a in range(args.num_agents)})
            env.reset(done['__all__'])
            env.check_consistency()

    loop(0)  # untimed: the first pass pays for the allocator's first hipMallocs of every output shape
    torch.cuda.synchronize()
    t0 = time.time()
    loop(args.num_steps)
    torch.cuda.synchronize()
    per_call = n * args.num_steps / (time.time() - t0)
    env.rollout(tape[:args.num_steps])  # untimed, for the same reason (6 GB of observations at 4096 envs)
    torch.cuda.synchronize()
    t0 = time.time()
    env.rollout(tape[2 * args.num_steps:])
    torch.cuda.synchronize()
    fused = n * args.num_steps / (time.time() - t0)
    print(n, f'{per_call:.0f} env-steps/s (step; reset; check_consistency)', f'{fused:.0f} env-steps/s (fused rollout)')
