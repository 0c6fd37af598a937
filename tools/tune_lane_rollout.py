#!/usr/bin/env python3
"""Sweep of the fused 9 x 9 `partial_2` rollout: batch size x envs per wave of `lane_rollout_kernel` (and the
one-env-per-wave `rollout_s9_kernel` it replaces for large batches), launches timed with events on the launch stream.
usage (GPU box): python tools/tune_lane_rollout.py [--noobs] > gpurun_out/tune_lane_rollout.jsonl"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

with_obs = '--noobs' not in sys.argv
dev = torch.device('cuda:0')
SIZES = [(1024, 256), (2048, 256), (4096, 128), (8192, 128), (16384, 128), (32768, 64), (65536, 64), (131072, 32)]
only = [int(a) for a in sys.argv[1:] if a.isdigit()]


def time_rollout(N, T, reps=12):
    env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device=dev, seed=0)
    actions = torch.randint(4, (T, N), device=dev, dtype=torch.int64)
    for _ in range(3):
        env.rollout(actions.clone(), return_observations=with_obs)
    acts = [actions.clone() for _ in range(reps)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for a in acts:
        out = env.rollout(a, return_observations=with_obs)
    e1.record()
    torch.cuda.synchronize()
    del out
    return e0.elapsed_time(e1) / reps


for N, T in SIZES:
    if only and N not in only:
        continue
    for epw in ['wave', 4, 8, 16, 32, 64]:
        if epw == 'wave':
            _lib.set_option('WURM_LANE_ROLLOUT_MIN_ENVS', 1 << 40)
        else:
            _lib.set_option('WURM_LANE_ROLLOUT_MIN_ENVS', 0)
            _lib.set_option('WURM_LANE_ROLLOUT_EPW', epw)
            if N // epw < 64:
                continue
        ms = time_rollout(N, T)
        bytes_per = (300 if with_obs else 0) + 23
        print(json.dumps({'N': N, 'T': T, 'epw': epw, 'obs': with_obs, 'ms_per_launch': round(ms, 4),
                          'us_per_batch_step': round(ms * 1e3 / T, 4), 'env_steps_per_s': round(N * T / (ms * 1e-3), 0),
                          'TB_per_s': round(N * T * bytes_per / (ms * 1e-3) / 1e12, 3)}), flush=True)
