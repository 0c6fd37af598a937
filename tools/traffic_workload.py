#!/usr/bin/env python3
"""Workload profiled under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) to measure the HBM
traffic of the env-step kernels.  Contains two calibration launches with KNOWN byte counts, as
MI355X_MICROARCH.md §HBM prescribes for gfx950 (FETCH_SIZE under-reports wide streaming reads by 2x; other access
widths must be calibrated in the kernel's own access pattern):
  * torch copy of 1 GiB        (16 B/lane streaming read + write)
  * wurm_single_check on a 1.02 GB state tensor (dword-per-lane coalesced reads — the env kernels' pattern)
then the measured launches: rollout (cfg2 512 envs, cfg3-share 8192 envs), per-call step/reset (cfg2), and the
36x36 step/reset (cfg5 shape)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd import _lib  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

dev = torch.device('cuda:0')
# calibration A: 1 GiB copy
src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
del src, dst
# calibration B: dword-per-lane read of a 1.02 GB state (65536 x 3 x 36 x 36 fp32)
big = SingleSnake(num_envs=65536, size=36, observation_mode='default', device=dev, seed=0)
err = torch.empty(65536, dtype=torch.int32, device=dev)
for _ in range(3):
    _lib.lib().wurm_single_check(_lib.ptr(big.envs), _lib.ptr(err), _lib.i64(65536), 36, None)
torch.cuda.synchronize()
del big

# measured: rollouts
for N, chunk in ((512, 1024), (8192, 128)):
    env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device=dev, seed=0)
    actions = torch.randint(4, (chunk * 5, N), device=dev, dtype=torch.int64)
    for c in range(0, chunk * 5, chunk):
        env.rollout(actions[c:c + chunk])
    torch.cuda.synchronize()
# measured: per-call step/reset at cfg2 and cfg3-share
for N in (512, 8192):
    env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device=dev, seed=0)
    actions = torch.randint(4, (50, N), device=dev, dtype=torch.int64)
    for t in range(50):
        _, _, d, _ = env.step(actions[t])
        env.reset(d)
    torch.cuda.synchronize()
# measured: cfg5 shape (8192 x 36 x 36, default RGB obs), per-call
env = SingleSnake(num_envs=8192, size=36, observation_mode='default', device=dev, seed=0)
actions = torch.randint(4, (20, 8192), device=dev, dtype=torch.int64)
for t in range(20):
    _, _, d, _ = env.step(actions[t])
    env.reset(d)
torch.cuda.synchronize()
print('traffic workload done')
