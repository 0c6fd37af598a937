#!/usr/bin/env python3
"""Workload profiled under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) to measure the HBM
traffic of the env-step kernels.  Contains two calibration launches with KNOWN byte counts, as
MI355X_MICROARCH.md §HBM prescribes for gfx950 (FETCH_SIZE under-reports wide streaming reads by 2x; other access
widths must be calibrated in the kernel's own access pattern):
  * torch copy of 1 GiB        (16 B/lane streaming read + write)
  * wurm_single_check on a 1.02 GB state tensor (dword-per-lane coalesced reads — the env kernels' pattern)
then the measured launches: rollouts (cfg2 512 envs, cfg3-share 8192 envs, cfg5 8192 x 36 x 36, cfg4 MultiSnake) and the
per-call loop (one fused launch per iteration) at cfg2 / cfg3-share / cfg3 / cfg5 and MultiSnake cfg4."""
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
for N, chunk in ((512, 1024), (8192, 128), (16384, 128), (32768, 64), (65536, 64)):  # bench.py's shapes at 1..8 GPUs
    env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device=dev, seed=0)
    actions = torch.randint(4, (chunk * 5, N), device=dev, dtype=torch.int64)
    for c in range(0, chunk * 5, chunk):
        env.rollout(actions[c:c + chunk])
    torch.cuda.synchronize()
# measured (round 4): the other whole-grid observations of 9 x 9 through the lane kernels — one_channel (the reference's
# constructor default) and default: rollout (32 batch-steps per launch) and the per-call loop in the reference's form
for mode in ('one_channel', 'default'):
    env = SingleSnake(num_envs=65536, size=9, observation_mode=mode, device=dev, seed=0)
    actions = torch.randint(4, (32 * 5, 65536), device=dev, dtype=torch.int64)
    for c in range(0, 32 * 5, 32):
        env.rollout(actions[c:c + 32])
    torch.cuda.synchronize()
    env = SingleSnake(num_envs=65536, size=9, observation_mode=mode, device=dev, seed=0)
    for t in range(30):
        _, _, d, _ = env.step(actions[t])
        env.reset(d)
    torch.cuda.synchronize()
    del env, actions
# measured: the big-grid rollouts — cfg5 (grid_rollout_kernel) and cfg4 (multi_rollout_kernel), 16 batch-steps per launch
env = SingleSnake(num_envs=8192, size=36, observation_mode='default', device=dev, seed=0)
actions = torch.randint(4, (16 * 5, 8192), device=dev, dtype=torch.int64)
for c in range(0, 16 * 5, 16):
    env.rollout(actions[c:c + 16])
torch.cuda.synchronize()
del env, actions
from wurm_amd.envs import MultiSnake  # noqa: E402
env = MultiSnake(4096, 4, 25, device=dev, seed=0)
actions = torch.randint(8, (16 * 5, 4, 4096), device=dev, dtype=torch.int64)
for c in range(0, 16 * 5, 16):
    env.rollout(actions[c:c + 16])
torch.cuda.synchronize()
del env, actions
# measured: the MultiSnake variants the reference itself runs — training dynamics with partial_5 crops
# (tests/test_multi_snake_env.py:100-104) and the experiments/speeds.py shape (10 agents on 36 x 36) — rollout and per call
def cfg4prime():
    return MultiSnake(4096, 4, 25, device=dev, seed=0, respawn_mode='any', food_mode='random_rate', boost_cost_prob=0.25,
                      observation_mode='partial_5', food_on_death_prob=0.33, food_rate=2.5e-4)


def speeds_env():
    return MultiSnake(4096, 10, 36, device=dev, seed=0, boost=True, respawn_mode='any')


for make, K, chunk in ((cfg4prime, 4, 16), (speeds_env, 10, 4)):
    env = make()
    actions = torch.randint(8, (chunk * 5, K, 4096), device=dev, dtype=torch.int64)
    for c in range(0, chunk * 5, chunk):
        env.rollout(actions[c:c + chunk])
    torch.cuda.synchronize()
    env = make()
    for t in range(20):
        _, _, d, _ = env.step({f'agent_{i}': actions[t, i] for i in range(K)})
        env.reset(d['__all__'], return_observations=False)
    torch.cuda.synchronize()
    del env, actions
# measured: the per-call loop (one fused launch per `step(a); reset(done)` iteration) at cfg2, cfg3-share, cfg3, cfg5.
# From 4096 envs of 9 x 9 the step runs on the resident mirror (lane_resident.hpp): first as shipped — without the reset
# observation, then in the reference form at 65 536 — then with the mirror switched off (lane_step_kernel / fused_step_kernel)
import os
def percall(N, S, mode, reset_obs=False):
    env = SingleSnake(num_envs=N, size=S, observation_mode=mode, device=dev, seed=0)
    actions = torch.randint(4, (30, N), device=dev, dtype=torch.int64)
    for t in range(30):
        _, _, d, _ = env.step(actions[t])
        env.reset(d, return_observations=reset_obs)
    torch.cuda.synchronize()
for N, S, mode in ((512, 9, 'partial_2'), (8192, 9, 'partial_2'), (65536, 9, 'partial_2'), (8192, 36, 'default')):
    percall(N, S, mode)
percall(65536, 9, 'partial_2', reset_obs=True)
_lib.set_option('WURM_RESIDENT_MIN_ENVS', 10 ** 9)
percall(8192, 9, 'partial_2')
percall(65536, 9, 'partial_2')
percall(8192, 36, 'default')   # grid_step_kernel a second time: the first 30 launches kept their grids in the mirror
_lib.set_option('WURM_RESIDENT_MIN_ENVS', None)
# measured: the per-call MultiSnake loop at cfg4 (multi_step_kernel with the postponed reset in front)
# — 20 launches on the resident mirror of foods / heads / bodies (lazy), then 20 with it switched off
for mirror in (True, False):
    if not mirror:
        _lib.set_option('WURM_RESIDENT_MIN_ENVS', 10 ** 9)
    env = MultiSnake(4096, 4, 25, device=dev, seed=0)
    actions = torch.randint(8, (20, 4, 4096), device=dev, dtype=torch.int64)
    for t in range(20):
        _, _, d, _ = env.step({f'agent_{i}': actions[t, i] for i in range(4)})
        env.reset(d['__all__'], return_observations=False)
    torch.cuda.synchronize()
_lib.set_option('WURM_RESIDENT_MIN_ENVS', None)
# measured (round 5): 'raw' and 'partial_3' of 65 536 x 9 x 9 through the lane kernel (byte slab / 7 x 7 bit planes), 32 steps per launch
for mode in ('raw', 'partial_3'):
    env = SingleSnake(num_envs=65536, size=9, observation_mode=mode, device=dev, seed=0)
    actions = torch.randint(4, (32 * 5, 65536), device=dev, dtype=torch.int64)
    for c in range(0, 32 * 5, 32):
        env.rollout(actions[c:c + 32])
    torch.cuda.synchronize()
    del env, actions
# measured (round 6): 10 x 10 'partial_2' and 11 x 11 'default' through the one-env-per-lane kernel of those sizes (lane_wide.hpp), 32 steps per launch
for S, mode in ((10, 'partial_2'), (11, 'default')):
    env = SingleSnake(num_envs=65536, size=S, observation_mode=mode, device=dev, seed=0)
    actions = torch.randint(4, (32 * 5, 65536), device=dev, dtype=torch.int64)
    for c in range(0, 32 * 5, 32):
        env.rollout(actions[c:c + 32])
    torch.cuda.synchronize()
    del env, actions
# measured (round 5): SimpleGridworld 65 536 x 9 x 9 through the one-env-per-lane rollout (gridworld_lane.hip), 16 steps per launch
from wurm_amd.envs import SimpleGridworld  # noqa: E402
for mode in ('default', 'raw'):
    env = SimpleGridworld(65536, 9, start_location=(4, 4), observation_mode=mode, device=dev, seed=0)
    actions = torch.randint(4, (16 * 5, 65536), device=dev, dtype=torch.int64)
    for c in range(0, 16 * 5, 16):
        env.rollout(actions[c:c + 16])
    torch.cuda.synchronize()
    del env, actions
# measured (round 6): the per-call step of SimpleGridworld 65 536 x 9 x 9 'default' on its mirror, reference form
env = SimpleGridworld(65536, 9, start_location=(4, 4), observation_mode='default', device=dev, seed=0)
actions = torch.randint(4, (30, 65536), device=dev, dtype=torch.int64)
for t in range(30):
    _, _, d, _ = env.step(actions[t])
    env.reset(d)
torch.cuda.synchronize()
del env, actions
print('traffic workload done')
