#!/usr/bin/env python3
"""Per-call throughput of every BASELINE.json config on one MI355X: the reference's loop `step(a); reset(done)` through
the Python classes, wall and HIP-event time, algorithmic GB/s.  Two forms per config: the reference's own
call (`reset(done)` returns observations) and `reset(done, return_observations=False)` (suffix "no reset obs": what the
reference's callers, who discard that observation, would write against this build).
Prints one JSON object per config.  Not the headline bench (bench.py); used for DESIGN.md §7 and profiles/."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import SingleSnake, SimpleGridworld, MultiSnake  # noqa: E402

dev = torch.device('cuda:0')
HBM = 8000.0


def timed_loop(step_fn, T, warm):
    for t in range(warm):
        step_fn(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for t in range(warm, warm + T):
        step_fn(t)
    e1.record()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, e0.elapsed_time(e1) * 1e-3


def single(N, S, mode, T, warm=20, grid=False, reset_obs=True):
    if grid:
        env = SimpleGridworld(N, S, start_location=(S // 2, S // 2), observation_mode=mode, device=dev, seed=0)
    else:
        env = SingleSnake(N, S, observation_mode=mode, device=dev, seed=0)
    actions = torch.randint(4, (warm + T, N), device=dev)

    def f(t):
        _, _, d, _ = env.step(actions[t])
        env.reset(d) if reset_obs else env.reset(d, return_observations=False)
    wall, gpu = timed_loop(f, T, warm)
    obs = env._obs_shape(mode)
    obs_bytes = 4
    for v in obs[1:]:
        obs_bytes *= v
    per = (12 if grid else 20) * S * S + obs_bytes + (31 if grid else 39)
    return dict(N=N, S=S, mode=mode, env_steps_per_s=N * T / wall, us_per_batch_step=wall / T * 1e6,
                gpu_us_per_batch_step=gpu / T * 1e6, algorithmic_bytes_per_env_step=per,
                algorithmic_GBs=per * N * T / wall / 1e9, frac_hbm=per * N * T / wall / 1e9 / HBM)


def multi(N, K, S, T, warm=10, reset_obs=True, **kw):
    env = MultiSnake(N, K, S, device=dev, seed=0, **kw)
    actions = {f'agent_{i}': torch.randint(8, (warm + T, N), device=dev) for i in range(K)}

    def f(t):
        _, _, d, _ = env.step({k: v[t] for k, v in actions.items()})
        env.reset(d['__all__']) if reset_obs else env.reset(d['__all__'], return_observations=False)
    wall, gpu = timed_loop(f, T, warm)
    mode = env.observation_mode
    obs_bytes = 12 * K * S * S if mode == 'full' else 12 * K * env.observation_size ** 2
    per = 8 * (1 + 2 * K) * S * S + obs_bytes + 40 * K
    return dict(N=N, K=K, S=S, mode=mode, env_steps_per_s=N * T / wall, us_per_batch_step=wall / T * 1e6,
                gpu_us_per_batch_step=gpu / T * 1e6, algorithmic_bytes_per_env_step=per,
                algorithmic_GBs=per * N * T / wall / 1e9, frac_hbm=per * N * T / wall / 1e9 / HBM)


which = sys.argv[1:] or ['cfg1', 'cfg2', 'cfg3', 'cfg4', 'cfg4b', 'cfg5', 'speeds']
out = {}
if 'cfg1' in which:
    out['cfg1 SimpleGridworld 64x9 default'] = single(64, 9, 'default', 2000, grid=True)
if 'cfg2' in which:
    out['cfg2 SingleSnake 512x9 partial_2'] = single(512, 9, 'partial_2', 2000)
if 'cfg3' in which:
    out['cfg3-share SingleSnake 8192x9 partial_2'] = single(8192, 9, 'partial_2', 1000)
    out['cfg3-full SingleSnake 65536x9 partial_2'] = single(65536, 9, 'partial_2', 300)
    out['cfg3-full SingleSnake 65536x9 partial_2, no reset obs'] = single(65536, 9, 'partial_2', 300, reset_obs=False)
if 'cfg4' in which:
    out['cfg4 MultiSnake 4096x25 K=4 defaults (full obs)'] = multi(4096, 4, 25, 200)
    out['cfg4 MultiSnake 4096x25 K=4 defaults (full obs), no reset obs'] = multi(4096, 4, 25, 200, reset_obs=False)
if 'cfg4b' in which:
    out['cfg4b MultiSnake 4096x25 K=4 train dynamics partial_5'] = multi(
        4096, 4, 25, 200, respawn_mode='any', food_mode='random_rate', boost_cost_prob=0.25,
        observation_mode='partial_5', food_on_death_prob=0.33, food_rate=2.5e-4)
if 'cfg5' in which:
    out['cfg5 SingleSnake 8192x36 default'] = single(8192, 36, 'default', 200)
    out['cfg5 SingleSnake 8192x36 default, no reset obs'] = single(8192, 36, 'default', 200, reset_obs=False)
if 'speeds' in which:
    out['speeds.py MultiSnake 4096x36 K=10 respawn any'] = multi(4096, 10, 36, 50, respawn_mode='any')
for k, v in out.items():
    print(json.dumps({k: v}))
