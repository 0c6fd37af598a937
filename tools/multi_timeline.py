#!/usr/bin/env python3
"""In-kernel timeline of the MultiSnake per-call step (multi_step_kernel, grouped 'full' writer): where a wave's 40 us go.

Needs the instrumented build (s_memtime stamps in LDS slots, wurm_amd/csrc/wurm_device.hpp WURM_TLS):
    make -C wurm_amd/csrc timeline
    WURM_HIP_LIBRARY=$PWD/wurm_amd/libwurm_hip_timeline.so python tools/multi_timeline.py [--envs 4096] [--snakes 4] [--size 25]
Every stepper wave overwrites the first 128 bytes of agent 0's observation of its env with 16 stamps once its stores have
drained; counters of different XCDs are not synchronised, so only differences within one wave are used.  100 MHz ticks
(10 ns).  The observations of a timeline run are garbage by construction."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--envs', type=int, default=4096)
ap.add_argument('--snakes', type=int, default=4)
ap.add_argument('--size', type=int, default=25)
ap.add_argument('--iters', type=int, default=24)
ap.add_argument('--rollout', type=int, default=0, help='fused rollout of this many steps with the training dynamics and partial_5 (cfg4-prime): per-segment TOTALS over the launch')
ap.add_argument('--train', action='store_true', help="per call with the reference's training dynamics and partial_5 crops (cfg4-prime): multi_step_kernel's plain path")
ap.add_argument('--speeds', action='store_true', help="experiments/speeds.py's env: boost, respawn_mode='any' (one env per workgroup at 10 x 36 x 36)")
args = ap.parse_args()
if 'timeline' not in os.environ.get('WURM_HIP_LIBRARY', ''):
    sys.exit('set WURM_HIP_LIBRARY to the instrumented library (see the docstring)')
import numpy as np  # noqa: E402
import torch  # noqa: E402
from wurm_amd.envs import MultiSnake  # noqa: E402

N, K, S = args.envs, args.snakes, args.size
names = ['entry', 'loaded', 'inputs', 'prologue', 'phase', 'death/delete', 'add_food', 'body done', 'outputs', 'state stored',
         'class codes', 'barrier', 'obs issued', 'image in LDS', 'food bits', 'drained']
order = [0, 13, 14, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15]  # the stamps in the order a wave passes them
kw = dict(boost=True, respawn_mode='any') if args.speeds else {}
if args.rollout:
    T = args.rollout
    env = MultiSnake(N, K, S, device=torch.device('cuda:0'), seed=0, observation_mode='partial_5', food_mode='random_rate',
                     respawn_mode='any', boost_cost_prob=0.25, food_on_death_prob=0.33, food_rate=2.5e-4)
    acts = torch.randint(8, (6, T, K, N), device='cuda')
    tot = []
    for it in range(6):
        out = env.rollout(acts[it])
        torch.cuda.synchronize()
        if it >= 2:
            tot.append(out['observations'][0, 0].reshape(N, -1)[:, :32].contiguous().view(torch.int64).cpu().numpy())
    tot = np.concatenate(tot).astype(np.float64) / T
    seg = {0: 'reset (prev step)', 2: 'actions', 3: 'boost phase', 4: 'phase', 5: 'death food / delete', 10: 'food count',
           11: 'free cells', 12: 'count + binomial', 6: 'food placed', 7: 'body tail', 8: 'outputs', 13: 'pixel codes + table',
           9: 'crops written'}
    life = tot[:, :15].sum(1)
    print(f'multi_rollout_kernel, {N} x {S} x {S} x {K}, training dynamics, partial_5, {T} steps: cycles per step (mean over waves; p90)')
    for k, name in seg.items():
        c = tot[:, k]
        print(f'  {name:>20s}  mean {c.mean():8.0f}  p90 {np.percentile(c, 90):8.0f}   {100 * c.mean() / life.mean():5.1f} %')
    print(f'  per step: mean {life.mean():.0f}  p90 {np.percentile(life, 90):.0f}')
    sys.exit(0)
if args.train:
    kw = dict(observation_mode='partial_5', food_mode='random_rate', respawn_mode='any', boost_cost_prob=0.25,
              food_on_death_prob=0.33, food_rate=2.5e-4)
    names = ['entry', 'loaded (mirror + inputs)', 'reset applied, inputs', 'prologue / boost phase', 'phase', 'death / delete',
             'food placed', 'body done', 'outputs', 'state stored', 'cell codes + counts', '-', 'crops issued', 'pixel table', '-', 'drained']
    order = [0, 1, 2, 3, 4, 5, 10, 6, 7, 8, 9, 13, 12, 15]
env = MultiSnake(N, K, S, device=torch.device('cuda:0'), seed=0, **kw)
if args.speeds:  # multi_step_wg_kernel stamps fewer points
    order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 15]
    names[9], names[1] = 'state stored', 'loaded'
keys = [f'agent_{i}' for i in range(K)]
acts = torch.randint(8, (args.iters, K, N), device='cuda')
rows = []
for it in range(args.iters):
    o = env.step(dict(zip(keys, acts[it].unbind(0))))
    env.reset(o[2]['__all__'], return_observations=False)
    torch.cuda.synchronize()
    if it < 6:
        continue
    st = o[0]['agent_0'].reshape(N, -1)[:, :32].contiguous().view(torch.int64).cpu().numpy()
    rows.append(st)
st = np.concatenate(rows)
idx = order
ok = (st[:, 15] > st[:, 0]) & (st[:, 15] - st[:, 0] < 10 ** 6)
st = st[ok]
life = st[:, 15] - st[:, 0]
print(f'multi_step_kernel, {N} x {S} x {S} x {K}: {len(st)} wave samples; 10 ns ticks')
for a, b in zip(idx[:-1], idx[1:]):
    c = st[:, b] - st[:, a]
    print(f'  {names[a]:>24s} -> {names[b]:24s} p10 {int(np.percentile(c, 10)):6d}  p50 {int(np.median(c)):6d}  p90 {int(np.percentile(c, 90)):6d}  '
          f'max {int(c.max()):6d}   {100.0 * np.median(c) / np.median(life):5.1f} % of p50 life')
print(f'  wave lifetime: p10 {int(np.percentile(life, 10))}  p50 {int(np.median(life))}  p90 {int(np.percentile(life, 90))}  max {int(life.max())}')
