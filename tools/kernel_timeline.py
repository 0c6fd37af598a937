#!/usr/bin/env python3
"""In-kernel timeline of the per-call lane kernels (SingleSnake N x 9 x 9 'partial_2').

Needs the instrumented build of the library (s_memtime stamps, wurm_amd/csrc/wurm_device.hpp WURM_TL):
    make -C wurm_amd/csrc timeline          # -> wurm_amd/libwurm_hip_timeline.so (cross-compiles without a GPU)
    WURM_HIP_LIBRARY=$PWD/wurm_amd/libwurm_hip_timeline.so python tools/kernel_timeline.py [--kernel resident|lane_step]
                                                            [--envs 65536] [--epw 32] [--form ref|noobs]
Every wave overwrites the first 64 bytes of its own observation block with eight stamps once its stores have drained; this
script reads them back from the observation `step` returned and prints, per segment between two stamps, the distribution
over the waves in s_memtime ticks — counters of different XCDs are not synchronised, so only differences within one wave
are used.  The observations of a timeline run are garbage by construction."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--kernel', default='resident', choices=['resident', 'lane_step', 'grid'])
ap.add_argument('--size', type=int, default=36, help='--kernel grid: the grid size (cfg5: 8192 envs of 36 x 36, default observation)')
ap.add_argument('--envs', type=int, default=65536)
ap.add_argument('--epw', type=int, default=0)
ap.add_argument('--form', default='ref', choices=['ref', 'noobs'])
ap.add_argument('--iters', type=int, default=12)
args = ap.parse_args()
if 'timeline' not in os.environ.get('WURM_HIP_LIBRARY', ''):
    sys.exit('set WURM_HIP_LIBRARY to the instrumented library (see the docstring)')
N = args.envs
if args.kernel == 'grid':
    os.environ['WURM_GRID_STEP_MIN_CELLS'] = '0'
    names = ['entry', 'grid in LDS', 'action', 'stepped', 'outputs', 'obs issued', 'state stored', 'drained']
    epw = 1
elif args.kernel == 'resident':
    os.environ['WURM_RESIDENT_MIN_ENVS'] = '0'
    nw = 2 if args.form == 'ref' else 1
    epw = args.epw or (64 // nw if N >= 32768 else (32 // nw if N >= 8192 else 16))
    os.environ['WURM_RESIDENT_EPW'] = str(epw)
    names = ['entry', 'loaded', 'stepped', 'outputs', 'state', 'bits', 'crops issued', 'drained']
else:
    os.environ['WURM_RESIDENT_MIN_ENVS'] = str(10 ** 9)
    os.environ['WURM_LANE_STEP_MIN_ENVS'] = '0'
    epw = 16 if N >= 49152 else (8 if N >= 24576 else 4)
    names = ['entry', 'state read', 'validated', 'moved', 'cells+outputs', 'rebuilt', 'crops issued', 'drained']

import numpy as np  # noqa: E402
import torch  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

S_, mode_ = (args.size, 'default') if args.kernel == 'grid' else (9, 'partial_2')
E_ = 3 * S_ * S_ if args.kernel == 'grid' else 75
env = SingleSnake(num_envs=N, size=S_, observation_mode=mode_, device='cuda', seed=0)
g = torch.Generator(device='cuda').manual_seed(1)
nw_ = N // epw
segs, life = [], []
for it in range(args.iters):
    a = torch.randint(0, 4, (N,), device='cuda', generator=g)
    obs, r, d, info = env.step(a)
    env.reset(d) if args.form == 'ref' else env.reset(d, return_observations=False)
    torch.cuda.synchronize()
    if it < 4:
        continue
    st = obs.detach().reshape(-1)[:nw_ * epw * E_].reshape(nw_, epw * E_)[:, :16].contiguous().view(torch.int64)
    st = st.cpu().numpy().astype(np.int64)
    ok = (st[:, 7] > st[:, 0]) & (st[:, 7] - st[:, 0] < 10 ** 7)
    st = st[ok]
    segs.append(st[:, 1:] - st[:, :-1])
    life.append(st[:, 7] - st[:, 0])
seg, life = np.concatenate(segs), np.concatenate(life)
print(f'{args.kernel} kernel, {N} envs, {epw} envs per wave, form {args.form}: {len(life)} wave samples; s_memtime ticks')
for k in range(7):
    c = seg[:, k]
    print(f'  {names[k]:>14s} -> {names[k + 1]:14s} p10 {int(np.percentile(c, 10)):6d}  p50 {int(np.median(c)):6d}  '
          f'p90 {int(np.percentile(c, 90)):6d}  max {int(c.max()):6d}   {100.0 * np.median(c) / np.median(life):5.1f} % of p50 life')
print(f'  wave lifetime: p10 {int(np.percentile(life, 10))}  p50 {int(np.median(life))}  p90 {int(np.percentile(life, 90))}  max {int(life.max())}')
