"""SimpleGridworld per call `obs, r, d, info = env.step(a); env.reset(d)` by batch size and observation mode: the lane kernel
(gridworld_lane_step_kernel, from WURM_LANE_STEP_MIN_ENVS envs) against the one-env-per-wave kernel"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from wurm_amd.envs import SimpleGridworld
from wurm_amd import _lib
dev = torch.device('cuda:0')
T = 200
for N in (16384, 65536):
    acts = torch.randint(4, (T + 10, N), device=dev)
    for mode in ('default', 'raw', 'positions'):
        for lane, epw, mirror in ((True, 4, None), (True, 8, None), (True, 16, None), (True, 32, None), (True, 64, None), (True, 4, False), (True, 16, False),
                                  (False, -1, False)):
            with _lib.knobs(WURM_LANE_STEP_MIN_ENVS=0 if lane else 1 << 40, WURM_GRIDWORLD_LANE_EPW=epw):
                env = SimpleGridworld(N, 9, start_location=(4, 4), observation_mode=mode, device=dev, seed=0, resident_mirror=mirror)
                for t in range(10):
                    _, _, d, _ = env.step(acts[t]); env.reset(d)
                ts = []
                for r in range(3):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for t in range(10, 10 + T):
                        _, _, d, _ = env.step(acts[t]); env.reset(d)
                    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / T)
                ts.sort()
                n0 = _lib.lib().wurm_launch_count()
                for t in range(10, 20):
                    _, _, d, _ = env.step(acts[t]); env.reset(d)
                launches = (_lib.lib().wurm_launch_count() - n0) / 10
                print(f'N {N:6d} {mode:10s} epw {epw:2d} mirror {env.mirror_state()["state"]:5s} launches {launches:.1f} {_lib.lib().wurm_single_last_route().decode():20s} {ts[1] * 1e6:7.2f} us per iteration  '
                      f'{N / ts[1]:.3e} env-steps/s', flush=True)
                del env
