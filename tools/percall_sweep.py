"""Wall time per iteration of the per-call loop `env.step(a); env.reset(d)` (SingleSnake N x 9 x 9 partial_2): the shipped
per-call kernels against the resident-mirror step (wurm_amd/csrc/lane_resident.hpp) at every envs-per-wave setting, in the
reference form (`reset(d)` returns its observation) and without the reset observation.  usage: percall_sweep.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wurm_amd import _lib
from wurm_amd.envs import SingleSnake

def run(N, form, iters=300):
    env = SingleSnake(num_envs=N, size=9, observation_mode='partial_2', device='cuda', seed=0)
    a = torch.randint(4, (iters + 50, N), device='cuda')
    for t in range(50):
        _, _, d, _ = env.step(a[t])
        env.reset(d) if form == 'ref' else env.reset(d, return_observations=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(50, 50 + iters):
        _, _, d, _ = env.step(a[t])
        env.reset(d) if form == 'ref' else env.reset(d, return_observations=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return dt * 1e6, env._mirror is not None

for N in [int(x) for x in sys.argv[1:]] or [4096, 8192, 16384, 32768, 65536]:
    for form in ('ref', 'noobs'):
        row = []
        for mirror in (10 ** 9, 0):
            _lib.set_option('WURM_RESIDENT_MIN_ENVS', mirror)
            for epw in ((None,) if mirror else (None, 16, 32, 64)):
                if epw is None: _lib.set_option('WURM_RESIDENT_EPW', None)
                else: _lib.set_option('WURM_RESIDENT_EPW', epw)
                us, on = run(N, form)
                row.append(f"{'mirror' if on else 'plain'}{'' if epw is None else '/' + str(epw)} {us:6.2f} us ({N / us / 1e3:5.2f}e9/s)")
        print(f'N={N:6d} {form:5s} ' + ' | '.join(row), flush=True)
