#!/usr/bin/env python3
"""Does the time of an HBM-bound rollout depend on WHERE its 2 GB observation tensor lies?  cfg5 (8 192 x 36 x 36 'default', 16 steps:
2.04 GB of observations): the launch is repeated into the SAME output tensor (the caching allocator hands the block back each
time) and into a series of DIFFERENT ones (earlier results are kept alive), each launch timed with HIP events; printed with the
tensor's address.  A user process cannot see physical pages; a dependence on the allocation shows up as plateaus per address."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from wurm_amd.envs import SingleSnake  # noqa: E402

dev = torch.device('cuda:0')
if '--prealloc' in sys.argv:           # what bench.py's host calibration does first: a 2 GB block, freed into the cache
    x = torch.empty(2 << 30, dtype=torch.uint8, device=dev); x.fill_(1); torch.cuda.synchronize(); del x
CFG4 = '--cfg4' in sys.argv   # MultiSnake 4 096 x 25 x 25 x 4 'full', 16 steps: 1.97 GB of observations
if CFG4:
    from wurm_amd.envs import MultiSnake
    env = MultiSnake(4096, 4, 25, device=dev, seed=0)
    acts = torch.randint(8, (16, 4, 4096), device=dev)
else:
    env = SingleSnake(8192, 36, observation_mode='default', device=dev, seed=0)
    acts = torch.randint(4, (16, 8192), device=dev)
env.rollout(acts.clone()); torch.cuda.synchronize()


from wurm_amd import _lib  # noqa: E402


def one(rotate=1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a = acts.clone()
    _lib.set_option('WURM_GRID_ROTATE', rotate)
    e0.record()
    out = env.rollout(a)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), out


numel = 16 * 4 * 4096 * 3 * 25 * 25 if CFG4 else 16 * 8192 * 3 * 36 * 36
held = []
for j in range(8):
    # both forms into the SAME block (results are dropped: the caching allocator hands the block back), three launches each
    first = [one(0 if '--swap' in sys.argv else 1) for _ in range(3)]   # (--swap: the other form takes the fresh block first)
    p = first[-1][1]['observations'].data_ptr()
    first = min(t for t, _ in first)
    second = min(one(1 if '--swap' in sys.argv else 0)[0] for _ in range(3))
    rot, same = (second, first) if '--swap' in sys.argv else (first, second)
    again = min(one(1)[0] for _ in range(3))
    skews = {k: min(one(k)[0] for _ in range(3)) for k in ((-1,) if '--linear' in sys.argv else (2, 4, 256))} if not CFG4 else {}
    x = torch.empty(numel, dtype=torch.float32, device=dev)   # takes that block out of the cache: the next launches get another
    assert x.data_ptr() == p
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fills = []
    for _ in range(3):
        e0.record(); x.fill_(0.5); e1.record(); torch.cuda.synchronize()
        fills.append(e0.elapsed_time(e1))
    held.append(x)
    print(f'block {j} at 0x{p:x}: rollout, every env\'s rows in the same order {same:.3f} ms, rows started at an env-dependent row {rot:.3f} ms (again, last: {again:.3f}), '
          + ''.join((f'plane by plane front to back {v:.3f} ms, ' if k < 0 else f'skew by id mod {k} {v:.3f} ms, ') for k, v in skews.items()) +
          f'  '
          f'linear fill of the same block {min(fills):.3f} ms = {numel * 4 / min(fills) / 1e9:.2f} TB/s', flush=True)
