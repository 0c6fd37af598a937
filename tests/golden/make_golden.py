"""Generates the golden fixtures under tests/golden/ by running the REAL reference.

Run only in the build container (where /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py [scenario ...]

The reference (oscarknagg/wurm, pure Python on torch) is imported through tests/golden/ref_shim.py; nothing
of it is copied here.  A fixture is DATA: the initial state, the action tape, the reference's recorded random
outcomes (which free cell food respawned in, where resets placed snakes, the uniform tensors it drew) and
the state / outputs the reference produced at every step.  tests/ replay the tape into the oracle (and, on
the GPU box, into the HIP kernels) with the recorded outcomes injected and require bit-equality.

Why injection: the reference picks food cells with randperm + an unstable argsort (wurm/utils.py:188,224-230
in the reference), so its picks are not a function of a seed that any other implementation could restate
(SURVEY.md §0 fact 6).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

SingleSnake, SimpleGridworld, MultiSnake, ref_utils = ref_shim.import_reference()

TAP = [(-1, 0), (0, 1), (1, 0), (0, -1)]  # ORIENTATION_FILTERS taps, orientation d <=> head = neck + TAP[d]


def pack(a: np.ndarray) -> np.ndarray:
    """Integer-valued float grids -> smallest unsigned dtype."""
    a = np.asarray(a)
    r = np.rint(a)
    assert np.array_equal(r, a), 'grid holds non-integers'
    assert r.min() >= 0
    return r.astype(np.uint8 if r.max() < 256 else np.uint16)


def save(name, **arrays):
    if name is None:  # in-memory use (tests/test_oracle_vs_live_reference.py)
        return
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')


def food_cell(food_2d: np.ndarray) -> int:
    idx = np.flatnonzero(food_2d.reshape(-1) > 0.5)
    return int(idx[0]) if len(idx) else -1


# ------------------------------------------------------------------------------------------- SingleSnake

def single_reset_triple(env3: np.ndarray):
    """(seed_y, seed_x, direction, food_cell) of a freshly created SingleSnake slab."""
    S = env3.shape[-1]
    body = env3[2]
    (sy,), (sx,) = np.nonzero(body == 2)
    (hy,), (hx,) = np.nonzero(body == 3)
    d = TAP.index((hy - sy, hx - sx))
    return [int(sy), int(sx), d, food_cell(env3[0])]


def record_single(name, N, S, T, mode, seed, reset_every=1, act_dtype=torch.long):
    torch.manual_seed(seed)
    env = SingleSnake(num_envs=N, size=S, observation_mode=mode, device='cpu')
    actions = torch.randint(4, size=(T, N)).to(act_dtype)
    rec = dict(state0=pack(env.envs.numpy()), actions_in=actions.numpy().copy())
    keys = ['actions_out', 'state_step', 'reward', 'done', 'self_collision', 'edge_collision', 'obs_step',
            'inject_food', 'reset_called', 'reset_mask', 'inject_reset', 'state_reset', 'obs_reset']
    out = {k: [] for k in keys}
    for t in range(T):
        a = actions[t].clone()
        obs, reward, done, info = env.step(a)
        out['actions_out'].append(a.numpy().copy())
        st = env.envs.numpy().copy()
        out['state_step'].append(pack(st))
        out['reward'].append(reward.numpy().reshape(N).copy())
        out['done'].append(done.numpy().reshape(N).astype(np.uint8))
        out['self_collision'].append(info['self_collision'].numpy().astype(np.uint8))
        out['edge_collision'].append(info['edge_collision'].numpy().astype(np.uint8))
        out['obs_step'].append(obs.numpy().copy())
        r = reward.numpy().reshape(N)
        out['inject_food'].append(np.array([food_cell(st[i, 0]) if r[i] > 0 else -1 for i in range(N)], np.int32))

        mask = np.zeros(N, np.uint8)
        inj = np.full((N, 4), -1, np.int32)
        if (t + 1) % reset_every == 0:
            mask = done.numpy().reshape(N).astype(np.uint8)
            obs_r = env.reset(done)
            st_r = env.envs.numpy()
            for i in np.flatnonzero(mask):
                inj[i] = single_reset_triple(st_r[i])
            called = 1
        else:
            obs_r, called = obs, 0  # no reset call at this step: placeholders, skipped by the tests
        out['reset_called'].append(np.uint8(called))
        out['reset_mask'].append(mask)
        out['inject_reset'].append(inj)
        out['state_reset'].append(pack(env.envs.numpy()))
        out['obs_reset'].append(np.asarray(obs_r.numpy() if hasattr(obs_r, 'numpy') else obs_r).copy())
    rec.update({k: np.stack(v) for k, v in out.items()})
    rec['meta'] = np.array([N, S, T, seed, reset_every], np.int64)
    rec['mode'] = np.array(mode)
    save(name, **rec)
    return rec


# ------------------------------------------------------------------------------------------- SimpleGridworld

def record_grid(name, N, S, T, mode, seed, start):
    torch.manual_seed(seed)
    env = SimpleGridworld(num_envs=N, size=S, observation_mode=mode, device='cpu', start_location=start)
    actions = torch.randint(4, size=(T, N)).long()
    rec = dict(state0=pack(env.envs.numpy()), actions_in=actions.numpy().copy())
    keys = ['actions_out', 'state_step', 'reward', 'done', 'edge_collision', 'obs_step', 'inject_food',
            'reset_mask', 'inject_reset', 'state_reset', 'obs_reset']
    out = {k: [] for k in keys}
    for t in range(T):
        a = actions[t].clone()
        obs, reward, done, info = env.step(a)
        out['actions_out'].append(a.numpy().copy())
        st = env.envs.numpy().copy()
        out['state_step'].append(pack(st))
        out['reward'].append(reward.numpy().reshape(N).copy())
        out['done'].append(done.numpy().reshape(N).astype(np.uint8))
        out['edge_collision'].append(info['edge_collision'].numpy().astype(np.uint8))
        out['obs_step'].append(obs.numpy().copy())
        r = reward.numpy().reshape(N)
        out['inject_food'].append(np.array([food_cell(st[i, 0]) if r[i] > 0 else -1 for i in range(N)], np.int32))
        mask = done.numpy().reshape(N).astype(np.uint8)
        obs_r = env.reset(done)
        st_r = env.envs.numpy()
        inj = np.full(N, -1, np.int32)
        for i in np.flatnonzero(mask):
            inj[i] = food_cell(st_r[i, 0])
        out['reset_mask'].append(mask)
        out['inject_reset'].append(inj)
        out['state_reset'].append(pack(st_r))
        out['obs_reset'].append(obs_r.numpy().copy())
    rec.update({k: np.stack(v) for k, v in out.items()})
    rec['meta'] = np.array([N, S, T, seed, start[0], start[1]], np.int64)
    rec['mode'] = np.array(mode)
    save(name, **rec)
    return rec


SCENARIOS = {
    # BASELINE.json cfg2 shape (SingleSnake 9x9 partial_2) at fixture size
    'single_s9_partial2': lambda: record_single('single_s9_partial2', N=48, S=9, T=150, mode='partial_2', seed=11),
    'single_s12_default': lambda: record_single('single_s12_default', N=32, S=12, T=150, mode='default', seed=12),
    'single_s12_one_channel': lambda: record_single('single_s12_one_channel', N=16, S=12, T=80, mode='one_channel',
                                                    seed=13),
    'single_s10_raw': lambda: record_single('single_s10_raw', N=8, S=10, T=60, mode='raw', seed=14),
    'single_s12_positions': lambda: record_single('single_s12_positions', N=16, S=12, T=80, mode='positions',
                                                  seed=15),
    'single_s11_partial3_i32': lambda: record_single('single_s11_partial3_i32', N=16, S=11, T=100, mode='partial_3',
                                                     seed=16, act_dtype=torch.int32),
    # BASELINE.json cfg5 shape (36x36, default RGB) at fixture size
    'single_s36_default': lambda: record_single('single_s36_default', N=6, S=36, T=80, mode='default', seed=17),
    # done envs are stepped again before they are reset: pins the behaviour on irregular states
    'single_s12_lazyreset': lambda: record_single('single_s12_lazyreset', N=24, S=12, T=120, mode='default', seed=18,
                                                  reset_every=4),
    # BASELINE.json cfg1 (SimpleGridworld 64 x 9 x 9, random actions, torch-CPU)
    'grid_s9_default': lambda: record_grid('grid_s9_default', N=64, S=9, T=100, mode='default', seed=21,
                                           start=(4, 4)),
    'grid_s7_raw': lambda: record_grid('grid_s7_raw', N=16, S=7, T=60, mode='raw', seed=22, start=(3, 3)),
}


if __name__ == '__main__':
    try:
        import make_golden_multi  # noqa: F401  (adds the MultiSnake scenarios)
        SCENARIOS.update(make_golden_multi.SCENARIOS)
    except ImportError:
        pass
    names = sys.argv[1:] or list(SCENARIOS)
    for n in names:
        SCENARIOS[n]()
