"""lane_step_kernel (wurm_amd/csrc/lane_step.hpp): the per-call SingleSnake step with ONE ENV PER LANE, which takes
launches of at least 12 288 envs of 9 <= size <= 11 ('partial_n' or no observation, RNG mode) — BASELINE configs[2]
whole on one GPU reaches it in tests/test_full_size_parity.py.

Here: (a) as shipped, the oracle follows single envs of batches that take 4, 8 and 16 envs per wave through step /
postponed reset / obs_after;
(b) in a child process with WURM_LANE_STEP_MIN_ENVS=0 every SingleSnake launch in its domain goes through it: the
dedicated cases below (ragged blocks, every size and crop width of the domain, irregular states that must fall back to
the one-env-per-wave code inside the same launch, hand-edited states) and the whole per-call parity suite."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.backends import OracleBackend

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, f'{what}: {a.shape}/{a.dtype} vs {b.shape}/{b.dtype}'
    if a.dtype == np.float32:
        a, b = a.view(np.int32), b.view(np.int32)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        raise AssertionError(f'{what}: {len(bad)} mismatches, first at {bad[0].tolist()}')


def _cmp(ro, rh, t):
    for k in ro:
        if ro[k] is None:
            assert rh[k] is None, k
        else:
            _same(ro[k], rh[k], f'{k} t={t}')


@pytest.mark.parametrize('N,S,mode,T', [
    (200, 9, 'partial_2', 90),    # three blocks of 64 and a ragged one of 8
    (131, 10, 'partial_1', 70),   # a ragged block whose float count is not a multiple of four
    (70, 11, 'partial_4', 60),    # the largest grid and the widest crop of the domain
    (64, 9, 'none', 60),
    (3, 11, 'partial_3', 50),
])
def test_step_postponed_reset_and_obs_after(hip, N, S, mode, T):
    rng = np.random.RandomState(N + S)
    o, h = OracleBackend(seed=31, env_offset=500), hip(seed=31, env_offset=500)
    eo = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(eo, np.ones(N, np.uint8), 'none')
    eh = eo.copy()
    prev = None
    deaths = eats = 0
    for t in range(T):
        a = rng.randint(-3, 9, size=N).astype(np.int64 if t % 2 else np.int32)  # hostile values included
        ao, ah = a.copy(), a.copy()
        kw = dict(call=1 + 2 * t, pre_done=prev, pre_call=2 * t, want_obs_after=(t % 3 != 1))
        ro = o.single_step_reset(eo, ao, mode, **kw)
        rh = h.single_step_reset(eh, ah, mode, **kw)
        _same(ah, ao, f'actions t={t}')
        _same(eh, eo, f'state t={t}')
        _cmp(ro, rh, t)
        deaths += int(ro['done'].sum())
        eats += int((ro['reward'] > 0).sum())
        # every fourth step the done envs are left alone: they are stepped again as they are (vanished heads, body
        # values above the head's) and must come out of the one-env-per-wave code
        prev = ro['done'] if t % 4 != 3 else None
        if t % 9 == 5:  # hand-edited states: a second food, no food, a broken body, food under the body
            eo[0, 0, 2, 2] = 1
            eo[1 % N, 0] = 0
            eo[2 % N, 2, 4, 4] = eo[2 % N, 2].max()
            b = eo[N - 1, 2]
            if b.max() >= 2 and (b == 1).any():
                y, x = np.argwhere(b == 1)[0]
                eo[N - 1, 0] = 0
                eo[N - 1, 0, y, x] = 1
            eh[...] = eo
    assert deaths > 0 and (eats > 0 or N < 10)


def test_plain_step_entry_point(hip):
    """wurm_single_step (no reset in the launch) takes the same kernel"""
    N, S, T = 150, 9, 60
    rng = np.random.RandomState(5)
    o, h = OracleBackend(seed=77), hip(seed=77)
    eo = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(eo, np.ones(N, np.uint8), 'none')
    eh = eo.copy()
    o.call = h.call = 1
    for t in range(T):
        a = rng.randint(0, 4, size=N).astype(np.int64)
        ao, ah = a.copy(), a.copy()
        ro, rh = o.single_step(eo, ao, 'partial_2'), h.single_step(eh, ah, 'partial_2')
        _same(ao, ah, f'actions t={t}')
        _same(eo, eh, f'state t={t}')
        for x, y, w in zip(ro, rh, ('obs', 'reward', 'done', 'self_collision', 'edge_collision')):
            _same(x, y, f'{w} t={t}')
        if t % 5 != 4:
            o.single_reset(eo, ro[2], 'none')
            h.single_reset(eh, rh[2], 'none')
            _same(eo, eh, f'reset state t={t}')
        else:
            o._next()
            h._next()


@pytest.mark.parametrize('N,T', [(12352, 30), (24640, 24), (49216, 24)])  # 4, 8 and 16 envs per wave (launch_lane_step)
def test_at_the_natural_threshold(hip, N, T):
    """batches the dispatch sends to lane_step_kernel as shipped (>= 12 288 envs): the oracle follows single envs of the
    batch by their global id"""
    import torch
    from oracle import oracle
    from wurm_amd.envs import SingleSnake
    S, seed = 9, 11
    env = SingleSnake(N, S, observation_mode='partial_2', device='cuda:0', seed=seed)
    ids = sorted({0, 1, 3, 4, 7, 8, 15, 16, 63, 64, 65, 127, 128, 4095, 4096, 8191, N // 2, N - 2, N - 1} |
                 set(int(i) for i in np.random.RandomState(0).randint(0, N, size=20)))
    refs = {}
    for gid in ids:
        refs[gid] = np.zeros((1, 3, S, S), np.float32)
        oracle.single_reset(refs[gid], np.ones(1, np.uint8), 'none', seed=seed, call=0, env_offset=gid)
    _same(env.envs[ids].cpu().numpy(), np.concatenate([refs[g] for g in ids]), 'fresh')
    g = torch.Generator(device='cuda:0').manual_seed(3)
    actions = torch.randint(4, (T, N), generator=g, device='cuda:0')
    a_host = actions.cpu().numpy()
    call = 1
    for t in range(T):
        a = actions[t].clone()
        obs, r, d, info = env.step(a)
        back = env.reset(d, return_observations=(t % 3 == 0))
        sub = [x[ids].cpu().numpy() for x in (obs, r, d, info['self_collision'], info['edge_collision'], a)]
        back = back[ids].cpu().numpy() if back is not None else None
        for j, gid in enumerate(ids):
            aj = np.ascontiguousarray(a_host[t, gid:gid + 1])
            ro = oracle.single_step(refs[gid], aj, 'partial_2', seed=seed, call=call, env_offset=gid)
            _same(sub[0][j:j + 1], ro[0], f'obs env {gid} t={t}')
            _same(sub[1][j, 0:1], ro[1], f'reward env {gid} t={t}')
            _same(sub[2][j, 0:1].astype(np.uint8), ro[2], f'done env {gid} t={t}')
            _same(sub[3][j:j + 1].astype(np.uint8), ro[3], f'selfc env {gid} t={t}')
            _same(sub[4][j:j + 1].astype(np.uint8), ro[4], f'edgec env {gid} t={t}')
            _same(sub[5][j:j + 1], aj, f'action env {gid} t={t}')
            bo = oracle.single_reset(refs[gid], ro[2], 'partial_2', seed=seed, call=call + 1, env_offset=gid)
            if back is not None:
                _same(back[j:j + 1], bo, f'reset obs env {gid} t={t}')
        call += 2
    _same(env.envs[ids].cpu().numpy(), np.concatenate([refs[g] for g in ids]), 'final state')
    env.check_consistency()


def test_per_call_parity_suite_on_the_lane_step_kernel():
    if os.environ.get('WURM_LANE_STEP_MIN_ENVS') == '0':
        pytest.skip('already inside the forced run')
    env = dict(os.environ, WURM_LANE_STEP_MIN_ENVS='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-m', 'gpu', '-x', 'tests/test_lane_step.py',
                        'tests/test_hip_vs_oracle.py', 'tests/test_hip_fused_step.py', 'tests/test_kat_single_snake.py',
                        'tests/test_rl_gpu.py'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
