"""The per-call SingleSnake step on a resident compact mirror of the state (wurm_amd/csrc/lane_resident.hpp;
wurm_single_call.resident / resident_valid, wurm_single_resident_bytes of include/wurm_hip.h): 9 x 9, 'partial_2' or no
observation.  `envs` is still written every call, so every check below compares the state too.

(a) through the C ABI: the oracle follows step / postponed reset / obs_after over ragged batches at every envs-per-wave
    setting, with finished envs that are stepped again without the reset, hostile action values, hand-edited states (the
    caller clears resident_valid) and other entry points in between;
(b) through the host classes: the loop of experiments/main.py:212-227 in its reset forms, in-place edits of a state
    tensor the caller holds (found through the tensor's version counter), rollouts and eager resets in between; a batch
    at the natural threshold;
(c) in a child process with WURM_RESIDENT_MIN_ENVS=0 the per-call parity suite as a whole."""
import contextlib
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


from wurm_amd._lib import knobs  # noqa: E402  (library options + environment for child processes)


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


@pytest.mark.parametrize('lazy', [False, True])
@pytest.mark.parametrize('epw', [0, 16, 32, 64])
@pytest.mark.parametrize('N,mode,T', [
    (200, 'partial_2', 90),   # whole blocks and a ragged one at every envs-per-wave setting
    (131, 'partial_2', 70),   # an odd count: the crops of the ragged block go out float by float
    (64, 'none', 60),
    (3, 'partial_2', 50),
    # round 4: every other observation (lr_obs_value) — 'one_channel' is the reference's constructor default
    (137, 'one_channel', 60),
    (70, 'default', 50),
    (90, 'positions', 50),
    (75, 'partial_0', 40),
    (75, 'partial_1', 40),
    # round 5: 'raw' through a byte slab, 'partial_3' through 7 x 7 bit planes
    (137, 'raw', 60),
    (70, 'raw', 45),
    (137, 'partial_3', 60),
    (64, 'partial_3', 45),
])
def test_abi_step_postponed_reset_and_obs_after(hip, epw, N, mode, T, lazy):
    S = 9
    rng = np.random.RandomState(N + epw)
    o, h = OracleBackend(seed=31, env_offset=500), hip(seed=31, env_offset=500)
    eo = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(eo, np.ones(N, np.uint8), 'none')
    eh = eo.copy()
    prev = None
    deaths = eats = 0
    mirror = {'valid': 0, 'lazy': lazy}
    with knobs(WURM_RESIDENT_EPW=epw or None):
        for t in range(T):
            a = rng.randint(-3, 9, size=N).astype(np.int64 if t % 2 else np.int32)  # hostile values included
            ao, ah = a.copy(), a.copy()
            kw = dict(call=1 + 2 * t, pre_done=prev, pre_call=2 * t, want_obs_after=(t % 3 != 1))
            # lazy: `envs` is written out of the mirror only now and then — and before the edits / other entry points below
            mirror['sync'] = not lazy or t % 5 == 4 or t % 9 == 5 or t % 11 == 7 or t == T - 1
            ro = o.single_step_reset(eo, ao, mode, **kw)
            rh = h.single_step_reset(eh, ah, mode, resident=mirror, **kw)
            assert mirror['valid'] == 1
            _same(ah, ao, f'actions t={t}')
            if mirror['sync']:
                _same(eh, eo, f'state t={t}')
            _cmp(ro, rh, t)
            deaths += int(ro['done'].sum())
            eats += int((ro['reward'] > 0).sum())
            # every fourth step the done envs are left alone: they are stepped again as they are and must come out of the
            # one-env-per-wave code, and keep doing so until they are rebuilt
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
                mirror['valid'] = 0
            if t % 11 == 7:  # another entry point writes the state in between: an eager reset of a few envs
                some = (rng.rand(N) < 0.2).astype(np.uint8)
                o.call = h.call = 100000 + t
                o.single_reset(eo, some, 'none')
                h.single_reset(eh, some, 'none')
                _same(eh, eo, f'eager reset t={t}')
                mirror['valid'] = 0
    assert deaths > 0 and (eats > 0 or N < 10)


@pytest.mark.parametrize('mode', ['partial_2', 'raw', 'partial_3'])
@pytest.mark.parametrize('lazy', [False, True])
def test_abi_long_snakes_and_never_reset(hip, lazy, mode):
    """no reset at all: every env ends up finished and is stepped on by the one-env-per-wave code; before that, snakes
    grow (a greedy walk towards the food) so that queues longer than one word are exercised"""
    N, S, T = 96, 9, 260
    o, h = OracleBackend(seed=5), hip(seed=5)
    eo = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(eo, np.ones(N, np.uint8), 'none')
    eh = eo.copy()
    mirror = {'valid': 0, 'lazy': lazy, 'sync': True}  # (the greedy walk reads the state every step)
    rng = np.random.RandomState(1)
    longest = 0
    prev = None
    for t in range(T):
        # greedy: move towards the food along rows, then columns (actions index the taps: 0 up? — whatever gets closer)
        head = np.argwhere(eo[:, 1] > 0.5)
        hy, hx = np.full(N, -1), np.full(N, -1)
        hy[head[:, 0]], hx[head[:, 0]] = head[:, 1], head[:, 2]
        food = np.argwhere(eo[:, 0] > 0.5)
        fy, fx = np.full(N, -1), np.full(N, -1)
        fy[food[:, 0]], fx[food[:, 0]] = food[:, 1], food[:, 2]
        a = rng.randint(0, 4, size=N)
        best = np.zeros(N, np.int64)
        for i in range(N):
            if hy[i] < 0 or fy[i] < 0:
                best[i] = a[i]
                continue
            body = eo[i, 2]
            cands = []
            for act, (dy, dx) in enumerate(((1, 0), (0, -1), (-1, 0), (0, 1))):  # -TAP[act] (lr_dy / lr_dx)
                y, x = hy[i] + dy, hx[i] + dx
                if 1 <= y <= S - 2 and 1 <= x <= S - 2 and body[y, x] <= 1:
                    cands.append((abs(y - fy[i]) + abs(x - fx[i]), act))
            best[i] = min(cands)[1] if cands else a[i]
        a = np.where(rng.rand(N) < 0.9, best, a).astype(np.int64)
        ao, ah = a.copy(), a.copy()
        late = t >= T - 40
        kw = dict(call=1 + 2 * t, pre_done=None if late else prev, pre_call=2 * t, want_obs_after=True)
        ro = o.single_step_reset(eo, ao, mode, **kw)
        rh = h.single_step_reset(eh, ah, mode, resident=mirror, **kw)
        _same(ah, ao, f'actions t={t}')
        _same(eh, eo, f'state t={t}')
        _cmp(ro, rh, t)
        longest = max(longest, int(eo[:, 2].max()))
        prev = ro['done']
    assert longest >= 18, longest  # more than 16 moves: the queue spills into its second word


def _make(num_envs, mode, **kw):
    from wurm_amd.envs import SingleSnake
    return SingleSnake(num_envs, 9, observation_mode=mode, device='cuda:0', **kw)


def _oracle_follow(env_ids, S, seed):
    from oracle import oracle
    refs = {}
    for gid in env_ids:
        refs[gid] = np.zeros((1, 3, S, S), np.float32)
        oracle.single_reset(refs[gid], np.ones(1, np.uint8), 'none', seed=seed, call=0, env_offset=gid)
    return refs


@pytest.mark.parametrize('form', ['obs', 'no_obs', 'eager', 'mixed'])
def test_host_loop_matches_the_path_without_a_mirror(form):
    """the loop of experiments/main.py:212-227 through the host class with the mirror (WURM_RESIDENT_MIN_ENVS=0) and
    without it (the shipped per-call kernels, themselves checked against the oracle): identical outputs and state,
    also across in-place edits of a state tensor the caller holds, rollouts and eager resets in between"""
    import torch
    N, T, seed = 300, 80, 9
    with knobs(WURM_RESIDENT_MIN_ENVS=0):
        a_env = _make(N, 'partial_2', seed=seed, lazy_reset=(form != 'eager'))
        g = torch.Generator(device='cuda:0').manual_seed(2)
        acts = torch.randint(-1, 5, (T, N), generator=g, device='cuda:0')
        outs_a = []
        alias = None
        for t in range(T):
            a = acts[t].clone()
            obs, r, d, info = a_env.step(a)
            if t == 0:
                assert a_env._mirror is not None and a_env._c.resident_valid == 1
            if form == 'obs' or (form == 'mixed' and t % 3 == 0):
                back = a_env.reset(d)
            elif form == 'mixed' and t % 3 == 1:
                back = a_env.reset(d.clone())  # not the step's own tensor: an eager reset
            else:
                back = a_env.reset(d, return_observations=False)
            outs_a.append([x.clone() for x in (obs, r, d, info['self_collision'], info['edge_collision'], a)] +
                          ([back.clone()] if back is not None else []))
            if t == 20:
                alias = a_env.envs          # from here on the caller can edit the state behind the class's back
            if t in (25, 40):
                alias[3, 0] = 0             # the food of env 3 disappears; version counter bumps
                alias[3, 0, 1 + t % 7, 3] = 1
            if t == 50:
                a_env.rollout(acts[:4].clone())
            if t < 9 and form == 'no_obs':
                assert a_env._c.resident_lazy == 1      # nobody has got hold of the state tensor, or looked at it twice, yet
            outs_a.append([a_env._observe('raw')] if t % 10 == 9 else [])  # (a look at the state that hands out no alias)
    with knobs(WURM_RESIDENT_MIN_ENVS=10 ** 9):
        b_env = _make(N, 'partial_2', seed=seed, lazy_reset=(form != 'eager'))
        k = 0
        for t in range(T):
            a = acts[t].clone()
            obs, r, d, info = b_env.step(a)
            assert b_env._mirror is None
            if form == 'obs' or (form == 'mixed' and t % 3 == 0):
                back = b_env.reset(d)
            elif form == 'mixed' and t % 3 == 1:
                back = b_env.reset(d.clone())
            else:
                back = b_env.reset(d, return_observations=False)
            got = [obs, r, d, info['self_collision'], info['edge_collision'], a] + ([back] if back is not None else [])
            assert len(got) == len(outs_a[k])
            for i, (x, y) in enumerate(zip(outs_a[k], got)):
                assert torch.equal(x, y), f'output {i} of step {t}'
            k += 1
            if t == 20:
                alias = b_env.envs
            if t in (25, 40):
                alias[3, 0] = 0
                alias[3, 0, 1 + t % 7, 3] = 1
            if t == 50:
                b_env.rollout(acts[:4].clone())
            if t % 10 == 9:
                assert torch.equal(outs_a[k][0], b_env._observe('raw')), f'state after step {t}'
            k += 1


def test_host_mirror_stays_current_in_the_plain_loop():
    """nothing but step / deferred reset touches the state: the mirror is built once"""
    import torch
    with knobs(WURM_RESIDENT_MIN_ENVS=0):
        env = _make(256, 'partial_2', seed=1)
        g = torch.Generator(device='cuda:0').manual_seed(2)
        seen = []
        for t in range(150):  # crosses a slab boundary (64 steps)
            a = torch.randint(0, 4, (256,), generator=g, device='cuda:0')
            seen.append(int(env._c.resident_valid))
            obs, r, d, info = env.step(a)
            env.reset(d)
        assert seen[:2] == [0, 0] and all(seen[2:]), seen  # (the first reset(d) that wants its observation runs eagerly)
        env.check_consistency()                      # looks at the state (and flushes the postponed reset): rebuilt next time
        assert env._c.resident_valid == 0
        env.step(torch.zeros(256, dtype=torch.long, device='cuda:0'))
        assert env._c.resident_valid == 1


def test_inference_mode_state_tensor_cannot_be_watched():
    import torch
    with knobs(WURM_RESIDENT_MIN_ENVS=0):
        with torch.inference_mode():
            env = _make(128, 'partial_2', seed=1)
            a = torch.zeros(128, dtype=torch.long, device='cuda:0')
            env.step(a)
            assert env._mirror is not None
            e = env.envs                              # no version counter: edits could not be seen, so no mirror from now on
            assert env._mirror is None and env._c.resident is None
            e[0, 0] = 0
            ref = _make(128, 'partial_2', seed=1)
            ref.step(a.clone())
            ref.envs[0, 0] = 0
            o1, o2 = env.step(a)[0], ref.step(a.clone())[0]
            assert torch.equal(o1, o2)


@pytest.mark.parametrize('lazy', [False, True])
@pytest.mark.parametrize('S,N,mode,T', [
    (12, 70, 'default', 80),
    (12, 33, 'partial_3', 80),
    (14, 40, 'one_channel', 60),     # S^2 not a multiple of four: the unaligned instantiation
    (20, 24, 'positions', 60),
    (25, 9, 'raw', 50),
    (36, 12, 'default', 50),
    (36, 6, 'none', 40),
    (64, 3, 'partial_6', 30),
])
def test_abi_grid_mirror(hip, S, N, mode, T, lazy):
    """12 x 12 and larger: grid_step_kernel keeps its clock grids in the mirror (grid_rollout.hip) — every observation mode,
    eager and lazy, postponed resets, un-reset finished envs (they leave the domain), hand-edited states (the caller says so),
    other entry points in between"""
    rng = np.random.RandomState(S * 100 + N)
    o, h = OracleBackend(seed=13, env_offset=77), hip(seed=13, env_offset=77)
    eo = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(eo, np.ones(N, np.uint8), 'none')
    eh = eo.copy()
    prev = None
    deaths = 0
    mirror = {'valid': 0, 'lazy': lazy}
    for t in range(T):
        a = rng.randint(-3, 9, size=N).astype(np.int64 if t % 2 else np.int32)
        if t % 3:  # mostly straight ahead, so that snakes live long enough to eat now and then
            a = np.where(rng.rand(N) < 0.7, 0, a).astype(a.dtype)
        ao, ah = a.copy(), a.copy()
        kw = dict(call=1 + 2 * t, pre_done=prev, pre_call=2 * t, want_obs_after=(t % 3 != 1))
        mirror['sync'] = not lazy or t % 5 == 4 or t % 9 == 5 or t % 11 == 7 or t == T - 1
        ro = o.single_step_reset(eo, ao, mode, **kw)
        rh = h.single_step_reset(eh, ah, mode, resident=mirror, **kw)
        assert mirror['valid'] == 1
        _same(ah, ao, f'actions t={t}')
        if mirror['sync']:
            _same(eh, eo, f'state t={t}')
        _cmp(ro, rh, t)
        deaths += int(ro['done'].sum())
        prev = ro['done'] if t % 4 != 3 else None
        if t % 9 == 5:
            eo[0, 0, 2, 2] = 1
            eo[1 % N, 0] = 0
            eo[2 % N, 2, 4, 4] = eo[2 % N, 2].max()
            eh[...] = eo
            mirror['valid'] = 0
        if t % 11 == 7:
            some = (rng.rand(N) < 0.2).astype(np.uint8)
            o.call = h.call = 100000 + t
            o.single_reset(eo, some, 'none')
            h.single_reset(eh, some, 'none')
            _same(eh, eo, f'eager reset t={t}')
            mirror['valid'] = 0
    assert deaths > 0


def test_abi_grid_mirror_clock_rebase(hip):
    """the 16-bit clocks of a mirrored grid are re-based before they reach the markers (EX_REBASE = 0xc000 in
    grid_rollout.hip): a record is planted with its clocks just below the threshold and stepped across it"""
    import torch
    S, N = 12, 4
    o, h = OracleBackend(seed=3), hip(seed=3)
    eo = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(eo, np.ones(N, np.uint8), 'none')
    eh = eo.copy()
    mirror = {'valid': 0, 'lazy': True, 'sync': True}
    a0 = np.zeros(N, np.int64)
    ro = o.single_step_reset(eo, a0.copy(), 'default', call=1)
    rh = h.single_step_reset(eh, a0.copy(), 'default', call=1, resident=mirror)
    _cmp(ro, rh, 0)
    # shift every live clock, T and G of the mirror by the same amount: the same state, later on the clock
    iters = (S * S + 255) >> 8
    buf = mirror['buf']
    grids = buf[:N * iters * 512].view(torch.int16).view(N, iters * 256)
    recs = buf[N * iters * 512:].view(torch.int32).view(N, 12)
    shift = 0xc000 - 40
    g32 = grids.to(torch.int32) & 0xffff
    T_ = recs[:, 7].clone()
    live = (g32 > T_[:, None]) & (g32 < 0xfffe)
    g32 = torch.where(live, g32 + shift, g32)
    grids.copy_(g32.to(torch.int16))
    recs[:, 6] += shift
    recs[:, 7] += shift
    prev = ro['done']
    rng = np.random.RandomState(0)
    for t in range(1, 120):
        a = rng.randint(0, 4, size=N).astype(np.int64)
        ao, ah = a.copy(), a.copy()
        kw = dict(call=1 + 2 * t, pre_done=prev, pre_call=2 * t, want_obs_after=True)
        ro = o.single_step_reset(eo, ao, 'default', **kw)
        rh = h.single_step_reset(eh, ah, 'default', resident=mirror, **kw)
        _same(eh, eo, f'state t={t}')
        _cmp(ro, rh, t)
        prev = ro['done']
    assert int((recs[:, 6] & 0xffffffff).max()) < 0xc000 + 8   # G came back down


@pytest.mark.parametrize('mode', ['default', 'partial_3'])
def test_host_loop_grid_mirror_matches_the_path_without(mode):
    import torch
    N, S, T, seed = 96, 16, 70, 4
    res = []
    for min_envs in (0, 10 ** 9):
        with knobs(WURM_RESIDENT_MIN_ENVS=min_envs):
            from wurm_amd.envs import SingleSnake
            env = SingleSnake(N, S, observation_mode=mode, device='cuda:0', seed=seed)
            g = torch.Generator(device='cuda:0').manual_seed(2)
            acts = torch.randint(-1, 5, (T, N), generator=g, device='cuda:0')
            outs = []
            alias = None
            for t in range(T):
                a = acts[t].clone()
                obs, r, d, info = env.step(a)
                if alias is None:   # (while the caller holds the state no reset is postponed — round 5, _alias_free — and a
                    # loop whose every step is followed by an eager reset gives the mirror up: the adaptive rule)
                    assert (env._mirror is not None) == (min_envs == 0)
                # (one reset form per phase: alternating them makes every other reset eager, and a loop that invalidates
                # the mirror every other step loses it — tests/test_host_lazy_reset.py)
                back = env.reset(d) if t >= 35 else env.reset(d, return_observations=False)
                outs.append([x.clone() for x in (obs, r, d, info['self_collision'], info['edge_collision'], a)] +
                            ([back.clone()] if back is not None else []))
                if t % 10 == 9:
                    outs.append([env._observe('raw')])
                if t == 30:
                    alias = env.envs
                if t in (35, 50):
                    alias[3, 0] = 0
                    alias[3, 0, 1 + t % 7, 3] = 1
                if t == 40:
                    env.rollout(acts[:4].clone())
            outs.append([env.envs.clone()])
            res.append(outs)
    assert len(res[0]) == len(res[1])
    for i, (x, y) in enumerate(zip(*res)):
        assert len(x) == len(y)
        for j, (u, v) in enumerate(zip(x, y)):
            assert torch.equal(u, v), (i, j)


def test_a_write_that_bypasses_the_version_counter_is_the_one_documented_deviation():
    """DESIGN.md §7 deviation 9: an in-place write through `tensor.data` (its own version counter), a raw pointer or a DLPack
    consumer between two steps is not seen by the mirror; the same write through the tensor itself is, and without the
    mirror (WURM_RESIDENT_MIN_ENVS=1000000000) there is nothing to see"""
    import torch
    N, seed = 128, 3
    a = torch.zeros(N, dtype=torch.long, device='cuda:0')

    def run(min_envs, bypass):
        with knobs(WURM_RESIDENT_MIN_ENVS=min_envs):
            env = _make(N, 'partial_2', seed=seed)
            env.step(a.clone())
            e = env.envs
            env.step(a.clone())
            if bypass:
                e.data[:, 0] = 0          # every food disappears behind the version counter's back
            else:
                e[:, 0] = 0
            return env.step(a.clone())[0].clone(), env.envs[:, 0].sum().item()

    ref_obs, ref_food = run(10 ** 9, True)            # no mirror: the edit is the state
    seen_obs, seen_food = run(0, False)               # mirror, edit through the tensor: found by the version counter
    assert torch.equal(seen_obs, ref_obs) and seen_food == ref_food
    miss_obs, _ = run(0, True)                        # mirror, edit behind the counter: the step still shows the food
    assert not torch.equal(miss_obs, ref_obs)


@pytest.mark.parametrize('N', [4096, 40000])
def test_at_the_natural_threshold(N):
    """as shipped (mirror from 4096 envs): the oracle follows single envs of the batch by their global id"""
    import torch
    from oracle import oracle
    S, seed, T = 9, 11, 40
    env = _make(N, 'partial_2', seed=seed)
    ids = sorted({0, 1, 15, 16, 31, 32, 63, 64, 65, 4095, N // 2, N - 2, N - 1} |
                 set(int(i) for i in np.random.RandomState(0).randint(0, N, size=20)))
    refs = _oracle_follow(ids, S, seed)
    g = torch.Generator(device='cuda:0').manual_seed(3)
    actions = torch.randint(4, (T, N), generator=g, device='cuda:0')
    a_host = actions.cpu().numpy()
    call = 1
    for t in range(T):
        a = actions[t].clone()
        obs, r, d, info = env.step(a)
        assert env._mirror is not None and (t > T // 2 or env._c.resident_lazy == 1)
        back = env.reset(d, return_observations=(t > T // 2 and t % 3 == 0))
        sub = [x[ids].cpu().numpy() for x in (obs, r, d, info['self_collision'], info['edge_collision'], a)]
        back_sub = back[ids].cpu().numpy() if back is not None else None
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
            if back_sub is not None:
                _same(back_sub[j:j + 1], bo, f'reset obs env {gid} t={t}')
        call += 2
    state = env.envs[ids].cpu().numpy()
    _same(state, np.concatenate([refs[g_] for g_ in ids]), 'final state')


def test_per_call_parity_suite_on_the_mirror():
    if os.environ.get('WURM_RESIDENT_MIN_ENVS') == '0':
        pytest.skip('already inside the forced run')
    env = dict(os.environ, WURM_RESIDENT_MIN_ENVS='0')
    # (round 6: without tests/test_fuzz_gpu.py — its own families force these routes themselves, and re-running all of it under
    # each forced knob was 300 of the GPU suite's 500 seconds)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                        'tests/test_hip_vs_oracle.py', 'tests/test_hip_fused_step.py', 'tests/test_kat_single_snake.py',
                        'tests/test_rl_gpu.py', 'tests/test_hip_golden.py'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_check_masks_of_the_step_launch_equal_the_checker(hip):
    """wurm_single_call.check_mask (round 4): the masks the resident step launch writes == wurm_single_check / the oracle's
    checker on the stepped state for every live env, 'not computed' (-1) for the envs that finished in the step; an env
    without food shows WURM_CHK_ONE_FOOD; SingleSnake.check_consistency(~done) is served by them and raises as the
    checker does"""
    import torch
    from wurm_amd.envs import SingleSnake
    from oracle import oracle as _orc
    N, T = 300, 40
    env = SingleSnake(N, 9, observation_mode='partial_2', device='cuda:0', seed=3, resident_mirror=True)
    g = torch.Generator().manual_seed(1)
    st = env.envs
    st[7, 0] = 0                      # env 7 loses its food (hand-edited: the mirror is rebuilt from the tensor)
    served = 0
    hungry = True                     # env 7 has no food until it finishes once and is rebuilt
    for t in range(T):
        a = torch.randint(4, (N,), generator=g).cuda()
        if hungry:
            a[7] = 3 if t % 2 == 0 else 1   # (keep it alive: back and forth — a reversal becomes a forward move)
        _, _, d, _ = env.step(a)
        live = ~d.squeeze(-1)
        hungry = hungry and bool(live[7])
        if t == 0:
            env.check_consistency(live & (torch.arange(N, device='cuda') != 7))   # arms the request (runs the checker once)
        else:
            chk = env._chk.clone()
            before = env._fs.steps
            if hungry:
                with pytest.raises(RuntimeError, match='exactly one food'):
                    env.check_consistency(live)
            env.check_consistency(live & (torch.arange(N, device='cuda') != 7))
            assert env._chk_void_at < before, 'the masks were not used (the state was touched)'
            served += 1
            ref = _orc.single_check(env.envs.cpu().numpy())        # (reading env.envs voids the masks: after the checks)
            lv = live.cpu().numpy()
            got = chk.cpu().numpy()
            assert (got[~lv] == -1).all(), 'finished envs must be "not computed"'
            assert np.array_equal(got[lv].astype(np.uint32), ref[lv]), f't={t}'
            assert not hungry or got[7] == _orc.CHK_ONE_FOOD
        env.reset(d)
    assert served >= T - 2
