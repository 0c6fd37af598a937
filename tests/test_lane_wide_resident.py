"""GPU parity for the per-call step of 10 x 10 and 11 x 11 SingleSnake batches on their resident mirror
(`lane_wide_resident_step_kernel`, wurm_amd/csrc/lane_wide_resident.hpp: 48 bytes per env, lazy form only): the C ABI
(`wurm_single_step_reset` with `resident`) against the oracle's fused step — postponed resets, `obs_after`, every observation
the mirror serves, both envs-per-wave settings, ragged batches, hostile actions, envs left un-reset (stepped again by the
one-env-per-wave code inside the launch), hand-edited states and other entry points in between (the mirror rebuilt), long
snakes — and the Python class with and without the mirror.
Loop being matched: /root/reference experiments/main.py:212-227 over wurm/envs/single_snake.py:197-342."""
import numpy as np
import pytest

from tests.backends import OracleBackend
from tests.test_lane_resident import _cmp, _same

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


def _route():
    from wurm_amd import _lib
    return _lib.lib().wurm_single_last_route().decode()


@pytest.mark.parametrize('S', [10, 11])
@pytest.mark.parametrize('epw', [0, 16, 32])
@pytest.mark.parametrize('N,mode,T', [
    (200, 'partial_2', 90),   # whole blocks and a ragged one at every envs-per-wave setting
    (131, 'partial_2', 70),   # an odd count: the crops of the ragged block go out float by float
    (64, 'none', 60),
    (3, 'partial_2', 50),
    (137, 'one_channel', 60),
    (70, 'default', 50),
    (96, 'default', 40),
    (90, 'positions', 50),
    (137, 'partial_3', 60),
    (64, 'partial_3', 45),
])
def test_abi_step_postponed_reset_and_obs_after(hip, S, epw, N, mode, T):
    from wurm_amd._lib import knobs
    rng = np.random.RandomState(N + epw + S)
    o, h = OracleBackend(seed=31, env_offset=500), hip(seed=31, env_offset=500)
    eo = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(eo, np.ones(N, np.uint8), 'none')
    eh = eo.copy()
    prev = None
    deaths = eats = 0
    mirror = {'valid': 0, 'lazy': True}
    with knobs(WURM_RESIDENT_EPW=epw or None):
        for t in range(T):
            a = rng.randint(-3, 9, size=N).astype(np.int64 if t % 2 else np.int32)  # hostile values included
            ao, ah = a.copy(), a.copy()
            kw = dict(call=1 + 2 * t, pre_done=prev, pre_call=2 * t, want_obs_after=(t % 3 != 1))
            # `envs` is written out of the mirror only now and then — and before the edits / other entry points below
            mirror['sync'] = t % 5 == 4 or t % 9 == 5 or t % 11 == 7 or t == T - 1
            ro = o.single_step_reset(eo, ao, mode, **kw)
            rh = h.single_step_reset(eh, ah, mode, resident=mirror, **kw)
            assert mirror['valid'] == 1 and _route() == 'lane_wide_resident'
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


@pytest.mark.parametrize('S', [10, 11])
def test_a_call_that_wants_envs_written_leaves_the_mirror_stale(hip, S):
    """resident_lazy = 0: the step runs without the mirror (envs written every call, as before) and the mirror is not kept —
    alternating with lazy calls that rebuild it"""
    N, mode, T = 150, 'partial_2', 40
    rng = np.random.RandomState(3)
    o, h = OracleBackend(seed=8), hip(seed=8)
    eo = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(eo, np.ones(N, np.uint8), 'none')
    eh = eo.copy()
    mirror = {'valid': 0, 'lazy': True, 'sync': True}
    prev = None
    for t in range(T):
        mirror['lazy'] = (t // 5) % 2 == 0
        a = rng.randint(0, 4, size=N).astype(np.int64)
        ao, ah = a.copy(), a.copy()
        kw = dict(call=1 + 2 * t, pre_done=prev, pre_call=2 * t, want_obs_after=True)
        ro = o.single_step_reset(eo, ao, mode, **kw)
        rh = h.single_step_reset(eh, ah, mode, resident=mirror, **kw)
        assert mirror['valid'] == int(mirror['lazy'])
        assert (_route() == 'lane_wide_resident') == mirror['lazy']
        _same(eh, eo, f'state t={t}')
        _cmp(ro, rh, t)
        prev = ro['done']


@pytest.mark.parametrize('S', [10, 11])
@pytest.mark.parametrize('mode', ['partial_2', 'default'])
def test_abi_long_snakes_and_never_reset(hip, S, mode):
    """no reset at all: every env ends up finished and is stepped on by the one-env-per-wave code; before that, snakes grow (a
    greedy walk towards the food) so that queues longer than one word are exercised; the state is written out of the mirror
    and compared every step (the self collision that adds the head's value on top of a segment included)"""
    N, T = 96, 300
    o, h = OracleBackend(seed=5), hip(seed=5)
    eo = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(eo, np.ones(N, np.uint8), 'none')
    eh = eo.copy()
    mirror = {'valid': 0, 'lazy': True, 'sync': True}  # (the greedy walk reads the state every step)
    rng = np.random.RandomState(1)
    longest = 0
    prev = None
    for t in range(T):
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
            for act, (dy, dx) in enumerate(((1, 0), (0, -1), (-1, 0), (0, 1))):  # -TAP[act]
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
    assert longest >= 18, longest


@pytest.mark.parametrize('S,mode', [(10, 'partial_2'), (11, 'default'), (10, 'one_channel'), (11, 'positions')])
@pytest.mark.parametrize('form', ['obs', 'no_obs', 'mixed'])
def test_host_loop_matches_the_path_without_a_mirror(S, mode, form):
    """the Python class: `env.step(a); env.reset(d)` with the mirror (lazy until something looks at env.envs) against the same
    object without one — every tensor returned, and the state at the end"""
    import torch
    from wurm_amd.envs import SingleSnake
    dev = torch.device('cuda:0')
    N, T = 300, 60
    g = torch.Generator().manual_seed(S)
    acts = torch.randint(-1, 5, (T, N), generator=g).to(dev)
    outs = []
    for policy in (True, False):
        env = SingleSnake(N, S, observation_mode=mode, device=dev, seed=9, resident_mirror=policy)
        rec = []
        for t in range(T):
            o, r, d, info = env.step(acts[t].clone())
            rec += [o.clone(), r.clone(), d.clone(), info['self_collision'].clone(), info['edge_collision'].clone()]
            keep = form == 'obs' or (form == 'mixed' and t % 3 == 0)
            ro = env.reset(d, return_observations=keep)
            if keep:
                rec.append(ro.clone())
            if form == 'mixed' and t % 17 == 16:
                rec.append(env.envs.clone())       # a look: the lazy mirror is written out (and the object turns eager)
        rec.append(env.envs.clone())
        if policy is True:
            st = env.mirror_state()
            assert st['state'] in ('lazy', 'eager')
        outs.append(rec)
    assert len(outs[0]) == len(outs[1])
    for i, (x, y) in enumerate(zip(*outs)):
        assert x.dtype == y.dtype and torch.equal(x, y), f'record {i}'


def test_at_the_natural_threshold():
    """from 4 096 envs on the class asks for the mirror by itself (wurm_single_resident_bytes), and the step runs on it"""
    import torch
    from wurm_amd import _lib
    from wurm_amd.envs import SingleSnake
    dev = torch.device('cuda:0')
    assert _lib.lib().wurm_single_resident_bytes(_lib.i64(4096), 10, *_lib.parse_obs_mode('partial_2')) == 4096 * 48
    assert _lib.lib().wurm_single_resident_bytes(_lib.i64(4095), 11, *_lib.parse_obs_mode('default')) == 0
    assert _lib.lib().wurm_single_resident_size(_lib.i64(5), 11, *_lib.parse_obs_mode('raw')) == 0
    N, T = 4096 + 7, 25
    g = torch.Generator().manual_seed(1)
    acts = torch.randint(4, (T, N), generator=g).to(dev)
    outs = []
    for policy in (None, False):
        env = SingleSnake(N, 10, observation_mode='partial_2', device=dev, seed=2, resident_mirror=policy)
        rec = []
        for t in range(T):
            o, r, d, _ = env.step(acts[t].clone())
            if policy is None:
                assert _route() == 'lane_wide_resident'
            rec += [o.clone(), r.clone(), d.clone()]
            env.reset(d, return_observations=False)
        rec.append(env.envs.clone())
        outs.append(rec)
    for i, (x, y) in enumerate(zip(*outs)):
        assert torch.equal(x, y), f'record {i}'


def test_check_masks_of_the_step_launch_equal_the_checker(hip):
    """check_consistency() served from the step launch's mask on the mirror (an env in the kernel's domain is a well-formed
    snake) agrees with the checker on the written-out state"""
    import torch
    from wurm_amd.envs import SingleSnake
    dev = torch.device('cuda:0')
    N, T = 500, 40
    env = SingleSnake(N, 11, observation_mode='partial_2', device=dev, seed=3, resident_mirror=True)
    g = torch.Generator().manual_seed(5)
    acts = torch.randint(4, (T, N), generator=g).to(dev)
    for t in range(T):
        o, r, d, _ = env.step(acts[t].clone())
        env.check_consistency(~d.squeeze(-1))
        env.reset(d, return_observations=False)
    env.check_consistency()


@pytest.mark.parametrize('S,mode', [(10, 'default'), (11, 'partial_2')])
def test_rollouts_between_per_call_steps_write_the_lazy_mirror_out(S, mode):
    """SingleSnake.rollout of these sizes runs on the planes (lane_wide.hpp has no mirror-keeping form): the entry point writes a
    lazy mirror out first and reports it stale, the next step rebuilds it — against the same object without a mirror"""
    import torch
    from wurm_amd.envs import SingleSnake
    dev = torch.device('cuda:0')
    N = 200
    g = torch.Generator().manual_seed(7)
    plan = [torch.randint(4, (6, N), generator=g).to(dev) for _ in range(5)]
    outs = []
    for policy in (True, False):
        env = SingleSnake(N, S, observation_mode=mode, device=dev, seed=4, resident_mirror=policy)
        rec = []
        for tape in plan:
            for t in range(3):
                o, r, d, _ = env.step(tape[t].clone())
                if policy is True:
                    assert _route() == 'lane_wide_resident'
                env.reset(d, return_observations=False)
                rec += [o.clone(), r.clone(), d.clone()]
            out = env.rollout(tape.clone())
            rec += [out['observations'].clone(), out['rewards'].clone(), out['dones'].clone()]
        rec.append(env.envs.clone())
        outs.append(rec)
    for i, (x, y) in enumerate(zip(*outs)):
        assert torch.equal(x, y), f'record {i}'


def test_deepcopy_and_pickle_with_the_mirror():
    """a copied / unpickled env object continues exactly like the original (the mirror is rebuilt or copied, never shared)"""
    import copy
    import pickle
    import torch
    from wurm_amd.envs import SingleSnake
    dev = torch.device('cuda:0')
    N = 150
    env = SingleSnake(N, 11, observation_mode='one_channel', device=dev, seed=6, resident_mirror=True)
    g = torch.Generator().manual_seed(2)
    acts = torch.randint(4, (30, N), generator=g).to(dev)
    for t in range(10):
        o, r, d, _ = env.step(acts[t].clone())
        env.reset(d, return_observations=False)
    twins = [copy.deepcopy(env), pickle.loads(pickle.dumps(env))]
    for t in range(10, 30):
        ref = env.step(acts[t].clone())
        env.reset(ref[2], return_observations=False)
        for tw in twins:
            got = tw.step(acts[t].clone())
            tw.reset(got[2], return_observations=False)
            for a, b in zip(ref[:3], got[:3]):
                assert torch.equal(a, b), f't={t}'
    for tw in twins:
        assert torch.equal(env.envs, tw.envs)
