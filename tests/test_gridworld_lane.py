"""SimpleGridworld rollouts of large batches: one env per lane (wurm_amd/csrc/gridworld_lane.hip) against the oracle, which
follows /root/reference wurm/envs/simple_gridworld.py:135-202 (step), :111-133 (_observe), :225-268 (reset).  The route
is forced with WURM_LANE_ROLLOUT_MIN_ENVS = 0 (it takes over from 6144 envs by default) and checked with
wurm_single_last_route().  Bit-exact: bytes, indices and floats that are 0.0 or 1.0."""
import numpy as np
import pytest

from tests.backends import OracleBackend

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


def _route():
    from wurm_amd import _lib
    return _lib.lib().wurm_single_last_route().decode()


def _same(a, b, what):
    if a is None:
        assert b is None, what
        return
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, what
    if a.dtype == np.float32:
        a, b = a.view(np.uint32), b.view(np.uint32)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        raise AssertionError(f'{what}: {len(bad)} elements differ, first at {bad[0].tolist()}: oracle {a[tuple(bad[0])]} '
                             f'hip {b[tuple(bad[0])]}; envs {sorted(set(bad[:, 1].tolist()))[:10] if bad.shape[1] > 1 else ""}')


def _fresh(o, N, S, start):
    envs = np.zeros((N, 2, S, S), np.float32)
    o.grid_reset(envs, np.ones(N, np.uint8), start, 'none')
    return envs


def _both(hip, envs, actions, start, mode, seed=8, call=40, offset=0, min_envs=0, row='gridworld_lane'):
    from wurm_amd._lib import knobs
    o, h = OracleBackend(seed=seed), hip(seed=seed)
    o.call = h.call = call
    o.env_offset = h.env_offset = offset
    eo, eh, ao, ah = envs.copy(), envs.copy(), actions.copy(), actions.copy()
    with knobs(WURM_LANE_ROLLOUT_MIN_ENVS=min_envs):
        rh = h.grid_rollout(eh, ah, start, mode)
        assert _route() == row
    ro = o.grid_rollout(eo, ao, start, mode)
    for k in ro:
        _same(ro[k], rh[k], k)
    _same(eo, eh, 'final state')
    _same(ao, ah, 'actions')
    return ro


@pytest.mark.parametrize('mode', ['default', 'raw', 'positions', 'none'])
@pytest.mark.parametrize('N,S,T', [(64, 9, 40), (131, 9, 33), (70, 5, 25), (257, 7, 20), (33, 12, 30), (3, 30, 50),
                                   (65, 5, 9)])
def test_lane_rollout_matches_the_oracle(hip, N, S, T, mode):
    """ragged and odd batches (an odd N puts every second step's observation block off the 16-byte boundary), sizes from
    the smallest the reference resets (5, simple_gridworld.py:249-250) to 30"""
    rng = np.random.RandomState(N * 31 + S)
    start = (S // 2, S // 2)
    o = OracleBackend(seed=8)
    envs = _fresh(o, N, S, start)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    ro = _both(hip, envs, actions, start, mode)
    assert ro['done'].sum() > 0 or S > 12


@pytest.mark.parametrize('mode', ['default', 'raw'])
@pytest.mark.parametrize('S,epw', [(9, 4), (9, 8), (9, 16), (9, 32), (9, 64), (12, 64), (20, 32), (7, 64)])
def test_every_envs_per_wave_of_the_image_modes(hip, S, epw, mode):
    """WURM_GRIDWORLD_LANE_EPW pins the envs per wave: runs that fit the 16 KB byte slab are composed in LDS, the others
    (12 x 12 at 64 envs per wave, 20 x 20 at 32) take the fill-and-patch form; ragged and odd batches in both"""
    from wurm_amd._lib import knobs
    N, T = 3 * epw + 5, 18
    rng = np.random.RandomState(S * 64 + epw)
    start = (S // 2, S // 2)
    envs = _fresh(OracleBackend(seed=8), N, S, start)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    with knobs(WURM_GRIDWORLD_LANE_EPW=epw):
        _both(hip, envs, actions, start, mode, seed=5, call=3)


@pytest.mark.parametrize('dtype', [np.int32, np.int64])
def test_action_dtypes_and_negative_actions(hip, dtype):
    """actions outside 0..3 wrap like the reference's `actions % 4` on the tensor would — the C-ABI takes them as they
    come (oracle/single_snake.c load_action)"""
    rng = np.random.RandomState(5)
    N, S, T = 100, 9, 24
    envs = _fresh(OracleBackend(seed=8), N, S, (4, 4))
    actions = rng.randint(0, 4, size=(T, N)).astype(dtype)
    _both(hip, envs, actions, (4, 4), 'default')


def test_start_location_off_centre_and_on_the_border(hip):
    rng = np.random.RandomState(6)
    N, S, T = 96, 9, 30
    for start in [(1, 7), (6, 2)]:
        envs = _fresh(OracleBackend(seed=8), N, S, start)
        actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
        _both(hip, envs, actions, start, 'default', seed=11, call=6)


def test_env_offset_and_call_counter_enter_the_draws(hip):
    rng = np.random.RandomState(7)
    N, S, T = 80, 9, 30
    envs = _fresh(OracleBackend(seed=8), N, S, (4, 4))
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    a = _both(hip, envs, actions, (4, 4), 'raw', seed=3, call=2 ** 33 + 5, offset=10 ** 7)
    b = _both(hip, envs, actions, (4, 4), 'raw', seed=3, call=2 ** 33 + 7, offset=10 ** 7)
    assert not np.array_equal(a['obs'], b['obs'])


@pytest.mark.parametrize('mode', ['default', 'raw', 'positions'])
def test_hand_made_envs_fall_back_to_the_generic_kernel(hip, mode):
    """envs outside the lane kernel's domain (two foods, no agent, other values, food under the agent) are rolled out by
    the one-env-per-wave kernel in a second launch; their neighbours in the same wave are not disturbed.  (Two agents in
    one env or a food of another value than 1 are outside the library's domain altogether: DESIGN.md, deviation 7.)"""
    rng = np.random.RandomState(9)
    N, S, T = 150, 9, 20
    envs = _fresh(OracleBackend(seed=8), N, S, (4, 4))
    envs[10, 0, 6, 6] = 1         # a second food (unless it was there)
    envs[64, 1] = 0               # no agent
    envs[65, 0] = 0               # no food: inside the domain
    envs[130, 1] *= 0.5           # an agent of value 0.5
    envs[140, 0] = envs[140, 1]   # the food under the agent
    envs[149, 0, 0, 3] = 1        # a food on the border ring as well
    envs[20, 0] = 0
    envs[20, 0, 3, 4] = 1         # the only food right above the agent, and (in the domain)
    envs[21, 0] = 0
    envs[21, 0, 0, 4] = 1         # the only food on the border ring (in the domain: invisible in 'default', eaten while dying)
    envs[21, 1] = 0
    envs[21, 1, 1, 4] = 1
    envs[30, 1] = 0
    envs[30, 1, 0, 0] = 1         # the agent in a corner (in the domain)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    _both(hip, envs, actions, (4, 4), mode)
    for a in range(4):
        actions[0, 20:22] = a
        _both(hip, envs, actions, (4, 4), mode)


def test_rollout_equals_the_loop(hip):
    """T fused iterations == T x (step; reset(done)) through the per-call entry points (which stay one env per wave)"""
    rng = np.random.RandomState(12)
    N, S, T, start = 90, 9, 25, (4, 4)
    from wurm_amd._lib import knobs
    h1, h2 = hip(seed=4), hip(seed=4)
    envs = _fresh(OracleBackend(seed=4), N, S, start)
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    e1, e2 = envs.copy(), envs.copy()
    h1.call = h2.call = 10
    with knobs(WURM_LANE_ROLLOUT_MIN_ENVS=0):
        r = h1.grid_rollout(e1, actions.copy(), start, 'default')
        assert _route() == 'gridworld_lane'
    for t in range(T):
        obs, rew, done, ec = h2.grid_step(e2, actions[t].copy(), 'default')
        _same(r['obs'][t], obs, f'obs {t}')
        _same(r['reward'][t], rew, f'reward {t}')
        _same(r['done'][t], done, f'done {t}')
        h2.grid_reset(e2, done, start, 'none')
    _same(e1, e2, 'final state')


def test_below_the_threshold_and_recorded_outcomes_stay_generic(hip):
    rng = np.random.RandomState(13)
    N, S, T = 70, 9, 10
    envs = _fresh(OracleBackend(seed=8), N, S, (4, 4))
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    _both(hip, envs, actions, (4, 4), 'default', min_envs=1 << 40, row='generic')


def test_a_large_batch_at_the_default_threshold(hip):
    """65536 x 9 x 9, the driver's `rollout_65536x9_gridworld_default` shape, with the knobs at their defaults"""
    rng = np.random.RandomState(14)
    N, S, T = 65536, 9, 8
    envs = _fresh(OracleBackend(seed=8), N, S, (4, 4))
    actions = rng.randint(0, 4, size=(T, N)).astype(np.int64)
    from wurm_amd import _lib
    assert _lib.lib().wurm_get_option(b'WURM_LANE_ROLLOUT_MIN_ENVS') <= N
    _both(hip, envs, actions, (4, 4), 'default', min_envs=_lib.lib().wurm_get_option(b'WURM_LANE_ROLLOUT_MIN_ENVS'))


# ---- the per-call step of large batches (gridworld_lane_step_kernel): wurm_grid_step_reset's deferred form and the plain step

def _cmp_step(ro, rh, t):
    for k in ro:
        _same(ro[k], rh[k], f'{k} t={t}')


@pytest.mark.parametrize('mode', ['default', 'raw', 'positions', 'none'])
@pytest.mark.parametrize('N,S,epw', [(200, 9, -1), (131, 9, 32), (65, 9, 64), (70, 5, 4), (33, 12, 16), (9, 30, -1), (257, 7, 8), (6, 40, -1)])
def test_per_call_step_with_the_postponed_reset(hip, N, S, epw, mode):
    """`step(a); reset(done)` as one launch per iteration in the deferred form: the previous call's done flags rebuild envs
    in front of the step (pre_call), the reset observation of THIS call's finished envs comes back as obs_after (call + 1)
    without being stored; every third call without obs_after, every fourth without the postponed reset (finished envs are
    stepped again: an agent on the ring walks on, or off the grid and vanishes — that env goes to the one-env-per-wave kernel
    from then on), hand-made states in between"""
    from wurm_amd._lib import knobs
    T = 60
    rng = np.random.RandomState(N + S)
    start = (S // 2, S // 2)
    o, h = OracleBackend(seed=12, env_offset=77), hip(seed=12, env_offset=77)
    eo = _fresh(o, N, S, start)
    eh = eo.copy()
    prev = None
    deaths = lane_calls = 0
    with knobs(WURM_LANE_STEP_MIN_ENVS=0, WURM_GRIDWORLD_LANE_EPW=epw):
        for t in range(T):
            a = rng.randint(-3, 9, size=N).astype(np.int64 if t % 2 else np.int32)
            ao, ah = a.copy(), a.copy()
            kw = dict(call=1 + 2 * t, pre_done=prev, pre_call=2 * t, want_obs_after=(t % 3 != 1), grid=start)
            ro = o.single_step_reset(eo, ao, mode, **kw)
            rh = h.single_step_reset(eh, ah, mode, **kw)
            lane_calls += _route() == 'gridworld_lane_step'
            _same(ah, ao, f'actions t={t}')
            _same(eh, eo, f'state t={t}')
            _cmp_step(ro, rh, t)
            deaths += int(ro['done'].sum())
            prev = ro['done'] if t % 4 != 3 else None
            if t % 9 == 5:   # hand-made states: two foods, no food, the food under the agent, the only food on the ring
                eo[0, 0, 1, 1] = 1
                eo[1 % N, 0] = 0
                eo[2 % N, 0] = eo[2 % N, 1]
                eo[N - 1, 0] = 0
                eo[N - 1, 0, 0, 2] = 1
                eh[...] = eo
    assert deaths > 0 or S > 12
    # (round 5's byte slab did not fit 40 x 40 'default' — 4 x 19 200 bytes — and those calls stayed with the one-env-per-wave
    # kernel; round 6 composes the run as one BIT per float: every size up to 64 x 64 fits)
    assert lane_calls == T


def test_per_call_plain_step_and_the_immediate_form(hip):
    """the plain wurm_grid_step (no reset in the call) takes the lane kernel too; the immediate form (post_reset: the rebuilt
    state is stored by the same launch) stays with the one-env-per-wave kernel"""
    from wurm_amd._lib import knobs
    N, S, start = 150, 9, (4, 4)
    rng = np.random.RandomState(4)
    o, h = OracleBackend(seed=3), hip(seed=3)
    eo = _fresh(o, N, S, start)
    eh = eo.copy()
    o.call = h.call = 1
    with knobs(WURM_LANE_STEP_MIN_ENVS=0):
        for t in range(30):
            a = rng.randint(0, 4, size=N).astype(np.int64)
            ro, rh = o.grid_step(eo, a.copy(), 'default'), h.grid_step(eh, a.copy(), 'default')
            assert _route() == 'gridworld_lane_step'
            for x, y in zip(ro, rh):
                _same(x, y, f'plain step t={t}')
            _same(eo, eh, f'state t={t}')
            o.call = h.call = 1000 + t
            o.grid_reset(eo, ro[2], start, 'none'); h.grid_reset(eh, rh[2], start, 'none')
        for t in range(20):
            a = rng.randint(0, 4, size=N).astype(np.int64)
            kw = dict(call=5000 + 2 * t, post_reset=True, want_obs_after=True, grid=start)
            ro = o.single_step_reset(eo, a.copy(), 'raw', **kw)
            rh = h.single_step_reset(eh, a.copy(), 'raw', **kw)
            assert _route() == 'generic'
            _cmp_step(ro, rh, t)
            _same(eo, eh, f'state t={t}')
    kw = dict(call=9001, pre_done=None, pre_call=9000, want_obs_after=True, grid=start)   # below the threshold: generic
    h.single_step_reset(eh, rng.randint(0, 4, size=N).astype(np.int64), 'default', **kw)
    assert _route() == 'generic'


def test_class_loop_at_the_default_threshold():
    """SimpleGridworld(16 384 envs) through the unchanged per-call loop `env.step(a); env.reset(done)` and a rollout: the C
    step machine's launches take the lane kernel (12 288 envs and more) — compared with the same loop on the
    one-env-per-wave kernels (themselves compared with the oracle above and in tests/test_hip_fused_step.py): identical
    outputs and state, also across an in-place edit of the state the caller holds"""
    import torch
    from wurm_amd import _lib
    from wurm_amd._lib import knobs
    from wurm_amd.envs import SimpleGridworld
    N, S, T, start, seed = 16384, 9, 40, (4, 4), 21
    g = torch.Generator(device='cuda:0').manual_seed(3)
    acts = torch.randint(0, 4, (T, N), generator=g, device='cuda:0')

    def loop(min_envs, want_route):
        routes, outs = set(), []
        with knobs(WURM_LANE_STEP_MIN_ENVS=min_envs, WURM_LANE_ROLLOUT_MIN_ENVS=min_envs):
            env = SimpleGridworld(N, S, start_location=start, observation_mode='default', device='cuda:0', seed=seed)
            for t in range(T):
                obs, r, d, info = env.step(acts[t].clone())
                routes.add(_lib.lib().wurm_single_last_route().decode())
                back = env.reset(d) if t % 3 else env.reset(d, return_observations=False)
                outs.append([x.clone() for x in (obs, r, d, info['edge_collision'])] + ([back.clone()] if back is not None else []))
                if t == 17:
                    e = env.envs
                    e[5, 0] = 0
                    e[5, 0, 2, 6] = 1        # the food of env 5 moves; version counter bumps
                    del e
                if t == 25:
                    ro = env.rollout(acts[:3].clone())
                    outs.append([ro['observations'].clone(), ro['rewards'].clone(), ro['dones'].clone()])
            outs.append([env.envs.clone()])
        assert routes == want_route, routes
        return outs

    a, b = loop(None, {'gridworld_lane_step'}), loop(1 << 40, {'generic'})
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        assert len(x) == len(y)
        for i, (u, v) in enumerate(zip(x, y)):
            assert torch.equal(u, v), f'record {k} output {i}'


# ---- round 6: the per-call step on the caller's mirror (wurm_grid_resident_bytes: one 32-bit record per env) ----

@pytest.mark.parametrize('mode', ['default', 'raw', 'positions', 'none'])
@pytest.mark.parametrize('N,S,lazy', [(200, 9, True), (131, 9, False), (70, 5, True), (33, 12, True), (9, 30, False), (257, 7, True)])
def test_per_call_step_on_the_mirror(hip, N, S, lazy, mode):
    """`step(a); reset(done)` on SimpleGridworld's mirror, against the oracle every call: the launch that builds the mirror scans
    the planes (two launches), every later one reads the records and is ONE launch (wurm_launch_count), lazy or eager; every
    fourth call comes without the postponed reset, so finished envs are stepped again and agents walk off the grid — an env
    without an agent is in the lane kernel's domain now (simple_gridworld.py:153-162 on an all-zero head plane)"""
    from wurm_amd import _lib
    from wurm_amd._lib import knobs
    T = 48
    rng = np.random.RandomState(N * 7 + S)
    start = (S // 2, S // 2)
    o, h = OracleBackend(seed=12, env_offset=77), hip(seed=12, env_offset=77)
    eo = _fresh(o, N, S, start)
    eh = eo.copy()
    res = {'lazy': lazy, 'sync': True}
    prev = None
    count = _lib.lib().wurm_launch_count
    gone = 0
    with knobs(WURM_LANE_STEP_MIN_ENVS=0):
        for t in range(T):
            a = rng.randint(-3, 9, size=N).astype(np.int64 if t % 2 else np.int32)
            ao, ah = a.copy(), a.copy()
            kw = dict(call=1 + 2 * t, pre_done=prev, pre_call=2 * t, want_obs_after=(t % 3 != 1), grid=start)
            ro = o.single_step_reset(eo, ao, mode, **kw)
            n0, was = count(), res.get('valid', 0)
            rh = h.single_step_reset(eh, ah, mode, resident=res, **kw)
            assert _route() == 'gridworld_lane_step' and res['valid'] == 1
            # (the flush of a lazy mirror the harness asks for is a launch of its own)
            assert count() - n0 == (1 if was == 1 else 2) + (1 if lazy else 0), (t, was, count() - n0)
            _same(ah, ao, f'actions t={t}')
            _same(eh, eo, f'state t={t}')
            _cmp_step(ro, rh, t)
            gone += int((eo[:, 1].reshape(N, -1).sum(1) == 0).sum())
            prev = ro['done'] if t % 4 != 3 else None
    assert gone > 0 or S > 12   # agents did walk off


def test_hand_made_states_refuse_the_mirror(hip):
    """an env outside the lane kernel's domain (two foods; the food under the agent; no food with the agent on the ring) in the planes the
    mirror is built from: the call is served as ever (lane kernel + the one-env-per-wave kernel for that env), returns
    WURM_MIRROR_REFUSED, and the mirror stays unused — results equal to the oracle's throughout — until the caller clears
    resident_valid after repairing the state, when it is built and used"""
    from wurm_amd import _lib
    from wurm_amd._lib import knobs
    N, S, start, mode = 150, 9, (4, 4), 'default'
    rng = np.random.RandomState(2)
    o, h = OracleBackend(seed=5), hip(seed=5)
    eo = _fresh(o, N, S, start)
    eo[3, 0, 2, 2] = 1; eo[3, 0, 6, 6] = 1          # two foods
    eo[40, 0] = eo[40, 1]                            # the food under the agent
    eo[77, 0] = 0; eo[77, 0, 1, 1] = 1; eo[77, 0, 7, 7] = 1   # two foods again, elsewhere
    eh = eo.copy()
    res = {'lazy': True, 'sync': False}
    prev = None
    count = _lib.lib().wurm_launch_count
    with knobs(WURM_LANE_STEP_MIN_ENVS=0):
        for t in range(12):
            a = rng.randint(0, 4, size=N).astype(np.int64)
            kw = dict(call=1 + 2 * t, pre_done=prev, pre_call=2 * t, want_obs_after=True, grid=start)
            ro = o.single_step_reset(eo, a.copy(), mode, **kw)
            n0 = count()
            rh = h.single_step_reset(eh, a.copy(), mode, resident=res, **kw)
            assert res['valid'] == 2 and count() - n0 == 2
            _same(eh, eo, f'state t={t}')
            _cmp_step(ro, rh, t)
            prev = ro['done']
        for e in (eo, eh):                            # the caller repairs the three envs and says so
            for i in (3, 40, 77):
                e[i] = 0
                e[i, 1, 4, 4] = 1
                e[i, 0, 2, 5] = 1
        res['valid'] = 0
        for t in range(12, 30):
            a = rng.randint(0, 4, size=N).astype(np.int64)
            kw = dict(call=1 + 2 * t, pre_done=prev, pre_call=2 * t, want_obs_after=True, grid=start)
            ro = o.single_step_reset(eo, a.copy(), mode, **kw)
            n0 = count()
            res['sync'] = t % 5 == 4                 # (lazy: the planes are written out only when somebody looks)
            rh = h.single_step_reset(eh, a.copy(), mode, resident=res, **kw)
            assert res['valid'] == 1 and count() - n0 == (2 if t == 12 else 1) + (1 if res['sync'] else 0)
            _cmp_step(ro, rh, t)
            if res['sync']:
                _same(eh, eo, f'state t={t}')
            prev = ro['done']


def test_class_loop_on_the_mirror():
    """SimpleGridworld(16 384 envs) through `env.step(a); env.reset(done)`: ONE launch per iteration on the mirror
    (mirror_state: lazy, current), equal outputs and state to the same loop with resident_mirror=False; a look at the state
    writes it out, a rollout and an eager reset go through the planes, an in-place edit that makes an env hand-made is served
    (the mirror refused) and healed again"""
    import torch
    from wurm_amd import _lib
    from wurm_amd.envs import SimpleGridworld
    N, S, T, start, seed = 16384, 9, 60, (4, 4), 33
    g = torch.Generator(device='cuda:0').manual_seed(4)
    acts = torch.randint(0, 4, (T, N), generator=g, device='cuda:0')
    count = _lib.lib().wurm_launch_count

    def loop(mirror):
        env = SimpleGridworld(N, S, start_location=start, observation_mode='default', device='cuda:0', seed=seed,
                              resident_mirror=mirror)
        outs, per_iter = [], []
        for t in range(T):
            n0 = count()
            obs, r, d, info = env.step(acts[t].clone())
            # (both forms of the reset between t = 22 and 38; switching between them costs an eager reset each time)
            back = env.reset(d) if (t % 3 or not 22 <= t < 38) else env.reset(d, return_observations=False)
            per_iter.append(count() - n0)
            outs.append([x.clone() for x in (obs, r, d, info['edge_collision'])] + ([back.clone()] if back is not None else []))
            if t == 20:
                outs.append([env.envs.clone()])                    # a look: written out, then eager while held? (dropped at once)
            if t == 30:
                ro = env.rollout(acts[:3].clone())
                outs.append([ro['observations'].clone(), ro['dones'].clone()])
            if t == 40:
                e = env.envs
                e[7, 0, 1, 1] = 1; e[7, 0, 1, 2] = 1             # env 7 gets extra foods: outside the domain
                del e
            if t == 45:
                e = env.envs
                e[7, 0] = 0; e[7, 0, 3, 3] = 1
                del e
        outs.append([env.envs.clone()])
        return outs, per_iter, env.mirror_state()

    (a, pa, ma), (b, pb, mb) = loop(None), loop(False)
    assert ma['state'] in ('lazy', 'eager') and ma['bytes'] == 16 + 4 * N and mb['state'] == 'off'
    assert pa[5:20] == [1] * 15 and pa[52:] == [1] * 8, pa     # one launch per iteration in the steady state
    assert min(pb[5:20]) == 2                                   # (without the mirror: lane kernel + the flag pass)
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        assert len(x) == len(y)
        for i, (u, v) in enumerate(zip(x, y)):
            assert torch.equal(u, v), f'record {k} output {i}'


def test_class_rollouts_and_steps_interleaved_on_the_mirror():
    """round 6: `env.rollout()` of a large SimpleGridworld batch reads and keeps the mirror of the per-call step
    (wurm_grid_rollout_resident): rollouts and `step; reset(done)` iterations interleaved stay ONE launch each, a look at the
    state in between writes the lazy mirror out, a hand-made env is served (the mirror refused until the next edit) — every
    output and the state equal to the same sequence with resident_mirror=False"""
    import torch
    from wurm_amd import _lib
    from wurm_amd.envs import SimpleGridworld
    N, S, start, seed = 16384, 9, (4, 4), 17
    g = torch.Generator(device='cuda:0').manual_seed(9)
    tape = torch.randint(0, 4, (12, 5, N), generator=g, device='cuda:0')
    acts = torch.randint(0, 4, (40, N), generator=g, device='cuda:0')
    count = _lib.lib().wurm_launch_count

    def run(mirror):
        env = SimpleGridworld(N, S, start_location=start, observation_mode='default', device='cuda:0', seed=seed,
                              resident_mirror=mirror)
        outs, launches = [], []
        k = 0
        for i in range(12):
            for j in range(2):   # two launches in a row: the second finds the mirror current, whatever came before the first
                n0 = count()
                ro = env.rollout(tape[i].clone(), return_observations=((i + j) % 2 == 0))
                if j == 1:
                    launches.append(count() - n0)
                outs.append([ro['rewards'].clone(), ro['dones'].clone()] +
                            ([ro['observations'].clone()] if ro['observations'] is not None else []))
            for _ in range(3):
                obs, r, d, info = env.step(acts[k].clone()); k += 1
                back = env.reset(d)
                outs.append([obs.clone(), r.clone(), d.clone(), back.clone()])
            if i == 4:
                outs.append([env.envs.clone()])
            if i == 7:
                e = env.envs
                e[11, 0, 1, 1] = 1; e[11, 0, 1, 2] = 1          # extra foods: outside the lane kernel's domain
                del e
            if i == 9:
                e = env.envs
                e[11, 0] = 0; e[11, 0, 3, 3] = 1
                del e
        outs.append([env.envs.clone()])
        return outs, launches, env.mirror_state()

    (a, la, ma), (b, lb, mb) = run(None), run(False)
    assert ma['state'] != 'off' and mb['state'] == 'off'
    # on the mirror: no flag pass behind the rollout kernel (launch 8 follows the hand-made env: refused, two launches)
    assert la[:8] == [1] * 8 and la[8] == 2 and la[10:] == [1, 1], la
    assert min(lb) == 2, lb
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        assert len(x) == len(y)
        for i, (u, v) in enumerate(zip(x, y)):
            assert torch.equal(u, v), f'record {k} output {i}'


@pytest.mark.parametrize('mode', ['default', 'raw', 'positions', 'none'])
@pytest.mark.parametrize('N,S,lazy', [(200, 9, True), (131, 9, False), (70, 5, True), (33, 12, True)])
def test_rollout_on_the_mirror_matches_the_oracle(N, S, lazy, mode):
    """wurm_grid_rollout_resident through the C ABI against the oracle: chained launches on the mirror (the first builds it),
    lazy (the planes written out by wurm_grid_resident_flush at the end) and eager, a hand-made env in between (refused, then
    rebuilt after the repair)"""
    import ctypes
    import torch
    from wurm_amd import _lib
    from wurm_amd._lib import knobs
    rng = np.random.RandomState(N + 3 * S)
    start = (S // 2, S // 2)
    o = OracleBackend(seed=21, env_offset=5)
    eo = _fresh(o, N, S, start)
    dev = torch.device('cuda:0')
    e_dev = torch.from_numpy(eo.copy()).to(dev)
    l = _lib.lib()
    m, n = _lib.parse_obs_mode(mode)
    res = torch.zeros(16 + 4 * N, dtype=torch.uint8, device=dev)
    valid = ctypes.c_int(0)
    call = 100
    stream = _lib.stream_ptr(0)
    with knobs(WURM_LANE_ROLLOUT_MIN_ENVS=0, WURM_LANE_STEP_MIN_ENVS=0):
        assert l.wurm_grid_resident_size(_lib.i64(N), S, m) == 16 + 4 * N
        for launch in range(7):
            T = int(rng.choice([1, 3, 8, 20]))
            a = rng.randint(-2, 7, size=(T, N)).astype(np.int64)
            o.call = call
            ro = o.grid_rollout(eo, a.copy(), start, mode)
            a_dev = torch.from_numpy(a.copy()).to(dev)
            shape = _o_shape(mode, N, S)
            obs = torch.empty((T,) + shape, dtype=torch.float32, device=dev) if shape else None
            reward = torch.empty((T, N), dtype=torch.float32, device=dev)
            done = torch.empty((T, N), dtype=torch.uint8, device=dev)
            edge = torch.empty((T, N), dtype=torch.uint8, device=dev)
            was = valid.value
            n0 = l.wurm_launch_count()
            rc = l.wurm_grid_rollout_resident(_lib.ptr(e_dev), _lib.ptr(a_dev), _lib.ACT_I64, _lib.ptr(reward), _lib.ptr(done),
                                              _lib.ptr(edge), _lib.ptr(obs), m, n, _lib.i64(N), S, _lib.i64(T), start[0], start[1],
                                              _lib.u64(21), _lib.u64(call), _lib.i64(5), _lib.ptr(res), ctypes.addressof(valid),
                                              int(lazy), stream)
            assert rc == 0 and _route() == 'gridworld_lane'
            assert l.wurm_launch_count() - n0 == (1 if was == 1 else 2)
            call += 2 * T
            _same(ro['reward'], reward.cpu().numpy(), f'reward launch {launch}')
            _same(ro['done'], done.cpu().numpy(), f'done launch {launch}')
            if obs is not None:
                _same(ro['obs'], obs.cpu().numpy(), f'obs launch {launch}')
            if launch == 2:          # a hand-made env: written into the planes the caller can see (flushed first), mirror cleared
                _flush(l, e_dev, res, valid, lazy, N, S, stream)
                _same(eo, e_dev.cpu().numpy(), 'state before the edit')
                eo[1 % N, 0, 1, 1] = 1; eo[1 % N, 0, 2, 2] = 1; eo[1 % N, 0, 3, 3] = 1
                e_dev.copy_(torch.from_numpy(eo))
                valid.value = 0
            elif launch == 3:
                assert valid.value == 2          # refused; the planes are the state
                _same(eo, e_dev.cpu().numpy(), 'state while refused')
                eo[1 % N, 0] = 0; eo[1 % N, 0, 2, 2] = 1
                e_dev.copy_(torch.from_numpy(eo))
                valid.value = 0
            elif launch > 3:
                assert valid.value == 1
        _flush(l, e_dev, res, valid, lazy, N, S, stream)
        _same(eo, e_dev.cpu().numpy(), 'final state')


def _o_shape(mode, N, S):
    from oracle import oracle as _orc
    return tuple(_orc.grid_obs_shape(mode, N, S)) if _orc.grid_obs_shape(mode, N, S) else None


def _flush(l, e_dev, res, valid, lazy, N, S, stream):
    import ctypes
    from wurm_amd import _lib
    c = _lib.SingleCall()
    c.envs, c.num_envs, c.size = _lib.ptr(e_dev), N, S
    c.resident, c.resident_valid, c.resident_lazy = _lib.ptr(res), valid.value, int(lazy)
    _lib.check(l.wurm_grid_resident_flush(ctypes.addressof(c), stream), 'flush')
