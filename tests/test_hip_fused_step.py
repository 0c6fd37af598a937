"""GPU parity of the one-launch loop iteration: wurm_single_step_reset / wurm_grid_step_reset in both
groupings — [step, observe, reset] and [postponed reset, step, observe] — against the per-call oracle functions with
the same counters (RNG mode), against the reference's recorded tape (injected outcomes), and the SingleSnake class
whose reset(done) is deferred into the next step's launch, including callers that look at or edit env.envs in
between.  Bit-exact everywhere."""
import numpy as np
import pytest

from tests.backends import OracleBackend
from tests import replay

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


def _same(a, b, what):
    if a is None and b is None:
        return
    assert a is not None and b is not None, what
    replay._eq(a, b, what, '-')


def _cmp(ro, rh, t):
    for k in ('obs', 'reward', 'done', 'self_collision', 'edge_collision', 'obs_after'):
        _same(rh[k], ro[k], f'{k} t={t}')


@pytest.mark.parametrize('N,S,T,mode,dtype', [
    (64, 9, 150, 'partial_2', np.int64),   # BASELINE cfg2 shape
    (33, 9, 60, 'default', np.int32),
    (17, 12, 80, 'one_channel', np.int64),
    (9, 16, 60, 'positions', np.int64),
    (6, 36, 50, 'default', np.int64),      # BASELINE cfg5 shape
    (2, 64, 20, 'partial_3', np.int64),
])
def test_step_then_reset_grouping(hip, N, S, T, mode, dtype):
    """post_reset: one launch == wurm_single_step(call) + wurm_single_reset(done, call + 1)"""
    rng = np.random.RandomState(S)
    o, h = OracleBackend(seed=31, env_offset=70), hip(seed=31, env_offset=70)
    eo = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(eo, np.ones(N, np.uint8), 'none')
    eh = eo.copy()
    for t in range(T):
        a = rng.randint(0, 4, size=N).astype(dtype)
        ao, ah = a.copy(), a.copy()
        want_after = t % 2 == 0
        ro = o.single_step_reset(eo, ao, mode, call=10 + 2 * t, post_reset=True, want_obs_after=want_after)
        rh = h.single_step_reset(eh, ah, mode, call=10 + 2 * t, post_reset=True, want_obs_after=want_after)
        _same(ah, ao, f'actions t={t}')
        _same(eh, eo, f'state t={t}')
        _cmp(ro, rh, t)


@pytest.mark.parametrize('N,S,T,mode', [(64, 9, 150, 'partial_2'), (20, 11, 80, 'partial_3'), (7, 25, 60, 'default'),
                                        (5, 36, 40, 'raw')])
def test_postponed_reset_grouping(hip, N, S, T, mode):
    """pre_done: one launch == wurm_single_reset(done of the previous step, pre_call) + wurm_single_step(call); the state
    between the two launches is the post-step, pre-reset state; obs_after is what that reset will return"""
    rng = np.random.RandomState(S + 1)
    o, h = OracleBackend(seed=8), hip(seed=8)
    eo = np.zeros((N, 3, S, S), np.float32)
    o.single_reset(eo, np.ones(N, np.uint8), 'none')
    eh = eo.copy()
    prev = None
    for t in range(T):
        a = rng.randint(0, 4, size=N).astype(np.int64)
        ao, ah = a.copy(), a.copy()
        kw = dict(call=1 + 2 * t, pre_done=prev, pre_call=2 * t, want_obs_after=(t % 3 != 1))
        ro = o.single_step_reset(eo, ao, mode, **kw)
        rh = h.single_step_reset(eh, ah, mode, **kw)
        _same(ah, ao, f'actions t={t}')
        _same(eh, eo, f'pre-reset state t={t}')
        _cmp(ro, rh, t)
        prev = ro['done']
        if t % 7 == 3:  # the caller edits the state between the calls (tests/test_single_snake_env.py:54 style)
            eo[0, 0] = 0
            eo[0, 0, 1, 1] = 1
            eh[...] = eo


def test_gridworld_both_groupings(hip):
    N, S, T, start = 24, 9, 80, (4, 4)
    rng = np.random.RandomState(3)
    o, h = OracleBackend(seed=2), hip(seed=2)
    eo = np.zeros((N, 2, S, S), np.float32)
    o.grid_reset(eo, np.ones(N, np.uint8), start, 'none')
    eh = eo.copy()
    prev = None
    for t in range(T):
        a = rng.randint(0, 4, size=N).astype(np.int64)
        ao, ah = a.copy(), a.copy()
        if t < T // 2:
            kw = dict(call=1 + 2 * t, post_reset=True, want_obs_after=True, grid=start)
        else:
            kw = dict(call=1 + 2 * t, pre_done=prev, pre_call=2 * t, grid=start)
        ro = o.single_step_reset(eo, ao, 'default', **kw)
        rh = h.single_step_reset(eh, ah, 'default', **kw)
        _same(eh, eo, f'state t={t}')
        _cmp(ro, rh, t)
        prev = ro['done'] if t >= T // 2 else None


@pytest.mark.parametrize('name', ['single_s9_partial2', 'single_s12_default', 'single_s36_default'])
def test_fused_replays_the_reference_tape(hip, name):
    """the reference's recorded outcomes injected into the one-launch iteration, both groupings"""
    fx = replay.load(name)
    if not fx['reset_called'].all():
        pytest.skip('tape with skipped resets')
    mode = str(fx['mode'])
    N, S, T = (int(v) for v in fx['meta'][:3])
    h = hip()
    # [step, observe, reset]
    envs = fx['state0'].astype(np.float32)
    for t in range(T):
        a = fx['actions_in'][t].copy()
        r = h.single_step_reset(envs, a, mode, call=2 * t, post_reset=True, want_obs_after=True,
                                inject_food=fx['inject_food'][t], inject_reset=fx['inject_reset'][t])
        replay._eq(a, fx['actions_out'][t], 'sanitised actions', t)
        replay._eq(r['obs'], fx['obs_step'][t], 'step observation', t)
        replay._eq(r['obs_after'], fx['obs_reset'][t], 'reset observation', t)
        replay._eq(r['reward'], fx['reward'][t], 'reward', t)
        replay._eq(r['done'], fx['done'][t], 'done', t)
        replay._eq(envs, fx['state_reset'][t].astype(np.float32), 'post-reset state', t)
    # [postponed reset, step, observe]
    envs = fx['state0'].astype(np.float32)
    for t in range(T):
        a = fx['actions_in'][t].copy()
        r = h.single_step_reset(envs, a, mode, call=2 * t, pre_done=fx['done'][t - 1] if t else None,
                                pre_call=2 * t - 1, inject_pre_reset=fx['inject_reset'][t - 1] if t else None,
                                want_obs_after=True, inject_food=fx['inject_food'][t], inject_reset=fx['inject_reset'][t])
        replay._eq(a, fx['actions_out'][t], 'sanitised actions', t)
        replay._eq(envs, fx['state_step'][t].astype(np.float32), 'post-step state', t)
        replay._eq(r['obs'], fx['obs_step'][t], 'step observation', t)
        replay._eq(r['obs_after'], fx['obs_reset'][t], 'reset observation', t)
        replay._eq(r['self_collision'], fx['self_collision'][t], 'self_collision', t)
        replay._eq(r['edge_collision'], fx['edge_collision'][t], 'edge_collision', t)


# ------------------------------------------------------------------------------------------- the class

def _oracle_loop_env(N, S, seed, offset=0):
    ref = np.zeros((N, 3, S, S), np.float32)
    from oracle import oracle
    oracle.single_reset(ref, np.ones(N, np.uint8), 'none', seed=seed, call=0, env_offset=offset)
    return ref


@pytest.mark.parametrize('lazy', [True, False])
@pytest.mark.parametrize('reset_obs', [True, False, 'mixed'])
def test_class_loop_equals_oracle_loop(lazy, reset_obs):
    """`obs, r, d, info = env.step(a); env.reset(d)` through the class == the oracle's step / reset pair, whether the
    reset is deferred into the next launch or not, whether its observation is asked for or not."""
    import torch
    from oracle import oracle
    from wurm_amd.envs import SingleSnake
    N, S, T, seed, mode = 96, 9, 150, 77, 'partial_2'
    env = SingleSnake(N, S, observation_mode=mode, device='cuda:0', seed=seed, env_offset=5, lazy_reset=lazy)
    ref = _oracle_loop_env(N, S, seed, 5)
    g = torch.Generator().manual_seed(1)
    call = 1
    kept = []
    for t in range(T):
        a = torch.randint(4, (N,), generator=g)
        a_dev, a_ref = a.cuda(), a.numpy().copy()
        obs, r, d, info = env.step(a_dev)
        o_ref, r_ref, d_ref, sc_ref, ec_ref = oracle.single_step(ref, a_ref, mode, seed=seed, call=call, env_offset=5)
        want = reset_obs if reset_obs != 'mixed' else (t // 5) % 2 == 0
        back = env.reset(d) if want else env.reset(d, return_observations=False)
        b_ref = oracle.single_reset(ref, d_ref, mode, seed=seed, call=call + 1, env_offset=5)
        call += 2
        assert obs.shape == (N, 75) and r.shape == (N, 1) and d.shape == (N, 1) and d.dtype == torch.bool
        replay._eq(a_dev.cpu().numpy(), a_ref, 'sanitised actions', t)
        replay._eq(obs.cpu().numpy(), o_ref, 'obs', t)
        replay._eq(r.cpu().numpy()[:, 0], r_ref, 'reward', t)
        replay._eq(d.cpu().numpy()[:, 0], d_ref, 'done', t)
        replay._eq(info['self_collision'].cpu().numpy(), sc_ref, 'self_collision', t)
        replay._eq(info['edge_collision'].cpu().numpy(), ec_ref, 'edge_collision', t)
        if want:
            replay._eq(back.cpu().numpy(), b_ref, 'reset observation', t)
        else:
            assert back is None
        if t % 11 == 0:  # looking at the state applies the postponed reset
            replay._eq(env.envs.cpu().numpy(), ref, 'state', t)
        if t % 20 == 0:
            kept.append((obs, o_ref.copy(), d, d_ref.copy()))
    replay._eq(env.envs.cpu().numpy(), ref, 'final state', T)
    for obs, o_ref, d, d_ref in kept:  # outputs are fresh tensors: later steps must not have overwritten them
        replay._eq(obs.cpu().numpy(), o_ref, 'kept obs', '-')
        replay._eq(d.cpu().numpy()[:, 0], d_ref, 'kept done', '-')


def test_class_state_edits_and_odd_call_patterns():
    """between step and reset the caller reads env.envs (pre-reset state), edits it after the reset, passes other masks,
    skips resets, resets twice, changes the observation mode: always the eager pair's results"""
    import torch
    from oracle import oracle
    from wurm_amd.envs import SingleSnake
    N, S, seed = 40, 10, 5
    env = SingleSnake(N, S, observation_mode='partial_1', device='cuda:0', seed=seed)
    ref = _oracle_loop_env(N, S, seed)
    g = torch.Generator().manual_seed(2)
    call, mode = 1, 'partial_1'
    for t in range(120):
        a = torch.randint(4, (N,), generator=g)
        a_dev, a_ref = a.cuda(), a.numpy().copy()
        obs, r, d, info = env.step(a_dev)
        o_ref, r_ref, d_ref, _, _ = oracle.single_step(ref, a_ref, mode, seed=seed, call=call)
        call += 1
        replay._eq(obs.cpu().numpy(), o_ref, 'obs', t)
        k = t % 8
        if k == 0:      # read the pre-reset state, then the deferred reset, then read again
            replay._eq(env.envs.cpu().numpy(), ref, 'pre-reset state', t)
            env.reset(d, return_observations=False)
            oracle.single_reset(ref, d_ref, 'none', seed=seed, call=call); call += 1
            replay._eq(env.envs.cpu().numpy(), ref, 'post-reset state', t)
        elif k == 1:    # a different mask object with the same content
            env.reset(d.clone())
            oracle.single_reset(ref, d_ref, 'none', seed=seed, call=call); call += 1
        elif k == 2:    # no reset at all: done envs are stepped again (irregular states)
            pass
        elif k == 3:    # deferred reset, then the caller edits the state in place (sees the reset state)
            env.reset(d, return_observations=False)
            oracle.single_reset(ref, d_ref, 'none', seed=seed, call=call); call += 1
            e = env.envs
            e[1, 0] = 0
            e[1, 0, 2, 2] = 1
            ref[1, 0] = 0
            ref[1, 0, 2, 2] = 1
        elif k == 4:    # the mask is modified in place before it is handed back
            d[:] = False
            env.reset(d)
            oracle.single_reset(ref, np.zeros(N, np.uint8), 'none', seed=seed, call=call); call += 1
        elif k == 5:    # reset with no argument (= self.done), twice
            env.reset()
            oracle.single_reset(ref, d_ref, 'none', seed=seed, call=call); call += 1
            env.reset()
            oracle.single_reset(ref, d_ref, 'none', seed=seed, call=call); call += 1
        elif k == 6:    # deferred, then the state tensor is replaced wholesale: the postponed reset must not touch it
            env.reset(d, return_observations=False)
            oracle.single_reset(ref, d_ref, 'none', seed=seed, call=call); call += 1
            env.envs = torch.from_numpy(ref).cuda()
        else:           # deferred, then an observation in another mode
            env.reset(d, return_observations=False)
            oracle.single_reset(ref, d_ref, 'none', seed=seed, call=call); call += 1
            replay._eq(env._observe('default').cpu().numpy(), oracle.single_observe(ref, 'default'), 'observe', t)
            mode = 'partial_2' if mode == 'partial_1' else 'partial_1'
            env.observation_mode = mode
    replay._eq(env.envs.cpu().numpy(), ref, 'final state', '-')
    env.check_consistency() if (oracle.single_check(ref) == 0).all() else None


@pytest.mark.parametrize('lazy', [True, False])
def test_gridworld_class_loop_equals_oracle_loop(lazy):
    """SimpleGridworld through the same one-launch machinery (wurm_grid_step_slot): step / reset pairs with deferred and
    eager resets, with and without the reset observation, reading env.envs in between."""
    import torch
    from oracle import oracle
    from wurm_amd.envs import SimpleGridworld
    N, S, T, seed, mode, start = 64, 9, 120, 21, 'default', (4, 4)
    env = SimpleGridworld(N, S, observation_mode=mode, device='cuda:0', start_location=start, seed=seed, lazy_reset=lazy)
    ref = np.zeros((N, 2, S, S), np.float32)
    oracle.grid_reset(ref, np.ones(N, np.uint8), start, 'none', seed=seed, call=0)
    replay._eq(env.envs.cpu().numpy(), ref, 'fresh envs', 0)
    g = torch.Generator().manual_seed(4)
    call = 1
    for t in range(T):
        a = torch.randint(4, (N,), generator=g)
        a_dev, a_ref = a.cuda(), a.numpy().copy()
        obs, r, d, info = env.step(a_dev)
        o_ref, r_ref, d_ref, ec_ref = oracle.grid_step(ref, a_ref, mode, seed=seed, call=call)
        want = (t // 7) % 2 == 0
        back = env.reset(d) if want else env.reset(d, return_observations=False)
        b_ref = oracle.grid_reset(ref, d_ref, start, mode, seed=seed, call=call + 1)
        call += 2
        replay._eq(a_dev.cpu().numpy(), a.numpy(), 'actions untouched', t)
        replay._eq(obs.cpu().numpy(), o_ref, 'obs', t)
        replay._eq(r.cpu().numpy()[:, 0], r_ref, 'reward', t)
        replay._eq(d.cpu().numpy()[:, 0], d_ref, 'done', t)
        replay._eq(info['edge_collision'].cpu().numpy(), ec_ref, 'edge_collision', t)
        if want:
            replay._eq(back.cpu().numpy(), b_ref, 'reset observation', t)
        if t % 9 == 0:
            replay._eq(env.envs.cpu().numpy(), ref, 'state', t)
    replay._eq(env.envs.cpu().numpy(), ref, 'final state', T)


def test_alias_taken_before_the_reset_shows_the_reset_state():
    """Round 5 (was DESIGN.md §5 deviation 9): while the caller holds a tensor alias of `env.envs` the reset is not
    postponed (`_alias_free`: the storage's use count), so the alias shows the reset state at once, with either setting of
    lazy_reset, as the reference's would."""
    import torch
    from oracle import oracle
    from wurm_amd.envs import SingleSnake
    N, S, seed = 64, 9, 13
    g = torch.Generator().manual_seed(3)
    for lazy in (True, False):
        env = SingleSnake(N, S, observation_mode='partial_2', device='cuda:0', seed=seed, lazy_reset=lazy)
        ref = _oracle_loop_env(N, S, seed)
        call = 1
        for t in range(40):   # until some env has finished
            a = torch.randint(4, (N,), generator=g)
            alias = env.envs                                  # taken before the iteration
            obs, r, d, info = env.step(a.cuda())
            _, _, d_ref, _, _ = oracle.single_step(ref, a.numpy().copy(), 'partial_2', seed=seed, call=call)
            pre_reset = ref.copy()
            env.reset(d, return_observations=False)
            oracle.single_reset(ref, d_ref, 'none', seed=seed, call=call + 1)
            call += 2
            if d_ref.any():
                # with an alias of the state alive the reset is NOT postponed (round 5: _alias_free; the storage's use count
                # tells): the alias shows what the reference's would, the reset state, and so does the attribute
                replay._eq(alias.cpu().numpy(), ref, 'alias right after reset(done)', t)
                replay._eq(env.envs.cpu().numpy(), ref, 'the attribute', t)
                assert alias.data_ptr() == env.envs.data_ptr()
                break
        else:
            raise AssertionError('no env finished in 40 steps')


def test_state_edited_between_step_and_reset_is_observed_by_the_reset():
    """`step(a); env.envs[i] = ...; obs = env.reset(done)`: the observation the step launch pre-computed for the reset is
    stale once the caller has been handed the state; the reset must observe the edited state, as the reference does."""
    import torch
    from oracle import oracle
    from wurm_amd.envs import SingleSnake
    N, S, seed, mode = 32, 9, 9, 'partial_2'
    env = SingleSnake(N, S, observation_mode=mode, device='cuda:0', seed=seed)
    ref = _oracle_loop_env(N, S, seed)
    g = torch.Generator().manual_seed(8)
    call = 1
    for t in range(30):
        a = torch.randint(4, (N,), generator=g)
        obs, r, d, info = env.step(a.cuda())
        _, _, d_ref, _, _ = oracle.single_step(ref, a.numpy().copy(), mode, seed=seed, call=call)
        if t % 3 == 2:        # move env 0's food next to nothing in particular: any visible edit will do
            e = env.envs
            e[:, 0] = 0
            e[:, 0, 1, 1] = 1
            ref[:, 0] = 0
            ref[:, 0, 1, 1] = 1
        back = env.reset(d)
        b_ref = oracle.single_reset(ref, d_ref, mode, seed=seed, call=call + 1)
        call += 2
        replay._eq(back.cpu().numpy(), b_ref, 'reset observation', t)
    replay._eq(env.envs.cpu().numpy(), ref, 'final state', '-')


def test_class_loop_under_inference_mode_and_with_awkward_action_tensors():
    """torch.inference_mode() (no version counters: resets run eagerly), actions as a strided view / (N,1) / on the CPU:
    the loop runs and equals the oracle's"""
    import torch
    from oracle import oracle
    from wurm_amd.envs import SingleSnake
    N, S, seed, mode = 48, 9, 4, 'partial_2'
    env = SingleSnake(N, S, observation_mode=mode, device='cuda:0', seed=seed)
    ref = _oracle_loop_env(N, S, seed)
    g = torch.Generator().manual_seed(5)
    call = 1
    for t in range(60):
        a = torch.randint(4, (N,), generator=g)
        a_ref = a.numpy().copy()
        kind = t % 4
        ctx = torch.inference_mode() if t >= 30 else torch.no_grad()
        with ctx:
            if kind == 0:
                a_in = a.cuda()
            elif kind == 1:
                a_in = torch.stack([a, a], dim=1).cuda()[:, 0]      # strided view
            elif kind == 2:
                a_in = a.cuda().view(N, 1)
            else:
                a_in = a.clone()                                      # CPU tensor
            obs, r, d, info = env.step(a_in)
            o_ref, r_ref, d_ref, _, _ = oracle.single_step(ref, a_ref, mode, seed=seed, call=call)
            back = env.reset(d)
            b_ref = oracle.single_reset(ref, d_ref, mode, seed=seed, call=call + 1)
        call += 2
        replay._eq(obs.cpu().numpy(), o_ref, 'obs', t)
        replay._eq(a_in.cpu().numpy().reshape(N), a_ref, 'sanitised actions', t)
        replay._eq(d.cpu().numpy()[:, 0], d_ref, 'done', t)
        replay._eq(back.cpu().numpy(), b_ref, 'reset observation', t)
    replay._eq(env.envs.cpu().numpy(), ref, 'final state', '-')
