"""GPU parity of MultiSnake's one-launch iteration: wurm_multi_step_reset ([postponed reset,] step, observe) against the
oracle's reset / step pair with the same counters, and the MultiSnake class whose reset(dones['__all__'],
return_observations=False) is deferred into the next step's launch — including callers that look at or edit the state
attributes in between, other masks, respawn_mode='any' and re-rolled colours.  Bit-exact everywhere."""
import numpy as np
import pytest

from oracle import oracle as _o
from tests.backends import OracleBackend
from tests.test_hip_multi_vs_oracle import CFGS, _same, _same_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


@pytest.mark.parametrize('N,K,S,T,mode,cfg', [
    (16, 4, 25, 60, 'full', 'default'),         # BASELINE cfg4 shape
    (12, 4, 25, 80, 'partial_5', 'train'),      # respawn 'any', random_rate food
    (10, 3, 10, 90, 'full', 'dense'),           # fixed colours, crowded
    (9, 2, 12, 90, 'full', 'noboost'),
    (7, 1, 5, 60, 'full', 'default'),
    (6, 10, 36, 40, 'full', 'train'),           # experiments/speeds.py shape: one env per workgroup
    (5, 4, 48, 30, 'full', 'default'),
    (4, 5, 40, 30, 'none', 'dense'),
])
def test_postponed_reset_in_front_of_the_step(hip, N, K, S, T, mode, cfg):
    cfg = CFGS[cfg]
    rng = np.random.RandomState(K * S)
    o, h = OracleBackend(seed=3, env_offset=11), hip(seed=3, env_offset=11)
    so = _o.multi_empty_state(N, K, S)
    so['colours'][...] = o.multi_colours(N, K, cfg['colour_mode'] == 'fixed', call=0)
    o.call = 1
    assert o.multi_reset(so, np.ones(N), cfg) == 0
    sh = {k: v.copy() for k, v in so.items()}
    call, prev, prev_call = 2, None, 0
    deaths = 0
    for t in range(T):
        a = rng.randint(0, 8, size=(K, N)).astype(np.int64)
        # oracle: the postponed reset (if any) with its counter, then the step with its own
        if prev is not None:
            o.call = prev_call
            o.multi_reset(so, prev, cfg)
        o.call = call
        ro = o.multi_step(so, a, cfg, mode)
        rh = h.multi_step_reset(sh, a, cfg, mode, call=call, pre_done=prev, pre_call=prev_call, want_obs_after=(t % 3 != 2))
        _same_state(so, sh, f'state t={t}')
        for k in ro:
            _same(ro[k], rh[k], f'{k} t={t}')
        if 'obs_after' in rh:  # what reset(all_done) with the next counter returns — computed, not applied
            tmp = {k: v.copy() for k, v in so.items()}
            o.call = call + 1
            o.multi_reset(tmp, ro['all_done'], cfg, mode=mode)
            _same(o.last_reset_obs, rh['obs_after'], f'obs_after t={t}')
        deaths += int(so['dones'].sum())
        if t % 5 == 4:      # no reset this time: dead snakes are stepped again
            prev = None
        elif t % 7 == 3:    # an arbitrary mask
            prev, prev_call = (rng.rand(N) < 0.3).astype(np.uint8), call + 1
        else:
            prev, prev_call = ro['all_done'], call + 1
        call += 2
    assert deaths > 0


@pytest.mark.parametrize('lazy', [True, False])
@pytest.mark.parametrize('cfg_name,mode', [('default', 'full'), ('train', 'partial_3')])
def test_class_loop_equals_oracle_loop(lazy, cfg_name, mode):
    import torch
    from wurm_amd.envs import MultiSnake
    cfg = CFGS[cfg_name]
    N, K, S, T, seed = 24, 3, 14, 120, 99
    env = MultiSnake(N, K, S, device='cuda:0', seed=seed, env_offset=7, observation_mode=mode, lazy_reset=lazy,
                     boost=cfg['boost'], food_on_death_prob=cfg['food_on_death_prob'],
                     boost_cost_prob=cfg['boost_cost_prob'], food_mode=cfg['food_mode'], food_rate=cfg['food_rate'],
                     respawn_mode=cfg['respawn_mode'], reward_on_death=cfg['reward_on_death'],
                     agent_colours=cfg['colour_mode'])
    o = OracleBackend(seed=seed, env_offset=7)
    st = _o.multi_empty_state(N, K, S)
    st['colours'][...] = o.multi_colours(N, K, cfg['colour_mode'] == 'fixed', call=0)
    o.call = 1
    assert o.multi_reset(st, np.ones(N), cfg) == 0
    g = torch.Generator().manual_seed(5)

    def check_state(what):
        _same(env.foods.cpu().numpy(), st['foods'], what + ' foods')
        _same(env.heads.cpu().numpy(), st['heads'], what + ' heads')
        _same(env.bodies.cpu().numpy(), st['bodies'], what + ' bodies')
        _same(env.dones.cpu().numpy().astype(np.uint8), st['dones'], what + ' dones')
        _same(env.orientations.cpu().numpy(), st['orientations'], what + ' orientations')
        _same(env.agent_colours.cpu().numpy(), st['colours'], what + ' colours')

    check_state('fresh')
    for t in range(T):
        a = torch.randint(8, (K, N), generator=g)
        if t % 2:           # separate tensors (stacked by the class) / rows of one tensor (used where they lie)
            obs, rew, dones, info = env.step({f'agent_{i}': a[i].cuda() for i in range(K)})
        else:
            ac = a.cuda()
            obs, rew, dones, info = env.step({f'agent_{i}': ac[i] for i in range(K)})
        r = o.multi_step(st, a.numpy(), cfg, mode)
        for i in range(K):
            _same(obs[f'agent_{i}'].cpu().numpy(), r['obs'][i], f'obs {i} t={t}')
            _same(rew[f'agent_{i}'].cpu().numpy(), r['rewards'].reshape(N, K)[:, i], f'reward {i} t={t}')
            _same(dones[f'agent_{i}'].cpu().numpy().astype(np.uint8), st['dones'].reshape(N, K)[:, i], f'done {i} t={t}')
            _same(info[f'size_{i}'].cpu().numpy(), r['size'].reshape(N, K)[:, i], f'size {i} t={t}')
            _same(info[f'food_{i}'].cpu().numpy(), r['food'].reshape(N, K)[:, i], f'food {i} t={t}')
            _same(info[f'snake_collision_{i}'].cpu().numpy().astype(np.uint8), r['snake_collision'].reshape(N, K)[:, i],
                  f'snake collision {i} t={t}')
            _same(info[f'edge_collision_{i}'].cpu().numpy().astype(np.uint8), r['edge_collision'].reshape(N, K)[:, i],
                  f'edge collision {i} t={t}')
            _same(info[f'boost_{i}'].cpu().numpy().astype(np.uint8), st['boost_this_step'].reshape(N, K)[:, i],
                  f'boost {i} t={t}')
        assert set(info) == {f'{k}_{i}' for k in ('snake_collision', 'edge_collision', 'food', 'boost', 'size')
                             for i in range(K)}                                # the reference's keys (:485-487, :661-729)
        _same(env.rewards.cpu().numpy(), r['rewards'], f'env.rewards t={t}')     # env-major attributes (:104-105)
        _same(env.boost_this_step.cpu().numpy().astype(np.uint8), st['boost_this_step'], f'env.boost_this_step t={t}')
        _same(dones['__all__'].cpu().numpy().astype(np.uint8), r['all_done'], f'all_done t={t}')
        k = t % 6
        if k == 4:          # the reference's default: the reset returns observations (eager)
            back = env.reset(dones['__all__'])
            o.multi_reset(st, r['all_done'], cfg, mode=mode)
            for i in range(K):
                _same(back[f'agent_{i}'].cpu().numpy(), o.last_reset_obs[i], f'reset obs {i} t={t}')
        elif k == 5:        # another mask object
            env.reset(dones['__all__'].clone(), return_observations=False)
            o.multi_reset(st, r['all_done'], cfg)
        else:               # deferred when lazy
            assert env.reset(dones['__all__'], return_observations=False) is None
            o.multi_reset(st, r['all_done'], cfg)
        if k == 1:          # looking at the state applies the postponed reset
            check_state(f't={t}')
        if k == 2:          # ... and so does editing it
            f = env.foods
            f[0, 0, 1, 1] = 1.0
            st['foods'][0, 0, 1, 1] = 1.0
    check_state('final')
    env.check_consistency() if (o.multi_check(st) == 0).all() else None


@pytest.mark.parametrize('cfg_name,mode,shape', [('default', 'full', (20, 3, 12, 100)), ('train', 'partial_3', (20, 3, 12, 100)),
                                                 ('dense', 'full', (20, 3, 12, 100)), ('train', 'full', (6, 10, 36, 40))])
def test_class_loop_with_reset_observations(cfg_name, mode, shape):
    """The reference's own call pattern, `obs, r, d, info = env.step(a); obs = env.reset(d['__all__'])` every iteration: the
    first reset runs at once, later ones are served by the step launch (obs_after) and deferred."""
    import torch
    from wurm_amd.envs import MultiSnake
    cfg = CFGS[cfg_name]
    (N, K, S, T), seed = shape, 41
    env = MultiSnake(N, K, S, device='cuda:0', seed=seed, env_offset=3, observation_mode=mode,
                     boost=cfg['boost'], food_on_death_prob=cfg['food_on_death_prob'],
                     boost_cost_prob=cfg['boost_cost_prob'], food_mode=cfg['food_mode'], food_rate=cfg['food_rate'],
                     respawn_mode=cfg['respawn_mode'], reward_on_death=cfg['reward_on_death'],
                     agent_colours=cfg['colour_mode'])
    o = OracleBackend(seed=seed, env_offset=3)
    st = _o.multi_empty_state(N, K, S)
    st['colours'][...] = o.multi_colours(N, K, cfg['colour_mode'] == 'fixed', call=0)
    o.call = 1
    assert o.multi_reset(st, np.ones(N), cfg) == 0
    g = torch.Generator().manual_seed(6)
    served = 0
    for t in range(T):
        a = torch.randint(8, (K, N), generator=g)
        ac = a.cuda()
        obs, rew, dones, info = env.step({f'agent_{i}': ac[i] for i in range(K)})
        r = o.multi_step(st, a.numpy(), cfg, mode)
        for i in range(K):
            _same(obs[f'agent_{i}'].cpu().numpy(), r['obs'][i], f'obs {i} t={t}')
            _same(rew[f'agent_{i}'].cpu().numpy(), r['rewards'].reshape(N, K)[:, i], f'reward {i} t={t}')
        _same(dones['__all__'].cpu().numpy().astype(np.uint8), r['all_done'], f'all_done t={t}')
        served += env._obs_after is not None
        back = env.reset(dones['__all__'])
        o.multi_reset(st, r['all_done'], cfg, mode=mode)
        for i in range(K):
            _same(back[f'agent_{i}'].cpu().numpy(), o.last_reset_obs[i], f'reset obs {i} t={t}')
        if t % 10 == 7:  # looking at the state applies the postponed reset
            _same(env.bodies.cpu().numpy(), st['bodies'], f'bodies t={t}')
            _same(env.agent_colours.cpu().numpy(), st['colours'], f'colours t={t}')
            _same(env.dones.cpu().numpy().astype(np.uint8), st['dones'], f'dones t={t}')
    assert served >= T - 2
    _same(env.foods.cpu().numpy(), st['foods'], 'final foods')
    _same(env.heads.cpu().numpy(), st['heads'], 'final heads')
    _same(env.orientations.cpu().numpy(), st['orientations'], 'final orientations')


def test_alias_taken_before_the_reset_shows_the_reset_state():
    """Round 5 (was DESIGN.md §5 deviation 9): while the caller holds aliases of `foods / heads / bodies` the reset is not
    postponed (`_alias_free`: the storages' use counts), so the aliases show the reset state at once, as the reference's."""
    import torch
    from wurm_amd.envs import MultiSnake
    cfg = CFGS['default']
    N, K, S, seed = 32, 2, 10, 17
    env = MultiSnake(N, K, S, device='cuda:0', seed=seed, observation_mode='full')
    o = OracleBackend(seed=seed)
    st = _o.multi_empty_state(N, K, S)
    st['colours'][...] = o.multi_colours(N, K, cfg['colour_mode'] == 'fixed', call=0)
    o.call = 1
    assert o.multi_reset(st, np.ones(N), cfg) == 0
    g = torch.Generator().manual_seed(2)
    for t in range(200):
        a = torch.randint(8, (K, N), generator=g)
        bodies_alias, foods_alias = env.bodies, env.foods           # taken before the iteration
        ac = a.cuda()
        _, _, dones, _ = env.step({f'agent_{i}': ac[i] for i in range(K)})
        r = o.multi_step(st, a.numpy(), cfg, 'full')
        pre = {k: st[k].copy() for k in ('bodies', 'foods')}
        assert env.reset(dones['__all__'], return_observations=False) is None
        o.multi_reset(st, r['all_done'], cfg)
        if r['all_done'].any():
            assert pre is not None
            _same(bodies_alias.cpu().numpy(), st['bodies'], 'alias right after reset(all_done)')
            _same(foods_alias.cpu().numpy(), st['foods'], 'alias right after reset(all_done)')
            _same(env.bodies.cpu().numpy(), st['bodies'], 'the attribute')
            return
    raise AssertionError('no env finished in 200 steps')


def test_class_loop_under_inference_mode():
    import torch
    from wurm_amd.envs import MultiSnake
    cfg = CFGS['default']
    N, K, S, seed = 16, 2, 10, 23
    env = MultiSnake(N, K, S, device='cuda:0', seed=seed, observation_mode='full')
    o = OracleBackend(seed=seed)
    st = _o.multi_empty_state(N, K, S)
    st['colours'][...] = o.multi_colours(N, K, cfg['colour_mode'] == 'fixed', call=0)
    o.call = 1
    assert o.multi_reset(st, np.ones(N), cfg) == 0
    g = torch.Generator().manual_seed(3)
    with torch.inference_mode():
        for t in range(40):
            a = torch.randint(8, (K, N), generator=g)
            ac = a.cuda()
            obs, _, dones, _ = env.step({f'agent_{i}': ac[i] for i in range(K)})
            r = o.multi_step(st, a.numpy(), cfg, 'full')
            _same(obs['agent_0'].cpu().numpy(), r['obs'][0], f'obs t={t}')
            env.reset(dones['__all__'], return_observations=bool(t % 2))
            o.multi_reset(st, r['all_done'], cfg)
    _same(env.bodies.cpu().numpy(), st['bodies'], 'final bodies')


@pytest.mark.gpu
def test_reading_a_state_attribute_does_not_cost_the_one_call_step_for_ever(monkeypatch):
    """`env.heads` read once (experiments/multiagent.py:531 does, for its heat maps) left the tensor on the watch
    list and every later step went the long way round (`_slow_step`: the version checks, the stacking test ...).  An alias
    the caller has dropped is pruned by the next step; one it keeps is watched as before — and the trajectories are those
    of an env nobody looked at."""
    import torch
    from wurm_amd.envs import MultiSnake
    N, K, S = 64, 2, 12
    g = torch.Generator().manual_seed(3)
    tape = torch.randint(8, (40, K, N), generator=g).cuda()
    keys = ['agent_%d' % i for i in range(K)]
    slow = []
    real = MultiSnake._slow_step
    monkeypatch.setattr(MultiSnake, '_slow_step', lambda self, a: (slow.append(1), real(self, a))[1])
    env, ref = MultiSnake(N, K, S, device='cuda:0', seed=9), MultiSnake(N, K, S, device='cuda:0', seed=9)

    def it(e, t):
        o = e.step(dict(zip(keys, tape[t].unbind(0))))
        e.reset(o[2]['__all__'], return_observations=False)
        return o
    for t in range(4):
        it(env, t), it(ref, t)
    n0 = len(slow)
    it(env, 4), it(ref, 4)
    assert len(slow) == n0                                  # two fast steps
    assert float(env.heads.sum()) == float(ref.heads.sum())  # both looked at, both dropped
    n1 = len(slow)
    for t in range(5, 15):
        a, b = it(env, t), it(ref, t)
    assert len(slow) - n1 <= 2                              # (at most the one step per env that finds the alias gone)
    h = env.heads                                           # kept: watched, slow, correct
    n2 = len(slow)
    for t in range(15, 20):
        a, b = it(env, t), it(ref, t)
        assert all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and all(torch.equal(a[1][k], b[1][k]) for k in a[1])
    assert len(slow) - n2 == 5
    del h
    for t in range(20, 30):
        a, b = it(env, t), it(ref, t)
        assert all(torch.equal(a[0][k], b[0][k]) for k in a[0])
    assert len(slow) - n2 <= 6
    for name in ('foods', 'heads', 'bodies', 'dones', 'orientations'):
        assert torch.equal(getattr(env, name), getattr(ref, name)), name


@pytest.mark.gpu
def test_a_caller_that_discards_what_reset_returns_stops_paying_for_it():
    """experiments/speeds.py:30-38: `env.step(a); env.reset(done['__all__'])` — the K observations reset returns (:834-836)
    are dropped.  After three such iterations the steps stop writing them (`want_obs_after` off, reset hands out a
    _LazyResetObs); a caller that then keeps or reads one gets exactly what an eager reset returns, and the env goes back to
    precomputing.  One launch per iteration throughout; trajectories equal to an env with `lazy_reset=False`."""
    import torch
    from wurm_amd import _lib
    from wurm_amd.envs import MultiSnake
    from wurm_amd.envs.multi_snake import _LazyResetObs
    N, K, S = 96, 3, 12
    g = torch.Generator().manual_seed(5)
    tape = torch.randint(8, (40, K, N), generator=g).cuda()
    keys = ['agent_%d' % i for i in range(K)]
    kw = dict(device='cuda:0', seed=21, respawn_mode='any', food_mode='random_rate', food_rate=2.5e-3)
    env, ref = MultiSnake(N, K, S, **kw), MultiSnake(N, K, S, lazy_reset=False, resident_mirror=False, **kw)
    count = _lib.lib().wurm_launch_count
    acts = lambda t: dict(zip(keys, tape[t].unbind(0)))   # noqa: E731
    for t in range(8):                                    # dropped: the lazy form sets in
        a = env.step(acts(t)); env.reset(a[2]['__all__'])
        b = ref.step(acts(t)); ref.reset(b[2]['__all__'])
        assert all(torch.equal(a[0][k], b[0][k]) for k in a[0])
    assert env._lazy_obs_mode and not env._fs.want_obs_after
    n0 = count()
    for t in range(8, 12):
        a = env.step(acts(t)); env.reset(a[2]['__all__'])
    assert count() - n0 == 4                              # one launch per iteration, no second observation stream
    for t in range(8, 12):
        b = ref.step(acts(t)); ref.reset(b[2]['__all__'])
    a = env.step(acts(12)); kept = env.reset(a[2]['__all__'])            # kept, not looked at ...
    b = ref.step(acts(12)); want = ref.reset(b[2]['__all__'])
    assert type(kept) is _LazyResetObs and kept._env is not None
    a = env.step(acts(13)); b = ref.step(acts(13))                       # ... across the next step: filled in front of it
    assert kept._env is None and list(kept) == list(want) and all(torch.equal(kept[k], want[k]) for k in want)
    assert all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and not env._lazy_obs_mode
    r1, r2 = env.reset(a[2]['__all__']), ref.reset(b[2]['__all__'])     # read at once: an ordinary dict again soon
    assert all(torch.equal(r1[k], r2[k]) for k in r2)
    for t in range(14, 24):
        a = env.step(acts(t)); r1 = env.reset(a[2]['__all__'])
        b = ref.step(acts(t)); r2 = ref.reset(b[2]['__all__'])
        assert all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and all(torch.equal(r1[k], r2[k]) for k in r2)
    assert not env._lazy_obs_mode                                        # (a caller that reads it is a caller that reads it)
    for name in ('foods', 'heads', 'bodies', 'dones', 'orientations', 'agent_colours'):
        assert torch.equal(getattr(env, name), getattr(ref, name)), name
