"""MultiSnake's per-call step on a resident mirror of foods / heads / bodies (wurm_multi_call.resident; the LDS image of
the env's grids kept between calls, wurm_amd/csrc/multi_snake.hip): eager and lazy, one env per wave and one env per
workgroup, every dynamics configuration of the parity suite.

(a) through the C ABI: the oracle follows [postponed reset,] step over many iterations while foods / heads / bodies are
    written out only now and then (lazy), with iterations without any reset, arbitrary reset masks, hand-edited states
    (the caller clears resident_valid) and other entry points in between;
(b) through the host class: the reference's loops with the mirror forced on against the oracle, state attributes read,
    edited in place through an alias, and replaced in between; long-lived snakes crossing the clock re-base;
(c) in a child process with WURM_RESIDENT_MIN_ENVS=0 the MultiSnake parity suites as a whole."""
import contextlib
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as _o
from tests.backends import OracleBackend
from tests.test_hip_multi_vs_oracle import CFGS, _same, _same_state

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def hip():
    from tests.hip_backend import HipBackend
    return HipBackend


from wurm_amd._lib import knobs  # noqa: E402  (library options + environment for child processes)


@pytest.mark.parametrize('lazy', [False, True])
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
def test_abi_postponed_reset_and_step_on_the_mirror(hip, N, K, S, T, mode, cfg, lazy):
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
    mirror = {'valid': 0, 'lazy': lazy}
    with knobs(WURM_RESIDENT_MIN_ENVS=0):
        for t in range(T):
            a = rng.randint(0, 8, size=(K, N)).astype(np.int64)
            if prev is not None:
                o.call = prev_call
                o.multi_reset(so, prev, cfg)
            o.call = call
            ro = o.multi_step(so, a, cfg, mode)
            edit = t % 13 == 6
            other = t % 17 == 9
            mirror['sync'] = not lazy or edit or other or t % 4 == 3 or t == T - 1
            rh = h.multi_step_reset(sh, a, cfg, mode, call=call, pre_done=prev, pre_call=prev_call,
                                    want_obs_after=(t % 3 != 2), resident=mirror)
            if mirror['sync']:
                _same_state(so, sh, f'state t={t}')
            else:
                for k in ('dones', 'orientations', 'colours', 'boost_this_step'):
                    _same(so[k], sh[k], f'{k} t={t}')
            for k in ro:
                _same(ro[k], rh[k], f'{k} t={t}')
            if 'obs_after' in rh:
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
            if edit:            # the caller edits the state (and says so): a food appears, one disappears
                so['foods'][0, 0, 2, 2] = 1
                so['foods'][(t // 13) % N, 0] *= 0
                for k in ('foods',):
                    sh[k][...] = so[k]
                mirror['valid'] = 0
            if other:           # another entry point writes the state in between: an eager reset of a few envs
                some = (rng.rand(N) < 0.3).astype(np.uint8)
                o.call = h.call = 50000 + t
                o.multi_reset(so, some, cfg)
                h.multi_reset(sh, some, cfg)
                _same_state(so, sh, f'eager reset t={t}')
                mirror['valid'] = 0
    assert deaths > 0


def _class_env(N, K, S, seed, mode, cfg, **kw):
    from wurm_amd.envs import MultiSnake
    return MultiSnake(N, K, S, device='cuda:0', seed=seed, env_offset=7, observation_mode=mode,
                      boost=cfg['boost'], food_on_death_prob=cfg['food_on_death_prob'],
                      boost_cost_prob=cfg['boost_cost_prob'], food_mode=cfg['food_mode'], food_rate=cfg['food_rate'],
                      respawn_mode=cfg['respawn_mode'], reward_on_death=cfg['reward_on_death'],
                      agent_colours=cfg['colour_mode'], **kw)


@pytest.mark.parametrize('cfg_name,mode,shape', [('default', 'full', (24, 4, 14, 150)), ('train', 'partial_3', (20, 3, 12, 150)),
                                                 ('dense', 'full', (20, 3, 12, 120)), ('train', 'full', (6, 10, 36, 50))])
def test_class_loop_on_the_lazy_mirror_against_the_oracle(cfg_name, mode, shape):
    """nobody looks at the state for many iterations (the lazy form all the way), then it is read, edited through the alias,
    and one tensor is replaced; outputs every step and the state at those points against the oracle"""
    import torch
    cfg = CFGS[cfg_name]
    (N, K, S, T), seed = shape, 23
    with knobs(WURM_RESIDENT_MIN_ENVS=0):
        env = _class_env(N, K, S, seed, mode, cfg)
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

        alias = None
        for t in range(T):
            a = torch.randint(8, (K, N), generator=g)
            ac = a.cuda()
            obs, rew, dones, info = env.step({f'agent_{i}': ac[i] for i in range(K)})
            assert env._mirror is not None
            r = o.multi_step(st, a.numpy(), cfg, mode)
            for i in range(K):
                _same(obs[f'agent_{i}'].cpu().numpy(), r['obs'][i], f'obs {i} t={t}')
                _same(rew[f'agent_{i}'].cpu().numpy(), r['rewards'].reshape(N, K)[:, i], f'reward {i} t={t}')
                _same(info[f'size_{i}'].cpu().numpy(), r['size'].reshape(N, K)[:, i], f'size {i} t={t}')
            _same(dones['__all__'].cpu().numpy().astype(np.uint8), r['all_done'], f'all_done t={t}')
            if t % 3 == 2 and t > T // 2:        # (first half: the deferred reset only, so that the lazy form lasts)
                back = env.reset(dones['__all__'])
                o.multi_reset(st, r['all_done'], cfg, mode=mode)
                for i in range(K):
                    _same(back[f'agent_{i}'].cpu().numpy(), o.last_reset_obs[i], f'reset obs {i} t={t}')
            else:
                env.reset(dones['__all__'], return_observations=False)
                o.multi_reset(st, r['all_done'], cfg)
            if t < T // 2:
                assert env._mc.resident_lazy == 1 and (t < 2 or env._mc.resident_valid == 1)
            if t == T // 2:
                check_state(f't={t}')            # the first look: written out, eager from now on
                assert env._mc.resident_lazy == 0
                alias = env.foods
            if t in (T // 2 + 5, T // 2 + 20):   # in-place edits through the alias, found by the version counter
                alias[1, 0, 3, 3] = 1.0
                st['foods'][1, 0, 3, 3] = 1.0
            if t == T // 2 + 30:                 # a state tensor is replaced
                nb = env.bodies.clone()
                env.bodies = nb
            if t == T - 10:
                env.check_consistency() if (o.multi_check(st) == 0).all() else None
        check_state('final')


# ------------------------------------------------------------------ rollouts on the mirror (wurm_multi_rollout_resident)

@pytest.mark.parametrize('cfg_name,mode,shape', [('default', 'full', (19, 4, 14, 9)), ('train', 'full', (10, 10, 36, 5)),
                                                 ('dense', 'full', (21, 3, 12, 11)), ('train', 'full', (13, 4, 25, 7)),
                                                 ('train', 'partial_5', (12, 4, 25, 8)), ('dense', 'partial_2', (9, 6, 14, 6)),
                                                 ('default', 'full', (8, 12, 20, 5))])   # (12 snakes: no class codes)
@pytest.mark.parametrize('group', [True, False])
def test_class_rollouts_and_steps_interleaved_on_the_mirror(cfg_name, mode, shape, group):
    """env.rollout on the resident mirror (wurm_multi_rollout_resident): fused rollouts and per-call steps take turns on the
    same env object — the mirror made by a rollout is read by the next step and the other way round, lazily (the fp32
    tensors are looked at only now and then), then with a tensor held by the caller (eager) and edited in place — every
    output of every step and the state at the looks against the oracle.  The grouped writer ('full', at most 10 snakes, large
    batches) and the one-wave-per-env rollout (crops, more snakes) keep the mirror; group=False with 'full' observations is
    the two-wave form, which does not — the library writes a lazy mirror out itself and works on the tensors."""
    import torch
    cfg = CFGS[cfg_name]
    (N, K, S, T), seed = shape, 41
    with knobs(WURM_RESIDENT_MIN_ENVS=0, WURM_MULTI_GROUP_MIN_ENVS=0 if group else 1 << 40):
        env = _class_env(N, K, S, seed, mode, cfg)
        o = OracleBackend(seed=seed, env_offset=7)
        st = _o.multi_empty_state(N, K, S)
        st['colours'][...] = o.multi_colours(N, K, cfg['colour_mode'] == 'fixed', call=0)
        o.call = 1
        assert o.multi_reset(st, np.ones(N), cfg) == 0
        g = torch.Generator().manual_seed(9)

        def check_state(what):
            _same(env.foods.cpu().numpy(), st['foods'], what + ' foods')
            _same(env.heads.cpu().numpy(), st['heads'], what + ' heads')
            _same(env.bodies.cpu().numpy(), st['bodies'], what + ' bodies')
            _same(env.dones.cpu().numpy().astype(np.uint8), st['dones'], what + ' dones')
            _same(env.orientations.cpu().numpy(), st['orientations'], what + ' orientations')

        def rollout(what, with_obs=True):
            a = torch.randint(8, (T, K, N), generator=g)
            out = env.rollout(a.cuda(), return_observations=with_obs)
            ref = o.multi_rollout(st, a.numpy(), cfg, mode)
            if with_obs:
                _same(out['observations'].cpu().numpy().reshape(ref['obs'].shape), ref['obs'], what + ' obs')
            else:
                assert out['observations'] is None
            _same(out['all_done'].cpu().numpy().astype(np.uint8), ref['all_done'], what + ' all_done')
            _same(out['rewards'].cpu().numpy().transpose(0, 2, 1).reshape(T, -1), ref['rewards'].reshape(T, -1), what + ' rewards')
            _same(out['size'].cpu().numpy().transpose(0, 2, 1).reshape(T, -1), ref['size'].reshape(T, -1), what + ' size')

        def steps(n, what):
            for t in range(n):
                a = torch.randint(8, (K, N), generator=g)
                ac = a.cuda()
                obs, rew, dones, info = env.step({f'agent_{i}': ac[i] for i in range(K)})
                r = o.multi_step(st, a.numpy(), cfg, mode)
                for i in range(K):
                    _same(obs[f'agent_{i}'].cpu().numpy(), r['obs'][i], f'{what} obs {i} t={t}')
                _same(dones['__all__'].cpu().numpy().astype(np.uint8), r['all_done'], f'{what} all_done t={t}')
                env.reset(dones['__all__'], return_observations=False)
                o.multi_reset(st, r['all_done'], cfg)

        rollout('first rollout (makes the mirror)')
        if cfg_name == 'dense':
            check_state('a look right after the first rollout')   # (no step() has run yet: the rollout itself must have told the
                                                                  # library which tensors a lazy mirror is written out to)
        kept = group or mode != 'full' or K > 10      # (else: the two-wave form, which works on the tensors)
        assert env._mirror is not None and (cfg_name == 'dense' or (env._mc.resident_lazy == 1 and env._mc.resident_valid == (1 if kept else 0)))
        steps(4, 'steps after a rollout')
        rollout('second rollout')                # (a postponed reset is pending: applied first)
        rollout('third rollout')
        rollout('a rollout without observations', with_obs=False)   # (the one-wave kernel, whatever the mode)
        steps(3, 'steps again')
        check_state('first look')                # written out; the lazy form may end here
        rollout('rollout after a look')
        alias = env.foods                        # the caller holds a tensor: eager from now on
        assert env._mc.resident_lazy == 0
        rollout('eager rollout')
        check_state('after the eager rollout')
        alias[1, 0, 3, 3] = 1.0                  # an in-place edit through the alias, found by the version counter
        st['foods'][1, 0, 3, 3] = 1.0
        rollout('rollout after an edit')
        steps(3, 'steps at the end')
        check_state('final')


# ------------------------------------------------------------------ check_consistency inside the step launch

@pytest.mark.parametrize('cfg_name,mode,shape', [('default', 'full', (24, 4, 14, 120)), ('train', 'partial_3', (20, 3, 12, 120)),
                                                 ('dense', 'full', (20, 3, 12, 100)), ('train', 'full', (6, 10, 36, 40))])
@pytest.mark.parametrize('reset_obs', [True, 'dropped', False])
def test_check_consistency_from_the_step_launch(cfg_name, mode, shape, reset_obs):
    """experiments/speeds.py:30-38 — `step; reset(done['__all__']); check_consistency()` every iteration: the masks come out
    of the step launch (no pass over the fp32 tensors, the postponed reset stays postponed) and equal the oracle's checker
    on the oracle's state, env by env"""
    import torch
    cfg = CFGS[cfg_name]
    (N, K, S, T), seed = shape, 31
    with knobs(WURM_RESIDENT_MIN_ENVS=0):
        env = _class_env(N, K, S, seed, mode, cfg)
        o = OracleBackend(seed=seed, env_offset=7)
        st = _o.multi_empty_state(N, K, S)
        st['colours'][...] = o.multi_colours(N, K, cfg['colour_mode'] == 'fixed', call=0)
        o.call = 1
        assert o.multi_reset(st, np.ones(N), cfg) == 0
        g = torch.Generator().manual_seed(5)
        from_launch = 0
        for t in range(T):
            a = torch.randint(8, (K, N), generator=g)
            ac = a.cuda()
            obs, rew, dones, info = env.step({f'agent_{i}': ac[i] for i in range(K)})
            r = o.multi_step(st, a.numpy(), cfg, mode)
            for i in range(K):
                _same(obs[f'agent_{i}'].cpu().numpy(), r['obs'][i], f'obs {i} t={t}')
            if reset_obs == 'dropped':
                # what reset returns is dropped, as speeds.py:33 does: after three of those the steps stop precomputing it
                # (round 6: _LazyResetObs) and with it the masks of the reset state — the checker then runs over the tensors
                env.reset(dones['__all__'])
                o.multi_reset(st, r['all_done'], cfg, mode=mode)
            elif reset_obs:
                kept = env.reset(dones['__all__'])   # (a caller that keeps it: the step launches go on precomputing it)
                assert len(kept) == K
                o.multi_reset(st, r['all_done'], cfg, mode=mode)
            else:
                env.reset(dones['__all__'], return_observations=False)
                o.multi_reset(st, r['all_done'], cfg)
            want = o.multi_check(st).astype(np.int64)
            m = env._step_check_mask()
            if m is not None:
                got = m.cpu().numpy().astype(np.int64)
                known = got != -1
                _same(got[known], want[known], f'mask t={t}')
                from_launch += int(known.all())
                assert reset_obs or bool((known | (r['all_done'] != 0)).all())
            launches_before = env._steps
            if (want == 0).all():
                env.check_consistency()
            else:
                with pytest.raises(RuntimeError):
                    env.check_consistency()
            assert env._steps == launches_before
        # (without the reset observation the launch cannot vouch for envs the postponed reset rebuilds: those iterations
        # run the checker over the tensors)
        assert from_launch > (T // 2 if reset_obs is True else 0), from_launch
        if reset_obs is True:
            assert env._pending            # the checker never forced the postponed reset out


def _mirror_views(buf, N, K, S):
    import torch
    C = S * S
    nb, nf = (2 * K * C + 15) // 16 * 16, (C + 15) // 16 * 16
    per = nb + nf + (12 * K + 15) // 16 * 16
    m = buf.view(N, per)
    return m, nb, nf


@pytest.mark.parametrize('K,S', [(3, 12), (10, 36)])
def test_abi_masks_of_a_planted_inconsistent_image_equal_the_checker(hip, K, S):
    """the dynamics keep every invariant, so non-zero masks are planted: the mirror is edited by hand (an overlap, a hole in
    a body, food under a head, a segment above the head's value), written out, and stepped — by the oracle from the fp32
    tensors, by the library from the image; outputs, state and masks must agree"""
    import torch
    cfg = CFGS['noboost']
    N, C = 12, S * S
    rng = np.random.RandomState(7)
    o, h = OracleBackend(seed=5, env_offset=3), hip(seed=5, env_offset=3)
    so = _o.multi_empty_state(N, K, S)
    so['colours'][...] = o.multi_colours(N, K, cfg['colour_mode'] == 'fixed', call=0)
    o.call = 1
    assert o.multi_reset(so, np.ones(N), cfg) == 0
    sh = {k: v.copy() for k, v in so.items()}
    mirror = {'valid': 0, 'lazy': True, 'sync': True}
    seen = 0
    with knobs(WURM_RESIDENT_MIN_ENVS=0):
        call = 2
        for t in range(40):
            a = rng.randint(0, 4, size=(K, N)).astype(np.int64)
            if t % 4 == 2:  # plant: edit the image, write it out, adopt what it says as the common state
                m, nb, nf = _mirror_views(mirror['buf'], N, K, S)
                body = m[:, :nb].contiguous().view(torch.int16)[:, :K * C].clone().view(N, K, C)
                food = m[:, nb:nb + C].clone()
                sc = m[:, nb + nf:nb + nf + 12 * K].contiguous().view(torch.int32).view(N, 3, K)
                tclk, hc = sc[:, 0], sc[:, 1]
                for e in range(N):
                    kind = (e + t) % 4
                    s0 = int(rng.randint(K))
                    live = (body[e, s0].to(torch.int32) > tclk[e, s0]) & (tclk[e, s0] < 0x7000)
                    cells = torch.nonzero(live).flatten()
                    if len(cells) < 3:
                        continue
                    order = torch.argsort(body[e, s0][cells].to(torch.int32))
                    mid = int(cells[order[len(cells) // 2]])            # a middle segment: still there after the next decay
                    if kind == 0 and K > 1:      # overlap: another living snake gets a segment on that cell
                        s1 = (s0 + 1) % K
                        if int(tclk[e, s1]) < 0x7000:
                            body[e, s1, mid] = int(tclk[e, s1]) + 3
                    elif kind == 1:              # a hole in the body
                        body[e, s0, mid] = 0
                    elif kind == 2:              # a middle segment larger than the head's: the head is not at the end
                        Ls = sc[:, 2]
                        body[e, s0, mid] = int(tclk[e, s0]) + int(Ls[e, s0]) + 2
                        Ls[e, s0] += 2           # (the image's length slot is the largest body value, as load_env derives it)
                m[:, :nb].view(torch.int16)[:, :K * C] = body.view(N, K * C)
                m[:, nb:nb + C] = food
                m[:, nb + nf:nb + nf + 12 * K] = sc.contiguous().view(N, 3 * K).view(torch.uint8).view(N, 12 * K)
                # write the edited image out through the library and take it as the state of both sides
                import ctypes
                from wurm_amd import _lib
                blk = _lib.MultiCall()
                d = mirror['dev']
                blk.foods, blk.heads, blk.bodies = d['foods'].data_ptr(), d['heads'].data_ptr(), d['bodies'].data_ptr()
                blk.num_envs, blk.num_snakes, blk.size = N, K, S
                blk.resident, blk.resident_valid, blk.resident_lazy = mirror['buf'].data_ptr(), 1, 1
                _lib.check(h.lib.wurm_multi_resident_flush(ctypes.addressof(blk), h._stream()), 'flush')
                torch.cuda.synchronize()
                for k in ('foods', 'heads', 'bodies'):
                    sh[k][...] = d[k].cpu().numpy()
                    so[k][...] = sh[k]
            o.call = call
            ro = o.multi_step(so, a, cfg, 'full')
            rh = h.multi_step_reset(sh, a, cfg, 'full', call=call, resident=mirror)
            _same_state(so, sh, f'state t={t}')
            for k in ro:
                _same(ro[k], rh[k], f'{k} t={t}')
            want = o.multi_check(so).astype(np.int64)
            got = mirror['masks'][0].cpu().numpy().astype(np.int64)
            _same(got, want, f'mask t={t}')   # (t = 0: read from fp32 planes that hold nothing the image cannot)
            seen += int((want != 0).sum())
            call += 2
    assert seen > 0


def test_long_lived_snakes_cross_the_clock_rebase():
    """a small WURM_MULTI_CLOCK_REBASE cannot be set at run time, so the snakes are kept alive long enough instead: two
    snakes circling for 0x3000 + steps would take too long — the record's clocks are moved forward by hand"""
    import torch
    from tests.test_hip_multi_vs_oracle import CFGS as _C
    cfg = _C['noboost']
    N, K, S, seed = 6, 2, 12, 4
    with knobs(WURM_RESIDENT_MIN_ENVS=0):
        env = _class_env(N, K, S, seed, 'full', cfg)
        o = OracleBackend(seed=seed, env_offset=7)
        st = _o.multi_empty_state(N, K, S)
        st['colours'][...] = o.multi_colours(N, K, cfg['colour_mode'] == 'fixed', call=0)
        o.call = 1
        assert o.multi_reset(st, np.ones(N), cfg) == 0
        g = torch.Generator().manual_seed(9)

        def one(t):
            a = torch.randint(4, (K, N), generator=g)
            ac = a.cuda()
            obs, rew, dones, info = env.step({f'agent_{i}': ac[i] for i in range(K)})
            r = o.multi_step(st, a.numpy(), cfg, 'full')
            for i in range(K):
                _same(obs[f'agent_{i}'].cpu().numpy(), r['obs'][i], f'obs {i} t={t}')
            env.reset(dones['__all__'], return_observations=False)
            o.multi_reset(st, r['all_done'], cfg)

        for t in range(5):
            one(t)
        # the mirror: [body u16 K*C | food u8 C | tclk, hc, L per snake] per env; move every live clock and tclk forward
        C = S * S
        nb, nf = (2 * K * C + 15) // 16 * 16, (C + 15) // 16 * 16
        per = nb + nf + (12 * K + 15) // 16 * 16
        m = env._mirror.view(N, per)
        body = m[:, :nb].contiguous().view(torch.int16)[:, :K * C].to(torch.int32).view(N, K, C)
        sc = m[:, nb + nf:nb + nf + 12 * K].contiguous().view(torch.int32).view(N, 3 * K)
        tclk = sc[:, :K].clone()
        alive = tclk < 0x7000
        shift = 0x3000 - 10
        live = (body > tclk[:, :, None]) & alive[:, :, None]
        body = torch.where(live, body + shift, body)
        tclk = torch.where(alive, tclk + shift, tclk)
        m[:, :nb].view(torch.int16)[:, :K * C] = body.view(N, K * C).to(torch.int16)
        sc[:, :K] = tclk
        m[:, nb + nf:nb + nf + 12 * K] = sc.contiguous().view(torch.uint8).view(N, 12 * K)
        for t in range(5, 60):
            one(t)
        sc = m[:, nb + nf:nb + nf + 12 * K].contiguous().view(torch.int32).view(N, 3 * K)
        assert int(sc[:, :K][sc[:, :K] < 0x7000].max()) < 0x3000 + 4     # re-based
        _same(env.bodies.cpu().numpy(), st['bodies'], 'final bodies')
        _same(env.foods.cpu().numpy(), st['foods'], 'final foods')
        _same(env.heads.cpu().numpy(), st['heads'], 'final heads')


def test_inference_mode_state_cannot_be_watched():
    import torch
    cfg = CFGS['default']
    with knobs(WURM_RESIDENT_MIN_ENVS=0):
        with torch.inference_mode():
            env = _class_env(8, 2, 12, 3, 'full', cfg)
            a = {f'agent_{i}': torch.zeros(8, dtype=torch.long, device='cuda:0') for i in range(2)}
            env.step(a)
            assert env._mirror is not None
            f = env.foods                     # no version counter: no mirror from now on
            assert env._mirror is None and not env._mc.resident
            f[0, 0, 5, 5] = 1.0
            ref = _class_env(8, 2, 12, 3, 'full', cfg)
            ref.step(a)
            ref.foods[0, 0, 5, 5] = 1.0
            o1, o2 = env.step(a)[0], ref.step(a)[0]
            for k in o1:
                assert torch.equal(o1[k], o2[k])


def test_multi_parity_suites_on_the_mirror():
    if os.environ.get('WURM_RESIDENT_MIN_ENVS') == '0':
        pytest.skip('already inside the forced run')
    env = dict(os.environ, WURM_RESIDENT_MIN_ENVS='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                        'tests/test_hip_multi_fused.py', 'tests/test_kat_multi_snake.py', 'tests/test_hip_multi_vs_oracle.py',
                        'tests/test_hip_golden.py', 'tests/test_recording.py'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
