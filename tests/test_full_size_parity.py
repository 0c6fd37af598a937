"""Parity AT BASELINE.json's FULL SIZES: cfg3 whole (65 536 x 9 x 9 partial_2), cfg4 (MultiSnake
4 096 x 25 x 25 x 4, 'full') and cfg5 (8 192 x 36 x 36, default RGB).

The oracle cannot step 65 536 envs in seconds, but it does not have to: every random draw is keyed by the GLOBAL env
id, so the oracle can follow any single env of the batch on its own (`env_offset = id`, one env) and must reproduce
that env's slice of the full-size GPU batch bit for bit.  Per config:
  * fused rollout == per-call step()/reset() loop over the whole batch (compared on the device, every output);
  * the device-side invariant checker reports 0 after every reset of the per-call loop;
  * the oracle on a strided subset of env ids (first / last env, wave, workgroup and XCD-round-robin boundaries,
    random ids) == the same envs of the full-size batch: state, sanitised actions, rewards, dones, flags, observations.
"""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests import replay
from tests.backends import OracleBackend

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _ids(N, extra=24, seed=0):
    fixed = [0, 1, 3, 4, 63, 64, 255, 256, 257, 1023, 1024, 2047, 2048, 4095, 4096, 8191, 8192, 32767, 32768,
             N // 2, N - 2, N - 1]
    rng = np.random.RandomState(seed)
    ids = sorted(set(i for i in fixed if 0 <= i < N) | set(int(i) for i in rng.randint(0, N, size=extra)))
    return ids


def _eq_dev(a, b, what):
    assert a.shape == b.shape and a.dtype == b.dtype, f'{what}: {a.shape}/{a.dtype} vs {b.shape}/{b.dtype}'
    if a.dtype == torch.float32:
        same = torch.equal(a.view(torch.int32), b.view(torch.int32))
    else:
        same = torch.equal(a, b)
    if not same:
        bad = (a != b).nonzero()
        raise AssertionError(f'{what}: {len(bad)} mismatches, first at {bad[0].tolist()}')


@pytest.mark.parametrize('N,S,mode,T', [
    (65536, 9, 'partial_2', 64),    # BASELINE configs[2], whole batch on one GPU
    (8192, 36, 'default', 16),      # BASELINE configs[4]
])
def test_single_snake_full_size(N, S, mode, T):
    from wurm_amd.envs import SingleSnake
    from wurm_amd import _lib
    seed = 4242
    g = torch.Generator(device=DEV).manual_seed(9)
    actions = torch.randint(4, (T, N), generator=g, device=DEV)
    a_host = actions.cpu().numpy()

    # (1) fused rollout over the whole batch
    env_r = SingleSnake(N, S, observation_mode=mode, device=DEV, seed=seed)
    state0 = env_r.envs.clone()
    a_roll = actions.clone()
    out = env_r.rollout(a_roll)

    # (2) the per-call loop over the whole batch, invariant checker after every reset
    env_l = SingleSnake(N, S, observation_mode=mode, device=DEV, seed=seed)
    _eq_dev(env_l.envs, state0, 'fresh envs')
    a_loop = actions.clone()
    err = torch.empty(N, dtype=torch.int32, device=DEV)
    total_done = 0
    for t in range(T):
        obs, r, d, info = env_l.step(a_loop[t])
        _eq_dev(obs.reshape(N, -1), out['observations'][t].reshape(N, -1), f'obs t={t}')
        _eq_dev(r[:, 0], out['rewards'][t], f'reward t={t}')
        _eq_dev(d[:, 0], out['dones'][t], f'done t={t}')
        _eq_dev(info['self_collision'], out['self_collision'][t], f'self_collision t={t}')
        _eq_dev(info['edge_collision'], out['edge_collision'][t], f'edge_collision t={t}')
        total_done += int(d.sum())
        env_l.reset(d, return_observations=(t % 4 == 0))
        e = env_l.envs
        rc = _lib.lib().wurm_single_check(_lib.ptr(e), _lib.ptr(err), _lib.i64(N), S, _lib.stream_ptr())
        assert rc == 0
        assert not bool(err.any()), f'invariant checker: {int((err != 0).sum())} inconsistent envs after reset t={t}'
    _eq_dev(a_loop, a_roll, 'sanitised actions')
    _eq_dev(env_l.envs, env_r.envs, 'final state')
    assert total_done > N // 8  # the reset path was exercised at scale

    # (3) the oracle follows single envs of the batch by their global id
    ids = _ids(N)
    final = env_r.envs[ids].cpu().numpy()
    obs_sub = out['observations'][:, ids].cpu().numpy()
    for j, gid in enumerate(ids):
        ref = np.zeros((1, 3, S, S), np.float32)
        oracle.single_reset(ref, np.ones(1, np.uint8), 'none', seed=seed, call=0, env_offset=gid)
        replay._eq(state0[gid:gid + 1].cpu().numpy(), ref, f'fresh env {gid}', 0)
        a = np.ascontiguousarray(a_host[:, gid:gid + 1])
        exp = oracle.single_rollout(ref, a, mode, seed=seed, call0=1, env_offset=gid)
        replay._eq(a_roll[:, gid].cpu().numpy(), a[:, 0], f'env {gid} sanitised actions', '-')
        replay._eq(obs_sub[:, j].reshape(T, -1), exp['obs'].reshape(T, -1), f'env {gid} observations', '-')
        replay._eq(out['rewards'][:, gid].cpu().numpy(), exp['reward'][:, 0], f'env {gid} rewards', '-')
        replay._eq(out['dones'][:, gid].cpu().numpy(), exp['done'][:, 0], f'env {gid} dones', '-')
        replay._eq(out['self_collision'][:, gid].cpu().numpy(), exp['self_collision'][:, 0], f'env {gid} selfc', '-')
        replay._eq(out['edge_collision'][:, gid].cpu().numpy(), exp['edge_collision'][:, 0], f'env {gid} edgec', '-')
        replay._eq(final[j:j + 1], ref, f'env {gid} final state', '-')


MULTI_DEFAULTS = dict(boost=True, food_on_death_prob=0.5, boost_cost_prob=0.5, food_mode='only_one', food_rate=5e-4,
                      reward_on_death=-1, respawn_mode='all', colour_mode='random')


def test_multi_snake_cfg4_full_size():
    """BASELINE configs[3]: MultiSnake(4096, 4, 25), constructor defaults, 'full' observations."""
    from wurm_amd.envs import MultiSnake
    N, K, S, T, seed = 4096, 4, 25, 16, 777
    g = torch.Generator(device=DEV).manual_seed(3)
    actions = torch.randint(8, (T, K, N), generator=g, device=DEV)
    a_host = actions.cpu().numpy()

    env_r = MultiSnake(N, K, S, device=DEV, seed=seed)
    start = {k: getattr(env_r, k).clone() for k in ('foods', 'heads', 'bodies', 'orientations', 'agent_colours')}
    out = env_r.rollout(actions)

    env_l = MultiSnake(N, K, S, device=DEV, seed=seed)
    for k, v in start.items():
        _eq_dev(getattr(env_l, k), v, f'fresh {k}')
    deaths = 0
    for t in range(T):
        obs, rew, dones, info = env_l.step({f'agent_{i}': actions[t, i] for i in range(K)})
        for i in range(K):
            _eq_dev(obs[f'agent_{i}'], out['observations'][t, i], f'obs agent {i} t={t}')
            _eq_dev(rew[f'agent_{i}'], out['rewards'][t, i], f'reward agent {i} t={t}')
            _eq_dev(dones[f'agent_{i}'], out['dones'][t, i], f'done agent {i} t={t}')
            _eq_dev(info[f'snake_collision_{i}'], out['snake_collision'][t, i], f'snake_collision {i} t={t}')
            _eq_dev(info[f'edge_collision_{i}'], out['edge_collision'][t, i], f'edge_collision {i} t={t}')
            _eq_dev(info[f'food_{i}'], out['food'][t, i], f'food {i} t={t}')
            _eq_dev(info[f'size_{i}'], out['size'][t, i], f'size {i} t={t}')
            _eq_dev(info[f'boost_{i}'], out['boost'][t, i], f'boost {i} t={t}')
        _eq_dev(dones['__all__'], out['all_done'][t], f'all_done t={t}')
        deaths += int(out['dones'][t].sum())
        env_l.reset(dones['__all__'], return_observations=False)
        env_l.check_consistency()   # device-side checker, raises on any inconsistent env
    for k in ('foods', 'heads', 'bodies', 'dones', 'orientations', 'agent_colours'):
        _eq_dev(getattr(env_l, k), getattr(env_r, k), f'final {k}')
    assert deaths > 0

    # the oracle follows single envs by global id
    C = S * S
    for gid in _ids(N, extra=12):
        o = OracleBackend(seed=seed, env_offset=gid)
        st = oracle.multi_empty_state(1, K, S)
        st['colours'][...] = o.multi_colours(1, K, False, call=0)
        o.call = 1
        assert o.multi_reset(st, np.ones(1), MULTI_DEFAULTS) == 0
        replay._eq(start['foods'][gid].cpu().numpy(), st['foods'][0], f'env {gid} fresh foods', 0)
        replay._eq(start['bodies'][gid * K:(gid + 1) * K].cpu().numpy(), st['bodies'], f'env {gid} fresh bodies', 0)
        replay._eq(start['agent_colours'][gid * K:(gid + 1) * K].cpu().numpy(), st['colours'], f'env {gid} colours', 0)
        exp = o.multi_rollout(st, np.ascontiguousarray(a_host[:, :, gid:gid + 1]), MULTI_DEFAULTS, 'full')
        replay._eq(out['observations'][:, :, gid].cpu().numpy().reshape(T, K, 1, 3, S, S), exp['obs'],
                   f'env {gid} observations', '-')
        replay._eq(out['rewards'][:, :, gid].cpu().numpy().reshape(T, K), exp['rewards'].reshape(T, K),
                   f'env {gid} rewards', '-')
        replay._eq(out['dones'][:, :, gid].cpu().numpy().reshape(T, K), exp['dones'].reshape(T, K), f'env {gid} dones', '-')
        replay._eq(out['size'][:, :, gid].cpu().numpy().reshape(T, K), exp['size'].reshape(T, K), f'env {gid} sizes', '-')
        replay._eq(out['all_done'][:, gid].cpu().numpy(), exp['all_done'][:, 0], f'env {gid} all_done', '-')
        replay._eq(env_r.foods[gid].cpu().numpy(), st['foods'][0], f'env {gid} final foods', '-')
        replay._eq(env_r.heads[gid * K:(gid + 1) * K].cpu().numpy(), st['heads'], f'env {gid} final heads', '-')
        replay._eq(env_r.bodies[gid * K:(gid + 1) * K].cpu().numpy(), st['bodies'], f'env {gid} final bodies', '-')
        assert C == st['foods'][0].size
