"""Replays a golden fixture (tests/golden/*.npz, recorded from the real reference by make_golden.py) into a
backend and asserts bit-equality with what the reference produced at every step.

A backend is anything with the numpy-in / numpy-out methods of tests/backends.py: the CPU oracle
(not-gpu tests) or the HIP kernels reached through the C-ABI (gpu tests).
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: z[k] for k in z.files}


def _eq(got, want, what, t):
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, f'{what} shape {got.shape} != {want.shape} at t={t}'
    if got.dtype.kind == 'f' or want.dtype.kind == 'f':
        # bit-exact: compare the fp32 bit patterns, not values within a tolerance
        g = np.ascontiguousarray(got, np.float32).view(np.uint32)
        w = np.ascontiguousarray(want, np.float32).view(np.uint32)
        bad = np.argwhere(g != w)
    else:
        bad = np.argwhere(got.astype(np.int64) != want.astype(np.int64))
    assert len(bad) == 0, f'{what} mismatch at t={t}: {len(bad)} elements, first at {bad[0].tolist()}: ' \
                          f'got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}'


def replay_single(backend, fx, check_reset_obs=True):
    mode = str(fx['mode'])
    N, S, T = (int(v) for v in fx['meta'][:3])
    envs = fx['state0'].astype(np.float32)
    for t in range(T):
        a = fx['actions_in'][t].copy()
        obs, reward, done, sc, ec = backend.single_step(envs, a, mode, inject_food=fx['inject_food'][t])
        _eq(a, fx['actions_out'][t], 'sanitised actions', t)
        _eq(envs, fx['state_step'][t].astype(np.float32), 'post-step state', t)
        _eq(reward, fx['reward'][t], 'reward', t)
        _eq(done, fx['done'][t], 'done', t)
        _eq(sc, fx['self_collision'][t], 'self_collision', t)
        _eq(ec, fx['edge_collision'][t], 'edge_collision', t)
        _eq(obs, fx['obs_step'][t], 'step observation', t)
        if fx['reset_called'][t]:
            obs_r = backend.single_reset(envs, fx['reset_mask'][t], mode, inject_reset=fx['inject_reset'][t])
            _eq(envs, fx['state_reset'][t].astype(np.float32), 'post-reset state', t)
            if check_reset_obs:
                _eq(obs_r, fx['obs_reset'][t], 'reset observation', t)
    return envs


def replay_single_rollout(backend, fx):
    """Same tape through the fused multi-step entry point (reset after every step only)."""
    mode = str(fx['mode'])
    N, S, T = (int(v) for v in fx['meta'][:3])
    assert int(fx['meta'][4]) == 1
    envs = fx['state0'].astype(np.float32)
    actions = fx['actions_in'].copy()
    out = backend.single_rollout(envs, actions, mode, inject_food=fx['inject_food'], inject_reset=fx['inject_reset'])
    _eq(actions, fx['actions_out'], 'sanitised actions', 'all')
    _eq(out['reward'], fx['reward'], 'reward', 'all')
    _eq(out['done'], fx['done'], 'done', 'all')
    _eq(out['self_collision'], fx['self_collision'], 'self_collision', 'all')
    _eq(out['edge_collision'], fx['edge_collision'], 'edge_collision', 'all')
    _eq(out['obs'], fx['obs_step'], 'step observations', 'all')
    _eq(envs, fx['state_reset'][-1].astype(np.float32), 'final state', T - 1)


def replay_single_rollout_vs_oracle(backend, oracle, fx, mode):
    """The tape's recorded random outcomes injected through the fused entry point in ANOTHER observation mode than the one
    the reference was recorded in: everything but the observation is still checked against the reference's record, the
    observation against the oracle driven by the same injection."""
    N, S, T = (int(v) for v in fx['meta'][:3])
    assert int(fx['meta'][4]) == 1
    envs, envs_o = fx['state0'].astype(np.float32), fx['state0'].astype(np.float32)
    actions, actions_o = fx['actions_in'].copy(), fx['actions_in'].copy()
    out = backend.single_rollout(envs, actions, mode, inject_food=fx['inject_food'], inject_reset=fx['inject_reset'])
    want = oracle.single_rollout(envs_o, actions_o, mode, inject_food=fx['inject_food'], inject_reset=fx['inject_reset'])
    _eq(actions, fx['actions_out'], 'sanitised actions', 'all')
    _eq(out['reward'], fx['reward'], 'reward', 'all')
    _eq(out['done'], fx['done'], 'done', 'all')
    _eq(out['self_collision'], fx['self_collision'], 'self_collision', 'all')
    _eq(out['edge_collision'], fx['edge_collision'], 'edge_collision', 'all')
    _eq(envs, fx['state_reset'][-1].astype(np.float32), 'final state', T - 1)
    _eq(out['obs'], want['obs'], f'step observations ({mode}) against the oracle', 'all')
    _eq(envs, envs_o, 'final state against the oracle', T - 1)


def replay_grid(backend, fx):
    mode = str(fx['mode'])
    N, S, T = (int(v) for v in fx['meta'][:3])
    start = (int(fx['meta'][4]), int(fx['meta'][5]))
    envs = fx['state0'].astype(np.float32)
    for t in range(T):
        a = fx['actions_in'][t].copy()
        obs, reward, done, ec = backend.grid_step(envs, a, mode, inject_food=fx['inject_food'][t])
        _eq(a, fx['actions_out'][t], 'actions (must be untouched)', t)
        _eq(envs, fx['state_step'][t].astype(np.float32), 'post-step state', t)
        _eq(reward, fx['reward'][t], 'reward', t)
        _eq(done, fx['done'][t], 'done', t)
        _eq(ec, fx['edge_collision'][t], 'edge_collision', t)
        _eq(obs, fx['obs_step'][t], 'step observation', t)
        obs_r = backend.grid_reset(envs, fx['reset_mask'][t], start, mode, inject_reset=fx['inject_reset'][t])
        _eq(envs, fx['state_reset'][t].astype(np.float32), 'post-reset state', t)
        _eq(obs_r, fx['obs_reset'][t], 'reset observation', t)
    return envs


def replay_grid_rollout(backend, fx):
    mode = str(fx['mode'])
    N, S, T = (int(v) for v in fx['meta'][:3])
    start = (int(fx['meta'][4]), int(fx['meta'][5]))
    envs = fx['state0'].astype(np.float32)
    actions = fx['actions_in'].copy()
    out = backend.grid_rollout(envs, actions, start, mode, inject_food=fx['inject_food'],
                               inject_reset=fx['inject_reset'])
    _eq(out['reward'], fx['reward'], 'reward', 'all')
    _eq(out['done'], fx['done'], 'done', 'all')
    _eq(out['edge_collision'], fx['edge_collision'], 'edge_collision', 'all')
    _eq(out['obs'], fx['obs_step'], 'step observations', 'all')
    _eq(envs, fx['state_reset'][-1].astype(np.float32), 'final state', T - 1)


# ------------------------------------------------------------------------------------------- MultiSnake

def load_multi(name):
    fx = load(name)
    N, K, S, T = (int(v) for v in fx['meta'][:4])
    for k in ('death_a', 'death_b', 'rate'):
        packed = fx['inj_' + k]
        fx['inj_' + k] = np.stack([np.unpackbits(packed[t])[:N * S * S].reshape(N, S, S) for t in range(T)])
    fx['cfg_dict'] = eval(str(fx['cfg']), {'__builtins__': {}}, {})
    return fx


def multi_state(fx, prefix, t=None):
    def g(k):
        a = fx[prefix + k]
        return a if t is None else a[t]
    return dict(foods=g('foods').astype(np.float32), heads=g('heads').astype(np.float32),
                bodies=g('bodies').astype(np.float32), dones=g('dones').astype(np.uint8).copy(),
                orientations=g('orientations').astype(np.int64).copy(),
                boost_this_step=g('boost_this_step').astype(np.uint8).copy(),
                colours=g('colours').astype(np.int16).copy())


def _eq_state(st, fx, prefix, t, what):
    for k in ('foods', 'heads', 'bodies'):
        _eq(st[k], fx[prefix + k][t].astype(np.float32), f'{what} {k}', t)
    for k in ('dones', 'orientations', 'boost_this_step', 'colours'):
        _eq(st[k], fx[prefix + k][t], f'{what} {k}', t)


def _obs_from_code(code):
    return code.astype(np.float32) / np.float32(255)


def replay_multi(backend, fx):
    mode = str(fx['mode'])
    N, K, S, T = (int(v) for v in fx['meta'][:4])
    cfg = fx['cfg_dict']
    st = multi_state(fx, 'state0_')
    for t in range(T):
        inj = {k: fx['inj_' + k][t] for k in ('death_a', 'cost', 'death_b', 'rate', 'food_cell')}
        r = backend.multi_step(st, fx['actions'][t], cfg, mode, inject=inj)
        _eq_state(st, fx, 'step_', t, 'post-step')
        _eq(st['dones'], fx['dones_out'][t], 'dones', t)
        _eq(r['rewards'], fx['rewards'][t], 'rewards', t)
        _eq(r['snake_collision'], fx['snake_collision'][t], 'snake_collision', t)
        _eq(r['edge_collision'], fx['edge_collision'][t], 'edge_collision', t)
        _eq(r['food'], fx['food'][t], 'food consumed', t)
        _eq(r['size'], fx['size'][t], 'sizes', t)
        _eq(st['boost_this_step'], fx['boost'][t], 'boost info', t)
        _eq(r['all_done'], fx['all_done'][t], '__all__ done', t)
        _eq(r['obs'], _obs_from_code(fx['obs_step'][t]), 'step observation', t)
        rinj = {k: fx['rinj_' + k][t] for k in ('create', 'create_food', 'colours', 'respawn')}
        backend.multi_reset(st, fx['all_done'][t], cfg, inject=rinj)
        _eq_state(st, fx, 'reset_', t, 'post-reset')
        if 'obs_reset' in fx:
            _eq(backend.multi_observe(st, mode), _obs_from_code(fx['obs_reset'][t]), 'reset observation', t)
    return st


def replay_multi_rollout(backend, fx):
    """The whole MultiSnake tape through the fused rollout entry point, with the reference's recorded outcomes injected."""
    mode = str(fx['mode'])
    N, K, S, T = (int(v) for v in fx['meta'][:4])
    cfg = fx['cfg_dict']
    st = multi_state(fx, 'state0_')
    inject = {k: fx['inj_' + k] for k in ('death_a', 'cost', 'death_b', 'rate', 'food_cell')}
    rinj = {k: fx['rinj_' + k] for k in ('create', 'create_food', 'colours', 'respawn')}
    r = backend.multi_rollout(st, fx['actions'], cfg, mode, inject=inject, reset_inject=rinj)
    _eq(r['rewards'], fx['rewards'], 'rewards', 'all')
    _eq(r['dones'], fx['dones_out'], 'dones', 'all')
    _eq(r['snake_collision'], fx['snake_collision'], 'snake_collision', 'all')
    _eq(r['edge_collision'], fx['edge_collision'], 'edge_collision', 'all')
    _eq(r['food'], fx['food'], 'food consumed', 'all')
    _eq(r['size'], fx['size'], 'sizes', 'all')
    _eq(r['boost'], fx['boost'], 'boost info', 'all')
    _eq(r['all_done'], fx['all_done'], '__all__ done', 'all')
    _eq(r['obs'], _obs_from_code(fx['obs_step']), 'step observations', 'all')
    _eq_state(st, fx, 'reset_', T - 1, 'final')
