"""MultiSnake scenarios for make_golden.py (see its docstring: runs the REAL reference in the build container and
records data only).

The reference's random outcomes are captured without touching its code: `torch.rand` / `torch.rand_like` are
wrapped while `step()` runs (the Bernoulli draws of food-on-death, boost cost and rate food), and the bound
methods `_add_food`, `_get_food_addition`, `_create_envs`, `_get_snake_addition` of the env OBJECT are wrapped to
see which envs they acted on and what they returned (food cell, spawn cells, directions).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

SingleSnake, SimpleGridworld, MultiSnake, ref_utils = ref_shim.import_reference()
EPS = 1e-6


def pack(a):
    a = np.asarray(a)
    r = np.rint(a)
    assert np.array_equal(r, a) and r.min() >= 0
    return r.astype(np.uint8 if r.max() < 256 else np.uint16)


def obs_to_u8(obs_dict, K):
    """(K,N,3,h,w) uint8 RGB codes; checks that code/255 reproduces the reference's fp32 bits exactly."""
    o = np.stack([obs_dict[f'agent_{i}'].numpy() for i in range(K)])
    code = np.rint(o * 255.0).astype(np.uint8)
    back = code.astype(np.float32) / np.float32(255)
    assert np.array_equal(back.view(np.uint32), o.view(np.uint32)), 'observation is not rgb/255'
    return code


class Recorder(object):
    """Wraps one reference MultiSnake object."""

    def __init__(self, env):
        self.env = env
        self.N, self.K, self.S = env.num_envs, env.num_snakes, env.size
        self.rand_log = []
        self.in_step = False
        self.add_food_envs = None
        self.food_addition = None
        self.in_create = False
        self.create_out = None
        self.respawn_out = None
        self.respawn_envs = None

        self._rand, self._rand_like = torch.rand, torch.rand_like
        orig_add_food, orig_gfa = env._add_food, env._get_food_addition
        orig_create, orig_gsa = env._create_envs, env._get_snake_addition

        def add_food():
            fsum = env.foods.view(self.N, -1).sum(dim=-1)
            if env.food_mode == 'only_one':
                self.add_food_envs = (fsum < EPS).numpy().copy()
            else:
                self.add_food_envs = (fsum < env.max_food).numpy().copy()
            return orig_add_food()

        def get_food_addition(*a, **kw):
            out = orig_gfa(*a, **kw)
            if not self.in_create:
                self.food_addition = out.numpy().copy()
            return out

        def create_envs(n):
            self.in_create = True
            out = orig_create(n)
            self.in_create = False
            (foods, heads, bodies), orient = out
            self.create_out = (foods.numpy().copy(), bodies.numpy().copy(), orient.numpy().copy())
            return out

        def get_snake_addition(pathing, exception_on_failure):
            first = (env.dones.view(self.N, self.K).cumsum(dim=1) == 1).flatten() & env.dones.bool()
            self.respawn_envs = np.flatnonzero(first.view(self.N, self.K).any(dim=1).numpy())
            out = orig_gsa(pathing, exception_on_failure)
            new_bodies, new_heads, dirs, ok = out
            self.respawn_out = (new_bodies.numpy().copy(), dirs.numpy().copy(), ok.numpy().copy())
            return out

        env._add_food, env._get_food_addition = add_food, get_food_addition
        env._create_envs, env._get_snake_addition = create_envs, get_snake_addition

    # -- torch.rand capture while step() runs
    def _patch_rand(self):
        def rand(*a, **kw):
            t = self._rand(*a, **kw)
            self.rand_log.append(('rand', t.clone()))
            return t

        def rand_like(*a, **kw):
            t = self._rand_like(*a, **kw)
            self.rand_log.append(('rand_like', t.clone()))
            return t

        torch.rand, torch.rand_like = rand, rand_like

    def _unpatch_rand(self):
        torch.rand, torch.rand_like = self._rand, self._rand_like

    def step(self, actions):
        env, N, K, S = self.env, self.N, self.K, self.S
        self.rand_log, self.add_food_envs, self.food_addition = [], None, None
        self._patch_rand()
        try:
            obs, rewards, dones, info = env.step(actions)
        finally:
            self._unpatch_rand()
        log = list(self.rand_log)
        inj = dict(death_a=np.zeros((N, S, S), np.uint8), cost=np.zeros(N * K, np.uint8),
                   death_b=np.zeros((N, S, S), np.uint8), rate=np.zeros((N, S, S), np.uint8),
                   food_cell=np.full(N, -1, np.int32))
        boost_ran = bool(env.boost) and bool(env.boost_this_step.any())
        p = env.food_on_death_prob
        if boost_ran:
            if p > 0:
                kind, u = log.pop(0)
                assert kind == 'rand_like' and u.shape == (N, 1, S, S)
                inj['death_a'] = (u > (1 - p)).numpy().reshape(N, S, S).astype(np.uint8)
            kind, u = log.pop(0)
            assert kind == 'rand' and u.shape == (N * K,)
            inj['cost'] = (u < env.boost_cost_prob).numpy().astype(np.uint8)
        if p > 0:
            kind, u = log.pop(0)
            assert kind == 'rand_like' and u.shape == (N, 1, S, S)
            inj['death_b'] = (u > (1 - p)).numpy().reshape(N, S, S).astype(np.uint8)
        if env.food_mode == 'random_rate':
            kind, u = log.pop(0)
            idx = np.flatnonzero(self.add_food_envs)
            assert kind == 'rand' and u.shape == (len(idx), 1, S, S)
            inj['rate'][idx] = u.lt(env.food_rate).numpy().reshape(len(idx), S, S).astype(np.uint8)
        else:
            idx = np.flatnonzero(self.add_food_envs)
            if len(idx):
                add = self.food_addition.reshape(len(idx), S * S)
                for j, e in enumerate(idx):
                    nz = np.flatnonzero(add[j] > 0.5)
                    inj['food_cell'][e] = nz[0] if len(nz) else -1
        assert not log, f'unparsed random draws: {[(k, tuple(t.shape)) for k, t in log]}'
        return obs, rewards, dones, info, inj

    def reset(self, done_env, return_observations=True):
        env, N, K, S = self.env, self.N, self.K, self.S
        self.create_out, self.respawn_out, self.respawn_envs = None, None, None
        obs = env.reset(done_env, return_observations=return_observations)
        inj = dict(create=np.full((N, K, 2), -1, np.int32), create_food=np.full(N, -1, np.int32),
                   colours=env.agent_colours.numpy().astype(np.int16).copy(),
                   respawn=np.full((N, 2), -1, np.int32))
        idx = np.flatnonzero(done_env.numpy().reshape(N) != 0)
        if self.create_out is not None:
            foods, bodies, orient = self.create_out
            bodies = bodies.reshape(len(idx), K, S * S)
            orient = orient.reshape(len(idx), K)
            for j, e in enumerate(idx):
                for s in range(K):
                    (seed,) = np.flatnonzero(bodies[j, s] == 2)
                    inj['create'][e, s] = (seed, orient[j, s])
                nz = np.flatnonzero(foods[j].reshape(-1) > 0.5)
                inj['create_food'][e] = nz[0] if len(nz) else -1
        if self.respawn_out is not None:
            new_bodies, dirs, ok = self.respawn_out
            new_bodies = new_bodies.reshape(len(self.respawn_envs), S * S)
            for j, e in enumerate(self.respawn_envs):
                seed = np.flatnonzero(new_bodies[j] == 2)
                inj['respawn'][e] = (seed[0] if len(seed) else -1, dirs[j])
        return obs, inj


def state_of(env):
    return dict(foods=pack(env.foods.numpy()), heads=pack(env.heads.numpy()), bodies=pack(env.bodies.numpy()),
                dones=env.dones.numpy().astype(np.uint8).copy(), orientations=env.orientations.numpy().astype(np.int64).copy(),
                boost_this_step=env.boost_this_step.numpy().astype(np.uint8).copy(),
                colours=env.agent_colours.numpy().astype(np.int16).copy())


def per_agent(d, key, K, dtype):
    """dict of per-agent (N,) tensors -> (N*K) array in the reference's agent order env*K + i."""
    return np.stack([d[f'{key}{i}'].numpy() for i in range(K)], axis=1).reshape(-1).astype(dtype)


def record_multi(name, N, K, S, T, seed, n_actions=8, return_reset_obs=True, **kwargs):
    torch.manual_seed(seed)
    env = MultiSnake(num_envs=N, num_snakes=K, size=S, device='cpu', **kwargs)
    rec = Recorder(env)
    all_actions = torch.randint(n_actions, size=(T, K, N)).long()
    out = {'state0_' + k: v for k, v in state_of(env).items()}
    out['actions'] = all_actions.numpy().copy()
    cols = {}

    def put(key, val):
        cols.setdefault(key, []).append(val)

    for t in range(T):
        actions = {f'agent_{i}': all_actions[t, i].clone() for i in range(K)}
        obs, rewards, dones, info, inj = rec.step(actions)
        for i in range(K):
            assert torch.equal(actions[f'agent_{i}'], all_actions[t, i]), 'MultiSnake.step must not touch actions'
        for k, v in state_of(env).items():
            put('step_' + k, v)
        put('obs_step', obs_to_u8(obs, K))
        put('rewards', per_agent(rewards, 'agent_', K, np.float32))
        put('dones_out', per_agent(dones, 'agent_', K, np.uint8))
        put('all_done', dones['__all__'].numpy().astype(np.uint8))
        put('snake_collision', per_agent(info, 'snake_collision_', K, np.uint8))
        put('edge_collision', per_agent(info, 'edge_collision_', K, np.uint8))
        put('food', per_agent(info, 'food_', K, np.float32))
        put('size', per_agent(info, 'size_', K, np.float32))
        put('boost', per_agent(info, 'boost_', K, np.uint8))
        for k, v in inj.items():
            put('inj_' + k, np.packbits(v, axis=None) if k in ('death_a', 'death_b', 'rate') else v)
        obs_r, rinj = rec.reset(dones['__all__'], return_observations=return_reset_obs)
        for k, v in state_of(env).items():
            put('reset_' + k, v)
        if return_reset_obs:
            put('obs_reset', obs_to_u8(obs_r, K))
        for k, v in rinj.items():
            put('rinj_' + k, v)
        env.check_consistency()
    out.update({k: np.stack(v) for k, v in cols.items()})
    out['meta'] = np.array([N, K, S, T, seed], np.int64)
    out['mode'] = np.array(env.observation_mode)
    cfgd = dict(boost=env.boost, food_on_death_prob=env.food_on_death_prob, boost_cost_prob=env.boost_cost_prob,
                food_mode=env.food_mode, food_rate=env.food_rate, reward_on_death=env.reward_on_death,
                respawn_mode=env.respawn_mode, colour_mode=env.colour_mode)
    out['cfg'] = np.array(repr(cfgd))
    if name is None:  # in-memory use (tests/test_oracle_vs_live_reference.py)
        return out
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB; deaths {int(out["dones_out"].sum())}, '
          f'boost steps {int(out["boost"].sum())}, env resets {int(out["all_done"].sum())}, '
          f'food eaten {float(out["food"].sum()):.0f}')
    return out


SCENARIOS = {
    'multi_k2_s12_default': lambda: record_multi('multi_k2_s12_default', N=24, K=2, S=12, T=100, seed=31),
    # BASELINE.json cfg4 shape (25x25, 4 agents, constructor defaults) at fixture size
    'multi_k4_s25_default': lambda: record_multi('multi_k4_s25_default', N=12, K=4, S=25, T=80, seed=32),
    # the training-style dynamics of the reference's tests/test_multi_snake_env.py:100-104
    'multi_k4_s25_train': lambda: record_multi(
        'multi_k4_s25_train', N=12, K=4, S=25, T=100, seed=33, respawn_mode='any', food_mode='random_rate',
        boost_cost_prob=0.25, observation_mode='partial_5', food_on_death_prob=0.33, food_rate=2.5e-4),
    'multi_k3_s14_noboost': lambda: record_multi(
        'multi_k3_s14_noboost', N=16, K=3, S=14, T=80, seed=34, boost=False, respawn_mode='any',
        observation_mode='partial_3', agent_colours='fixed', food_on_death_prob=0.0),
    # crowded board: many collisions, head-to-head, failed respawns
    'multi_k6_s10_crowded': lambda: record_multi(
        'multi_k6_s10_crowded', N=8, K=6, S=14, T=80, seed=35, respawn_mode='any', food_mode='random_rate',
        food_rate=5e-3, food_on_death_prob=0.8, boost_cost_prob=0.9),
}

if __name__ == '__main__':
    for n in (sys.argv[1:] or list(SCENARIOS)):
        SCENARIOS[n]()
