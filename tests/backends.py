"""Numpy-in / numpy-out adapters with one interface over the two implementations the tests compare:

* OracleBackend — the CPU oracle (oracle/oracle.py); the checker.
* HipBackend    — the product's C-ABI (wurm_amd/csrc -> libwurm_hip.so) on cuda:0; defined in
                  tests/hip_backend.py so that not-gpu runs never touch the GPU library's compute calls.
"""
from oracle import oracle as _o


class OracleBackend(object):
    name = 'oracle'

    def __init__(self, seed=0, env_offset=0):
        self.seed = seed
        self.env_offset = env_offset
        self.call = 0

    def _next(self, n=1):
        c = self.call
        self.call += n
        return c

    # SingleSnake
    def single_step(self, envs, actions, mode, inject_food=None):
        return _o.single_step(envs, actions, mode, self.seed, self._next(), self.env_offset, inject_food)

    def single_reset(self, envs, done, mode, inject_reset=None):
        return _o.single_reset(envs, done, mode, self.seed, self._next(), self.env_offset, inject_reset)

    def single_observe(self, envs, mode):
        return _o.single_observe(envs, mode)

    def single_step_reset(self, envs, actions, mode, call, pre_done=None, pre_call=None, post_reset=False,
                          want_obs_after=False, inject_food=None, inject_reset=None, inject_pre_reset=None, grid=None):
        """The definition wurm_single_step_reset / wurm_grid_step_reset must reproduce, composed from the per-call
        oracle functions with explicit counters: [reset(pre_done, pre_call)]; step(call); [reset(done, call + 1)].
        grid: None for SingleSnake, the start location for SimpleGridworld."""
        def reset(e, d, m, c, inj):
            if grid is None:
                return _o.single_reset(e, d, m, self.seed, c, self.env_offset, inj)
            return _o.grid_reset(e, d, grid, m, self.seed, c, self.env_offset, inj)
        if pre_done is not None:
            reset(envs, pre_done, 'none', pre_call, inject_pre_reset)
        if grid is None:
            obs, reward, done, sc, ec = _o.single_step(envs, actions, mode, self.seed, call, self.env_offset, inject_food)
        else:
            obs, reward, done, ec = _o.grid_step(envs, actions, mode, self.seed, call, self.env_offset, inject_food)
            sc = None
        out = dict(obs=obs, reward=reward, done=done, self_collision=sc, edge_collision=ec, obs_after=None)
        if post_reset or want_obs_after:
            target = envs if post_reset else envs.copy()
            oa = reset(target, done, mode, call + 1, inject_reset)
            out['obs_after'] = oa if want_obs_after else None
        return out

    def single_rollout(self, envs, actions, mode, inject_food=None, inject_reset=None):
        return _o.single_rollout(envs, actions, mode, self.seed, self._next(2 * actions.shape[0]), self.env_offset,
                                 inject_food, inject_reset)

    def single_policy_rollout(self, envs, obs0, params, T, obs_n):
        return _o.single_policy_rollout(envs, obs0, params, T, obs_n, self.seed, self._next(2 * T), self.env_offset)

    def single_check(self, envs):
        return _o.single_check(envs)

    # SimpleGridworld
    def grid_step(self, envs, actions, mode, inject_food=None):
        return _o.grid_step(envs, actions, mode, self.seed, self._next(), self.env_offset, inject_food)

    def grid_reset(self, envs, done, start, mode, inject_reset=None):
        return _o.grid_reset(envs, done, start, mode, self.seed, self._next(), self.env_offset, inject_reset)

    def grid_observe(self, envs, mode):
        return _o.grid_observe(envs, mode)

    def grid_rollout(self, envs, actions, start, mode, inject_food=None, inject_reset=None):
        return _o.grid_rollout(envs, actions, start, mode, self.seed, self._next(2 * actions.shape[0]),
                               self.env_offset, inject_food, inject_reset)

    # MultiSnake
    def multi_cfg(self, K, cfg):
        return _o.multi_cfg(K, boost=cfg['boost'], food_on_death_prob=cfg['food_on_death_prob'],
                            boost_cost_prob=cfg['boost_cost_prob'], food_mode=cfg['food_mode'],
                            food_rate=cfg['food_rate'], reward_on_death=cfg['reward_on_death'],
                            respawn_mode=cfg['respawn_mode'], colour_mode=cfg['colour_mode'])

    def multi_step(self, st, actions, cfg, mode, inject=None):
        K = st['heads'].shape[0] // st['foods'].shape[0]
        return _o.multi_step(st, actions, self.multi_cfg(K, cfg), mode, self.seed, self._next(), self.env_offset,
                             inject)

    def multi_reset(self, st, done_env, cfg, inject=None, mode=None):
        K = st['heads'].shape[0] // st['foods'].shape[0]
        rc = _o.multi_reset(st, done_env, self.multi_cfg(K, cfg), self.seed, self._next(), self.env_offset, inject)
        self.last_reset_obs = _o.multi_observe(st, mode) if mode else None
        return rc

    def multi_colours(self, N, K, fixed=False, call=0):
        return _o.multi_colours(N, K, fixed, self.seed, call, self.env_offset)

    def orientations(self, envs):
        return _o.orientations(envs)

    def multi_observe(self, st, mode):
        return _o.multi_observe(st, mode)

    def multi_check(self, st):
        return _o.multi_check(st)

    def multi_rollout(self, st, actions, cfg, mode, inject=None, reset_inject=None):
        """Loop of multi_step + multi_reset(all_done) — the definition the fused GPU rollout must reproduce.
        actions (T,K,N); inject / reset_inject: dicts of arrays with a leading T dimension."""
        import numpy as np
        T = actions.shape[0]
        outs = []
        for t in range(T):
            inj = {k: v[t] for k, v in inject.items()} if inject is not None else None
            r = self.multi_step(st, actions[t], cfg, mode, inject=inj)
            r['dones'] = st['dones'].copy()
            r['boost'] = st['boost_this_step'].copy()
            outs.append(r)
            rinj = {k: v[t] for k, v in reset_inject.items()} if reset_inject is not None else None
            self.multi_reset(st, r['all_done'], cfg, inject=rinj)
        return {k: np.stack([o[k] for o in outs]) for k in outs[0] if outs[0][k] is not None}
