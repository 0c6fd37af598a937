"""HipBackend — numpy-in / numpy-out adapter over the product's C ABI (include/wurm_hip.h -> libwurm_hip.so)
with the same interface as tests/backends.OracleBackend, so that the same replay / comparison code drives
both.  Every call goes through ctypes into the shared library with raw device pointers: this is the drop-in
boundary under test.  GPU only.
"""
import numpy as np
import torch

from wurm_amd import _lib
from oracle import oracle as _o  # only for the obs-shape helpers (pure Python arithmetic)


class HipBackend(object):
    name = 'hip'

    def __init__(self, seed=0, env_offset=0, device='cuda:0'):
        self.seed = seed
        self.env_offset = env_offset
        self.call = 0
        self.dev = torch.device(device)
        self.lib = _lib.lib()

    def _next(self, n=1):
        c = self.call
        self.call += n
        return c

    def _t(self, a, dtype=None):
        if a is None:
            return None
        t = torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.dev)

    def _empty(self, shape, dtype):
        # poison outputs so that unwritten elements are caught
        if dtype == torch.float32:
            return torch.full(shape, float('nan'), dtype=dtype, device=self.dev)
        return torch.full(shape, 77, dtype=dtype, device=self.dev)

    @staticmethod
    def _act(a):
        return _lib.ACT_I64 if a.dtype == torch.int64 else _lib.ACT_I32

    def _stream(self):
        return _lib.stream_ptr()

    # ------------------------------------------------------------------ SingleSnake
    def single_step(self, envs, actions, mode, inject_food=None):
        N, _, S, _ = envs.shape
        m, n = _lib.parse_obs_mode(mode)
        e, a = self._t(envs), self._t(actions)
        shape = _o.single_obs_shape(mode, N, S)
        obs = self._empty(shape, torch.float32) if shape else None
        reward = self._empty((N,), torch.float32)
        done, sc, ec = (self._empty((N,), torch.uint8) for _ in range(3))
        inj = self._t(inject_food, torch.int32)
        rc = self.lib.wurm_single_step(_lib.ptr(e), _lib.ptr(a), self._act(a), _lib.ptr(reward), _lib.ptr(done),
                                       _lib.ptr(sc), _lib.ptr(ec), _lib.ptr(obs), m, n, _lib.i64(N), S,
                                       _lib.u64(self.seed), _lib.u64(self._next()), _lib.i64(self.env_offset),
                                       _lib.ptr(inj), self._stream())
        _lib.check(rc, 'wurm_single_step')
        torch.cuda.synchronize()
        envs[...] = e.cpu().numpy()
        actions[...] = a.cpu().numpy()
        return (obs.cpu().numpy() if obs is not None else None, reward.cpu().numpy(), done.cpu().numpy(),
                sc.cpu().numpy(), ec.cpu().numpy())

    def single_reset(self, envs, done, mode, inject_reset=None):
        N, _, S, _ = envs.shape
        m, n = _lib.parse_obs_mode(mode)
        e = self._t(envs)
        d = self._t((np.asarray(done).reshape(N) != 0).astype(np.uint8))
        shape = _o.single_obs_shape(mode, N, S)
        obs = self._empty(shape, torch.float32) if shape else None
        inj = self._t(inject_reset, torch.int32)
        rc = self.lib.wurm_single_reset(_lib.ptr(e), _lib.ptr(d), _lib.ptr(obs), m, n, _lib.i64(N), S,
                                        _lib.u64(self.seed), _lib.u64(self._next()), _lib.i64(self.env_offset),
                                        _lib.ptr(inj), self._stream())
        _lib.check(rc, 'wurm_single_reset')
        torch.cuda.synchronize()
        envs[...] = e.cpu().numpy()
        return obs.cpu().numpy() if obs is not None else None

    def single_step_reset(self, envs, actions, mode, call, pre_done=None, pre_call=None, post_reset=False,
                          want_obs_after=False, inject_food=None, inject_reset=None, inject_pre_reset=None, grid=None,
                          resident=None):
        """wurm_single_step_reset / wurm_grid_step_reset through the wurm_single_call argument block.
        resident: a dict the caller keeps across calls ({'valid': 0 | 1}; the mirror buffer is created in it): the call is
        given wurm_single_call.resident / resident_valid, and 'valid' is set as the protocol of the header says.  With
        resident['lazy'] the device copy of the state is kept in the dict across calls (the step does not write it) and is
        written out (wurm_single_resident_flush) and copied back into `envs` only when resident['sync'] is true"""
        N, _, S, _ = envs.shape
        m, n = _lib.parse_obs_mode(mode)
        lazy = resident is not None and bool(resident.get('lazy'))
        if lazy and resident.get('valid') == 1 and resident.get('envs_dev') is not None:
            e = resident['envs_dev']  # stale by design; the mirror describes the state
        else:
            e = self._t(envs)
        a = self._t(actions)
        shape = (_o.single_obs_shape if grid is None else _o.grid_obs_shape)(mode, N, S)
        obs = self._empty(shape, torch.float32) if shape else None
        obs_after = self._empty(shape, torch.float32) if (shape and want_obs_after) else None
        reward = self._empty((N,), torch.float32)
        done, sc, ec, copy = (self._empty((N,), torch.uint8) for _ in range(4))
        pd = self._t((np.asarray(pre_done).reshape(N) != 0).astype(np.uint8)) if pre_done is not None else None
        inj_f, inj_r, inj_p = (self._t(x, torch.int32) for x in (inject_food, inject_reset, inject_pre_reset))
        c = _lib.SingleCall()
        c.envs, c.actions, c.reward, c.done = _lib.ptr(e), _lib.ptr(a), _lib.ptr(reward), _lib.ptr(done)
        c.self_collision, c.edge_collision, c.obs, c.obs_after = _lib.ptr(sc), _lib.ptr(ec), _lib.ptr(obs), _lib.ptr(obs_after)
        c.done_copy, c.pre_done = _lib.ptr(copy), _lib.ptr(pd)
        c.inject_food, c.inject_reset, c.inject_pre_reset = _lib.ptr(inj_f), _lib.ptr(inj_r), _lib.ptr(inj_p)
        c.num_envs, c.env_offset, c.seed, c.call = N, self.env_offset, _lib.u64(self.seed), _lib.u64(call)
        c.pre_call = _lib.u64(pre_call if pre_call is not None else 0)
        c.actions_dtype, c.obs_mode, c.obs_n, c.size, c.post_reset = self._act(a), m, n, S, int(bool(post_reset))
        c.start_y, c.start_x = (-1, -1) if grid is None else grid
        if resident is not None:
            nbytes = 32 * N if S == 9 else 48 * N if S in (10, 11) else N * ((((S * S + 255) >> 8) * 512) + 48)  # (lane_resident.hpp / lane_wide_resident.hpp / grid_rollout.hip)
            if grid is not None:
                nbytes = 16 + 4 * N                                                 # (gridworld_lane.hip: header + records)
            if resident.get('buf') is None or resident['buf'].numel() != nbytes:
                resident['buf'], resident['valid'] = self._empty((nbytes,), torch.uint8), 0
            c.resident, c.resident_valid = _lib.ptr(resident['buf']), int(resident.get('valid', 0))
            c.resident_lazy = int(lazy)
        import ctypes
        fn = self.lib.wurm_single_step_reset if grid is None else self.lib.wurm_grid_step_reset
        rc = fn(ctypes.addressof(c), self._stream())
        refused = grid is not None and resident is not None and rc == 1   # WURM_MIRROR_REFUSED: the call ran
        _lib.check(0 if refused else rc, 'wurm_single_step_reset')
        if resident is not None:
            served = int(self.lib.wurm_single_resident_size(_lib.i64(N), S, m, n)) > 0 if grid is None else \
                (int(self.lib.wurm_grid_resident_size(_lib.i64(N), S, m)) > 0 and
                 N >= int(self.lib.wurm_get_option(b'WURM_LANE_STEP_MIN_ENVS')))
            if grid is None and S in (10, 11) and not lazy:
                served = False   # (include/wurm_hip.h: the mirror of 10 x 10 / 11 x 11 is maintained by lazy calls only)
            was = int(resident.get('valid', 0))
            resident['valid'] = int(inject_food is None and inject_reset is None and inject_pre_reset is None and
                                    not post_reset and served)
            if grid is not None and resident['valid'] and (refused or was == 2):
                resident['valid'] = 2          # (stays refused until the caller clears it: include/wurm_hip.h)
            flush = self.lib.wurm_single_resident_flush if grid is None else self.lib.wurm_grid_resident_flush
            if lazy:
                resident['envs_dev'] = e
                c.resident_valid = resident['valid']
                if resident.get('sync', True):
                    _lib.check(flush(ctypes.addressof(c), self._stream()), 'flush')
        torch.cuda.synchronize()
        if not lazy or resident.get('sync', True) or resident['valid'] != 1:  # (not valid: the call wrote envs itself)
            envs[...] = e.cpu().numpy()
        actions[...] = a.cpu().numpy()
        assert torch.equal(copy, done), 'done_copy != done'
        return dict(obs=obs.cpu().numpy() if obs is not None else None, reward=reward.cpu().numpy(),
                    done=done.cpu().numpy(), self_collision=sc.cpu().numpy() if grid is None else None,
                    edge_collision=ec.cpu().numpy(),
                    obs_after=obs_after.cpu().numpy() if obs_after is not None else None)

    def single_observe(self, envs, mode):
        N, _, S, _ = envs.shape
        m, n = _lib.parse_obs_mode(mode)
        e = self._t(envs)
        obs = self._empty(_o.single_obs_shape(mode, N, S), torch.float32)
        rc = self.lib.wurm_single_observe(_lib.ptr(e), _lib.ptr(obs), m, n, _lib.i64(N), S, self._stream())
        _lib.check(rc, 'wurm_single_observe')
        return obs.cpu().numpy()

    def single_rollout(self, envs, actions, mode, inject_food=None, inject_reset=None):
        N, _, S, _ = envs.shape
        T = actions.shape[0]
        m, n = _lib.parse_obs_mode(mode)
        e, a = self._t(envs), self._t(actions)
        shape = _o.single_obs_shape(mode, N, S)
        obs = self._empty((T,) + shape, torch.float32) if shape else None
        reward = self._empty((T, N), torch.float32)
        done, sc, ec = (self._empty((T, N), torch.uint8) for _ in range(3))
        inj_f, inj_r = self._t(inject_food, torch.int32), self._t(inject_reset, torch.int32)
        rc = self.lib.wurm_single_rollout(_lib.ptr(e), _lib.ptr(a), self._act(a), _lib.ptr(reward), _lib.ptr(done),
                                          _lib.ptr(sc), _lib.ptr(ec), _lib.ptr(obs), m, n, _lib.i64(N), S,
                                          _lib.i64(T), _lib.u64(self.seed), _lib.u64(self._next(2 * T)),
                                          _lib.i64(self.env_offset), _lib.ptr(inj_f), _lib.ptr(inj_r),
                                          self._stream())
        _lib.check(rc, 'wurm_single_rollout')
        torch.cuda.synchronize()
        envs[...] = e.cpu().numpy()
        actions[...] = a.cpu().numpy()
        return dict(obs=obs.cpu().numpy() if obs is not None else None, reward=reward.cpu().numpy(),
                    done=done.cpu().numpy(), self_collision=sc.cpu().numpy(), edge_collision=ec.cpu().numpy())

    def single_policy_rollout(self, envs, obs0, params, T, obs_n):
        N, _, S, _ = envs.shape
        E = 3 * (2 * obs_n + 1) ** 2
        e, x0, w = self._t(envs), self._t(np.asarray(obs0, np.float32).reshape(N, E)), self._t(np.asarray(params, np.float32))
        actions = self._empty((T, N), torch.int64)
        probs = self._empty((T, N, 4), torch.float32)
        values, reward = self._empty((T, N), torch.float32), self._empty((T, N), torch.float32)
        done, sc, ec = (self._empty((T, N), torch.uint8) for _ in range(3))
        obs = self._empty((T, N, E), torch.float32)
        status = self._empty((N,), torch.uint8)
        rc = self.lib.wurm_single_policy_rollout(_lib.ptr(e), _lib.ptr(x0), _lib.ptr(w), _lib.ptr(actions), _lib.ptr(probs),
                                                 _lib.ptr(values), _lib.ptr(reward), _lib.ptr(done), _lib.ptr(sc),
                                                 _lib.ptr(ec), _lib.ptr(obs), _lib.ptr(status), int(obs_n), _lib.i64(N), S,
                                                 _lib.i64(T), _lib.u64(self.seed), _lib.u64(self._next(2 * T)),
                                                 _lib.i64(self.env_offset), self._stream())
        _lib.check(rc, 'wurm_single_policy_rollout')
        torch.cuda.synchronize()
        envs[...] = e.cpu().numpy()
        return dict(actions=actions.cpu().numpy(), probs=probs.cpu().numpy(), values=values.cpu().numpy(),
                    reward=reward.cpu().numpy(), done=done.cpu().numpy(), self_collision=sc.cpu().numpy(),
                    edge_collision=ec.cpu().numpy(), obs=obs.cpu().numpy(), status=status.cpu().numpy())

    def single_check(self, envs):
        N, _, S, _ = envs.shape
        e = self._t(envs)
        err = self._empty((N,), torch.int32)
        rc = self.lib.wurm_single_check(_lib.ptr(e), _lib.ptr(err), _lib.i64(N), S, self._stream())
        _lib.check(rc, 'wurm_single_check')
        return err.cpu().numpy().astype(np.uint32)

    # ------------------------------------------------------------------ SimpleGridworld
    def grid_step(self, envs, actions, mode, inject_food=None):
        N, _, S, _ = envs.shape
        m, n = _lib.parse_obs_mode(mode)
        e, a = self._t(envs), self._t(actions)
        shape = _o.grid_obs_shape(mode, N, S)
        obs = self._empty(shape, torch.float32) if shape else None
        reward = self._empty((N,), torch.float32)
        done, ec = (self._empty((N,), torch.uint8) for _ in range(2))
        inj = self._t(inject_food, torch.int32)
        rc = self.lib.wurm_grid_step(_lib.ptr(e), _lib.ptr(a), self._act(a), _lib.ptr(reward), _lib.ptr(done),
                                     _lib.ptr(ec), _lib.ptr(obs), m, n, _lib.i64(N), S, _lib.u64(self.seed),
                                     _lib.u64(self._next()), _lib.i64(self.env_offset), _lib.ptr(inj),
                                     self._stream())
        _lib.check(rc, 'wurm_grid_step')
        torch.cuda.synchronize()
        envs[...] = e.cpu().numpy()
        actions[...] = a.cpu().numpy()
        return (obs.cpu().numpy() if obs is not None else None, reward.cpu().numpy(), done.cpu().numpy(),
                ec.cpu().numpy())

    def grid_reset(self, envs, done, start, mode, inject_reset=None):
        N, _, S, _ = envs.shape
        m, n = _lib.parse_obs_mode(mode)
        e = self._t(envs)
        d = self._t((np.asarray(done).reshape(N) != 0).astype(np.uint8))
        shape = _o.grid_obs_shape(mode, N, S)
        obs = self._empty(shape, torch.float32) if shape else None
        inj = self._t(inject_reset, torch.int32)
        sy, sx = (-1, -1) if start is None else start
        rc = self.lib.wurm_grid_reset(_lib.ptr(e), _lib.ptr(d), _lib.ptr(obs), m, n, _lib.i64(N), S, int(sy),
                                      int(sx), _lib.u64(self.seed), _lib.u64(self._next()),
                                      _lib.i64(self.env_offset), _lib.ptr(inj), self._stream())
        _lib.check(rc, 'wurm_grid_reset')
        torch.cuda.synchronize()
        envs[...] = e.cpu().numpy()
        return obs.cpu().numpy() if obs is not None else None

    def grid_observe(self, envs, mode):
        N, _, S, _ = envs.shape
        m, n = _lib.parse_obs_mode(mode)
        e = self._t(envs)
        obs = self._empty(_o.grid_obs_shape(mode, N, S), torch.float32)
        rc = self.lib.wurm_grid_observe(_lib.ptr(e), _lib.ptr(obs), m, n, _lib.i64(N), S, self._stream())
        _lib.check(rc, 'wurm_grid_observe')
        return obs.cpu().numpy()

    def grid_rollout(self, envs, actions, start, mode, inject_food=None, inject_reset=None):
        N, _, S, _ = envs.shape
        T = actions.shape[0]
        m, n = _lib.parse_obs_mode(mode)
        e, a = self._t(envs), self._t(actions)
        shape = _o.grid_obs_shape(mode, N, S)
        obs = self._empty((T,) + shape, torch.float32) if shape else None
        reward = self._empty((T, N), torch.float32)
        done, ec = (self._empty((T, N), torch.uint8) for _ in range(2))
        inj_f, inj_r = self._t(inject_food, torch.int32), self._t(inject_reset, torch.int32)
        rc = self.lib.wurm_grid_rollout(_lib.ptr(e), _lib.ptr(a), self._act(a), _lib.ptr(reward), _lib.ptr(done),
                                        _lib.ptr(ec), _lib.ptr(obs), m, n, _lib.i64(N), S, _lib.i64(T),
                                        int(start[0]), int(start[1]), _lib.u64(self.seed),
                                        _lib.u64(self._next(2 * T)), _lib.i64(self.env_offset), _lib.ptr(inj_f),
                                        _lib.ptr(inj_r), self._stream())
        _lib.check(rc, 'wurm_grid_rollout')
        torch.cuda.synchronize()
        envs[...] = e.cpu().numpy()
        actions[...] = a.cpu().numpy()
        return dict(obs=obs.cpu().numpy() if obs is not None else None, reward=reward.cpu().numpy(),
                    done=done.cpu().numpy(), edge_collision=ec.cpu().numpy())


# ------------------------------------------------------------------ MultiSnake (methods added to HipBackend below)

def _multi_cfg(K, cfg):
    return _lib.multi_config(K, cfg['boost'], cfg['food_on_death_prob'], cfg['boost_cost_prob'], cfg['food_mode'],
                             cfg['food_rate'], cfg['reward_on_death'], cfg['respawn_mode'], cfg['colour_mode'])


def _multi_to_dev(self, st):
    return {k: self._t(v) for k, v in st.items()}


def _multi_back(st, dev):
    for k in st:
        st[k][...] = dev[k].cpu().numpy()


def multi_step(self, st, actions, cfg, mode, inject=None):
    import ctypes
    N, _, S, _ = st['foods'].shape
    K = st['heads'].shape[0] // N
    m, n = _lib.parse_obs_mode(mode)
    d = _multi_to_dev(self, st)
    act = self._t(np.ascontiguousarray(actions, np.int64))
    shape = _o.multi_obs_shape(mode, N, K, S)
    obs = self._empty(shape, torch.float32) if shape else None
    rewards, food, size = (self._empty((N * K,), torch.float32) for _ in range(3))
    sc, ec = (self._empty((N * K,), torch.uint8) for _ in range(2))
    all_done = self._empty((N,), torch.uint8)
    am_f, am_b = self._empty((3, K, N), torch.float32), self._empty((4, K, N), torch.uint8)
    c = _multi_cfg(K, cfg)
    inj_ref, keep = None, []
    if inject is not None:
        keep = [self._t(np.ascontiguousarray(inject[k], np.uint8)) for k in ('death_a', 'cost', 'death_b', 'rate')]
        keep.append(self._t(np.ascontiguousarray(inject['food_cell'], np.int32)))
        inj = _lib.MultiInject(*[t.data_ptr() for t in keep])
        inj_ref = ctypes.byref(inj)
    rc = self.lib.wurm_multi_step(
        _lib.ptr(d['foods']), _lib.ptr(d['heads']), _lib.ptr(d['bodies']), _lib.ptr(d['dones']),
        _lib.ptr(d['orientations']), _lib.ptr(act), _lib.ptr(d['boost_this_step']), _lib.ptr(rewards), _lib.ptr(sc),
        _lib.ptr(ec), _lib.ptr(food), _lib.ptr(size), _lib.ptr(all_done), _lib.ptr(d['colours']), _lib.ptr(obs), m, n,
        _lib.i64(N), K, S, ctypes.byref(c), _lib.u64(self.seed), _lib.u64(self._next()), _lib.i64(self.env_offset),
        inj_ref, _lib.ptr(am_f), _lib.ptr(am_b), self._stream())
    _lib.check(rc, 'wurm_multi_step')
    torch.cuda.synchronize()
    _multi_back(st, d)
    # the agent-major copies must be the transposes of the env-major outputs
    t = lambda x: x.view(N, K).t().contiguous().view(-1)
    for row, ref in zip(am_f, (rewards, food, size)):
        assert torch.equal(row.view(-1), t(ref)), 'agent-major f32 copy'
    for row, ref in zip(am_b, (d['dones'], d['boost_this_step'], sc, ec)):
        assert torch.equal(row.view(-1), t(ref)), 'agent-major u8 copy'
    return dict(obs=obs.cpu().numpy() if obs is not None else None, rewards=rewards.cpu().numpy(),
                snake_collision=sc.cpu().numpy(), edge_collision=ec.cpu().numpy(), food=food.cpu().numpy(),
                size=size.cpu().numpy(), all_done=all_done.cpu().numpy())


def multi_step_reset(self, st, actions, cfg, mode, call, pre_done=None, pre_call=0, want_obs_after=False, resident=None):
    """wurm_multi_step_reset through the wurm_multi_call block: [wurm_multi_reset(pre_done, pre_call),] step(call).
    resident: a dict the caller keeps across calls ({'valid': 0 | 1, 'lazy': bool, 'sync': bool}): the call is given
    wurm_multi_call.resident; with 'lazy' the device copy of the state lives in the dict and foods / heads / bodies are
    written out (wurm_multi_resident_flush) and copied back into `st` only when 'sync' is true"""
    import ctypes
    N, _, S, _ = st['foods'].shape
    K = st['heads'].shape[0] // N
    m, n = _lib.parse_obs_mode(mode)
    lazy = resident is not None and bool(resident.get('lazy'))
    if lazy and resident.get('valid') and resident.get('dev') is not None:
        d = resident['dev']  # foods / heads / bodies stale by design; the small arrays are always current
    else:
        d = _multi_to_dev(self, st)
    act = self._t(np.ascontiguousarray(actions, np.int64))
    shape = _o.multi_obs_shape(mode, N, K, S)
    obs = self._empty(shape, torch.float32) if shape else None
    obs_after = self._empty(shape, torch.float32) if (shape and want_obs_after) else None
    rewards, food, size = (self._empty((N * K,), torch.float32) for _ in range(3))
    sc, ec = (self._empty((N * K,), torch.uint8) for _ in range(2))
    all_done, copy = self._empty((N,), torch.uint8), self._empty((N,), torch.uint8)
    am_f, am_b = self._empty((3, K, N), torch.float32), self._empty((4, K, N), torch.uint8)
    pd = self._t((np.asarray(pre_done).reshape(N) != 0).astype(np.uint8)) if pre_done is not None else None
    c = _lib.MultiCall()
    for name in ('foods', 'heads', 'bodies', 'dones', 'orientations', 'colours', 'boost_this_step'):
        setattr(c, name, d[name].data_ptr())
    c.actions, c.rewards, c.snake_collision, c.edge_collision = act.data_ptr(), rewards.data_ptr(), sc.data_ptr(), ec.data_ptr()
    c.food_consumed, c.sizes, c.all_done, c.all_done_copy = food.data_ptr(), size.data_ptr(), all_done.data_ptr(), copy.data_ptr()
    c.obs, c.agent_major_f32, c.agent_major_u8 = _lib.ptr(obs), am_f.data_ptr(), am_b.data_ptr()
    c.obs_after = _lib.ptr(obs_after)
    c.pre_done = _lib.ptr(pd)
    c.num_envs, c.env_offset, c.seed, c.call, c.pre_call = N, self.env_offset, _lib.u64(self.seed), _lib.u64(call), _lib.u64(pre_call)
    c.num_snakes, c.size, c.obs_mode, c.obs_n = K, S, m, n
    c.cfg = _multi_cfg(K, cfg)
    if resident is not None:
        nbytes = int(self.lib.wurm_multi_resident_bytes(_lib.i64(N), K, S))
        assert nbytes > 0, 'set WURM_RESIDENT_MIN_ENVS=0 for small batches'
        if resident.get('buf') is None or resident['buf'].numel() != nbytes:
            resident['buf'], resident['valid'] = self._empty((nbytes,), torch.uint8), 0
        c.resident, c.resident_valid, c.resident_lazy = resident['buf'].data_ptr(), int(resident.get('valid', 0)), int(lazy)
        masks = self._empty((2, N), torch.int32)
        masks.fill_(-7)
        c.check_mask = masks[0].data_ptr()
        c.check_mask_after = masks[1].data_ptr() if obs_after is not None else None
        resident['masks'] = masks
    rc = self.lib.wurm_multi_step_reset(ctypes.addressof(c), self._stream())
    _lib.check(rc, 'wurm_multi_step_reset')
    synced = True
    if resident is not None:
        resident['valid'] = 1
        if lazy:
            resident['dev'] = d
            c.resident_valid = 1
            synced = bool(resident.get('sync', True))
            if synced:
                _lib.check(self.lib.wurm_multi_resident_flush(ctypes.addressof(c), self._stream()), 'flush')
    torch.cuda.synchronize()
    if synced:
        _multi_back(st, d)
    else:  # the small state arrays are written by every launch
        for k in st:
            if k not in ('foods', 'heads', 'bodies'):
                st[k][...] = d[k].cpu().numpy()
    assert torch.equal(copy, all_done), 'all_done_copy != all_done'
    return dict(obs=obs.cpu().numpy() if obs is not None else None, rewards=rewards.cpu().numpy(),
                snake_collision=sc.cpu().numpy(), edge_collision=ec.cpu().numpy(), food=food.cpu().numpy(),
                size=size.cpu().numpy(), all_done=all_done.cpu().numpy(),
                **({'obs_after': obs_after.cpu().numpy()} if obs_after is not None else {}))


def multi_reset(self, st, done_env, cfg, inject=None, mode=None):
    import ctypes
    N, _, S, _ = st['foods'].shape
    K = st['heads'].shape[0] // N
    d = _multi_to_dev(self, st)
    de = self._t((np.asarray(done_env).reshape(N) != 0).astype(np.uint8))
    status = torch.zeros(1, dtype=torch.int32, device=self.dev)
    c = _multi_cfg(K, cfg)
    m, n = _lib.parse_obs_mode(mode)
    shape = _o.multi_obs_shape(mode, N, K, S) if mode else None
    obs = self._empty(shape, torch.float32) if shape else None
    inj_ref, keep = None, []
    if inject is not None:
        keep = [self._t(np.ascontiguousarray(inject['create'], np.int32)),
                self._t(np.ascontiguousarray(inject['create_food'], np.int32)),
                self._t(np.ascontiguousarray(inject['colours'], np.int16)),
                self._t(np.ascontiguousarray(inject['respawn'], np.int32))]
        inj = _lib.MultiResetInject(*[t.data_ptr() for t in keep])
        inj_ref = ctypes.byref(inj)
    rc = self.lib.wurm_multi_reset(
        _lib.ptr(d['foods']), _lib.ptr(d['heads']), _lib.ptr(d['bodies']), _lib.ptr(d['dones']),
        _lib.ptr(d['orientations']), _lib.ptr(d['colours']), _lib.ptr(de), _lib.ptr(status),
        _lib.ptr(d['boost_this_step']), _lib.ptr(obs), m, n, _lib.i64(N), K, S, ctypes.byref(c), _lib.u64(self.seed),
        _lib.u64(self._next()), _lib.i64(self.env_offset), inj_ref, self._stream())
    _lib.check(rc, 'wurm_multi_reset')
    torch.cuda.synchronize()
    _multi_back(st, d)
    self.last_reset_obs = obs.cpu().numpy() if obs is not None else None
    return int(status.item())


def multi_observe(self, st, mode):
    N, _, S, _ = st['foods'].shape
    K = st['heads'].shape[0] // N
    m, n = _lib.parse_obs_mode(mode)
    d = _multi_to_dev(self, st)
    obs = self._empty(_o.multi_obs_shape(mode, N, K, S), torch.float32)
    rc = self.lib.wurm_multi_observe(_lib.ptr(d['foods']), _lib.ptr(d['heads']), _lib.ptr(d['bodies']),
                                     _lib.ptr(d['dones']), _lib.ptr(d['boost_this_step']), _lib.ptr(d['colours']),
                                     _lib.ptr(obs), m, n, _lib.i64(N), K, S, self._stream())
    _lib.check(rc, 'wurm_multi_observe')
    return obs.cpu().numpy()


def multi_check(self, st):
    N, _, S, _ = st['foods'].shape
    K = st['heads'].shape[0] // N
    d = _multi_to_dev(self, st)
    err = self._empty((N,), torch.int32)
    rc = self.lib.wurm_multi_check(_lib.ptr(d['foods']), _lib.ptr(d['heads']), _lib.ptr(d['bodies']),
                                   _lib.ptr(d['dones']), _lib.ptr(err), _lib.i64(N), K, S, self._stream())
    _lib.check(rc, 'wurm_multi_check')
    return err.cpu().numpy().astype(np.uint32)


def multi_colours(self, N, K, fixed=False, call=0):
    col = self._empty((N * K, 3), torch.int16)
    rc = self.lib.wurm_multi_colours(_lib.ptr(col), _lib.i64(N), K, int(fixed), _lib.u64(self.seed), _lib.u64(call),
                                     _lib.i64(self.env_offset), self._stream())
    _lib.check(rc, 'wurm_multi_colours')
    return col.cpu().numpy()


def orientations(self, envs):
    n, _, S, _ = envs.shape
    e = self._t(envs)
    out = self._empty((n,), torch.int64)
    rc = self.lib.wurm_orientations(_lib.ptr(e), _lib.ptr(out), _lib.i64(n), S, self._stream())
    _lib.check(rc, 'wurm_orientations')
    return out.cpu().numpy()


def multi_rollout(self, st, actions, cfg, mode, inject=None, reset_inject=None):
    """wurm_multi_rollout; returns the same dict of (T, ...) env-major arrays as OracleBackend.multi_rollout."""
    import ctypes
    N, _, S, _ = st['foods'].shape
    K = st['heads'].shape[0] // N
    T = actions.shape[0]
    m, n = _lib.parse_obs_mode(mode)
    d = _multi_to_dev(self, st)
    act = self._t(np.ascontiguousarray(actions, np.int64))
    shape = _o.multi_obs_shape(mode, N, K, S)
    obs = self._empty((T,) + shape, torch.float32) if shape else None
    out_f, out_b = self._empty((T, 3, K, N), torch.float32), self._empty((T, 4, K, N), torch.uint8)
    all_done = self._empty((T, N), torch.uint8)
    c = _multi_cfg(K, cfg)
    inj_ref = rinj_ref = None
    keep = []
    if inject is not None:
        a = [self._t(np.ascontiguousarray(inject[k], np.uint8)) for k in ('death_a', 'cost', 'death_b', 'rate')]
        a.append(self._t(np.ascontiguousarray(inject['food_cell'], np.int32)))
        b = [self._t(np.ascontiguousarray(reset_inject['create'], np.int32)),
             self._t(np.ascontiguousarray(reset_inject['create_food'], np.int32)),
             self._t(np.ascontiguousarray(reset_inject['colours'], np.int16)),
             self._t(np.ascontiguousarray(reset_inject['respawn'], np.int32))]
        keep = a + b
        inj, rinj = _lib.MultiInject(*[t.data_ptr() for t in a]), _lib.MultiResetInject(*[t.data_ptr() for t in b])
        inj_ref, rinj_ref = ctypes.byref(inj), ctypes.byref(rinj)
    rc = self.lib.wurm_multi_rollout(
        _lib.ptr(d['foods']), _lib.ptr(d['heads']), _lib.ptr(d['bodies']), _lib.ptr(d['dones']),
        _lib.ptr(d['orientations']), _lib.ptr(d['colours']), _lib.ptr(d['boost_this_step']), _lib.ptr(act),
        _lib.ptr(out_f), _lib.ptr(out_b), _lib.ptr(all_done), _lib.ptr(obs), m, n, _lib.i64(N), K, S, _lib.i64(T),
        ctypes.byref(c), _lib.u64(self.seed), _lib.u64(self._next(2 * T)), _lib.i64(self.env_offset), inj_ref, rinj_ref,
        self._stream())
    _lib.check(rc, 'wurm_multi_rollout')
    torch.cuda.synchronize()
    _multi_back(st, d)
    em = lambda x: x.permute(0, 2, 1).reshape(T, N * K).cpu().numpy()  # (T,K,N) agent-major -> (T,N*K) env-major
    return dict(obs=obs.cpu().numpy() if obs is not None else None, rewards=em(out_f[:, 0]), food=em(out_f[:, 1]),
                size=em(out_f[:, 2]), dones=em(out_b[:, 0]), boost=em(out_b[:, 1]), snake_collision=em(out_b[:, 2]),
                edge_collision=em(out_b[:, 3]), all_done=all_done.cpu().numpy())


for _f in (multi_step, multi_step_reset, multi_reset, multi_observe, multi_check, multi_colours, orientations, multi_rollout):
    setattr(HipBackend, _f.__name__, _f)
