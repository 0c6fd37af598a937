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
