"""SimpleGridworld — drop-in for the reference's wurm.envs.SimpleGridworld (wurm/envs/simple_gridworld.py:15-271)
on the gfx950 kernels of include/wurm_hip.h.  State `envs` is (num_envs, 2, size, size) fp32 = [food, agent].
Deviations are the ones listed in wurm_amd/envs/single_snake.py.  Per-call path: one launch per `step(a); reset(done)`
iteration, see wurm_amd/envs/_fast_step.py."""
from collections import namedtuple
from typing import Tuple

import ctypes

import torch

from wurm_amd import _lib
from wurm_amd.constants import DEFAULT_DEVICE
from wurm_amd.envs._fast_step import FastStepMixin
from wurm_amd.envs.single_snake import _draw_seed

Spec = namedtuple('Spec', ['reward_threshold'])


class SimpleGridworld(FastStepMixin):
    """Batched gridworld: the agent moves in the 4 cardinal directions, +1 reward on a food square (which then
    respawns), moving on to the border ring ends the episode (reference simple_gridworld.py:16-42)."""

    _CHANNELS = 2
    _STEP_SLOT = 'wurm_grid_step_slot'
    _RESIDENT_FNS = ('wurm_grid_resident_bytes', 'wurm_grid_resident_size', 'wurm_grid_resident_flush')

    spec = Spec(float('inf'))

    def __init__(self,
                 num_envs: int,
                 size: int,
                 on_death: str = 'restart',
                 observation_mode: str = 'default',
                 device: str = DEFAULT_DEVICE,
                 start_location: Tuple[int, int] = None,
                 manual_setup: bool = False,
                 verbose: int = 0,
                 seed: int = None,
                 env_offset: int = 0,
                 lazy_reset: bool = True,
                 resident_mirror=None):
        # (`resident_mirror`: as SingleSnake's — large batches step on a mirror of one 32-bit record per env,
        # wurm_grid_resident_bytes, include/wurm_hip.h; `env.mirror_state()` tells what is in effect)
        self._resident_policy = resident_mirror
        self.num_envs = num_envs
        self.size = size
        self.on_death = on_death
        self.observation_mode = observation_mode
        self.start_location = start_location
        self.device = _lib.require_device(device)
        self.verbose = verbose
        self.seed = _draw_seed() if seed is None else int(seed)
        self.env_offset = int(env_offset)
        self.lazy_reset = bool(lazy_reset)
        self._mode_cache = {}
        self._fast_init()

        self.t = 0

        if not manual_setup:
            self._reset(torch.ones(num_envs, dtype=torch.bool, device=self.device), observe=False)

        self.viewer = None

        self.head_colour = torch.tensor((0, 255, 0), dtype=torch.short, device=self.device)
        self.food_colour = torch.tensor((255, 0, 0), dtype=torch.short, device=self.device)
        self.edge_colour = torch.tensor((0, 0, 0), dtype=torch.short, device=self.device)

    @property
    def start_location(self):
        return self._start_location

    @start_location.setter
    def start_location(self, value):
        """reference :254-262 reads it at reset time: a reset(done) that was postponed is applied with the start location
        of the moment it was called (the old one), and the observation the last step's launch pre-computed for
        `reset(done)` assumed the old one as well"""
        fs = getattr(self, '_fs', None)
        if fs is not None:
            if fs.pending:
                self._flush()
            fs.obs_after = None
            fs.ok = False       # the next step goes through _slow_step, which hands the call block the new location
        self._start_location = value
        if fs is not None:
            fs.lazy_ok = self._lazy_reset and self._lazy_supported()

    def _lazy_supported(self) -> bool:
        return self.size > 4 and self.start_location is not None

    def _configure_call(self, c):
        sy, sx = self.start_location if self.start_location is not None else (-1, -1)
        c.start_y, c.start_x = int(sy), int(sx)

    def _launch_reset(self, envs, done, obs, m, n, call):
        rc = _lib.call(self.device.index, _lib.lib().wurm_grid_reset, _lib.ptr(envs), _lib.ptr(done), _lib.ptr(obs), m, n,
                       _lib.i64(self.num_envs), self.size, int(self.start_location[0]), int(self.start_location[1]),
                       _lib.u64(self.seed), _lib.u64(call), _lib.i64(self.env_offset), None,
                       _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'SimpleGridworld.reset')

    def _mode_info(self, observation_mode: str):
        info = self._mode_cache.get(observation_mode)
        if info is None:
            shape = self._obs_shape(observation_mode)
            m, n = _lib.parse_obs_mode(observation_mode)
            info = self._mode_cache[observation_mode] = (m, n, shape)
        return info

    def _obs_shape(self, mode: str):
        N, S = self.num_envs, self.size
        if mode == 'default':
            return (N, 3, S, S)
        if mode == 'raw':
            return (N, 2, S, S)
        if mode == 'positions':
            return (N, 4)
        raise Exception  # reference :132-133

    def _observe(self, observation_mode: str = 'default') -> torch.Tensor:
        """reference :111-133 ('positions' is generalised from num_envs == 1 to (N, 4))"""
        shape = self._obs_shape(observation_mode)
        m, n = _lib.parse_obs_mode(observation_mode)
        obs = torch.empty(shape, dtype=torch.float32, device=self.device)
        rc = _lib.call(self.device.index, _lib.lib().wurm_grid_observe, _lib.ptr(self._state()), _lib.ptr(obs), m, n, _lib.i64(self.num_envs),
                                          self.size, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'SimpleGridworld._observe')
        return obs

    def _get_rgb(self) -> torch.Tensor:
        """reference :88-109"""
        return (self._observe('default') * 255).round().short()

    def step(self, actions: torch.Tensor) -> (torch.Tensor, torch.Tensor, torch.Tensor, dict):
        """reference :135-202 (actions are not modified).  One launch; a reset(done) postponed by the previous iteration
        (wurm_amd/envs/_fast_step.py) is applied in front of the transition."""
        return self._fast_step(actions, 'SimpleGridworld.step')

    def _make_out(self, i: int):
        return self._v_obs[i], self._v_reward[i], self._v_done2[i], {'edge_collision': self._v_edgec[i]}

    def _reset(self, done: torch.Tensor, observe: bool = True):
        if self.size <= 4 or self.start_location is None:
            # reference :249-260 raises only when an env actually has to be created
            if bool(done.any()):
                if self.size <= 4:
                    raise NotImplementedError('Environemnts smaller than this don\'t make sense.')
                raise NotImplementedError("Haven't implemented random starting locations")
            return self._observe(self.observation_mode) if observe else None
        envs = self._state()
        if observe:
            m, n = _lib.parse_obs_mode(self.observation_mode)
            obs = torch.empty(self._obs_shape(self.observation_mode), dtype=torch.float32, device=self.device)
        else:
            m, n, obs = _lib.OBS_NONE, 0, None
        self._launch_reset(envs, done, obs, m, n, self._next_call())
        return obs

    def reset(self, done: torch.Tensor = None, return_observations: bool = True):
        """reference :225-245"""
        if done is None:
            done = self.done
        handled, obs = self._try_lazy_reset(done, return_observations)
        if handled:
            return obs
        done = done.view((done.shape[0]))
        if done.dtype != torch.bool:
            done = done != 0
        if done.device != self.device:
            done = done.to(self.device)
        return self._reset(done.contiguous(), observe=return_observations)

    def rollout(self, actions: torch.Tensor, return_observations: bool = True) -> dict:
        """T iterations of `step(actions[t]); reset(done)` in one launch (see SingleSnake.rollout)."""
        if actions.dtype not in (torch.int, torch.long):
            raise TypeError('actions Tensor must be an integer type i.e. {torch.IntTensor, torch.LongTensor}')
        if actions.dim() != 2 or actions.shape[1] != self.num_envs:
            raise RuntimeError('Must have the same number of actions as environments.')
        if not actions.is_contiguous() or actions.device != self.device:
            raise RuntimeError('rollout actions must be a contiguous device tensor')
        if self.start_location is None:
            raise NotImplementedError("Haven't implemented random starting locations")
        T, N = actions.shape
        if return_observations:
            m, n = _lib.parse_obs_mode(self.observation_mode)
            obs = torch.empty((T,) + self._obs_shape(self.observation_mode), dtype=torch.float32, device=self.device)
        else:
            m, n, obs = _lib.OBS_NONE, 0, None
        reward = torch.empty((T, N), dtype=torch.float32, device=self.device)
        flags = torch.empty((2, T, N), dtype=torch.bool, device=self.device)
        dt = _lib.ACT_I64 if actions.dtype == torch.long else _lib.ACT_I32
        # The mirror (round 6; large batches: wurm_grid_resident_bytes): the launch reads the records instead of scanning the
        # planes when they describe the state, keeps them current, and — lazy — does not write the planes; no flag pass
        # behind it (wurm_grid_rollout_resident).  Same protocol as step(): a postponed reset is applied first, a watched
        # tensor is checked for in-place edits, nothing is "touched".
        if self._fs.pending:
            self._flush()
        self._mirror_sync()
        self._setup_mirror(*_lib.parse_obs_mode(self.observation_mode))
        c = self._c
        if c.resident and T > 0:
            envs = self._checked(self._envs)
            self._fs.last_fresh = False
            valid = ctypes.c_int(c.resident_valid)
            rc = _lib.call(self.device.index, _lib.lib().wurm_grid_rollout_resident,
                _lib.ptr(envs), _lib.ptr(actions), dt, _lib.ptr(reward), _lib.ptr(flags[0]), _lib.ptr(flags[1]), _lib.ptr(obs),
                m, n, _lib.i64(N), self.size, _lib.i64(T), int(self.start_location[0]), int(self.start_location[1]),
                _lib.u64(self.seed), _lib.u64(self._next_call(2 * T)), _lib.i64(self.env_offset), c.resident,
                ctypes.addressof(valid), int(c.resident_lazy), _lib.stream_ptr(self.device.index))
            c.resident_valid = valid.value if rc == _lib.OK else 0
        else:
            envs = self._state()
            rc = _lib.call(self.device.index, _lib.lib().wurm_grid_rollout,
                _lib.ptr(envs), _lib.ptr(actions), dt,
                _lib.ptr(reward), _lib.ptr(flags[0]), _lib.ptr(flags[1]), _lib.ptr(obs), m, n, _lib.i64(N), self.size,
                _lib.i64(T), int(self.start_location[0]), int(self.start_location[1]), _lib.u64(self.seed),
                _lib.u64(self._next_call(2 * T)), _lib.i64(self.env_offset), None, None, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'SimpleGridworld.rollout')
        self._done_all_false()
        return {'observations': obs, 'rewards': reward, 'dones': flags[0], 'edge_collision': flags[1]}

    def _consistent(self):
        pass
