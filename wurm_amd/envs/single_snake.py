"""SingleSnake — drop-in for the reference's wurm.envs.SingleSnake (wurm/envs/single_snake.py:17-428) whose
step / reset / _observe run as fused gfx950 kernels behind the C ABI of include/wurm_hip.h.

Same constructor keywords, public attributes (`envs`, `done`, `num_envs`, `size`, ...), return conventions and
error types as the reference.  Intentional deviations:
  * `done` and `info[...]` are torch.bool (the faithful translation of torch-1.1 uint8 masks: `~done` is a
    logical not, as experiments/main.py:215 needs);
  * randomness comes from a counter-based Philox generator (`seed`, `env_offset` keywords) instead of torch's
    global RNG stream, so trajectories do not depend on how the batch is sharded over GPUs;
  * `partial_n` observations of an env whose head left the grid are zeros (the reference raises at :191);
  * there is no CPU path: `device` must be a HIP GPU.

Per-call cost (the loop of experiments/main.py:212-227 at 512 envs is bound by the host, not by the GPU):
  * `step` is ONE launch (wurm_single_step_reset) described by a persistent argument block; outputs are carved from
    slabs that hold the next few dozen steps' worth of fresh tensors (never reused: a slab is dropped once used up and
    lives as long as any tensor handed out from it);
  * `reset(done)` with the `done` that `step` just returned is DEFERRED (`lazy_reset=True`, the default): the next
    `step` rebuilds those envs in front of its own transition, in the same launch, with the RNG counters the eager
    call would have used — bit-identical results.  Anything that looks at the state in between (`env.envs`, `_observe`,
    `check_consistency`, `rollout`, `render`, another `reset`) first flushes the postponed reset with the ordinary
    reset kernel, so `env.envs` always is what the reference would show.  The one thing that is not tracked is a
    tensor alias taken BEFORE the deferred reset (`e = env.envs; env.step(a); env.reset(d); e[...]`): read the
    attribute again, or pass `lazy_reset=False`.
"""
import ctypes
from collections import namedtuple

import torch

from wurm_amd import _lib
from wurm_amd.constants import DEFAULT_DEVICE

Spec = namedtuple('Spec', ['reward_threshold'])

_INT_TYPES = (torch.short, torch.int, torch.long)


def _draw_seed() -> int:
    # one draw from torch's global generator: `torch.manual_seed(k)` before construction pins the trajectories
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class SingleSnake(object):
    """Batched snake environment: state `envs` is (num_envs, 3, size, size) fp32 = [food, head, body]
    (reference single_snake.py:22-47)."""

    spec = Spec(float('inf'))
    metadata = {
        'render.modes': ['rgb_array'],
        'video.frames_per_second': 12
    }

    def __init__(self,
                 num_envs: int,
                 size: int,
                 max_timesteps: int = None,
                 initial_snake_length: int = 3,
                 on_death: str = 'restart',
                 observation_mode: str = 'one_channel',
                 device: str = DEFAULT_DEVICE,
                 manual_setup: bool = False,
                 verbose: int = 0,
                 render_args: dict = None,
                 seed: int = None,
                 env_offset: int = 0,
                 lazy_reset: bool = True):
        self.num_envs = num_envs
        self.size = size
        self.max_timesteps = max_timesteps
        self.initial_snake_length = initial_snake_length
        self.on_death = on_death
        self.observation_mode = observation_mode
        self.device = _lib.require_device(device)
        self.verbose = verbose
        self.seed = _draw_seed() if seed is None else int(seed)
        self.env_offset = int(env_offset)
        self._call = 0
        self._mode_cache = {}
        self.lazy_reset = bool(lazy_reset)
        # deferred reset(done) of the last step (see the module docstring)
        self._pending = False          # a reset is postponed; its flags are in self._pend, its counter in _pend_call
        self._pend = None              # (N) bytes: the kernels' own copy of the last step's `done`
        self._pend_call = 0
        self._last_done2 = None        # the (N,1) `done` the last step returned
        self._done = None              # env.done: its (N) view (made on first use) or whatever the caller assigned
        self._slab_version = -1
        self._last_fresh = False       # no call since that step has consumed an RNG counter or changed the state
        self._obs_after = None         # what reset(done) will return, if the last step produced it
        self._want_obs_after = False   # adaptive: callers that read reset()'s observation get it from the step launch
        self._R = self._slot = 0       # output slabs (see _new_slab)
        self._slab_mode = None
        self._v_obs_after = None
        self._c = None

        if render_args is None:
            self.render_args = {'num_rows': 1, 'num_cols': 1, 'size': 256}
        else:
            self.render_args = render_args

        self._envs = torch.zeros((num_envs, 3, size, size), device=self.device)
        self._envs_ok = self._envs
        self.t = 0
        self._done = torch.zeros(num_envs, dtype=torch.bool, device=self.device)
        self._done_from_step = False

        if not manual_setup:
            # reference :90-93 _create_envs(num_envs): every env is built by the reset kernel
            self._reset(torch.ones(num_envs, dtype=torch.bool, device=self.device), observe=False)

        self.viewer = None

        self.body_colour = torch.tensor((0, 127, 0), dtype=torch.short, device=self.device)
        self.head_colour = torch.tensor((0, 255, 0), dtype=torch.short, device=self.device)
        self.food_colour = torch.tensor((255, 0, 0), dtype=torch.short, device=self.device)
        self.edge_colour = torch.tensor((0, 0, 0), dtype=torch.short, device=self.device)

    # ------------------------------------------------------------------ helpers

    def _next_call(self, n: int = 1) -> int:
        c = self._call
        self._call += n
        self._last_fresh = False  # the counter the last step's `obs_after` assumed for its reset is gone
        return c

    @property
    def envs(self) -> torch.Tensor:
        """(num_envs, 3, size, size) fp32 = [food, head, body]; caller-visible and caller-mutable as in the reference
        (tests/test_single_snake_env.py:54, experiments/main.py:215,273).  Reading it applies a postponed reset."""
        if self._pending:
            self._flush()
        return self._envs

    @envs.setter
    def envs(self, value: torch.Tensor):
        # the reference would have applied reset(done) to the OLD tensor before this assignment replaced it
        self._pending = False
        self._last_fresh = False
        self._envs = value

    def _flush(self):
        """Applies the postponed reset(done) now, with the ordinary reset kernel and the counter it was given."""
        self._pending = False
        rc = _lib.call(self.device.index, _lib.lib().wurm_single_reset,
                       _lib.ptr(self._checked(self._envs)), _lib.ptr(self._pend), None, _lib.OBS_NONE, 0,
                       _lib.i64(self.num_envs), self.size, _lib.u64(self.seed), _lib.u64(self._pend_call),
                       _lib.i64(self.env_offset), None, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'SingleSnake.reset')

    def _state(self) -> torch.Tensor:
        """The state tensor, validated, with any postponed reset applied; the caller is about to change it or to
        consume an RNG counter."""
        if self._pending:
            self._flush()
        self._last_fresh = False
        return self._checked(self._envs)

    def _checked(self, e: torch.Tensor) -> torch.Tensor:
        if e is self._envs_ok:
            return e
        if e.shape != (self.num_envs, 3, self.size, self.size):
            raise RuntimeError(f'env.envs has shape {tuple(e.shape)}, expected '
                               f'{(self.num_envs, 3, self.size, self.size)}')
        if e.dtype != torch.float32 or e.device != self.device or not e.is_contiguous():
            # callers rebind env.envs (reference tests/test_single_snake_env.py:54): normalise once
            e = e.to(device=self.device, dtype=torch.float32).contiguous()
            self._envs = e
        self._envs_ok = e
        if self._c is not None:
            self._c.envs = e.data_ptr()
        return e

    def _obs_shape(self, mode: str):
        m, n = _lib.parse_obs_mode(mode)
        N, S = self.num_envs, self.size
        if m in (_lib.OBS_DEFAULT, _lib.OBS_RAW):
            return (N, 3, S, S)
        if m == _lib.OBS_ONE_CHANNEL:
            return (N, 1, S, S)
        if m == _lib.OBS_POSITIONS:
            return (N, 4)
        if m == _lib.OBS_PARTIAL:
            return (N, 3 * (2 * n + 1) ** 2)
        raise Exception  # reference :194-195

    def _parse_mode(self, observation_mode: str):
        if observation_mode in ('default', 'raw', 'one_channel', 'positions'):
            return _lib.parse_obs_mode(observation_mode)
        if isinstance(observation_mode, str) and observation_mode.startswith('partial_'):
            # reference :167 reads the window size from self.observation_mode
            src = self.observation_mode if self.observation_mode.startswith('partial_') else observation_mode
            return _lib.OBS_PARTIAL, int(src.split('_')[-1])
        raise Exception  # reference :194-195

    def _mode_info(self, observation_mode: str):
        """(mode code, window size, observation shape) of an observation mode string, cached per string."""
        key = (observation_mode, self.observation_mode if isinstance(observation_mode, str) and
               observation_mode.startswith('partial_') else None)
        info = self._mode_cache.get(key)
        if info is None:
            m, n = self._parse_mode(observation_mode)
            shape = self._obs_shape(observation_mode if m != _lib.OBS_PARTIAL else f'partial_{n}')
            info = self._mode_cache[key] = (m, n, shape)
        return info

    # ------------------------------------------------------------------ observations

    def _observe(self, observation_mode: str = 'default') -> torch.Tensor:
        """reference :130-195"""
        m, n, shape = self._mode_info(observation_mode)
        envs = self._state()
        obs = torch.empty(shape, dtype=torch.float32, device=self.device)
        rc = _lib.call(self.device.index, _lib.lib().wurm_single_observe, _lib.ptr(envs), _lib.ptr(obs), m, n, _lib.i64(self.num_envs),
                                            self.size, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'SingleSnake._observe')
        return obs

    def _get_rgb(self) -> torch.Tensor:
        """reference :104-128 — int16 RGB image (N,3,S,S)"""
        return (self._observe('default') * 255).round().short()

    # ------------------------------------------------------------------ step

    def _new_slab(self):
        """Fresh output tensors for the next R steps in a few allocations (instead of four per step): R observations,
        R (N,1) rewards, 3 x R flag vectors (wurm_single_slabs).  A slab is never written twice; it is released when the
        last tensor carved from it is."""
        N = self.num_envs
        m, n, shape = self._mode_info(self.observation_mode)
        elems = int(torch.Size(shape).numel()) // N
        per_step = N * (4 * elems + 4 + 3)
        R = max(1, min(64, (32 << 20) // max(per_step, 1)))
        dev = self.device
        obs = torch.empty((R,) + shape, dtype=torch.float32, device=dev)
        reward = torch.empty((R, N, 1), dtype=torch.float32, device=dev)
        flags = torch.empty((3, R, N), dtype=torch.bool, device=dev)
        obs_after = torch.empty((R,) + shape, dtype=torch.float32, device=dev) if self._want_obs_after else None
        self._v_obs, self._v_reward = obs.unbind(0), reward.unbind(0)
        self._v_done2 = flags[0].unsqueeze(-1).unbind(0)
        self._v_selfc, self._v_edgec = flags[1].unbind(0), flags[2].unbind(0)
        self._v_obs_after = obs_after.unbind(0) if obs_after is not None else None
        self._slab_version = flags._version
        self._R, self._slot = R, 0
        if self._c is None:
            c = self._c = _lib.SingleCall()
            c.num_envs, c.env_offset, c.seed, c.size = N, self.env_offset, _lib.u64(self.seed), self.size
            self._pend = torch.zeros(N, dtype=torch.uint8, device=dev)
            c.done_copy = self._pend.data_ptr()
            self._sl = _lib.SingleSlabs()
            self._c_addr, self._sl_addr = ctypes.addressof(c), ctypes.addressof(self._sl)
            self._fn = _lib.step_slot_fn()
            self._get_device, self._get_stream = _lib.accessors()
            self._dev_index = dev.index
            c.envs = self._checked(self._envs).data_ptr()
        self._c.obs_mode, self._c.obs_n = m, n
        sl = self._sl
        sl.obs, sl.reward, sl.flags, sl.steps = obs.data_ptr(), reward.data_ptr(), flags.data_ptr(), R
        sl.obs_after = obs_after.data_ptr() if obs_after is not None else None
        self._slab_mode = self.observation_mode

    @property
    def done(self) -> torch.Tensor:
        """(num_envs,) bool — the `done` of the last step (reference :300 `self.done = done`)."""
        d = self._done
        if d is None:  # the 1-D view of what the last step returned, made on first use
            d = self._done = self._last_done2.view(self.num_envs)
            self._done_from_step = True
        return d

    @done.setter
    def done(self, value: torch.Tensor):
        self._done, self._done_from_step = value, False

    def step(self, actions: torch.Tensor) -> (torch.Tensor, torch.Tensor, torch.Tensor, dict):
        """reference :197-304.  `actions` is sanitised in place (reverse moves become forward moves).  One launch:
        a reset(done) postponed by the previous iteration (module docstring) is applied in front of the transition."""
        dt = actions.dtype
        if dt is torch.int64:
            code = 0  # _lib.ACT_I64
        elif dt is torch.int32:
            code = 1  # _lib.ACT_I32
        elif dt is torch.short:
            # the reference passes its own dtype check and then fails inside scatter_ (:229)
            raise RuntimeError('scatter_(): Expected dtype int32/int64 for index')
        else:
            raise TypeError('actions Tensor must be an integer type i.e. '
                            '{torch.ShortTensor, torch.IntTensor, torch.LongTensor}')
        N = self.num_envs
        if actions.size(0) != N:
            raise RuntimeError('Must have the same number of actions as environments.')
        i = self._slot
        if i >= self._R or self._slab_mode != self.observation_mode or \
                (self._want_obs_after and self._v_obs_after is None):
            self._new_slab()
            i = 0
        self._slot = i + 1
        idx = self._dev_index
        act = actions
        if act.get_device() != idx or act.dim() != 1 or not act.is_contiguous():
            act = actions.to(self.device).reshape(N).contiguous()
        e = self._envs
        if e is not self._envs_ok:
            e = self._checked(e)
        call = self._call
        self._call = call + 1
        pending = self._pending
        if pending:
            self._pending = False
        want_after = self._want_obs_after
        if self._get_device() != idx:  # a process driving several GPUs has another device current
            rc = _lib.call(idx, self._fn, self._c_addr, self._sl_addr, i, act.data_ptr(), code, call, pending,
                           self._pend_call, want_after, _lib.stream_ptr(idx))
        else:
            rc = self._fn(self._c_addr, self._sl_addr, i, act.data_ptr(), code, call, pending, self._pend_call,
                          want_after, self._get_stream(idx))
        if rc:
            _lib.check(rc, 'SingleSnake.step')
        if act is not actions:
            actions.copy_(act.view_as(actions))  # keep the in-place side effect (:222)

        done2 = self._last_done2 = self._v_done2[i]
        self._done = None
        self._last_fresh = True
        self._obs_after = self._v_obs_after[i] if want_after else None
        return self._v_obs[i], self._v_reward[i], done2, {'self_collision': self._v_selfc[i],
                                                           'edge_collision': self._v_edgec[i]}

    # ------------------------------------------------------------------ reset

    def _reset(self, done: torch.Tensor, observe: bool = True):
        if self.initial_snake_length != 3:
            raise NotImplementedError('Only initial snake length = 3 has been implemented.')
        if self.size <= 8:
            # reference :346-347 raises only when an env actually has to be created
            if bool(done.any()):
                raise NotImplementedError('Cannot make an env this small without making this code more clever')
            return self._observe(self.observation_mode) if observe else None
        envs = self._state()
        if observe:
            m, n, shape = self._mode_info(self.observation_mode)
            obs = torch.empty(shape, dtype=torch.float32, device=self.device)
        else:
            m, n, obs = _lib.OBS_NONE, 0, None
        rc = _lib.call(self.device.index, _lib.lib().wurm_single_reset, 
            _lib.ptr(envs), _lib.ptr(done), _lib.ptr(obs), m, n, _lib.i64(self.num_envs), self.size,
            _lib.u64(self.seed), _lib.u64(self._next_call()), _lib.i64(self.env_offset), None, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'SingleSnake.reset')
        return obs

    def reset(self, done: torch.Tensor = None, return_observations: bool = True):
        """Resets environments in which the snake has died (reference :322-342).

        Args:
            done: A 1D Tensor of length self.num_envs (any dtype; (N,1) is accepted). A non-zero value means the
                corresponding environment needs to be reset.  None: use the `done` of the last step.
            return_observations: extension — pass False to skip the observation the reference's callers discard
                (experiments/main.py:227).

        Called with the very `done` the last `step` returned (and nothing in between), the reset is postponed into the
        next step's launch (module docstring); the observation it returns then comes from the step's launch as well.
        """
        if done is None:
            done = self.done
        if self._last_fresh and (done is self._last_done2 or (self._done_from_step and done is self._done)) and \
                self.lazy_reset and self.size > 8 and self.initial_snake_length == 3 and \
                done._version == self._slab_version:
            if not return_observations:
                self._want_obs_after = False
                self._pending, self._pend_call, self._last_fresh = True, self._next_call(), False
                return None
            if self._obs_after is not None:
                obs, self._obs_after = self._obs_after, None
                self._pending, self._pend_call, self._last_fresh = True, self._next_call(), False
                return obs
            self._want_obs_after = True  # from the next step on, the step launch also writes this observation
        done = done.view((done.shape[0]))
        if done.dtype != torch.bool:
            done = done != 0
        if done.device != self.device:
            done = done.to(self.device)
        return self._reset(done.contiguous(), observe=return_observations)

    def _create_envs(self, num_envs: int) -> torch.Tensor:
        """reference :344-387 — a fresh batch of `num_envs` environments (does not touch self.envs)."""
        if self.size <= 8:
            raise NotImplementedError('Cannot make an env this small without making this code more clever')
        if self.initial_snake_length != 3:
            raise NotImplementedError('Only initial snake length = 3 has been implemented.')
        envs = torch.zeros((num_envs, 3, self.size, self.size), device=self.device)
        done = torch.ones(num_envs, dtype=torch.bool, device=self.device)
        rc = _lib.call(self.device.index, _lib.lib().wurm_single_reset, 
            _lib.ptr(envs), _lib.ptr(done), None, _lib.OBS_NONE, 0, _lib.i64(num_envs), self.size,
            _lib.u64(self.seed), _lib.u64(self._next_call()), _lib.i64(self.env_offset), None, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'SingleSnake._create_envs')
        return envs

    # ------------------------------------------------------------------ fused multi-step loop (extension)

    def rollout(self, actions: torch.Tensor, return_observations: bool = True) -> dict:
        """T iterations of `obs, r, d, info = env.step(actions[t]); env.reset(d)` in one kernel launch.

        actions: (T, num_envs) int64/int32 on the device, sanitised in place.  Returns a dict of (T, N, ...) tensors
        (`observations`, `rewards`, `dones`, `self_collision`, `edge_collision`), bit-identical to the Python loop.
        """
        if actions.dtype not in (torch.int, torch.long):
            raise TypeError('actions Tensor must be an integer type i.e. {torch.IntTensor, torch.LongTensor}')
        if actions.dim() != 2 or actions.shape[1] != self.num_envs:
            raise RuntimeError('Must have the same number of actions as environments.')
        if not actions.is_contiguous() or actions.device != self.device:
            raise RuntimeError('rollout actions must be a contiguous device tensor')
        envs = self._state()
        T, N = actions.shape
        if return_observations:
            m, n, shape = self._mode_info(self.observation_mode)
            obs = torch.empty((T,) + shape, dtype=torch.float32, device=self.device)
        else:
            m, n, obs = _lib.OBS_NONE, 0, None
        reward = torch.empty((T, N), dtype=torch.float32, device=self.device)
        flags = torch.empty((3, T, N), dtype=torch.bool, device=self.device)
        rc = _lib.call(self.device.index, _lib.lib().wurm_single_rollout, 
            _lib.ptr(envs), _lib.ptr(actions), _lib.ACT_I64 if actions.dtype == torch.long else _lib.ACT_I32,
            _lib.ptr(reward), _lib.ptr(flags[0]), _lib.ptr(flags[1]), _lib.ptr(flags[2]), _lib.ptr(obs), m, n,
            _lib.i64(N), self.size, _lib.i64(T), _lib.u64(self.seed), _lib.u64(self._next_call(2 * T)),
            _lib.i64(self.env_offset), None, None, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'SingleSnake.rollout')
        self.done = torch.zeros(N, dtype=torch.bool, device=self.device)  # every done env was reset
        return {'observations': obs, 'rewards': reward, 'dones': flags[0], 'self_collision': flags[1],
                'edge_collision': flags[2]}

    def policy_rollout(self, params: torch.Tensor, state: torch.Tensor, num_steps: int, check: bool = True) -> dict:
        """T iterations of the acting half of experiments/main.py:207-227 in one kernel launch:

            probs, value = model(state); action = Categorical(probs).sample()
            state, reward, done, info = env.step(action); env.reset(done)

        params: `wurm_amd.agents.pack_policy_params(agent)` (inputs -> 64 -> 64 -> {4, 1}); state: the observation the
        policy acts on first, (num_envs, 3, 2n+1, 2n+1) as returned by reset / step; the env must be in a
        `partial_n` mode with n <= 3 and size <= 11.  Returns (T, N, ...) tensors: `actions` (sanitised, int64),
        `probs`, `values` (no grad — the learner recomputes them from `observations`), `rewards`, `dones`,
        `self_collision`, `edge_collision`, `observations` (what step t returned, i.e. the policy input of step t+1)
        and `state` = observations[-1].  `check=True` synchronises once and raises if any env was outside the
        kernel's domain (not a well-formed snake — only possible if `env.envs` was edited by hand)."""
        m, n, shape = self._mode_info(self.observation_mode)
        if m != _lib.OBS_PARTIAL or n > 3 or self.size > 11:
            raise NotImplementedError('policy_rollout: partial_n observation with n <= 3 on grids of size <= 11')
        E = 3 * (2 * n + 1) ** 2
        N, T = self.num_envs, int(num_steps)
        if params.dtype != torch.float32 or params.device != self.device or not params.is_contiguous() or \
                params.numel() != 64 * E + 64 + 64 * 64 + 64 + 4 * 64 + 4 + 64 + 1:
            raise RuntimeError('params must be the contiguous fp32 device tensor of pack_policy_params for this observation size')
        if state.device != self.device or state.numel() != N * E:
            raise RuntimeError('state must be the current observation of every env on the env device')
        state = state.to(torch.float32).contiguous()
        envs = self._state()
        dev = self.device
        actions = torch.empty((T, N), dtype=torch.long, device=dev)
        probs = torch.empty((T, N, 4), dtype=torch.float32, device=dev)
        values = torch.empty((T, N), dtype=torch.float32, device=dev)
        reward = torch.empty((T, N), dtype=torch.float32, device=dev)
        flags = torch.empty((3, T, N), dtype=torch.bool, device=dev)
        obs = torch.empty((T,) + shape, dtype=torch.float32, device=dev)
        status = torch.empty(N, dtype=torch.uint8, device=dev)
        rc = _lib.call(dev.index, _lib.lib().wurm_single_policy_rollout,
                       _lib.ptr(envs), _lib.ptr(state), _lib.ptr(params), _lib.ptr(actions), _lib.ptr(probs),
                       _lib.ptr(values), _lib.ptr(reward), _lib.ptr(flags[0]), _lib.ptr(flags[1]), _lib.ptr(flags[2]),
                       _lib.ptr(obs), _lib.ptr(status), n, _lib.i64(N), self.size, _lib.i64(T), _lib.u64(self.seed),
                       _lib.u64(self._next_call(2 * T)), _lib.i64(self.env_offset), _lib.stream_ptr(dev.index))
        _lib.check(rc, 'SingleSnake.policy_rollout')
        if check and T > 0 and bool(status.any()):
            raise RuntimeError('policy_rollout: some envs are not well-formed snakes (status != 0); they were left untouched')
        self.done = torch.zeros(N, dtype=torch.bool, device=dev)  # every done env was reset
        return {'actions': actions, 'probs': probs, 'values': values, 'rewards': reward, 'dones': flags[0],
                'self_collision': flags[1], 'edge_collision': flags[2], 'observations': obs,
                'state': obs[-1] if T > 0 else state.reshape(shape), 'status': status}

    # ------------------------------------------------------------------ invariants

    def check_consistency(self):
        """wurm.utils.env_consistency on self.envs (reference wurm/utils.py:167-178)."""
        from wurm_amd.utils import env_consistency
        env_consistency(self._state())

    # ------------------------------------------------------------------ rendering (host side)

    def render(self, mode: str = 'human'):
        """reference :389-428.  Only 'rgb_array' is provided (the 'human' viewer needs gym/pyglet)."""
        if mode == 'human':
            raise NotImplementedError("render('human') needs gym's SimpleImageViewer; use mode='rgb_array'")
        if mode != 'rgb_array':
            raise ValueError('Render mode not recognised.')
        from wurm_amd._render import frame
        return frame(self._get_rgb().cpu().numpy(), self.num_envs, self.render_args)
