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

Per-call path: one launch per `step(a); reset(done)` iteration, see wurm_amd/envs/_fast_step.py.
"""
from collections import namedtuple

import ctypes

import torch

from wurm_amd import _lib
from wurm_amd.constants import DEFAULT_DEVICE
from wurm_amd.envs._fast_step import FastStepMixin

Spec = namedtuple('Spec', ['reward_threshold'])

_INT_TYPES = (torch.short, torch.int, torch.long)


def _draw_seed() -> int:
    # one draw from torch's global generator: `torch.manual_seed(k)` before construction pins the trajectories
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class SingleSnake(FastStepMixin):
    """Batched snake environment: state `envs` is (num_envs, 3, size, size) fp32 = [food, head, body]
    (reference single_snake.py:22-47)."""

    _CHANNELS = 3
    _STEP_SLOT = 'wurm_single_step_slot'

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
                 lazy_reset: bool = True,
                 resident_mirror=None):
        """Reference keywords (single_snake.py:55-65) plus this build's: `seed`, `env_offset` (module docstring),
        `lazy_reset` (envs/_fast_step.py) and `resident_mirror` — None: large batches step on a compact mirror of the state
         chosen by batch size with adaptive rules; False: never; True / 'lazy' / 'eager': whenever the
        shape is served, without the adaptive rules (`env.mirror_state()` tells what is in effect and why)."""
        self._resident_policy = resident_mirror
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
        self._mode_cache = {}
        self.lazy_reset = bool(lazy_reset)
        self._fast_init()

        if render_args is None:
            self.render_args = {'num_rows': 1, 'num_cols': 1, 'size': 256}
        else:
            self.render_args = render_args

        self.t = 0

        if not manual_setup:
            # reference :90-93 _create_envs(num_envs): every env is built by the reset kernel
            self._reset(torch.ones(num_envs, dtype=torch.bool, device=self.device), observe=False)

        self.viewer = None

        self.body_colour = torch.tensor((0, 127, 0), dtype=torch.short, device=self.device)
        self.head_colour = torch.tensor((0, 255, 0), dtype=torch.short, device=self.device)
        self.food_colour = torch.tensor((255, 0, 0), dtype=torch.short, device=self.device)
        self.edge_colour = torch.tensor((0, 0, 0), dtype=torch.short, device=self.device)

    # ------------------------------------------------------------------ helpers

    def _lazy_supported(self) -> bool:
        return self.size > 8 and self.initial_snake_length == 3

    def _configure_call(self, c):
        pass

    def _launch_reset(self, envs, done, obs, m, n, call):
        rc = _lib.call(self.device.index, _lib.lib().wurm_single_reset, _lib.ptr(envs), _lib.ptr(done), _lib.ptr(obs),
                       m, n, _lib.i64(self.num_envs), self.size, _lib.u64(self.seed), _lib.u64(call),
                       _lib.i64(self.env_offset), None, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'SingleSnake.reset')

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
        envs = self._state(write=False)
        obs = torch.empty(shape, dtype=torch.float32, device=self.device)
        rc = _lib.call(self.device.index, _lib.lib().wurm_single_observe, _lib.ptr(envs), _lib.ptr(obs), m, n, _lib.i64(self.num_envs),
                                            self.size, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'SingleSnake._observe')
        return obs

    def _get_rgb(self) -> torch.Tensor:
        """reference :104-128 — int16 RGB image (N,3,S,S)"""
        return (self._observe('default') * 255).round().short()

    # ------------------------------------------------------------------ step

    def step(self, actions: torch.Tensor) -> (torch.Tensor, torch.Tensor, torch.Tensor, dict):
        """reference :197-304.  `actions` is sanitised in place (reverse moves become forward moves).  One launch:
        a reset(done) postponed by the previous iteration (wurm_amd/envs/_fast_step.py) is applied in front of the
        transition."""
        return self._fast_step(actions, 'SingleSnake.step')

    def _make_out(self, i: int):
        """what `step` returns for slot i of the current output slab (built once per slab)"""
        return (self._v_obs[i], self._v_reward[i], self._v_done2[i],
                {'self_collision': self._v_selfc[i], 'edge_collision': self._v_edgec[i]})

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
        self._launch_reset(envs, done, obs, m, n, self._next_call())
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
        handled, obs = self._try_lazy_reset(done, return_observations)
        if handled:
            return obs
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
        T, N = actions.shape
        if return_observations:
            m, n, shape = self._mode_info(self.observation_mode)
            obs = torch.empty((T,) + shape, dtype=torch.float32, device=self.device)
        else:
            m, n, obs = _lib.OBS_NONE, 0, None
        reward = torch.empty((T, N), dtype=torch.float32, device=self.device)
        flags = torch.empty((3, T, N), dtype=torch.bool, device=self.device)
        dt = _lib.ACT_I64 if actions.dtype == torch.long else _lib.ACT_I32
        # Grids of 12 x 12 and larger with a mirror (large batches; round 6): the launch reads the clock grids of the per-call
        # step instead of the planes and keeps them current, lazy: without writing the planes (wurm_single_rollout_resident).
        # Same protocol as step(): a postponed reset is applied first, a watched tensor is checked for in-place edits, nothing
        # is "touched".  (9 x 9 has another mirror format and a launch that costs 11 us besides its steps: the library writes
        # a lazy mirror out, rolls out on the planes and leaves the mirror stale — what _state() did on this side before.)
        mirrored = False
        if T > 0:
            if self._fs.pending:
                self._flush()
            self._mirror_sync()
            self._setup_mirror(*self._mode_info(self.observation_mode)[:2])
            mirrored = bool(self._c.resident)
        if mirrored:
            c = self._c
            envs = self._checked(self._envs)
            self._fs.last_fresh = False
            self._chk_void_at = self._fs.steps
            valid = ctypes.c_int(c.resident_valid)
            rc = _lib.call(self.device.index, _lib.lib().wurm_single_rollout_resident,
                _lib.ptr(envs), _lib.ptr(actions), dt, _lib.ptr(reward), _lib.ptr(flags[0]), _lib.ptr(flags[1]),
                _lib.ptr(flags[2]), _lib.ptr(obs), m, n, _lib.i64(N), self.size, _lib.i64(T), _lib.u64(self.seed),
                _lib.u64(self._next_call(2 * T)), _lib.i64(self.env_offset), c.resident, ctypes.addressof(valid),
                int(c.resident_lazy), _lib.stream_ptr(self.device.index))
            c.resident_valid = valid.value if rc == _lib.OK else 0
        else:
            envs = self._state()
            rc = _lib.call(self.device.index, _lib.lib().wurm_single_rollout,
                _lib.ptr(envs), _lib.ptr(actions), dt,
                _lib.ptr(reward), _lib.ptr(flags[0]), _lib.ptr(flags[1]), _lib.ptr(flags[2]), _lib.ptr(obs), m, n,
                _lib.i64(N), self.size, _lib.i64(T), _lib.u64(self.seed), _lib.u64(self._next_call(2 * T)),
                _lib.i64(self.env_offset), None, None, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'SingleSnake.rollout')
        self._done_all_false()  # every done env was reset
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
        self._done_all_false()  # every done env was reset
        return {'actions': actions, 'probs': probs, 'values': values, 'rewards': reward, 'dones': flags[0],
                'self_collision': flags[1], 'edge_collision': flags[2], 'observations': obs,
                'state': obs[-1] if T > 0 else state.reshape(shape), 'status': status}

    # ------------------------------------------------------------------ invariants

    def check_consistency(self, mask: torch.Tensor = None):
        """wurm.utils.env_consistency on self.envs (reference wurm/utils.py:167-178).

        mask (extension): a (num_envs,) bool tensor — only the envs it selects are checked, on the device and without
        gathering them: `env.check_consistency(~done.squeeze(-1))` is what experiments/main.py:214-215 asks with
        `env_consistency(env.envs[~done.squeeze(-1)])`, at the cost of one checker launch and one `any()`."""
        from wurm_amd.utils import consistency_mask, _or_reduce, _raise_for
        sel = None
        if mask is not None:
            sel = mask.view(self.num_envs)
            sel = sel if sel.dtype == torch.bool else sel != 0
        err = self._step_check_mask()  # from inside the last step's launch, where it computed them (resident mirror)
        if err is not None:
            if sel is not None:
                err = err * sel.to(err.dtype)
            if not bool(err.any()):
                return
            if bool((err == -1).any()):  # an env the launch could not vouch for (or a finished one that was asked about)
                err = None
        if err is None:
            err = consistency_mask(self._state(write=False))
            if sel is not None:
                err = err * sel.to(err.dtype)
        _raise_for(_or_reduce(err), one_food=True)

    # ------------------------------------------------------------------ rendering (host side)

    def render(self, mode: str = 'human'):
        """reference :389-428.  Only 'rgb_array' is provided (the 'human' viewer needs gym/pyglet)."""
        if mode == 'human':
            raise NotImplementedError("render('human') needs gym's SimpleImageViewer; use mode='rgb_array'")
        if mode != 'rgb_array':
            raise ValueError('Render mode not recognised.')
        from wurm_amd._render import frame
        return frame(self._get_rgb().cpu().numpy(), self.num_envs, self.render_args)
