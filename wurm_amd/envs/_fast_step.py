"""Per-call machinery shared by SingleSnake and SimpleGridworld: one launch per `step(a); reset(done)` iteration.

At a few hundred envs the loop of experiments/main.py:212-227 / tests/test_single_snake_env.py:24-31 is bound by host
time per call, not by the GPU, so:
  * `step` is ONE launch (wurm_single_step_slot / wurm_grid_step_slot, include/wurm_hip.h) described by a persistent
    argument block; the tensors it returns are carved from slabs that hold the next few dozen steps' worth of FRESH
    outputs (a slab is never written twice: it is dropped once used up and lives as long as any tensor handed out from
    it), and the call itself goes through the CPython shim wurm_amd/_fastcall instead of ctypes;
  * `reset(done)` with the very `done` that `step` just returned is DEFERRED (`lazy_reset=True`, the default): the next
    `step` rebuilds those envs in front of its own transition, in the same launch, with the RNG counter the eager call
    would have used — bit-identical results.  Anything that looks at the state in between (`env.envs`, `_observe`,
    `check_consistency`, `rollout`, `render`, another `reset`) first flushes the postponed reset with the ordinary reset
    kernel, so `env.envs` always is what the reference would show.  If the caller reads the observation `reset(done)`
    returns, the step launch writes that one too (the observation of every env once the finished ones are rebuilt).
    The one thing that is not tracked is a tensor alias taken BEFORE the deferred reset
    (`e = env.envs; env.step(a); env.reset(d); e[...]`): read the attribute again, or pass `lazy_reset=False`.

The host class provides: num_envs, size, device, seed, env_offset, observation_mode, lazy_reset, _CHANNELS,
_STEP_SLOT (entry point name), _mode_info(mode) -> (mode code, n, obs shape), _lazy_supported(), _launch_reset(envs,
done, obs, mode code, n, call), _configure_call(block).
"""
import ctypes

import torch

from wurm_amd import _lib


class FastStepMixin(object):
    def _fast_init(self):
        self._call = 0
        self._pending = False          # a reset is postponed; its flags are in self._pend, its counter in _pend_call
        self._pend = None              # (N) bytes: the kernels' own copy of the last step's `done`
        self._pend_call = 0
        self._last_done2 = None        # the (N,1) `done` the last step returned
        self._done = None              # env.done: its (N) view (made on first use) or whatever the caller assigned
        self._done_from_step = False
        self._slab_version = -1
        self._last_fresh = False       # no call since that step has consumed an RNG counter or changed the state
        self._obs_after = None         # what reset(done) will return, if the last step produced it
        self._want_obs_after = False   # adaptive: callers that read reset()'s observation get it from the step launch
        self._R = self._slot = 0       # output slabs (see _new_slab)
        self._slab_mode = None
        self._v_obs_after = None
        self._c = None
        self._envs = torch.zeros((self.num_envs, self._CHANNELS, self.size, self.size), device=self.device)
        self._envs_ok = self._envs
        self._done = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)

    def _next_call(self, n: int = 1) -> int:
        c = self._call
        self._call += n
        self._last_fresh = False  # the counter the last step's `obs_after` assumed for its reset is gone
        return c

    # ------------------------------------------------------------------ state

    @property
    def envs(self) -> torch.Tensor:
        """The state tensor, caller-visible and caller-mutable as in the reference (tests/test_single_snake_env.py:54,
        experiments/main.py:215,273).  Reading it applies a postponed reset."""
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
        self._launch_reset(self._checked(self._envs), self._pend, None, _lib.OBS_NONE, 0, self._pend_call)

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
        want = (self.num_envs, self._CHANNELS, self.size, self.size)
        if e.shape != want:
            raise RuntimeError(f'env.envs has shape {tuple(e.shape)}, expected {want}')
        if e.dtype != torch.float32 or e.device != self.device or not e.is_contiguous():
            # callers rebind env.envs (reference tests/test_single_snake_env.py:54): normalise once
            e = e.to(device=self.device, dtype=torch.float32).contiguous()
            self._envs = e
        self._envs_ok = e
        if self._c is not None:
            self._c.envs = e.data_ptr()
        return e

    @property
    def done(self) -> torch.Tensor:
        """(num_envs,) bool — the `done` of the last step (the reference assigns `self.done = done` in step)."""
        d = self._done
        if d is None:  # the 1-D view of what the last step returned, made on first use
            d = self._done = self._last_done2.view(self.num_envs)
            self._done_from_step = True
        return d

    @done.setter
    def done(self, value: torch.Tensor):
        self._done, self._done_from_step = value, False

    # ------------------------------------------------------------------ step

    def _new_slab(self):
        """Fresh output tensors for the next R steps in a few allocations (instead of four per step): R observations,
        R (N,1) rewards, 3 x R flag vectors (wurm_single_slabs).  A slab is never written twice; it is released when the
        last tensor carved from it is."""
        N = self.num_envs
        m, n, shape = self._mode_info(self.observation_mode)
        elems = int(torch.Size(shape).numel()) // max(N, 1)
        per_step = N * (4 * elems + 4 + 3)
        R = max(1, min(64, (32 << 20) // max(per_step, 1)))
        dev = self.device
        obs = torch.empty((R,) + tuple(shape), dtype=torch.float32, device=dev)
        reward = torch.empty((R, N, 1), dtype=torch.float32, device=dev)
        flags = torch.empty((3, R, N), dtype=torch.bool, device=dev)
        obs_after = torch.empty((R,) + tuple(shape), dtype=torch.float32, device=dev) if self._want_obs_after else None
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
            self._fn = _lib.step_slot_fn(self._STEP_SLOT)
            self._get_device, self._get_stream = _lib.accessors()
            self._dev_index = dev.index
            c.envs = self._checked(self._envs).data_ptr()
        self._configure_call(self._c)
        self._c.obs_mode, self._c.obs_n = m, n
        sl = self._sl
        sl.obs, sl.reward, sl.flags, sl.steps = obs.data_ptr(), reward.data_ptr(), flags.data_ptr(), R
        sl.obs_after = obs_after.data_ptr() if obs_after is not None else None
        self._slab_mode = self.observation_mode

    def _fast_step(self, actions: torch.Tensor, what: str):
        """One launch: [postponed reset] + step + observation (+ the observation reset(done) will return).  Returns the
        slot index of the outputs in the current slab."""
        dt = actions.dtype
        if dt is torch.int64:
            code = 0  # _lib.ACT_I64
        elif dt is torch.int32:
            code = 1  # _lib.ACT_I32
        elif dt is torch.short:
            # the reference passes its own dtype check and then fails inside scatter_ (single_snake.py:229)
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
            _lib.check(rc, what)
        if act is not actions:
            actions.copy_(act.view_as(actions))  # SingleSnake sanitises actions in place (single_snake.py:222)
        self._last_done2 = self._v_done2[i]
        self._done = None
        self._last_fresh = True
        self._obs_after = self._v_obs_after[i] if want_after else None
        return i

    # ------------------------------------------------------------------ reset

    def _try_lazy_reset(self, done: torch.Tensor, return_observations: bool):
        """(True, obs) if reset(done) could be postponed into the next step's launch, else (False, None)."""
        if self._last_fresh and (done is self._last_done2 or (self._done_from_step and done is self._done)) and \
                self.lazy_reset and self._lazy_supported() and done._version == self._slab_version:
            if not return_observations:
                self._want_obs_after = False
                self._pending, self._pend_call, self._last_fresh = True, self._next_call(), False
                return True, None
            if self._obs_after is not None:
                obs, self._obs_after = self._obs_after, None
                self._pending, self._pend_call, self._last_fresh = True, self._next_call(), False
                return True, obs
            self._want_obs_after = True  # from the next step on, the step launch also writes this observation
        return False, None
