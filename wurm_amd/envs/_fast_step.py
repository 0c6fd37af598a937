"""Per-call machinery shared by SingleSnake and SimpleGridworld: one launch per `step(a); reset(done)` iteration.

At a few hundred envs the loop of experiments/main.py:212-227 / tests/test_single_snake_env.py:24-31 is bound by host
time per call, not by the GPU, so:
  * `step` is ONE launch (wurm_single_step_slot / wurm_grid_step_slot, include/wurm_hip.h) described by a persistent
    argument block; the tensors it returns are carved from slabs that hold the next few dozen steps' worth of FRESH
    outputs (a slab is never written twice: it is dropped once used up and lives as long as any tensor handed out from
    it) and are handed back as one PREBUILT tuple per slot;
  * `reset(done)` with the very `done` that `step` just returned is DEFERRED (`lazy_reset=True`, the default): the next
    `step` rebuilds those envs in front of its own transition, in the same launch, with the RNG counter the eager call
    would have used — bit-identical results.  Anything that looks at the state in between (`env.envs`, `_observe`,
    `check_consistency`, `rollout`, `render`, another `reset`) first flushes the postponed reset with the ordinary reset
    kernel, so `env.envs` always is what the reference would show.  If the caller reads the observation `reset(done)`
    returns (the reference's `reset` always returns it, single_snake.py:322-342), the step launch writes that one too
    (the observation of every env once the finished ones are rebuilt).
    A reset is only postponed while nobody else holds a tensor on the state's storage (`_alias_free`: the storage's use
    count; the `envs` attribute hands out a tensor object of the caller's own on the same storage and version counter), so an
    alias the caller keeps — `e = env.envs; env.step(a); env.reset(d); e[...]` — shows the reset state, as the reference's
    would (round 5; rounds 2-4 documented this as a deviation);
  * the bodies of `step` and of the deferred `reset` are a small state machine, `Stepper`: in C where the extension
    wurm_amd/_fastcall is built (wurm_amd/csrc/fastcall.c), else `PyStepper` below — the same logic, and the
    specification of the C type (tests/test_host_lazy_reset.py drives both through the same call patterns).

  * large batches (SingleSnake 9 x 9 `partial_2` / no observation from 4096 envs, 12 x 12 and larger from 2^20 cells:
    wurm_single_resident_bytes) hand the launch a compact MIRROR of the state (wurm_single_call.resident) which it steps
    instead of re-reading the fp32 tensor every call; the tensor stays the state: reading `env.envs` writes it out (and from
    then on every step writes it), any other entry point goes through `_touch()`, a tensor the caller holds is watched for
    in-place edits through its version counter (DESIGN.md §5, §7 deviation 9).

The host class provides: num_envs, size, device, seed, env_offset, observation_mode, lazy_reset, _CHANNELS,
_STEP_SLOT (entry point name), _mode_info(mode) -> (mode code, n, obs shape), _lazy_supported(), _launch_reset(envs,
done, obs, mode code, n, call), _configure_call(block), _make_out(i) -> what step returns for slot i.
"""
import ctypes
import os

import torch

from wurm_amd import _lib


_storage_use_count = torch._C._storage_Use_Count


def parse_mirror_policy(value):
    """`resident_mirror` keyword of the env classes -> None (automatic: by batch size, with the adaptive rules), False
    (never), 'lazy' (whenever the shape is served, no adaptive rule; True means this) or 'eager' (the same, but every step
    also writes the state tensors)."""
    if value is None or value is False:
        return value
    if value is True or value == 'lazy':
        return 'lazy'
    if value == 'eager':
        return 'eager'
    raise ValueError("resident_mirror must be None, True, False, 'lazy' or 'eager'")


class PyStepper(object):
    """Pure-Python twin of wurm_amd._fastcall.Stepper (same attributes, same methods, same results).

    `fn` is a callable fn(call_block_addr, slabs_addr, slot, actions_ptr, actions_dtype, call, apply_pending, pre_call,
    want_obs_after, stream) -> return code: the ctypes function, or a recording stand-in in the host tests."""

    def __init__(self, fn, blk, slabs, get_device, get_stream, dt_i64, dt_i32, dt_i16):
        self.fn, self.blk, self.slabs = fn, blk, slabs
        self.get_device, self.get_stream = get_device, get_stream
        self.dt_i64, self.dt_i32, self.dt_i16 = dt_i64, dt_i32, dt_i16
        self.slot = self.R = self.num_envs = self.dev_index = 0
        self.slab_version = -1
        self.call = self.pend_call = self.steps = 0
        self.ok = self.pending = self.last_fresh = self.want_obs_after = self.lazy_ok = False
        self.outs = self.done2s = self.obs_afters = None
        self.last_done2 = self.done_view = self.obs_after = None
        self.watch, self.watch_version = None, -1  # a tensor the caller may edit in place, and its version when last looked at

    def step(self, actions):
        """The prebuilt output tuple of this step; None: the caller has to prepare something first (new slab, state tensor
        to re-validate or edited in place, actions on another device / not a contiguous vector, another device current) and call again; a
        non-zero int: the entry point's error code.  Argument errors as the reference raises them
        (single_snake.py:198-203; int16 passes its check and then fails in scatter_ at :229)."""
        dt = actions.dtype
        if dt is self.dt_i64:
            code = 0  # _lib.ACT_I64
        elif dt is self.dt_i32:
            code = 1  # _lib.ACT_I32
        elif dt is self.dt_i16:
            raise RuntimeError('scatter_(): Expected dtype int32/int64 for index')
        else:
            raise TypeError('actions Tensor must be an integer type i.e. '
                            '{torch.ShortTensor, torch.IntTensor, torch.LongTensor}')
        if actions.size(0) != self.num_envs:
            raise RuntimeError('Must have the same number of actions as environments.')
        i = self.slot
        if not self.ok or i >= self.R or (self.want_obs_after and self.obs_afters is None):
            return None
        w = self.watch
        if w is not None:
            try:
                ver = w._version
            except RuntimeError:
                ver = -1
            if ver != self.watch_version:
                return None  # the state tensor has been edited in place since the caller took it
        idx = self.dev_index
        if actions.get_device() != idx or actions.dim() != 1 or not actions.is_contiguous():
            return None
        if self.get_device() != idx:  # kernels launch on the current device
            return None
        rc = self.fn(self.blk, self.slabs, i, actions.data_ptr(), code, self.call, self.pending, self.pend_call,
                     self.want_obs_after, self.get_stream(idx))
        if rc:
            return int(rc)  # nothing consumed: the counter, the postponed reset and the slot stay
        self.call += 1
        self.steps += 1
        self.pending = False
        self.slot = i + 1
        self.last_done2 = self.done2s[i]
        self.done_view = None
        self.last_fresh = True
        self.obs_after = self.obs_afters[i] if self.want_obs_after else None
        return self.outs[i]

    # ---- MultiSnake: the same machine over wurm_multi_step_slot (`mfn`), actions a dict of K tensors
    mfn = None
    num_agents = 0
    last_out = None
    state = None        # (the C machine: the tuple of state tensors for its own storage-use-count check)
    alias_free = None   # callable: nobody else holds a tensor on the state's storage

    def step_multi(self, actions):
        """MultiSnake.step(actions: dict): the prebuilt (observations, rewards, dones, info) of this step; None: the caller
        has to prepare something first (as `step`; also: action tensors of another type or device) and calls launch_multi;
        False: everything is in place but the action block — K int64 device vectors that are not the rows of one (K, N)
        tensor: the caller stacks them and calls launch_multi; a non-zero int: the entry point's error code.
        Argument errors as the reference raises them (multi_snake.py:463-472)."""
        if actions.__class__ is not dict:
            return None  # (an OrderedDict may iterate in another order than its dict storage: the caller's generic path)
        if len(actions) != self.num_agents:
            raise RuntimeError('Must have a Tensor of actions for each snake')
        a_ptr, row, n = 0, 8 * self.num_envs, 0
        rows_ok = stackable = True
        for act in actions.values():
            dt = act.dtype
            if dt is not self.dt_i64 and dt is not self.dt_i32 and dt is not self.dt_i16:
                raise TypeError('actions Tensor must be an integer type i.e. '
                                '{torch.ShortTensor, torch.IntTensor, torch.LongTensor}')
            if act.size(0) != self.num_envs:
                raise RuntimeError('Must have the same number of actions as environments.')
            if dt is not self.dt_i64 or act.dim() != 1 or act.get_device() != self.dev_index:
                stackable = False
            if rows_ok:
                if dt is not self.dt_i64 or act.dim() != 1 or not act.is_contiguous() or act.get_device() != self.dev_index:
                    rows_ok = False
                else:
                    ptr = act.data_ptr()
                    if n == 0:
                        a_ptr = ptr
                    elif ptr != a_ptr + n * row:
                        rows_ok = False
            n += 1
        i = self.slot
        if not self.ok or i >= self.R or (self.want_obs_after and self.obs_afters is None):
            return None
        if self.get_device() != self.dev_index:
            return None
        if not rows_ok:   # False: all that is missing is the (K, N) action block (the caller stacks the K vectors)
            return False if stackable else None
        return self.launch_multi(a_ptr)

    def launch_multi(self, a_ptr):
        """the launch of slot `slot` on a prepared (K, N) int64 action block, and the bookkeeping after it"""
        i = self.slot
        rc = self.mfn(self.blk, self.slabs, i, a_ptr, self.call, self.pending, self.pend_call, self.want_obs_after,
                      self.get_stream(self.dev_index))
        if rc:
            return int(rc)
        self.call += 1
        self.steps += 1
        self.pending = False
        self.slot = i + 1
        self.last_done2 = self.done2s[i]
        self.done_view = None
        self.last_fresh = True
        self.obs_after = self.obs_afters[i] if self.want_obs_after else None
        self.last_out = self.outs[i]
        return self.last_out

    def reset_lazy(self, done, return_observations):
        """What reset(done) returns if the reset could be postponed into the next step's launch (None, or the observation
        the last step's launch already wrote), else NotImplemented."""
        if self.last_fresh and self.lazy_ok and (done is self.last_done2 or
                                                 (self.done_view is not None and done is self.done_view)):
            # only while nobody else holds a tensor on the state's storage: through an alias the caller could read or edit the
            # un-reset state, which the reference would show reset (single_snake.py:322-342, multi_snake.py:771-836)
            if self.alias_free is not None and not self.alias_free():
                return NotImplemented
            try:
                ver = done._version
            except RuntimeError:  # inference tensors do not track versions: cannot prove `done` is unmodified
                ver = -1
            if ver >= 0 and ver == self.slab_version:
                if not return_observations:
                    self.want_obs_after = False
                    self.pending, self.pend_call, self.last_fresh = True, self.call, False
                    self.call += 1
                    return None
                if self.obs_after is not None:
                    obs, self.obs_after = self.obs_after, None
                    self.pending, self.pend_call, self.last_fresh = True, self.call, False
                    self.call += 1
                    return obs
                self.want_obs_after = True  # from the next step on, the step launch also writes this observation
            elif ver >= 0:
                self.slab_version = ver  # an in-place edit of one step's flags costs one eager reset, not the slab's rest
        return NotImplemented


def _make_stepper(name, blk_addr, slabs_addr):
    """C Stepper over the entry point's address when wurm_amd/_fastcall is built and the entry point is a real C function;
    else PyStepper over whatever callable _lib.step_slot_fn hands out (ctypes function / test stand-in).  MultiSnake's
    entry point (wurm_multi_step_slot) has another signature: the machine's `mfn` / step_multi / launch_multi side."""
    get_device, get_stream = _lib.accessors()
    fn = _lib.step_slot_fn(name)
    multi = name == 'wurm_multi_step_slot'
    addr = getattr(fn, 'c_address', None)
    if addr is not None:
        try:
            from wurm_amd import _fastcall
            helpers = _lib.torch_helpers() or (None, None, None)
            fs = _fastcall.Stepper(addr, blk_addr, slabs_addr, get_device, get_stream, torch.int64, torch.int32,
                                   torch.int16, *helpers)
            if multi:
                fs.multi = True  # (AttributeError with an older build of the extension: the Python machine)
            return fs
        except (ImportError, AttributeError):
            pass
    fs = PyStepper(None if multi else fn, blk_addr, slabs_addr, get_device, get_stream, torch.int64, torch.int32, torch.int16)
    if multi:
        fs.mfn = fn
    return fs


class FastStepMixin(object):
    # the library's mirror entry points of the class: size with / without the batch-size threshold, flush
    _RESIDENT_FNS = ('wurm_single_resident_bytes', 'wurm_single_resident_size', 'wurm_single_resident_flush')

    def _fast_init(self):
        N = self.num_envs
        c = self._c = _lib.SingleCall()
        c.num_envs, c.env_offset, c.seed, c.size = N, self.env_offset, _lib.u64(self.seed), self.size
        self._sl = _lib.SingleSlabs()
        self._fs = _make_stepper(self._STEP_SLOT, ctypes.addressof(c), ctypes.addressof(self._sl))
        self._fs.alias_free = self._alias_free
        self._fs.num_envs = N
        self._fs.dev_index = -1 if self.device.index is None else self.device.index
        self._get_device = _lib.accessors()[0]
        self._pend = None              # (N) bytes: the kernels' own copy of the last step's `done`
        self._slab_mode = None
        self._envs = torch.zeros((N, self._CHANNELS, self.size, self.size), device=self.device)
        self._fs.state = (self._envs,)
        self._envs_ok = None           # the state tensor that has been validated (None: validate before the next launch)
        # env.done: what the caller (or the constructor / rollout) assigned, valid until the next step overwrites it
        self._done, self._done_stamp = torch.zeros(N, dtype=torch.bool, device=self.device), 0
        # compact mirror of the state the step launch reads instead of `envs` (wurm_single_call.resident; large 9 x 9
        # batches).  The library marks it current after each step launch; everything else that writes the state clears
        # the mark (_touch); a state tensor the caller got hold of is watched for in-place edits through its version counter
        self._mirror, self._mirror_key, self._mirror_off = None, None, False
        pol = parse_mirror_policy(getattr(self, '_resident_policy', None))
        self._resident_policy = pol
        if pol is False:
            self._mirror_off = True
        self._lazy_mirror = pol != 'eager' and os.environ.get('WURM_RESIDENT_LAZY', '1') != '0'
        self._write_outs = self._touches = self._mirror_step0 = 0
        # check_consistency()'s masks from inside the step launch (wurm_single_call.check_mask): asked for once the caller
        # has called check_consistency(), valid for the state the last step left until anything else touches the state
        self._chk, self._chk_armed_at, self._chk_void_at, self._check_calls, self._check_step = None, 1 << 62, -1, 0, 0
        self._mirror_why = 'resident_mirror=False' if pol is False else 'no step yet'
        self._stor = None

    # ------------------------------------------------------------------ copy / pickle
    # The reference's envs are plain attribute bags: copy.deepcopy(env) and pickle work on them (single_snake.py:55-102).
    # Here the object also owns ctypes blocks, a step machine and device buffers that are derived state: they are left out
    # and rebuilt on the other side (`_fast_init`), after the postponed reset and a lazy mirror have been applied to the
    # tensor that IS the state.
    _DERIVED = ('_c', '_sl', '_fs', '_get_device', '_pend', '_slab_mode', '_envs_ok', '_done', '_done_stamp', '_mirror',
                '_mirror_key', '_mirror_off', '_lazy_mirror', '_write_outs', '_touches', '_mirror_step0', '_chk',
                '_chk_armed_at', '_chk_void_at', '_check_calls', '_check_step', '_mirror_why', '_stor', '_v_obs', '_v_reward',
                '_v_done2', '_v_selfc', '_v_edgec')

    def __getstate__(self):
        envs = self._state()   # (applies a postponed reset, writes a lazy mirror out)
        st = {k: v for k, v in self.__dict__.items() if k not in self._DERIVED}
        st['_envs'] = envs
        st['_copy_call'], st['_copy_done'] = int(self._fs.call), self.done
        return st

    def __setstate__(self, st):
        st = dict(st)
        call, done, envs = st.pop('_copy_call'), st.pop('_copy_done'), st.pop('_envs')
        self.__dict__.update(st)
        self._fast_init()
        self.envs = envs
        self._fs.watch, self._fs.watch_version = None, -1   # (nobody holds the restored tensor but this object)
        self._fs.call = call
        self.done = done

    # state of the step machine that other methods of the classes read and write
    _call = property(lambda self: self._fs.call, lambda self, v: setattr(self._fs, 'call', v))
    _pending = property(lambda self: self._fs.pending, lambda self, v: setattr(self._fs, 'pending', v))
    _pend_call = property(lambda self: self._fs.pend_call, lambda self, v: setattr(self._fs, 'pend_call', v))
    _last_fresh = property(lambda self: self._fs.last_fresh, lambda self, v: setattr(self._fs, 'last_fresh', v))

    @property
    def observation_mode(self) -> str:
        return self._observation_mode

    @observation_mode.setter
    def observation_mode(self, value: str):
        self._observation_mode = value
        fs = getattr(self, '_fs', None)
        if fs is not None:
            fs.ok = False  # the next step builds output slabs of the new shape
            # the observation the last step's launch pre-computed for `reset(done)` is one of the OLD mode: the reference
            # observes at reset time in the mode of that moment (single_snake.py:342) — that reset runs eagerly
            fs.obs_after = None

    @property
    def lazy_reset(self) -> bool:
        return self._lazy_reset

    @lazy_reset.setter
    def lazy_reset(self, value: bool):
        self._lazy_reset = bool(value)
        fs = getattr(self, '_fs', None)
        if fs is not None:
            fs.lazy_ok = self._lazy_reset and self._lazy_supported()

    def _next_call(self, n: int = 1) -> int:
        fs = self._fs
        c = fs.call
        fs.call = c + n
        fs.last_fresh = False  # the counter the last step's `obs_after` assumed for its reset is gone
        return c

    # ------------------------------------------------------------------ state

    @property
    def envs(self) -> torch.Tensor:
        """The state tensor, caller-visible and caller-mutable as in the reference (tests/test_single_snake_env.py:54,
        experiments/main.py:215,273).  Reading it applies a postponed reset; the caller may edit what it gets, so an
        observation the last step's launch pre-computed for `reset(done)` is dropped (that reset then runs eagerly)."""
        fs = self._fs
        if fs.pending:
            self._flush()
        fs.obs_after = None
        self._chk_void_at = fs.steps  # (the caller may edit what it gets)
        self._write_out()  # (lazy mirror: the step launches have not been writing `envs`; eager from now on, _watch)
        self._mirror_sync()  # (an edit through an alias since the last look must not be forgotten when _watch takes the
                             # tensor's version again: edit, look, step would step on a stale mirror)
        self._watch(self._envs)
        # a tensor object of the caller's own on the same storage (and version counter): `_alias_free` can then tell from
        # the storage's use count whether the caller still holds the state — or any view of it
        return self._envs.detach()

    @envs.setter
    def envs(self, value: torch.Tensor):
        # the reference would have applied reset(done) to the OLD tensor before this assignment replaced it
        fs = self._fs
        fs.pending = False
        fs.last_fresh = False
        fs.ok = False
        self._touch()  # (a lazy mirror is written out to the tensor that is being replaced, which is still ours here)
        self._envs_ok = None
        self._envs = value.detach() if isinstance(value, torch.Tensor) else value  # (our own tensor object: see the getter)
        self._fs.state = (self._envs,)
        self._watch(self._envs)

    def _alias_free(self) -> bool:
        """Nobody but this object holds a tensor on the state's storage (the caller's `e = env.envs`, a slice of it, the
        tensor it assigned): only then may a reset be postponed — with an alias alive the caller could read or edit the
        un-reset state through it, which the reference would show reset (single_snake.py:322-342)."""
        e = self._envs
        st = self._stor
        if st is None or st[0] is not e:
            s = e.untyped_storage()
            st = self._stor = (e, s, s._cdata)
        return _storage_use_count(st[2]) <= 2  # this object's tensor + the storage handle kept in `_stor`

    def _flush(self):
        """Applies the postponed reset(done) now, with the ordinary reset kernel and the counter it was given."""
        fs = self._fs
        self._touch()
        self._launch_reset(self._checked(self._envs), self._pend, None, _lib.OBS_NONE, 0, fs.pend_call)
        fs.pending = False  # (only once the launch is known to have been accepted)

    def _state(self, write: bool = True) -> torch.Tensor:
        """The state tensor, validated, with any postponed reset applied; the caller is about to change it or to
        consume an RNG counter (write=False: it only reads it with another kernel — the mirror stays current)."""
        fs = self._fs
        if fs.pending:
            self._flush()
        fs.last_fresh = False
        if write:
            self._touch()
        else:
            self._write_out()
        return self._checked(self._envs)

    def _touch(self):
        """Something other than the step launch is about to look at the state or to write it: a lazy mirror is written
        out to `envs` first (the step launches have not been writing them), and the next step rebuilds the mirror."""
        self._chk_void_at = self._fs.steps  # the masks of the last step's launch no longer describe the state
        self._write_out()
        c = self._c
        if c.resident_valid and c.resident:
            # a loop in which (nearly) every step is followed by something that writes the state some other way — an eager
            # reset, a rollout — rebuilds the mirror every step for nothing (SimpleGridworld: with a synchronous read of the
            # build's verdict each time, and 2 = refused again while a hand-made env stays): switch it off for this env object
            self._touches += 1
            if self._resident_policy is None and self._touches >= 8 and \
                    2 * self._touches >= self._fs.steps - self._mirror_step0:
                self._mirror_off, self._mirror, self._mirror_key = True, None, None
                c.resident = None
                self._mirror_why = ('adaptive: the state was written by something other than the step launch after %d of the '
                                    'last %d steps' % (self._touches, self._fs.steps - self._mirror_step0))
        c.resident_valid = 0

    def _write_out(self):
        """`envs` from a lazy mirror (which stays current)"""
        c = self._c
        if c.resident_valid == 1 and c.resident and c.resident_lazy:   # (2: SimpleGridworld's mirror refused — the planes are the state)
            rc = _lib.call(self.device.index, getattr(_lib.lib(), self._RESIDENT_FNS[2]), ctypes.addressof(c),
                           _lib.stream_ptr(self.device.index))
            _lib.check(rc, self._RESIDENT_FNS[2])
            # a caller that keeps looking at the state (check_consistency() every step, experiments/main.py:214-215) pays
            # a whole-state write per look in the lazy form: from the second one on the steps write `envs` themselves
            self._write_outs += 1
            if self._write_outs >= 2 and self._resident_policy is None:
                c.resident_lazy = 0
                self._mirror_why = 'adaptive: the state was looked at twice, every step writes `envs` from now on'

    def _watch(self, t):
        """The caller holds the state tensor `t` from now on and may edit it in place at any time: every step compares its
        version counter (in-place torch ops bump it) and rebuilds the mirror after a change.  Tensors without version
        counters (made under torch.inference_mode()) cannot be watched: no mirror from then on."""
        fs = self._fs
        self._c.resident_lazy = 0  # the caller may READ it at any time as well: the step launches write `envs` from now on
        if self._mirror is not None:
            self._mirror_why = 'the caller holds the state tensor: every step writes `envs`, in-place edits are watched'
        try:
            fs.watch, fs.watch_version = t, t._version
        except (RuntimeError, AttributeError):
            fs.watch, fs.watch_version = None, -1
            self._mirror_off, self._mirror, self._mirror_key = True, None, None
            self._c.resident, self._c.resident_valid = None, 0
            self._mirror_why = 'the state tensor has no version counter (inference mode): cannot be watched'

    def _mirror_sync(self):
        """the step machine saw another version of the watched state tensor: the mirror is stale"""
        fs = self._fs
        w = fs.watch
        if w is not None:
            try:
                ver = w._version
            except RuntimeError:
                ver = -1
            if ver != fs.watch_version:
                self._touch()
                self._watch(w)

    def _setup_mirror(self, m: int, n: int):
        key = (m, n)
        if key == self._mirror_key:
            return
        # another observation mode: a LAZY mirror holds steps `envs` has not seen — written out while the old mirror is
        # still the one the call block names, before it is replaced or dropped
        self._touch()
        self._mirror_key = key
        nbytes = 0
        if not self._mirror_off:
            size_fn = getattr(_lib.lib(), self._RESIDENT_FNS[0 if self._resident_policy is None else 1])
            args = (m, n) if self._CHANNELS == 3 else (m,)   # (SimpleGridworld's observations have no crop radius)
            nbytes = int(size_fn(_lib.i64(self.num_envs), self.size, *args))
            if not self._mirror_off:
                self._mirror_why = ('shape / observation mode not served by the mirror kernels, or batch below the threshold'
                                    if nbytes == 0 else 'on')
        self._mirror = torch.empty(nbytes, dtype=torch.uint8, device=self.device) if nbytes > 0 else None
        self._touches, self._mirror_step0 = 0, self._fs.steps
        self._c.resident = self._mirror.data_ptr() if self._mirror is not None else None
        self._c.resident_valid = 0
        # lazy (the step launches do not write `envs`, _touch() brings them up to date) as long as the caller has never
        # got hold of the state tensor
        self._c.resident_lazy = int(self._mirror is not None and self._fs.watch is None and self._lazy_mirror and
                                    (self._write_outs < 2 or self._resident_policy is not None))
        if self._mirror is not None and self._fs.watch is not None:
            self._mirror_why = 'the caller holds the state tensor: every step writes `envs`, in-place edits are watched'

    def mirror_state(self) -> dict:
        """What the resident mirror of the state is doing for this env object right now:
        `state` 'off' (every step reads `envs`), 'eager' (steps read the mirror and write `envs`) or 'lazy' (steps do
        not write `envs`; they are written out when something looks at them); `why`; `policy` (the `resident_mirror`
        keyword: None = automatic); `current`: the mirror describes the state (False: the next step rebuilds it)."""
        c = self._c
        on = bool(c.resident) and self._mirror is not None
        why = self._mirror_why
        if on and c.resident_valid == 2:
            why = ("refused: the launch that built it found envs outside the lane kernel's domain (hand-made states) — the "
                   'state tensor is stepped as it is, two launches per call, until something writes the state again')
        return {'state': 'off' if not on else ('lazy' if c.resident_lazy else 'eager'), 'why': why,
                'policy': self._resident_policy, 'current': bool(on and c.resident_valid == 1),
                'bytes': int(self._mirror.numel()) if on else 0}

    def _step_check_mask(self):
        """(N) int32 masks of wurm_single_check for the state as it is now, computed inside the last step's launch
        (wurm_single_call.check_mask) — -1 where the launch could not vouch for an env or the env finished in that step —
        or None if they do not describe the current state; arms the request for the following steps."""
        fs, c = self._fs, self._c
        self._check_calls += 1
        self._check_step = fs.steps
        if self._mirror is None or not c.resident or self.size != 9:
            return None      # (only the 9 x 9 lane-resident step computes masks: a grid mirror would pay a fill per step for -1s)
        self._mirror_sync()  # (an in-place edit of a state tensor the caller holds voids the masks)
        if self._chk is None:
            self._chk = torch.full((self.num_envs,), -1, dtype=torch.int32, device=self.device)
        if not c.check_mask:
            c.check_mask = self._chk.data_ptr()
            self._chk_armed_at = fs.steps          # steps from here on write the masks
            return None
        if not c.resident_valid or fs.steps <= self._chk_armed_at or fs.steps <= self._chk_void_at:
            return None
        if fs.pending:  # the postponed reset(done) rebuilds exactly the envs the masks left out: fresh envs are consistent
            return self._chk.masked_fill(self._pend != 0, 0)
        return self._chk

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
            self._fs.state = (e,)
        self._envs_ok = e
        self._c.envs = e.data_ptr()
        self._c.resident_valid = 0  # another tensor (its setter already dealt with a lazy mirror of the old one)
        return e

    @property
    def done(self) -> torch.Tensor:
        """(num_envs,) bool — the `done` of the last step (the reference assigns `self.done = done` in step), or whatever
        was assigned to the attribute since."""
        fs = self._fs
        if self._done_stamp == fs.steps:
            if self._done is None:  # after a rollout: every finished env was reset (made on first use: no fill kernel per launch)
                self._done = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
            return self._done
        self._done = None  # a step has run since the assignment
        d = fs.done_view
        if d is None:  # the 1-D view of what the last step returned, made on first use
            d = fs.done_view = fs.last_done2.view(self.num_envs)
        return d

    @done.setter
    def done(self, value: torch.Tensor):
        self._done, self._done_stamp = value, self._fs.steps

    def _done_all_false(self):
        """env.done after a fused rollout (every finished env has been reset)"""
        self._done, self._done_stamp = None, self._fs.steps

    # ------------------------------------------------------------------ step

    def _new_slab(self):
        """Fresh output tensors for the next R steps in a few allocations (instead of four per step): R observations,
        R (N,1) rewards, 3 x R flag vectors (wurm_single_slabs), and the R tuples `step` returns.  A slab is never written
        twice; it is released when the last tensor carved from it is."""
        N = self.num_envs
        fs = self._fs
        m, n, shape = self._mode_info(self.observation_mode)
        elems = int(torch.Size(shape).numel()) // max(N, 1)
        per_step = N * (4 * elems + 4 + 3)
        R = max(1, min(64, (256 << 20) // max(per_step, 1)))  # (65 536 x 9 x 9 'partial_2': 20 MB per step)
        dev = self.device
        want_after = bool(fs.want_obs_after)
        obs = torch.empty((R,) + tuple(shape), dtype=torch.float32, device=dev)
        reward = torch.empty((R, N, 1), dtype=torch.float32, device=dev)
        flags = torch.empty((3, R, N), dtype=torch.bool, device=dev)
        obs_after = torch.empty((R,) + tuple(shape), dtype=torch.float32, device=dev) if want_after else None
        self._v_obs, self._v_reward = obs.unbind(0), reward.unbind(0)
        self._v_done2 = flags[0].unsqueeze(-1).unbind(0)
        self._v_selfc, self._v_edgec = flags[1].unbind(0), flags[2].unbind(0)
        if self._pend is None:
            self._pend = torch.zeros(N, dtype=torch.uint8, device=dev)
            self._c.done_copy = self._pend.data_ptr()
        self._configure_call(self._c)
        if self._c.check_mask and fs.steps - self._check_step > 64:
            # check_consistency() has not been called for a while: the step launches stop writing its masks
            self._c.check_mask, self._chk_armed_at = None, 1 << 62
        self._c.obs_mode, self._c.obs_n = m, n
        self._setup_mirror(m, n)
        sl = self._sl
        sl.obs, sl.reward, sl.flags, sl.steps = obs.data_ptr(), reward.data_ptr(), flags.data_ptr(), R
        sl.obs_after = obs_after.data_ptr() if obs_after is not None else None
        self._slab_mode = self.observation_mode
        try:
            fs.slab_version = flags._version
        except RuntimeError:   # allocated under torch.inference_mode(): no version counters, so no deferral (eager resets)
            fs.slab_version = -1
        fs.outs = [self._make_out(i) for i in range(R)]
        fs.done2s = list(self._v_done2)
        fs.obs_afters = list(obs_after.unbind(0)) if obs_after is not None else None
        fs.R, fs.slot = R, 0
        fs.lazy_ok = self._lazy_reset and self._lazy_supported()

    def _fast_step(self, actions: torch.Tensor, what: str):
        """`step`: one launch — [postponed reset] + step + observation (+ the observation reset(done) will return)."""
        fs = self._fs
        out = fs.step(actions)
        if out.__class__ is tuple:
            return out
        return self._slow_step(actions, out, what)

    def _slow_step(self, actions: torch.Tensor, out, what: str):
        """What the step machine could not do by itself (it returned None), or an error code of the entry point."""
        fs = self._fs
        if out is None:
            self._mirror_sync()
            if fs.slot >= fs.R or self._slab_mode != self.observation_mode or \
                    (fs.want_obs_after and fs.obs_afters is None):
                self._new_slab()
            else:
                self._configure_call(self._c)  # (attributes the call block carries may have been assigned: start_location)
            e = self._envs
            if e is not self._envs_ok:
                self._checked(e)
            fs.ok = True
            idx = fs.dev_index
            act = actions
            if act.get_device() != idx or act.dim() != 1 or not act.is_contiguous():
                act = actions.to(self.device).reshape(self.num_envs).contiguous()
            if self._get_device() != idx:
                with torch.cuda.device(idx):  # a process driving several GPUs has another device current
                    out = fs.step(act)
            else:
                out = fs.step(act)
            if out.__class__ is tuple:
                if act is not actions:
                    actions.copy_(act.view_as(actions))  # SingleSnake sanitises actions in place (single_snake.py:222)
                return out
        if out is None:
            raise RuntimeError(f'{what}: the step machine refused a prepared call')  # not reachable
        _lib.check(int(out), what)
        raise RuntimeError(f'{what}: unexpected return {out!r}')

    # ------------------------------------------------------------------ reset

    def _try_lazy_reset(self, done: torch.Tensor, return_observations: bool):
        """(True, obs) if reset(done) could be postponed into the next step's launch, else (False, None)."""
        obs = self._fs.reset_lazy(done, return_observations)
        if obs is NotImplemented:
            return False, None
        return True, obs
