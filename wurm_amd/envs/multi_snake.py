"""MultiSnake — drop-in for the reference's wurm.envs.MultiSnake (wurm/envs/multi_snake.py:18-1019): K snakes per
environment, boost moves, food-on-death, respawning; step / reset / _observe run as fused gfx950 kernels behind
the C ABI of include/wurm_hip.h (wurm_multi_*).

State tensors keep the reference's names, shapes and agent order (agent = env * num_snakes + i):
`foods (N,1,S,S)`, `heads` / `bodies (N*K,1,S,S)`, `dones (N*K)`, `orientations (N*K)` int64, `boost_this_step`,
`rewards`, `agent_colours (N*K,3)` int16.  Callers may write into them or rebind them between calls (the
reference's tests do, tests/test_multi_snake_env.py:21-47).  Dynamics attributes (`boost`, `food_on_death_prob`,
`boost_cost_prob`, `food_mode`, `food_rate`, `respawn_mode`, `reward_on_death`) are read at every call and may be
changed after construction, as the reference allows.

Deviations: see wurm_amd/envs/single_snake.py (bool masks, Philox RNG via `seed` / `env_offset`, GPU only);
`dtype=torch.half` (reference :64, experiments/multiagent.py:124-129) is the type of what the env hands to the agents —
observations, `info['food_i']`, `info['size_i']`, converted from the kernels' fp32 outputs (the same two roundings as
the reference's `.to(dtype) / 255`, :281) — while the state tensors stay fp32 (every body value is an exact integer
either way); a snake that finds no room during env creation stays dead instead of raising
(the constructor checks and raises like the reference, `reset()` does not sync to check).

Per-call path: `step` is one launch (wurm_multi_step_reset).  `reset(done, return_observations=False)` called with the very
`dones['__all__']` the last `step` returned is DEFERRED into the next step's launch (`lazy_reset=True`, the default), with
the RNG counter the eager call would have used — bit-identical results; the state attributes (`foods`, `heads`, `bodies`,
`dones`, `orientations`, `agent_colours`) are properties that first apply a postponed reset, as does every method that
looks at the state.  A reset whose observation is asked for (the reference's default) is executed at once the first
time; from then on the step launch also writes that observation (`obs_after`: the reset is applied to the on-chip copy
of the env after the step's own observation, with the counter the eager call would use) and `reset` hands it out and
is deferred like the other form.  A reset is only postponed while nobody else holds a tensor on a state tensor's storage
(`_alias_free`), so an alias the caller keeps shows what the reference's would.
Large batches (from 2^20 cells: wurm_multi_resident_bytes) hand the step launch a compact MIRROR of foods / heads / bodies
(wurm_multi_call.resident) which it steps instead of converting the fp32 tensors every call, and ask it for
check_consistency()'s masks once the caller has used that method; the tensors stay the state — reading one of the three
attributes writes them out (and from then on every step writes them), every other entry point goes through `_touch()`,
tensors the caller holds are watched for in-place edits (DESIGN.md §5, §7 deviation 9).
"""
import ctypes
import os
import sys
import weakref
from collections import namedtuple, OrderedDict
from typing import Dict, Optional, Tuple

import torch

from wurm_amd import _lib
from wurm_amd.constants import DEFAULT_DEVICE, EPS
from wurm_amd.envs.single_snake import _draw_seed

Spec = namedtuple('Spec', ['reward_threshold'])

_INT_TYPES = (torch.short, torch.int, torch.long)
_storage_use_count = torch._C._storage_Use_Count


def _version_of(t: torch.Tensor) -> int:
    """the tensor's version counter, or -1 where there is none (tensors made under torch.inference_mode()): the deferred
    reset needs it to prove that the flags it was handed are still the kernel's own"""
    try:
        return t._version
    except RuntimeError:
        return -1


class _Flushing(object):
    """State attribute of MultiSnake: reading or assigning it first applies a postponed reset(done)."""

    def __init__(self, name):
        self.slot = '_' + name

    def __get__(self, obj, cls):
        if obj is None:
            return self
        if obj._pending:
            obj._flush()
        obj._obs_after = None  # the caller may edit what it gets: reset(done) then observes again instead of reusing it
        obj._chk_fresh = False  # ... and the consistency mask the step launch computed no longer describes the state
        t = getattr(obj, self.slot)
        if self.slot in _MIRRORED:
            obj._escape(t)     # the caller holds a state tensor from now on (resident mirror: written out, watched)
        # a tensor object of the caller's own on the same storage (and version counter): `_alias_free` then tells from the
        # storages' use counts whether the caller still holds a state tensor, or any view of one
        return t.detach()

    def __set__(self, obj, value):
        if obj._pending:
            obj._flush()  # the reference applied the reset before this assignment; the other tensors still need it
        if self.slot in _MIRRORED:
            obj._touch()   # (a lazy mirror is written out to the tensors as they are before one of them is replaced)
            old = getattr(obj, self.slot, None)  # the tensor being replaced is no longer part of the state: not watched
            if old is not None and obj._watched:
                obj._watched = tuple((x, v) for x, v in obj._watched if x is not old)
        obj._last_fresh = False
        obj._chk_fresh = False
        obj._state_dirty = True  # step() re-validates the layout and re-reads the pointers
        if isinstance(value, torch.Tensor):
            value = value.detach()   # (our own tensor object: see __get__)
        setattr(obj, self.slot, value)
        obj._sync_state()
        if self.slot in _MIRRORED:
            obj._escape(value)


class _LazyInfo(dict):
    """`info` of one step (reference :707-729: snake_collision_i, edge_collision_i, food_i, size_i, boost_i).  A dict whose 5 K
    tensors are carved from the step's output block the first time anything looks at it: a view object per row is most of
    what a step costs on the host, and callers that read `info` every step are rare (the reference's loops log from it every
    few hundred steps).  Every reading method fills it first, so it behaves as the plain dict it then is."""
    __slots__ = ('_src',)

    def __init__(self, src):
        dict.__init__(self)
        self._src = src

    def _fill(self):
        src = self._src
        if src is not None:
            self._src = None
            f, b, keys, K = src  # the step's packed floats (6K, N) / flags (7K + 1, N): agent-major rows from 3K on
            rf, rb = f[3 * K:].unbind(0), b[3 * K:7 * K].unbind(0)
            k_snake, k_edge, k_food, k_boost, k_size = keys
            dict.update(self, zip(k_snake, rb[2 * K:3 * K]))
            dict.update(self, zip(k_edge, rb[3 * K:]))
            dict.update(self, zip(k_food, rf[K:2 * K]))
            dict.update(self, zip(k_size, rf[2 * K:]))
            dict.update(self, zip(k_boost, rb[K:2 * K]))

    def __missing__(self, key):
        if self._src is None:
            raise KeyError(key)
        self._fill()
        return dict.__getitem__(self, key)

    def __reduce__(self):
        self._fill()
        return (dict, (dict(self),))


def _filling(name):
    method = getattr(dict, name)

    def wrapper(self, *a, **kw):
        self._fill()
        return method(self, *a, **kw)
    wrapper.__name__ = name
    return wrapper


for _name in ('__contains__', '__iter__', '__len__', '__eq__', '__ne__', '__repr__', '__setitem__', '__delitem__', '__or__',
              '__ror__', '__ior__', '__reversed__', 'get', 'keys', 'values', 'items', 'copy', 'pop', 'popitem', 'setdefault',
              'update', 'clear'):
    setattr(_LazyInfo, _name, _filling(_name))


class _LazyResetObs(dict):
    """What `reset(done)` returns once the caller has been seen to DISCARD it (experiments/speeds.py:30-38 does: `env.reset(
    done['__all__'])` as a statement): the K observations of the reference's return value (:834-836) are not worth a second
    observation stream out of every step launch for a caller that drops them.  The dict fills itself — the postponed reset
    applied, `_observe` of the mode the reset was called in — the first time anything reads it; and if the caller still
    holds it when the state is about to change (the next step, anything that flushes the postponed reset) the env fills it
    first, so it always holds what the reference would have returned.  Either way the env goes back to precomputing."""
    __slots__ = ('_env', '_mode', '__weakref__')

    def __init__(self, env, mode):
        dict.__init__(self)
        self._env, self._mode = env, mode

    def _fill(self):
        env = self._env
        if env is not None:
            self._env = None
            env._settle_reset_obs(self)

    def __missing__(self, key):
        if self._env is None:
            raise KeyError(key)
        self._fill()
        return dict.__getitem__(self, key)

    def __reduce__(self):
        self._fill()
        return (dict, (dict(self),))


for _name in ('__contains__', '__iter__', '__len__', '__eq__', '__ne__', '__repr__', '__setitem__', '__delitem__', '__or__',
              '__ror__', '__ior__', '__reversed__', 'get', 'keys', 'values', 'items', 'copy', 'pop', 'popitem', 'setdefault',
              'update', 'clear'):
    setattr(_LazyResetObs, _name, _filling(_name))


class _Dynamic(object):
    """Attribute of MultiSnake that callers assign after construction (reference tests/test_multi_snake_env.py:180,288,
    401-403,618; experiments/multiagent.py:340,345 anneal `food_rate` / `food_on_death_prob` every step).  The reference reads
    it at the moment of each call.  `reset` — the kernels' and the reference's (:771-836) — reads only `respawn_mode`,
    `colour_mode` and `initial_snake_length`: before one of THOSE changes a postponed reset(done) is applied with the old
    value (that reset was called then) and the observation the last step's launch pre-computed for `reset(done)` is dropped
    (`reset=True`); `observation_mode` only drops that observation; the step dynamics only mark the configuration block."""

    def __init__(self, name, reset=False, observes=False):
        self.slot, self.reset, self.drops = '_dyn_' + name, reset, reset or observes

    def __get__(self, obj, cls):
        if obj is None:
            return self
        try:
            return obj.__dict__[self.slot]
        except KeyError:
            raise AttributeError(self.slot[5:]) from None

    def __set__(self, obj, value):
        if self.reset and obj._pending:
            obj._flush()
        if self.drops:
            obj._obs_after = None
        obj._cfg_dirty = True
        obj._fs.ok = False   # the next step goes through _slow_step (configuration block, observation shape)
        obj.__dict__[self.slot] = value


_MIRRORED = ('_foods', '_heads', '_bodies')  # what wurm_multi_call.resident mirrors
_SLOTS = _MIRRORED + ('_dones', '_orientations', '_agent_colours')


class MultiSnake(object):
    """Batched multi-agent snake environment (reference multi_snake.py:18-48)."""

    # The deferred-reset protocol lives in the step machine `_fs` (wurm_amd/envs/_fast_step.py: PyStepper, the specification;
    # wurm_amd/csrc/fastcall.c: Stepper, its C twin) — the same one SingleSnake and SimpleGridworld use.
    _pending = property(lambda self: self._fs.pending, lambda self, v: setattr(self._fs, 'pending', v))
    _pend_call = property(lambda self: self._fs.pend_call, lambda self, v: setattr(self._fs, 'pend_call', v))
    _call = property(lambda self: self._fs.call, lambda self, v: setattr(self._fs, 'call', v))
    _last_fresh = property(lambda self: self._fs.last_fresh, lambda self, v: setattr(self._fs, 'last_fresh', v))
    _obs_after = property(lambda self: self._fs.obs_after, lambda self, v: setattr(self._fs, 'obs_after', v))
    _want_after = property(lambda self: self._fs.want_obs_after, lambda self, v: setattr(self._fs, 'want_obs_after', v))
    _steps = property(lambda self: self._fs.steps)
    _mirror = None          # the compact mirror of foods / heads / bodies the step launch reads instead of them
    _mirror_off = False
    _lazy_mirror = os.environ.get('WURM_RESIDENT_LAZY', '1') != '0'
    _watched = ()           # (tensor, version) of state tensors the caller holds: in-place edits make the mirror stale
    _write_outs = _touches = 0
    _chk = None             # (2, N) int32: check_consistency's masks of the post-step / post-reset state, from the step launch
    _chk_has_after = False
    _chk_armed_at, _chk_void_at = 1 << 62, -1
    _check_calls = _check_step = 0
    _rewards_t = _boost_t = _mc_mode = _info = _rewards_src = None
    _rewards_at = _boost_at = 0
    _slab = None            # (out_f32, out_u8) of the current output slab
    _mc_ready = False
    # what reset(done) returned last / whether the caller keeps it (module docstring: "discarded reset observations")
    _reset_obs_probe = None  # the precomputed dict the last reset(done) handed out: looked at by the next step
    _reset_obs_drops = 0     # consecutive resets whose returned dict nobody held at the next step
    _RESET_OBS_DROPS = 3     # ... this many: reset(done) returns a _LazyResetObs and the steps stop precomputing
    _lazy_obs_mode = False
    _lazy_obs_ref = None     # weak reference to the _LazyResetObs of the postponed reset, while it may still need filling
    _abuf = None             # (K, N) int64: the action block of a step whose K action tensors had to be stacked

    @property
    def _state_dirty(self):
        return not self._fs.ok

    @_state_dirty.setter
    def _state_dirty(self, v):
        if v:
            self._fs.ok = False  # the next step goes through _slow_step, which re-validates the layout and re-reads the pointers

    @property
    def _chk_fresh(self):
        """the masks the last step's launch wrote describe the state as it is (nothing has looked at or written it since)"""
        c, st = self._mc, self._fs.steps
        return bool(c.resident) and bool(c.check_mask) and st > self._chk_armed_at and st > self._chk_void_at

    @_chk_fresh.setter
    def _chk_fresh(self, v):
        if not v:
            self._chk_void_at = self._fs.steps

    foods = _Flushing('foods')
    heads = _Flushing('heads')
    bodies = _Flushing('bodies')
    dones = _Flushing('dones')
    orientations = _Flushing('orientations')
    agent_colours = _Flushing('agent_colours')
    # (observation_mode: nothing to apply — a postponed reset does not observe — but what was pre-computed is of the old mode)
    observation_mode = _Dynamic('observation_mode', observes=True)
    respawn_mode = _Dynamic('respawn_mode', reset=True)
    food_on_death_prob = _Dynamic('food_on_death_prob')
    boost = _Dynamic('boost')
    boost_cost_prob = _Dynamic('boost_cost_prob')
    food_mode = _Dynamic('food_mode')
    food_rate = _Dynamic('food_rate')
    max_food = _Dynamic('max_food')
    reward_on_death = _Dynamic('reward_on_death')
    colour_mode = _Dynamic('colour_mode', reset=True)
    initial_snake_length = _Dynamic('initial_snake_length', reset=True)
    spec = Spec(float('inf'))
    metadata = {
        'render.modes': ['rgb_array'],
        'video.frames_per_second': 12
    }

    def __init__(self,
                 num_envs: int,
                 num_snakes: int,
                 size: int,
                 initial_snake_length: int = 3,
                 on_death: str = 'restart',
                 observation_mode: str = 'full',
                 device: str = DEFAULT_DEVICE,
                 dtype: torch.dtype = torch.float,
                 manual_setup: bool = False,
                 food_on_death_prob: float = 0.5,
                 boost: bool = True,
                 boost_cost_prob: float = 0.5,
                 food_mode: str = 'only_one',
                 food_rate: float = 5e-4,
                 respawn_mode: str = 'all',
                 reward_on_death: int = -1,
                 verbose: int = 0,
                 render_args: dict = None,
                 agent_colours: str = 'random',
                 seed: int = None,
                 env_offset: int = 0,
                 lazy_reset: bool = True,
                 resident_mirror=None):
        """Reference keywords (multi_snake.py:56-75) plus this build's: `seed`, `env_offset`, `lazy_reset` and
        `resident_mirror` — None: large batches step on a compact mirror of foods / heads / bodies chosen
        by batch size with adaptive rules; False: never; True / 'lazy' / 'eager': always, without the adaptive rules
        (`env.mirror_state()` tells what is in effect and why)."""
        from wurm_amd.envs._fast_step import parse_mirror_policy
        self.device = _lib.require_device(device)
        self._make_machine(num_envs, num_snakes)
        self._resident_policy = pol = parse_mirror_policy(resident_mirror)
        self._mirror_off = pol is False
        self._lazy_mirror = pol != 'eager' and MultiSnake._lazy_mirror
        self._mirror_why = 'resident_mirror=False' if pol is False else 'no step yet'
        self.num_envs = num_envs
        self.num_snakes = num_snakes
        self.lazy_reset = bool(lazy_reset)
        self._pend = None            # (N) bytes: the kernels' own copy of the last step's dones['__all__']
        self._cfg_cache = (None, None)
        self._lifetimes_touched = False
        self._stor = [None] * len(_SLOTS)
        self.size = size
        self.initial_snake_length = initial_snake_length
        self.on_death = on_death
        self.verbose = verbose
        if dtype not in (torch.float, torch.half):
            raise NotImplementedError('wurm_amd.MultiSnake: dtype must be torch.float or torch.half')
        self.dtype = dtype
        self._half = dtype == torch.half
        self.observation_mode = observation_mode
        if observation_mode.startswith('partial_'):
            self.observation_width = int(observation_mode.split('_')[1])
            self.observation_size = 2 * int(observation_mode.split('_')[1]) + 1
        self.seed = _draw_seed() if seed is None else int(seed)
        self.env_offset = int(env_offset)

        if render_args is None:
            self.render_args = {'num_rows': 1, 'num_cols': 1, 'size': 256}
        else:
            self.render_args = render_args

        N, K, S, dev = num_envs, num_snakes, size, self.device
        # (the raw slots: the class's own tensors are not "held by the caller" until an attribute is read or assigned)
        self._foods = torch.zeros((N, 1, S, S), dtype=torch.float32, device=dev)    # fp32 whatever `dtype` (docstring)
        self._heads = torch.zeros((N * K, 1, S, S), dtype=torch.float32, device=dev)
        self._bodies = torch.zeros((N * K, 1, S, S), dtype=torch.float32, device=dev)
        self.dones = torch.zeros(N * K, dtype=torch.bool, device=dev)
        self.boost_this_step = torch.zeros(N * K, dtype=torch.bool, device=dev)
        self.rewards = torch.zeros(N * K, dtype=torch.float, device=dev)
        self._env_lifetimes = torch.zeros(N, dtype=torch.long, device=dev)
        self.snake_lifetimes = torch.zeros((N, K), dtype=torch.long, device=dev)
        self.orientations = torch.zeros(N * K, dtype=torch.long, device=dev)
        self.viewer = None

        ###################################
        # Environment dynamics parameters #
        ###################################
        self.respawn_mode = respawn_mode
        self.food_on_death_prob = food_on_death_prob
        self.boost = boost
        self.boost_cost_prob = boost_cost_prob
        self.food_mode = food_mode
        self.food_rate = food_rate
        self.max_food = self.num_snakes * 8
        self.max_env_lifetime = 5000
        self.reward_on_death = reward_on_death

        # Rendering parameters (reference :131-141)
        self.self_colour = torch.tensor((0, 192, 0), dtype=torch.short, device=dev)
        self.self_boost_colour = torch.tensor((0, 255, 0), dtype=torch.short, device=dev)
        self.other_colour = torch.tensor((0, 0, 192), dtype=torch.short, device=dev)
        self.other_boost_colour = torch.tensor((0, 0, 255), dtype=torch.short, device=dev)
        self.food_colour = torch.tensor((255, 0, 0), dtype=torch.short, device=dev)
        self.edge_colour = torch.tensor((0, 0, 0), dtype=torch.short, device=dev)

        if agent_colours == 'random':
            self.colour_mode = 'random'
        elif agent_colours == 'fixed':
            self.colour_mode = 'fixed'
        else:
            raise ValueError('agent_colours must in {random, fixed}')
        self.agent_colours = torch.empty((N * K, 3), dtype=torch.short, device=dev)
        rc = _lib.call(self.device.index, _lib.lib().wurm_multi_colours, _lib.ptr(self.agent_colours), _lib.i64(N), K,
                                           int(self.colour_mode == 'fixed'), _lib.u64(self.seed),
                                           _lib.u64(self._next_call()), _lib.i64(self.env_offset), _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'MultiSnake.get_n_colours')
        self.num_colours = self.agent_colours.shape[0]

        self.info = {}

        self.edge_locations_mask = torch.zeros((1, 1, S, S), dtype=self.dtype, device=dev)
        self.edge_locations_mask[:, :, :1, :] = 1
        self.edge_locations_mask[:, :, :, :1] = 1
        self.edge_locations_mask[:, :, -1:, :] = 1
        self.edge_locations_mask[:, :, :, -1:] = 1

        if not manual_setup:
            # reference :111-116 _create_envs(num_envs); raises if a snake cannot be placed (:946-947)
            failures = self._reset_kernel(torch.ones(N, dtype=torch.bool, device=dev), observe=False,
                                          want_status=True)
            if failures:
                raise RuntimeError('There is no available locations to create snake!')

    def _make_machine(self, num_envs: int, num_snakes: int):
        """the argument block of the per-call launches, the output slabs and the step machine over them (the block is filled
        in on first use: _ensure_call) — derived state: made by the constructor and again by __setstate__"""
        from wurm_amd.envs._fast_step import _make_stepper
        self._cfg_dirty = True
        self._mc, self._sl = _lib.MultiCall(), _lib.MultiSlabs()
        self._mc_addr = ctypes.addressof(self._mc)
        self._fs = fs = _make_stepper('wurm_multi_step_slot', self._mc_addr, ctypes.addressof(self._sl))
        fs.num_envs, fs.num_agents = num_envs, num_snakes
        fs.alias_free = self._alias_free
        fs.dev_index = -1 if self.device.index is None else self.device.index

    # ------------------------------------------------------------------ copy / pickle
    # The reference's env is a plain attribute bag: copy.deepcopy(env) and pickle work on it (multi_snake.py:56-160).  The
    # ctypes blocks, the step machine, the mirror and the output slabs are derived state: left out here and rebuilt on the
    # other side, after the postponed reset and a lazy mirror have been applied to the tensors that ARE the state.
    _DERIVED = ('_mc', '_sl', '_mc_addr', '_fs', '_pend', '_cfg_cache', '_mc_cfg', '_mc_ready', '_mc_mode', '_mirror',
                '_mirror_off', '_lazy_mirror', '_mirror_why', '_watched', '_write_outs', '_touches', '_chk', '_chk_has_after',
                '_chk_armed_at', '_chk_void_at', '_check_calls', '_check_step', '_slab', '_stor', '_keys', '_get_device',
                '_rewards_t', '_boost_t', '_rewards_at', '_boost_at', '_rewards_src', '_info', '_cfg_dirty',
                '_reset_obs_probe', '_reset_obs_drops', '_lazy_obs_mode', '_lazy_obs_ref', '_abuf')

    def __getstate__(self):
        self._state()   # (applies a postponed reset, writes a lazy mirror out)
        st = {k: v for k, v in self.__dict__.items() if k not in self._DERIVED}
        st['_copy_call'] = int(self._fs.call)
        st['_copy_rewards'], st['_copy_boost'] = self.rewards, self.boost_this_step
        return st

    def __setstate__(self, st):
        st = dict(st)
        call, rewards, boost = st.pop('_copy_call'), st.pop('_copy_rewards'), st.pop('_copy_boost')
        self.__dict__.update(st)
        self._make_machine(self.num_envs, self.num_snakes)
        self._pend, self._cfg_cache, self._stor = None, (None, None), [None] * len(_SLOTS)
        self._mirror_off = self._resident_policy is False
        self._lazy_mirror = self._resident_policy != 'eager' and MultiSnake._lazy_mirror
        self._mirror_why = 'resident_mirror=False' if self._resident_policy is False else 'no step yet'
        self._fs.call = call
        self._fs.lazy_ok = self._lazy_ok()
        self.rewards, self.boost_this_step = rewards, boost
        self._sync_state()

    # ------------------------------------------------------------------ helpers

    @property
    def lazy_reset(self) -> bool:
        return self._lazy_reset

    @lazy_reset.setter
    def lazy_reset(self, value: bool):
        self._lazy_reset = bool(value)
        self._fs.lazy_ok = self._lazy_ok()

    def _lazy_ok(self) -> bool:
        # (env_lifetimes: `env_lifetimes[done] = 0`, :797, is a no-op only while nobody has asked for that tensor)
        return self._lazy_reset and getattr(self, 'size', 0) >= 5 and not self._lifetimes_touched

    def _next_call(self, n: int = 1) -> int:
        fs = self._fs
        c = fs.call
        fs.call = c + n
        fs.last_fresh = False  # the counter the last step's `obs_after` assumed for its reset is gone
        return c

    @property
    def info(self) -> dict:
        """reference :131 / :729: the info dict of the last step"""
        if self._info is not None:
            return self._info
        out = self._fs.last_out
        return out[3] if out is not None else {}

    @info.setter
    def info(self, value):
        self._info = value

    @property
    def env_lifetimes(self) -> torch.Tensor:
        """reference :106.  The reference never increments it (its use at :705 is dead code): as long as no
        caller has asked for the tensor it is all zeros and `dones['__all__'] |= env_lifetimes > max_env_lifetime`
        (:705) / `env_lifetimes[done] = 0` (:797) are skipped; once somebody has, both run as in the reference."""
        self._touch_lifetimes()
        return self._env_lifetimes

    @env_lifetimes.setter
    def env_lifetimes(self, value):
        self._touch_lifetimes()
        self._env_lifetimes = value

    def _touch_lifetimes(self):
        if not self._lifetimes_touched:
            if self._pending:
                self._flush()   # (a reset postponed as a no-op on lifetimes: applied before somebody can see them)
            self._lifetimes_touched = True
            self._fs.lazy_ok = False
            self._fs.last_fresh = False

    @property
    def rewards(self) -> torch.Tensor:
        """reference :105/:478: (num_envs*num_snakes,) float, env-major — a view of the last step's output block"""
        st = self._fs.steps
        if self._rewards_at != st:  # (the env-major block at the start of the last step's packed floats)
            self._rewards_t, self._rewards_at = self._slab[0][self._fs.slot - 1].view(-1)[:self.num_envs * self.num_snakes], st
        elif self._rewards_t is None:  # (after a rollout: its last step's (K, N) block, env-major on demand)
            self._rewards_t, self._rewards_src = self._rewards_src.t().reshape(-1), None
        return self._rewards_t

    @rewards.setter
    def rewards(self, value):
        self._rewards_t, self._rewards_at = value, self._fs.steps

    @property
    def boost_this_step(self) -> torch.Tensor:
        """reference :104: (num_envs*num_snakes,) env-major — a view of the last step's output block"""
        st = self._fs.steps
        if self._boost_at != st:
            self._boost_t, self._boost_at = self._slab[1][self._fs.slot - 1].view(-1)[:self.num_envs * self.num_snakes], st
        return self._boost_t

    @boost_this_step.setter
    def boost_this_step(self, value):
        self._boost_t, self._boost_at = value, self._fs.steps

    def _sync_state(self):
        """the state tensors as the step machine knows them (its own storage-use-count check, wurm_torch_alias_free)"""
        d = self.__dict__
        self._fs.state = tuple(d[n] for n in _SLOTS) if all(n in d for n in _SLOTS) else None

    def _alias_free(self) -> bool:
        """Nobody but this object holds a tensor on the storage of a state tensor (what the caller read from `env.foods`
        ..., a slice of it, a tensor it assigned): only then may a reset be postponed — through an alias the caller could
        read or edit the un-reset state, which the reference would show reset (multi_snake.py:771-836)."""
        st = self._stor
        for i, name in enumerate(_SLOTS):
            t = getattr(self, name)
            if st[i] is None or st[i][0] is not t:
                s = t.untyped_storage()
                st[i] = (t, s, s._cdata)
            if _storage_use_count(st[i][2]) > 2:  # this object's tensor + the storage handle kept in `_stor`
                return False
        return True

    def _flush(self):
        """Applies the postponed reset(done) now, with the ordinary reset kernel and the counter it was given."""
        fs = self._fs
        fs.pending = False
        self._launch_reset(self._pend, None, _lib.OBS_NONE, 0, fs.pend_call, None)
        if self._lazy_obs_ref is not None:   # the dict that reset returned, if the caller still holds it: filled NOW, while
            lz = self._lazy_obs_ref()        # the state is what the reset left
            if lz is not None:
                lz._fill()
            self._lazy_obs_ref = None

    def _settle_reset_obs(self, lz):
        """fills the _LazyResetObs `lz`: the postponed reset applied (if it still is postponed), then the observation of the
        mode that reset was called in — what the reference's reset returned (:834-836).  The caller reads, or still holds,
        what reset returns after all: back to precomputing it in the step launch."""
        self._lazy_obs_ref = None
        if self._pending:
            self._flush()
        obs = self._observe(lz._mode)
        dict.update(lz, obs)
        self._lazy_obs_mode = False
        self._reset_obs_drops = -(1 << 20)   # (no flapping: a caller that reads it once is a caller that reads it)
        self._fs.want_obs_after = True

    def _reset_obs_bookkeeping(self):
        """in front of a step: did the caller keep what the last reset(done) returned?"""
        probe, self._reset_obs_probe = self._reset_obs_probe, None
        if probe is not None:
            # the references of a dict nobody else holds: the slab's list of precomputed dicts, `probe`, getrefcount's argument
            if sys.getrefcount(probe) <= 3:
                self._reset_obs_drops += 1
                if self._reset_obs_drops >= self._RESET_OBS_DROPS and not self._half:
                    self._lazy_obs_mode = True
            else:
                self._reset_obs_drops = min(self._reset_obs_drops, 0)
        if self._lazy_obs_ref is not None:
            lz = self._lazy_obs_ref()
            self._lazy_obs_ref = None
            if lz is not None:
                lz._fill()   # still held: the state is about to change

    # ---- the resident mirror (include/wurm_hip.h: wurm_multi_call.resident; protocol as in envs/_fast_step.py)

    def _write_out(self):
        """foods / heads / bodies from a lazy mirror (which stays current)"""
        c = self._mc
        if c is not None and c.resident and c.resident_lazy and c.resident_valid:
            rc = _lib.call(self.device.index, _lib.lib().wurm_multi_resident_flush, self._mc_addr,
                           _lib.stream_ptr(self.device.index))
            _lib.check(rc, 'wurm_multi_resident_flush')
            # a caller that keeps looking at the state (experiments/speeds.py:30-38: check_consistency() every step) pays a
            # whole-state write per look in the lazy form: from the second one on the steps write the tensors themselves
            self._write_outs += 1
            if self._write_outs >= 2 and self._resident_policy is None:
                c.resident_lazy = 0
                self._lazy_mirror = False
                self._mirror_why = 'adaptive: the state was looked at twice, every step writes the tensors from now on'

    def _touch(self):
        """something other than the step launch is about to read or write the state tensors"""
        self._write_out()
        self._chk_fresh = False
        c = self._mc
        if c is not None:
            if c.resident and c.resident_valid:
                # a loop in which (nearly) every step is followed by something that writes the state some other way (an
                # eager reset: experiments/speeds.py's check_consistency() flushes the postponed one every step) rebuilds
                # the mirror every step for nothing: switch it off for this env object
                self._touches += 1
                if self._resident_policy is None and self._touches >= 8 and 2 * self._touches >= self._steps:
                    self._mirror_off, self._mirror = True, None
                    c.resident = None
                    self._mirror_why = ('adaptive: the state was written by something other than the step launch after %d '
                                        'of %d steps' % (self._touches, self._steps))
            c.resident_valid = 0

    def _escape(self, t):
        """The caller holds the state tensor `t` from now on: it is brought up to date, written by every step (no lazy form
        any more) and watched for in-place edits through its version counter; without one (inference tensors) no mirror."""
        if self._watched:
            self._watch_ok()   # (an edit through a tensor the caller already holds must not be forgotten when its version is
                               # taken again below: edit, look, step would step on a stale mirror)
        self._write_out()
        self._chk_fresh = False
        if self._mc is not None:
            self._mc.resident_lazy = 0
        self._lazy_mirror = False
        if not self._mirror_off:
            self._mirror_why = 'the caller holds a state tensor: every step writes them, in-place edits are watched'
        ver = _version_of(t) if isinstance(t, torch.Tensor) else -1
        if ver < 0:
            self._mirror_off, self._mirror = True, None
            self._mirror_why = 'a state tensor has no version counter (inference mode): cannot be watched'
            if self._mc is not None:
                self._mc.resident, self._mc.resident_valid = None, 0
            return
        self._watched = tuple((x, v) for x, v in self._watched if x is not t) + ((t, ver),)

    def _watch_ok(self) -> bool:
        """False (and the mirror marked stale) if a watched tensor has been edited in place since the last look"""
        ok = True
        for x, v in self._watched:
            if _version_of(x) != v:
                ok = False
        if not ok:
            self._touch()
            self._watched = tuple((x, _version_of(x)) for x, _ in self._watched)
            if any(v < 0 for _, v in self._watched):
                self._mirror_off, self._mirror, self._watched = True, None, ()
                self._mc.resident, self._mc.resident_valid = None, 0
        return ok

    def mirror_state(self) -> dict:
        """What the resident mirror of foods / heads / bodies is doing for this env object right now: `state` 'off' / 'eager'
        (steps read the mirror and write the tensors) / 'lazy' (steps do not write them; they are written out when something
        looks at them), `why`, `policy` (the `resident_mirror` keyword; None = automatic), `current` (the mirror describes
        the state; False: the next step rebuilds it)."""
        c = self._mc
        on = c is not None and bool(c.resident) and self._mirror is not None
        return {'state': 'off' if not on else ('lazy' if c.resident_lazy else 'eager'), 'why': self._mirror_why,
                'policy': self._resident_policy, 'current': bool(on and c.resident_valid),
                'bytes': int(self._mirror.numel()) if on else 0}

    def _log(self, msg: str):
        if self.verbose > 0:
            print(msg)

    def _cfg(self) -> _lib.MultiConfig:
        if self._cfg_dirty:  # dynamics attributes may be changed between calls (reference tests do): _Dynamic marks it
            key = (self.boost, self.food_on_death_prob, self.boost_cost_prob, self.food_mode, self.food_rate,
                   self.reward_on_death, self.respawn_mode, self.colour_mode)
            self._cfg_cache = (key, _lib.multi_config(self.num_snakes, *key, max_food=self.max_food))
            self._cfg_dirty = False
        return self._cfg_cache[1]

    def _norm(self, name: str, shape, dtype):
        """State tensors may have been rebound by the caller: bring them to the layout the kernels read."""
        if self._pending:
            self._flush()
        raw = '_' + name in _SLOTS
        t = getattr(self, '_' + name if raw else name)  # (the raw slot: the class's own use hands out no alias)
        if tuple(t.shape) != tuple(shape):
            raise RuntimeError(f'env.{name} has shape {tuple(t.shape)}, expected {tuple(shape)}')
        if t.dtype != dtype or t.device != self.device or not t.is_contiguous():
            if dtype == torch.bool and t.dtype != torch.bool:
                t = t != 0
            t = t.to(device=self.device, dtype=dtype).contiguous()
            if raw:  # (not through the descriptor: nobody else holds the normalised copy)
                setattr(self, '_' + name, t)
                self._sync_state()
                self._state_dirty = True
                if '_' + name in _MIRRORED and self._mc is not None:
                    self._mc.resident_valid = 0
            else:
                fresh = self._last_fresh
                setattr(self, name, t)
                self._last_fresh = fresh
        return t

    def _state(self, write: bool = True):
        """the state tensors, normalised, a postponed reset applied; the caller is about to write them with another entry
        point, or (write=False) only to read them — then the mirror stays current (the step launch uses _step_state)"""
        if self._pending:
            self._flush()
        if write:
            self._touch()
        else:
            self._write_out()
        return self._step_state()

    def _step_state(self):
        N, K, S = self.num_envs, self.num_snakes, self.size
        return (self._norm('foods', (N, 1, S, S), torch.float32), self._norm('heads', (N * K, 1, S, S), torch.float32),
                self._norm('bodies', (N * K, 1, S, S), torch.float32), self._norm('dones', (N * K,), torch.bool),
                self._norm('orientations', (N * K,), torch.long),
                self._norm('agent_colours', (N * K, 3), torch.short),
                self._norm('boost_this_step', (N * K,), torch.bool))

    def _obs_args(self, mode: Optional[str]):
        if mode is None:
            return _lib.OBS_NONE, 0, None
        if mode == 'full':
            m, n = _lib.OBS_DEFAULT, 0
            shape = (self.num_snakes, self.num_envs, 3, self.size, self.size)
        elif isinstance(mode, str) and mode.startswith('partial_'):
            m, n = _lib.OBS_PARTIAL, int(mode.split('_')[1])
            shape = (self.num_snakes, self.num_envs, 3, 2 * n + 1, 2 * n + 1)
        else:
            raise ValueError('Unrecognised observation mode.')
        return m, n, torch.empty(shape, dtype=torch.float32, device=self.device)

    def _obs_dict(self, obs: torch.Tensor) -> Dict[str, torch.Tensor]:
        if self._half:
            obs = obs.to(torch.half)
        return OrderedDict([(f'agent_{i}', o) for i, o in enumerate(obs.unbind(0))])

    # ------------------------------------------------------------------ colours / rendering (host side, torch ops)

    def get_n_colours(self, n: int) -> torch.Tensor:
        """reference :163-169"""
        colours = torch.rand((n, 3), device=self.device)
        colours[:, 0] /= 1.5  # Reduce red
        colours /= colours.norm(2, dim=1, keepdim=True)
        colours *= 192
        return colours.short()

    def _get_env_images(self) -> torch.Tensor:
        """reference :194-227 — (N,3,S,S) int16 image of every env; not on the step path (used by render())."""
        N, K, S = self.num_envs, self.num_snakes, self.size
        foods, heads, bodies, dones, _, colours, boost = self._state(write=False)
        inten = bodies.gt(EPS).float() * 1 / 3 + heads.gt(EPS).float() * 1 / 3
        inten = (inten * (1 + 0.5 * boost.float())[:, None, None, None]).squeeze(1)
        img = (inten[:, None] * colours.float()[:, :, None, None]).reshape(N, K, 3, S, S).sum(dim=1).short()
        img[:, 0] += (foods.gt(EPS).squeeze(1) * 255).short()
        black = (img == 0).all(dim=1, keepdim=True)
        img = torch.where(black, torch.full_like(img, 255), img)
        img = img * (1 - self.edge_locations_mask).short()
        return img

    def render(self, mode: str = 'human', env: int = None):
        """reference :229-266 ('rgb_array' only; the 'human' viewer needs gym/pyglet)"""
        if mode == 'human':
            raise NotImplementedError("render('human') needs gym's SimpleImageViewer; use mode='rgb_array'")
        if mode != 'rgb_array':
            raise ValueError('Render mode not recognised.')
        from wurm_amd._render import frame
        return frame(self._get_env_images().cpu().numpy(), self.num_envs, self.render_args, env)

    # ------------------------------------------------------------------ observations

    def sanitize_movements(self, movements: torch.Tensor, orientations: torch.Tensor) -> torch.Tensor:
        """reference :336-339: a move straight back into the snake's own neck (movement == stored orientation) becomes the
        opposite direction.  A helper of the reference's `step` (:493) that callers can reach; here the step kernel does it
        itself (multi_snake.hip: multi_step_body), so this is the same three torch ops for whoever calls it."""
        mask = orientations == movements
        return (movements + (mask * 2).long()).fmod(4)

    def _observe(self, mode: str = None) -> Dict[str, torch.Tensor]:
        """reference :283-334"""
        if mode is None:
            mode = self.observation_mode
        m, n, obs = self._obs_args(mode)
        foods, heads, bodies, dones, _, colours, boost = self._state(write=False)
        rc = _lib.call(self.device.index, _lib.lib().wurm_multi_observe, _lib.ptr(foods), _lib.ptr(heads), _lib.ptr(bodies), _lib.ptr(dones),
                                           _lib.ptr(boost), _lib.ptr(colours), _lib.ptr(obs), m, n,
                                           _lib.i64(self.num_envs), self.num_snakes, self.size, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'MultiSnake._observe')
        return self._obs_dict(obs)

    def _observe_agent(self, agent: int) -> torch.Tensor:
        """reference :268-281"""
        return self._observe('full')[f'agent_{agent}']

    # ------------------------------------------------------------------ step

    def _ensure_call(self):
        """the argument block of the per-call launches (wurm_multi_call), filled in on first use — and with it the resident
        mirror of foods / heads / bodies, where the library offers one for this batch (or the caller asked for one)"""
        c = self._mc
        if self._mc_ready:
            return c
        self._mc_ready = True
        N, K, S, dev = self.num_envs, self.num_snakes, self.size, self.device
        c.num_envs, c.env_offset, c.seed = N, self.env_offset, _lib.u64(self.seed)
        c.num_snakes, c.size = K, S
        self._pend = torch.zeros(N, dtype=torch.uint8, device=dev)
        c.all_done_copy = self._pend.data_ptr()
        self._mc_cfg = None
        self._get_device = _lib.accessors()[0]
        ks = [str(i) for i in range(K)]
        self._keys = tuple([p + k for k in ks] for p in ('agent_', 'snake_collision_', 'edge_collision_', 'food_',
                                                         'boost_', 'size_'))
        # the resident mirror of foods / heads / bodies (large batches): the launch reads it instead of them;
        # lazy (they are not written either) as long as the caller has never got hold of one of them
        size_fn = _lib.lib().wurm_multi_resident_bytes if self._resident_policy is None else \
            _lib.lib().wurm_multi_resident_size
        nbytes = 0 if self._mirror_off else int(size_fn(_lib.i64(N), K, S))
        if not self._mirror_off:
            self._mirror_why = 'on' if nbytes > 0 else 'batch below the threshold (2^20 cells), or shape not served'
        if nbytes > 0:
            self._mirror = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            c.resident, c.resident_valid = self._mirror.data_ptr(), 0
            c.resident_lazy = int(self._lazy_mirror and not self._watched)
            # check_consistency()'s masks can come out of the step launch (the image it steps is its own): asked for
            # once the caller has called check_consistency(), dropped again if it stops doing so
            self._chk = torch.empty((2, N), dtype=torch.int32, device=dev)
        return c

    def _arm_masks(self):
        """check_consistency()'s masks from inside the step launch: asked for while the caller keeps calling that method"""
        c = self._mc
        if self._chk is None:
            return
        want = self._check_calls > 0 and self._fs.steps - self._check_step <= 64
        if want != bool(c.check_mask):
            c.check_mask = self._chk[0].data_ptr() if want else None
            c.check_mask_after = self._chk[1].data_ptr() if want else None
            self._chk_armed_at = self._fs.steps if want else 1 << 62  # (steps from here on write them)

    def _new_slab(self):
        """Fresh output tensors for the next R steps in four allocations, and the R x (8 K + 1) per-agent views and 4 dicts
        `step` returns (reference :701-729) made ONCE per slab — a view object per row was most of what a step cost on the
        host.  A slab is never written twice; it is released when the last tensor carved from it is."""
        N, K, dev, fs = self.num_envs, self.num_snakes, self.device, self._fs
        if self._slab is not None and fs.steps > 0:
            _, _ = self.rewards, self.boost_this_step  # (views of the slab that is being replaced: made while it is at hand)
        c = self._mc
        mode = self.observation_mode
        c.obs_mode, c.obs_n, o = self._obs_args(mode)
        inner = tuple(o.shape[2:])
        self._mc_mode = mode
        elems = int(torch.Size(inner).numel())
        want_after = bool(fs.want_obs_after)
        per_step = K * N * (4 * elems * (2 if want_after else 1) + 24 + 7) + N
        # 64 MB per slab at most: a caller that drops what a step returned before the next one gets the SAME block back from
        # torch's allocator, and one call's outputs (123 MB at cfg4 'full': one step per slab) then stay in the 256 MB
        # Infinity Cache; a slab of several such steps would stream to HBM (measured: 39 against 35 us per iteration at cfg4)
        R = max(1, min(64, (64 << 20) // max(per_step, 1)))
        # TWO allocations for the slab's four blocks (round 6: four torch.empty were 10 of the 28 us a step of cfg4 'full' — one
        # step per slab — cost on the host, which is what bounded that loop): one block of floats (rewards / food / sizes,
        # observations, reset observations; each part on a 256-byte boundary) and the flags on their own — their version
        # counter is what proves `dones['__all__']` unmodified, and an in-place edit of an observation must not touch it
        n_of, n_obs = R * 6 * K * N, R * K * N * elems
        n_of_pad, n_obs_pad = (n_of + 63) & ~63, (n_obs + 63) & ~63
        floats = torch.empty(n_of_pad + n_obs_pad + (n_obs if want_after else 0), dtype=torch.float32, device=dev)
        of = floats[:n_of].view(R, 6 * K, N)
        obs = floats[n_of_pad:n_of_pad + n_obs].view((R, K, N) + inner)
        after = floats[n_of_pad + n_obs_pad:].view((R, K, N) + inner) if want_after else None
        ob = torch.empty((R, 7 * K + 1, N), dtype=torch.bool, device=dev)
        sl = self._sl
        sl.out_f32, sl.out_u8, sl.obs, sl.steps, sl.obs_elems = of.data_ptr(), ob.data_ptr(), obs.data_ptr(), R, elems
        sl.obs_after = after.data_ptr() if after is not None else None
        self._slab = (of.unbind(0), ob.unbind(0))
        try:
            fs.slab_version = ob._version
        except RuntimeError:   # allocated under torch.inference_mode(): no version counters, so no deferral (eager resets)
            fs.slab_version = -1
        # what callers read every step — observations, rewards, dones — is carved now (row_views: ~0.3 us per tensor); the
        # 5 K tensors of `info` when somebody looks (_LazyInfo)
        K3, K4 = 3 * K, 4 * K
        ofs, obs_ = self._slab
        rf = _lib.row_views(of, 2, K3, K4)                      # rewards (agent-major rows 3K .. 4K)
        ob2 = ob.view(R, 7 * K + 1, N)
        rd = _lib.row_views(ob2, 2, K3, K4)                     # dones
        ra = _lib.row_views(ob2, 2, 7 * K, 7 * K + 1)           # all_done
        ov = _lib.row_views(obs, 2)
        av = _lib.row_views(after, 2) if after is not None else None
        k_agent, k_snake, k_edge, k_food, k_boost, k_size = self._keys
        info_keys = (k_snake, k_edge, k_food, k_boost, k_size)
        outs, done2s, afters = [], [], []
        for i in range(R):
            f, b = rf[i * K:(i + 1) * K], rd[i * K:(i + 1) * K] + [ra[i]]
            dones_out = dict(zip(k_agent, b[:K]))
            dones_out['__all__'] = b[K]
            info = _LazyInfo((ofs[i], obs_[i], info_keys, K))
            outs.append((OrderedDict(zip(k_agent, ov[i * K:(i + 1) * K])), dict(zip(k_agent, f)), dones_out, info))
            done2s.append(b[K])
            if av is not None:
                afters.append(OrderedDict(zip(k_agent, av[i * K:(i + 1) * K])))
        fs.outs, fs.done2s, fs.obs_afters = outs, done2s, (afters if av is not None else None)
        fs.R, fs.slot = R, 0
        fs.lazy_ok = self._lazy_ok()
        self._arm_masks()

    def step(self, actions: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], dict, dict, dict]:
        """reference :462-731.  One launch (wurm_multi_step_slot) from the step machine; the four dicts it returns were
        built when the output slab was."""
        if self._reset_obs_probe is not None or self._lazy_obs_ref is not None:
            self._reset_obs_bookkeeping()
        if self._watched and self._alias_free():
            # The caller read a state attribute earlier (experiments/multiagent.py:531 reads `env.heads`) and has let go of
            # the tensor since: nothing can edit the state behind the machine's back any more, so nothing is left to watch —
            # after one last look at the version counters (an edit made before the alias was dropped must not be forgotten)
            self._watch_ok()
            self._watched = ()
        if not (self._watched or self._lifetimes_touched or self._half):
            fs = self._fs
            out = fs.step_multi(actions)
            if out is False:
                # K int64 device vectors that are not the rows of one tensor (a policy that emits one tensor per agent,
                # experiments/multiagent.py:285-300): stacked into this object's own (K, N) block — reference :492 — and
                # launched; the stream orders the block's reuse behind the previous launch
                buf = self._abuf
                if buf is None:
                    buf = self._abuf = torch.empty((self.num_snakes, self.num_envs), dtype=torch.long, device=self.device)
                torch.stack(tuple(actions.values()), out=buf)
                out = fs.launch_multi(buf.data_ptr())
            if out.__class__ is tuple:
                self._info = None  # (reference :729 rebinds `self.info` every step: an assigned one does not outlive it)
                return out
        return self._slow_step(actions)

    def _slow_step(self, actions):
        """What the step machine cannot do by itself: the first call, a new output slab, state tensors or configuration that
        changed, tensors the caller holds (watched for in-place edits), actions that have to be stacked, `dtype=torch.half`,
        `env_lifetimes` in use — then the launch through the machine, and the error codes of the entry point."""
        if len(actions) != self.num_snakes:
            raise RuntimeError('Must have a Tensor of actions for each snake')

        for agent, act in actions.items():
            if act.dtype not in _INT_TYPES:
                raise TypeError('actions Tensor must be an integer type i.e. '
                                '{torch.ShortTensor, torch.IntTensor, torch.LongTensor}')

            if act.shape[0] != self.num_envs:
                raise RuntimeError('Must have the same number of actions as environments.')

        N, K, dev, fs = self.num_envs, self.num_snakes, self.device, self._fs
        # reference :492: stack in dict order -> (K, N); the kernel reads agent i's action of env e at [i*N + e].
        # Rows of one (K, N) int64 tensor (e.g. `tape[t, i]`) are used where they lie.
        vals = list(actions.values())
        a0 = vals[0]
        a_ptr = a0.data_ptr() if (a0.dtype is torch.long and a0.device == dev) else 0
        if a_ptr:
            row = 8 * N
            for i, v in enumerate(vals):
                if v.dtype is not torch.long or v.dim() != 1 or not v.is_contiguous() or v.data_ptr() != a_ptr + i * row:
                    a_ptr = 0
                    break
        if not a_ptr:
            vals = torch.stack([v.reshape(N) for v in vals]).to(device=dev, dtype=torch.long).contiguous()
            a_ptr = vals.data_ptr()
        # (if anything below raises, the postponed reset and the counter are still owed: the machine consumes them only
        # once the launch has been accepted)
        c = self._ensure_call()
        if self._watched:
            self._watch_ok()
        if not fs.ok:  # (first call, a state tensor rebound, the configuration or the observation mode assigned)
            pend, fs.pending = fs.pending, False  # (the raw tensors: the postponed reset belongs in front of THIS launch)
            try:
                foods, heads, bodies, dones, orientations, colours, _ = self._step_state()
            finally:
                fs.pending = pend
            c.foods, c.heads, c.bodies = foods.data_ptr(), heads.data_ptr(), bodies.data_ptr()
            c.dones, c.orientations, c.colours = dones.data_ptr(), orientations.data_ptr(), colours.data_ptr()
        if fs.slot >= fs.R or self.observation_mode != self._mc_mode or (fs.want_obs_after and fs.obs_afters is None):
            self._new_slab()
        else:
            self._arm_masks()
        cfg = self._cfg()
        if cfg is not self._mc_cfg:
            c.cfg = self._mc_cfg = cfg
        fs.ok = True
        idx = fs.dev_index
        if self._get_device() != idx:  # a process driving several GPUs has another device current
            with torch.cuda.device(idx):
                out = fs.launch_multi(a_ptr)
        else:
            out = fs.launch_multi(a_ptr)
        if out.__class__ is not tuple:
            _lib.check(int(out), 'MultiSnake.step')
            raise RuntimeError(f'MultiSnake.step: unexpected return {out!r}')
        self._info = None
        if self._lifetimes_touched or self._half:
            obs, rewards, dones_out, info = out
            dones_out = dict(dones_out)
            if self._lifetimes_touched:
                dones_out['__all__'] = dones_out['__all__'] | (self._env_lifetimes > self.max_env_lifetime)  # :703-705
                fs.last_fresh = False  # (not the kernel's own flags any more)
            if self._half:  # reference :477 / :724: observations, food and size carry the env's dtype
                info = dict(info.items())
                for k in self._keys[3] + self._keys[5]:
                    info[k] = info[k].to(torch.half)
                obs = OrderedDict((k, v.to(torch.half)) for k, v in obs.items())
            self._info = info
            out = (obs, rewards, dones_out, info)
        return out

    # ------------------------------------------------------------------ fused multi-step loop (extension)

    def rollout(self, actions: torch.Tensor, return_observations: bool = True) -> dict:
        """T iterations of `obs, r, d, info = env.step(a_t); env.reset(d['__all__'], return_observations=False)` in one
        kernel launch, with the environments resident on chip (the loop of experiments/speeds.py:30-37).

        actions: (T, num_snakes, num_envs) int64 on the device (actions[t, i] = agent_i's actions of step t).
        Returns a dict of tensors with a leading T dimension, per-agent quantities as (T, num_snakes, num_envs):
        `observations` (T, K, N, 3, h, w), `rewards`, `dones`, `boost`, `snake_collision`, `edge_collision`, `food`,
        `size`, and `all_done` (T, N).  Bit-identical to the Python loop.
        """
        N, K, S, dev = self.num_envs, self.num_snakes, self.size, self.device
        if actions.dtype != torch.long:
            raise TypeError('rollout actions must be a LongTensor of shape (T, num_snakes, num_envs)')
        if actions.dim() != 3 or actions.shape[1] != K or actions.shape[2] != N:
            raise RuntimeError('rollout actions must have shape (T, num_snakes, num_envs)')
        if not actions.is_contiguous() or actions.device != dev:
            raise RuntimeError('rollout actions must be a contiguous device tensor')
        T = actions.shape[0]
        # The resident mirror (large batches): the launch reads the compact image instead of the fp32 tensors when it
        # describes them, and keeps it current — in the lazy form without writing the tensors (wurm_multi_rollout_resident;
        # the library itself falls back to the tensors, writing a lazy mirror out first, for the shapes its mirror-keeping
        # kernel does not serve).  Same protocol as step(): a postponed reset is applied first, watched tensors are checked
        # for in-place edits, nothing is "touched".
        if self._pending:
            self._flush()
        c = self._ensure_call() if T > 0 else self._mc
        if c is not None and c.resident and self._watched:
            self._watch_ok()
        mirrored = c is not None and bool(c.resident)
        if mirrored:
            foods, heads, bodies, dones, orientations, colours, boost = self._step_state()
            # (the argument block names the tensors for wurm_multi_resident_flush — a look at the state right after this
            # rollout writes the mirror out through it, whether or not a step() has ever filled it in)
            c.foods, c.heads, c.bodies = foods.data_ptr(), heads.data_ptr(), bodies.data_ptr()
            c.dones, c.orientations, c.colours = dones.data_ptr(), orientations.data_ptr(), colours.data_ptr()
            self._state_dirty = False
            self._chk_fresh = False  # (the masks of the last step launch describe an older state)
        else:
            foods, heads, bodies, dones, orientations, colours, boost = self._state()
        m, n, obs1 = self._obs_args(self.observation_mode if return_observations else None)
        obs = torch.empty((T,) + tuple(obs1.shape), dtype=torch.float32, device=dev) if obs1 is not None else None
        out_f = torch.empty((T, 3, K, N), dtype=torch.float32, device=dev)
        out_b = torch.empty((T, 4, K, N), dtype=torch.bool, device=dev)
        all_done = torch.empty((T, N), dtype=torch.bool, device=dev)
        cfg = self._cfg()
        if mirrored:
            rc = _lib.call(self.device.index, _lib.lib().wurm_multi_rollout_resident,
                _lib.ptr(foods), _lib.ptr(heads), _lib.ptr(bodies), _lib.ptr(dones), _lib.ptr(orientations),
                _lib.ptr(colours), _lib.ptr(boost), _lib.ptr(actions), _lib.ptr(out_f), _lib.ptr(out_b), _lib.ptr(all_done),
                _lib.ptr(obs), m, n, _lib.i64(N), K, S, _lib.i64(T), ctypes.byref(cfg), _lib.u64(self.seed),
                _lib.u64(self._next_call(2 * T)), _lib.i64(self.env_offset), c.resident,
                self._mc_addr + _lib.MultiCall.resident_valid.offset, int(c.resident_lazy),
                _lib.stream_ptr(self.device.index))
        else:
            rc = _lib.call(self.device.index, _lib.lib().wurm_multi_rollout,
                _lib.ptr(foods), _lib.ptr(heads), _lib.ptr(bodies), _lib.ptr(dones), _lib.ptr(orientations),
                _lib.ptr(colours), _lib.ptr(boost), _lib.ptr(actions), _lib.ptr(out_f), _lib.ptr(out_b), _lib.ptr(all_done),
                _lib.ptr(obs), m, n, _lib.i64(N), K, S, _lib.i64(T), ctypes.byref(cfg), _lib.u64(self.seed),
                _lib.u64(self._next_call(2 * T)), _lib.i64(self.env_offset), None, None, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'MultiSnake.rollout')
        if T > 0:
            # (reference :105: env-major; the last step's rewards are agent-major here — transposed when somebody asks for
            # the attribute, not by a kernel between two rollout launches)
            self._rewards_src, self._rewards_t, self._rewards_at = out_f[-1, 0], None, self._fs.steps
        if self._lifetimes_touched:  # (else it is all zeros already: env_lifetimes' docstring)
            self._env_lifetimes.zero_()
        if self._half:
            obs = obs.to(torch.half) if obs is not None else None
            out_h = out_f[:, 1:].to(torch.half)
            food, size = out_h[:, 0], out_h[:, 1]
        else:
            food, size = out_f[:, 1], out_f[:, 2]
        return {'observations': obs, 'rewards': out_f[:, 0], 'food': food, 'size': size,
                'dones': out_b[:, 0], 'boost': out_b[:, 1], 'snake_collision': out_b[:, 2],
                'edge_collision': out_b[:, 3], 'all_done': all_done}

    # ------------------------------------------------------------------ invariants

    def check_consistency(self):
        """reference :733-769: raises RuntimeError if any env is inconsistent.  Right after a step (and the deferred reset
        of its own `dones['__all__']`) on the resident mirror the masks come from that step's launch; anything else — the
        state read from the fp32 tensors, looked at or edited since — runs the checker over the tensors."""
        from wurm_amd.utils import _raise_for, _or_reduce
        self._check_calls += 1
        self._check_step = self._steps
        err = self._step_check_mask()
        if err is not None:
            if not bool(err.any()):
                return
            if bool((err == -1).any()):  # an env whose mask the launch could not vouch for
                err = None
        if err is None:
            foods, heads, bodies, dones, _, _, _ = self._state(write=False)
            err = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
            rc = _lib.call(self.device.index, _lib.lib().wurm_multi_check, _lib.ptr(foods), _lib.ptr(heads), _lib.ptr(bodies),
                           _lib.ptr(dones), _lib.ptr(err), _lib.i64(self.num_envs), self.num_snakes, self.size,
                           _lib.stream_ptr(self.device.index))
            _lib.check(rc, 'MultiSnake.check_consistency')
        mask = _or_reduce(err, 10)
        _raise_for(mask & 0x7f, one_food=False)
        if mask & 0x100:
            raise RuntimeError('An environment contains overlapping snakes')
        if mask & 0x200:
            raise RuntimeError('Dead snake contains non-zero elements.')

    def _step_check_mask(self):
        """(N) int32 masks of the state as it is now, computed inside the last step's launch (wurm_multi_call.check_mask /
        check_mask_after), or None if they do not apply; -1 marks an env the launch could not vouch for"""
        c = self._mc
        self._arm_masks()   # (from the next step on, if they are not being written yet)
        if self._watched and c is not None and not self._watch_ok():
            return None                   # an alias the caller holds was edited in place since the launch (_touch() ran)
        if not self._chk_fresh or c is None or not c.resident or not c.resident_valid:
            return None
        if not self._pending:
            return self._chk[0]
        if self._chk_has_after:       # the postponed reset of the step's own mask: what obs_after observed
            return self._chk[1]
        # postponed, but the launch did not build that state: envs the reset rebuilds are not vouched for
        return self._chk[0].masked_fill(self._pend != 0, -1)

    # ------------------------------------------------------------------ reset

    def _launch_reset(self, done: torch.Tensor, obs, m, n, call, status):
        foods, heads, bodies, dones, orientations, colours, boost = self._state()
        rc = _lib.call(self.device.index, _lib.lib().wurm_multi_reset,
            _lib.ptr(foods), _lib.ptr(heads), _lib.ptr(bodies), _lib.ptr(dones), _lib.ptr(orientations),
            _lib.ptr(colours), _lib.ptr(done), _lib.ptr(status), _lib.ptr(boost), _lib.ptr(obs), m, n,
            _lib.i64(self.num_envs), self.num_snakes, self.size, ctypes.byref(self._cfg()), _lib.u64(self.seed),
            _lib.u64(call), _lib.i64(self.env_offset), None, _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'MultiSnake.reset')

    def _reset_kernel(self, done: torch.Tensor, observe: bool, want_status: bool = False):
        if self._pending:
            self._flush()
        m, n, obs = self._obs_args(self.observation_mode if observe else None)
        status = torch.zeros(1, dtype=torch.int32, device=self.device) if want_status else None
        self._launch_reset(done, obs, m, n, self._next_call(), status)
        if want_status:
            return int(status.item())
        return obs

    def reset(self, done: torch.Tensor = None, return_observations: bool = True) -> Optional[Dict[str, torch.Tensor]]:
        """Resets environments in which the snake has died (reference :771-836)

        Args:
            done: A 1D Tensor of length self.num_envs. A value of 1 means the corresponding environment needs to be
                reset.  None: every env whose snakes are all dead.
            return_observations: extension — pass False to skip the K observations the reference's callers discard.

        Called with the very `dones['__all__']` the last `step` returned, nothing in between and
        `return_observations=False`, the reset is postponed into the next step's launch (module docstring).
        """
        if self.initial_snake_length != 3:
            raise NotImplementedError('Only initial snake length = 3 has been implemented.')
        if done is not None:
            fs = self._fs
            had_after = fs.obs_after is not None
            # (the machine checks that `done` is the very tensor the last step returned, unmodified, that nothing has consumed a
            # counter since, and that nobody else holds a state tensor: _alias_free / wurm_torch_alias_free)
            if fs.last_fresh:
                if return_observations and self._lazy_obs_mode:
                    # the caller has been discarding what reset returns: postponed without the precomputed observation; the
                    # dict fills itself if it is looked at after all (_LazyResetObs)
                    if fs.reset_lazy(done, False) is None:
                        self._chk_has_after = False
                        lz = _LazyResetObs(self, self.observation_mode)
                        self._lazy_obs_ref = weakref.ref(lz)
                        return lz
                else:
                    obs = fs.reset_lazy(done, return_observations)
                    if obs is not NotImplemented:
                        # env_lifetimes is all zeros here (nobody has asked for it): `env_lifetimes[done] = 0` (:797) is a no-op
                        self._chk_has_after = had_after  # the launch also left the masks of the state this reset produces
                        if obs is not None:
                            self._reset_obs_probe = obs
                            if self._half:
                                obs = OrderedDict((k, v.to(torch.half)) for k, v in obs.items())
                        return obs
        if done is None:
            done = self._norm('dones', (self.num_envs * self.num_snakes,), torch.bool) \
                .view(self.num_envs, self.num_snakes).all(dim=1)
        done = done.view((done.shape[0]))
        if done.dtype != torch.bool:
            done = done != 0
        done = done.to(self.device).contiguous()
        self._env_lifetimes.masked_fill_(done, 0)  # :797
        obs = self._reset_kernel(done, observe=return_observations)
        if return_observations:
            return self._obs_dict(obs)

    def _create_envs(self, num_envs: int):
        """reference :996-1019 — a fresh batch: ((foods, heads, bodies), orientations); self is not modified."""
        if self.initial_snake_length != 3:
            raise NotImplementedError('Only initial snake length = 3 has been implemented.')
        K, S, dev = self.num_snakes, self.size, self.device
        foods = torch.zeros((num_envs, 1, S, S), device=dev)
        heads = torch.zeros((num_envs * K, 1, S, S), device=dev)
        bodies = torch.zeros((num_envs * K, 1, S, S), device=dev)
        dones = torch.zeros(num_envs * K, dtype=torch.bool, device=dev)
        orientations = torch.zeros(num_envs * K, dtype=torch.long, device=dev)
        colours = torch.zeros((num_envs * K, 3), dtype=torch.short, device=dev)
        ones = torch.ones(num_envs, dtype=torch.bool, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        cfg = _lib.MultiConfig.from_buffer_copy(self._cfg())  # a copy: the cached block stays as it is
        cfg.respawn_any = 0
        rc = _lib.call(self.device.index, _lib.lib().wurm_multi_reset, 
            _lib.ptr(foods), _lib.ptr(heads), _lib.ptr(bodies), _lib.ptr(dones), _lib.ptr(orientations),
            _lib.ptr(colours), _lib.ptr(ones), _lib.ptr(status), None, None, _lib.OBS_NONE, 0, _lib.i64(num_envs), K,
            S, ctypes.byref(cfg), _lib.u64(self.seed), _lib.u64(self._next_call()), _lib.i64(self.env_offset), None,
            _lib.stream_ptr(self.device.index))
        _lib.check(rc, 'MultiSnake._create_envs')
        if int(status.item()):
            raise RuntimeError('There is no available locations to create snake!')
        return (foods, heads, bodies), orientations
