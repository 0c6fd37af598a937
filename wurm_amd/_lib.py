"""Loader of libwurm_hip.so (the gfx950 kernels behind a C ABI, include/wurm_hip.h).

There is deliberately NO fallback: if the library is missing, or a call is made without a HIP device, the
product raises.  The CPU oracle under oracle/ is test infrastructure and is never used from here.
"""
import contextlib
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# WURM_HIP_LIBRARY: another build of the same library (the instrumented one of `make -C wurm_amd/csrc timeline`)
LIB_PATH = os.environ.get('WURM_HIP_LIBRARY') or os.path.join(_HERE, 'libwurm_hip.so')
CSRC = os.path.join(_HERE, 'csrc')

OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_DTYPE = 0, -1, -2, -3, -4

OBS_DEFAULT, OBS_RAW, OBS_ONE_CHANNEL, OBS_POSITIONS, OBS_PARTIAL, OBS_NONE = range(6)
ACT_I64, ACT_I32 = 0, 1

# every symbol include/wurm_hip.h declares (tests/test_abi.py checks the header against this list and the .so)
SYMBOLS = [
    'wurm_version', 'wurm_set_option', 'wurm_get_option', 'wurm_reset_option', 'wurm_launch_count', 'wurm_single_last_route', 'wurm_single_obs_elems', 'wurm_grid_obs_elems',
    'wurm_single_step', 'wurm_single_reset', 'wurm_single_observe', 'wurm_single_rollout', 'wurm_single_rollout_resident', 'wurm_single_check',
    'wurm_single_step_reset', 'wurm_single_resident_bytes', 'wurm_single_resident_size', 'wurm_single_resident_flush', 'wurm_grid_step_reset', 'wurm_grid_resident_bytes', 'wurm_grid_resident_size', 'wurm_grid_resident_flush', 'wurm_single_step_slot', 'wurm_grid_step_slot',
    'wurm_grid_step', 'wurm_grid_reset', 'wurm_grid_observe', 'wurm_grid_rollout', 'wurm_grid_rollout_resident',
    'wurm_multi_obs_elems', 'wurm_multi_step', 'wurm_multi_step_reset', 'wurm_multi_step_packed', 'wurm_multi_step_slot', 'wurm_multi_resident_bytes', 'wurm_multi_resident_size', 'wurm_multi_resident_flush', 'wurm_multi_reset', 'wurm_multi_observe', 'wurm_multi_check',
    'wurm_multi_rollout', 'wurm_multi_rollout_resident',
    'wurm_multi_colours', 'wurm_orientations',
    'wurm_a2c_returns', 'wurm_a2c_returns_backward', 'wurm_single_stats', 'wurm_single_policy_rollout',
]


class SingleCall(ctypes.Structure):
    """wurm_single_call of include/wurm_hip.h (host struct of device pointers and sizes)"""
    _fields_ = [('envs', ctypes.c_void_p), ('actions', ctypes.c_void_p), ('reward', ctypes.c_void_p),
                ('done', ctypes.c_void_p), ('self_collision', ctypes.c_void_p), ('edge_collision', ctypes.c_void_p),
                ('obs', ctypes.c_void_p), ('obs_after', ctypes.c_void_p), ('done_copy', ctypes.c_void_p),
                ('pre_done', ctypes.c_void_p), ('inject_food', ctypes.c_void_p), ('inject_reset', ctypes.c_void_p),
                ('inject_pre_reset', ctypes.c_void_p), ('num_envs', ctypes.c_int64), ('env_offset', ctypes.c_int64),
                ('seed', ctypes.c_uint64), ('call', ctypes.c_uint64), ('pre_call', ctypes.c_uint64),
                ('actions_dtype', ctypes.c_int), ('obs_mode', ctypes.c_int), ('obs_n', ctypes.c_int),
                ('size', ctypes.c_int), ('post_reset', ctypes.c_int), ('start_y', ctypes.c_int),
                ('start_x', ctypes.c_int), ('resident', ctypes.c_void_p), ('resident_valid', ctypes.c_int),
                ('resident_lazy', ctypes.c_int), ('check_mask', ctypes.c_void_p)]


class SingleSlabs(ctypes.Structure):
    """wurm_single_slabs of include/wurm_hip.h"""
    _fields_ = [('obs', ctypes.c_void_p), ('obs_after', ctypes.c_void_p), ('reward', ctypes.c_void_p),
                ('flags', ctypes.c_void_p), ('steps', ctypes.c_int64)]


class MultiConfig(ctypes.Structure):
    """wurm_multi_config of include/wurm_hip.h (host struct)"""
    _fields_ = [('boost', ctypes.c_int), ('food_on_death', ctypes.c_int), ('death_threshold', ctypes.c_float),
                ('boost_cost_prob', ctypes.c_float), ('food_mode', ctypes.c_int), ('food_rate', ctypes.c_float),
                ('max_food', ctypes.c_int), ('reward_on_death', ctypes.c_float), ('respawn_any', ctypes.c_int),
                ('colour_random', ctypes.c_int)]


class MultiInject(ctypes.Structure):
    """wurm_multi_inject: device pointers"""
    _fields_ = [('death_a', ctypes.c_void_p), ('cost', ctypes.c_void_p), ('death_b', ctypes.c_void_p),
                ('rate', ctypes.c_void_p), ('food_cell', ctypes.c_void_p)]


class MultiResetInject(ctypes.Structure):
    """wurm_multi_reset_inject: device pointers"""
    _fields_ = [('create', ctypes.c_void_p), ('create_food', ctypes.c_void_p), ('colours', ctypes.c_void_p),
                ('respawn', ctypes.c_void_p)]


class MultiCall(ctypes.Structure):
    """wurm_multi_call of include/wurm_hip.h"""
    _fields_ = [(n, ctypes.c_void_p) for n in (
        'foods', 'heads', 'bodies', 'dones', 'orientations', 'colours', 'actions', 'boost_this_step', 'rewards',
        'snake_collision', 'edge_collision', 'food_consumed', 'sizes', 'all_done', 'all_done_copy', 'obs', 'obs_after',
        'agent_major_f32', 'agent_major_u8', 'pre_done', 'inject', 'pre_inject')] + [
        ('num_envs', ctypes.c_int64), ('env_offset', ctypes.c_int64), ('seed', ctypes.c_uint64),
        ('call', ctypes.c_uint64), ('pre_call', ctypes.c_uint64), ('num_snakes', ctypes.c_int), ('size', ctypes.c_int),
        ('obs_mode', ctypes.c_int), ('obs_n', ctypes.c_int), ('cfg', MultiConfig), ('resident', ctypes.c_void_p),
        ('resident_valid', ctypes.c_int), ('resident_lazy', ctypes.c_int), ('check_mask', ctypes.c_void_p),
        ('check_mask_after', ctypes.c_void_p)]


class MultiSlabs(ctypes.Structure):
    """wurm_multi_slabs of include/wurm_hip.h"""
    _fields_ = [('out_f32', ctypes.c_void_p), ('out_u8', ctypes.c_void_p), ('obs', ctypes.c_void_p),
                ('obs_after', ctypes.c_void_p), ('steps', ctypes.c_int64), ('obs_elems', ctypes.c_int64)]


def multi_config(num_snakes, boost, food_on_death_prob, boost_cost_prob, food_mode, food_rate, reward_on_death,
                 respawn_mode, colour_mode, max_food=None):
    import numpy as np
    if food_mode not in ('only_one', 'random_rate'):
        raise ValueError('food_mechanics not recognised')
    return MultiConfig(int(bool(boost)), int(food_on_death_prob > 0), float(np.float32(1 - food_on_death_prob)),
                       float(np.float32(boost_cost_prob)), 0 if food_mode == 'only_one' else 1,
                       float(np.float32(food_rate)), 8 * num_snakes if max_food is None else int(max_food),
                       float(np.float32(reward_on_death)),
                       int(respawn_mode == 'any'), int(colour_mode == 'random'))


class WurmHipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compiles wurm_amd/csrc/*.hip for gfx950 into wurm_amd/libwurm_hip.so (hipcc cross-compiles without a GPU)."""
    import sys
    cmd = ['make', '-j4', '-C', CSRC, 'PYTHON=' + sys.executable] + (['-B'] if force else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise WurmHipError('building libwurm_hip.so failed:\n' + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WurmHipError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or `make -C wurm_amd/csrc`). There is no CPU fallback.')
        # torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's): import torch FIRST so that this
        # library binds to the HIP runtime that owns torch's device memory and streams, whatever the load order.
        import torch  # noqa: F401
        l = ctypes.CDLL(LIB_PATH)
        _declare_prototypes(l)
        _lib = l
    return _lib


_CTYPES = {'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'uint64_t': ctypes.c_uint64, 'float': ctypes.c_float}


def _declare_prototypes(l):
    """argtypes / restype of every entry point, read from include/wurm_hip.h itself (pointers of any kind ->
    c_void_p).  With them ctypes converts plain Python ints / None per call instead of wrapper objects built in Python
    (~4 us per call on the per-step path), and a call with the wrong number of arguments fails loudly."""
    import re
    # the repo's include/wurm_hip.h, or the copy `make` places next to the library (wurm_amd/ installed on its own)
    header = os.path.join(os.path.dirname(_HERE), 'include', 'wurm_hip.h')
    if not os.path.exists(header):
        header = os.path.join(_HERE, 'wurm_hip.h')
    text = re.sub(r'/\*.*?\*/', ' ', open(header).read(), flags=re.S)
    declared = set()
    for ret, name, params in re.findall(r'\b(const char \*|int64_t|int)\s*(wurm_[a-z_0-9]+)\s*\(([^)]*)\)\s*;', text):
        declared.add(name)
        fn = getattr(l, name)
        fn.restype = {'const char *': ctypes.c_char_p, 'int64_t': ctypes.c_int64, 'int': ctypes.c_int}[ret]
        args = [a.strip() for a in params.split(',')]
        if args == ['void']:
            args = []
        fn.argtypes = [ctypes.c_void_p if '*' in a else _CTYPES[a.replace('const ', '').split()[0]] for a in args]
    missing = [n for n in SYMBOLS if n not in declared]
    if missing:  # a symbol without argtypes would get ctypes' default int conversion for 64-bit arguments
        raise WurmHipError(f'{header}: no prototype found for {missing}')


_step_slot = {}


class _SlotFn(object):
    """wurm_single_step_slot / wurm_grid_step_slot as a Python callable (through the CPython shim wurm_amd/_fastcall when
    it is built, else ctypes) that also carries the C address of the entry point for wurm_amd._fastcall.Stepper."""

    def __init__(self, cfn, shim=True):
        self.c_address = ctypes.cast(cfn, ctypes.c_void_p).value
        self._call = cfn
        if shim:  # (wurm_multi_step_slot has another signature: the ctypes function itself)
            try:
                import functools
                from wurm_amd import _fastcall
                self._call = functools.partial(_fastcall.step_slot, self.c_address)
            except ImportError:
                pass

    def __call__(self, *args):
        return self._call(*args)


def step_slot_fn(name: str = 'wurm_single_step_slot'):
    """The per-step entry point `name` of libwurm_hip.so (see _SlotFn) — same arguments, same library, same kernels
    whichever way it ends up being called."""
    fn = _step_slot.get(name)
    if fn is None:
        fn = _step_slot[name] = _SlotFn(getattr(lib(), name), shim=name != 'wurm_multi_step_slot')
    return fn


def multi_step_fn():
    """wurm_multi_step_packed through the CPython shim when it is built, else the ctypes function (same arguments)."""
    fn = _step_slot.get('wurm_multi_step_packed')
    if fn is None:
        cfn = lib().wurm_multi_step_packed
        try:
            import functools
            from wurm_amd import _fastcall
            fn = functools.partial(_fastcall.multi_step, ctypes.cast(cfn, ctypes.c_void_p).value)
        except (ImportError, AttributeError):
            fn = cfn
        _step_slot['wurm_multi_step_packed'] = fn
    return fn


def set_option(name: str, value):
    """Sets (value=None: back to its default) one of the library's knobs (include/wurm_hip.h: wurm_set_option); returns the
    previous value.  The environment variable of the same name is read once, when the library is loaded."""
    l = lib()
    old = l.wurm_get_option(name.encode())
    rc = l.wurm_reset_option(name.encode()) if value is None else l.wurm_set_option(name.encode(), int(value))
    if rc != OK:
        raise ValueError(f'unknown option {name}')
    return old


@contextlib.contextmanager
def knobs(**kw):
    """`with knobs(WURM_RESIDENT_MIN_ENVS=0): ...` — options set for the duration (None: the default), and the environment
    variables of the same names with them, so that child processes started inside see the same values."""
    old_opt = {k: set_option(k, v) for k, v in kw.items()}
    old_env = {k: os.environ.get(k) for k in kw}
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    try:
        yield
    finally:
        for k, v in old_opt.items():
            set_option(k, v)
        for k, v in old_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def get_option(name: str) -> int:
    return lib().wurm_get_option(name.encode())


def check(rc: int, what: str):
    """Maps C-ABI error codes onto the exception types the reference raises for the same conditions."""
    if rc == OK:
        return
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(f'{what}: configuration not supported')
    if rc == ERR_DTYPE:
        raise TypeError(f'{what}: actions Tensor must be an integer type')
    if rc == ERR_HIP:
        raise WurmHipError(f'{what}: HIP kernel launch failed')
    raise RuntimeError(f'{what}: invalid argument (code {rc})')


def parse_obs_mode(mode):
    if mode is None or mode == 'none':
        return OBS_NONE, 0
    if mode.startswith('partial_'):
        return OBS_PARTIAL, int(mode.split('_')[-1])
    table = {'default': OBS_DEFAULT, 'raw': OBS_RAW, 'one_channel': OBS_ONE_CHANNEL, 'positions': OBS_POSITIONS,
             'full': OBS_DEFAULT}
    if mode not in table:
        raise ValueError(f'Unrecognised observation mode: {mode}')
    return table[mode], 0


def require_device(device):
    """Resolves `device` to a torch.device and insists that it is a HIP GPU."""
    import torch
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise WurmHipError(f"wurm_amd runs on MI355X only: device must be 'cuda[:i]', got '{device}'. "
                           f'There is no CPU fallback (the CPU oracle under oracle/ is test infrastructure).')
    if not torch.cuda.is_available():
        raise WurmHipError('wurm_amd: no HIP device visible (torch.cuda.is_available() is False)')
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    lib()
    return dev


def ptr(t):
    """device address of a tensor (or NULL) as ctypes converts it for a pointer parameter"""
    return t.data_ptr() if t is not None else None


_get_raw_stream = None
_get_device = None


def _torch_accessors():
    global _get_raw_stream, _get_device
    import torch
    _get_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    _get_device = getattr(torch._C, '_cuda_getDevice', None)
    if _get_raw_stream is None:  # public (slower) route if the private accessors ever go away
        _get_raw_stream = lambda index: torch.cuda.current_stream(index).cuda_stream
    if _get_device is None:
        _get_device = torch.cuda.current_device


_torchinfo = None


def _load_torchinfo():
    """(library, addresses of its three per-step helpers) of wurm_amd/libwurm_torchinfo.so, or False if that optional
    helper is not built / switched off (WURM_TORCHINFO=0)"""
    global _torchinfo
    if _torchinfo is None:
        _torchinfo = False
        path = os.path.join(_HERE, 'libwurm_torchinfo.so')
        if os.path.exists(path) and os.environ.get('WURM_TORCHINFO', '1') != '0':
            try:
                import torch  # noqa: F401  (its libraries first: the helper links against them)
                l = ctypes.PyDLL(path)
                _torchinfo = (l, tuple(ctypes.cast(getattr(l, n), ctypes.c_void_p).value for n in
                                       ('wurm_torch_tensor_info', 'wurm_torch_raw_stream', 'wurm_torch_current_device',
                                        'wurm_torch_alias_free')))
            except (OSError, AttributeError):
                _torchinfo = False
    return _torchinfo


def torch_helpers():
    """(tensor_info, raw_stream, current_device, alias_free) addresses of wurm_amd/libwurm_torchinfo.so (csrc/torchinfo.cpp: facts about
    a tensor and torch's current device / stream straight from ATen), or None if that optional helper is not built."""
    t = _load_torchinfo()
    return t[1] if t else None


_row_views = None


def row_views(t, lead: int, lo: int = 0, hi: int = -1):
    """The leading `lead` dimensions of the contiguous tensor `t` unbound at once: a list of prod(t.shape[:lead]) views of
    shape t.shape[lead:] that share t's storage and version counter — through wurm_amd/libwurm_torchinfo.so
    (csrc/torchinfo.cpp: wurm_torch_row_views, ~0.3 us per view) where it is built, else `unbind` (~0.65 us per view).
    lead == 2 with lo / hi: only the rows lo <= j < hi of the second dimension (for each index of the first)."""
    global _row_views
    if _row_views is None:
        _row_views = False
        info = _load_torchinfo()
        if info:
            try:
                fn = info[0].wurm_torch_row_views
                fn.restype, fn.argtypes = ctypes.py_object, [ctypes.py_object] + [ctypes.c_longlong] * 3
                _row_views = fn
            except AttributeError:
                pass
    if _row_views:
        return _row_views(t, lead, lo, hi)
    if lead == 2 and hi >= 0:
        return [v for part in t.unbind(0) for v in part[lo:hi].unbind(0)]
    return list(t.reshape((-1,) + tuple(t.shape[lead:])).unbind(0))


def accessors():
    """(current-device getter, raw-stream getter) for callers that inline `call` / `stream_ptr` on a hot path"""
    if _get_raw_stream is None:
        _torch_accessors()
    return _get_device, _get_raw_stream


def stream_ptr(device_index=None):
    """torch's current HIP stream on `device_index` as a raw hipStream_t.  Uses the C accessor directly:
    torch.cuda.current_stream() costs ~8 us per call (device resolution, availability checks)."""
    if _get_raw_stream is None:
        _torch_accessors()
    if device_index is None:
        device_index = _get_device()
    return _get_raw_stream(device_index)


def call(device_index, fn, *args):
    """Calls a C-ABI entry point with `device_index` as the current HIP device (kernels launch on the current device;
    the stream handed over belongs to `device_index`).  The check is one cheap C call; the guard is only entered when
    a process that drives several GPUs has another device current."""
    if _get_device is None:
        _torch_accessors()
    if _get_device() != device_index:
        import torch
        with torch.cuda.device(device_index):
            return fn(*args)
    return fn(*args)


def u64(x):
    """a Python int as the unsigned 64-bit argument it stands for (prototypes are declared: ctypes converts)"""
    return int(x) & 0xFFFFFFFFFFFFFFFF


def i64(x):
    return int(x)
