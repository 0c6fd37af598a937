"""ctypes front-end of the CPU oracle (oracle/*.c -> oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline
leg as the checker / reported baseline.  The product package (wurm_amd/) never imports this module.

All functions take and return numpy arrays (state arrays are modified in place, like the reference's
tensors).  Argument meaning mirrors the C functions, which cite the reference lines they restate.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle.so')

OBS_DEFAULT, OBS_RAW, OBS_ONE_CHANNEL, OBS_POSITIONS, OBS_PARTIAL, OBS_NONE = range(6)
ACT_I64, ACT_I32 = 0, 1

# bits of the per-env error mask returned by single_check (wurm/utils.py:113-178 in the reference)
CHK_FOOD_VALUE, CHK_ONE_HEAD, CHK_HAS_SNAKE, CHK_HEAD_AT_END, CHK_BODY_RANGE, CHK_MIN_LENGTH, CHK_HEAD_ON_FOOD, \
    CHK_ONE_FOOD = (1 << i for i in range(8))


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(('.c', '.h'))]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.run(['make', '-C', _HERE] + (['-B'] if force else []), check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_single_obs_elems.restype = ctypes.c_int64
        _lib.oracle_grid_obs_elems.restype = ctypes.c_int64
        if hasattr(_lib, 'oracle_multi_obs_elems'):
            _lib.oracle_multi_obs_elems.restype = ctypes.c_int64
    return _lib


def parse_obs_mode(mode: str):
    """'partial_2' -> (OBS_PARTIAL, 2) etc."""
    if mode is None or mode == 'none':
        return OBS_NONE, 0
    if mode.startswith('partial_'):
        return OBS_PARTIAL, int(mode.split('_')[-1])
    return {'default': OBS_DEFAULT, 'raw': OBS_RAW, 'one_channel': OBS_ONE_CHANNEL, 'positions': OBS_POSITIONS,
            'full': OBS_DEFAULT}[mode], 0


def _p(a):
    if a is None:
        return None
    assert a.flags['C_CONTIGUOUS'], 'oracle arrays must be C-contiguous'
    return a.ctypes.data_as(ctypes.c_void_p)


def _act_dtype(a):
    if a.dtype == np.int64:
        return ACT_I64
    if a.dtype == np.int32:
        return ACT_I32
    raise TypeError('actions must be int64 or int32')


def _check(rc):
    if rc == -2:
        raise NotImplementedError('oracle: unsupported configuration')
    if rc != 0:
        raise RuntimeError(f'oracle error {rc}')


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


_u64 = ctypes.c_uint64
_i64 = ctypes.c_int64


# ---------------------------------------------------------------- SingleSnake

def single_obs_shape(mode: str, N: int, S: int):
    m, n = parse_obs_mode(mode)
    if m in (OBS_DEFAULT, OBS_RAW):
        return (N, 3, S, S)
    if m == OBS_ONE_CHANNEL:
        return (N, 1, S, S)
    if m == OBS_POSITIONS:
        return (N, 4)
    if m == OBS_PARTIAL:
        return (N, 3 * (2 * n + 1) ** 2)
    return None


def single_observe(envs, mode: str):
    N, _, S, _ = envs.shape
    m, n = parse_obs_mode(mode)
    obs = np.empty(single_obs_shape(mode, N, S), np.float32)
    _check(lib().oracle_single_observe(_p(envs), _p(obs), m, n, _i64(N), S))
    return obs


def single_step(envs, actions, mode='default', seed=0, call=0, env_offset=0, inject_food=None):
    """Returns (obs, reward (N,), done (N,) u8, self_collision, edge_collision); envs and actions in place."""
    N, _, S, _ = envs.shape
    m, n = parse_obs_mode(mode)
    shape = single_obs_shape(mode, N, S)
    obs = np.empty(shape, np.float32) if shape else None
    reward = np.empty(N, np.float32)
    done, sc, ec = (np.empty(N, np.uint8) for _ in range(3))
    inj = _i32(inject_food)
    _check(lib().oracle_single_step(_p(envs), _p(actions), _act_dtype(actions), _p(reward), _p(done), _p(sc), _p(ec),
                                    _p(obs), m, n, _i64(N), S, _u64(seed), _u64(call), _i64(env_offset), _p(inj)))
    return obs, reward, done, sc, ec


def single_reset(envs, done, mode='default', seed=0, call=0, env_offset=0, inject_reset=None):
    N, _, S, _ = envs.shape
    m, n = parse_obs_mode(mode)
    shape = single_obs_shape(mode, N, S)
    obs = np.empty(shape, np.float32) if shape else None
    d = np.ascontiguousarray(np.asarray(done).reshape(N) != 0, dtype=np.uint8)
    inj = _i32(inject_reset)
    _check(lib().oracle_single_reset(_p(envs), _p(d), _p(obs), m, n, _i64(N), S, _u64(seed), _u64(call),
                                     _i64(env_offset), _p(inj)))
    return obs


def single_rollout(envs, actions, mode='default', seed=0, call0=0, env_offset=0, inject_food=None,
                   inject_reset=None):
    """actions (T,N) sanitised in place. Returns dict of (T,N,...) arrays."""
    N, _, S, _ = envs.shape
    T = actions.shape[0]
    m, n = parse_obs_mode(mode)
    shape = single_obs_shape(mode, N, S)
    obs = np.empty((T,) + shape, np.float32) if shape else None
    reward = np.empty((T, N), np.float32)
    done, sc, ec = (np.empty((T, N), np.uint8) for _ in range(3))
    inj_f, inj_r = _i32(inject_food), _i32(inject_reset)
    _check(lib().oracle_single_rollout(_p(envs), _p(actions), _act_dtype(actions), _p(reward), _p(done), _p(sc),
                                       _p(ec), _p(obs), m, n, _i64(N), S, _i64(T), _u64(seed), _u64(call0),
                                       _i64(env_offset), _p(inj_f), _p(inj_r)))
    return dict(obs=obs, reward=reward, done=done, self_collision=sc, edge_collision=ec)


def single_check(envs):
    N, _, S, _ = envs.shape
    err = np.empty(N, np.uint32)
    _check(lib().oracle_single_check(_p(envs), _p(err), _i64(N), S))
    return err


# ---------------------------------------------------------------- SimpleGridworld

def grid_obs_shape(mode: str, N: int, S: int):
    m, _ = parse_obs_mode(mode)
    if m == OBS_DEFAULT:
        return (N, 3, S, S)
    if m == OBS_RAW:
        return (N, 2, S, S)
    if m == OBS_POSITIONS:
        return (N, 4)
    return None


def grid_observe(envs, mode: str):
    N, _, S, _ = envs.shape
    m, n = parse_obs_mode(mode)
    obs = np.empty(grid_obs_shape(mode, N, S), np.float32)
    _check(lib().oracle_grid_observe(_p(envs), _p(obs), m, n, _i64(N), S))
    return obs


def grid_step(envs, actions, mode='default', seed=0, call=0, env_offset=0, inject_food=None):
    N, _, S, _ = envs.shape
    m, n = parse_obs_mode(mode)
    shape = grid_obs_shape(mode, N, S)
    obs = np.empty(shape, np.float32) if shape else None
    reward = np.empty(N, np.float32)
    done, ec = (np.empty(N, np.uint8) for _ in range(2))
    inj = _i32(inject_food)
    _check(lib().oracle_grid_step(_p(envs), _p(actions), _act_dtype(actions), _p(reward), _p(done), _p(ec), _p(obs),
                                  m, n, _i64(N), S, _u64(seed), _u64(call), _i64(env_offset), _p(inj)))
    return obs, reward, done, ec


def grid_reset(envs, done, start_location, mode='default', seed=0, call=0, env_offset=0, inject_reset=None):
    N, _, S, _ = envs.shape
    m, n = parse_obs_mode(mode)
    shape = grid_obs_shape(mode, N, S)
    obs = np.empty(shape, np.float32) if shape else None
    d = np.ascontiguousarray(np.asarray(done).reshape(N) != 0, dtype=np.uint8)
    inj = _i32(inject_reset)
    sy, sx = (-1, -1) if start_location is None else start_location
    _check(lib().oracle_grid_reset(_p(envs), _p(d), _p(obs), m, n, _i64(N), S, int(sy), int(sx), _u64(seed),
                                   _u64(call), _i64(env_offset), _p(inj)))
    return obs


def grid_rollout(envs, actions, start_location, mode='default', seed=0, call0=0, env_offset=0, inject_food=None,
                 inject_reset=None):
    N, _, S, _ = envs.shape
    T = actions.shape[0]
    m, n = parse_obs_mode(mode)
    shape = grid_obs_shape(mode, N, S)
    obs = np.empty((T,) + shape, np.float32) if shape else None
    reward = np.empty((T, N), np.float32)
    done, ec = (np.empty((T, N), np.uint8) for _ in range(2))
    inj_f, inj_r = _i32(inject_food), _i32(inject_reset)
    sy, sx = start_location
    _check(lib().oracle_grid_rollout(_p(envs), _p(actions), _act_dtype(actions), _p(reward), _p(done), _p(ec),
                                     _p(obs), m, n, _i64(N), S, _i64(T), int(sy), int(sx), _u64(seed), _u64(call0),
                                     _i64(env_offset), _p(inj_f), _p(inj_r)))
    return dict(obs=obs, reward=reward, done=done, edge_collision=ec)


# ---------------------------------------------------------------- MultiSnake

class MultiCfg(ctypes.Structure):
    _fields_ = [('boost', ctypes.c_int), ('food_on_death', ctypes.c_int), ('death_threshold', ctypes.c_float),
                ('boost_cost_prob', ctypes.c_float), ('food_mode', ctypes.c_int), ('food_rate', ctypes.c_float),
                ('max_food', ctypes.c_int), ('reward_on_death', ctypes.c_float), ('respawn_any', ctypes.c_int),
                ('colour_random', ctypes.c_int)]


class MultiInject(ctypes.Structure):
    _fields_ = [('death_a', ctypes.c_void_p), ('cost', ctypes.c_void_p), ('death_b', ctypes.c_void_p),
                ('rate', ctypes.c_void_p), ('food_cell', ctypes.c_void_p)]


class MultiResetInject(ctypes.Structure):
    _fields_ = [('create', ctypes.c_void_p), ('create_food', ctypes.c_void_p), ('colours', ctypes.c_void_p),
                ('respawn', ctypes.c_void_p)]


def multi_cfg(num_snakes, boost=True, food_on_death_prob=0.5, boost_cost_prob=0.5, food_mode='only_one',
              food_rate=5e-4, reward_on_death=-1, respawn_mode='all', colour_mode='random'):
    """Dynamics parameters with the reference's defaults (multi_snake.py:56-75)."""
    return MultiCfg(int(bool(boost)), int(food_on_death_prob > 0), np.float32(1 - food_on_death_prob),
                    np.float32(boost_cost_prob), {'only_one': 0, 'random_rate': 1}[food_mode], np.float32(food_rate),
                    8 * num_snakes, np.float32(reward_on_death), int(respawn_mode == 'any'),
                    int(colour_mode == 'random'))


def _addr(a):
    return None if a is None else a.ctypes.data


def multi_obs_shape(mode, N, K, S):
    m, n = parse_obs_mode(mode)
    if m == OBS_DEFAULT:
        return (K, N, 3, S, S)
    if m == OBS_PARTIAL:
        return (K, N, 3, 2 * n + 1, 2 * n + 1)
    return None


def multi_observe(st, mode):
    """st: dict with foods, heads, bodies, dones, boost_this_step, colours (numpy, reference shapes)."""
    N, _, S, _ = st['foods'].shape
    K = st['heads'].shape[0] // N
    m, n = parse_obs_mode(mode)
    obs = np.empty(multi_obs_shape(mode, N, K, S), np.float32)
    _check(lib().oracle_multi_observe(_p(st['foods']), _p(st['heads']), _p(st['bodies']), _p(st['dones']),
                                      _p(st['boost_this_step']), _p(st['colours']), _p(obs), m, n, _i64(N), K, S))
    return obs


def multi_step(st, actions, cfg, mode='full', seed=0, call=0, env_offset=0, inject=None):
    """st: dict of numpy state arrays modified in place: foods (N,1,S,S), heads/bodies (N*K,1,S,S), dones (N*K) u8,
    orientations (N*K) i64, boost_this_step (N*K) u8, colours (N*K,3) i16.  actions: (K,N) int64.
    inject: dict with death_a, cost, death_b, rate (uint8) and food_cell (int32), or None.
    Returns dict(obs (K,N,...), rewards, snake_collision, edge_collision, food, size (all (N*K)), all_done (N))."""
    N, _, S, _ = st['foods'].shape
    K = st['heads'].shape[0] // N
    m, n = parse_obs_mode(mode)
    shape = multi_obs_shape(mode, N, K, S)
    obs = np.empty(shape, np.float32) if shape else None
    rewards, food, size = (np.empty(N * K, np.float32) for _ in range(3))
    sc, ec = (np.empty(N * K, np.uint8) for _ in range(2))
    all_done = np.empty(N, np.uint8)
    actions = np.ascontiguousarray(actions, np.int64)
    assert actions.shape == (K, N)
    keep = []
    inj_ref = None
    if inject is not None:
        arrs = [np.ascontiguousarray(inject[k], np.uint8) for k in ('death_a', 'cost', 'death_b', 'rate')]
        arrs.append(np.ascontiguousarray(inject['food_cell'], np.int32))
        keep.extend(arrs)
        inj = MultiInject(*[_addr(a) for a in arrs])
        inj_ref = ctypes.byref(inj)
    _check(lib().oracle_multi_step(_p(st['foods']), _p(st['heads']), _p(st['bodies']), _p(st['dones']),
                                   _p(st['orientations']), _p(actions), _p(st['boost_this_step']), _p(rewards),
                                   _p(sc), _p(ec), _p(food), _p(size), _p(all_done), _p(st['colours']), _p(obs), m, n,
                                   _i64(N), K, S, ctypes.byref(cfg), _u64(seed), _u64(call), _i64(env_offset),
                                   inj_ref))
    return dict(obs=obs, rewards=rewards, snake_collision=sc, edge_collision=ec, food=food, size=size,
                all_done=all_done)


def multi_reset(st, done_env, cfg, seed=0, call=0, env_offset=0, inject=None):
    """Rebuilds the envs flagged in done_env (N), re-rolls colours of dead snakes, respawns (respawn_mode 'any').
    inject: dict with create (N,K,2), create_food (N), colours (N*K,3) i16, respawn (N,2), or None.
    Returns the number of envs in which a snake could not be placed."""
    N, _, S, _ = st['foods'].shape
    K = st['heads'].shape[0] // N
    d = np.ascontiguousarray(np.asarray(done_env).reshape(N) != 0, dtype=np.uint8)
    status = np.zeros(1, np.int32)
    keep = []
    inj_ref = None
    if inject is not None:
        arrs = [np.ascontiguousarray(inject['create'], np.int32), np.ascontiguousarray(inject['create_food'], np.int32),
                np.ascontiguousarray(inject['colours'], np.int16), np.ascontiguousarray(inject['respawn'], np.int32)]
        keep.extend(arrs)
        inj = MultiResetInject(*[_addr(a) for a in arrs])
        inj_ref = ctypes.byref(inj)
    _check(lib().oracle_multi_reset(_p(st['foods']), _p(st['heads']), _p(st['bodies']), _p(st['dones']),
                                    _p(st['orientations']), _p(st['colours']), _p(d), _p(status), _i64(N), K, S,
                                    ctypes.byref(cfg), _u64(seed), _u64(call), _i64(env_offset), inj_ref))
    return int(status[0])


def multi_check(st):
    N, _, S, _ = st['foods'].shape
    K = st['heads'].shape[0] // N
    err = np.empty(N, np.uint32)
    _check(lib().oracle_multi_check(_p(st['foods']), _p(st['heads']), _p(st['bodies']), _p(st['dones']), _p(err),
                                    _i64(N), K, S))
    return err


def multi_empty_state(N, K, S):
    return dict(foods=np.zeros((N, 1, S, S), np.float32), heads=np.zeros((N * K, 1, S, S), np.float32),
                bodies=np.zeros((N * K, 1, S, S), np.float32), dones=np.zeros(N * K, np.uint8),
                orientations=np.zeros(N * K, np.int64), boost_this_step=np.zeros(N * K, np.uint8),
                colours=np.zeros((N * K, 3), np.int16))


def orientations(envs):
    n, _, S, _ = envs.shape
    out = np.empty(n, np.int64)
    _check(lib().oracle_orientations(_p(envs), _p(out), _i64(n), S))
    return out


def multi_colours(N, K, fixed=False, seed=0, call=0, env_offset=0):
    col = np.empty((N * K, 3), np.int16)
    _check(lib().oracle_multi_colours(_p(col), _i64(N), K, int(fixed), _u64(seed), _u64(call), _i64(env_offset)))
    return col


# ---------------------------------------------------------------- RL glue next to the env

def a2c_returns(bootstrap, rewards, values, dones, gamma, use_gae=False, gae_lambda=None):
    """(T,N) returns of the reference's A2C.loss (wurm/rl/a2c.py:49-66).  fp32 arrays; dones uint8/bool."""
    rewards = np.ascontiguousarray(rewards, np.float32)
    T, N = rewards.shape
    values = np.ascontiguousarray(values, np.float32).reshape(T, N)
    bootstrap = np.ascontiguousarray(bootstrap, np.float32).reshape(N)
    d = np.ascontiguousarray(np.asarray(dones).reshape(T, N) != 0, dtype=np.uint8)
    out = np.empty((T, N), np.float32)
    gl = np.float32(gamma * gae_lambda) if use_gae else np.float32(0)
    _check(lib().oracle_a2c_returns(_p(bootstrap), _p(rewards), _p(values), _p(d), ctypes.c_float(np.float32(gamma)),
                                    int(bool(use_gae)), ctypes.c_float(gl), _p(out), _i64(T), _i64(N)))
    return out


def single_stats(envs, reward, done, self_collision, edge_collision):
    N, _, S, _ = envs.shape
    out = np.empty(5, np.float64)
    _check(lib().oracle_single_stats(_p(envs), _p(np.ascontiguousarray(reward, np.float32)),
                                     _p(np.ascontiguousarray(done, np.uint8)),
                                     _p(np.ascontiguousarray(self_collision, np.uint8)),
                                     _p(np.ascontiguousarray(edge_collision, np.uint8)), _p(out), _i64(N), S))
    return out


# ---------------------------------------------------------------- policy in the loop (oracle/policy.c)

def policy_param_count(E: int) -> int:
    return 64 * E + 64 + 64 * 64 + 64 + 4 * 64 + 4 + 64 + 1


def policy_forward(params, x):
    """x (M,E) -> probs (M,4), values (M): the build's arithmetic spec of FeedforwardAgent + softmax."""
    x = np.ascontiguousarray(x, np.float32)
    params = np.ascontiguousarray(params, np.float32)
    M, E = x.shape
    assert params.size == policy_param_count(E)
    probs, values = np.empty((M, 4), np.float32), np.empty(M, np.float32)
    f = lib().oracle_policy_forward
    f.restype = None
    for i in range(M):
        f(_p(params), int(E), _p(x[i]), _p(probs[i]), _p(values[i:i + 1]))
    return probs, values


def exp_spec(x):
    f = lib().oracle_exp_spec
    f.restype = ctypes.c_float
    f.argtypes = [ctypes.c_float]
    return np.asarray([f(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32).reshape(np.shape(x))


def single_policy_rollout(envs, obs0, params, T, obs_n=2, seed=0, call0=0, env_offset=0):
    """T iterations of (policy forward, sample, step, reset) over the envs (updated in place).  Returns dict of
    (T,N,...) arrays: actions, probs, values, reward, done, self_collision, edge_collision, obs."""
    N, _, S, _ = envs.shape
    E = 3 * (2 * obs_n + 1) ** 2
    obs0 = np.ascontiguousarray(obs0, np.float32).reshape(N, E)
    params = np.ascontiguousarray(params, np.float32)
    assert params.size == policy_param_count(E)
    actions = np.empty((T, N), np.int64)
    probs = np.empty((T, N, 4), np.float32)
    values, reward = np.empty((T, N), np.float32), np.empty((T, N), np.float32)
    done, sc, ec = (np.empty((T, N), np.uint8) for _ in range(3))
    obs = np.empty((T, N, E), np.float32)
    _check(lib().oracle_single_policy_rollout(_p(envs), _p(obs0), _p(params), _p(actions), _p(probs), _p(values),
                                              _p(reward), _p(done), _p(sc), _p(ec), _p(obs), int(obs_n), _i64(N), S,
                                              _i64(T), _u64(seed), _u64(call0), _i64(env_offset)))
    return dict(actions=actions, probs=probs, values=values, reward=reward, done=done, self_collision=sc,
                edge_collision=ec, obs=obs)
