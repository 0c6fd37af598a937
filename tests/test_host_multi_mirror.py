"""Host logic of MultiSnake around the resident mirror and the in-launch consistency masks (wurm_amd/envs/multi_snake.py:
_touch / _write_out / _escape / _watch_ok / _step_check_mask) WITHOUT a GPU: the class runs on CPU tensors over a stand-in
of the library that records every entry point it is asked for and plays the library's side of the protocol
(wurm_multi_step_packed marks the mirror current; the masks are whatever the test plants).  The kernels behind it are
tested on the GPU in tests/test_multi_resident.py."""
import ctypes

import pytest
import torch

from wurm_amd import _lib


class _Lib(object):
    """records (name, details); returns WURM_OK"""

    def __init__(self):
        self.calls = []
        self.mirror_bytes = 64

    def wurm_multi_colours(self, colours, N, K, fixed, seed, call, off, stream):
        return 0

    def wurm_multi_reset(self, *a):
        self.calls.append(('reset', {}))
        return 0

    def wurm_multi_observe(self, *a):
        self.calls.append(('observe', {}))
        return 0

    def wurm_multi_check(self, foods, heads, bodies, dones, err, N, K, S, stream):
        self.calls.append(('check', {}))
        ctypes.memset(err.value if hasattr(err, 'value') else err, 0, 4 * int(getattr(N, 'value', N)))
        return 0

    def wurm_multi_rollout(self, *a):
        self.calls.append(('rollout', {}))
        return 0

    def wurm_multi_resident_bytes(self, N, K, S):
        return self.mirror_bytes * int(getattr(N, 'value', N))

    def wurm_multi_resident_flush(self, c_addr, stream):
        c = _lib.MultiCall.from_address(c_addr)
        assert c.resident and c.resident_lazy and c.resident_valid
        self.calls.append(('flush', {}))
        return 0

    def step_packed(self, c_addr, of, ob, obs, obs_after, actions, call, pending, pre_call, stream):
        c = _lib.MultiCall.from_address(c_addr)
        self.calls.append(('step', dict(mirror=bool(c.resident), valid=bool(c.resident_valid), lazy=bool(c.resident_lazy),
                                        masks=bool(c.check_mask), pending=bool(pending), after=bool(obs_after))))
        if c.resident:
            c.resident_valid = 1
        return 0


N, K, S = 6, 2, 12


@pytest.fixture
def env_and_log(monkeypatch):
    lib = _Lib()
    monkeypatch.setattr(_lib, 'lib', lambda: lib)
    monkeypatch.setattr(_lib, 'require_device', lambda d: torch.device('cpu'))
    monkeypatch.setattr(_lib, 'stream_ptr', lambda i=None: 0)
    monkeypatch.setattr(_lib, 'call', lambda idx, fn, *a: fn(*a))
    monkeypatch.setattr(_lib, 'accessors', lambda: ((lambda: None), (lambda i: 0)))
    monkeypatch.setattr(_lib, 'multi_step_fn', lambda: lib.step_packed)
    monkeypatch.setattr(_lib, 'ptr', lambda t: None if t is None else ctypes.c_void_p(t.data_ptr()))
    from wurm_amd.envs import MultiSnake
    env = MultiSnake(N, K, S, device='cpu', seed=1)
    lib.calls.clear()
    env._lib = lib
    return env, lib.calls


def _step(env):
    a = {f'agent_{i}': torch.zeros(N, dtype=torch.long) for i in range(K)}
    return env.step(a)


def _steps(log):
    return [c[1] for c in log if c[0] == 'step']


def _names(log):
    return [c[0] for c in log]


def test_plain_loop_keeps_the_lazy_mirror(env_and_log):
    env, log = env_and_log
    for t in range(12):
        _, _, d, _ = _step(env)
        assert env.reset(d['__all__'], return_observations=False) is None
    st = _steps(log)
    assert all(s['mirror'] and s['lazy'] for s in st)
    assert [s['valid'] for s in st] == [False] + [True] * 11
    assert [s['pending'] for s in st] == [False] + [True] * 11
    assert 'flush' not in _names(log) and 'reset' not in _names(log)
    assert not any(s['masks'] for s in st)      # nobody has asked for check_consistency()


def test_reading_a_state_attribute_writes_the_mirror_out_and_ends_the_lazy_form(env_and_log):
    env, log = env_and_log
    _step(env); _step(env)
    f = env.foods
    assert _names(log)[-1] == 'flush'
    _step(env)
    assert not _steps(log)[-1]['lazy'] and _steps(log)[-1]['valid']     # written out, still current
    _step(env)
    f[0, 0, 3, 3] = 1.0                                                  # an in-place edit: found by the version counter
    _step(env)
    assert not _steps(log)[-1]['valid']
    _step(env)
    assert _steps(log)[-1]['valid']
    env.dones[0] = True                                                  # not mirrored: the kernels read it in place
    _step(env)
    assert _steps(log)[-1]['valid']
    env.bodies = env.bodies.clone()                                      # a tensor replaced
    _step(env)
    assert not _steps(log)[-1]['valid']
    assert _names(log).count('flush') == 1


def test_other_entry_points(env_and_log):
    env, log = env_and_log
    _step(env); _step(env)
    env._observe('full')                       # only reads: written out, the mirror stays current
    assert _names(log)[-2:] == ['flush', 'observe']
    _step(env)
    assert _steps(log)[-1]['valid'] and _steps(log)[-1]['lazy']
    env.reset(torch.ones(N, dtype=torch.bool), return_observations=False)   # an eager reset writes the state
    assert _names(log)[-2:] == ['flush', 'reset']
    _step(env)
    assert not _steps(log)[-1]['valid'] and not _steps(log)[-1]['lazy']     # the second write-out ended the lazy form
    _step(env)
    env.rollout(torch.zeros((2, K, N), dtype=torch.long))
    _step(env)
    assert not _steps(log)[-1]['valid']


def test_check_consistency_uses_the_masks_of_the_step_launch(env_and_log):
    env, log = env_and_log
    _, _, d, _ = _step(env)
    env.reset(d['__all__'])                    # (the first reset that wants its observations runs at once)
    env.check_consistency()                    # no masks yet: the checker over the tensors
    assert 'check' in _names(log)
    n_checks = _names(log).count('check')
    _, _, d, _ = _step(env)
    assert _steps(log)[-1]['masks'] and _steps(log)[-1]['after']
    env.reset(d['__all__'])                    # served by the step launch, postponed
    env._chk.zero_()                           # what the launch wrote: consistent
    env.check_consistency()
    assert _names(log).count('check') == n_checks and env._pending      # no pass over the tensors, nothing forced out
    _, _, d, _ = _step(env)
    env.reset(d['__all__'])
    env._chk.zero_()
    env._chk[1, 2] = 0x100                     # an overlap in env 2 of the state the reset observation shows
    with pytest.raises(RuntimeError, match='overlapping'):
        env.check_consistency()
    _, _, d, _ = _step(env)
    env.reset(d['__all__'])
    env._chk.zero_()
    env._chk[1, 0] = -1                        # an env the launch could not vouch for: the checker over the tensors
    env.check_consistency()
    assert _names(log).count('check') == n_checks + 1
    # a look at the state in between: the masks no longer describe it
    _, _, d, _ = _step(env)
    env.reset(d['__all__'])
    env._chk.zero_()
    _ = env.heads
    env.check_consistency()
    assert _names(log).count('check') == n_checks + 2


def test_masks_are_dropped_when_nobody_checks_any_more(env_and_log):
    env, log = env_and_log
    _step(env)
    env.check_consistency()
    _step(env)
    assert _steps(log)[-1]['masks']
    for _ in range(70):
        _step(env)
    assert not _steps(log)[-1]['masks']


def test_a_loop_that_invalidates_the_mirror_every_step_loses_it(env_and_log):
    env, log = env_and_log
    for t in range(12):
        _step(env)
        env.reset(torch.ones(N, dtype=torch.bool), return_observations=False)
    assert _steps(log)[0]['mirror'] and not _steps(log)[-1]['mirror']


def test_no_mirror_for_small_batches(env_and_log):
    env, log = env_and_log
    env._lib.mirror_bytes = 0
    from wurm_amd.envs import MultiSnake
    e2 = MultiSnake(N, K, S, device='cpu', seed=1)
    a = {f'agent_{i}': torch.zeros(N, dtype=torch.long) for i in range(K)}
    e2.step(a)
    assert not _steps(log)[-1]['mirror'] and e2._mirror is None
