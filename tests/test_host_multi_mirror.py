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

    rollout_keeps_mirror = False   # True: the batch is one the library's mirror-keeping rollout kernel serves

    def wurm_multi_rollout_resident(self, *a):
        resident, valid_addr, lazy = a[22], a[23], a[24]
        valid = ctypes.c_int.from_address(valid_addr)
        self.calls.append(('rollout_resident', dict(mirror=bool(resident), valid=bool(valid.value), lazy=bool(lazy))))
        if self.rollout_keeps_mirror:
            valid.value = 1
        else:                        # the library's fallback: a lazy mirror is written out first, and it is stale afterwards
            if lazy and valid.value:
                self.calls.append(('flush', {}))
            valid.value = 0
        return 0

    def wurm_multi_resident_bytes(self, N, K, S):
        return self.mirror_bytes * int(getattr(N, 'value', N))

    size_calls = 0

    def wurm_multi_resident_size(self, N, K, S):      # `resident_mirror=True`: no batch-size threshold
        self.size_calls += 1
        return self.mirror_bytes * int(getattr(N, 'value', N))

    def wurm_multi_resident_flush(self, c_addr, stream):
        c = _lib.MultiCall.from_address(c_addr)
        assert c.resident and c.resident_lazy and c.resident_valid
        self.calls.append(('flush', {}))
        return 0

    def step_slot(self, c_addr, sl_addr, slot, actions, call, pending, pre_call, want_after, stream):
        c = _lib.MultiCall.from_address(c_addr)
        self.calls.append(('step', dict(mirror=bool(c.resident), valid=bool(c.resident_valid), lazy=bool(c.resident_lazy),
                                        masks=bool(c.check_mask), pending=bool(pending), after=bool(want_after))))
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
    monkeypatch.setattr(_lib, 'step_slot_fn', lambda name='wurm_multi_step_slot': lib.step_slot)
    monkeypatch.setattr(_lib, 'torch_helpers', lambda: None)
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


def test_rollout_on_the_mirror(env_and_log):
    """env.rollout hands the mirror to wurm_multi_rollout_resident: nothing is written out in front of it, the mirror stays
    current where the library's mirror-keeping kernel serves the batch, and is stale (a lazy one written out by the library
    itself) where it does not"""
    env, log = env_and_log
    env._lib.rollout_keeps_mirror = True
    env.rollout(torch.zeros((2, K, N), dtype=torch.long))                 # a fresh env: the rollout makes the mirror
    assert _names(log) == ['rollout_resident'] and log[-1][1] == dict(mirror=True, valid=False, lazy=True)
    assert env._mc.foods and env._mc.heads and env._mc.bodies             # (what a write-out of this mirror will need)
    _step(env)
    assert _steps(log)[-1]['valid'] and _steps(log)[-1]['lazy']            # ... and the step reads it
    env.rollout(torch.zeros((3, K, N), dtype=torch.long))
    assert log[-1] == ('rollout_resident', dict(mirror=True, valid=True, lazy=True)) and 'flush' not in _names(log)
    _, _, d, _ = _step(env)
    env.reset(d['__all__'], return_observations=False)                    # postponed into the next launch ...
    env.rollout(torch.zeros((2, K, N), dtype=torch.long))                 # ... which is not a step: applied first
    assert _names(log)[-2:] == ['reset', 'rollout_resident']
    b = env.bodies                                                        # the caller holds a tensor: eager from now on
    _step(env)
    env.rollout(torch.zeros((2, K, N), dtype=torch.long))
    assert log[-1][1]['lazy'] is False
    b[0, 0, 1, 1] = 3                                                     # an in-place edit is noticed by the rollout too
    env.rollout(torch.zeros((2, K, N), dtype=torch.long))
    assert log[-1][1]['valid'] is False
    env._lib.rollout_keeps_mirror = False                                 # a batch the kernel does not serve
    _step(env)
    env.rollout(torch.zeros((2, K, N), dtype=torch.long))
    _step(env)
    assert not _steps(log)[-1]['valid']
    assert env.mirror_state()['state'] in ('eager', 'lazy')


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


def test_an_in_place_edit_of_an_alias_after_the_step_is_seen_by_check_consistency(env_and_log):
    """the masks of the step launch describe the state the launch left; an alias of a state tensor the caller
    holds may have been edited since — the version counters are compared before the masks are trusted"""
    env, log = env_and_log
    _step(env)
    env.check_consistency()
    b = env.bodies                             # the caller holds it from now on (watched)
    _step(env)
    env._chk.zero_()
    n = _names(log).count('check')
    env.check_consistency()                    # nothing edited: served by the launch masks
    assert _names(log).count('check') == n
    _step(env)
    env._chk.zero_()
    b[0] = 7                                   # in-place edit AFTER the step
    env.check_consistency()
    assert _names(log).count('check') == n + 1, 'the launch masks were trusted over an edited tensor'
    _step(env)
    assert not _steps(log)[-1]['valid']        # and the mirror is rebuilt from the edited tensors


def test_an_edit_followed_by_a_look_is_not_forgotten(env_and_log):
    """edit through an alias, then read an attribute again (which takes the tensor's version anew), then step / rollout: the
    launch must not be told the mirror is current (found by tools/fuzz_parity.py's class-level family in round 4)"""
    env, log = env_and_log
    _step(env)
    b = env.bodies
    _step(env); _step(env)
    assert _steps(log)[-1]['valid']
    b[0, 0, 3, 3] = 2.0
    assert env.bodies.data_ptr() == b.data_ptr()   # a look between the edit and the next launch
    _step(env)
    assert not _steps(log)[-1]['valid']
    _step(env)
    assert _steps(log)[-1]['valid']
    b[1, 0, 4, 4] = 1.0
    assert env.foods is not None     # (a look at ANOTHER attribute)
    env._lib.rollout_keeps_mirror = True
    env.rollout(torch.zeros((2, K, N), dtype=torch.long))
    assert log[-1][0] == 'rollout_resident' and log[-1][1]['valid'] is False


def test_a_replaced_state_tensor_is_no_longer_watched(env_and_log):
    """rebinding env.foods / heads / bodies every iteration must not keep every old tensor alive and watched"""
    env, log = env_and_log
    olds = []
    for t in range(5):
        _step(env)
        new = torch.zeros(N, 1, S, S)
        olds.append(new)
        env.foods = new
    assert len(env._watched) == 1 and env._watched[0][0].data_ptr() == olds[-1].data_ptr()
    _step(env); _step(env)
    assert _steps(log)[-1]['valid']
    olds[0][0, 0, 1, 1] = 1.0                  # an edit of a tensor that is no longer part of the state: nothing to do
    _step(env)
    assert _steps(log)[-1]['valid']
    olds[-1][0, 0, 1, 1] = 1.0                 # the current one: the mirror is stale
    _step(env)
    assert not _steps(log)[-1]['valid']


def test_resident_mirror_keyword_and_mirror_state(env_and_log):
    env, log = env_and_log
    lib = env._lib
    from wurm_amd.envs import MultiSnake
    assert env.mirror_state()['state'] == 'off' and env.mirror_state()['why'] == 'no step yet'
    _step(env)
    assert env.mirror_state() == {'state': 'lazy', 'why': 'on', 'policy': None, 'current': True, 'bytes': 64 * N}

    off = MultiSnake(N, K, S, device='cpu', seed=1, resident_mirror=False)
    a = len(log)
    _step(off); _step(off)
    assert not any(s['mirror'] for s in _steps(log[a:])) and off.mirror_state()['why'] == 'resident_mirror=False'

    forced = MultiSnake(N, K, S, device='cpu', seed=1, resident_mirror=True)
    a = len(log)
    for t in range(12):                        # eager resets after every step: the automatic policy would switch it off
        _step(forced)
        forced.reset(torch.ones(N, dtype=torch.bool), return_observations=False)
    assert lib.size_calls == 1 and all(s['mirror'] and s['lazy'] for s in _steps(log[a:]))
    forced._observe(); _step(forced); forced._observe(); _step(forced)
    assert _steps(log)[-1]['lazy'] and forced.mirror_state()['state'] == 'lazy'
    _ = forced.heads                           # correctness, not a heuristic
    _step(forced)
    assert not _steps(log)[-1]['lazy'] and 'caller holds' in forced.mirror_state()['why']

    eager = MultiSnake(N, K, S, device='cpu', seed=1, resident_mirror='eager')
    a = len(log)
    _step(eager); _step(eager)
    assert all(s['mirror'] and not s['lazy'] for s in _steps(log[a:]))

    auto = MultiSnake(N, K, S, device='cpu', seed=1)
    for t in range(12):
        _step(auto)
        auto.reset(torch.ones(N, dtype=torch.bool), return_observations=False)
    assert auto.mirror_state()['state'] == 'off' and auto.mirror_state()['why'].startswith('adaptive:')


def test_reference_test_access_patterns_never_step_on_a_stale_mirror(env_and_log):
    """The access patterns of the reference's MultiSnake tests (tests/test_multi_snake_env.py:26-44 rebinding foods / heads /
    bodies and in-place edits of what the attributes return, :130 check_consistency() every step, :180 / :288 / :401-403
    dynamics attributes set after construction) with the mirror FORCED on: no step launch is told the mirror is current
    after anything wrote the state some other way."""
    env, log = env_and_log
    from wurm_amd.envs import MultiSnake
    env = MultiSnake(N, K, S, device='cpu', seed=1, resident_mirror=True, manual_setup=True)
    wrote = [False]

    def step():
        n = len(log)
        out = _step(env)
        s = _steps(log[n:])[-1]
        assert not (wrote[0] and s['valid']), 'a step launch was told the mirror is current after a write'
        wrote[0] = False
        return out

    # :26-44: the board is written into the tensors the attributes return, orientations assigned
    env.foods[:, 0, 1, 1] = 1
    env.heads[0, 0, 5, 5] = 1
    env.bodies[0, 0, 5, 5] = 4
    env.orientations = torch.zeros(N * K, dtype=torch.long); wrote[0] = True
    step()
    # :130 check_consistency() after every step, actions in between
    for _ in range(3):
        step()
        env._chk.zero_()
        env.check_consistency()
    # in-place edits between steps through the attribute and through an alias taken earlier
    h = env.heads
    step()
    h[1, 0, 3, 3] = 1; wrote[0] = True
    step()
    env.bodies[2, 0, 4, 4] = 2; wrote[0] = True
    step()
    # rebinding (tests/test_multi_snake_env.py:29-36 assigns fresh tensors)
    env.foods = torch.zeros(N, 1, S, S); wrote[0] = True
    step()
    # :180 / :288 / :401-403: dynamics attributes after construction do not touch the state
    env.food_on_death_prob, env.boost, env.boost_cost_prob, env.respawn_mode = 1.0, False, 0.0, 'any'
    step()
    assert _steps(log)[-1]['valid']
    # reset with flags of the caller's own, reset() and a rollout write the state with other kernels
    env.reset(torch.ones(N, dtype=torch.bool), return_observations=False); wrote[0] = True
    step()
    env.reset(); wrote[0] = True
    step()
    del h   # (while the caller holds an alias of a state tensor no reset is postponed: _alias_free)
    _, _, d, _ = step()
    env.reset(d['__all__'], return_observations=False)     # the step's own flags: postponed, nothing written
    step()
    assert _steps(log)[-1]['pending'] and _steps(log)[-1]['valid']


def test_a_dropped_alias_gives_the_fast_path_back(env_and_log):
    """one read of `env.heads` (experiments/multiagent.py:531) put the tensor on the watch list for ever, and
    `step` takes its one-C-call path only while that list is empty.  Once the caller has let go of the tensor nothing is
    left to watch (the GPU twin of this test counts the slow steps: tests/test_hip_multi_fused.py)."""
    env, log = env_and_log
    for t in range(3):
        _step(env)
    assert float(env.heads.sum()) >= 0.0               # read and dropped at once
    assert env._watched
    _step(env)
    assert not env._watched and _steps(log)[-1]['valid']
    h = env.heads                                      # held: watched for as long as the caller holds it
    for t in range(4):
        _step(env)
    assert env._watched and _steps(log)[-1]['valid']
    h[0, 0, 2, 2] = 1.0                                # an edit made before the alias is dropped is not forgotten
    del h
    _step(env)
    assert not _steps(log)[-1]['valid'] and not env._watched
    _step(env)
    assert _steps(log)[-1]['valid']


def test_an_assigned_info_lasts_until_the_next_step(env_and_log):
    """reference multi_snake.py:729 rebinds `self.info` in every step (the fast path kept a user's dict)"""
    env, log = env_and_log
    _step(env); _step(env)
    mine = {'x': 1}
    env.info = mine
    assert env.info is mine
    out = _step(env)
    assert env.info is not mine and env.info is out[3]


def test_sanitize_movements_is_the_reference_formula(env_and_log):
    """reference multi_snake.py:336-339 (public; its own step calls it at :493)"""
    env, _ = env_and_log
    mv = torch.tensor([0, 1, 2, 3, 0, 1, 2, 3])
    ori = torch.tensor([0, 1, 2, 3, 1, 2, 3, 0])
    got = env.sanitize_movements(movements=mv, orientations=ori)
    assert got.tolist() == [2, 3, 0, 1, 0, 1, 2, 3] and got.dtype == torch.long
