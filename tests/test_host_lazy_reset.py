"""Host logic of the one-launch step path (wurm_amd/envs/_fast_step.py) WITHOUT a GPU: the C ABI is replaced by a
recording stand-in (no compute), the env lives on CPU tensors, and the SEQUENCE of entry points, counters and flags the
class hands to the library is checked for the call patterns a caller can produce — the contract the GPU tests then check
for bit-exact results."""
import ctypes

import pytest
import torch

from wurm_amd import _lib


class _Recorder(object):
    """Stands in for libwurm_hip.so: every entry point records (name, selected arguments) and returns WURM_OK."""

    def __init__(self):
        self.calls = []
        self.flushes = []

    def wurm_single_reset(self, envs, done, obs, m, n, N, S, seed, call, off, inj, stream):
        self.calls.append(('reset', dict(call=call, obs=obs is not None, mode=m)))
        return 0

    def wurm_single_observe(self, envs, obs, m, n, N, S, stream):
        self.calls.append(('observe', dict(mode=m)))
        return 0

    fail_next = False
    mirror_bytes_per_env = 32

    def wurm_single_resident_bytes(self, N, S, m, n):
        return self.mirror_bytes_per_env * int(getattr(N, 'value', N))

    size_calls = 0

    def wurm_single_resident_size(self, N, S, m, n):      # what `resident_mirror=True` asks: no batch-size threshold
        self.size_calls += 1
        return self.mirror_bytes_per_env * int(getattr(N, 'value', N))

    def wurm_single_resident_flush(self, c_addr, stream):
        c = _lib.SingleCall.from_address(c_addr)
        assert c.resident and c.resident_lazy and c.resident_valid  # (the library would do nothing otherwise)
        self.flushes.append(len(self.calls))   # kept apart from `calls`: position in the call sequence
        return 0

    def step_slot(self, c_addr, sl_addr, slot, actions, dtype, call, pending, pre_call, want_after, stream):
        c = _lib.SingleCall.from_address(c_addr)
        if self.fail_next:
            self.fail_next = False
            if c.resident:
                c.resident_valid = 0
            return -3
        self.calls.append(('step', dict(slot=slot, call=call, pending=bool(pending), pre_call=pre_call,
                                        want_after=bool(want_after), dtype=dtype, obs_mode=c.obs_mode, envs=c.envs,
                                        mirror=bool(c.resident), mirror_valid=bool(c.resident_valid),
                                        lazy=bool(c.resident_lazy))))
        if c.resident:  # what wurm_single_step_slot does: the mirror is current after a launch that was given it
            c.resident_valid = 1
        return 0


_STEP_SLOT_T = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p)


class _CSlot(object):
    """the recorder's step_slot behind a real C function pointer, as wurm_amd._fastcall.Stepper needs it"""

    def __init__(self, rec):
        self._cb = _STEP_SLOT_T(lambda c, sl, slot, a, dt, call, pend, pre, want, st: rec.step_slot(c, sl, slot, a, dt, call,
                                                                                              pend, pre, want, st))
        self.c_address = ctypes.cast(self._cb, ctypes.c_void_p).value

    def __call__(self, *args):
        return self._cb(*args)


def _have_c_stepper():
    try:
        from wurm_amd import _fastcall
        return hasattr(_fastcall, 'Stepper')
    except ImportError:
        return False


def _torchinfo_addresses():
    """the real tensor-facts helper of wurm_amd/libwurm_torchinfo.so (it works on CPU tensors) with stand-ins for the two
    that ask the HIP runtime: current device -1 (as the patched accessor), stream 0"""
    import os
    path = os.path.join(os.path.dirname(_lib.__file__), 'libwurm_torchinfo.so')
    if not os.path.exists(path):
        return None
    l = ctypes.PyDLL(path)
    keep = (l, ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_int)(lambda idx: 0), ctypes.CFUNCTYPE(ctypes.c_int)(lambda: -1))
    return keep, (ctypes.cast(l.wurm_torch_tensor_info, ctypes.c_void_p).value,
                  ctypes.cast(keep[1], ctypes.c_void_p).value, ctypes.cast(keep[2], ctypes.c_void_p).value,
                  ctypes.cast(l.wurm_torch_alias_free, ctypes.c_void_p).value)   # (the real one: it works on CPU tensors)


@pytest.fixture(params=['python', 'c', 'c+torchinfo'])
def env_and_log(monkeypatch, request):
    """the env on CPU tensors over a recording stand-in of the library, once with the Python step machine (PyStepper) and
    once with the C one (wurm_amd._fastcall.Stepper calling the recorder through a C function pointer)"""
    if request.param != 'python' and not _have_c_stepper():
        pytest.skip('wurm_amd/_fastcall is not built')
    helpers = _torchinfo_addresses() if request.param == 'c+torchinfo' else None
    if request.param == 'c+torchinfo' and helpers is None:
        pytest.skip('wurm_amd/libwurm_torchinfo.so is not built')
    rec = _Recorder()
    slot = _CSlot(rec) if request.param != 'python' else rec.step_slot
    monkeypatch.setattr(_lib, 'lib', lambda: rec)
    monkeypatch.setattr(_lib, 'require_device', lambda d: torch.device('cpu'))
    monkeypatch.setattr(_lib, 'stream_ptr', lambda i=None: 0)
    monkeypatch.setattr(_lib, 'call', lambda idx, fn, *a: fn(*a))
    monkeypatch.setattr(_lib, 'accessors', lambda: ((lambda: -1), (lambda i: 0)))
    monkeypatch.setattr(_lib, 'step_slot_fn', lambda name='wurm_single_step_slot': slot)
    # (the real helpers ask the HIP runtime for the device and the stream: stand-ins, or the generic attribute path)
    monkeypatch.setattr(_lib, 'torch_helpers', lambda: helpers[1] if helpers else None)
    from wurm_amd.envs import SingleSnake
    from wurm_amd.envs._fast_step import PyStepper
    env = SingleSnake(num_envs=8, size=9, observation_mode='partial_2', device='cpu', seed=5)
    assert isinstance(env._fs, PyStepper) == (request.param == 'python')
    rec.calls.clear()
    env._rec = rec
    env._keep = helpers
    return env, rec.calls


def _step(env, a=None):
    a = torch.zeros(8, dtype=torch.int64) if a is None else a
    return env.step(a)


def test_reset_with_the_steps_own_done_is_deferred_into_the_next_step(env_and_log):
    env, log = env_and_log
    obs, r, d, info = _step(env)
    assert env.reset(d, return_observations=False) is None
    assert [c[0] for c in log] == ['step']                      # no reset launch
    _step(env)
    s0, s1 = log[0][1], log[1][1]
    assert not s0['pending'] and s1['pending'] and s1['pre_call'] == s0['call'] + 1 and s1['call'] == s0['call'] + 2
    assert obs.shape == (8, 75) and r.shape == (8, 1) and d.shape == (8, 1) and d.dtype == torch.bool
    assert set(info) == {'self_collision', 'edge_collision'}


def test_reading_envs_flushes_the_postponed_reset(env_and_log):
    env, log = env_and_log
    _, _, d, _ = _step(env)
    env.reset(d, return_observations=False)
    _ = env.envs                                                # looks at the state
    assert [c[0] for c in log] == ['step', 'reset'] and log[1][1]['call'] == log[0][1]['call'] + 1
    assert not log[1][1]['obs']
    _step(env)
    assert not log[2][1]['pending']                             # nothing left to apply


def test_reset_observation_moves_into_the_step_launch_once_asked_for(env_and_log):
    env, log = env_and_log
    _, _, d, _ = _step(env)
    back = env.reset(d)                                         # first time: eager launch with an observation
    assert back is not None and [c[0] for c in log] == ['step', 'reset'] and log[1][1]['obs']
    _, _, d, _ = _step(env)
    assert log[2][1]['want_after'] and not log[2][1]['pending']
    back = env.reset(d)                                         # now served by the step launch, reset deferred
    assert back is not None and len(log) == 3
    _step(env)
    assert log[3][1]['pending'] and log[3][1]['want_after']
    env.reset(env.done, return_observations=False)              # env.done is the step's own flags too
    _step(env)
    assert log[4][1]['pending'] and not log[4][1]['want_after']


def test_anything_else_is_an_eager_reset(env_and_log):
    env, log = env_and_log
    _, _, d, _ = _step(env)
    env.reset(d.clone(), return_observations=False)             # another tensor object
    _, _, d, _ = _step(env)
    d[:] = False                                                # edited in place: version counter moved
    env.reset(d, return_observations=False)
    _, _, d, _ = _step(env)
    env._observe('default')                                     # consumed nothing, but looked at the state
    env.reset(d, return_observations=False)
    names = [c[0] for c in log]
    assert names == ['step', 'reset', 'step', 'reset', 'step', 'observe', 'reset']
    assert not any(c[1]['pending'] for c in log if c[0] == 'step')
    calls = [c[1]['call'] for c in log if 'call' in c[1]]
    assert calls == sorted(calls) and len(set(calls)) == len(calls)   # one counter value per call, increasing


def test_lazy_reset_can_be_switched_off_and_assignment_drops_the_pending_reset(env_and_log):
    env, log = env_and_log
    _, _, d, _ = _step(env)
    env.reset(d, return_observations=False)
    env.envs = torch.zeros(8, 3, 9, 9)                          # wholesale replacement: the postponed reset is void
    _step(env)
    assert [c[0] for c in log] == ['step', 'step'] and not log[1][1]['pending']
    assert log[1][1]['envs'] == env.envs.data_ptr()
    env.lazy_reset = False
    _, _, d, _ = _step(env)
    env.reset(d, return_observations=False)
    assert log[-1][0] == 'reset'


def test_an_edited_done_costs_one_eager_reset_not_the_rest_of_the_slab(env_and_log):
    """the steps of a slab share one version counter: after an in-place edit the deferral resumes with the next step"""
    env, log = env_and_log
    _, _, d, _ = _step(env)
    d[0] = True
    env.reset(d, return_observations=False)                     # eager: the flags are no longer the kernel's own copy
    _, _, d, _ = _step(env)
    env.reset(d, return_observations=False)                     # deferred again
    _step(env)
    assert [c[0] for c in log] == ['step', 'reset', 'step', 'step'] and log[3][1]['pending']


def test_step_and_reset_under_inference_mode(env_and_log):
    """tensors allocated under torch.inference_mode() have no version counter: resets run eagerly, nothing raises"""
    env, log = env_and_log
    with torch.inference_mode():
        for _ in range(3):
            _, _, d, _ = _step(env)
            assert env.reset(d) is not None
    assert [c[0] for c in log] == ['step', 'reset'] * 3


def test_editing_the_state_between_step_and_reset_drops_the_precomputed_observation(env_and_log):
    env, log = env_and_log
    _, _, d, _ = _step(env)
    env.reset(d)                                                # asks for the reset observation: eager the first time
    _, _, d, _ = _step(env)                                     # this launch pre-computes it ...
    assert log[-1][1]['want_after']
    env.envs[0, 0, 4, 4] = 1                                    # ... but the caller edits the state it was computed from
    n = len(log)
    assert env.reset(d) is not None
    assert [c[0] for c in log[n:]] == ['reset'] and log[-1][1]['obs']   # observed after the edit, as the reference would


def test_done_attribute_is_the_last_steps_flags_unless_assigned_since(env_and_log):
    env, log = env_and_log
    _, _, d, _ = _step(env)
    assert env.done.shape == (8,) and env.done.data_ptr() == d.data_ptr()
    mine = torch.ones(8, dtype=torch.bool)
    env.done = mine
    assert env.done is mine
    env.reset()                                                 # uses the assigned tensor: eager
    assert log[-1][0] == 'reset'
    _, _, d2, _ = _step(env)
    assert env.done.data_ptr() == d2.data_ptr()                 # a step overwrites the attribute, as in the reference


def test_a_failed_launch_consumes_nothing(env_and_log):
    env, log = env_and_log
    _, _, d, _ = _step(env)
    env.reset(d, return_observations=False)
    call, slot = env._fs.call, env._fs.slot
    c = _lib.SingleCall.from_address(ctypes.addressof(env._c))
    envs_ptr = c.envs
    env._rec.fail_next = True
    with pytest.raises(Exception):
        _step(env)
    assert env._fs.pending and env._fs.call == call and env._fs.slot == slot and c.envs == envs_ptr
    _step(env)                                                   # the postponed reset is still applied
    assert log[-1][1]['pending']


def test_argument_errors_match_the_reference(env_and_log):
    env, _ = env_and_log
    with pytest.raises(TypeError):
        env.step(torch.zeros(8))
    with pytest.raises(RuntimeError):
        env.step(torch.zeros(7, dtype=torch.int64))
    with pytest.raises(RuntimeError):
        env.step(torch.zeros(8, dtype=torch.int16))


def _steps(log):
    return [c[1] for c in log if c[0] == 'step']


def test_the_mirror_is_built_once_in_the_plain_loop(env_and_log):
    """wurm_single_call.resident: the step launch keeps the compact mirror of the state current; nothing in the loop of
    experiments/main.py:212-227 (step, deferred reset in either form, a new output slab) makes it stale"""
    env, log = env_and_log
    for t in range(70):  # 64 steps per slab
        _, _, d, _ = _step(env)
        if t < 35:
            env.reset(d)
        else:
            env.reset(d, return_observations=False)
    st = _steps(log)
    assert all(s['mirror'] for s in st)
    # (the very first reset(d) that wants its observation runs eagerly, test_reset_observation_moves_into_the_step_launch...)
    assert [s['mirror_valid'] for s in st] == [False, False] + [True] * 68


def test_everything_else_that_writes_the_state_makes_the_mirror_stale(env_and_log):
    env, log = env_and_log
    _step(env); _step(env)
    assert _steps(log)[-1]['mirror_valid']
    env.reset(torch.ones(8, dtype=torch.bool), return_observations=False)   # an eager reset
    _step(env)
    assert not _steps(log)[-1]['mirror_valid']
    _, _, d, _ = _step(env)
    assert _steps(log)[-1]['mirror_valid']
    env.reset(d, return_observations=False)                                  # postponed ...
    env.check_consistency() if False else env._observe('default')          # ... and flushed by a look at the state
    _step(env)
    assert not _steps(log)[-1]['mirror_valid']
    _step(env)
    env.envs = torch.zeros(8, 3, 9, 9)                                       # a new state tensor
    _step(env)
    assert not _steps(log)[-1]['mirror_valid']
    _step(env)
    assert _steps(log)[-1]['mirror_valid']
    env._rec.fail_next = True                                                # a launch that failed: rebuilt next time
    with pytest.raises(Exception):
        _step(env)
    _step(env)
    assert not _steps(log)[-1]['mirror_valid']


def test_a_state_tensor_the_caller_holds_is_watched_for_in_place_edits(env_and_log):
    env, log = env_and_log
    _step(env)
    e = env.envs                     # reading alone changes nothing ...
    _step(env)
    assert _steps(log)[-1]['mirror_valid']
    _step(env)
    e[0, 0, 3, 3] = 1.0              # ... an in-place edit, at any later time, does: the next step rebuilds the mirror
    _step(env)
    assert not _steps(log)[-1]['mirror_valid']
    _step(env)
    assert _steps(log)[-1]['mirror_valid']
    e.view(-1)[5:7].zero_()          # through a view as well (views share the version counter)
    _step(env)
    assert not _steps(log)[-1]['mirror_valid']
    t2 = torch.zeros(8, 3, 9, 9)
    env.envs = t2                    # a tensor handed in is held by the caller too
    _step(env); _step(env)
    assert _steps(log)[-1]['mirror_valid']
    t2[1, 2, 4, 4] = 2.0
    _step(env)
    assert not _steps(log)[-1]['mirror_valid']


def test_an_edit_followed_by_a_look_is_not_forgotten(env_and_log):
    """edit through an alias, then READ env.envs again (which takes the tensor's version anew), then step: the step must not
    be told the mirror is current (found by tools/fuzz_parity.py's class-level family in round 4)"""
    env, log = env_and_log
    _step(env)
    e = env.envs
    _step(env); _step(env)
    assert _steps(log)[-1]['mirror_valid']
    e[0, 0, 3, 3] = 1.0
    assert env.envs.data_ptr() == e.data_ptr()   # a look between the edit and the next step
    _step(env)
    assert not _steps(log)[-1]['mirror_valid']
    _step(env)
    assert _steps(log)[-1]['mirror_valid']


def test_a_state_tensor_without_version_counter_switches_the_mirror_off(env_and_log):
    env, log = env_and_log
    _step(env)
    assert _steps(log)[-1]['mirror']
    with torch.inference_mode():
        env.envs = torch.zeros(8, 3, 9, 9)   # an inference tensor: edits could not be seen
    _step(env)
    assert not _steps(log)[-1]['mirror']
    _step(env)
    assert not _steps(log)[-1]['mirror']


def test_no_mirror_for_shapes_the_library_does_not_serve(env_and_log):
    env, log = env_and_log
    env._rec.mirror_bytes_per_env = 0
    env.observation_mode = 'default'         # (the recorder decides; the library serves size 9, partial_2 / none)
    _step(env)
    assert not _steps(log)[-1]['mirror']


def test_a_lazy_mirror_is_written_out_before_anything_else_looks_at_the_state(env_and_log):
    """wurm_single_call.resident_lazy: the step launches do not write `envs`; whatever else reads or writes the state is
    preceded by wurm_single_resident_flush — once — and the caller getting hold of the tensor ends the lazy form"""
    env, log = env_and_log
    rec = env._rec
    for _ in range(3):
        _, _, d, _ = _step(env)
        env.reset(d, return_observations=False)
    assert all(s['lazy'] for s in _steps(log)) and rec.flushes == []
    env._observe('default')                      # flush, (postponed) reset, observe
    assert rec.flushes == [3] and [c[0] for c in log[3:]] == ['reset', 'observe']
    env._observe('default')                      # envs are current now: no second flush
    assert rec.flushes == [3]
    _step(env)
    assert _steps(log)[-1]['lazy'] and not _steps(log)[-1]['mirror_valid']
    _step(env)
    n = len(log)
    e = env.envs                                 # written out, handed out, eager from now on
    assert rec.flushes == [3, n]
    _step(env)
    assert not _steps(log)[-1]['lazy'] and _steps(log)[-1]['mirror_valid']   # (writing out leaves the mirror current)
    env._observe('default')
    assert rec.flushes == [3, n]                 # nothing to write out in the eager form
    del e


def test_replacing_the_state_tensor_writes_a_lazy_mirror_out_to_the_old_one_first(env_and_log):
    env, log = env_and_log
    rec = env._rec
    _step(env); _step(env)
    old_ptr = env._c.envs
    seen = []
    orig = rec.wurm_single_resident_flush
    rec.wurm_single_resident_flush = lambda c_addr, stream: (seen.append(_lib.SingleCall.from_address(c_addr).envs),
                                                            orig(c_addr, stream))[1]
    env.envs = torch.zeros(8, 3, 9, 9)
    assert seen == [old_ptr]
    _step(env)
    assert not _steps(log)[-1]['lazy'] and not _steps(log)[-1]['mirror_valid'] and _steps(log)[-1]['envs'] != old_ptr


def test_a_look_that_only_reads_keeps_the_mirror_and_the_second_one_ends_the_lazy_form(env_and_log):
    """_observe / check_consistency read the state with another kernel: a lazy mirror is written out for them but stays
    current; a caller that keeps doing that (the reference's loops check consistency every step) gets the eager form"""
    env, log = env_and_log
    rec = env._rec
    _step(env); _step(env)
    env._observe('default')
    assert rec.flushes == [2]
    _step(env)
    assert _steps(log)[-1]['mirror_valid'] and _steps(log)[-1]['lazy']
    env._observe('default')
    assert len(rec.flushes) == 2
    _step(env)
    assert _steps(log)[-1]['mirror_valid'] and not _steps(log)[-1]['lazy']
    env._observe('default')
    assert len(rec.flushes) == 2          # eager: nothing to write out
    _step(env)
    assert _steps(log)[-1]['mirror_valid']


def test_a_loop_that_invalidates_the_mirror_every_step_loses_it(env_and_log):
    """eager resets after (nearly) every step: the mirror would be rebuilt every step for nothing"""
    env, log = env_and_log
    for t in range(12):
        _step(env)
        env.reset(torch.ones(8, dtype=torch.bool), return_observations=False)   # not the step's own flags: eager
    assert _steps(log)[0]['mirror'] and not _steps(log)[-1]['mirror']
    # ... but an occasional one does not cost it
    env2_log_start = len(log)
    from wurm_amd.envs import SingleSnake
    e2 = SingleSnake(num_envs=8, size=9, observation_mode='partial_2', device='cpu', seed=5)
    for t in range(200):
        _, _, d, _ = _step(e2)
        e2.reset(d, return_observations=False)
        if t % 20 == 19:
            e2.reset(torch.ones(8, dtype=torch.bool), return_observations=False)
    assert all(s['mirror'] for s in _steps(log)[env2_log_start + 1:])


def test_changing_the_observation_mode_writes_a_lazy_mirror_out_first(env_and_log):
    """in the lazy form `envs` lag behind; a mode change replaces (or drops) the mirror, so the old one has to be
    written out to `envs` while the call block still names it — otherwise the batch rolls back to the last written state"""
    env, log = env_and_log
    rec = env._rec
    for _ in range(3):
        _, _, d, _ = _step(env)
        env.reset(d, return_observations=False)
    assert all(s['lazy'] and s['mirror'] for s in _steps(log)) and rec.flushes == []
    env.observation_mode = 'partial_3'           # (the recorder serves every mode: a NEW mirror is made)
    n = len(log)
    _step(env)
    assert rec.flushes == [n], 'the lazy mirror was not written out before the step of the new mode'
    assert _steps(log)[-1]['mirror'] and not _steps(log)[-1]['mirror_valid']   # the new mirror starts stale
    # ... and when the new mode has no mirror at all
    for _ in range(2):
        _, _, d, _ = _step(env)
        env.reset(d, return_observations=False)
    assert _steps(log)[-1]['lazy'] and _steps(log)[-1]['mirror_valid']
    rec.mirror_bytes_per_env = 0
    env.observation_mode = 'default'
    n = len(log)
    _step(env)
    assert rec.flushes == [3, n] and not _steps(log)[-1]['mirror']


def _make(monkeypatch_env, **kw):
    from wurm_amd.envs import SingleSnake
    return SingleSnake(num_envs=8, size=9, observation_mode='partial_2', device='cpu', seed=5, **kw)


def test_resident_mirror_keyword_and_mirror_state(env_and_log):
    env, log = env_and_log          # (the fixture's patches are in force for the envs made here)
    rec = env._rec
    assert env.mirror_state()['state'] == 'off' and env.mirror_state()['policy'] is None
    _step(env)
    st = env.mirror_state()
    assert st['state'] == 'lazy' and st['current'] and st['why'] == 'on' and st['bytes'] == 8 * 32

    off = _make(None, resident_mirror=False)
    a = len(log)
    _step(off); _step(off)
    assert not any(s['mirror'] for s in _steps(log[a:])) and off.mirror_state() == {
        'state': 'off', 'why': 'resident_mirror=False', 'policy': False, 'current': False, 'bytes': 0}

    # True / 'lazy': asked for explicitly — sized without the batch threshold, and none of the adaptive rules applies
    forced = _make(None, resident_mirror=True)
    before = rec.size_calls
    a = len(log)
    for t in range(12):
        _step(forced)
        forced.reset(torch.ones(8, dtype=torch.bool), return_observations=False)   # eager resets after every step
    assert rec.size_calls == before + 1
    assert all(s['mirror'] and s['lazy'] for s in _steps(log[a:])), 'the adaptive switch-off applied under a keyword'
    forced._observe('default'); _step(forced); forced._observe('default'); _step(forced)
    assert _steps(log)[-1]['lazy'], "'second look ends the lazy form' applied under a keyword"
    assert forced.mirror_state()['state'] == 'lazy' and forced.mirror_state()['policy'] == 'lazy'
    e = forced.envs                  # correctness, not a heuristic: a tensor the caller holds ends the lazy form
    _step(forced)
    assert not _steps(log)[-1]['lazy'] and forced.mirror_state()['state'] == 'eager'
    assert 'caller holds' in forced.mirror_state()['why']
    del e

    eager = _make(None, resident_mirror='eager')
    a = len(log)
    _step(eager); _step(eager)
    assert all(s['mirror'] and not s['lazy'] for s in _steps(log[a:])) and eager.mirror_state()['state'] == 'eager'
    with pytest.raises(ValueError):
        _make(None, resident_mirror='sometimes')


def test_the_adaptive_rules_say_why(env_and_log):
    env, log = env_and_log
    for t in range(12):
        _step(env)
        env.reset(torch.ones(8, dtype=torch.bool), return_observations=False)
    st = env.mirror_state()
    assert st['state'] == 'off' and st['why'].startswith('adaptive:')


def test_reference_test_access_patterns_never_step_on_a_stale_mirror(env_and_log):
    """The access patterns of the reference's own SingleSnake tests (tests/test_single_snake_env.py: rebinding `env.envs`
    :54, in-place edits of what `env.envs` returned, `env.envs[...]` reads for env_consistency every step :24-31, reset(done)
    / reset() in between) against the recorder with the mirror FORCED on: no step launch may be told the mirror is current
    (resident_valid = 1) after anything wrote the state some other way, and no lazy launch may follow a write-less look."""
    env, log = env_and_log
    rec = env._rec
    from wurm_amd.envs import SingleSnake
    env = SingleSnake(num_envs=8, size=9, observation_mode='partial_2', device='cpu', seed=5, resident_mirror=True)
    env._rec = rec
    wrote = [False]          # something other than a step launch has written the state since the last step

    def step():
        n = len(log)
        _step(env)
        s = _steps(log[n:])[-1]
        assert not (wrote[0] and s['mirror_valid']), 'a step launch was told the mirror is current after a write'
        wrote[0] = False
        return s

    step(); step()
    # :54 `env.envs = get_test_env(size, 'up').to(DEFAULT_DEVICE)`: rebinding
    env.envs = torch.zeros(8, 3, 9, 9); wrote[0] = True
    s = step()
    assert not s['lazy']                                  # the caller holds that tensor
    # in-place edit through an alias of the handed-in tensor and through what the attribute returns
    env.envs[0, 0, 1, 1] = 1.0; wrote[0] = True
    step()
    e = env.envs
    step()
    e[1, 2, 3, 3] = 2.0; wrote[0] = True
    step()
    del e   # (while the caller holds an alias of the state no reset is postponed: _alias_free)
    # :24-31 / experiments/main.py:212-227: step, consistency on a gathered copy, reset(done)
    for _ in range(4):
        n = len(log)
        _, _, d, _ = _step(env)
        assert not (wrote[0] and _steps(log[n:])[-1]['mirror_valid'])
        wrote[0] = False
        _ = env.envs[~d.squeeze(-1)]                      # a read: eager mirror, nothing to write out, stays current
        env.reset(d)                                      # the step's own flags: postponed, applied by the next launch
    # an eager reset (not the step's flags) and reset() with no argument write the state with another kernel
    env.reset(torch.ones(8, dtype=torch.bool)); wrote[0] = True
    step()
    env.reset()                                           # = the last step's own flags (env.done): postponed as well
    assert step()['pending']
    # .data edits bypass the version counter: the documented hole (DESIGN.md §7 deviation 9) — the mirror stays "current"
    env.envs.data[2, 0, 5, 5] = 1.0
    s = step()
    assert s['mirror_valid']


def test_check_consistency_uses_the_masks_of_the_step_launch(env_and_log, monkeypatch):
    """SingleSnake.check_consistency(mask) — experiments/main.py:214-215 checks the live envs every step — is served by the
    masks the resident step launch writes (wurm_single_call.check_mask) whenever they describe the current state; anything
    else (first call, a look at / edit of the state in between, an env the launch could not vouch for) runs the checker"""
    env, log = env_and_log
    import wurm_amd.utils as U
    from wurm_amd.envs import SingleSnake
    env = SingleSnake(num_envs=8, size=9, observation_mode='partial_2', device='cpu', seed=5, resident_mirror=True)
    ran = []
    monkeypatch.setattr(U, 'consistency_mask', lambda e: (ran.append(1), torch.zeros(e.shape[0], dtype=torch.int32))[1])
    _, _, d, _ = _step(env)
    live = ~d.squeeze(-1)
    env.check_consistency(live)
    assert len(ran) == 1 and env._c.check_mask            # no masks yet: the checker ran, the request is armed
    env.reset(d)
    _, _, d, _ = _step(env)
    env._chk.zero_()                                       # what the launch writes for consistent live envs
    env.check_consistency(~d.squeeze(-1))
    assert len(ran) == 1                                   # served by the launch's masks
    env.reset(d)                                           # (the first reset after an undisturbed step asks for obs_after)
    _, _, d, _ = _step(env)
    env._chk.zero_()
    env.check_consistency(~d.squeeze(-1))
    assert len(ran) == 1
    assert env.reset(d) is not None and env._fs.pending    # postponed: the masks still apply (rebuilt envs are consistent)
    env._chk[3] = -1
    env._pend[3] = 1
    env.check_consistency()
    assert len(ran) == 1
    env._pend[3] = 0
    _, _, d, _ = _step(env)
    env._chk.zero_(); env._chk[2] = -1                     # an env the launch could not vouch for, and it is asked about
    env.check_consistency(torch.ones(8, dtype=torch.bool))
    assert len(ran) == 2
    env.check_consistency(torch.tensor([1, 1, 0, 1, 1, 1, 1, 1], dtype=torch.bool))   # ... not asked about: fine
    assert len(ran) == 2
    env._chk[5] = U.CHK_ONE_FOOD
    with pytest.raises(RuntimeError, match='exactly one food'):
        env.check_consistency(torch.ones(8, dtype=torch.bool) & (torch.arange(8) != 2))
    env._chk.zero_()
    _ = env.envs                                           # the caller looked at the state (and may have edited it)
    env.check_consistency()
    assert len(ran) == 3
    _step(env)
    env._chk.zero_()
    env.check_consistency()
    assert len(ran) == 3                                   # a new step: its masks are good again
    e = env.envs
    _step(env)
    env._chk.zero_()
    e[0, 0, 1, 1] = 1.0                                    # in-place edit through an alias after the step
    env.check_consistency()
    assert len(ran) == 4
