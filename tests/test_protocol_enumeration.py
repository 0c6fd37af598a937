"""The host protocols (deferred reset, resident mirror, attribute changes) by BOUNDED-EXHAUSTIVE ENUMERATION on the CPU:
every caller event sequence up to a length, object under test against a twin with `lazy_reset=False,
resident_mirror=False`, on the simulating stand-in of the library (tests/protocol_sim.py, tests/protocol_enum.py).

Here: every sequence of length <= 3 for each class x mirror policy (Python step machine; the C machines on the automatic
policy), plus the longer sequences that found something once.  tools/protocol_enumerate.py runs the long lengths over all
cores; its record is profiles/r05_protocol_enumeration.json.
"""
import itertools

import pytest

from tests import protocol_enum as pe
from tests.test_host_lazy_reset import _have_c_stepper, _torchinfo_addresses


def _machine_or_skip(machine):
    if machine != 'python' and not _have_c_stepper():
        pytest.skip('wurm_amd/_fastcall is not built')
    if machine == 'c+torchinfo' and _torchinfo_addresses() is None:
        pytest.skip('wurm_amd/libwurm_torchinfo.so is not built')


def _run_all(make, events, max_len, extra=()):
    bad, n = [], 0
    seqs = itertools.chain(*(itertools.product(events, repeat=L) for L in range(1, max_len + 1)), extra)
    for seq in seqs:
        r = pe.run_sequence(make, lambda: make(True), seq)
        n += 1
        if r is not None and r is not pe.SKIP:
            bad.append(r)
    assert not bad, '%d of %d sequences differ from the twin; first: %s' % (len(bad), n, bad[:5])


# sequences (length 4-6) that exposed a hole at some point — VERDICT r04's two repros first
SINGLE_REGRESSIONS = [
    ('step', 'reset_d', 'step', 'mode', 'reset_d'),              # pre-computed reset observation of the OLD mode
    ('step', 'reset_d', 'step', 'mode', 'reset_view'),
    ('step', 'reset_d', 'step', 'reset_d', 'mode', 'step'),
    ('look', 'step', 'reset_d_noobs', 'edit_alias'),              # alias across a postponed reset (was deviation 9)
    ('step', 'look', 'reset_d_noobs', 'edit_alias', 'step'),
    ('step', 'reset_d', 'step', 'look', 'edit_alias', 'reset_d'),
    ('step', 'reset_d_noobs', 'lazy', 'step', 'reset_d', 'step'),
    ('step', 'check', 'step', 'reset_d_noobs', 'check', 'step'),
    ('step', 'reset_d', 'rollout', 'step', 'reset_d', 'look'),
    ('step', 'edit_done', 'reset_d', 'step', 'reset_d', 'step'),
    ('assign', 'step', 'reset_d', 'step', 'reset_d', 'observe'),
]
GRID_REGRESSIONS = [
    ('step', 'reset_d_noobs', 'start'),                          # start_location assigned after the reset was postponed
    ('step', 'reset_d_noobs', 'start', 'look'),
    ('step', 'reset_d', 'step', 'start', 'reset_d', 'step'),
    ('step', 'reset_d', 'step', 'mode', 'reset_d'),
]
MULTI_REGRESSIONS = [
    ('step', 'reset_d', 'step', 'mode', 'reset_d'),              # 'full' -> 'partial_n' between step and reset
    ('step', 'reset_d_noobs', 'respawn'),                        # dynamics attribute assigned after the reset was postponed
    ('step', 'reset_d_noobs', 'food_mode', 'step'),
    ('step', 'reset_d', 'step', 'respawn', 'reset_d', 'step'),
    ('look', 'step', 'reset_d_noobs', 'edit_alias'),
    ('step', 'look', 'reset_d_noobs', 'edit_alias', 'step'),
    ('step', 'check', 'step', 'reset_d_noobs', 'check', 'step'),
    ('step', 'reset_d', 'rollout', 'step', 'reset_d', 'look'),
]


@pytest.mark.parametrize('mirror,machine,keeps', [(False, 'python', True), ('lazy', 'python', True), ('eager', 'python', True),
                                                  (None, 'python', True), (None, 'c', True), (None, 'c+torchinfo', True),
                                                  ('lazy', 'c+torchinfo', True), ('lazy', 'python', False),
                                                  (None, 'c+torchinfo', False)])
def test_single_snake_every_sequence_up_to_3(monkeypatch, mirror, machine, keeps):
    """(keeps — round 6: `rollout` on the mirror, wurm_single_rollout_resident: grids of 12 x 12 and larger keep the mirror current;
    False: the library's fallback for 9 x 9 — a lazy mirror written out first, the mirror stale afterwards)"""
    _machine_or_skip(machine)
    sim = pe.install_single(monkeypatch, 'single', machine)
    sim.rollout_serves_mirror = keeps
    _run_all(lambda twin=False: pe.make_single('single', mirror, twin), pe.SingleDriver.EVENTS, 3, SINGLE_REGRESSIONS)


@pytest.mark.parametrize('mirror,refuse,machine', [(False, False, 'python'), (False, False, 'c'), (False, False, 'c+torchinfo'),
                                                   (None, False, 'python'), ('lazy', False, 'python'), ('eager', False, 'python'),
                                                   ('lazy', False, 'c+torchinfo'), (None, True, 'python'), ('lazy', True, 'c+torchinfo')])
def test_gridworld_every_sequence_up_to_3(monkeypatch, mirror, refuse, machine):
    """(round 6: SimpleGridworld's mirror — wurm_grid_resident_bytes; refuse: every launch that builds it reports envs outside
    the lane kernel's domain, resident_valid == 2, and the planes stay the state)"""
    _machine_or_skip(machine)
    sim = pe.install_single(monkeypatch, 'grid', machine)
    sim.refuse_builds = refuse
    _run_all(lambda twin=False: pe.make_single('grid', mirror, twin), pe.GridDriver.EVENTS, 3, GRID_REGRESSIONS)


@pytest.mark.parametrize('mirror,keep,machine', [(False, True, 'python'), ('lazy', True, 'python'), ('eager', True, 'python'),
                                                 (None, True, 'python'), ('lazy', False, 'python'), (None, False, 'python'),
                                                 (None, True, 'c'), (None, True, 'c+torchinfo'), ('lazy', False, 'c+torchinfo')])
def test_multi_snake_every_sequence_up_to_3(monkeypatch, mirror, keep, machine):
    _machine_or_skip(machine)
    pe.install_multi(monkeypatch, rollout_keeps_mirror=keep, machine=machine)
    _run_all(lambda twin=False: pe.make_multi(mirror, twin), pe.MultiDriver.EVENTS, 3, MULTI_REGRESSIONS)


def _with_extra(events, extra, max_len, first=None):
    """every sequence over events + extra of length <= max_len that contains at least one of `extra` (length max_len: only
    those that start with `first`, if given — most of the extra events need a step in front of them to mean anything)"""
    allev = tuple(events) + tuple(extra)
    xs = set(extra)
    for L in range(1, max_len + 1):
        for seq in itertools.product(allev, repeat=L):
            if L == max_len and first is not None and seq[0] != first:
                continue
            if xs.intersection(seq):
                yield seq


# VERDICT r05's probe — caller events outside the base alphabet (edit through a temporary, dropped alias, held slice,
# deepcopy / pickle mid-sequence, cleared / reshaped / assigned done, int32 and strided actions; MultiSnake: assigned
# orientations / heads, edited dones, dynamics attributes, separately allocated actions) — and round 6's reset-observation
# events (kept / dropped / read later).  Here: length <= 2 whole, length 3 behind a step; tools/protocol_enumerate.py
# --extra runs length 4 (profiles/r06_protocol_enumeration_extra.json).
@pytest.mark.parametrize('kind,mirror,machine', [('single', None, 'python'), ('single', 'lazy', 'c+torchinfo'),
                                                 ('grid', None, 'python'), ('grid', 'lazy', 'c+torchinfo')])
def test_extra_events_single(monkeypatch, kind, mirror, machine):
    _machine_or_skip(machine)
    pe.install_single(monkeypatch, kind, machine)
    drv = pe.SingleDriver if kind == 'single' else pe.GridDriver
    seqs = list(_with_extra(drv.EVENTS, drv.EXTRA, 3, first='step'))
    _run_all(lambda twin=False: pe.make_single(kind, mirror, twin), (), 0, seqs)


MULTI_EXTRA_REGRESSIONS = [
    ('step', 'reset_d_keep', 'step', 'read_kept'),               # held across the next step: filled in front of it
    ('step', 'reset_d_keep', 'mode', 'read_kept'),               # the mode the reset was called in
    ('step', 'reset_d_keep', 'look', 'edit_alias', 'read_kept'),
    ('step', 'reset_d_keep', 'rollout', 'read_kept'),
    ('step', 'reset_d_keep', 'deepcopy', 'step', 'read_kept'),
    ('step', 'reset_d_drop', 'step', 'reset_d_keep', 'reset_none', 'read_kept'),
    ('step', 'reset_d_keep', 'respawn', 'step', 'read_kept'),
    ('step', 'reset_d_drop', 'step', 'reset_d_drop', 'step', 'reset_d_drop', 'step', 'reset_d_drop', 'step', 'reset_d_keep',
     'step', 'read_kept', 'reset_d_keep', 'step', 'read_kept'),  # the adaptive rule itself: into the lazy form and out again
]


@pytest.mark.parametrize('mirror,lazy_obs,machine', [(None, False, 'python'), (None, True, 'python'), ('lazy', True, 'python'),
                                                     (None, True, 'c+torchinfo'), (False, True, 'c')])
def test_extra_events_multi(monkeypatch, mirror, lazy_obs, machine):
    _machine_or_skip(machine)
    pe.install_multi(monkeypatch, rollout_keeps_mirror=True, machine=machine)
    seqs = list(_with_extra(pe.MultiDriver.EVENTS, pe.MultiDriver.EXTRA, 3, first='step')) + MULTI_EXTRA_REGRESSIONS
    _run_all(lambda twin=False: pe.make_multi(mirror, twin, lazy_obs), (), 0, seqs)


def test_the_harness_sees_a_wrong_counter_and_a_stale_mirror(monkeypatch):
    """the simulator must make protocol mistakes VISIBLE: a deferred reset applied with another counter, and a step on a
    mirror that was not invalidated after a foreign write, both change what the caller sees"""
    import torch
    from tests import protocol_sim as ps
    sim = pe.install_single(monkeypatch, 'single', 'python')

    def broken_counter(twin=False):
        d = pe.make_single('single', False, twin)
        if not twin:
            fs = d.env._fs
            orig = fs.fn
            fs.fn = lambda blk, sl, i, a, dt, call, pend, pre, want, st: orig(blk, sl, i, a, dt, call, pend, pre + (1 if pend else 0), want, st)
        return d
    long = ('step', 'reset_d_noobs') * 3 + ('step', 'look')
    assert pe.run_sequence(lambda: pe.make_single('single', False), lambda: pe.make_single('single', False, True), long) is None
    r = pe.run_sequence(broken_counter, lambda: broken_counter(True), long)
    assert r is not None and r is not pe.SKIP

    def stale_mirror(twin=False):
        d = pe.make_single('single', 'lazy', twin)
        if not twin:
            d.env._touch = lambda: None             # foreign writes no longer invalidate the mirror
        return d
    seq = ('step', 'reset_other', 'step', 'reset_other', 'step', 'look')
    assert pe.run_sequence(lambda: pe.make_single('single', 'lazy'), lambda: pe.make_single('single', 'lazy', True), seq) is None
    r = pe.run_sequence(stale_mirror, lambda: stale_mirror(True), seq)
    assert r is not None and r is not pe.SKIP
    assert isinstance(sim, ps.SimSingle) and torch is not None


def test_the_harness_sees_a_reset_observation_that_was_not_filled_in_time(monkeypatch):
    """round 6 (_LazyResetObs): what reset(done) returned must be filled in front of the next state change while the caller
    holds it — and the enumeration above must be able to tell when it is not"""
    from wurm_amd.envs import multi_snake as ms
    pe.install_multi(monkeypatch, True, 'python')
    d = pe.make_multi(None, False, True)
    d.step(); d.reset_d_keep()
    assert type(d.kept) is ms._LazyResetObs and d.kept._env is not None
    d.step()
    assert d.kept._env is None and len(dict.keys(d.kept)) == pe.M_K and not d.env._lazy_obs_mode
    seq = ('step', 'reset_d_keep', 'step', 'read_kept')
    assert pe.run_sequence(lambda: pe.make_multi(None, False, True), lambda: pe.make_multi(None, True), seq) is None
    monkeypatch.setattr(ms.MultiSnake, '_reset_obs_bookkeeping', lambda self: None)
    r = pe.run_sequence(lambda: pe.make_multi(None, False, True), lambda: pe.make_multi(None, True), seq)
    assert r is not None and r is not pe.SKIP
