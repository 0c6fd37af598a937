"""Bounded-exhaustive enumeration of the host protocols (deferred reset, resident mirror, attribute changes) against the
simulating stand-in of the library — no GPU needed.  See tests/protocol_enum.py; the test suite runs the short lengths,
this tool the long ones over all cores and writes a record:

    python tools/protocol_enumerate.py --length 6 --out profiles/r05_protocol_enumeration.json

Every sequence of exactly 1..L events over the alphabet of the class, for each configuration
(class x mirror policy x step machine), is compared with a twin object that defers nothing and mirrors nothing.
"""
import argparse
import itertools
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _configs():
    out = []
    for mirror in (False, 'lazy', 'eager', None):
        for machine in ('python', 'c', 'c+torchinfo'):
            out.append(('single', mirror, machine))
    for machine in ('python', 'c', 'c+torchinfo'):
        out.append(('grid', False, machine))
    for mirror in ('lazy', 'eager', None):                     # round 6: SimpleGridworld's mirror, built valid / refused
        out.append(('grid', mirror, 'python'))
        out.append(('grid', mirror, 'c+torchinfo'))
        out.append(('grid', mirror, 'refuse/c+torchinfo'))
    for machine in ('python', 'c', 'c+torchinfo'):
        for mirror in (False, 'lazy', 'eager', None):
            out.append(('multi', mirror, 'keep/' + machine))
        out.append(('multi', 'lazy', 'nokeep/' + machine))
        out.append(('multi', None, 'nokeep/' + machine))
    return out


def _work(job):
    (kind, mirror, machine), length, prefix, extra = job
    import pytest
    from tests import protocol_enum as pe
    mpatch = pytest.MonkeyPatch()
    try:
        if kind == 'multi':
            keep, mach = machine.split('/')
            pe.install_multi(mpatch, rollout_keeps_mirror=keep != 'nokeep', machine=mach)
            drv = pe.MultiDriver
            make = lambda twin=False: pe.make_multi(mirror, twin, lazy_obs=keep == 'lazyobs')  # noqa: E731
        else:
            sim = pe.install_single(mpatch, kind, machine.split('/')[-1])
            sim.refuse_builds = machine.startswith('refuse/')
            drv = pe.SingleDriver if kind == 'single' else pe.GridDriver
            make = lambda twin=False: pe.make_single(kind, mirror, twin)  # noqa: E731
        events = tuple(drv.EVENTS) + (tuple(drv.EXTRA) if extra else ())
        xs = set(drv.EXTRA)
        n = skipped = 0
        bad = []
        for rest in itertools.product(events, repeat=length - len(prefix)):
            seq = tuple(prefix) + rest
            if extra and not xs.intersection(seq):
                continue   # (--extra: only the sequences that contain an event outside the base alphabet; the rest is the base run)
            r = pe.run_sequence(make, lambda: make(True), seq)
            n += 1
            if r is pe.SKIP:
                skipped += 1
            elif r:
                bad.append(r)
        return (kind, mirror, machine), length, n, skipped, bad
    finally:
        mpatch.undo()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--length', type=int, default=5)
    ap.add_argument('--only', default=None, help='single / grid / multi')
    ap.add_argument('--machines', default=None, help="comma list restricting the step machines, e.g. 'python'")
    ap.add_argument('--procs', type=int, default=os.cpu_count())
    ap.add_argument('--out', default=None)
    ap.add_argument('--extra', action='store_true', help="the events outside the base alphabet (tests/protocol_enum.py: EXTRA): every "
                    'sequence that contains at least one of them; MultiSnake also started in the lazy-observation form')
    args = ap.parse_args()
    from tests import protocol_enum as pe
    jobs = []
    configs = _configs()
    if args.extra:   # MultiSnake as an object whose caller has been dropping what reset(done) returns (_LazyResetObs)
        configs += [('multi', None, 'lazyobs/python'), ('multi', 'lazy', 'lazyobs/c+torchinfo'), ('multi', False, 'lazyobs/c')]
    for cfg in configs:
        if args.only and cfg[0] != args.only:
            continue
        if args.machines and cfg[2].split('/')[-1] not in args.machines.split(','):
            continue
        drv = pe.MultiDriver if cfg[0] == 'multi' else (pe.SingleDriver if cfg[0] == 'single' else pe.GridDriver)
        events = tuple(drv.EVENTS) + (tuple(drv.EXTRA) if args.extra else ())
        for L in range(1, args.length + 1):
            if L <= (2 if args.extra else 3):
                jobs.append((cfg, L, (), args.extra))
            else:  # split by the first two events; a sequence that does not start with a step cannot have deferred anything,
                   # but it can have looked at / assigned the state first: all of them are run
                for p in itertools.product(events, repeat=2):
                    jobs.append((cfg, L, p, args.extra))
    t0 = time.time()
    totals = {}
    failures = []
    with mp.Pool(args.procs) as pool:
        for cfg, L, n, skipped, bad in pool.imap_unordered(_work, jobs, chunksize=1):
            key = '%s mirror=%s %s' % cfg
            t = totals.setdefault(key, {'sequences': 0, 'inapplicable': 0, 'differences': 0, 'max_length': 0})
            t['sequences'] += n
            t['inapplicable'] += skipped
            t['differences'] += len(bad)
            t['max_length'] = max(t['max_length'], L)
            failures.extend(bad[:20])
    rec = {'what': 'tools/protocol_enumerate.py: every caller event sequence up to the length given, object under test vs a twin '
                   'with lazy_reset=False, resident_mirror=False, on the simulating stand-in (tests/protocol_sim.py)',
           'length': args.length, 'seconds': round(time.time() - t0, 1), 'configurations': totals,
           'sequences_total': sum(t['sequences'] for t in totals.values()),
           'differences_total': sum(t['differences'] for t in totals.values()), 'first_differences': failures[:50]}
    print(json.dumps(rec, indent=1))
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(rec, f, indent=1)
    return 1 if failures else 0


if __name__ == '__main__':
    sys.exit(main())
