#!/usr/bin/env python3
"""Long live sweep: the CPU oracle against the REAL reference on random configurations (build container only).
Same procedure as tests/test_oracle_vs_live_reference.py with many more seeds.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/fuzz_oracle_vs_reference.py --seconds 600"""
import argparse
import os
import sys
import time
import traceback
import warnings

warnings.filterwarnings('ignore')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
os.environ.setdefault('MPLBACKEND', 'Agg')
import pytest  # noqa: E402
import tests.test_oracle_vs_live_reference as live  # noqa: E402

sys.path.insert(0, live.GOLD)
import make_golden  # noqa: E402
import make_golden_multi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--seconds', type=float, default=300)
ap.add_argument('--first-seed', type=int, default=10000)
args = ap.parse_args()
rec = (make_golden, make_golden_multi)
t0, n, fails, seed = time.time(), dict(single=0, grid=0, multi=0), 0, args.first_seed
while time.time() - t0 < args.seconds and fails < 5:
    for kind, fn in (('single', live.test_single_snake_random_config), ('multi', live.test_multi_snake_random_config),
                     ('grid', live.test_gridworld_random_config), ('multi', live.test_multi_snake_random_config)):
        try:
            fn(rec, seed)
            n[kind] += 1
        except pytest.skip.Exception:
            n['skipped'] = n.get('skipped', 0) + 1
        except RuntimeError as e:
            if 'no available locations' in str(e):  # the reference itself cannot build this (too crowded) config
                n['skipped'] = n.get('skipped', 0) + 1
                seed += 1
                continue
            fails += 1
            print('FAIL', kind, 'seed', seed)
            traceback.print_exc()
        except Exception:
            fails += 1
            print('FAIL', kind, 'seed', seed)
            traceback.print_exc()
        seed += 1
print(f'oracle vs live reference: {n} random configurations, {fails} failures, {time.time() - t0:.0f} s')
