"""A seeded slice of tools/fuzz_parity.py in the regular GPU suite (about a minute, a few thousand random cases): random
shapes / modes / dynamics / seeds / call patterns, HIP (C ABI) vs oracle, bit-exact on every output.  The long runs
(`python tools/fuzz_parity.py --seconds N --summary profiles/rNN_fuzz_summary.json`) record library hash, seeds, cases
per family and mismatches under profiles/."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


@pytest.mark.parametrize('kind,seed,cases', [('single', 101, 800), ('fused', 106, 700), ('grid', 102, 300), ('multi', 103, 800),
                                             ('lean', 104, 500), ('lane', 107, 1200), ('policy', 105, 150), ('resident', 108, 1000),
                                             ('multi_resident', 109, 500), ('multi_group', 110, 400), ('grid_lane', 111, 600)])
def test_random_cases(kind, seed, cases):
    import fuzz_parity
    rng = np.random.RandomState(seed)
    fn = getattr(fuzz_parity, 'fuzz_' + kind)
    for _ in range(cases):
        fn(rng)
