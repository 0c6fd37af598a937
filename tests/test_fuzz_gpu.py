"""A short, seeded slice of tools/fuzz_parity.py in the regular GPU suite: random shapes / modes / dynamics / seeds,
HIP (C ABI) vs oracle, bit-exact on every output (the full fuzz runs for minutes: `python tools/fuzz_parity.py`)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


@pytest.mark.parametrize('kind,seed,cases', [('single', 101, 40), ('grid', 102, 15), ('multi', 103, 40), ('lean', 104, 40),
                                             ('policy', 105, 20)])
def test_random_cases(kind, seed, cases):
    import fuzz_parity
    rng = np.random.RandomState(seed)
    fn = getattr(fuzz_parity, 'fuzz_' + kind)
    for _ in range(cases):
        fn(rng)
