"""`div255` of wurm_amd/csrc/multi_snake.hip (the pixels of MultiSnake's partial_n observations, reference
wurm/envs/multi_snake.py:194-227 `/ 255`): q = x * fl(1/255); q' = fma(fma(-q, 255, x), fl(1/255), q) is the correctly rounded
IEEE quotient x / 255 for every integer the fast path takes (0 <= x < 70 000).  Checked here in exact rational arithmetic
(no GPU): the fused operations are rounded once, from the exact value."""
import math
from fractions import Fraction

import numpy as np

F32 = np.float32


def _round_f32(fr):
    """nearest float32, ties to even, of an exact rational"""
    if fr == 0:
        return F32(0)
    sign = 1 if fr > 0 else -1
    a = abs(fr)
    e = math.floor(math.log2(a.numerator) - math.log2(a.denominator))
    sh = 23 - e
    m = a * (Fraction(2) ** sh)
    while m >= 2 ** 24:
        m /= 2
        sh -= 1
    while m < 2 ** 23:
        m *= 2
        sh += 1
    n = m.numerator // m.denominator
    rem = m - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return F32(sign * float(n) * (2.0 ** (-sh)))


def test_div255_is_the_ieee_quotient_on_its_whole_domain():
    rc = F32(1.0) / F32(255.0)
    rc_exact = Fraction(float(rc))
    xs = np.arange(0, 70000, dtype=np.float32)
    want = xs / F32(255.0)
    for x in range(0, 70000):
        q = F32(x) * rc                                        # one rounding
        q_exact = Fraction(float(q))
        r = _round_f32(Fraction(x) - q_exact * 255)            # fma(-q, 255, x)
        got = _round_f32(Fraction(float(r)) * rc_exact + q_exact)   # fma(r, rc, q)
        assert got == want[x], (x, got, want[x])
