"""Searches small inputs on which a reading of the Scala source and a plausible alternative reading give different
answers (the inputs frozen in tests/test_reading_cpu.py and listed in oracle/READING.md).  Pure Python, exact
arithmetic through fractions where an alternative needs a fused operation."""
import math
import os
import random
import sys
from fractions import Fraction

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyref as P


def alloc_with(total, weights, sum_fn=None, raw_fn=None):
    n = len(weights)
    sum_w = (sum_fn or P.seq_sum)(weights)
    if total <= 0 or sum_w <= 0.0:
        return [0] * n
    raw = [(raw_fn or (lambda w, s, t: (w / s) * t))(w, sum_w, total) for w in weights]
    base = [P.d2i(math.floor(r)) for r in raw]
    rem = total - sum(base)
    order = sorted(range(n), key=lambda j: raw[j] - math.floor(raw[j]), reverse=True)
    k = 0
    while rem > 0 and k < n:
        base[order[k]] += 1
        rem -= 1
        k += 1
    return base


def pairwise(xs):
    xs = list(xs)
    if len(xs) == 1:
        return xs[0]
    h = len(xs) // 2
    return pairwise(xs[:h]) + pairwise(xs[h:])


def fma(a, b, c):
    """a * b + c rounded once."""
    x = Fraction(a) * Fraction(b) + Fraction(c)
    return float(x)   # Fraction -> float rounds to nearest, ties to even


def main():
    rnd = random.Random(1)
    # (1) weights.sum is a left-to-right fold: against pairwise summation
    for _ in range(200000):
        n = rnd.choice((3, 4, 5))
        w = [round(rnd.random(), 2) for _ in range(n)]
        t = rnd.randrange(1, 40)
        a, b = alloc_with(t, w), alloc_with(t, w, sum_fn=pairwise)
        if a != b:
            print("fold shape:", t, w, a, b, P.seq_sum(w), pairwise(w))
            break
    # (2) raw = (w / sumW) * total, two roundings: against w * total / sumW
    for _ in range(200000):
        n = rnd.choice((2, 3))
        w = [float(rnd.randrange(1, 30)) for _ in range(n)]
        t = rnd.randrange(1, 60)
        a, b = alloc_with(t, w), alloc_with(t, w, raw_fn=lambda x, s, tt: x * tt / s)
        if a != b:
            print("two roundings:", t, w, a, b)
            break
    # (3) export = floor(S * sumW + 1e-9): a multiply and an add, against one fused operation
    for _ in range(2000000):
        S = rnd.randrange(1, 2000)
        sw = round(rnd.random() * 0.2, 3)
        a = math.floor(S * sw + 1e-9)
        b = math.floor(fma(float(S), sw, 1e-9))
        if a != b:
            print("export fma:", S, sw, a, b, repr(S * sw))
            break
    # (4) 1.0 - pInf - pVax is (1.0 - pInf) - pVax: against 1.0 - (pInf + pVax)
    for _ in range(200000):
        pi, pv = round(rnd.random() * 0.5, 3), round(rnd.random() * 0.1, 3)
        s0 = rnd.randrange(1, 5000)
        w1, w2 = max(0.0, 1.0 - pi - pv), max(0.0, 1.0 - (pi + pv))
        a, b = P.allocate_int(s0, [w1, pi, pv]), P.allocate_int(s0, [w2, pi, pv])
        if a != b:
            print("left assoc:", s0, pi, pv, a, b, repr(w1), repr(w2))
            break


if __name__ == "__main__":
    main()
