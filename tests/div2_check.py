"""Division by a known divisor in two operations, and the proof obligation that goes with it.

For a divisor d with zh = RN(1/d), zl = RN(1/d - zh):   q = fma(x, zh, RN(x * zl))
differs from x/d by at most 2^-105 relative before its final rounding, so q = RN(x/d) unless x/d lies within 2^-52 ulp of a
midpoint of two adjacent floats.  For p-bit mantissas X (of x) and D (odd part of d's mantissa, L bits) that means
|2^t X - (2M+1) D| < 2 D 2^-(p-1) for t in {L-1, L}: a handful of residues r, each with at most a few X.  candidates() lists
them; verify() runs the actual two-operation sequence on every one of them (and their neighbours).  This file is the
prototype and the self-test of that argument (test infrastructure: only tests/ import it): toy_exhaustive() replays it in a 10..12-bit floating-point format where ALL
(X, D) pairs can be tried, and checks that every failing pair is among the candidates.  The product's C++ twin is
fastdiv2_ok() in stmpc.hip.  (Brisebarre & Muller, "Correctly rounded multiplication by arbitrary precision constants",
IEEE TC 2008, give the general theory; only the elementary bound above is used here.)"""
from fractions import Fraction
import math, random, sys


def rn(fr, p):
    """Round a positive Fraction to p significant bits, nearest-even; returns a Fraction."""
    if fr == 0:
        return Fraction(0)
    sign = 1
    if fr < 0:
        sign, fr = -1, -fr
    e = fr.numerator.bit_length() - fr.denominator.bit_length()
    if Fraction(2) ** e > fr:
        e -= 1
    # fr in [2^e, 2^(e+1)); scale so that the integer part has p bits
    scale = Fraction(2) ** (p - 1 - e)
    y = fr * scale
    f = y.numerator // y.denominator
    rem = y - f
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (f & 1)):
        f += 1
    return sign * Fraction(f) / scale


def two_op(x, d, p):
    z = Fraction(1) / d
    zh = rn(z, p)
    zl = rn(z - zh, p)
    u1 = rn(x * zl, p)
    return rn(x * zh + u1, p)


def candidates(Dm, p, safety=4):
    """Mantissas X (p bits) for which X / Dm may sit within `safety` x 2^-(p-1) ulp of a midpoint.  Dm: p-bit mantissa of d."""
    D = Dm
    while D % 2 == 0:
        D //= 2
    if D == 1:
        return []
    L = D.bit_length()
    R = (2 * D * safety) >> (p - 1)            # |r| < 2 D 2^-(p-1) x safety
    out = set()
    lo, hi = 1 << (p - 1), 1 << p
    for t in (L - 1, L, L + 1):
        inv = pow(pow(2, t, D), -1, D)
        for r in range(-R - 1, R + 2):
            if r == 0:
                continue
            x0 = (r * inv) % D
            k0 = (lo - x0 + D - 1) // D
            X = x0 + k0 * D
            n = 0
            while X < hi:
                out.add(X); X += D; n += 1
                if n > 4096:
                    return None                # too many to try: the caller falls back to ordinary division
    return sorted(out)


def toy_exhaustive(p):
    """All (X, D) of a p-bit format: every X where the two-operation quotient is wrong must be a candidate."""
    bad = missed = 0
    for Dm in range(1 << (p - 1), 1 << p):
        d = Fraction(Dm)
        c = candidates(Dm, p)
        cs = set(c)
        for X in range(1 << (p - 1), 1 << p):
            x = Fraction(X)
            if two_op(x, d, p) != rn(x / d, p):
                bad += 1
                if X not in cs:
                    missed += 1
    return bad, missed


def verify_double(d):
    """True if q = fma(x, zh, x*zl) == x/d for every double x (no overflow/underflow), by the candidate argument + hardware arithmetic."""
    m, e = math.frexp(d)
    Dm = int(m * (1 << 53))
    c = candidates(Dm, 53)
    if c is None:
        return False
    zh = 1.0 / d
    zl = math.fma(-d, zh, 1.0) / d if hasattr(math, "fma") else None
    if zl is None:
        zl = float(Fraction(1) / Fraction(d) - Fraction(zh))
    def two(x):
        u1 = x * zl
        return float(rn(Fraction(x) * Fraction(zh) + Fraction(u1), 53)) if not hasattr(math, "fma") else math.fma(x, zh, u1)
    for X in c:
        for dX in (-1, 0, 1):
            for sc in (1.0, 2.0 ** -40, 2.0 ** 30):
                x = float(X + dX) * sc
                for s in (x, -x):
                    if two(s) != s / d:
                        return False
    rnd = random.Random(1)
    for _ in range(20000):
        x = rnd.uniform(-1e4, 1e4)
        if two(x) != x / d:
            return False
    return True


if __name__ == "__main__":
    for p in (8, 9, 10):
        print("toy p=%d: (failing, failing but not listed) =" % p, toy_exhaustive(p))
    for d in (0.3, 0.3 * 0.3, 0.3 * 0.3 * 0.3, 0.05, 0.2, 0.04, 0.008, 0.1, 1.0 / 3.0):
        m, e = math.frexp(d)
        c = candidates(int(m * (1 << 53)), 53)
        print("d=%r: %s candidates, verified=%s" % (d, "too many" if c is None else len(c), verify_double(d)))
