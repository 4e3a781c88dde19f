#!/usr/bin/env python3
"""Generates tests/golden/stat_goldens.json -- independent known answers for the
floating-point statistics of the hot path (chi-square, odds ratio, p-values,
Fisher's exact test, Hardy-Weinberg), computed WITHOUT the oracle or the
product: scipy / mpmath / exact rational arithmetic.

The reference (opencb/hpg-variant) has no test for these functions
(SURVEY.md section 4), so these vectors are what pins the oracle's restatement of
assoc_basic_test.c:23-41,58-61 and the unpinned third-party pieces.

Run:  python3 tests/golden/make_goldens.py     (needs scipy + mpmath; offline)
"""
import json
import math
import os
import random
from fractions import Fraction

import mpmath as mp
from scipy import stats

mp.mp.dps = 50
rng = random.Random(20261003)


def chi2_case(a, b, c, d):
    """2x2 table rows = (affected, unaffected), cols = (allele1, allele2):
    a=A1, b=U1, c=A2, d=U2 in assoc_basic_test(a,b,c,d) naming."""
    n = a + b + c + d
    A, U, R1, R2 = a + c, b + d, a + b, c + d
    out = {"a": a, "b": b, "c": c, "d": d}
    if 0 in (A, U, R1, R2):
        out["chisq"] = None            # 0/0 -> NaN in the reference
        out["p"] = None
    else:
        x = mp.mpf(0)
        for o, e in ((a, mp.mpf(A) * R1 / n), (c, mp.mpf(A) * R2 / n),
                     (b, mp.mpf(U) * R1 / n), (d, mp.mpf(U) * R2 / n)):
            x += (o - e) ** 2 / e
        out["chisq"] = float(x)
        out["p"] = float(mp.erfc(mp.sqrt(x / 2)))
        # scipy cross-check of the closed form
        sc = stats.chi2_contingency([[a, c], [b, d]], correction=False)
        assert abs(sc[0] - out["chisq"]) <= 1e-9 * max(1.0, out["chisq"])
    # odds ratio as assoc_basic_test.c:58-59 with (A1,A2,U1,U2) = (a,c,b,d)
    out["odds"] = None if (c == 0 or b == 0) else float(Fraction(a, c) * Fraction(d, b))
    return out


def fisher_exact_two_sided(a, b, c, d):
    """table [[a,b],[c,d]]; exact rational arithmetic; ties are exact ties."""
    r1, r2, c1, n = a + b, c + d, a + c, a + b + c + d
    lo, hi = max(0, c1 - r2), min(r1, c1)
    den = math.comb(n, c1)
    w = {x: math.comb(r1, x) * math.comb(r2, c1 - x) for x in range(lo, hi + 1)}
    obs = w[a]
    tot = sum(v for v in w.values() if v <= obs)
    # distance of the closest non-tie on the inclusion side, to flag near-ties
    others = [v for v in w.values() if v > obs]
    near = min((Fraction(v, obs) for v in others), default=None)
    return float(Fraction(tot, den)), (float(near) if near is not None else None)


def hwe_case(n_AA, n_Aa, n_aa):
    n = n_AA + n_Aa + n_aa
    out = {"n_AA": n_AA, "n_Aa": n_Aa, "n_aa": n_aa}
    if n == 0:
        out["chi2"] = None; out["p"] = None
        return out
    p = mp.mpf(2 * n_AA + n_Aa) / (2 * n)
    q = 1 - p
    x = mp.mpf(0)
    for o, e in ((n_AA, p * p * n), (n_Aa, 2 * p * q * n), (n_aa, q * q * n)):
        if e > 0:
            x += (o - e) ** 2 / e
    out["chi2"] = float(x)
    out["p"] = float(mp.erfc(mp.sqrt(x / 2))) if x > 0 else 1.0
    return out


def main():
    chi2 = [chi2_case(30, 20, 10, 40),          # SURVEY 8c probe: 16.666666666666668
            chi2_case(10, 10, 0, 0), chi2_case(0, 0, 0, 0), chi2_case(5, 0, 7, 0),
            chi2_case(1, 1, 1, 1), chi2_case(100000, 99000, 100000, 101000),
            chi2_case(0, 12, 9, 3)]
    for _ in range(60):
        scale = rng.choice([10, 100, 1000, 10000, 100000])
        chi2.append(chi2_case(*[rng.randrange(0, scale + 1) for _ in range(4)]))

    pvals = []
    for x in [1e-300, 1e-12, 1e-6, 0.001, 0.1, 0.5, 0.999, 1.0, 1.001, 2.0, 3.841458820694124,
              10.0, 16.666666666666668, 30.0, 70.0, 100.0, 500.0, 1400.0, 1e4]:
        pvals.append({"x": x, "p": float(mp.erfc(mp.sqrt(mp.mpf(x) / 2)))})
    pvals.append({"x": 0.0, "p": 1.0})
    pvals.append({"x": -1.0, "p": 1.0})             # tdt.c:255,292: chi2 = -1 -> p = 1

    fisher = []
    tables = [(3, 1, 1, 3), (0, 0, 0, 0), (5, 0, 0, 5), (1, 9, 11, 3), (10, 10, 10, 10),
              (0, 7, 7, 0), (12, 0, 0, 0), (100, 200, 150, 120), (2, 3, 4, 5)]
    for _ in range(50):
        scale = rng.choice([5, 20, 100, 400])
        tables.append(tuple(rng.randrange(0, scale + 1) for _ in range(4)))
    for (a, b, c, d) in tables:
        p, near = fisher_exact_two_sided(a, b, c, d)
        # keep only cases where no excluded table is within 1e-6 (relative) of the
        # observed one: near-ties are decided by the (unpinned) tolerance rule.
        if near is not None and near < 1 + 1e-6:
            continue
        sc = stats.fisher_exact([[a, b], [c, d]])[1]
        assert abs(sc - p) < 1e-9, (a, b, c, d, sc, p)
        fisher.append({"a": a, "b": b, "c": c, "d": d, "p": p})

    hwe = [hwe_case(0, 0, 0), hwe_case(10, 0, 0), hwe_case(0, 10, 0), hwe_case(25, 50, 25),
           hwe_case(298, 489, 213), hwe_case(1469, 138, 5)]
    for _ in range(30):
        scale = rng.choice([10, 100, 1000, 50000])
        hwe.append(hwe_case(*[rng.randrange(0, scale + 1) for _ in range(3)]))

    out = {"chi2": chi2, "pvalue": pvals, "fisher": fisher, "hwe": hwe,
           "note": "generated by tests/golden/make_goldens.py (scipy/mpmath/fractions)"}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stat_goldens.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, {k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
