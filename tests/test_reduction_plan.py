"""The compile-time lazy-reduction plans of the inverse transforms, checked from outside the planner.

`tools/gs_plan_dump` (host code, built by `make all`) prints the masks that `make_gs_plan`
(tfhe_amd/csrc/ntt512.hpp) hands to the kernels.  This file does not repeat the planner's decisions: it
takes the masks as given and propagates magnitude bounds through the butterflies in exact rational
arithmetic, asserting what exactness of the FP64 transform needs:

  * every sum and difference of a Gentleman-Sande butterfly stays at or below 2^53 (so it is an exact
    double and a legal operand of the twiddle product);
  * the transform's outputs stay at or below 4 p (far inside the 2^53 the conversion after it can take).

Magnitudes are in units of p = 6432^4 + 1.  A reduction leaves at most p/2 (plus rounding, `EPS`).  A
twiddle product of an operand d by a twiddle |w| <= p/2 with q = rint(d * fl(w/p)) leaves at most
(1/2 + 3/4 |d| / 2^53) p: 1/2 from rint, |d| 2^-55 <= 1/4 |d|/2^53 from the rounding of w/p, and
|d| 2^-54 <= 1/2 |d|/2^53 from the rounding of the product d * fl(w/p).
"""
import pathlib
import subprocess
from fractions import Fraction

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
TOOL = ROOT / "tools" / "gs_plan_dump"

P = 6432**4 + 1
LIMIT = Fraction(2**53, P)  # 5.2627...
EPS = Fraction(1, 10**9)
HALF = Fraction(1, 2) + EPS


def _plans():
    if not TOOL.exists():  # built by `make all`; host code only, a few seconds
        subprocess.run(["make", "-C", str(ROOT), "tools/gs_plan_dump"], check=True, capture_output=True)
    out = subprocess.run([str(TOOL)], check=True, capture_output=True, text=True).stdout
    plans = []
    for line in out.splitlines():
        f = line.split()
        log2n, groups, last = int(f[0]), int(f[1]), int(f[2])
        masks = [int(v) for v in f[3 : 3 + 4 * groups]]
        pre = {groups - 1 - i: masks[3 * i : 3 * i + 3] for i in range(groups)}
        carry = {groups - 1 - i: masks[3 * groups + i] for i in range(groups)}
        plans.append((log2n, groups, last, pre, carry, float(f[3 + 4 * groups])))
    return plans


def _propagate(groups, last, pre, carry):
    """Largest magnitude of any butterfly operand sum, and the bound of the outputs."""
    b = [LIMIT] * 8  # inputs: anything below 2^53
    worst = Fraction(0)
    for g in range(groups - 1, -1, -1):
        stages = last if g == groups - 1 else 3
        for s in range(3):
            if s >= stages:
                assert pre[g][s] == 0, "a mask on a stage the group does not have"
                continue
            for k in range(8):
                if pre[g][s] >> k & 1:
                    b[k] = HALF
            dist = 1 << s
            for lo in range(8):
                if lo & dist:
                    continue
                d = b[lo] + b[lo + dist]
                worst = max(worst, d)
                b[lo] = d
                b[lo + dist] = HALF + Fraction(3, 4) * d / LIMIT
        for k in range(8):
            if carry[g] >> k & 1:
                b[k] = HALF
        if g > 0:
            # The transposition hands a thread values that other threads held in any of their
            # registers (the last group of N = 2048 even mixes two register classes per thread).
            b = [max(b)] * 8
    return worst, max(b)


def test_tool_lists_every_transform_length():
    lengths = [pl[0] for pl in _plans()]
    assert lengths == [9, 9, 10, 11]


@pytest.mark.parametrize("index", range(4))
def test_plan_keeps_every_butterfly_exact(index):
    log2n, groups, last, pre, carry, claimed = _plans()[index]
    assert 3 * (groups - 1) + last == log2n
    assert carry[0] == 0, "nothing follows group 0"
    worst, out = _propagate(groups, last, pre, carry)
    assert worst <= LIMIT, f"N = 2^{log2n}: a butterfly operand reaches {float(worst):.4f} p > 2^53"
    # 4 p up to the rounding allowance of eight reduced values (the plain scheme of N = 1024 is exactly there)
    assert out <= 4 + 8 * EPS, f"N = 2^{log2n}: outputs reach {float(out):.4f} p"
    # the planner works with 5.26 for 2^53 / p and a larger rounding allowance: its own bound is the looser one
    assert float(out) <= claimed + 1e-6


def test_limit_constant_is_conservative():
    # kGsTwo53 in ntt512.hpp
    assert Fraction(526, 100) <= LIMIT


def test_dropping_any_reduction_breaks_a_planned_transform():
    """The plans are minimal in the sense that matters for a regression: clearing any single mask bit
    of a planned transform (not the plain all-eight scheme of N = 1024) violates one of the two limits
    or raises the output bound, so a planner change that loses a reduction cannot pass silently."""
    for log2n, groups, last, pre, carry, _ in _plans():
        if log2n == 10:
            continue
        _, base_out = _propagate(groups, last, pre, carry)
        for g in range(groups):
            for s in range(3):
                for k in range(8):
                    if not pre[g][s] >> k & 1:
                        continue
                    trial = {gg: list(v) for gg, v in pre.items()}
                    trial[g][s] &= ~(1 << k)
                    worst, out = _propagate(groups, last, trial, carry)
                    assert worst > LIMIT or out > base_out, (log2n, g, s, k)
