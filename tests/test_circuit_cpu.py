"""Level scheduling of NAND netlists (host logic), checked with the oracle as the gate evaluator
at toy parameters."""
import numpy as np
import pytest

import oracle_py as O
from tfhe_amd import circuit as C


def test_adder_shape_and_levels():
    gates, n_wires, a, b, zero, sums, cout = C.ripple_carry_adder(32)
    assert len(gates) == 288 and n_wires == 65 + 288
    levels = C.levelize(gates, n_wires)
    assert sum(len(l) for l in levels) == 288
    assert [len(l) for l in levels[:3]] == [32, 64, 32]          # carry-independent prefix
    assert len(levels) == 3 + 2 * 32 + 1                           # two levels per carry, one for the last sum
    assert max(len(l) for l in levels) == 64
    # every gate's operands are produced at an earlier level
    produced = {}
    for li, lv in enumerate(levels):
        for g in lv:
            produced[gates[g][0]] = li
    for li, lv in enumerate(levels):
        for g in lv:
            for w in gates[g][1:]:
                assert produced.get(w, -1) < li


def test_levelize_rejects_loops_and_double_drivers():
    with pytest.raises(ValueError):
        C.levelize([(2, 0, 3), (3, 1, 2)], 4)
    with pytest.raises(ValueError):
        C.levelize([(2, 0, 1), (2, 1, 0)], 3)


def test_plaintext_adder_semantics():
    gates, n_wires, a, b, zero, sums, cout = C.ripple_carry_adder(8)
    rs = np.random.RandomState(1)
    for _ in range(20):
        x, y = int(rs.randint(0, 256)), int(rs.randint(0, 256))
        w = np.zeros((n_wires, 1), dtype=np.uint32)
        for i in range(8):
            w[a[i], 0], w[b[i], 0] = (x >> i) & 1, (y >> i) & 1
        C.evaluate(gates, w, lambda p, q: (1 - (p & q)).astype(np.uint32))
        got = sum(int(w[s, 0]) << i for i, s in enumerate(sums)) + (int(w[cout, 0]) << 8)
        assert got == x + y


def test_encrypted_adder_with_oracle_evaluator():
    p = O.params_from(10, 16, 2, 8, 2, 5, 2)
    s0, sr, bk, ks = O.keyset(p, seed=5)
    ctx = O.Context(p, bk, ks)
    bits = 4
    gates, n_wires, a, b, zero, sums, cout = C.ripple_carry_adder(bits)
    r = O.rng(8)
    for x, y in [(5, 9), (15, 15), (0, 0), (7, 8)]:
        w = np.zeros((n_wires, p.n + 1), dtype=np.uint32)
        w[a] = O.encrypt_bits([(x >> i) & 1 for i in range(bits)], s0, p, r)
        w[b] = O.encrypt_bits([(y >> i) & 1 for i in range(bits)], s0, p, r)
        w[zero] = C.trivial_bit(0, p.n + 1)
        n_levels = C.evaluate(gates, w, lambda u, v: ctx.nand_batch(u, v, threads=2))
        assert n_levels == 3 + 2 * bits + 1
        out_bits = O.decrypt_bits(w[sums + [cout]], s0, p)
        assert sum(int(v) << i for i, v in enumerate(out_bits)) == x + y
