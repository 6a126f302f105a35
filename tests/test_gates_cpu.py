"""Two-input gates other than NAND, NOT and MUX on the CPU oracle (toy parameters, seconds), and the
netlist forms that use them.  The reference has NAND only (src/hom_nand.hpp:74-81): these gates are
pinned by their truth tables under decryption and, for NAND, by the reference golden vectors."""
import itertools

import numpy as np
import pytest

import oracle_py as O
from tfhe_amd import binding as B
from tfhe_amd import circuit as C


@pytest.fixture(scope="module")
def toy():
    p = O.params_from(10, 16, 2, 8, 2, 5, 2)
    s0, sr, bk, ks = O.keyset(p, seed=21)
    return p, s0, O.Context(p, bk, ks)


def test_gate_tables_match_between_oracle_and_binding():
    assert O.GATES == B.GATES


def test_nand_code_is_the_reference_combination(golden):
    g = golden("combine636")
    p = O.params_default()
    for a, b, c in zip(g["a"], g["b"], g["c"]):
        assert np.array_equal(O.gate_combine("nand", a, b, p), c)
        assert np.array_equal(O.gate_combine(0, a, b, p), O.nand_combine(a, b, p))


def test_unknown_gate_code_rejected():
    p = O.params_default()
    z = np.zeros(p.n + 1, dtype=np.uint32)
    with pytest.raises(ValueError):
        O.gate_combine(10, z, z, p)


@pytest.mark.parametrize("op", sorted(O.GATES))
def test_truth_table(toy, op):
    p, s0, ctx = toy
    r = O.rng(3)
    abits = np.array([0, 0, 1, 1] * 4)
    bbits = np.array([0, 1, 0, 1] * 4)
    a, b = O.encrypt_bits(abits, s0, p, r), O.encrypt_bits(bbits, s0, p, r)
    out = ctx.gate_batch(op, a, b)
    assert np.array_equal(O.decrypt_bits(out, s0, p), O.GATE_TRUTH[op](abits, bbits))


def test_mixed_codes_equal_uniform_calls(toy):
    p, s0, ctx = toy
    r = O.rng(4)
    n = 20
    rs = np.random.RandomState(0)
    a, b = O.encrypt_bits(rs.randint(0, 2, n), s0, p, r), O.encrypt_bits(rs.randint(0, 2, n), s0, p, r)
    ops = rs.randint(0, 10, n).astype(np.uint8)
    got = ctx.gate_batch(ops, a, b)
    for g in range(n):
        assert np.array_equal(got[g], ctx.gate_batch(int(ops[g]), a[g:g + 1], b[g:g + 1])[0])


def test_not_and_mux(toy):
    p, s0, ctx = toy
    r = O.rng(5)
    combos = np.array(list(itertools.product([0, 1], repeat=3)) * 2)
    sel, x1, x0 = (O.encrypt_bits(combos[:, i], s0, p, r) for i in range(3))
    out = ctx.mux_batch(sel, x1, x0)
    assert np.array_equal(O.decrypt_bits(out, s0, p), np.where(combos[:, 0] == 1, combos[:, 1], combos[:, 2]))
    assert np.array_equal(O.decrypt_bits((0 - sel.astype(np.int64)).astype(np.uint32), s0, p), 1 - combos[:, 0])


class OracleBackend:
    def __init__(self, ctx):
        self.ctx = ctx

    def gates(self, ops, a, b):
        return self.ctx.gate_batch(np.asarray(ops, dtype=np.uint8), a, b)

    def not_(self, a):
        return (0 - a.astype(np.int64)).astype(np.uint32)

    def mux(self, sel, x1, x0):
        return self.ctx.mux_batch(sel, x1, x0)


def test_mux_adder_shape():
    gates, n_wires, a, b, zero, sums, cout = C.mux_adder(32)
    levels = C.levelize(gates, n_wires)
    assert len(gates) == 96 and len(levels) == 32 + 1        # xor level, then one mux (+ the previous sum) per bit
    assert len(levels[0]) == 32


def test_plaintext_semantics_of_every_gate_kind():
    gates = [(3, 0, 1, "xor"), (4, 3, None, "not"), (5, 2, 3, "mux", 4), (6, 5, 0, "andyn"), (7, 6, 1)]
    for bits in itertools.product([0, 1], repeat=3):
        w = np.zeros((8, 1), dtype=np.uint32)
        w[:3, 0] = bits

        class Plain:
            def gates(self, ops, a, b):
                names = {v: k for k, v in B.GATES.items()}
                return np.stack([O.GATE_TRUTH[names[int(o)]](x, y) for o, x, y in zip(ops, a, b)]).astype(np.uint32)

            def not_(self, a):
                return 1 - a

            def mux(self, s, x1, x0):
                return np.where(s == 1, x1, x0)

        C.evaluate(gates, w, Plain())
        x = bits[0] ^ bits[1]
        m = x if bits[2] else 1 - x
        g6 = m & (1 - bits[0])
        assert list(w[3:, 0]) == [x, 1 - x, m, g6, 1 - (g6 & bits[1])]


def test_encrypted_mux_adder_with_oracle(toy):
    p, s0, ctx = toy
    bits = 4
    gates, n_wires, a, b, zero, sums, cout = C.mux_adder(bits)
    r = O.rng(9)
    for x, y in [(5, 9), (15, 15), (0, 0), (6, 11)]:
        w = np.zeros((n_wires, p.n + 1), dtype=np.uint32)
        w[a] = O.encrypt_bits([(x >> i) & 1 for i in range(bits)], s0, p, r)
        w[b] = O.encrypt_bits([(y >> i) & 1 for i in range(bits)], s0, p, r)
        w[zero] = C.trivial_bit(0, p.n + 1)
        C.evaluate(gates, w, OracleBackend(ctx))
        got = sum(int(v) << i for i, v in enumerate(O.decrypt_bits(w[sums], s0, p)))
        got += int(O.decrypt_bits(w[[cout]], s0, p)[0]) << bits
        assert got == x + y


def test_loop_detection_still_works():
    with pytest.raises(ValueError):
        C.levelize([(2, 0, 3, "xor"), (3, 1, 2, "mux", 0)], 4)
    with pytest.raises(ValueError):
        C.normalize((1, 0, 0, "frob"))


def test_prefix_adder_shape_and_plaintext_sum():
    """The parallel-prefix adder: log2(bits) + 2 levels, no dead gates, and the sum of every pair of a
    random sample of operands (plaintext evaluation of the netlist)."""
    gates, n_wires, a, b, zero, sums, cout = C.prefix_adder(32)
    levels = C.levelize(gates, n_wires)
    assert len(levels) == 5 + 2
    assert len(C.prune(gates, sums + [cout])) == len(gates)
    names = {v: k for k, v in B.GATES.items()}

    class Plain:
        def gates(self, ops, x, y):
            return np.stack([O.GATE_TRUTH[names[int(o)]](u, v) for o, u, v in zip(ops, x, y)]).astype(np.uint32)

        def mux(self, s, x1, x0):
            return np.where(s == 1, x1, x0)

    rs = np.random.RandomState(4)
    cases = [(0, 0), (2**32 - 1, 1), (2**32 - 1, 2**32 - 1), (0x55555555, 0xAAAAAAAA), (0x0000FFFF, 1)]
    cases += [(int(rs.randint(0, 2**32, dtype=np.uint64)), int(rs.randint(0, 2**32, dtype=np.uint64))) for _ in range(20)]
    for x, y in cases:
        w = np.zeros((n_wires, 1), dtype=np.uint32)
        for i in range(32):
            w[a[i], 0], w[b[i], 0] = (x >> i) & 1, (y >> i) & 1
        C.evaluate(gates, w, Plain())
        got = sum(int(w[s_, 0]) << i for i, s_ in enumerate(sums)) + (int(w[cout, 0]) << 32)
        assert got == x + y, (x, y)


def test_encrypted_prefix_adder_with_oracle(toy):
    p, s0, ctx = toy
    bits = 8
    gates, n_wires, a, b, zero, sums, cout = C.prefix_adder(bits)
    assert len(C.levelize(gates, n_wires)) == 3 + 2
    r = O.rng(10)
    for x, y in [(200, 100), (255, 255), (0, 0), (170, 85)]:
        w = np.zeros((n_wires, p.n + 1), dtype=np.uint32)
        w[a] = O.encrypt_bits([(x >> i) & 1 for i in range(bits)], s0, p, r)
        w[b] = O.encrypt_bits([(y >> i) & 1 for i in range(bits)], s0, p, r)
        C.evaluate(gates, w, OracleBackend(ctx))
        got = sum(int(v) << i for i, v in enumerate(O.decrypt_bits(w[sums], s0, p)))
        got += int(O.decrypt_bits(w[[cout]], s0, p)[0]) << bits
        assert got == x + y
