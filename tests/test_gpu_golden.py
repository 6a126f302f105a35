"""GPU path, called through the C ABI, against vectors produced by the REAL reference
(tests/golden/*.npz): the parameter-generic exact kernels at toy ring degrees, where the
reference itself is exact (reference src/types.hpp:225-236).  Bit-exact."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu

SETS = ["toyA", "toyB", "toyC"]


def _params(arr):
    import tfhe_amd

    return tfhe_amd.Params(*[int(v) for v in arr])


@pytest.mark.parametrize("name", SETS)
def test_external_product_and_cmux_vs_reference(golden, name):
    # reference src/trgsw.hpp:100-117
    import tfhe_amd

    g = golden(name)
    p = _params(g["params"])
    assert np.array_equal(tfhe_amd.external_product(p, g["ep_trgsw"], g["ep_in"]), g["ep_out"])
    assert np.array_equal(tfhe_amd.cmux(p, g["ep_trgsw"], g["cmux_d0"], g["cmux_d1"]), g["cmux_out"])


@pytest.mark.parametrize("name", SETS)
def test_nand_pipeline_vs_reference(golden, name):
    # reference src/hom_nand.hpp:74-81 and every stage below it
    import tfhe_amd

    g = golden(name)
    p = _params(g["params"])
    with tfhe_amd.Context(p, g["bk"], g["ks"]) as ctx:
        assert np.array_equal(ctx.blind_rotate(g["c"]), g["acc"])
        assert np.array_equal(ctx.gate_bootstrapping_tlwe_to_tlwe(g["c"]), g["lv1"])
        assert np.array_equal(ctx.identity_key_switch(g["lv1"]), g["out"])
        assert np.array_equal(ctx.bootstrap(g["c"]), g["out"])
        assert np.array_equal(ctx.nand(g["a"], g["b"]), g["out"])


@pytest.mark.parametrize("tag", ["t5", "t8", "t3b3", "wide:t4b4", "wide:t6b5", "wide:t3b7", "wide:t1b8"])
def test_identity_key_switch_vs_reference(golden, tag):
    # reference src/tlwe_key_switching.hpp:47-65; context with a key-switching key only.  "wide": BASE_BIT 4 .. 8
    # (the gather kernel; the others run on the tiled kernel)
    import tfhe_amd

    g = golden("kswide" if tag.startswith("wide:") else "ks")
    tag = tag.split(":")[-1]
    n, nlv1, _, _, _, T, bb = [int(v) for v in g[f"{tag}_params"]]
    N = nlv1 & -nlv1  # the key switch only sees N*K = nlv1; pick a power-of-two N dividing it
    p = tfhe_amd.Params(n, N, nlv1 // N, 8, 1, T, bb)
    with tfhe_amd.Context(p, None, g[f"{tag}_ks"]) as ctx:
        assert np.array_equal(ctx.identity_key_switch(g[f"{tag}_in"]), g[f"{tag}_out"])
        with pytest.raises(tfhe_amd.TfheAmdError):
            ctx.bootstrap(np.zeros((1, n + 1), np.uint32))  # no bootstrapping key in this context


def test_blind_rotate_custom_test_vector_vs_reference(golden):
    # reference src/trgsw.hpp:155-158 with n = 0 steps is only the initial rotation
    import tfhe_amd

    g = golden("rot512")
    p = tfhe_amd.Params(1, 512, 2, 8, 2, 5, 2)
    tv = np.zeros((3, 512), dtype=np.uint32)
    tv[0] = np.arange(1, 513)
    # one all-zero TRGSW row: CMUX with a zero selector leaves the accumulator unchanged
    bk = np.zeros(p.bk_shape(), dtype=np.uint32)
    c = np.zeros((len(g["b"]), 2), dtype=np.uint32)
    c[:, 0] = g["b"]
    c[:, 1] = np.arange(len(g["b"])) * 0x01234567
    for generic in (False, True):
        with tfhe_amd.Context(p, bk, None, generic_kernels=generic) as ctx:
            acc = ctx.blind_rotate(c, tv)
        assert np.array_equal(acc[:, 0, :], g["out"]), generic
        assert not acc[:, 1:, :].any()


def test_fast_and_generic_kernels_agree_at_default_parameters():
    """The NTT kernels and the O(N^2) wrap-around kernels are independent implementations."""
    import tfhe_amd

    p = O.params_default()
    rs = np.random.RandomState(9)
    bk = rs.randint(0, 2**32, size=(24,) + p.bk_shape()[1:], dtype=np.uint64).astype(np.uint32)
    p24 = tfhe_amd.Params(24, 512, 2, 8, 2, 5, 2)  # 24 blind-rotate steps keep the O(N^2) path short
    c = rs.randint(0, 2**32, size=(6, 25), dtype=np.uint64).astype(np.uint32)
    tv = rs.randint(0, 2**32, size=(3, 512), dtype=np.uint64).astype(np.uint32)
    with tfhe_amd.Context(p24, bk, None) as fast, tfhe_amd.Context(p24, bk, None, generic_kernels=True) as gen:
        for t in (None, tv):
            assert np.array_equal(fast.blind_rotate(c, t), gen.blind_rotate(c, t))
        assert np.array_equal(fast.gate_bootstrapping_tlwe_to_tlwe(c), gen.gate_bootstrapping_tlwe_to_tlwe(c))
    # and the oracle agrees with both
    op = O.params_from(24, 512, 2, 8, 2, 5, 2)
    with tfhe_amd.Context(p24, bk, None) as fast:
        got = fast.blind_rotate(c, tv)
    for i in range(2):
        assert np.array_equal(got[i], O.blind_rotate(c[i], bk, op, O.NTT, tv=tv))


def test_external_product_default_size_vs_oracle_and_reference_fft(golden):
    """N = 512: exact result equals the oracle; the reference's own floating-point product
    (reference src/types.hpp:237-292) is the exact one minus at most 6 (documented relation)."""
    import tfhe_amd

    g = golden("extprod512")
    p = tfhe_amd.params_default()
    got = tfhe_amd.external_product(p, g["trgsw"], g["in"])
    op = O.params_default()
    for i in range(got.shape[0]):
        assert np.array_equal(got[i], O.external_product(g["trgsw"][i], g["in"][i], op, O.NTT))
    delta = (g["out"].astype(np.int64) - got.astype(np.int64) + (1 << 31)) % (1 << 32) - (1 << 31)
    assert delta.min() >= -6 and delta.max() <= 0


@pytest.mark.parametrize("K", [2, 1])
def test_extreme_operands_stay_exact(K):
    """Worst-case magnitudes for the transform's lazy-reduction bounds: key words at the ends
    of the centred range, accumulators whose rotated difference decomposes to digits of -128
    or +127 everywhere.  NTT kernels (both forms, K = 2 and K = 1) vs the O(N^2) wrap-around kernels vs
    the oracle."""
    import tfhe_amd

    n = 6
    p6 = tfhe_amd.Params(n, 512, K, 8, 2, 5, 2)
    op = O.params_from(n, 512, K, 8, 2, 5, 2)
    rs = np.random.RandomState(21)
    patterns = {
        "min": np.full(p6.bk_shape(), 0x80000000, dtype=np.uint32),
        "max": np.full(p6.bk_shape(), 0x7FFFFFFF, dtype=np.uint32),
        "alt": np.where(rs.randint(0, 2, size=p6.bk_shape()) == 1, 0x80000000, 0x7FFFFFFF).astype(np.uint32),
    }
    # test vectors that make (acc * X^a - acc) hit the digit extremes: +-2^31 steps and 0x7F7F.. runs
    tvs = [np.full((K + 1, 512), 0x80000000, dtype=np.uint32),
           np.tile(np.array([0x7F7F8000, 0x80808000], dtype=np.uint32), (K + 1, 256)),
           rs.randint(0, 2**32, size=(K + 1, 512), dtype=np.uint64).astype(np.uint32)]
    c = rs.randint(0, 2**32, size=(8, n + 1), dtype=np.uint64).astype(np.uint32)
    c[0, 1:] = 0x00400000  # rotate by exactly one position every step
    c[1, 1:] = 0x80000000  # rotate by N: acc * X^N = -acc, difference = -2 acc
    for name, bk in patterns.items():
        with tfhe_amd.Context(p6, bk, None, generic_kernels=True) as gen:
            want = [gen.blind_rotate(c, tv) for tv in tvs]
        for tv, a_gen in zip(tvs, want):
            assert np.array_equal(a_gen[0], O.blind_rotate(c[0], bk, op, O.SCHOOLBOOK, tv=tv)), name
        # both forms of the NTT kernels: a batch this small would only take the workgroup-per-gate one
        for form in ("throughput", "latency"):
            with tfhe_amd.Context(p6, bk, None, form=form) as fast:
                assert fast.kernel_family() == "ntt512"
                for tv, a_gen in zip(tvs, want):
                    assert np.array_equal(fast.blind_rotate(c, tv), a_gen), (name, form)


def test_n2048_ntt_path_vs_generic_and_oracle():
    """The N = 2048 set (K = 1, L = 3): workgroup NTT kernels with the split key vs the O(N^2)
    wrap-around kernels vs the oracle, including key words at the ends of the range (each key
    half then sits at the end of its own range)."""
    import tfhe_amd

    n = 3
    p = tfhe_amd.Params(n, 2048, 1, 8, 3, 8, 2)
    op = O.params_from(n, 2048, 1, 8, 3, 8, 2)
    rs = np.random.RandomState(31)
    keys = [rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32),
            np.full(p.bk_shape(), 0x80000000, dtype=np.uint32),
            np.full(p.bk_shape(), 0x7FFF8000, dtype=np.uint32)]
    c = rs.randint(0, 2**32, size=(5, n + 1), dtype=np.uint64).astype(np.uint32)
    c[0, 1:] = 0x00100000  # rotate by one position every step
    tv = rs.randint(0, 2**32, size=(2, 2048), dtype=np.uint64).astype(np.uint32)
    for bk in keys:
        with tfhe_amd.Context(p, bk, None) as fast, tfhe_amd.Context(p, bk, None, generic_kernels=True) as gen:
            for t in (None, tv):
                got = fast.blind_rotate(c, t)
                assert np.array_equal(got, gen.blind_rotate(c, t))
                assert np.array_equal(got[0], O.blind_rotate(c[0], bk, op, O.NTT, tv=t))
            assert np.array_equal(fast.gate_bootstrapping_tlwe_to_tlwe(c), gen.gate_bootstrapping_tlwe_to_tlwe(c))


@pytest.mark.parametrize("N,K,L", [(512, 1, 2), (512, 1, 3), (512, 2, 3), (1024, 1, 2), (1024, 1, 3), (1024, 2, 2),
                                   (1024, 2, 3), (2048, 1, 2), (2048, 2, 2), (2048, 2, 3)])
def test_workgroup_ntt_sets_vs_generic_and_oracle(N, K, L):
    """The other parameter sets with workgroup NTT kernels (one wave per gate at N = 512, two at
    N = 1024; split key except for (512, 1, 2)) vs the O(N^2) wrap-around kernels vs the oracle,
    including key words at the ends of the range, and a NAND end to end on generated keys."""
    import tfhe_amd

    n = 4
    p = tfhe_amd.Params(n, N, K, 8, L, 5, 2)
    op = O.params_from(n, N, K, 8, L, 5, 2)
    rs = np.random.RandomState(77 + L + N + K)
    keys = [rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32),
            np.full(p.bk_shape(), 0x80000000, dtype=np.uint32),
            np.full(p.bk_shape(), 0x7FFF8000, dtype=np.uint32)]
    c = rs.randint(0, 2**32, size=(7, n + 1), dtype=np.uint64).astype(np.uint32)
    c[0, 1:] = 0x80000000 // N  # rotate by one position every step
    tv = rs.randint(0, 2**32, size=(K + 1, N), dtype=np.uint64).astype(np.uint32)
    for bk in keys:
        with tfhe_amd.Context(p, bk, None) as fast, tfhe_amd.Context(p, bk, None, generic_kernels=True) as gen:
            assert fast.kernel_family() == ("ntt512" if (N, L) == (512, 2) else "ntt-workgroup")
            assert gen.kernel_family() == "generic"
            for t in (None, tv):
                got = fast.blind_rotate(c, t)
                assert np.array_equal(got, gen.blind_rotate(c, t))
                assert np.array_equal(got[0], O.blind_rotate(c[0], bk, op, O.NTT, tv=t))
            assert np.array_equal(fast.gate_bootstrapping_tlwe_to_tlwe(c), gen.gate_bootstrapping_tlwe_to_tlwe(c))
    # end to end: n = 200 keeps the oracle's key generation short
    op = O.params_from(200, N, K, 8, L, 8, 2)
    s0, sr, bk, ks = O.keyset(op, seed=6 + L)
    p = tfhe_amd.Params(op.n, op.N, op.K, op.bg_bit, op.L, op.T, op.base_bit)
    r = O.rng(2)
    bits_a, bits_b = np.arange(8) % 2, (np.arange(8) // 2) % 2
    a, b = O.encrypt_bits(bits_a, s0, op, r), O.encrypt_bits(bits_b, s0, op, r)
    with tfhe_amd.Context(p, bk, ks) as ctx:
        got = ctx.nand(a, b)
    assert np.array_equal(got, O.Context(op, bk, ks).nand_batch(a, b))
    assert np.array_equal(O.decrypt_bits(got, s0, op), 1 - (bits_a & bits_b))


def test_n512_k1_both_forms_vs_generic():
    """N = 512, K = 1, L = 2 runs on the wave-per-gate kernels of the default set: throughput and
    latency forms against the O(N^2) kernels, accumulator and extracted sample."""
    import tfhe_amd

    n = 9
    p = tfhe_amd.Params(n, 512, 1, 8, 2, 5, 2)
    rs = np.random.RandomState(12)
    bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
    c = rs.randint(0, 2**32, size=(70, n + 1), dtype=np.uint64).astype(np.uint32)
    tv = rs.randint(0, 2**32, size=(2, 512), dtype=np.uint64).astype(np.uint32)
    with tfhe_amd.Context(p, bk, None, generic_kernels=True) as gen:
        want_acc, want_acc_tv, want_lv1 = gen.blind_rotate(c), gen.blind_rotate(c, tv), gen.gate_bootstrapping_tlwe_to_tlwe(c)
    for form in ("throughput", "latency"):
        with tfhe_amd.Context(p, bk, None, form=form) as ctx:
            assert ctx.kernel_family() == "ntt512"
            assert np.array_equal(ctx.blind_rotate(c), want_acc), form
            assert np.array_equal(ctx.blind_rotate(c, tv), want_acc_tv), form
            assert np.array_equal(ctx.gate_bootstrapping_tlwe_to_tlwe(c), want_lv1), form


def test_config5_n2048_nand_vs_oracle():
    """N = 2048 stress set end to end (n = 800, K = 1, L = 3, T = 8): NAND gates bit-exact vs the
    oracle and decrypting correctly."""
    import tfhe_amd

    op = O.params_n2048()
    s0, sr, bk, ks = O.keyset(op, seed=4)
    p = tfhe_amd.Params(op.n, op.N, op.K, op.bg_bit, op.L, op.T, op.base_bit)
    r = O.rng(1)
    bits_a, bits_b = np.arange(12) % 2, (np.arange(12) // 2) % 2
    a, b = O.encrypt_bits(bits_a, s0, op, r), O.encrypt_bits(bits_b, s0, op, r)
    want = O.Context(op, bk, ks).nand_batch(a, b)
    with tfhe_amd.Context(p, bk, ks) as ctx:
        got = ctx.nand(a, b)
    assert np.array_equal(got, want)
    assert np.array_equal(O.decrypt_bits(got, s0, op), 1 - (bits_a & bits_b))


@pytest.mark.parametrize("N,K,L", [(512, 2, 2), (512, 1, 2), (512, 2, 3), (1024, 1, 2), (1024, 2, 2), (2048, 1, 3),
                                   (2048, 2, 2)])
def test_stand_alone_external_product_and_cmux_on_ntt_kernels(N, K, L):
    """tfhe_amd_external_product_batch / _cmux_batch (reference src/trgsw.hpp:100-117) at the ring
    sizes with NTT kernels, against the oracle's exact product: random operands, operands at the
    ends of the centred range, and a batch that grows and shrinks (the per-device scratch is reused)."""
    import tfhe_amd

    p = tfhe_amd.Params(16, N, K, 8, L, 5, 2)
    op = O.params_from(16, N, K, 8, L, 5, 2)
    rs = np.random.RandomState(N + K + L)
    R = (K + 1) * L
    for batch in (3, 5, 1):
        trgsw = rs.randint(0, 2**32, size=(batch, R, K + 1, N), dtype=np.uint64).astype(np.uint32)
        d0 = rs.randint(0, 2**32, size=(batch, K + 1, N), dtype=np.uint64).astype(np.uint32)
        d1 = rs.randint(0, 2**32, size=(batch, K + 1, N), dtype=np.uint64).astype(np.uint32)
        # element 0: extreme magnitudes (key words +-2^31, digits of -128 everywhere)
        trgsw[0] = np.where(rs.randint(0, 2, size=trgsw[0].shape) == 1, 0x80000000, 0x7FFFFFFF).astype(np.uint32)
        d0[0] = 0x80808080 if L * 8 == 32 else (0x80800000 if L == 2 else 0x80808000)
        got_ep = tfhe_amd.external_product(p, trgsw, d0)
        got_cm = tfhe_amd.cmux(p, trgsw, d0, d1)
        for i in range(batch):
            assert np.array_equal(got_ep[i], O.external_product(trgsw[i], d0[i], op, O.NTT)), (batch, i)
            assert np.array_equal(got_cm[i], O.cmux(trgsw[i], d0[i], d1[i], op, O.NTT)), (batch, i)


@pytest.mark.parametrize("N,K", [(512, 1), (512, 2), (1024, 1), (1024, 2), (2048, 1), (2048, 2)])
def test_sixteen_bit_gadget_on_workgroup_ntt_kernels(N, K):
    """BG_BIT = 16 (L = 1), the reference's other gadget base below 32 (reference src/types.hpp:11-19,
    src/trgsw.hpp:21): blind rotation and stand-alone external product / cmux on the workgroup NTT
    kernels (every column with its own hi accumulator) against the O(N^2) kernels and the oracle,
    including key words and digits at the ends of their ranges."""
    import tfhe_amd

    n = 4
    p = tfhe_amd.Params(n, N, K, 16, 1, 5, 2)
    op = O.params_from(n, N, K, 16, 1, 5, 2)
    rs = np.random.RandomState(16 + N + K)
    keys = [rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32),
            np.full(p.bk_shape(), 0x80000000, dtype=np.uint32),
            np.full(p.bk_shape(), 0x7FFFFFFF, dtype=np.uint32)]
    c = rs.randint(0, 2**32, size=(5, n + 1), dtype=np.uint64).astype(np.uint32)
    tv = rs.randint(0, 2**32, size=(K + 1, N), dtype=np.uint64).astype(np.uint32)
    tv[0, :] = 0x80000000   # digits of -32768 after the first rotation's difference
    for bk in keys:
        with tfhe_amd.Context(p, bk, None) as fast, tfhe_amd.Context(p, bk, None, generic_kernels=True) as gen:
            assert fast.kernel_family() == "ntt-workgroup" and gen.kernel_family() == "generic"
            for t in (None, tv):
                got = fast.blind_rotate(c, t)
                assert np.array_equal(got, gen.blind_rotate(c, t))
                assert np.array_equal(got[0], O.blind_rotate(c[0], bk, op, O.NTT, tv=t))
    R = K + 1
    trgsw = rs.randint(0, 2**32, size=(3, R, K + 1, N), dtype=np.uint64).astype(np.uint32)
    d0 = rs.randint(0, 2**32, size=(3, K + 1, N), dtype=np.uint64).astype(np.uint32)
    d1 = rs.randint(0, 2**32, size=(3, K + 1, N), dtype=np.uint64).astype(np.uint32)
    trgsw[0] = 0x80000000
    d0[0] = 0x80000000
    got_ep, got_cm = tfhe_amd.external_product(p, trgsw, d0), tfhe_amd.cmux(p, trgsw, d0, d1)
    for i in range(3):
        assert np.array_equal(got_ep[i], O.external_product(trgsw[i], d0[i], op, O.NTT))
        assert np.array_equal(got_cm[i], O.cmux(trgsw[i], d0[i], d1[i], op, O.NTT))


def _random_parameter_sets(count, seed):
    """Parameter sets drawn from everything the reference's constraints allow (8- or 16-bit gadget with
    L * bg_bit <= 31, T * base_bit <= 31, base_bit <= 3 for the table size, power-of-two ring)."""
    rs = np.random.RandomState(seed)
    sets = []
    while len(sets) < count:
        N = 1 << rs.randint(1, 11)          # 2 .. 1024
        K = int(rs.randint(1, 5))           # 1 .. 4
        bg = int(rs.choice([8, 16]))
        L = 1 if bg == 16 else int(rs.randint(1, 4))
        bb = int(rs.randint(1, 4))
        T = int(rs.randint(1, min(31 // bb, 10) + 1))
        n = int(rs.randint(1, 13))
        if N * K > 2048 or (N * K * L * (K + 1)) > 16384:
            continue                         # keeps the O(N^2) oracle and kernels to a second or two per set
        sets.append((n, int(N), K, bg, L, T, bb))
    return sets


def _wide_parameter_sets(count, seed):
    """The corners the draw above leaves out: key-switch bases of 4 .. 6 bits (table size allowing) and
    rings of 2048 .. 8192 (small n, K = 1 .. 2: the O(N^2) oracle takes seconds there)."""
    rs = np.random.RandomState(seed)
    sets = []
    while len(sets) < count:
        big_ring = len(sets) % 2 == 1
        N = 1 << (rs.randint(11, 14) if big_ring else rs.randint(3, 10))
        K = int(rs.randint(1, 3))
        bg = int(rs.choice([8, 16]))
        L = 1 if bg == 16 else int(rs.randint(1, 4))
        bb = int(rs.randint(1, 4) if big_ring else rs.randint(4, 7))
        T = int(rs.randint(1, min(31 // bb, 6) + 1))
        n = int(rs.randint(1, 4) if big_ring else rs.randint(1, 13))
        if N * K * T * (1 << bb) * (n + 1) > (1 << 24) or (big_ring and N * K * L > 16384):
            continue
        sets.append((n, int(N), K, bg, L, T, bb))
    return sets


@pytest.mark.parametrize("shape", _random_parameter_sets(24, seed=2024) + _wide_parameter_sets(10, seed=2025))
def test_random_parameter_sets_vs_oracle(shape):
    """The whole gate (reference src/hom_nand.hpp:74-81) at parameter sets nobody picked by hand: the kernels
    chosen by the library for each set (NTT where one exists, the exact O(N^2) ones elsewhere) against the
    oracle's schoolbook multiplier, bit for bit, on random keys and ciphertexts."""
    import tfhe_amd

    n, N, K, bg, L, T, bb = shape
    p = tfhe_amd.Params(*shape)
    op = O.params_from(*shape)
    rs = np.random.RandomState(n * 7919 + N + 31 * K + bg + L + T + bb)
    bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
    ks = rs.randint(0, 2**32, size=p.ks_shape(), dtype=np.uint64).astype(np.uint32)
    a = rs.randint(0, 2**32, size=(5, n + 1), dtype=np.uint64).astype(np.uint32)
    b = rs.randint(0, 2**32, size=(5, n + 1), dtype=np.uint64).astype(np.uint32)
    a[0], b[0] = 0, 0xFFFFFFFF                       # extreme words
    with tfhe_amd.Context(p, bk, ks) as ctx:
        got = ctx.nand(a, b)
        c = np.stack([O.nand_combine(x, y, op) for x, y in zip(a, b)])
        acc = ctx.blind_rotate(c)
    for i in range(a.shape[0]):
        assert np.array_equal(acc[i], O.blind_rotate(c[i], bk, op, O.SCHOOLBOOK)), (shape, i)
        assert np.array_equal(got[i], O.nand(a[i], b[i], bk, ks, op, O.SCHOOLBOOK)), (shape, i)
