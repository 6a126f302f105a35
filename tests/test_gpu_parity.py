"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle.
Bit-exact (integer work only)."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def keys():
    p = O.params_default()
    s0, sr, bk, ks = O.keyset(p, seed=1234)
    return p, s0, sr, bk, ks


@pytest.fixture(scope="module")
def ctx(keys):
    import tfhe_amd

    p, _, _, bk, ks = keys
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, ks, device=0) as c:
        yield c


@pytest.fixture(scope="module")
def octx(keys):
    p, _, _, bk, ks = keys
    return O.Context(p, bk, ks)


def _inputs(p, s0, n, seed):
    rs = np.random.RandomState(seed)
    bits_a, bits_b = rs.randint(0, 2, n), rs.randint(0, 2, n)
    r = O.rng(seed)
    return bits_a, bits_b, O.encrypt_bits(bits_a, s0, p, r), O.encrypt_bits(bits_b, s0, p, r)


def test_nand_small_batch_bit_exact(keys, ctx, octx):
    p, s0, _, _, _ = keys
    bits_a, bits_b, a, b = _inputs(p, s0, 8, seed=7)
    got = ctx.nand(a, b)
    want = octx.nand_batch(a, b)
    assert np.array_equal(got, want)
    assert np.array_equal(O.decrypt_bits(got, s0, p), 1 - (bits_a & bits_b))


def test_stages_bit_exact(keys, ctx, octx):
    """gate_bootstrapping_tlwe_to_tlwe and identity_key_switch separately, device pointers."""
    import torch

    p, s0, _, bk, ks = keys
    _, _, a, b = _inputs(p, s0, 5, seed=11)
    c = np.stack([O.nand_combine(x, y, p) for x, y in zip(a, b)])
    want_lv1 = np.stack([O.sample_extract_index(octx.blind_rotate(ci), 0, p) for ci in c])
    want_out = np.stack([O.identity_key_switch(l, ks, p) for l in want_lv1])
    dc = torch.from_numpy(c.view(np.int32)).cuda()
    dlv1 = torch.empty((5, p.lv1 + 1), dtype=torch.int32, device="cuda")
    dout = torch.empty((5, p.n + 1), dtype=torch.int32, device="cuda")
    ctx.gate_bootstrapping_tlwe_to_tlwe_device(dc, dlv1)
    ctx.identity_key_switch_device(dlv1, dout)
    ctx.synchronize()
    assert np.array_equal(dlv1.cpu().numpy().view(np.uint32), want_lv1)
    assert np.array_equal(dout.cpu().numpy().view(np.uint32), want_out)


def test_key_switch_random_lv1(keys, ctx):
    """identity_key_switch on uniformly random lv1 words (every digit pattern, ragged tile)."""
    import torch

    p, _, _, _, ks = keys
    rs = np.random.RandomState(5)
    lv1 = rs.randint(0, 2**32, size=(37, p.lv1 + 1), dtype=np.uint64).astype(np.uint32)
    lv1[0, 1:] = 0
    lv1[1, 1:] = 0xFFFFFFFF
    lv1[2, 1:] = 0x001FFFFF  # just below the rounding boundary
    lv1[3, 1:] = 0x00200000
    want = np.stack([O.identity_key_switch(l, ks, p) for l in lv1])
    dlv1 = torch.from_numpy(lv1.view(np.int32)).cuda()
    dout = torch.empty((37, p.n + 1), dtype=torch.int32, device="cuda")
    ctx.identity_key_switch_device(dlv1, dout)
    ctx.synchronize()
    assert np.array_equal(dout.cpu().numpy().view(np.uint32), want)


def test_blind_rotate_adversarial_inputs(keys, ctx, octx):
    """Ciphertext words chosen to hit rotation-amount edge cases (0, 2N wrap, rounding ties)."""
    p, _, _, _, _ = keys
    rs = np.random.RandomState(3)
    c = rs.randint(0, 2**32, size=(6, p.n + 1), dtype=np.uint64).astype(np.uint32)
    c[0, :] = 0
    c[1, :] = 0xFFFFFFFF
    c[2, 1:] = 0x001FFFFF
    c[3, 1:] = 0x00200000
    c[4, 1:] = 0xFFE00000
    got = ctx.bootstrap(c)
    want = octx.bootstrap_batch(c)
    assert np.array_equal(got, want)


def test_config2_1024_gates_bit_exact(keys, ctx, octx):
    """1,024 independent NAND gates, default parameters, one GPU, every output word compared."""
    p, s0, _, _, _ = keys
    bits_a, bits_b, a, b = _inputs(p, s0, 1024, seed=1)
    got = ctx.nand(a, b)
    want = octx.nand_batch(a, b)
    assert np.array_equal(got, want)
    assert np.array_equal(O.decrypt_bits(got, s0, p), 1 - (bits_a & bits_b))


def test_empty_batch(ctx, keys):
    p = keys[0]
    out = ctx.nand(np.zeros((0, p.n + 1), np.uint32), np.zeros((0, p.n + 1), np.uint32))
    assert out.shape == (0, p.n + 1)


def test_latency_and_throughput_forms_agree(keys, octx):
    """One gate per workgroup (latency form) and one gate per wave (throughput form) are the
    same arithmetic: identical ciphertexts, both equal to the oracle."""
    import tfhe_amd

    p, s0, _, bk, ks = keys
    _, _, a, b = _inputs(p, s0, 7, seed=23)
    want = octx.nand_batch(a, b)
    for form in ("latency", "throughput"):
        with tfhe_amd.Context(tfhe_amd.params_default(), bk, ks, form=form) as ctx:
            assert np.array_equal(ctx.nand(a, b), want), form
            c = np.stack([O.nand_combine(x, y, p) for x, y in zip(a, b)])
            assert np.array_equal(ctx.blind_rotate(c[:2])[0], octx.blind_rotate(c[0])), form


def test_config4_ripple_carry_adder(keys, ctx, octx):
    """32-bit ripple-carry adder from NAND only, gates batched per dependency level, wires
    resident on the GPU.  Decrypted sum = a + b; on an 8-bit instance every wire equals the
    oracle evaluated in the same level order."""
    import torch

    from tfhe_amd import circuit as C

    p, s0, _, _, _ = keys
    rs = np.random.RandomState(3)
    for bits, check_wires in ((8, True), (32, False)):
        gates, n_wires, a, b, zero, sums, cout = C.ripple_carry_adder(bits)
        x, y = int(rs.randint(0, 2**bits, dtype=np.uint64)), int(rs.randint(0, 2**bits, dtype=np.uint64))
        r = O.rng(100 + bits)
        w = np.zeros((n_wires, p.n + 1), dtype=np.uint32)
        w[a] = O.encrypt_bits([(x >> i) & 1 for i in range(bits)], s0, p, r)
        w[b] = O.encrypt_bits([(y >> i) & 1 for i in range(bits)], s0, p, r)
        w[zero] = C.trivial_bit(0, p.n + 1)
        dw = torch.from_numpy(w.view(np.int32).copy()).cuda()

        def gpu_nand(u, v):
            # on torch's current stream, so the gathers before and the scatter after are ordered
            out = torch.empty_like(u)
            ctx.nand_device(u, v, out, stream=torch.cuda.current_stream())
            return out

        levels = C.levelize(gates, n_wires)
        assert C.evaluate(gates, dw, gpu_nand, levels) == 3 + 2 * bits + 1
        torch.cuda.synchronize()
        got = dw.cpu().numpy().view(np.uint32)
        out_bits = O.decrypt_bits(got[sums + [cout]], s0, p)
        assert sum(int(v) << i for i, v in enumerate(out_bits)) == x + y
        if check_wires:
            C.evaluate(gates, w, lambda u, v: octx.nand_batch(u, v), levels)
            assert np.array_equal(got, w)
