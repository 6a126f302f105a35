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


@pytest.mark.parametrize("rows", [1, 4, 16, 17, 37, 130])
def test_key_switch_random_lv1(keys, ctx, rows):
    """identity_key_switch on uniformly random lv1 words (every digit pattern, ragged tile), at
    batch sizes on both sides of the switch between the per-gate kernel and the tiled one."""
    import torch

    p, _, _, _, ks = keys
    rs = np.random.RandomState(5 + rows)
    lv1 = rs.randint(0, 2**32, size=(rows, p.lv1 + 1), dtype=np.uint64).astype(np.uint32)
    for i, word in enumerate((0, 0xFFFFFFFF, 0x001FFFFF, 0x00200000)):  # 0x001FFFFF: just below the rounding boundary
        if i < rows:
            lv1[i, 1:] = word
    want = np.stack([O.identity_key_switch(l, ks, p) for l in lv1])
    dlv1 = torch.from_numpy(lv1.view(np.int32)).cuda()
    dout = torch.empty((rows, p.n + 1), dtype=torch.int32, device="cuda")
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


@pytest.mark.parametrize("gates", [2048 + 5, 2048 + 512, 2048 + 513])
def test_ragged_batch_tail_on_the_workgroup_form(keys, ctx, octx, gates):
    """A batch just above a whole generation of resident waves: a remainder of up to 512 gates runs in
    the workgroup-per-gate form behind the wave-per-gate kernel (513: it does not).
    Every word equals the pinned wave-per-gate form's, the first and last gates (both sides of the split)
    equal the oracle's, and the same buffers are reused twice without a device-wide synchronisation."""
    import tfhe_amd
    import torch

    p, s0, _, bk, ks = keys
    bits_a, bits_b, a, b = _inputs(p, s0, gates, seed=gates)
    da = torch.from_numpy(a.view(np.int32)).cuda()
    db = torch.from_numpy(b.view(np.int32)).cuda()
    out1, out2 = torch.empty_like(da), torch.empty_like(da)
    ctx.nand_device(da, db, out1)
    ctx.nand_device(out1, db, out2)          # consumes the first call's output on the same stream
    ctx.synchronize()
    got1, got2 = out1.cpu().numpy().view(np.uint32), out2.cpu().numpy().view(np.uint32)
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, ks, form="throughput") as pinned:
        ref1, ref2 = torch.empty_like(da), torch.empty_like(da)
        pinned.nand_device(da, db, ref1)
        pinned.nand_device(ref1, db, ref2)
        pinned.synchronize()
    assert np.array_equal(got1, ref1.cpu().numpy().view(np.uint32))
    assert np.array_equal(got2, ref2.cpu().numpy().view(np.uint32))
    sample = [0, 1, 2047, 2048, gates - 2, gates - 1]
    assert np.array_equal(got1[sample], octx.nand_batch(a[sample], b[sample]))
    assert np.array_equal(O.decrypt_bits(got1, s0, p), 1 - (bits_a & bits_b))


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


# ---- gates other than NAND (the reference has none: parity is against the oracle, semantics by
# ---- decryption) -----------------------------------------------------------------------------

@pytest.mark.parametrize("op", sorted(O.GATES))
def test_gate_bit_exact_and_truth_table(keys, ctx, octx, op):
    p, s0, _, _, _ = keys
    abits = np.array([0, 0, 1, 1, 1, 0])
    bbits = np.array([0, 1, 0, 1, 0, 0])
    r = O.rng(100 + O.GATES[op])
    a, b = O.encrypt_bits(abits, s0, p, r), O.encrypt_bits(bbits, s0, p, r)
    got = ctx.gate(op, a, b)
    assert np.array_equal(got, octx.gate_batch(op, a, b))
    assert np.array_equal(O.decrypt_bits(got, s0, p), O.GATE_TRUTH[op](abits, bbits))


def test_per_gate_codes_not_and_mux_device(keys, ctx, octx):
    import torch

    p, s0, _, _, _ = keys
    n = 24
    rs = np.random.RandomState(5)
    r = O.rng(55)
    bits = rs.randint(0, 2, (3, n))
    a, b, c = (O.encrypt_bits(bits[i], s0, p, r) for i in range(3))
    ops = rs.randint(0, 10, n).astype(np.uint8)
    da, db, dc = (torch.from_numpy(x.view(np.int32)).cuda() for x in (a, b, c))
    out = torch.empty_like(da)
    ctx.gate_device(torch.from_numpy(ops).cuda(), da, db, out)
    ctx.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), octx.gate_batch(ops, a, b))
    ctx.mux_device(da, db, dc, out)
    ctx.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    assert np.array_equal(got, octx.mux_batch(a, b, c))
    assert np.array_equal(O.decrypt_bits(got, s0, p), np.where(bits[0] == 1, bits[1], bits[2]))
    assert np.array_equal(ctx.mux(a, b, c), got)
    ctx.not_device(da, out)
    ctx.synchronize()
    assert np.array_equal(O.decrypt_bits(out.cpu().numpy().view(np.uint32), s0, p), 1 - bits[0])
    assert np.array_equal(ctx.not_(a), out.cpu().numpy().view(np.uint32))


def test_unknown_gate_code_is_an_error(keys, ctx):
    import tfhe_amd

    p = keys[0]
    z = np.zeros((1, p.n + 1), dtype=np.uint32)
    with pytest.raises(tfhe_amd.TfheAmdError):
        ctx.gate(10, z, z)
    with pytest.raises(ValueError):
        ctx.gate("frob", z, z)


def test_mux_adder_32_bits_on_device(keys, ctx, octx):
    """32-bit adder with XOR and MUX gates, every level one batched device call; ciphertexts equal
    to the oracle run in the same level order, sum correct after decryption."""
    import torch
    from tfhe_amd import circuit as C

    p, s0, _, _, _ = keys
    bits = 32
    gates, n_wires, a, b, zero, sums, cout = C.mux_adder(bits)
    levels = C.levelize(gates, n_wires)
    rs = np.random.RandomState(3)
    x, y = int(rs.randint(0, 2**32, dtype=np.uint64)), int(rs.randint(0, 2**32, dtype=np.uint64))
    r = O.rng(33)
    w = np.zeros((n_wires, p.n + 1), dtype=np.uint32)
    w[a] = O.encrypt_bits([(x >> i) & 1 for i in range(bits)], s0, p, r)
    w[b] = O.encrypt_bits([(y >> i) & 1 for i in range(bits)], s0, p, r)
    w[zero] = C.trivial_bit(0, p.n + 1)
    dw = torch.from_numpy(w.view(np.int32).copy()).cuda()
    C.evaluate(gates, dw, C.DeviceBackend(ctx), levels)
    torch.cuda.synchronize()
    got = dw.cpu().numpy().view(np.uint32)

    class OracleBackend:
        def gates(self, ops, u, v):
            return octx.gate_batch(np.asarray(ops, dtype=np.uint8), u, v)

        def mux(self, s, x1, x0):
            return octx.mux_batch(s, x1, x0)

    C.evaluate(gates, w, OracleBackend(), levels)
    assert np.array_equal(got, w)
    total = sum(int(v) << i for i, v in enumerate(O.decrypt_bits(got[sums], s0, p)))
    total += int(O.decrypt_bits(got[[cout]], s0, p)[0]) << bits
    assert total == x + y


def test_prefix_adder_32_bits_on_device(keys, ctx, octx):
    """32-bit parallel-prefix adder (AND / XOR / MUX gates, 7 levels of up to 95 gates), every level one
    batched device call; every wire equals the oracle run in the same level order, sum correct."""
    import torch
    from tfhe_amd import circuit as C

    p, s0, _, _, _ = keys
    bits = 32
    gates, n_wires, a, b, zero, sums, cout = C.prefix_adder(bits)
    levels = C.levelize(gates, n_wires)
    assert len(levels) == 7
    rs = np.random.RandomState(12)
    x, y = int(rs.randint(0, 2**32, dtype=np.uint64)), int(rs.randint(0, 2**32, dtype=np.uint64))
    r = O.rng(34)
    w = np.zeros((n_wires, p.n + 1), dtype=np.uint32)
    w[a] = O.encrypt_bits([(x >> i) & 1 for i in range(bits)], s0, p, r)
    w[b] = O.encrypt_bits([(y >> i) & 1 for i in range(bits)], s0, p, r)
    dw = torch.from_numpy(w.view(np.int32).copy()).cuda()
    C.evaluate(gates, dw, C.DeviceBackend(ctx), levels)
    torch.cuda.synchronize()
    got = dw.cpu().numpy().view(np.uint32)

    class OracleBackend:
        def gates(self, ops, u, v):
            return octx.gate_batch(np.asarray(ops, dtype=np.uint8), u, v)

        def mux(self, s, x1, x0):
            return octx.mux_batch(s, x1, x0)

    C.evaluate(gates, w, OracleBackend(), levels)
    assert np.array_equal(got, w)
    total = sum(int(v) << i for i, v in enumerate(O.decrypt_bits(got[sums], s0, p)))
    total += int(O.decrypt_bits(got[[cout]], s0, p)[0]) << bits
    assert total == x + y


# ---- the 65,536-gate configuration at full size on one GPU ---------------------------------------

def test_config3_65536_gates_decrypt_and_sampled_parity(keys, ctx, octx):
    """All 65,536 outputs decrypt to NAND of the inputs; every output word equals what an independent
    kernel returns for the same gates (the workgroup-per-gate form, run in 128 chunks of 512 gates and
    compared on the GPU); 512 gates, one per stride of 128 at a moving offset plus the first and last
    eight, are bit-exact with the oracle."""
    import tfhe_amd
    import torch

    p, s0, _, bk, ks = keys
    G = 65536
    rs = np.random.RandomState(2)
    ba, bb = rs.randint(0, 2, G), rs.randint(0, 2, G)
    r = O.rng(2)
    a, b = O.encrypt_bits(ba, s0, p, r), O.encrypt_bits(bb, s0, p, r)
    da, db = torch.from_numpy(a.view(np.int32)).cuda(), torch.from_numpy(b.view(np.int32)).cuda()
    out = torch.empty_like(da)
    ctx.nand_device(da, db, out)
    ctx.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    s = s0.astype(np.uint32)
    phase = got[:, 0] - (got[:, 1:] * s[None, :]).sum(axis=1, dtype=np.uint32)
    assert np.array_equal((phase >> 31) == 0, (1 - (ba & bb)).astype(bool))
    # the whole batch against the other blind-rotation form (different kernel, different work split)
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, ks, form="latency") as lat:
        chunk = torch.empty((512, p.n + 1), dtype=torch.int32, device="cuda")
        for lo in range(0, G, 512):
            lat.nand_device(da[lo:lo + 512], db[lo:lo + 512], chunk)
            assert torch.equal(chunk, out[lo:lo + 512]), f"gates {lo}..{lo + 511} differ between the two kernel forms"
    pick = np.unique(np.concatenate([np.arange(0, 8), np.arange(G - 8, G),
                                     np.arange(0, G, 128) + (np.arange(G // 128) * 37) % 128]))
    assert pick.size >= 512
    assert np.array_equal(got[pick], octx.nand_batch(a[pick], b[pick]))


def test_full_size_closed_form_on_noiseless_inputs(keys, ctx):
    """Size-independent property at 65,536 gates: for a ciphertext with an all-zero mask every
    rotation amount is zero, every decomposed difference is zero, so the blind rotation returns the
    test vector rotated by -floor(b * 2N) exactly, whatever the key: acc_0[j] = +-1/8 with the sign
    of the wrap-around, acc_1 = acc_2 = 0."""
    import torch

    p = keys[0]
    G = 65536
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    c = torch.zeros((G, p.n + 1), dtype=torch.int32, device="cuda")
    c[:, 0] = torch.randint(-2**31, 2**31 - 1, (G,), dtype=torch.int32, device="cuda", generator=gen)
    lv1 = torch.empty((G, p.lv1 + 1), dtype=torch.int32, device="cuda")
    ctx.gate_bootstrapping_tlwe_to_tlwe_device(c, lv1)
    ctx.synchronize()
    bbar = (c[:, 0].to(torch.int64) & 0xFFFFFFFF) >> 22                       # floor(b * 2N), in [0, 2N)
    mu = 1 << 29
    # acc_0[j] = sign(j + bbar) / 8; the extracted sample is [acc_0[0], acc_1[0], -acc_1[N-1..1], acc_2 ..]
    want0 = torch.where((bbar & p.N) != 0, -mu, mu).to(torch.int32)
    assert torch.equal(lv1[:, 0], want0)
    assert not lv1[:, 1:].any()
