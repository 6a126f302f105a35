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
