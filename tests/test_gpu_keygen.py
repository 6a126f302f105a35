"""Evaluation-key generation on the GPU (reference HomNAND::make_evaluate_key,
src/hom_nand.hpp:56-64).  The reference draws masks and noise from std::random_device
(src/random.hpp:10) and cannot be reproduced, so there are no reference bytes to compare with.
Pinned instead: the GPU keys against the CPU oracle's restatement of the same generator (ChaCha20
under a 256-bit seed): every mask word identical, every noise word identical or one unit apart (the
Gaussian goes through log / sin / cos of two different math libraries), at the default and the
N = 2048 parameter sets.  Also checked: the keys decrypt, under the secret key and with the
oracle's phase functions, to exactly the messages the reference's make_bk / make_ks encrypt, with
noise of the prescribed deviation; the result is a pure function of (secret, seed); and gates
evaluated with them are correct and bit-exact between the GPU and the oracle on the same key bytes."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu


def _signed(x):
    return x.astype(np.uint32).view(np.int32).astype(np.int64)


@pytest.fixture(scope="module")
def generated():
    import tfhe_amd

    p = O.params_default()
    s0, sr = O.keygen_secret(p, O.rng(77))
    bk, ks = tfhe_amd.keygen(tfhe_amd.params_default(), s0, sr, p.alpha_lv0, p.alpha_lv1, seed=2024)
    return p, s0, sr, bk, ks


def _compare_with_oracle(p, tp, s0, sr, seed):
    import tfhe_amd

    bk, ks = tfhe_amd.keygen(tp, s0, sr, p.alpha_lv0, p.alpha_lv1, seed=seed)
    obk, oks = O.keygen_chacha(s0, sr, p, p.alpha_lv0, p.alpha_lv1, tfhe_amd.seed_words(seed))
    # masks: pure generator words
    assert np.array_equal(bk[:, :, 1:, :], obk[:, :, 1:, :])
    assert np.array_equal(ks[..., 1:], oks[..., 1:])
    # b words: same masks and messages, so the difference is the difference of the noise words
    d_bk = (bk[:, :, 0, :] - obk[:, :, 0, :]).view(np.int32)
    d_ks = (ks[..., 0] - oks[..., 0]).view(np.int32)
    assert np.abs(d_bk).max() <= 1 and np.abs(d_ks).max() <= 1
    # ... and all but a vanishing share of them are identical
    assert np.count_nonzero(d_bk) <= 1e-6 * d_bk.size + 2 and np.count_nonzero(d_ks) <= 1e-6 * d_ks.size + 2
    return bk, ks


def test_keys_equal_the_oracle_generator_default_set(generated):
    import tfhe_amd

    p, s0, sr, bk, ks = generated
    bk2, ks2 = _compare_with_oracle(p, tfhe_amd.params_default(), s0, sr, seed=2024)
    assert np.array_equal(bk, bk2) and np.array_equal(ks, ks2)
    # a full-width seed: all eight words matter
    _compare_with_oracle(p, tfhe_amd.params_default(), s0, sr, seed=(1 << 255) | (0xABCDEF << 100) | 7)


def test_keys_equal_the_oracle_generator_n2048_set():
    import tfhe_amd

    p = O.params_n2048()
    tp = tfhe_amd.Params(p.n, p.N, p.K, p.bg_bit, p.L, p.T, p.base_bit)
    s0, sr = O.keygen_secret(p, O.rng(8))
    _compare_with_oracle(p, tp, s0, sr, seed=31337)


def test_deterministic_in_the_seed(generated):
    import tfhe_amd

    p, s0, sr, bk, ks = generated
    bk2, ks2 = tfhe_amd.keygen(tfhe_amd.params_default(), s0, sr, p.alpha_lv0, p.alpha_lv1, seed=2024)
    assert np.array_equal(bk, bk2) and np.array_equal(ks, ks2)
    bk3, _ = tfhe_amd.keygen(tfhe_amd.params_default(), s0, sr, p.alpha_lv0, p.alpha_lv1, seed=2025, want_ks=False)
    assert _ is None and not np.array_equal(bk, bk3)
    # seed=None draws the 256 bits from the operating system: two calls never agree
    bk4, _ = tfhe_amd.keygen(tfhe_amd.params_default(), s0, sr, p.alpha_lv0, p.alpha_lv1, seed=None, want_ks=False)
    bk5, _ = tfhe_amd.keygen(tfhe_amd.params_default(), s0, sr, p.alpha_lv0, p.alpha_lv1, seed=None, want_ks=False)
    assert not np.array_equal(bk4[0, 0, 1], bk5[0, 0, 1])


def test_bootstrapping_key_rows_encrypt_the_gadget(generated):
    """Row comp*L + level of bk[i] is TRLWE(0) plus s0_i * 2^(32 - 8 (level + 1)) on coefficient 0
    of polynomial comp (reference src/trgsw.hpp:61-82)."""
    p, s0, sr, bk, ks = generated
    sigma = p.alpha_lv1 * 2.0**32
    noise = []
    for i in list(range(0, p.n, 53)) + [p.n - 1]:
        for r in range((p.K + 1) * p.L):
            comp, level = divmod(r, p.L)
            row = bk[i, r].copy()
            row[comp, 0] -= np.uint32(int(s0[i]) << (32 - p.bg_bit * (level + 1)))   # remove the gadget
            e = _signed(O.trlwe_phase(row, sr, p))
            assert np.abs(e).max() < 7 * sigma + 2
            noise.append(e)
    noise = np.concatenate(noise)
    assert 0.9 * sigma < noise.std() < 1.1 * sigma
    assert abs(noise.mean() + 0.5) < 0.05 * sigma     # the reference's double -> torus rule truncates
    masks = bk[:, :, 1:, :].reshape(-1)[::97]
    assert abs((masks >> 31).mean() - 0.5) < 0.01 and abs((masks & 1).mean() - 0.5) < 0.01
    assert len(np.unique(masks)) > 0.999 * masks.size


def test_key_switching_rows_encrypt_the_digits(generated):
    """ks[i][m][o] is a TLWE_lv0 encryption of o * s1_i * 2^(32 - 2 (m + 1)); o = 0 rows are zero
    (reference src/tlwe_key_switching.hpp:26-45)."""
    p, s0, sr, bk, ks = generated
    s1 = sr.reshape(-1)
    sigma = p.alpha_lv0 * 2.0**32
    assert not ks[:, :, 0, :].any()
    noise = []
    for i in range(0, p.lv1, 37):
        for m in range(p.T):
            for o in range(1, 1 << p.base_bit):
                msg = (int(s1[i]) * o << (32 - p.base_bit * (m + 1))) & 0xFFFFFFFF
                ph = int(O.tlwe_phase(ks[i, m, o], s0))
                noise.append(((ph - msg + 2**31) % 2**32) - 2**31)
    noise = np.array(noise, dtype=np.float64)
    assert np.abs(noise).max() < 7 * sigma
    assert 0.85 * sigma < noise.std() < 1.15 * sigma


def test_gates_with_generated_keys(generated):
    import tfhe_amd

    p, s0, sr, bk, ks = generated
    rs = np.random.RandomState(4)
    r = O.rng(44)
    ba, bb = rs.randint(0, 2, 12), rs.randint(0, 2, 12)
    a, b = O.encrypt_bits(ba, s0, p, r), O.encrypt_bits(bb, s0, p, r)
    octx = O.Context(p, bk, ks)
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, ks) as ctx:
        got = ctx.nand(a, b)
    assert np.array_equal(got, octx.nand_batch(a, b))
    assert np.array_equal(O.decrypt_bits(got, s0, p), 1 - (ba & bb))
    # the same keys generated straight into a context, never copied to the host
    with tfhe_amd.Context.from_secret(tfhe_amd.params_default(), s0, sr, p.alpha_lv0, p.alpha_lv1, seed=2024) as ctx2:
        assert np.array_equal(ctx2.nand(a, b), got)
        x = ctx2.gate("xor", a, b)
        assert np.array_equal(O.decrypt_bits(x, s0, p), ba ^ bb)


def test_n2048_keys_evaluate_a_gate():
    import tfhe_amd

    p = O.params_n2048()
    tp = tfhe_amd.Params(p.n, p.N, p.K, p.bg_bit, p.L, p.T, p.base_bit)
    s0, sr = O.keygen_secret(p, O.rng(5))
    r = O.rng(6)
    ba, bb = np.array([0, 0, 1, 1]), np.array([0, 1, 0, 1])
    a, b = O.encrypt_bits(ba, s0, p, r), O.encrypt_bits(bb, s0, p, r)
    with tfhe_amd.Context.from_secret(tp, s0, sr, p.alpha_lv0, p.alpha_lv1, seed=9) as ctx:
        assert np.array_equal(O.decrypt_bits(ctx.nand(a, b), s0, p), 1 - (ba & bb))


def test_argument_checks():
    import tfhe_amd

    tp = tfhe_amd.params_default()
    with pytest.raises(ValueError):
        tfhe_amd.keygen(tp, np.zeros(3, dtype=np.uint8), np.zeros((2, 512), dtype=np.uint8), 1e-6, 1e-8, 1)
    with pytest.raises(ValueError):
        tfhe_amd.keygen(tp, np.full(636, 2, dtype=np.uint8), np.zeros((2, 512), dtype=np.uint8), 1e-6, 1e-8, 1)
    with pytest.raises(tfhe_amd.TfheAmdError):
        tfhe_amd.keygen(tp, np.zeros(636, dtype=np.uint8), np.zeros((2, 512), dtype=np.uint8), 0.7, 1e-8, 1)
