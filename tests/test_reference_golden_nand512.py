"""Known-answer test at the DEFAULT parameters against the compiled reference, without needing the
reference at test time: tests/golden/refnand512.npz (made by oracle/make_golden_nand512.py) holds 48
input pairs and the ciphertexts HomNAND<>::nand of the real reference returned for them under an
evaluation key regenerated here from the recorded seed (its checksum is recorded too).  With the
reference's FFT product replayed, the oracle and the GPU must return exactly those ciphertexts;
with the exact product, the same decrypted bits.  Nothing on this path depends on the machine's
math library (the oracle's seeded key generator uses additions only, the FFT power tables start
from constants), so a key that does not regenerate, or an output that differs, is a failure."""
import sys

import numpy as np
import pytest

import oracle_py as O

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent / "oracle"))
from make_golden_nand512 import key_checksum  # noqa: E402


@pytest.fixture(scope="module")
def case(golden):
    g = golden("refnand512")
    p = O.params_default()
    s0, sr, bk, ks = O.keyset(p, seed=int(g["key_seed"][0]))
    assert np.array_equal(s0, g["s0"]), "the oracle's seeded key generator does not regenerate the recorded secret key"
    assert np.array_equal(key_checksum(bk, ks), g["key_checksum"]), \
        "the oracle's seeded key generator does not regenerate the recorded evaluation key"
    return p, g, bk, ks


def test_oracle_replay_equals_recorded_reference_output(case):
    p, g, bk, ks = case
    for i in (0, 5, 8, 47):
        assert np.array_equal(O.nand(g["a"][i], g["b"][i], bk, ks, p, 2), g["out"][i])
    exact = O.Context(p, bk, ks).nand_batch(g["a"], g["b"])
    want = 1 - (g["bits_a"] & g["bits_b"])
    assert np.array_equal(O.decrypt_bits(exact, g["s0"], p), want)
    assert np.array_equal(O.decrypt_bits(g["out"], g["s0"], p), want)


@pytest.mark.gpu
def test_gpu_replay_equals_recorded_reference_output(case):
    import tfhe_amd

    p, g, bk, ks = case
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, ks, reference_fft=True) as ctx:
        assert np.array_equal(ctx.nand(g["a"], g["b"]), g["out"])
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, ks) as ctx:
        exact = ctx.nand(g["a"], g["b"])
    assert np.array_equal(exact, O.Context(p, bk, ks).nand_batch(g["a"], g["b"]))
    assert np.array_equal(O.decrypt_bits(exact, g["s0"], p), 1 - (g["bits_a"] & g["bits_b"]))


@pytest.mark.gpu
def test_wave_and_workgroup_replay_kernels_agree(case):
    """The wave-per-gate replay kernel of the default set against the parameter-generic
    one-workgroup-per-gate replay kernel, on encrypted and on arbitrary inputs, both blind-rotation
    outputs (accumulator and extracted sample)."""
    import tfhe_amd

    p, g, bk, ks = case
    rs = np.random.RandomState(8)
    c = np.concatenate([np.stack([O.nand_combine(x, y, p) for x, y in zip(g["a"], g["b"])]),
                        rs.randint(0, 2**32, size=(9, p.n + 1), dtype=np.uint64).astype(np.uint32)])
    tv = rs.randint(0, 2**32, size=(p.K + 1, p.N), dtype=np.uint64).astype(np.uint32)
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, None, reference_fft=True) as fast, \
            tfhe_amd.Context(tfhe_amd.params_default(), bk, None, reference_fft=True, generic_kernels=True) as slow:
        assert np.array_equal(fast.blind_rotate(c), slow.blind_rotate(c))
        assert np.array_equal(fast.blind_rotate(c, tv), slow.blind_rotate(c, tv))
        assert np.array_equal(fast.gate_bootstrapping_tlwe_to_tlwe(c), slow.gate_bootstrapping_tlwe_to_tlwe(c))
