#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- tests/golden/refnand512.npz: what the compiled reference's
HomNAND<>::nand (oracle/_ref/ref_nand, built from /root/reference/src with g++ -std=gnu++17 -O2)
returns at the DEFAULT parameters for 48 input pairs under an evaluation key that is too large to
commit (75 MB) and is therefore regenerated from a seed by the oracle's keygen.  The fixture holds
the seed, a checksum of the regenerated key, the input ciphertexts and the reference's outputs.
Runs only in the development container (needs the reference to build ref_nand).
"""
import pathlib
import subprocess
import sys
import tempfile

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import oracle_py as O  # noqa: E402

KEY_SEED, INPUT_SEED, BIT_SEED, G = 606, 66, 6, 8
G_MORE = 40  # a second group of pairs from the next seeds, appended (the first 8 stay what they were)


def key_checksum(bk, ks):
    w = np.concatenate([bk.reshape(-1), ks.reshape(-1)]).astype(np.uint64)
    idx = np.arange(w.size, dtype=np.uint64)
    return np.array([int(w.sum() % (1 << 61)), int(((w * (idx % np.uint64(65521) + np.uint64(1))).sum()) % (1 << 61))],
                    dtype=np.uint64)


def main() -> int:
    subprocess.run(["make", "-C", str(HERE), "ref"], check=True)
    p = O.params_default()
    s0, sr, bk, ks = O.keyset(p, seed=KEY_SEED)
    rs = np.random.RandomState(BIT_SEED)
    ba, bb = rs.randint(0, 2, G), rs.randint(0, 2, G)
    r = O.rng(INPUT_SEED)
    a, b = O.encrypt_bits(ba, s0, p, r), O.encrypt_bits(bb, s0, p, r)
    rs2, r2 = np.random.RandomState(BIT_SEED + 1), O.rng(INPUT_SEED + 1)
    ba2, bb2 = rs2.randint(0, 2, G_MORE), rs2.randint(0, 2, G_MORE)
    ba, bb = np.concatenate([ba, ba2]), np.concatenate([bb, bb2])
    a = np.concatenate([a, O.encrypt_bits(ba2, s0, p, r2)])
    b = np.concatenate([b, O.encrypt_bits(bb2, s0, p, r2)])
    with tempfile.TemporaryDirectory() as tmp:
        t = pathlib.Path(tmp)
        bk.tofile(t / "bk.bin"); ks.tofile(t / "ks.bin"); np.stack([a, b]).tofile(t / "in.bin")
        subprocess.run([str(HERE / "_ref" / "ref_nand"), str(t / "bk.bin"), str(t / "ks.bin"), str(t / "in.bin"),
                        str(t / "out.bin"), "8"], check=True)
        out = np.fromfile(t / "out.bin", dtype=np.uint32).reshape(G + G_MORE, p.n + 1)
    dst = HERE.parent / "tests" / "golden" / "refnand512.npz"
    np.savez_compressed(dst, key_seed=np.array([KEY_SEED]), key_checksum=key_checksum(bk, ks), s0=s0, bits_a=ba,
                        bits_b=bb, a=a, b=b, out=out)
    print(dst, dst.stat().st_size, "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
