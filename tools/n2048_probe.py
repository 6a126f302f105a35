"""Config 5 probe: the N = 2048 stress set on the generic exact kernels vs the oracle."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "oracle")
import oracle_py as O  # noqa: E402
import tfhe_amd  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
op = O.params_n2048()
t0 = time.time()
s0, sr, bk, ks = O.keyset(op, seed=4)
print("keygen", time.time() - t0, flush=True)
p = tfhe_amd.Params(op.n, op.N, op.K, op.bg_bit, op.L, op.T, op.base_bit)
r = O.rng(1)
bits_a, bits_b = np.arange(G) % 2, (np.arange(G) // 2) % 2
a, b = O.encrypt_bits(bits_a, s0, op, r), O.encrypt_bits(bits_b, s0, op, r)
octx = O.Context(op, bk, ks)
t0 = time.time()
want = octx.nand_batch(a, b)
print("oracle", time.time() - t0, flush=True)
with tfhe_amd.Context(p, bk, ks) as ctx:
    ctx.nand(a[:1], b[:1])
    t0 = time.time()
    got = ctx.nand(a, b)
    dt = time.time() - t0
print(f"gpu generic: {G} gates in {dt:.3f} s; equal to oracle: {np.array_equal(got, want)}; "
      f"decrypt ok: {np.array_equal(O.decrypt_bits(got, s0, op), 1 - (bits_a & bits_b))}")
