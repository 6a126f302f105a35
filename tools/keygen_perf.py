"""Evaluation-key generation on the GPU: time to a ready context, default and N = 2048 sets."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import tfhe_amd  # noqa: E402

for name, p, a0, a1 in (("default", tfhe_amd.params_default(), 0.0000925119974676756, 2.9802322387695312e-08),
                        ("N=2048", tfhe_amd.Params(800, 2048, 1, 8, 3, 8, 2), 0.0000925119974676756, 2.9802322387695312e-08)):
    rs = np.random.RandomState(1)
    s0, sr = rs.randint(0, 2, p.n).astype(np.uint8), rs.randint(0, 2, (p.K, p.N)).astype(np.uint8)
    tfhe_amd.Context.from_secret(p, s0, sr, a0, a1, seed=1).close()
    t0 = time.time()
    ctx = tfhe_amd.Context.from_secret(p, s0, sr, a0, a1, seed=2)
    dt = time.time() - t0
    ctx.close()
    t0 = time.time()
    bk, ks = tfhe_amd.keygen(p, s0, sr, a0, a1, seed=2)
    dt2 = time.time() - t0
    print(f"{name}: context from secret key {dt*1e3:.1f} ms (generate + transform + pack on the GPU); "
          f"raw keys to host {dt2*1e3:.1f} ms ({(bk.nbytes + ks.nbytes)/1e6:.0f} MB)", flush=True)
