"""Speed of the verification mode (products through a replay of the reference's FFT)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import tfhe_amd  # noqa: E402

p = tfhe_amd.params_default()
rs = np.random.RandomState(0)
bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
ks = rs.randint(0, 2**32, size=p.ks_shape(), dtype=np.uint64).astype(np.uint32)
ctx = tfhe_amd.Context(p, bk, ks, reference_fft=True)
for G in (1, 1024, 2048, 8192):
    a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    out = torch.empty_like(a)
    ctx.nand_device(a, b, out); ctx.synchronize()
    t0 = time.time()
    ctx.nand_device(a, b, out); ctx.synchronize()
    dt = time.time() - t0
    print(f"reference-FFT mode G={G:5d} {dt*1e3:9.2f} ms {G/dt:9.0f} gates/s", flush=True)
