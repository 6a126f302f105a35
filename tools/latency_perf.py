"""Latency of small batches with each blind-rotation form."""
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
for form in ("latency", "throughput"):
    ctx = tfhe_amd.Context(p, bk, ks, form=form)
    ctx.enable_timing(True)
    for G in (1, 4, 16, 32, 64, 128, 256, 512, 1024, 2048):
        a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
        b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
        out = torch.empty_like(a)
        ctx.nand_device(a, b, out); ctx.synchronize(); ctx.timing()
        t0 = time.time()
        for _ in range(3):
            ctx.nand_device(a, b, out)
        ctx.synchronize()
        dt = (time.time() - t0) / 3
        br, ksm, n = ctx.timing()
        print(f"{form:10s} G={G:5d} {dt*1e3:8.2f} ms  BR {br/n:8.2f}  KS {ksm/n:6.2f}  {G/dt:9.0f} gates/s", flush=True)
    ctx.close()
