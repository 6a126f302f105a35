"""Throughput of the parameter sets beside the default one that have NTT kernels (n = 630, T = 8),
random keys, device resident; the exact O(N^2) kernels beside them."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import tfhe_amd  # noqa: E402

for N, K, L in ((512, 1, 2), (512, 2, 3), (1024, 1, 2), (1024, 1, 3), (1024, 2, 2)):
    p = tfhe_amd.Params(630, N, K, 8, L, 8, 2)
    rs = np.random.RandomState(0)
    bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
    ks = rs.randint(0, 2**32, size=p.ks_shape(), dtype=np.uint64).astype(np.uint32)
    for kw, sizes in (({}, (1, 8192)), ({"generic_kernels": True}, (256,))):
        with tfhe_amd.Context(p, bk, ks, **kw) as ctx:
            ctx.enable_timing(True)
            for G in sizes:
                a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
                b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
                out = torch.empty_like(a)
                ctx.nand_device(a, b, out); ctx.synchronize(); ctx.timing()
                t0 = time.time()
                ctx.nand_device(a, b, out)
                ctx.synchronize()
                dt = time.time() - t0
                br, ksm, n = ctx.timing()
                print(f"N={N} K={K} L={L} {ctx.kernel_family():14s} G={G:5d} {dt*1e3:9.2f} ms  BR {br/n:9.2f}  KS {ksm/n:7.2f}  {G/dt:9.0f} gates/s", flush=True)
