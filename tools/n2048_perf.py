"""Config 5 throughput: N = 2048 stress set (n=800, K=1, L=3, T=8), random keys, device resident."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import tfhe_amd  # noqa: E402

p = tfhe_amd.Params(800, 2048, 1, 8, 3, 8, 2)
rs = np.random.RandomState(0)
bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
ks = rs.randint(0, 2**32, size=p.ks_shape(), dtype=np.uint64).astype(np.uint32)
ctx = tfhe_amd.Context(p, bk, ks)
ctx.enable_timing(True)
for G in [int(x) for x in sys.argv[1:]] or [1, 64, 256, 512, 1024, 2048]:
    a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    out = torch.empty_like(a)
    ctx.nand_device(a, b, out); ctx.synchronize(); ctx.timing()
    t0 = time.time()
    ctx.nand_device(a, b, out)
    ctx.synchronize()
    dt = time.time() - t0
    br, ksm, n = ctx.timing()
    print(f"G={G:5d} {dt*1e3:9.2f} ms  BR {br/n:9.2f}  KS {ksm/n:7.2f}  {G/dt:9.0f} gates/s", flush=True)
