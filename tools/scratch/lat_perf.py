import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import tfhe_amd
p = tfhe_amd.params_default()
rs = np.random.RandomState(0)
bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
ks = rs.randint(0, 2**32, size=p.ks_shape(), dtype=np.uint64).astype(np.uint32)
ctx = tfhe_amd.Context(p, bk, ks, form="latency")
ctx.enable_timing(True)
for G in (1, 64, 256):
    a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    out = torch.empty_like(a)
    ctx.nand_device(a, b, out); ctx.synchronize(); ctx.timing()
    t0 = time.time()
    for _ in range(5):
        ctx.nand_device(a, b, out)
    ctx.synchronize()
    dt = (time.time() - t0) / 5
    br, ksm, n = ctx.timing()
    print(f"G={G:5d} {dt*1e3:8.3f} ms  BR {br/n:8.3f}  KS {ksm/n:6.3f}", flush=True)
