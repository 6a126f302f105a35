import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import tfhe_amd
p = tfhe_amd.params_default()
rs = np.random.RandomState(0)
bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
ks = rs.randint(0, 2**32, size=p.ks_shape(), dtype=np.uint64).astype(np.uint32)
for form in (None, "throughput"):
    ctx = tfhe_amd.Context(p, bk, ks, form=form)
    for G in (8192, 8193, 8192 + 256, 8192 + 512, 2048, 2048 + 100, 1, 100, 256):
        a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
        b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
        out = torch.empty_like(a)
        ctx.enable_timing(True)
        ctx.nand_device(a, b, out); ctx.synchronize(); ctx.timing()
        t0 = time.time()
        for _ in range(3):
            ctx.nand_device(a, b, out)
        ctx.synchronize()
        dt = (time.time() - t0) / 3
        br, ksm, n = ctx.timing()
        print(f"form={form} G={G:6d} {dt*1e3:8.2f} ms  {dt/G*1e6:7.3f} us/gate  BR {br/n:7.2f} KS {ksm/n:6.2f}", flush=True)
    ctx.close()
