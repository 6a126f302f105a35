"""Blind-rotation forms side by side at the default parameters: identical outputs, time per NAND batch.

  python3 tools/forms_perf.py
"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import tfhe_amd
p = tfhe_amd.params_default()
rs = np.random.RandomState(0)
bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
ks = rs.randint(0, 2**32, size=p.ks_shape(), dtype=np.uint64).astype(np.uint32)
ctxs = {f: tfhe_amd.Context(p, bk, ks, form=f) for f in ("throughput", "latency")}
for G in (1, 3, 130, 513):
    a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    outs = {}
    for f, c in ctxs.items():
        o = torch.empty_like(a); c.nand_device(a, b, o); c.synchronize(); outs[f] = o
    print(G, "latency==throughput", bool((outs["latency"] == outs["throughput"]).all()), flush=True)
for G in (256, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096, 8192):
    a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    o = torch.empty_like(a)
    line = [str(G)]
    for f, c in ctxs.items():
        if f == "latency" and G > 2048: line.append("-"); continue
        c.nand_device(a, b, o); c.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): c.nand_device(a, b, o)
        c.synchronize()
        line.append("%s %.2f ms" % (f, (time.perf_counter() - t0) / 3 * 1e3))
    print("  ".join(line), flush=True)
