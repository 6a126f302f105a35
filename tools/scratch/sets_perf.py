"""NTT-kernel throughput of the non-default parameter sets (random keys), one line per set."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import tfhe_amd
sets = [(800, 2048, 1, 8, 3, 2048), (630, 2048, 1, 8, 2, 2048), (630, 2048, 2, 8, 2, 2048), (630, 512, 2, 8, 3, 8192), (630, 1024, 1, 8, 2, 4096),
        (630, 1024, 1, 8, 3, 4096), (630, 1024, 2, 8, 2, 4096), (630, 2048, 1, 16, 1, 2048), (630, 1024, 2, 16, 1, 4096)]
for n, N, K, bg, L, G in sets:
    p = tfhe_amd.Params(n, N, K, bg, L, 8, 2)
    rs = np.random.RandomState(0)
    bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
    with tfhe_amd.Context(p, bk, None) as ctx:
        c = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
        lv1 = torch.empty((G, p.lv1 + 1), dtype=torch.int32, device="cuda")
        ctx.gate_bootstrapping_tlwe_to_tlwe_device(c, lv1); ctx.synchronize()
        t0 = time.time()
        for _ in range(2):
            ctx.gate_bootstrapping_tlwe_to_tlwe_device(c, lv1)
        ctx.synchronize()
        dt = (time.time() - t0) / 2
        print(f"N={N} K={K} bg={bg} L={L} n={n} {ctx.kernel_family():14s} G={G:5d} {dt*1e3:9.2f} ms {G/dt:9.0f} gates/s", flush=True)
