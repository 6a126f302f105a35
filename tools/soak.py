"""Determinism soak: the same batches evaluated many times must give identical ciphertexts (the
blind-rotation paths of the default set: throughput form, latency form, reference-FFT mode; and the
workgroup kernels of the N = 2048 set)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import tfhe_amd  # noqa: E402

p = tfhe_amd.params_default()
rs = np.random.RandomState(0)
bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
ks = rs.randint(0, 2**32, size=p.ks_shape(), dtype=np.uint64).astype(np.uint32)
for name, kw, G, reps in (("throughput", {}, 8192, 30), ("latency", {}, 200, 60), ("reference-fft", {"reference_fft": True}, 2048, 20)):
    ctx = tfhe_amd.Context(p, bk, ks, **kw)
    a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    first = torch.empty_like(a)
    ctx.nand_device(a, b, first); ctx.synchronize()
    out = torch.empty_like(a)
    bad = 0
    for _ in range(reps):
        out.zero_()
        ctx.nand_device(a, b, out); ctx.synchronize()
        bad += int((out != first).any())
    print(f"{name}: {reps} repetitions of {G} gates, {bad} differing", flush=True)
    ctx.close()
    assert bad == 0

p2 = tfhe_amd.Params(800, 2048, 1, 8, 3, 8, 2)
bk2 = rs.randint(0, 2**32, size=p2.bk_shape(), dtype=np.uint64).astype(np.uint32)
ks2 = rs.randint(0, 2**32, size=p2.ks_shape(), dtype=np.uint64).astype(np.uint32)
with tfhe_amd.Context(p2, bk2, ks2) as ctx:
    G, reps = 512, 12
    a = torch.randint(-2**31, 2**31 - 1, (G, p2.n1), dtype=torch.int32, device="cuda")
    b = torch.randint(-2**31, 2**31 - 1, (G, p2.n1), dtype=torch.int32, device="cuda")
    first = torch.empty_like(a)
    ctx.nand_device(a, b, first); ctx.synchronize()
    out = torch.empty_like(a)
    bad = 0
    for _ in range(reps):
        out.zero_()
        ctx.nand_device(a, b, out); ctx.synchronize()
        bad += int((out != first).any())
    print(f"N = 2048 set: {reps} repetitions of {G} gates, {bad} differing", flush=True)
    assert bad == 0
