"""Throughput probe: random (not encrypted) inputs, device-resident, several batch sizes."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "oracle")
import tfhe_amd  # noqa: E402

p = tfhe_amd.params_default()
rs = np.random.RandomState(0)
bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
ks = rs.randint(0, 2**32, size=p.ks_shape(), dtype=np.uint64).astype(np.uint32)
ctx = tfhe_amd.Context(p, bk, ks)
ctx.enable_timing(True)
for G in [int(x) for x in sys.argv[1:]] or [256, 1024, 2048, 4096, 8192]:
    a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
    out = torch.empty_like(a)
    ctx.nand_device(a, b, out)
    ctx.synchronize()
    ctx.timing()
    reps = 2
    t0 = time.time()
    for _ in range(reps):
        ctx.nand_device(a, b, out)
    ctx.synchronize()
    dt = (time.time() - t0) / reps
    br, ksm, n = ctx.timing()
    print(f"G={G:6d}  {dt*1e3:9.2f} ms/batch  {G/dt:10.0f} gates/s   BR {br/n:9.2f} ms  KS {ksm/n:8.2f} ms", flush=True)

# host-pointer entry point (PCIe inclusive)
G = 8192
a = rs.randint(0, 2**32, size=(G, p.n1), dtype=np.uint64).astype(np.uint32)
b = rs.randint(0, 2**32, size=(G, p.n1), dtype=np.uint64).astype(np.uint32)
ctx.nand(a[:64], b[:64])
t0 = time.time()
ctx.nand(a, b)
dt = time.time() - t0
print(f"host-pointer nand, G={G}: {dt*1e3:.2f} ms  {G/dt:.0f} gates/s (PCIe inclusive, pageable host memory)", flush=True)

# same call on page-locked host memory (torch pinned tensors viewed as numpy)
ta = torch.from_numpy(a.view(np.int32)).pin_memory()
tb = torch.from_numpy(b.view(np.int32)).pin_memory()
pa, pb = ta.numpy().view(np.uint32), tb.numpy().view(np.uint32)
ctx.nand(pa[:64], pb[:64])
t0 = time.time()
ctx.nand(pa, pb)
dt = time.time() - t0
print(f"host-pointer nand, G={G}: {dt*1e3:.2f} ms  {G/dt:.0f} gates/s (PCIe inclusive, page-locked inputs, pageable output)", flush=True)
