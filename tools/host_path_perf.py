"""Where the time of the host-pointer entry point goes: device-resident batch, the same batch in
chunks, and the host-pointer call."""
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
ctx = tfhe_amd.Context(p, bk, ks)
G = 8192
a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
out = torch.empty_like(a)
ha, hb = a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32)


def timed(f, reps=3):
    f()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"device, one call of {G}: {timed(lambda: ctx.nand_device(a, b, out)):.2f} ms")
for c in (2048, 4096):
    def chunks():
        for s in range(0, G, c):
            ctx.nand_device(a[s:s + c], b[s:s + c], out[s:s + c])
    print(f"device, chunks of {c}: {timed(chunks):.2f} ms")
print(f"host pointers: {timed(lambda: ctx.nand(ha, hb)):.2f} ms")
t0 = time.perf_counter(); da = torch.from_numpy(ha.view(np.int32)).cuda(); torch.cuda.synchronize()
print(f"pageable H2D of one operand ({ha.nbytes / 1e6:.1f} MB): {(time.perf_counter() - t0) * 1e3:.2f} ms")
t0 = time.perf_counter(); _ = out.cpu(); print(f"pageable D2H of the result: {(time.perf_counter() - t0) * 1e3:.2f} ms")
