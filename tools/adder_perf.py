"""Config 4: 32-bit ripple-carry adder, gates batched per dependency level, timed on one GPU."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import tfhe_amd  # noqa: E402
from tfhe_amd import circuit as C  # noqa: E402

p = tfhe_amd.params_default()
rs = np.random.RandomState(0)
bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
ks = rs.randint(0, 2**32, size=p.ks_shape(), dtype=np.uint64).astype(np.uint32)
ctx = tfhe_amd.Context(p, bk, ks)
gates, n_wires, a, b, zero, sums, cout = C.ripple_carry_adder(32)
levels = C.levelize(gates, n_wires)
dw = torch.randint(-2**31, 2**31 - 1, (n_wires, p.n1), dtype=torch.int32, device="cuda")


idx = [tuple(torch.tensor([gates[g][k] for g in lv], device="cuda") for k in range(3)) for lv in levels]


# everything on torch's current stream: gathers, the library's kernels, scatters
def run():
    st = torch.cuda.current_stream()
    for o, ia, ib in idx:
        u, v = dw.index_select(0, ia), dw.index_select(0, ib)
        out = torch.empty_like(u)
        ctx.nand_device(u, v, out, stream=st)
        dw.index_copy_(0, o, out)


run()
torch.cuda.synchronize()
t0 = time.time()
run()
torch.cuda.synchronize()
dt = time.time() - t0
print(f"32-bit adder: {len(gates)} gates in {len(levels)} levels: {dt*1e3:.1f} ms ({dt/len(levels)*1e3:.2f} ms per level, "
      f"{len(gates)/dt:.0f} gates/s)")
