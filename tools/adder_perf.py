"""Config 4: 32-bit ripple-carry adder, gates batched per dependency level, timed on one GPU.
Two netlists: the NAND-only form (288 gates, 68 levels) and the XOR/MUX form (96 gates,
128 blind rotations, 33 levels)."""
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
backend = C.DeviceBackend(ctx)
for name, build in (("NAND-only", C.ripple_carry_adder), ("XOR/MUX", C.mux_adder)):
    gates, n_wires, a, b, zero, sums, cout = build(32)
    plan = C.Plan(gates, n_wires)
    levels = plan.levels
    dw = torch.randint(-2**31, 2**31 - 1, (n_wires, p.n1), dtype=torch.int32, device="cuda")
    C.evaluate(gates, dw, backend, plan=plan)
    torch.cuda.synchronize()
    t0 = time.time()
    C.evaluate(gates, dw, backend, plan=plan)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"32-bit adder, {name}: {len(gates)} gates in {len(levels)} levels: {dt*1e3:.1f} ms "
          f"({dt/len(levels)*1e3:.2f} ms per level)", flush=True)
