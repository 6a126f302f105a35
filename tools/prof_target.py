"""Small fixed workload for counter passes: 2 NAND batches of 8192 gates, device resident.

  python3 tools/prof_target.py [gates [reference-fft]]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import tfhe_amd  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = tfhe_amd.params_default()
rs = np.random.RandomState(0)
bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
ks = rs.randint(0, 2**32, size=p.ks_shape(), dtype=np.uint64).astype(np.uint32)
ctx = tfhe_amd.Context(p, bk, ks, reference_fft=len(sys.argv) > 2 and sys.argv[2] == "reference-fft")
a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
out = torch.empty_like(a)
for _ in range(2):
    ctx.nand_device(a, b, out)
ctx.synchronize()
print("done")
