"""Small fixed workloads for rocprofv3 kernel-trace and counter passes, device resident.

  python3 tools/prof_target.py [mode]

  default        2 NAND batches of 8,192 gates, default parameters (wave-per-gate kernels, tiled key switch)
  latency1       8 single-gate NAND calls (workgroup-per-gate kernel, small key switch)
  latency256     4 NAND batches of 256 gates (workgroup-per-gate kernel, tiled key switch with slices)
  n2048          2 NAND batches of 2,048 gates at the N = 2048 set (n=800, K=1, L=3, T=8)
  reference-fft  2 NAND batches of 8,192 gates in the verification mode (reference's FFT replayed)
  keygen         4 evaluation-key generations from a secret key, default parameters
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import tfhe_amd  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
p = tfhe_amd.Params(800, 2048, 1, 8, 3, 8, 2) if mode == "n2048" else tfhe_amd.params_default()
rs = np.random.RandomState(0)
if mode == "keygen":
    s0, sr = rs.randint(0, 2, p.n).astype(np.uint8), rs.randint(0, 2, (p.K, p.N)).astype(np.uint8)
    for seed in range(4):
        tfhe_amd.Context.from_secret(p, s0, sr, 9.25119974676756e-5, 2.0 ** -25, seed=seed).close()
    print("done")
    sys.exit(0)
G, reps = {"default": (8192, 2), "latency1": (1, 8), "latency256": (256, 4), "n2048": (2048, 2),
           "reference-fft": (8192, 2)}[mode]
bk = rs.randint(0, 2**32, size=p.bk_shape(), dtype=np.uint64).astype(np.uint32)
ks = rs.randint(0, 2**32, size=p.ks_shape(), dtype=np.uint64).astype(np.uint32)
ctx = tfhe_amd.Context(p, bk, ks, reference_fft=mode == "reference-fft")
a = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
b = torch.randint(-2**31, 2**31 - 1, (G, p.n1), dtype=torch.int32, device="cuda")
out = torch.empty_like(a)
torch.cuda.synchronize()
for _ in range(reps):
    ctx.nand_device(a, b, out)
ctx.synchronize()
print("done")
