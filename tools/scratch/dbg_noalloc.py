import sys, pathlib
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import oracle_py as O, tfhe_amd
po = O.params_default()
s0, _, bk, ks = O.keyset(po, seed=99)
p = tfhe_amd.params_default()
dev = torch.device("cuda", 0)
octx = O.Context(po, bk, ks)
r = O.rng(31)
bits = [(np.array([0, 1, 1]), np.array([1, 1, 0])), (np.array([1, 0, 1]), np.array([1, 0, 0])), (np.array([0, 0, 1]), np.array([0, 1, 1]))]
enc = [(O.encrypt_bits(x, s0, po, r), O.encrypt_bits(y, s0, po, r)) for x, y in bits]
want = [octx.nand_batch(x, y) for x, y in enc]
G = 3
a = torch.empty((G, p.n1), dtype=torch.int32, device=dev); b = torch.empty_like(a); out = torch.empty_like(a)
def load(i):
    a.copy_(torch.from_numpy(enc[i][0].view(np.int32))); b.copy_(torch.from_numpy(enc[i][1].view(np.int32)))
with tfhe_amd.Context(p, bk, ks, device=0) as ctx:
    for i in range(3):
        load(i); out.fill_(0x5A5A5A5A)
        ctx.nand_device(a, b, out); torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint32)
        print("eager alloc", i, np.array_equal(got, want[i]), np.count_nonzero(got != want[i]))
        out.fill_(0x5A5A5A5A)
        ctx.nand_device_noalloc(a, b, out); torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint32)
        print("eager noalloc", i, np.array_equal(got, want[i]), np.count_nonzero(got != want[i]))
    side = torch.cuda.Stream(device=0)
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream(0))
    with torch.cuda.graph(graph, stream=side, capture_error_mode="global"):
        ctx.nand_device_noalloc(a, b, out, stream=side)
    for rep in range(3):
        for i in (1, 2, 0):
            load(i); out.fill_(0x5A5A5A5A)
            graph.replay(); torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint32)
            d = got != want[i]
            print("replay", rep, i, np.array_equal(got, want[i]), np.count_nonzero(d), "per gate", d.sum(axis=1), "first diffs", np.nonzero(d[0])[0][:10], (got[0][d[0]][:6].astype(np.int64) - want[i][0][d[0]][:6].astype(np.int64)))
            print("   decrypt", O.decrypt_bits(got, s0, po), 1 - (bits[i][0] & bits[i][1]))
