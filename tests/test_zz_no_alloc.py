"""tfhe_amd_nand_batch_device_noalloc (the entry point that allocates nothing) against
tfhe_amd_nand_batch_device: same words, on two calls in a row, the second with a smaller batch, plus
the size limit.  Bit-exact (integer path)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
import tfhe_amd
from tfhe_amd.binding import TfheAmdError

pytestmark = pytest.mark.gpu


def _rand(shape, seed):
    return np.random.RandomState(seed).randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)


def test_noalloc_matches_allocating_entry_point():
    p = tfhe_amd.params_default()
    bk, ks = _rand(p.bk_shape(), 1), _rand(p.ks_shape(), 2)
    dev = torch.device("cuda", 0)
    with tfhe_amd.Context(p, bk, ks, device=0) as ctx:
        assert ctx.kernel_family() == "ntt512"
        # 700 gates: the wave-per-gate kernels; 37: the workgroup-per-gate form and the tiled key switch; 3: the
        # small key switch.  Each smaller call reuses the front of the workspace the larger one left behind.
        for g, seed in ((700, 10), (37, 11), (3, 12)):
            a = torch.from_numpy(_rand((g, p.n1), seed).view(np.int32)).to(dev)
            b = torch.from_numpy(_rand((g, p.n1), seed + 100).view(np.int32)).to(dev)
            want = torch.empty_like(a)
            ctx.nand_device(a, b, want)
            ctx.synchronize()
            got = torch.full_like(a, 0x5A5A5A5A)
            ctx.nand_device_noalloc(a, b, got)
            ctx.synchronize()
            assert torch.equal(got, want), f"batch of {g}: no-allocation entry point differs"
        # a second context on the same GPU shares the array
        with tfhe_amd.Context(p, bk, ks, device=0) as ctx2:
            got2 = torch.empty_like(a)
            ctx2.nand_device_noalloc(a, b, got2)
            ctx2.synchronize()
            assert torch.equal(got2, want)
        # size limit: (n+1) + (N*K+1) words per gate in 16 MiB
        limit = (ctx.NOALLOC_SCRATCH_BYTES // 4) // (p.n1 + p.lv1 + 1)
        a = torch.zeros((limit + 1, p.n1), dtype=torch.int32, device=dev)
        out = torch.empty_like(a)
        with pytest.raises(TfheAmdError, match="static workspace"):
            ctx.nand_device_noalloc(a, a, out)
        ctx.nand_device_noalloc(a[:limit], a[:limit], out[:limit])   # the largest accepted batch runs
        ctx.synchronize()
        ref = torch.empty_like(out[:limit])
        ctx.nand_device(a[:limit], a[:limit], ref)
        ctx.synchronize()
        assert torch.equal(out[:limit], ref)


def test_noalloc_allocates_nothing_and_is_capturable():
    """The property the entry point exists for.  (1) On a fresh context that has no workspace of
    its own yet, a call leaves the free device memory unchanged.  (2) The call is recorded into a
    HIP graph (stream capture in global mode, which turns any allocation, synchronisation or other
    unsafe API call on any thread into a capture error), the graph is replayed twice on new inputs
    written into the captured buffers, the second time with the output poisoned first, and every
    word is compared with the CPU oracle and decrypted."""
    import sys, pathlib
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / "oracle"))
    import oracle_py as O  # the checker

    po = O.params_default()
    s0, _, bk, ks = O.keyset(po, seed=99)
    p = tfhe_amd.params_default()
    dev = torch.device("cuda", 0)
    octx = O.Context(po, bk, ks)
    r = O.rng(31)
    G = 3
    a = torch.empty((G, p.n1), dtype=torch.int32, device=dev)
    b = torch.empty_like(a)
    out = torch.empty_like(a)
    # another context runs the same kernels once, so that what the HIP runtime sets up on a kernel's
    # first launch in a process (code objects, kernel-argument pools) is not charged to the call below
    with tfhe_amd.Context(p, bk, ks, device=0) as warm:
        a.zero_(); b.zero_()
        warm.nand_device_noalloc(a, b, out)
        warm.synchronize()
    with tfhe_amd.Context(p, bk, ks, device=0) as ctx:
        bits = [(np.array([0, 1, 1]), np.array([1, 1, 0])), (np.array([1, 0, 1]), np.array([1, 0, 0])),
                (np.array([0, 0, 1]), np.array([0, 1, 1]))]
        enc = [(O.encrypt_bits(x, s0, po, r), O.encrypt_bits(y, s0, po, r)) for x, y in bits]
        want = [octx.nand_batch(x, y) for x, y in enc]

        def load(i):
            a.copy_(torch.from_numpy(enc[i][0].view(np.int32)))
            b.copy_(torch.from_numpy(enc[i][1].view(np.int32)))

        # (1) nothing allocated: this context has evaluated nothing yet, so it holds no workspace lane
        load(0)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        ctx.nand_device_noalloc(a, b, out)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(0)[0] == free_before, "the call changed the free device memory"
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want[0])

        # (2) captured into a graph on a side stream
        side = torch.cuda.Stream(device=0)
        graph = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream(0))
        with torch.cuda.graph(graph, stream=side, capture_error_mode="global"):
            ctx.nand_device_noalloc(a, b, out, stream=side)
        for i in (1, 2):
            load(i)
            out.fill_(0x5A5A5A5A)
            graph.replay()
            torch.cuda.synchronize()
            got = out.cpu().numpy().view(np.uint32)
            assert np.array_equal(got, want[i]), f"graph replay {i} differs from the oracle"
            assert np.array_equal(O.decrypt_bits(got, s0, po), 1 - (bits[i][0] & bits[i][1]))
        # after the capture the plain call still works and is ordered behind the graph's launches
        load(0)
        ctx.nand_device_noalloc(a, b, out)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want[0])


def test_noalloc_rejects_sets_that_need_a_launch_workspace():
    """A ring beyond LDS (N = 4096) runs on kernels that take a workspace per launch: the entry point that
    allocates nothing refuses it, the allocating one evaluates it."""
    p = tfhe_amd.Params(2, 4096, 1, 8, 1, 2, 2)
    bk, ks = _rand(p.bk_shape(), 3), _rand(p.ks_shape(), 4)
    dev = torch.device("cuda", 0)
    with tfhe_amd.Context(p, bk, ks, device=0) as ctx:
        a = torch.from_numpy(_rand((2, p.n1), 5).view(np.int32)).to(dev)
        out = torch.empty_like(a)
        with pytest.raises(TfheAmdError, match="per-launch workspace"):
            ctx.nand_device_noalloc(a, a, out)
        ctx.nand_device(a, a, out)
        ctx.synchronize()
