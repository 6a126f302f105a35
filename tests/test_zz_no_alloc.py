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
