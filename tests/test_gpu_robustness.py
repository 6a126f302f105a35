"""Behaviour of the C ABI around the hot path on a real GPU: error returns, several contexts,
caller streams, workspace growth.  Results are checked against the CPU oracle, bit for bit."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def keysets():
    p = O.params_default()
    return p, O.keyset(p, seed=41), O.keyset(p, seed=42)


def _enc(p, s0, n, seed):
    rs = np.random.RandomState(seed)
    r = O.rng(seed)
    ba, bb = rs.randint(0, 2, n), rs.randint(0, 2, n)
    return ba, bb, O.encrypt_bits(ba, s0, p, r), O.encrypt_bits(bb, s0, p, r)


def test_context_without_a_key_refuses_the_stage_that_needs_it(keysets):
    import tfhe_amd

    p, (s0, sr, bk, ks), _ = keysets
    _, _, a, b = _enc(p, s0, 2, 1)
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, None) as ctx:
        with pytest.raises(tfhe_amd.TfheAmdError, match="key-switching key"):
            ctx.nand(a, b)
        c = np.stack([O.nand_combine(x, y, p) for x, y in zip(a, b)])
        lv1 = ctx.gate_bootstrapping_tlwe_to_tlwe(c)       # needs the bootstrapping key only
        assert lv1.shape == (2, p.lv1 + 1)
    with tfhe_amd.Context(tfhe_amd.params_default(), None, ks) as ctx:
        with pytest.raises(tfhe_amd.TfheAmdError, match="bootstrapping key"):
            ctx.nand(a, b)
        assert np.array_equal(ctx.identity_key_switch(lv1), np.stack([O.identity_key_switch(l, ks, p) for l in lv1]))


def test_unsupported_and_invalid_parameter_sets():
    import tfhe_amd

    with pytest.raises(tfhe_amd.TfheAmdError):      # ring beyond every kernel
        tfhe_amd.Context(tfhe_amd.Params(8, 4096, 1, 8, 2, 5, 2), None, None)
    with pytest.raises(tfhe_amd.TfheAmdError):      # L * bg_bit > 31
        tfhe_amd.Context(tfhe_amd.Params(8, 16, 1, 8, 4, 5, 2), None, None)
    with pytest.raises(tfhe_amd.TfheAmdError):      # no such device
        tfhe_amd.Context(tfhe_amd.params_default(), None, None, device=99)
    with pytest.raises(ValueError):                 # key size does not match the parameters
        tfhe_amd.Context(tfhe_amd.params_default(), np.zeros(10, dtype=np.uint32), None)


def test_two_contexts_with_different_keys_interleaved(keysets):
    import tfhe_amd

    p, k1, k2 = keysets
    ctxs = [tfhe_amd.Context(tfhe_amd.params_default(), k[2], k[3]) for k in (k1, k2)]
    octxs = [O.Context(p, k[2], k[3]) for k in (k1, k2)]
    try:
        for rnd in range(2):
            for i, k in enumerate((k1, k2)):
                ba, bb, a, b = _enc(p, k[0], 3, 10 * rnd + i)
                got = ctxs[i].nand(a, b)
                assert np.array_equal(got, octxs[i].nand_batch(a, b))
                assert np.array_equal(O.decrypt_bits(got, k[0], p), 1 - (ba & bb))
    finally:
        for c in ctxs:
            c.close()


def test_caller_streams_and_workspace_growth(keysets):
    """Device-pointer calls on a torch side stream and on the default stream; batch sizes 3, 700, 5
    so that the workspace grows and both blind-rotation forms run on one context."""
    import tfhe_amd
    import torch

    p, (s0, sr, bk, ks), _ = keysets
    octx = O.Context(p, bk, ks)
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, ks) as ctx:
        side = torch.cuda.Stream()
        for n, stream in ((3, side), (700, None), (5, torch.cuda.default_stream()), (3, side)):
            ba, bb, a, b = _enc(p, s0, n, 100 + n)
            with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
                da = torch.from_numpy(a.view(np.int32)).cuda(non_blocking=False)
                db = torch.from_numpy(b.view(np.int32)).cuda(non_blocking=False)
                out = torch.empty_like(da)
            torch.cuda.synchronize()
            ctx.nand_device(da, db, out, stream=stream)
            if stream is None:
                ctx.synchronize()
            else:
                stream.synchronize()
            got = out.cpu().numpy().view(np.uint32)
            sample = np.arange(n) if n <= 8 else np.array([0, 1, n // 2, n - 2, n - 1])
            assert np.array_equal(got[sample], octx.nand_batch(a[sample], b[sample]))
            assert np.array_equal(O.decrypt_bits(got, s0, p), 1 - (ba & bb))


def test_device_pointer_argument_checks(keysets):
    import tfhe_amd
    import torch

    p, (s0, sr, bk, ks), _ = keysets
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, ks) as ctx:
        a = torch.zeros((2, p.n + 1), dtype=torch.int32, device="cuda")
        with pytest.raises(ValueError):            # host tensor where a GPU tensor is required
            ctx.nand_device(a.cpu(), a, a)
        with pytest.raises(ValueError):            # wrong row length
            ctx.nand_device(a[:, :-1].contiguous(), a, a)
        with pytest.raises(ValueError):            # per-gate codes of the wrong length
            ctx.gate_device(torch.zeros(3, dtype=torch.uint8, device="cuda"), a, a, torch.empty_like(a))
        lib = tfhe_amd.library()
        assert lib.tfhe_amd_nand_batch_device(ctx._h, None, None, None, 2, None) == 1   # null arrays
        assert b"null" in lib.tfhe_amd_last_error()
        assert lib.tfhe_amd_nand_batch_device(ctx._h, None, None, None, 0, None) == 0   # empty batch: nothing to check
        assert lib.tfhe_amd_level_batch_device(ctx._h, None, None, None, None, None, 1, 2, None) == 1


def test_two_host_threads_each_with_its_own_context(keysets):
    """One context per host thread is the supported way to call from several threads; the library
    keeps its error text per thread and its device state per context."""
    import threading

    import tfhe_amd

    p, k1, k2 = keysets
    results, errors = {}, []

    def work(tag, k, seed):
        try:
            with tfhe_amd.Context(tfhe_amd.params_default(), k[2], k[3]) as ctx:
                ba, bb, a, b = _enc(p, k[0], 40, seed)
                outs = [ctx.nand(a, b) for _ in range(3)]
                assert all(np.array_equal(outs[0], o) for o in outs[1:])
                results[tag] = (ba, bb, a, b, outs[0])
        except Exception as e:  # surfaced in the main thread
            errors.append((tag, repr(e)))

    ts = [threading.Thread(target=work, args=(i, k, 300 + i)) for i, k in enumerate((k1, k2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for i, k in enumerate((k1, k2)):
        ba, bb, a, b, got = results[i]
        assert np.array_equal(O.decrypt_bits(got, k[0], p), 1 - (ba & bb))
        assert np.array_equal(got[:4], O.Context(p, k[2], k[3]).nand_batch(a[:4], b[:4]))


def test_back_to_back_calls_on_two_streams_share_the_workspace_safely(keysets):
    """Two device-pointer calls on different streams with no synchronisation between them: the
    second must not overwrite the context's workspace while the first still reads it."""
    import tfhe_amd
    import torch

    p, (s0, sr, bk, ks), _ = keysets
    octx = O.Context(p, bk, ks)
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, ks) as ctx:
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        ba1, bb1, a1, b1 = _enc(p, s0, 2300, 901)
        ba2, bb2, a2, b2 = _enc(p, s0, 600, 902)
        d = [torch.from_numpy(x.view(np.int32)).cuda() for x in (a1, b1, a2, b2)]
        o1, o2 = torch.empty_like(d[0]), torch.empty_like(d[2])
        torch.cuda.synchronize()
        for _ in range(2):
            ctx.nand_device(d[0], d[1], o1, stream=sa)
            ctx.nand_device(d[2], d[3], o2, stream=sb)
        torch.cuda.synchronize()
        g1, g2 = o1.cpu().numpy().view(np.uint32), o2.cpu().numpy().view(np.uint32)
        assert np.array_equal(O.decrypt_bits(g1, s0, p), 1 - (ba1 & bb1))
        assert np.array_equal(O.decrypt_bits(g2, s0, p), 1 - (ba2 & bb2))
        pick1, pick2 = np.array([0, 1, 1150, 2299]), np.array([0, 299, 599])
        assert np.array_equal(g1[pick1], octx.nand_batch(a1[pick1], b1[pick1]))
        assert np.array_equal(g2[pick2], octx.nand_batch(a2[pick2], b2[pick2]))


def test_each_parameter_set_runs_on_the_kernels_meant_for_it(keysets):
    import tfhe_amd

    p, (s0, sr, bk, ks), _ = keysets
    dp = tfhe_amd.params_default()
    cases = [({}, "ntt512"), ({"generic_kernels": True}, "generic"), ({"reference_fft": True}, "reference-fft-wave"),
             ({"reference_fft": True, "generic_kernels": True}, "reference-fft-workgroup")]
    for kw, want in cases:
        with tfhe_amd.Context(dp, bk, ks, **kw) as ctx:
            assert ctx.kernel_family() == want
    with tfhe_amd.Context(dp, None, ks) as ctx:
        assert ctx.kernel_family() == "none"
    big = tfhe_amd.Params(800, 2048, 1, 8, 3, 8, 2)
    rs = np.random.RandomState(0)
    with tfhe_amd.Context(big, rs.randint(0, 2**32, size=big.bk_shape(), dtype=np.uint64).astype(np.uint32), None) as ctx:
        assert ctx.kernel_family() == "ntt-workgroup"
    toy = tfhe_amd.Params(10, 16, 2, 8, 2, 5, 2)
    with tfhe_amd.Context(toy, rs.randint(0, 2**32, size=toy.bk_shape(), dtype=np.uint64).astype(np.uint32), None) as ctx:
        assert ctx.kernel_family() == "generic"


def test_host_pointer_calls_return_what_device_resident_calls_return(keysets):
    """A batch that is not a multiple of anything the kernels tile by (gates in flight, key-switch
    tiles), through the host-pointer entry points and through the device-pointer ones."""
    import torch

    import tfhe_amd

    p, (s0, sr, bk, ks), _ = keysets
    n = 2 * 2048 + 700
    rs = np.random.RandomState(5)
    a, b, x0 = (rs.randint(0, 2**32, size=(n, p.n + 1), dtype=np.uint64).astype(np.uint32) for _ in range(3))
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, ks) as ctx:
        da, db, dx = (torch.from_numpy(v.view(np.int32)).cuda() for v in (a, b, x0))
        out = torch.empty_like(da)
        ctx.nand_device(da, db, out)
        ctx.synchronize()
        assert np.array_equal(ctx.nand(a, b), out.cpu().numpy().view(np.uint32))
        ctx.gate_device("xor", da, db, out)
        ctx.synchronize()
        assert np.array_equal(ctx.gate("xor", a, b), out.cpu().numpy().view(np.uint32))
        ctx.mux_device(da, db, dx, out)
        ctx.synchronize()
        assert np.array_equal(ctx.mux(a, b, x0), out.cpu().numpy().view(np.uint32))
        # and against the oracle on a sample of rows
        octx = O.Context(p, bk, ks)
        rows = np.array([0, 2047, 2048, 4095, 4096, n - 1])
        assert np.array_equal(ctx.nand(a, b)[rows], octx.nand_batch(a[rows], b[rows]))
        # every row-wise host entry point returns what two smaller calls return
        h = 2500
        for call in (ctx.bootstrap, ctx.not_, ctx.gate_bootstrapping_tlwe_to_tlwe):
            assert np.array_equal(call(a), np.concatenate([call(a[:h]), call(a[h:])])), call.__name__
        lv1 = ctx.gate_bootstrapping_tlwe_to_tlwe(a)
        assert np.array_equal(ctx.identity_key_switch(lv1),
                              np.concatenate([ctx.identity_key_switch(lv1[:h]), ctx.identity_key_switch(lv1[h:])]))
        assert np.array_equal(ctx.identity_key_switch(lv1), ctx.bootstrap(a))
