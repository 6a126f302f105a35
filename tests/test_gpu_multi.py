"""Gates sharded over several GPUs (or over several contexts on one GPU) and several host threads
on one shared key, against the CPU oracle.  Bit-exact.  The reference allows both usages: a gate
reads only its two inputs and the read-only keys (reference src/hom_nand.hpp:74-81)."""
import os
import socket
import threading

import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def keys():
    p = O.params_default()
    s0, sr, bk, ks = O.keyset(p, seed=4321)
    return p, s0, bk, ks


def _inputs(p, s0, n, seed):
    rs = np.random.RandomState(seed)
    bits_a, bits_b = rs.randint(0, 2, n), rs.randint(0, 2, n)
    r = O.rng(seed)
    return bits_a, bits_b, O.encrypt_bits(bits_a, s0, p, r), O.encrypt_bits(bits_b, s0, p, r)


def _device_list():
    """Every GPU of the box, and at least three entries (aliased on a small box) so that the split is ragged."""
    import torch

    n = torch.cuda.device_count()
    devs = list(range(n))
    while len(devs) < 3:
        devs.append(devs[-1])
    return devs


def test_multi_context_ragged_batch_bit_exact(keys):
    import tfhe_amd

    p, s0, bk, ks = keys
    devs = _device_list()
    G = 8 * len(devs) + 5   # not a multiple of the device count
    bits_a, bits_b, a, b = _inputs(p, s0, G, seed=21)
    with tfhe_amd.MultiContext(tfhe_amd.params_default(), bk, ks, devs) as m:
        assert m.device_count() == len(devs)
        assert all(m.kernel_family(i) == "ntt512" for i in range(len(devs)))
        bounds = [m.shard(G, i) for i in range(len(devs))]
        assert bounds[0][0] == 0 and bounds[-1][1] == G
        assert all(x[1] == y[0] for x, y in zip(bounds, bounds[1:]))
        assert bounds == [(G * i // len(devs), G * (i + 1) // len(devs)) for i in range(len(devs))]
        got = m.nand(a, b)
        got_xor = m.gate("xor", a, b)
        # fewer gates than devices: some blocks are empty
        got_small = m.nand(a[:2], b[:2])
        assert m.nand(a[:0], b[:0]).shape == (0, p.n + 1)
    octx = O.Context(p, bk, ks)
    want = octx.nand_batch(a, b)
    assert np.array_equal(got, want)
    assert np.array_equal(got_small, want[:2])
    assert np.array_equal(O.decrypt_bits(got, s0, p), 1 - (bits_a & bits_b))
    assert np.array_equal(O.decrypt_bits(got_xor, s0, p), bits_a ^ bits_b)


def test_multi_context_device_resident_blocks(keys):
    """tfhe_amd_multi_nand_batch_device: one device-resident block per entry, keys already on the first GPU."""
    import tfhe_amd
    import torch

    p, s0, bk, ks = keys
    devs = _device_list()
    _, _, a, b = _inputs(p, s0, 4 * len(devs) + 1, seed=22)
    d0 = torch.device("cuda", devs[0])
    bk_t = torch.from_numpy(bk.reshape(-1).view(np.int32)).to(d0)
    ks_t = torch.from_numpy(ks.reshape(-1).view(np.int32)).to(d0)
    torch.cuda.synchronize()
    with tfhe_amd.MultiContext(tfhe_amd.params_default(), bk_t, ks_t, devs, keys_on_device=True) as m:
        ta, tb, to = [], [], []
        for i, d in enumerate(devs):
            lo, hi = m.shard(a.shape[0], i)
            dev = torch.device("cuda", d)
            ta.append(torch.from_numpy(a[lo:hi].view(np.int32)).to(dev))
            tb.append(torch.from_numpy(b[lo:hi].view(np.int32)).to(dev))
            to.append(torch.empty_like(ta[-1]))
        torch.cuda.synchronize()
        m.nand_device(ta, tb, to)
        m.synchronize()
        got = np.concatenate([t.cpu().numpy().view(np.uint32) for t in to])
    assert np.array_equal(got, O.Context(p, bk, ks).nand_batch(a, b))


def test_multi_context_device_blocks_ordered_on_producer_streams(keys):
    """The operands are produced by torch kernels on a side stream of every device and consumed
    there again, with no device-wide synchronisation anywhere: the call is given the producing
    streams (the default: torch's current stream on every entry's GPU)."""
    import tfhe_amd
    import torch

    p, s0, bk, ks = keys
    devs = _device_list()
    _, _, a, b = _inputs(p, s0, 3 * len(devs) + 2, seed=23)
    want = O.Context(p, bk, ks).nand_batch(a, b)
    with tfhe_amd.MultiContext(tfhe_amd.params_default(), bk, ks, devs) as m:
        side = {d: torch.cuda.Stream(device=d) for d in set(devs)}
        host, ta, tb, to, neg = [], [], [], [], []
        for i, d in enumerate(devs):
            lo, hi = m.shard(a.shape[0], i)
            # pinned host copies: the uploads below are really asynchronous
            host.append((torch.from_numpy((0 - a[lo:hi]).view(np.int32)).pin_memory(),
                         torch.from_numpy((b[lo:hi] ^ np.uint32(0x5a5a5a5a)).view(np.int32)).pin_memory()))
        for rep in range(2):   # the second round reuses the workspaces the first one left in flight
            ta, tb, to, neg = [], [], [], []
            for i, d in enumerate(devs):
                with torch.cuda.stream(side[d]):
                    big = torch.empty((1 << 22,), dtype=torch.int32, device=f"cuda:{d}")
                    for _ in range(8):   # keeps the stream busy while the host runs ahead
                        big.add_(1)
                    na = host[i][0].to(f"cuda:{d}", non_blocking=True)
                    xb = host[i][1].to(f"cuda:{d}", non_blocking=True)
                    ta.append(0 - na)              # a, produced by a torch kernel on the side stream
                    tb.append(xb ^ 0x5a5a5a5a)     # b likewise
                    to.append(torch.empty_like(ta[-1]))
            m.nand_device(ta, tb, to, streams=[side[d] for d in devs])
            for i, d in enumerate(devs):
                with torch.cuda.stream(side[d]):
                    neg.append(0 - to[i])          # consumed on the same stream, no synchronisation in between
            for s in side.values():
                s.synchronize()
            got = np.concatenate([(0 - t).cpu().numpy().view(np.uint32) for t in neg])
            assert np.array_equal(got, want), f"round {rep}"


def test_eight_host_threads_share_one_context(keys):
    """Concurrent calls on ONE context from eight host threads (ctypes releases the GIL): batches of
    different sizes, host-pointer and device-pointer entry points mixed, every word as the oracle's."""
    import tfhe_amd
    import torch

    p, s0, bk, ks = keys
    octx = O.Context(p, bk, ks)
    sizes = [1, 3, 17, 40, 2, 9, 64, 5]
    work = [_inputs(p, s0, g, seed=100 + i) for i, g in enumerate(sizes)]
    want = [octx.nand_batch(w[2], w[3]) for w in work]
    got, errors = [None] * len(sizes), []
    with tfhe_amd.Context(tfhe_amd.params_default(), bk, ks) as ctx:
        start = threading.Barrier(len(sizes))

        def run(i):
            try:
                _, _, a, b = work[i]
                start.wait()
                for rep in range(3):
                    if i % 2 == 0:
                        out = ctx.nand(a, b)
                    else:
                        stream = torch.cuda.Stream()
                        with torch.cuda.stream(stream):
                            ta = torch.from_numpy(a.view(np.int32)).cuda()
                            tb = torch.from_numpy(b.view(np.int32)).cuda()
                            to = torch.empty_like(ta)
                        ctx.nand_device(ta, tb, to, stream=stream)
                        stream.synchronize()
                        out = to.cpu().numpy().view(np.uint32)
                    if got[i] is not None and not np.array_equal(got[i], out):
                        raise AssertionError(f"thread {i}: repetition {rep} differs")
                    got[i] = out
            except Exception as e:  # surfaced in the main thread
                errors.append((i, repr(e)))

        ts = [threading.Thread(target=run, args=(i,)) for i in range(len(sizes))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    assert not errors, errors
    for i in range(len(sizes)):
        assert np.array_equal(got[i], want[i]), f"thread {i} ({sizes[i]} gates)"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gloo_worker(rank, world, port, n_gates, q):
    """One rank of distributed.nand_sharded with Context.nand_device as the evaluator.  Ranks share
    the box's GPUs round robin (the RCCL run has one GPU per rank; gloo moves host tensors)."""
    import torch
    import torch.distributed as dist

    import tfhe_amd
    from tfhe_amd import distributed as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev_index = rank % torch.cuda.device_count()
        dev = torch.device("cuda", dev_index)
        p = O.params_default()
        if rank == 0:
            s0, sr, bk, ks = O.keyset(p, seed=4321)
            bits_a, bits_b, a, b = _inputs(p, s0, n_gates, seed=33)
            a_t, b_t = torch.from_numpy(a.view(np.int32)), torch.from_numpy(b.view(np.int32))
        else:
            bk = ks = a_t = b_t = None
        bk_t, ks_t = D.broadcast_keys(p, bk, ks, src=0)          # host tensors over gloo
        ctx = tfhe_amd.Context(tfhe_amd.params_default(), bk_t.to(dev), ks_t.to(dev), device=dev_index,
                               keys_on_device=True)

        def evaluate(x, y):
            xd, yd = x.to(dev), y.to(dev)
            out = torch.empty_like(xd)
            ctx.nand_device(xd, yd, out, stream=torch.cuda.current_stream(dev))
            torch.cuda.synchronize(dev)
            return out.cpu()

        full = D.nand_sharded(evaluate, a_t, b_t, n_gates, p.n + 1, src=0)
        ctx.close()
        if rank == 0:
            want = O.Context(p, bk, ks).nand_batch(a, b)
            ok = np.array_equal(full.numpy().view(np.uint32), want)
            ok = ok and np.array_equal(O.decrypt_bits(want, s0, p), 1 - (bits_a & bits_b))
            q.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_nand_sharded_gloo_world2_on_gpu():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, 11, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(240)
        assert pr.exitcode == 0
    assert q.get(timeout=5) is True


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher's environment: the script starts its two ranks itself
    (fresh processes, before the parent touches a GPU), rank 0's JSON line comes back, every rank's
    outputs decrypt correctly.  gloo stands in for RCCL here (one GPU shared by the ranks); the strong
    (fixed total, block split) and the weak sizing are both run at a small size."""
    import json
    import pathlib
    import subprocess
    import sys

    root = pathlib.Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["TFHE_BENCH_BACKEND"] = "gloo"
    for extra, scaling, per_rank in ((["--gates-per-gpu", "96"], "weak", [96, 96]), (["--total-gates", "131"], "strong", [65, 66])):
        r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                            "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["scaling"] == scaling
        assert line["distributed"]["world_size"] == 2 and line["distributed"]["launcher"] == "self"
        assert [e["gates"] for e in line["distributed"]["per_rank"]] == per_rank
        assert line["verified"] == {**line["verified"], "gates_decrypted": sum(per_rank), "wrong": 0}
    # a rank that fails takes the whole launch down with a non-zero status
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--gates-per-gpu", "-5"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode != 0


def test_bench_single_rank_through_rccl():
    """The benchmark's collective path with the real backend: one rank creates an RCCL process group on its
    GPU and runs the key broadcasts, the barriers, the gather of the per-rank times and the reduction of the
    wrong-bit count through it (gloo stands in for RCCL in the two-rank test above, because two RCCL ranks
    cannot share one GPU).  The line must say that RCCL ran, and every output must decrypt correctly."""
    import json
    import pathlib
    import subprocess
    import sys

    root = pathlib.Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TFHE_BENCH_BACKEND")}
    env["TFHE_BENCH_FORCE_DIST"] = "1"
    env["MASTER_PORT"] = "29547"
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                        "--gates-per-gpu", "160", "--no-cpu-baseline", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["distributed"]["backend"] == "nccl" and line["distributed"]["world_size"] == 1
    assert line["distributed"]["key_broadcast_s"] > 0.0
    assert line["verified"] == {**line["verified"], "gates_decrypted": 160, "wrong": 0}


def _rccl_worker(port, n_gates, q):
    """distributed.broadcast_keys and nand_sharded over an RCCL process group of one rank: device
    tensors through the real backend (two RCCL ranks cannot share one GPU)."""
    import torch
    import torch.distributed as dist

    import tfhe_amd
    from tfhe_amd import distributed as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        p = O.params_default()
        s0, sr, bk, ks = O.keyset(p, seed=4321)
        bits_a, bits_b, a, b = _inputs(p, s0, n_gates, seed=35)
        bk_t, ks_t = D.broadcast_keys(p, bk, ks, src=0, device=dev)   # device tensors over RCCL
        assert bk_t.is_cuda and ks_t.is_cuda
        ctx = tfhe_amd.Context(tfhe_amd.params_default(), bk_t, ks_t, device=0, keys_on_device=True)

        def evaluate(x, y):
            out = torch.empty_like(x)
            ctx.nand_device(x, y, out, stream=torch.cuda.current_stream(dev))
            return out

        a_t = torch.from_numpy(a.view(np.int32)).to(dev)
        b_t = torch.from_numpy(b.view(np.int32)).to(dev)
        full = D.nand_sharded(evaluate, a_t, b_t, n_gates, p.n + 1, src=0, device=dev)
        torch.cuda.synchronize(dev)
        ctx.close()
        want = O.Context(p, bk, ks).nand_batch(a, b)
        q.put(bool(np.array_equal(full.cpu().numpy().view(np.uint32), want)))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharding_over_rccl_single_rank():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_rccl_worker, args=(_free_port(), 9, q))
    pr.start()
    pr.join(240)
    assert pr.exitcode == 0
    assert q.get(timeout=5) is True
