"""world_size-2 test of the gate-sharding host logic on CPU (gloo).  The per-rank evaluator is
the oracle here (no GPU in this suite); on a GPU box the same functions drive Context.nand_device."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_py as O
from tfhe_amd import distributed as D


def test_shard_bounds():
    assert D.shard_bounds(65536, 8) == [8192 * r for r in range(9)]
    assert D.shard_bounds(10, 4) == [0, 2, 5, 7, 10]
    assert D.shard_bounds(0, 3) == [0, 0, 0, 0]
    assert D.shard_bounds(2, 4) == [0, 0, 1, 1, 2]
    for n, w in [(1, 1), (7, 2), (1024, 8), (65537, 8)]:
        b = D.shard_bounds(n, w)
        assert b[0] == 0 and b[-1] == n and all(0 <= y - x <= -(-n // w) for x, y in zip(b, b[1:]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_gates, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = O.params_from(10, 16, 2, 8, 2, 5, 2)
        if rank == 0:
            s0, sr, bk, ks = O.keyset(p, seed=77)
            r = O.rng(3)
            bits_a = np.arange(n_gates) % 2
            bits_b = (np.arange(n_gates) // 2) % 2
            a = O.encrypt_bits(bits_a, s0, p, r)
            b = O.encrypt_bits(bits_b, s0, p, r)
            a_t, b_t = torch.from_numpy(a.view(np.int32)), torch.from_numpy(b.view(np.int32))
        else:
            bk = ks = a_t = b_t = None
        bk_t, ks_t = D.broadcast_keys(p, bk, ks, src=0)
        ctx = O.Context(p, bk_t.numpy().view(np.uint32), ks_t.numpy().view(np.uint32))

        def evaluate(x, y):
            out = ctx.nand_batch(x.numpy().view(np.uint32), y.numpy().view(np.uint32), threads=1)
            return torch.from_numpy(out.view(np.int32))

        full = D.nand_sharded(evaluate, a_t, b_t, n_gates, p.n + 1, src=0)
        if rank == 0:
            want = O.Context(p, bk, ks).nand_batch(a, b, threads=1)
            ok = np.array_equal(full.numpy().view(np.uint32), want)
            ok = ok and np.array_equal(O.decrypt_bits(want, s0, p), 1 - (bits_a & bits_b))
            q.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_gates", [9, 1, 0])
def test_sharded_nand_gloo_world2(n_gates):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_gates, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    assert q.get(timeout=5) is True
