"""Multi-GPU host logic: one process per GPU, gates sharded, keys replicated.

The path shards naturally: a gate reads only its two input ciphertexts and the shared
read-only evaluation key (reference src/hom_nand.hpp:74-81), so ranks never exchange data
while evaluating.  The only collective is the one-time broadcast of the bootstrapping and
key-switching keys from the rank that holds them (RCCL over xGMI when the tensors are on
GPUs, gloo for CPU tensors in tests).  Each rank then transforms the bootstrapping key to the
NTT domain locally, which is cheaper than broadcasting the twice-larger transformed key.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_gates, world):
    """Contiguous block split: rank r owns gates [bounds[r], bounds[r+1])."""
    return [n_gates * r // world for r in range(world + 1)]


def shard(n_gates, rank, world):
    b = shard_bounds(n_gates, world)
    return b[rank], b[rank + 1]


def broadcast_keys(params, bk, ks, src=0, device=None, group=None):
    """Returns (bk, ks) as flat int32 tensors on `device` on every rank.  On rank `src` the
    arguments are the keys (numpy uint32 arrays or tensors); elsewhere they are ignored."""
    import numpy as np

    rank = dist.get_rank(group)
    bk_words, ks_words = int(np.prod(params.bk_shape())), int(np.prod(params.ks_shape()))

    def as_tensor(x, words):
        if isinstance(x, torch.Tensor):
            t = x.reshape(-1)
            t = t.view(torch.int32) if t.dtype != torch.int32 else t
        else:
            t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.uint32).reshape(-1).view(np.int32))
        if t.numel() != words:
            raise ValueError("key size does not match the parameter set")
        return t.to(device) if device is not None else t

    if rank == src:
        bk_t, ks_t = as_tensor(bk, bk_words), as_tensor(ks, ks_words)
    else:
        bk_t = torch.empty(bk_words, dtype=torch.int32, device=device)
        ks_t = torch.empty(ks_words, dtype=torch.int32, device=device)
    dist.broadcast(bk_t, src=src, group=group)
    dist.broadcast(ks_t, src=src, group=group)
    return bk_t, ks_t


def scatter_gates(full, n_gates, n1, src=0, device=None, group=None):
    """Rank `src` holds `full` = [n_gates, n1] int32; every rank receives its contiguous shard.
    Point-to-point sends: shards may differ in size by one gate."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = shard_bounds(n_gates, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    if rank == src:
        reqs = []
        for r in range(world):
            if r != src and bounds[r + 1] > bounds[r]:
                reqs.append(dist.isend(full[bounds[r]:bounds[r + 1]].contiguous(), dst=r, group=group))
        local = full[lo:hi].clone()
        for q in reqs:
            q.wait()
        return local
    local = torch.empty((hi - lo, n1), dtype=torch.int32, device=device)
    if hi > lo:
        dist.recv(local, src=src, group=group)
    return local


def gather_gates(local, n_gates, n1, dst=0, group=None):
    """Inverse of scatter_gates: rank `dst` returns the [n_gates, n1] result, others None."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = shard_bounds(n_gates, world)
    if rank != dst:
        if local.shape[0]:
            dist.send(local.contiguous(), dst=dst, group=group)
        return None
    full = torch.empty((n_gates, n1), dtype=torch.int32, device=local.device)
    full[bounds[dst]:bounds[dst + 1]] = local
    for r in range(world):
        if r != dst and bounds[r + 1] > bounds[r]:
            dist.recv(full[bounds[r]:bounds[r + 1]], src=r, group=group)
    return full


def nand_sharded(evaluate, a_full, b_full, n_gates, n1, src=0, device=None, group=None):
    """out = NAND(a, b) over a batch held by rank `src`, evaluated shard-wise by every rank.
    `evaluate(a_shard, b_shard) -> out_shard` is the per-rank evaluator (on a GPU box,
    Context.nand_device on that rank's device).  Returns the full result on `src`."""
    a = scatter_gates(a_full, n_gates, n1, src, device, group)
    b = scatter_gates(b_full, n_gates, n1, src, device, group)
    out = evaluate(a, b) if a.shape[0] else a.clone()
    return gather_gates(out, n_gates, n1, src, group)
