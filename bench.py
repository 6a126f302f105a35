#!/usr/bin/env python3
"""Benchmark of the gate-bootstrapping hot path (HomNAND::nand, reference src/hom_nand.hpp:74-81).

  python bench.py --gpus N --steps K --warmup W

One process per GPU (launched by torch.distributed.run for N > 1).  A step is one batch of
independent NAND gates per GPU, inputs and outputs resident in HBM.  Keys are generated on
rank 0, broadcast once over RCCL, and every rank builds its own device context from the
broadcast copy; there is no collective inside the timed region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import pathlib
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import tfhe_amd  # noqa: E402

# Algorithmic work per gate at the default parameter set: per blind-rotate step 6 forward and
# 3 inverse length-512 NTTs (2,304 butterflies each) plus 18 length-512 pointwise
# multiply-accumulates; 636 steps.
MODMUL_PER_GATE = 636 * (9 * 2304 + 18 * 512)
BUTTERFLIES_PER_GATE = 636 * 9 * 2304
MACS_PER_GATE = 636 * 18 * 512
# FP64 vector instructions per unit in the canonical formulation the kernel uses:
# a butterfly is one 6-instruction modular multiply plus an add and a subtract, a
# multiply-accumulate is the 6-instruction multiply plus one add.
F64_INSTR_PER_GATE = BUTTERFLIES_PER_GATE * 8 + MACS_PER_GATE * 7
# FP64 vector peak of MI355X: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz (= 78.6 TFLOP/s as FMAs)
F64_LANE_INSTR_PEAK = 256 * 4 * 16 * 2.4e9
IO_BYTES_PER_GATE = 3 * 637 * 4
# counter summary of the dominant kernel quoted in roofline.traffic (tools/pmc_passes.sh + tools/pmc_summarize.py)
PMC_SUMMARY = "r02_pmc_summary.json"


def cpu_baseline(params, sample_gates, threads):
    """The reference itself (oracle/_ref/ref_nand, built from /root/reference in the development
    container) or, if that binary is absent, the oracle port, timed on this host's cores."""
    rs = np.random.RandomState(17)
    n1 = params.n + 1
    bk = rs.randint(0, 2**32, size=params.bk_shape(), dtype=np.uint64).astype(np.uint32)
    ks = rs.randint(0, 2**32, size=params.ks_shape(), dtype=np.uint64).astype(np.uint32)
    ab = rs.randint(0, 2**32, size=(2, sample_gates, n1), dtype=np.uint64).astype(np.uint32)
    ref_bin = ROOT / "oracle" / "_ref" / "ref_nand"
    res = {}
    if ref_bin.exists():
        with tempfile.TemporaryDirectory() as tmp:
            t = pathlib.Path(tmp)
            bk.tofile(t / "bk.bin"); ks.tofile(t / "ks.bin"); ab.tofile(t / "in.bin")
            out = subprocess.run([str(ref_bin), str(t / "bk.bin"), str(t / "ks.bin"), str(t / "in.bin"),
                                  str(t / "out.bin"), str(threads)], check=True, capture_output=True, text=True)
            r = json.loads(out.stdout.strip().splitlines()[-1])
        res = {"value": r["gates_per_s"], "unit": "gates/s", "cores": threads, "kind": "reference",
               "sample": f"{sample_gates} independent NAND gates, default params, {threads} threads sharing one "
                         f"EvaluateKey, reference built with g++ -std=gnu++17 -O2; {r['seconds']:.2f} s wall"}
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle_py as O  # checker / baseline only

    octx = O.Context(O.params_default(), bk, ks)
    t0 = time.time()
    octx.nand_batch(ab[0], ab[1], threads=threads)
    dt = time.time() - t0
    port = {"value": sample_gates / dt, "unit": "gates/s", "cores": threads, "kind": "port",
            "sample": f"{sample_gates} gates, exact-integer oracle (NTT multiplier), {threads} OpenMP threads; {dt:.2f} s wall"}
    if not res:
        return port
    res["port"] = port
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gates-per-gpu", type=int, default=8192,
                    help="batch per GPU per step (8192 = one GPU's shard of the 65,536-gate configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the 1,024-gate and single-gate timings (for clean rocprofv3 kernel averages)")
    ap.add_argument("--cpu-sample-gates", type=int, default=64)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # TFHE_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than
    # ranks (ranks share devices, keys are broadcast as host tensors); the real run uses RCCL.
    backend = os.environ.get("TFHE_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    bcast_dev = dev if backend == "nccl" else torch.device("cpu")

    p = tfhe_amd.params_default()
    n1, G = p.n + 1, args.gates_per_gpu

    # A real key set (random binary secrets, evaluation keys generated on rank 0's GPU with the
    # reference's noise parameters, src/tlwe.hpp:13 and src/trlwe.hpp:12) and real encryptions, so
    # that every rank can check what it computed by decrypting it.  Synthetic: nothing is loaded.
    alpha_lv0, alpha_lv1 = 9.25119974676756e-5, 2.0 ** -25
    bk_words, ks_words = int(np.prod(p.bk_shape())), int(np.prod(p.ks_shape()))
    rs = np.random.RandomState(2026)
    s_lv0 = rs.randint(0, 2, p.n).astype(np.uint8)      # same secret on every rank (same seed)
    s_ring = rs.randint(0, 2, (p.K, p.N)).astype(np.uint8)
    if rank == 0:
        bk_h, ks_h = tfhe_amd.keygen(p, s_lv0, s_ring, alpha_lv0, alpha_lv1, seed=2026, device=dev_index)
        bk = torch.from_numpy(bk_h.reshape(-1).view(np.int32)).to(bcast_dev)
        ks = torch.from_numpy(ks_h.reshape(-1).view(np.int32)).to(bcast_dev)
        del bk_h, ks_h
    else:
        bk = torch.empty(bk_words, dtype=torch.int32, device=bcast_dev)
        ks = torch.empty(ks_words, dtype=torch.int32, device=bcast_dev)
    t_bcast = 0.0
    if world > 1:
        # the first collective creates the communicator: keep it out of the broadcast time
        warm = torch.zeros(1, dtype=torch.int32, device=bcast_dev)
        dist.broadcast(warm, src=0)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        dist.broadcast(bk, src=0)   # RCCL over xGMI, once per key
        dist.broadcast(ks, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
    bk, ks = bk.to(dev), ks.to(dev)
    torch.cuda.synchronize()
    ctx = tfhe_amd.Context(p, bk, ks, device=dev_index, keys_on_device=True)
    if ctx.kernel_family() != "ntt512":
        raise SystemExit(f"bench: expected the NTT kernels of the default parameter set, got {ctx.kernel_family()}")
    # verification mode (products through a replay of the reference's FFT: output bit-identical to the
    # compiled reference), timed beside the default path on one GPU
    ctx_ref = None
    if world == 1 and not args.no_extras:
        ctx_ref = tfhe_amd.Context(p, bk, ks, device=dev_index, keys_on_device=True, reference_fft=True)
    del bk, ks

    # this rank's gates: fresh encryptions of random bits, b = <a, s> + m + e (reference src/tlwe.hpp:55-61)
    rs_in = np.random.RandomState(1000 + rank)

    def encrypt(bits):
        ct = rs_in.randint(0, 2**32, size=(bits.size, n1), dtype=np.uint64).astype(np.uint32)
        noise = np.floor(np.mod(rs_in.normal(0.0, alpha_lv0, bits.size), 1.0) * 2.0**32).astype(np.uint64).astype(np.uint32)
        msg = np.where(bits == 1, np.uint32(1 << 29), np.uint32((1 << 32) - (1 << 29))).astype(np.uint32)
        ct[:, 0] = (ct[:, 1:].astype(np.uint64) @ s_lv0.astype(np.uint64)).astype(np.uint32) + msg + noise
        return ct

    def decrypt(ct):   # top bit of the phase clear = true (reference src/tlwe.hpp:67-75, src/types.hpp:46)
        phase = ct[:, 0] - (ct[:, 1:].astype(np.uint64) @ s_lv0.astype(np.uint64)).astype(np.uint32)
        return (phase < np.uint32(1 << 31)).astype(np.int64)

    bits_a, bits_b = rs_in.randint(0, 2, G), rs_in.randint(0, 2, G)
    a = torch.from_numpy(encrypt(bits_a).view(np.int32)).to(dev)
    b = torch.from_numpy(encrypt(bits_b).view(np.int32)).to(dev)
    out = torch.empty_like(a)
    torch.cuda.synchronize()

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        ctx.nand_device(a, b, out)
    ctx.enable_timing(True)
    barrier()
    ctx.timing()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.nand_device(a, b, out)
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    barrier()
    br_ms, ks_ms, calls = ctx.timing()
    ctx.enable_timing(False)
    el = torch.tensor([elapsed_local], dtype=torch.float64, device=bcast_dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    # every rank decrypts what its last timed step produced and checks it against the plaintext NAND
    wrong_local = int(np.count_nonzero(decrypt(out.cpu().numpy().view(np.uint32)) != 1 - (bits_a & bits_b)))
    wrong = torch.tensor([wrong_local], dtype=torch.int64, device=bcast_dev)
    if world > 1:
        dist.all_reduce(wrong, op=dist.ReduceOp.SUM)
    wrong = int(wrong.item())

    if rank == 0:
        total_gates = G * world * args.steps
        value = total_gates / elapsed
        br_avg_s = br_ms / max(calls, 1) / 1e3
        achieved = MODMUL_PER_GATE * G / br_avg_s
        peak = F64_LANE_INSTR_PEAK / (F64_INSTR_PER_GATE / MODMUL_PER_GATE)
        line = {
            "metric": "bootstrapped NAND gates/sec (whole node)",
            "value": value, "unit": "gates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 torus; NTT over a 51-bit prime evaluated in f64", "data": "synthetic (keys generated on the GPU from random secrets, fresh encryptions of random bits)",
            "config": {"workload": f"{G} independent NAND gates per GPU per step, default params (n=636, N=512, K=2, "
                                   f"L=2, T=5): one GPU's shard of the 65,536-gate configuration",
                       "gates_per_gpu": G, "params": "default", "parallelism": f"gate-sharded x{world}, keys replicated"},
            "roofline": {"bound": "valu_fp64", "kernel": "blind_rotate_512_kernel",
                         "achieved": achieved / 1e9, "peak": peak / 1e9, "unit": "Gmodmul/s",
                         "frac": achieved / peak, "traffic": None,
                         "kernel_ms": br_avg_s * 1e3, "modmul_per_launch": MODMUL_PER_GATE * G,
                         "key_switch_kernel_ms": ks_ms / max(calls, 1),
                         "io_GBps": IO_BYTES_PER_GATE * G / (elapsed / args.steps) / 1e9,
                         "note": "peak = 39.3e12 FP64 lane-instructions/s / 7.69 FP64 instructions per modmul-equivalent"},
            "key_broadcast_s": t_bcast,
            "verified": {"gates_decrypted": G * world, "wrong": wrong,
                         "how": "every rank decrypts the outputs of its last timed step and compares with NAND of the plaintext bits"},
        }
        # HBM traffic of the dominant kernel comes from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE),
        # which cannot run inside this process: the committed summary of the latest passes is quoted
        pmc = [ROOT / "profiles" / PMC_SUMMARY] if (ROOT / "profiles" / PMC_SUMMARY).exists() else []
        if pmc:
            k = json.loads(pmc[-1].read_text())["kernels"].get("blind_rotate_512_kernel", {})
            if "hbm_bytes_per_launch" in k:
                line["roofline"]["traffic"] = k["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = (f"profiles/{pmc[-1].name}: (2 x FETCH_SIZE + WRITE_SIZE) per launch of "
                                                      "8192 gates; the NTT-domain key (46.9 MB) is re-streamed from the "
                                                      "Infinity Cache once per XCD and wave generation, L2 hit rate 99 %")
                line["roofline"]["valu_pipe_busy"] = 2 * k.get("valu_active_share_of_wave_cycles", 0.0)
                # the same kernel against the HBM roofline, to show which of the two binds
                hbm_gbps = k["hbm_bytes_per_launch"] * (G / 8192.0) / br_avg_s / 1e9
                line["roofline"]["hbm"] = {"achieved": hbm_gbps, "peak": 8000.0, "unit": "GB/s", "frac": hbm_gbps / 8000.0}
        # the 1,024-gate single-GPU configuration and single-gate latency, measured beside it
        if world == 1 and not args.no_extras:
            extra = {}
            for g_small in (1024, 1):
                ctx.nand_device(a[:g_small], b[:g_small], out[:g_small]); ctx.synchronize()
                t0 = time.perf_counter()
                reps = 3
                for _ in range(reps):
                    ctx.nand_device(a[:g_small], b[:g_small], out[:g_small])
                ctx.synchronize()
                dt = (time.perf_counter() - t0) / reps
                extra[f"batch_{g_small}"] = {"ms": dt * 1e3, "gates_per_s": g_small / dt}
            # the whole 65,536-gate configuration on this one GPU (8 x the timed batch)
            g_big = 65536
            a8, b8 = a.repeat(g_big // G + 1, 1)[:g_big].contiguous(), b.repeat(g_big // G + 1, 1)[:g_big].contiguous()
            out8 = torch.empty_like(a8)
            ctx.nand_device(a8, b8, out8); ctx.synchronize()
            t0 = time.perf_counter()
            ctx.nand_device(a8, b8, out8)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            extra[f"batch_{g_big}"] = {"ms": dt * 1e3, "gates_per_s": g_big / dt}
            del a8, b8, out8
            line["other_batches"] = extra
            line["single_gate_latency_ms"] = extra["batch_1"]["ms"]
            # host-pointer entry point: PCIe transfers of inputs and outputs included (pageable memory)
            ha, hb = a.cpu().numpy().view(np.uint32), b.cpu().numpy().view(np.uint32)
            ctx.nand(ha[:64], hb[:64])
            t0 = time.perf_counter()
            ctx.nand(ha, hb)
            dt = time.perf_counter() - t0
            line["pcie_inclusive"] = {"gates_per_s": G / dt, "ms": dt * 1e3, "gates": G,
                                      "note": "tfhe_amd_nand_batch on host arrays: H2D of both operands, kernels, D2H"}
            del ha, hb
            ctx_ref.nand_device(a, b, out); ctx_ref.synchronize()
            t0 = time.perf_counter()
            ctx_ref.nand_device(a, b, out)
            ctx_ref.synchronize()
            dt = time.perf_counter() - t0
            ctx_ref.close()
            line["reference_bit_identical_mode"] = {
                "gates_per_s": G / dt, "ms": dt * 1e3, "gates": G,
                "note": "TFHE_AMD_CTX_REFERENCE_FFT: ciphertexts bit-identical to the reference built with g++ -O2 "
                        "(tests/test_reference_golden_nand512.py); the headline value is the exact-product path"}
        if world == 1:
            if not args.no_cpu_baseline:
                threads = min(16, os.cpu_count() or 1)
                line["cpu_baseline"] = cpu_baseline(p, args.cpu_sample_gates, threads)
        print(json.dumps(line), flush=True)
    ctx.close()
    if wrong:
        raise SystemExit(f"bench: {wrong} gates decrypted to the wrong bit")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
