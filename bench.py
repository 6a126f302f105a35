#!/usr/bin/env python3
"""Benchmark of the gate-bootstrapping hot path (HomNAND::nand, reference src/hom_nand.hpp:74-81).

  python bench.py --gpus N --steps K --warmup W [--total-gates T]

One process per GPU.  For N > 1 either torch.distributed.run starts the ranks (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment), or this script does: a plain `python bench.py --gpus N`
starts N fresh child processes of itself, before anything in the parent has touched a GPU, relays
rank 0's JSON line and exits non-zero if any rank did.

A step is one batch of independent NAND gates per GPU, inputs and outputs resident in HBM.  Two
ways to size it:
  weak   (default)          --gates-per-gpu gates on every GPU (8,192 = one GPU's shard of the
                            65,536-gate configuration when eight GPUs run)
  strong (--total-gates T)  T gates in all, GPU g of P evaluating the contiguous block
                            [g T / P, (g + 1) T / P) (T = 65536: the sharded configuration as written)
Keys are generated on rank 0's GPU, broadcast once over RCCL, and every rank builds its own device
context from the broadcast copy; there is no collective inside the timed region.  Rank 0 prints
one JSON line.
"""
import argparse
import json
import os
import pathlib
import socket
import subprocess
import sys
import tempfile
import time

ROOT = pathlib.Path(__file__).resolve().parent

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
# The N = 2048 stress set (n = 800, K = 1, L = 3): per step 6 forward + 2 inverse length-2048 NTTs
# (11,264 butterflies each) and 12 length-2048 multiply-accumulates; 800 steps.
N2048 = dict(n=800, N=2048, K=1, bg_bit=8, L=3, T=8, base_bit=2)
MODMUL_PER_GATE_2048 = 800 * (8 * 11264 + 12 * 2048)
F64_INSTR_PER_GATE_2048 = 800 * (8 * 11264 * 8 + 12 * 2048 * 7)
# FP64 vector peak of MI355X: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz (= 78.6 TFLOP/s as FMAs)
F64_LANE_INSTR_PEAK = 256 * 4 * 16 * 2.4e9
IO_BYTES_PER_GATE = 3 * 637 * 4
ALPHA_LV0, ALPHA_LV1 = 9.25119974676756e-5, 2.0 ** -25   # reference src/tlwe.hpp:13, src/trlwe.hpp:12
DOMINANT_KERNEL = "blind_rotate_512_kernel"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gates-per-gpu", type=int, default=8192,
                    help="weak scaling: batch per GPU per step (8192 = one GPU's shard of the 65,536-gate configuration)")
    ap.add_argument("--total-gates", type=int, default=0,
                    help="strong scaling: gates per step in all, split in contiguous blocks over the GPUs "
                         "(65536 = the sharded configuration as written)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the other configurations (for clean rocprofv3 kernel averages)")
    ap.add_argument("--cpu-sample-gates", type=int, default=64)
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher's environment: start the N ranks as child
    processes.  Nothing here imports torch or the library, so the parent never touches a GPU."""
    world = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    rank0_file = tempfile.TemporaryFile()   # rank 0's standard output (the JSON line)
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", str(pathlib.Path(__file__).resolve())] + sys.argv[1:],
                                      env=env, stdout=rank0_file if r == 0 else sys.stderr))
    rc = 0
    alive = set(range(world))
    try:
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0:
                    print(f"bench: rank {r} exited with status {code}", file=sys.stderr)
                    rc = rc or (code if 0 < code < 256 else 1)
            if rc:
                break
            time.sleep(0.05)
    finally:
        for r in alive:   # a rank failed: the others would wait for it in a collective forever
            procs[r].kill()
        for pr in procs:
            pr.wait()
    rank0_file.seek(0)
    sys.stdout.write(rank0_file.read().decode())
    sys.stdout.flush()
    return rc


def cpu_baseline(params, sample_gates, threads):
    """The reference itself (oracle/_ref/ref_nand, built from /root/reference in the development
    container) or, if that binary is absent, the oracle port, timed on this host's cores."""
    import numpy as np

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


def newest_pmc_summary(kernel):
    """The newest committed counter summary (profiles/r*_pmc_summary.json) that holds `kernel`:
    rocprofv3 --pmc passes cannot run inside this process, so the traffic figure is quoted."""
    for f in sorted((ROOT / "profiles").glob("r*_pmc_summary.json"), reverse=True):
        try:
            k = json.loads(f.read_text())["kernels"].get(kernel)
        except (ValueError, KeyError):
            continue
        if k and "hbm_bytes_per_launch" in k:
            return f, k
    return None, None


class Encryptor:
    """TLWE encryption / decryption of bits under a binary key, on the host
    (reference src/tlwe.hpp:55-75, src/types.hpp:25,34,46)."""

    def __init__(self, s, alpha, rs):
        import numpy as np

        self.np, self.s, self.alpha, self.rs = np, s.astype(np.uint64), alpha, rs

    def encrypt(self, bits):
        np, rs = self.np, self.rs
        ct = rs.randint(0, 2**32, size=(bits.size, self.s.size + 1), dtype=np.uint64).astype(np.uint32)
        noise = np.floor(np.mod(rs.normal(0.0, self.alpha, bits.size), 1.0) * 2.0**32).astype(np.uint64).astype(np.uint32)
        msg = np.where(bits == 1, np.uint32(1 << 29), np.uint32((1 << 32) - (1 << 29))).astype(np.uint32)
        ct[:, 0] = (ct[:, 1:].astype(np.uint64) @ self.s).astype(np.uint32) + msg + noise
        return ct

    def decrypt(self, ct):   # top bit of the phase clear = true
        np = self.np
        phase = ct[:, 0] - (ct[:, 1:].astype(np.uint64) @ self.s).astype(np.uint32)
        return (phase < np.uint32(1 << 31)).astype(np.int64)


def extra_n2048(tfhe_amd, torch, np, dev_index):
    """The N = 2048 stress set on this GPU: 2,048 gates under real keys, decrypt-checked."""
    p = tfhe_amd.Params(*[N2048[k] for k in ("n", "N", "K", "bg_bit", "L", "T", "base_bit")])
    rs = np.random.RandomState(5)
    s0 = rs.randint(0, 2, p.n).astype(np.uint8)
    sr = rs.randint(0, 2, (p.K, p.N)).astype(np.uint8)
    ctx = tfhe_amd.Context.from_secret(p, s0, sr, ALPHA_LV0, ALPHA_LV1, seed=5, device=dev_index)
    family = ctx.kernel_family()
    G = 2048
    enc = Encryptor(s0, ALPHA_LV0, rs)
    ba, bb = rs.randint(0, 2, G), rs.randint(0, 2, G)
    dev = torch.device("cuda", dev_index)
    a = torch.from_numpy(enc.encrypt(ba).view(np.int32)).to(dev)
    b = torch.from_numpy(enc.encrypt(bb).view(np.int32)).to(dev)
    out = torch.empty_like(a)
    ctx.nand_device(a, b, out); ctx.synchronize()
    ctx.enable_timing(True); ctx.timing()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.nand_device(a, b, out)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    br_ms, ks_ms, calls = ctx.timing()
    wrong = int(np.count_nonzero(enc.decrypt(out.cpu().numpy().view(np.uint32)) != 1 - (ba & bb)))
    one = torch.empty_like(a[:1])
    ctx.nand_device(a[:1], b[:1], one); ctx.synchronize()
    t0 = time.perf_counter()
    ctx.nand_device(a[:1], b[:1], one); ctx.synchronize()
    single_ms = (time.perf_counter() - t0) * 1e3
    ctx.close()
    k_s = br_ms / max(calls, 1) / 1e3
    peak = F64_LANE_INSTR_PEAK / (F64_INSTR_PER_GATE_2048 / MODMUL_PER_GATE_2048)
    achieved = MODMUL_PER_GATE_2048 * G / k_s
    return {"params": N2048, "gates": G, "ms": dt * 1e3, "gates_per_s": G / dt, "kernel_family": family,
            "single_gate_ms": single_ms,
            "roofline": {"bound": "valu_fp64", "kernel": "blind_rotate_wg_kernel", "kernel_ms": k_s * 1e3,
                         "key_switch_kernel_ms": ks_ms / max(calls, 1),
                         "achieved": achieved / 1e9, "peak": peak / 1e9, "unit": "Gmodmul/s", "frac": achieved / peak,
                         "modmul_per_gate": MODMUL_PER_GATE_2048,
                         "note": "peak = 39.3e12 FP64 lane-instructions/s / 7.79 FP64 instructions per modmul-equivalent"},
            "verified": {"gates_decrypted": G, "wrong": wrong}}, wrong


def extra_adders(tfhe_amd, torch, np, ctx, enc, dev):
    """The 32-bit adder, gates batched per dependency level: the NAND-only ripple-carry netlist
    (9 gates per bit), the XOR/MUX ripple-carry one, and the parallel-prefix one (7 levels of wide
    batches instead of 33 narrow ones); operands encrypted, sum decrypted and compared."""
    from tfhe_amd import circuit as CC

    res, wrong_total = {}, 0
    rs = np.random.RandomState(3)
    x, y = int(rs.randint(0, 2**32, dtype=np.uint64)), int(rs.randint(0, 2**32, dtype=np.uint64))
    bits = lambda v: np.array([(v >> i) & 1 for i in range(32)])  # noqa: E731
    backend = CC.DeviceBackend(ctx)
    n1 = ctx.params.n1
    for name, build in (("nand_only", CC.ripple_carry_adder), ("xor_mux", CC.mux_adder), ("prefix", CC.prefix_adder)):
        gates, n_wires, aw, bw, zero, sums, cout = build(32)
        plan = CC.Plan(gates, n_wires)
        wires = np.zeros((n_wires, n1), dtype=np.uint32)
        wires[aw], wires[bw] = enc.encrypt(bits(x)), enc.encrypt(bits(y))
        wires[zero] = CC.trivial_bit(0, n1)
        w0 = torch.from_numpy(wires.view(np.int32)).to(dev)
        dw = w0.clone()
        CC.evaluate(gates, dw, backend, plan=plan)
        torch.cuda.synchronize()
        dw = w0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        CC.evaluate(gates, dw, backend, plan=plan)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        got_bits = enc.decrypt(dw.cpu().numpy().view(np.uint32)[sums + [cout]])
        got = sum(int(b) << i for i, b in enumerate(got_bits))
        ok = got == x + y
        wrong_total += 0 if ok else 1
        n_rot = sum(2 if (len(g) > 3 and g[3] == "mux") else 1 for g in gates)
        res[name] = {"ms": dt * 1e3, "gates": len(gates), "blind_rotations": n_rot, "levels": len(plan.levels),
                     "ms_per_level": dt * 1e3 / len(plan.levels), "sum_correct": ok}
    return res, wrong_total


def main_rank(args):
    # Rank 0's standard output carries exactly one line, the JSON line: whatever a library prints there
    # while the ranks set up (gloo announces its connections on stdout, for one) goes to standard error.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    sys.path.insert(0, str(ROOT))
    import tfhe_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but the launcher started {world} ranks")
    # TFHE_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than
    # ranks (ranks share devices, keys are broadcast as host tensors); the real run uses RCCL.
    backend = os.environ.get("TFHE_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"bench: rank {rank} has no GPU (local rank {local_rank}, {n_dev} visible)")
    dev_index = local_rank if backend == "nccl" else local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # TFHE_BENCH_FORCE_DIST=1: a single rank still creates the process group and runs the key broadcast,
    # the gathers and the barriers through it (RCCL with one GPU), so that path can be tested on one GPU
    use_dist = world > 1 or os.environ.get("TFHE_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    bcast_dev = dev if backend == "nccl" else torch.device("cpu")

    p = tfhe_amd.params_default()
    n1 = p.n + 1
    strong = args.total_gates > 0
    if strong:
        lo, hi = args.total_gates * rank // world, args.total_gates * (rank + 1) // world
        G, total_per_step = hi - lo, args.total_gates
    else:
        G, total_per_step = args.gates_per_gpu, args.gates_per_gpu * world

    # A real key set (random binary secrets, evaluation keys generated on rank 0's GPU with the
    # reference's noise parameters, src/tlwe.hpp:13 and src/trlwe.hpp:12) and real encryptions, so
    # that every rank can check what it computed by decrypting it.  Synthetic: nothing is loaded.
    bk_words, ks_words = int(np.prod(p.bk_shape())), int(np.prod(p.ks_shape()))
    rs = np.random.RandomState(2026)
    s_lv0 = rs.randint(0, 2, p.n).astype(np.uint8)      # same secret on every rank (same seed)
    s_ring = rs.randint(0, 2, (p.K, p.N)).astype(np.uint8)
    if rank == 0:
        bk_h, ks_h = tfhe_amd.keygen(p, s_lv0, s_ring, ALPHA_LV0, ALPHA_LV1, seed=2026, device=dev_index)
        bk = torch.from_numpy(bk_h.reshape(-1).view(np.int32)).to(bcast_dev)
        ks = torch.from_numpy(ks_h.reshape(-1).view(np.int32)).to(bcast_dev)
        del bk_h, ks_h
    else:
        bk = torch.empty(bk_words, dtype=torch.int32, device=bcast_dev)
        ks = torch.empty(ks_words, dtype=torch.int32, device=bcast_dev)
    t_bcast = 0.0
    if use_dist:
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
    enc = Encryptor(s_lv0, ALPHA_LV0, rs_in)
    bits_a, bits_b = rs_in.randint(0, 2, G), rs_in.randint(0, 2, G)
    a = torch.from_numpy(enc.encrypt(bits_a).view(np.int32)).to(dev)
    b = torch.from_numpy(enc.encrypt(bits_b).view(np.int32)).to(dev)
    out = torch.empty_like(a)
    torch.cuda.synchronize()

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
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
    # per-rank times and kernel times to rank 0; the job's time is the slowest rank's
    mine = torch.tensor([elapsed_local, br_ms / max(calls, 1), float(G)], dtype=torch.float64, device=bcast_dev)
    if use_dist:
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
    else:
        every = [mine]
    every = [[float(v) for v in t.cpu()] for t in every]
    elapsed = max(e[0] for e in every)
    # every rank decrypts what its last timed step produced and checks it against the plaintext NAND
    wrong_local = int(np.count_nonzero(enc.decrypt(out.cpu().numpy().view(np.uint32)) != 1 - (bits_a & bits_b)))
    wrong = torch.tensor([wrong_local], dtype=torch.int64, device=bcast_dev)
    if use_dist:
        dist.all_reduce(wrong, op=dist.ReduceOp.SUM)
    wrong = int(wrong.item())

    if rank == 0:
        value = total_per_step * args.steps / elapsed
        br_avg_s = br_ms / max(calls, 1) / 1e3
        achieved = MODMUL_PER_GATE * G / br_avg_s
        peak = F64_LANE_INSTR_PEAK / (F64_INSTR_PER_GATE / MODMUL_PER_GATE)
        if strong:
            workload = (f"{total_per_step} independent NAND gates per step in all, default params (n=636, N=512, K=2, L=2, "
                        f"T=5), split in contiguous blocks over {world} GPU(s) (strong scaling; 65,536 = the sharded "
                        f"configuration as written)")
        else:
            workload = (f"{G} independent NAND gates per GPU per step, default params (n=636, N=512, K=2, L=2, T=5): "
                        f"one GPU's shard of the 65,536-gate configuration (weak scaling; eight GPUs run exactly it)")
        line = {
            "metric": "bootstrapped NAND gates/sec (whole node)",
            "value": value, "unit": "gates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u32 torus; NTT over a 51-bit prime evaluated in f64",
            "data": "synthetic (keys generated on the GPU from random secrets, fresh encryptions of random bits)",
            "config": {"workload": workload, "gates_per_gpu": G, "gates_per_step": total_per_step, "params": "default",
                       "parallelism": f"gate-sharded x{world}, keys replicated"},
            "roofline": {"bound": "valu_fp64", "kernel": DOMINANT_KERNEL,
                         "achieved": achieved / 1e9, "peak": peak / 1e9, "unit": "Gmodmul/s",
                         "frac": achieved / peak, "traffic": None,
                         "kernel_ms": br_avg_s * 1e3, "modmul_per_launch": MODMUL_PER_GATE * G,
                         "key_switch_kernel_ms": ks_ms / max(calls, 1),
                         "io_GBps": IO_BYTES_PER_GATE * G / (elapsed / args.steps) / 1e9,
                         "note": "peak = 39.3e12 FP64 lane-instructions/s / 7.69 FP64 instructions per modmul-equivalent"},
            "distributed": {"backend": backend if use_dist else "none",
                            "world_size": dist.get_world_size() if use_dist else 1,
                            "launcher": "self" if os.environ.get("TFHE_BENCH_SELF_LAUNCHED") else
                                        ("external" if world > 1 else "none"),
                            "key_broadcast_s": t_bcast,
                            "per_rank": [{"gates": int(e[2]), "ms_per_step": e[0] / args.steps * 1e3,
                                          "blind_rotate_kernel_ms": e[1]} for e in every]},
            "key_broadcast_s": t_bcast,
            "verified": {"gates_decrypted": total_per_step, "wrong": wrong,
                         "how": "every rank decrypts the outputs of its last timed step and compares with NAND of the plaintext bits"},
        }
        # HBM traffic of the dominant kernel comes from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE)
        pmc_file, k = newest_pmc_summary(DOMINANT_KERNEL)
        if k:
            line["roofline"]["traffic"] = k["hbm_bytes_per_launch"]
            line["roofline"]["traffic_source"] = (f"profiles/{pmc_file.name}: (2 x FETCH_SIZE + WRITE_SIZE) per launch of "
                                                  "8192 gates; the NTT-domain key (46.9 MB) is re-streamed from the "
                                                  "Infinity Cache once per XCD and wave generation, L2 hit rate 99 %")
            line["roofline"]["valu_pipe_busy"] = 2 * k.get("valu_active_share_of_wave_cycles", 0.0)
            # the same kernel against the HBM roofline, to show which of the two binds
            hbm_gbps = k["hbm_bytes_per_launch"] * (G / 8192.0) / br_avg_s / 1e9
            line["roofline"]["hbm"] = {"achieved": hbm_gbps, "peak": 8000.0, "unit": "GB/s", "frac": hbm_gbps / 8000.0}
        # the other configurations, measured beside it on one GPU
        if world == 1 and not args.no_extras:
            extra = {}
            for g_small in (1024, 1):
                g_small = min(g_small, G)
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
            if G != g_big:
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
            gh = min(G, 8192)
            ha, hb = a[:gh].cpu().numpy().view(np.uint32), b[:gh].cpu().numpy().view(np.uint32)
            ctx.nand(ha[:64], hb[:64])
            t0 = time.perf_counter()
            ctx.nand(ha, hb)
            dt = time.perf_counter() - t0
            line["pcie_inclusive"] = {"gates_per_s": gh / dt, "ms": dt * 1e3, "gates": gh,
                                      "note": "tfhe_amd_nand_batch on host arrays: H2D of both operands, kernels, D2H"}
            del ha, hb
            ctx_ref.nand_device(a[:gh], b[:gh], out[:gh]); ctx_ref.synchronize()
            t0 = time.perf_counter()
            ctx_ref.nand_device(a[:gh], b[:gh], out[:gh])
            ctx_ref.synchronize()
            dt = time.perf_counter() - t0
            ctx_ref.close()
            line["reference_bit_identical_mode"] = {
                "gates_per_s": gh / dt, "ms": dt * 1e3, "gates": gh,
                "note": "TFHE_AMD_CTX_REFERENCE_FFT: ciphertexts bit-identical to the reference built with g++ -O2 "
                        "(tests/test_reference_golden_nand512.py); the headline value is the exact-product path"}
            # the level-batched 32-bit adders and the N = 2048 stress set
            line["adder_32bit"], w4 = extra_adders(tfhe_amd, torch, np, ctx, enc, dev)
            line["n2048_set"], w5 = extra_n2048(tfhe_amd, torch, np, dev_index)
            wrong += w4 + w5
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            line["cpu_baseline"] = cpu_baseline(p, args.cpu_sample_gates, threads)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    ctx.close()
    if use_dist:
        dist.destroy_process_group()
    if wrong:
        raise SystemExit(f"bench: {wrong} results decrypted to the wrong value")


if __name__ == "__main__":
    _args = parse_args()
    if _args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        os.environ["TFHE_BENCH_SELF_LAUNCHED"] = "1"
        sys.exit(self_launch(_args))
    main_rank(_args)
