#!/usr/bin/env python3
"""Benchmark of the MI355X prover hot path ("Measurement").

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): 2^20-point Goldilocks LDE, blowup 8 -- B = 135 columns
("wires"-shaped batch, access_set.rs:73) of n = 2^17 coefficients each are extended to N = 2^20
evaluations on the coset 7<omega_N>, written in commitment (bit-reversed) order.  One step = one
such batch per GPU, inputs and outputs resident in HBM.  value = algorithmic GB/s of the whole job:
8*B*(n+N) bytes per step per GPU (one read of the coefficients + one write of the evaluations).

Multi-GPU: the batch shards by columns/batches with no data-path collective (weak scaling); after
the timed steps every rank contributes one Poseidon digest of its result and rank 0 folds the
gathered digests into an aggregation root with the HIP Merkle kernel (RCCL all_gather of 32 B/rank).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N, RATE_BITS, BATCH = 17, 3, 135
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(seconds=4.0):
    """The CPU restatement (oracle, OpenMP over columns like plonky2's rayon) on the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle, rand_field
    orc = Oracle()
    threads = orc.L.orc_num_threads()
    rng = np.random.default_rng(0x355)
    n, N = 1 << LOG_N, 1 << (LOG_N + RATE_BITS)
    # bounded sample: as many columns as threads (at most the full batch), repeated for ~`seconds`
    cols = max(1, min(BATCH, threads))
    coeffs = rand_field(rng, (cols, n))
    out = np.empty((cols, N), dtype=np.uint64)
    u64p = C.POINTER(C.c_uint64)
    reps, t0 = 0, time.perf_counter()
    while True:
        orc.L.orc_lde(coeffs.ctypes.data_as(u64p), C.c_uint32(LOG_N), C.c_uint32(RATE_BITS), C.c_uint64(7),
                      C.c_uint32(cols), out.ctypes.data_as(u64p))
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or reps >= 50:
            break
    gbs = 8.0 * cols * (n + N) * reps / dt / 1e9
    return {"value": round(gbs, 3), "unit": "GB/s", "cores": int(threads), "kind": "port",
            "sample": "%d columns of the same 2^%d->2^%d LDE, %d repetitions, %.1f s wall (C restatement of "
                      "plonky2's lde+coset_fft, OpenMP over columns; not the Rust binary)" % (cols, LOG_N, LOG_N + RATE_BITS, reps, dt)}


class SemaphoreProvers:
    """K concurrent prover contexts (one HIP stream each) on one GPU proving depth-20 Semaphore signals
    (make_signal, access_set.rs:61-104): the unit of BASELINE's proofs/s metric, without the recursive wrap
    (the recursive verifier circuit is not built yet -- DESIGN.md section 7)."""

    def __init__(self, gl, device, threads, log_members=20, seed=0x357):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_lib import rand_field  # only the seeded RNG helper, no oracle arithmetic
        sem = importlib.import_module("stark-verifier_amd.semaphore")
        rng = np.random.default_rng(seed)
        ctx0 = gl.Context(device)
        self.sks = rand_field(rng, (1 << log_members, 4))
        keys = ctx0.hash_no_pad(np.concatenate([self.sks, np.zeros_like(self.sks)], axis=1))
        self.topic = rand_field(rng, 4)
        self.sets = []
        for t in range(threads):
            a = sem.AccessSet(gl.Context(device), keys)
            a.build(np.random.default_rng(1))
            a.make_signal_fast(self.sks[t], self.topic, t, t)   # warm-up
            self.sets.append(a)
        self.root = self.sets[0].tree.cap[0].copy()

    def prove_batch(self, first, count):
        """proves members first..first+count-1, returns their (nullifier | topic) leaves [count][8]"""
        import threading
        k = len(self.sets)
        leaves = np.zeros((count, 8), dtype=np.uint64)

        def worker(t):
            for j in range(t, count, k):
                i = first + j
                sig, _ = self.sets[t].make_signal_fast(self.sks[i], self.topic, i, 0x358 + i, flat_only=True)
                leaves[j, :4] = sig.nullifier[0]
                leaves[j, 4:] = self.topic
        ths = [threading.Thread(target=worker, args=(t,)) for t in range(k)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        return leaves


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["lde", "semaphore"], default="lde",
                    help="lde = BASELINE configs[1] (default); semaphore = depth-20 proofs, sharded over the GPUs (configs[4] shape)")
    ap.add_argument("--proofs-per-step", type=int, default=32, help="semaphore workload: proofs per GPU per step")
    ap.add_argument("--threads", type=int, default=12, help="semaphore workload: concurrent prover contexts per GPU")
    args = ap.parse_args()
    if args.workload == "semaphore":
        return main_semaphore(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    gl = importlib.import_module("stark-verifier_amd")
    ctx = gl.Context(local_rank)
    lib = ctx.lib
    n, N = 1 << LOG_N, 1 << (LOG_N + RATE_BITS)

    # synthetic coefficients, uniform in [0, p) up to the negligible rejection tail (seeded per rank)
    g = torch.Generator(device=dev)
    g.manual_seed(0x355 + rank)
    coeffs = torch.randint(0, (1 << 63) - 1, (BATCH, n), dtype=torch.int64, device=dev, generator=g)
    out = torch.empty((BATCH, N), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def step():
        ctx.check(lib.gl355_lde_bitrev(ctx.h, C.c_void_p(coeffs.data_ptr()), LOG_N, RATE_BITS, 7, BATCH,
                                       C.c_void_p(out.data_ptr())))

    for _ in range(args.warmup):
        step()
    ctx.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides ----------------------
    ctx.profile_enable(True)
    ctx.profile_read()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    # aggregation root over one digest per rank (the only exchange step of the sharded job)
    digest = torch.empty(4, dtype=torch.int64, device=dev)
    ctx.check(lib.gl355_hash_no_pad(ctx.h, C.c_void_p(out.data_ptr()), 1, 8, C.c_void_p(digest.data_ptr())))
    ctx.sync()
    root = None
    if dist is not None:
        gathered = [torch.empty_like(digest) for _ in range(world)]
        dist.all_gather(gathered, digest)
        if rank == 0:
            leaves = torch.stack(gathered).contiguous()
            pad = 1
            while pad < world:
                pad *= 2
            if pad != world:
                leaves = torch.cat([leaves, torch.zeros((pad - world, 4), dtype=torch.int64, device=dev)])
            capbuf = torch.empty(4, dtype=torch.int64, device=dev)
            digs = torch.empty((max(1, 2 * (pad - 1)), 4), dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            ctx.check(lib.gl355_merkle_build(ctx.h, C.c_void_p(leaves.data_ptr()), pad, 4, 0, C.c_void_p(digs.data_ptr()),
                                             C.c_void_p(capbuf.data_ptr())))
            ctx.sync()
            root = [int(x) & ((1 << 64) - 1) for x in capbuf.cpu().tolist()]
    barrier()
    t1 = time.perf_counter()
    prof = ctx.profile_read()
    ctx.profile_enable(False)

    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        alg_bytes_step = 8.0 * BATCH * (n + N)
        value = alg_bytes_step * args.steps * world / elapsed / 1e9
        # dominant kernel = the kernel group with the largest HIP-event time in the timed region
        dom_name, (dom_cnt, dom_ms) = max(prof.items(), key=lambda kv: kv[1][1]) if prof else ("none", (1, 0.0))
        total_kernel_ms = sum(v[1] for v in prof.values())
        # algorithmic bytes of one launch of each pass (DESIGN.md "NTT"): pass 1 reads the n coefficients
        # once and owns the coset expansion; pass 2 turns them into the N evaluations.  A launch of either
        # pass is charged the FULL algorithmic traffic of the LDE it belongs to divided between the two
        # passes in proportion to what each must move at minimum: pass1 = 8*B*n, pass2 = 8*B*N.
        alg_by_kernel = {"ntt_cols_pass1": 8.0 * BATCH * n, "ntt_rows_pass2": 8.0 * BATCH * N,
                         "ntt_rows_single_pass": alg_bytes_step}
        per_launch_ms = dom_ms / max(1, dom_cnt)
        # roofline of the whole LDE (both passes are needed to produce one unit of output): algorithmic
        # bytes of one LDE over the summed average launch durations of its kernels
        lde_ms = sum(v[1] / max(1, v[0]) for k, v in prof.items() if k.startswith("ntt_"))
        achieved = alg_bytes_step / (lde_ms * 1e-3) / 1e9 if lde_ms > 0 else 0.0
        line = {
            "metric": "NTT HBM GB/s (2^20-point Goldilocks LDE, blowup 8, bit-exact)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
            "config": {"workload": "lde n=2^17 -> N=2^20 (rate_bits 3, coset 7), batch 135 columns per GPU, "
                                   "bit-reversed (commitment) output order, operands resident in HBM",
                       "algorithmic_bytes_per_step_per_gpu": alg_bytes_step, "parallelism": "independent batches per GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "lde = ntt_cols_kernel<5> (pass 1) + ntt_rows_kernel<12,12> (pass 2)",
                         "dominant_kernel": dom_name,
                         "dominant_avg_launch_ms": round(per_launch_ms, 4),
                         "dominant_alg_GBps": round(alg_by_kernel.get(dom_name, alg_bytes_step) / (per_launch_ms * 1e-3) / 1e9, 2)
                         if per_launch_ms > 0 else None,
                         "kernels_ms_per_launch": {k: round(v[1] / max(1, v[0]), 4) for k, v in prof.items()},
                         "kernel_time_fraction_of_wall": round(total_kernel_ms * 1e-3 / elapsed, 3)},
        }
        if root is not None:
            line["aggregation_root"] = ["%016x" % x for x in root]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            # secondary figure (not the timed region): end-to-end Semaphore proofs/s on this GPU
            try:
                pr = SemaphoreProvers(gl, local_rank, 12)
                pr.prove_batch(100, 16)
                t_a = time.perf_counter()
                pr.prove_batch(200, 128)
                dt = time.perf_counter() - t_a
                line["semaphore_proofs"] = {"value": round(128 / dt, 1), "unit": "proofs/s", "proofs": 128, "contexts": 12,
                                            "what": "make_signal (depth-20 membership + nullifier, n = 2^13, blowup 8, 28 queries, "
                                                    "16 PoW bits, zk) incl. witness generation, every proof bit-checked stage-wise in "
                                                    "tests; no recursive wrap; reference README: ~1.05 proofs/s on an M1"}
            except Exception as exc:  # the headline line must still be printed
                line["semaphore_proofs"] = {"error": repr(exc)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_semaphore(args):
    """proofs sharded over ranks (recursion.rs:300-308 maps to one block of members per GPU), one RCCL
    all_gather of the (nullifier | topic) leaves, aggregation root on rank 0."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    gl = importlib.import_module("stark-verifier_amd")
    par = importlib.import_module("stark-verifier_amd.parallel")
    pr = SemaphoreProvers(gl, local_rank, args.threads)
    per = args.proofs_per_step
    total = per * world
    lo, hi = par.shard_range(total, rank, world)
    for w in range(args.warmup):
        pr.prove_batch(1000 + lo, hi - lo)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    root = None
    for step in range(args.steps):
        leaves = pr.prove_batch(2000 + step * total + lo, hi - lo)
        lt = torch.from_numpy(leaves.view(np.int64)).to(dev)
        allv = par.gather_leaves(lt, dist)
        if rank == 0:
            root = par.aggregation_root(pr.sets[0].ctx, allv.cpu().numpy().view(np.uint64))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        line = {"metric": "plonky2 proofs/sec (Semaphore d=20, no recursive wrap)", "value": round(total * args.steps / elapsed, 2),
                "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
                "config": {"workload": "make_signal: group of 2^20 members, %d proofs per GPU per step, %d prover contexts per GPU, "
                                       "all_gather of (nullifier|topic) leaves + Poseidon aggregation root per step" % (per, args.threads)},
                "aggregation_root": ["%016x" % int(x) for x in root[0]]}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
