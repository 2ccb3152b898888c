#!/usr/bin/env python3
"""Aggregation tree across ranks: torchrun --nproc-per-node G tools/aggregate_distributed.py [signals per rank] [contexts].
Each rank proves and aggregates its own signals, one all_gather moves the per-rank proofs, rank 0 finishes the tree and checks the
public inputs.  GL355_BENCH_ONE_DEVICE=1 rehearses it with every rank on cuda:0 (gloo instead of RCCL)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
rank, world, local_rank = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
rehearsal = os.environ.get("GL355_BENCH_ONE_DEVICE") == "1"
if rehearsal:
    local_rank = 0
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
if world > 1:
    dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world, **({} if rehearsal else {"device_id": dev}))
gl = importlib.import_module("stark-verifier_amd")
sem = importlib.import_module("stark-verifier_amd.semaphore")
rec = importlib.import_module("stark-verifier_amd.recursion")
plonk = importlib.import_module("stark-verifier_amd.plonk")
par = importlib.import_module("stark-verifier_amd.parallel")
from oracle_lib import rand_field
per_rank = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 4
log_members = 12
ctxs = [gl.Context(local_rank) for _ in range(n_ctx)]
ctx = ctxs[0]
rng = np.random.default_rng(0x357)                       # the same access set and circuits on every rank
sks = rand_field(rng, (1 << log_members, 4))
keys = ctx.hash_no_pad(np.concatenate([sks, np.zeros_like(sks)], axis=1))
aset = sem.AccessSet(ctx, keys)
topic = rand_field(rng, 4)
data, rows = aset.build(np.random.default_rng(1))
semc = plonk.NativeCircuit(ctx, data.export_blob(aset.witness_rows(rows, sks[0], topic, 0)[0]))
agg = rec.Aggregator(ctx, data.common())
members = np.arange(rank * per_rank, (rank + 1) * per_rank, dtype=np.uint64)
for attempt in ("first pass (builds the level circuits)", "circuits cached"):
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    leaves, proofs, _ = plonk.semaphore_units(ctxs, semc, None, sks, topic, aset.tree.digests, members, 7000 + 2 * per_rank * rank, want_proofs=True)
    signals = [(proofs[j], np.concatenate([aset.tree.cap[0], leaves[j]])) for j in range(per_rank)]
    out = par.aggregate_distributed(agg, signals, dist if world > 1 else None, dev if not rehearsal else None, ctxs, seed=100, rng=np.random.default_rng(7))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        proof, pis, cd = out
        n = per_rank * world
        assert pis.size == 4 + 8 * n and np.array_equal(pis[:4], aset.tree.cap[0])
        want_null = [plonk.host_hash_no_pad(np.concatenate([sks[i], topic])) for i in range(n)]
        assert np.array_equal(pis[4:4 + 4 * n].reshape(n, 4), np.stack(want_null))
        print("%s: %d ranks x %d signals -> one proof (%d tree levels: %d local + %d after the all_gather) in %.2f s" % (
            attempt, world, per_rank, len(agg.levels), per_rank.bit_length() - 1, world.bit_length() - 1, dt), flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
