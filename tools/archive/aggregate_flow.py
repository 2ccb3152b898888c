#!/usr/bin/env python3
"""The reference's whole flow on one GPU (README "Aggregation" table): N Semaphore signals (depth 20) -> pairwise aggregation tree
of recursive proofs (recursion.rs:187-247) -> final wrap under the BN254-Poseidon config (wrapper.rs:35-56).  Circuits are built
on the first pass (one per tree level) and reused."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.init()
gl = importlib.import_module("stark-verifier_amd")
sem = importlib.import_module("stark-verifier_amd.semaphore")
rec = importlib.import_module("stark-verifier_amd.recursion")
plonk = importlib.import_module("stark-verifier_amd.plonk")
from oracle_lib import rand_field
n_signals = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n_ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 12
log_members = 20
ctxs = [gl.Context(0) for _ in range(n_ctx)]
ctx = ctxs[0]
rng = np.random.default_rng(0x357)
sks = rand_field(rng, (1 << log_members, 4))
keys = ctx.hash_no_pad(np.concatenate([sks, np.zeros_like(sks)], axis=1))
aset = sem.AccessSet(ctx, keys)
topic = rand_field(rng, 4)
data, rows = aset.build(rng)
idx, _, _ = aset.witness_rows(rows, sks[0], topic, 0)
semc = plonk.NativeCircuit(ctx, data.export_blob(idx))
agg = rec.Aggregator(ctx, data.common())
wrap = None
for attempt in ("first pass (builds every level's circuit)", "circuits cached"):
    t0 = time.perf_counter()
    leaves, proofs, _ = plonk.semaphore_units(ctxs, semc, None, sks, topic, aset.tree.digests, np.arange(n_signals, dtype=np.uint64), 7000, want_proofs=True)
    signals = [(proofs[j], np.concatenate([aset.tree.cap[0], leaves[j]])) for j in range(n_signals)]
    t1 = time.perf_counter()
    proof, pis, cd = agg.aggregate(signals, seed=100, rng=rng, ctxs=ctxs)
    t2 = time.perf_counter()
    if wrap is None:
        wrap = rec.WrapperCircuit(ctx, cd).build([(proof, pis)], rng)
    wflat, wpis = wrap.native().prove_tape(ctx, np.concatenate([proof, pis]), 9)
    t3 = time.perf_counter()
    print("%s: %d signals %.2f s, aggregation tree (%d proofs, %d levels) %.2f s, BN254 wrap %.2f s; total %.2f s" % (
        attempt, n_signals, t1 - t0, n_signals - 1, len(agg.levels), t2 - t1, t3 - t2, t3 - t0))
assert np.array_equal(wpis[:4], aset.tree.cap[0]) and wpis.size == 4 + 8 * n_signals
print("final proof: %d words, %d public inputs (root | %d nullifiers | %d topics); level degrees %s, wrap degree 2^%d" % (
    wflat.size, wpis.size, n_signals, n_signals, [l.data.degree_bits for l in agg.levels], wrap.data.degree_bits))
