#!/usr/bin/env python3
"""Timing of recursive proofs on one GPU: a wrapper proof of a depth-D Semaphore proof (wrapper.rs:35-56), the
pairwise aggregation tree (recursion.rs:187-247), and multi-context throughput of level-1 aggregation proofs."""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
gl = importlib.import_module("stark-verifier_amd")
sem = importlib.import_module("stark-verifier_amd.semaphore")
rec = importlib.import_module("stark-verifier_amd.recursion")
plonk = importlib.import_module("stark-verifier_amd.plonk")
from oracle_lib import rand_field
log_members = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_signals = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = gl.Context(0)
rng = np.random.default_rng(0x357)
sks = rand_field(rng, (1 << log_members, 4))
keys = ctx.hash_no_pad(np.concatenate([sks, np.zeros_like(sks)], axis=1))
aset = sem.AccessSet(ctx, keys)


def flat_signal(a, i, seed):
    topic = rand_field(rng, 4)
    sig, data = a.make_signal_fast(sks[i], topic, i, seed, flat_only=True)
    return (sig.proof, np.concatenate([a.tree.cap[0], sig.nullifier[0], sig.topics[0]])), data


t0 = time.perf_counter()
sigs = []
for i in range(n_signals):
    s, data = flat_signal(aset, i, 10 + i)
    sigs.append(s)
t1 = time.perf_counter()
print("%d Semaphore proofs (depth %d): %.1f ms each" % (n_signals, log_members, (t1 - t0) / n_signals * 1e3))
rc = rec.RecursiveCircuit(ctx, data.common(), k=1)
rc.build([sigs[0]], rng)
t2 = time.perf_counter()
print("wrapper circuit build + tape record: %.2f s; degree 2^%d, %d witness rows, %d tape entries, %d input words" % (
    t2 - t1, rc.data.degree_bits, rc.row_idx.size, rc.tape.shape[0], rc.n_inputs))
for k in range(1, 4):
    ta = time.perf_counter()
    rows, pis = rc.witness([sigs[k]])
    tb = time.perf_counter()
    plonk.prove_sparse(ctx, rc.data, rc.row_idx, rows, pis, 9 + k, flat_only=True)
    tc = time.perf_counter()
    print("wrapper proof %d: witness replay (C) %.2f ms, gl355_prove_sparse %.2f ms" % (k, (tb - ta) * 1e3, (tc - tb) * 1e3))

agg = rec.Aggregator(ctx, data.common())
ta = time.perf_counter()
proof, pis, cd = agg.aggregate(sigs, seed=100, rng=rng)
tb = time.perf_counter()
print("aggregate %d signals, first pass (builds %d circuits): %.2f s" % (n_signals, len(agg.levels), tb - ta))
proof, pis, cd = agg.aggregate(sigs, seed=200)
tc = time.perf_counter()
print("aggregate %d signals, circuits cached: %.1f ms (%d proofs, degrees %s)" % (
    n_signals, (tc - tb) * 1e3, n_signals - 1, [l.data.degree_bits for l in agg.levels]))
lvl = agg.levels[0]
for k in range(2):
    ta = time.perf_counter()
    rows, p2 = lvl.witness(sigs[2 * k:2 * k + 2])
    tb = time.perf_counter()
    plonk.prove_sparse(ctx, lvl.data, lvl.row_idx, rows, p2, 9 + k, flat_only=True)
    tc = time.perf_counter()
    print("level-1 aggregation proof: witness replay %.2f ms, gl355_prove_sparse %.2f ms" % ((tb - ta) * 1e3, (tc - tb) * 1e3))

# multi-context throughput of level-1 aggregation proofs (independent pairs, one context + host thread each)
ctxs = [gl.Context(0) for _ in range(n_ctx)]
pds = [lvl.data.prover_data(c) for c in ctxs]
per = 6
def worker(c):
    for j in range(per):
        rows, p2 = lvl.witness(sigs[0:2])
        plonk.prove_sparse(c, lvl.data, lvl.row_idx, rows, p2, j, flat_only=True)
for c in ctxs:
    worker.__call__(c) if False else None
ths = [threading.Thread(target=worker, args=(c,)) for c in ctxs]
ta = time.perf_counter()
[t.start() for t in ths]; [t.join() for t in ths]
tb = time.perf_counter()
print("level-1 aggregation throughput, %d contexts: %.1f proofs/s" % (n_ctx, n_ctx * per / (tb - ta)))
