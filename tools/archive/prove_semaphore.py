#!/usr/bin/env python3
"""End-to-end Semaphore proof(s) on one GPU with per-stage timings (BASELINE cfg-4 building block).
  python tools/prove_semaphore.py [log_members=20] [n_proofs=3] [--verify]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
gl = importlib.import_module("stark-verifier_amd")
sem = importlib.import_module("stark-verifier_amd.semaphore")
from oracle_lib import rand_field

log_members = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = gl.Context(0)
rng = np.random.default_rng(0x357)
t0 = time.perf_counter()
sks = rand_field(rng, (1 << log_members, 4))
keys = ctx.hash_no_pad(np.concatenate([sks, np.zeros_like(sks)], axis=1))
aset = sem.AccessSet(ctx, keys)
t1 = time.perf_counter()
print("access set: 2^%d members, keys + tree %.3f s" % (log_members, t1 - t0))
data, rows = aset.build(np.random.default_rng(1))
t2 = time.perf_counter()
print("circuit build: degree 2^%d, gates %s, %.3f s" % (data.degree_bits, data.gates, t2 - t1))
topic = rand_field(rng, 4)
for k in range(n_proofs):
    ta = time.perf_counter()
    sig, _ = aset.make_signal(sks[40 + k], topic, 40 + k, np.random.default_rng(0x458 + k))
    print("proof %d (gl355_prove): %.1f ms total incl. witness generation" % (k, (time.perf_counter() - ta) * 1e3))
for k in range(n_proofs):
    tm = {}
    ta = time.perf_counter()
    sig, _ = aset.make_signal(sks[12 + k], topic, 12 + k, np.random.default_rng(0x358 + k), timings=tm)
    tb = time.perf_counter()
    print("proof %d: %.1f ms total | " % (k, (tb - ta) * 1e3) + ", ".join("%s %.1f" % (a, b * 1e3) for a, b in tm.items()))
if "--verify" in sys.argv:
    import plonk_verifier as pv
    from oracle_lib import Oracle
    ta = time.perf_counter()
    pv.verify(Oracle(), data.common(), sig.proof)
    print("verifier restatement: OK (%.2f s)" % (time.perf_counter() - ta))
