#!/usr/bin/env python3
"""Parity under load: M random units (random members, one topic, seeds by unit) are proven by the native batch runtime on 22
concurrent prover contexts and again one by one on a single context; all 2 M proofs must be byte-identical between the two runs
(determinism under concurrency), and K sampled units are proven by the CPU restatement of prove() (oracle/gl_prover.c) and
compared byte for byte (Semaphore proof and recursive proof).  Test infrastructure (it drives the oracle), run by hand on a GPU box:
  python tests/parity_sweep.py [M=256] [K=6]          (output of the last run: profiles/r01f_parity_sweep.txt)"""
import hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
lib = importlib.import_module("stark-verifier_amd._lib").load(init_torch=False)
assert lib.gl355_runtime_config(0, 22, 1) == 0
import bench
gl = importlib.import_module("stark-verifier_amd")
plonk = importlib.import_module("stark-verifier_amd.plonk")
from oracle_lib import CpuProver, Oracle

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pr = bench.RecursiveProvers(gl, 0, 22, replay_threads=2)
rng = np.random.default_rng(0x5EED)
members = rng.integers(0, pr.sks.shape[0], size=M, dtype=np.uint64)
seed_base = 77000
t0 = time.perf_counter()
leaves, proofs, per = plonk.semaphore_units(pr.sets, pr.sem, pr.nat, pr.sks, pr.topic, pr.aset.tree.digests, members, seed_base, want_proofs=True)
t1 = time.perf_counter()
print("22 contexts: %d units in %.2f s (%.1f units/s), units per context %s" % (M, t1 - t0, M / (t1 - t0), per))
leaves1, proofs1, _ = plonk.semaphore_units(pr.sets[:1], pr.sem, pr.nat, pr.sks, pr.topic, pr.aset.tree.digests, members, seed_base, want_proofs=True)
t2 = time.perf_counter()
same = np.array_equal(proofs, proofs1) and np.array_equal(leaves, leaves1)
print("1 context: %.2f s; recursive proofs and leaves identical to the concurrent run: %s" % (t2 - t1, same))
print("sha256 of the %d recursive proofs: %s" % (M, hashlib.sha256(np.ascontiguousarray(proofs).tobytes()).hexdigest()))
assert same
# ---- CPU restatement on K sampled units ------------------------------------------------------------------------------------
orc = Oracle()
orc.L.orc_set_num_threads(bench.host_cores())
cpu_in = CpuProver.from_circuit_data(orc, pr.inner_data)
cpu_out = CpuProver.from_circuit_data(orc, pr.rc.data)
_, rows = pr.aset.build(np.random.default_rng(1))
ok = True
for j in rng.choice(M, size=K, replace=False):
    m = int(members[j])
    idx, vals, pi = pr.aset.witness_rows(rows, pr.sks[m], pr.topic, m)
    assert np.array_equal(idx, pr.inner_rows[0])
    flat_in = cpu_in.prove_sparse(idx, vals, pi, seed_base + 2 * int(j))
    g_in, g_pis = pr.sem.semaphore_prove(pr.sets[0], pr.sks[m], pr.topic, m, pr.aset.tree.prove_host(m), seed_base + 2 * int(j))
    wrows, wpis = pr.rc.witness([(flat_in, np.asarray(g_pis))])
    flat_out = cpu_out.prove_sparse(pr.rc.row_idx, wrows, wpis, seed_base + 2 * int(j) + 1)
    e_in, e_out = bool(np.array_equal(g_in, flat_in)), bool(np.array_equal(proofs[j], flat_out))
    print("unit %4d (member %7d): Semaphore proof GPU == CPU: %s; recursive proof (concurrent run) == CPU: %s" % (j, m, e_in, e_out), flush=True)
    ok &= e_in and e_out
assert ok
print("PARITY SWEEP OK: %d units deterministic under 22-way concurrency, %d units byte-identical to the CPU prover" % (M, K))
