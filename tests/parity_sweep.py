#!/usr/bin/env python3
"""Parity under load: M random units (random members, one topic, per-unit blinding keys derived from one batch key) are proven by
the native batch runtime on `contexts` concurrent prover contexts and again one by one on a single context; all M recursive
proofs must be byte-identical between the two runs (determinism under concurrency), and K sampled units are proven by the CPU
restatement of prove() (oracle/gl_prover.c) and compared byte for byte (Semaphore proof and recursive proof).
Test infrastructure (it drives the oracle).  Collected at reduced size by tests/test_gpu_sweep.py; by hand on a GPU box:
  python tests/parity_sweep.py [M=256] [K=6]"""
import hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def run_sweep(M=256, K=6, contexts=22, log_members=20, blocking_sync=False, out=print, check_root=False):
    import bench
    gl = importlib.import_module("stark-verifier_amd")
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    from oracle_lib import CpuProver, Oracle
    pr = bench.RecursiveProvers(gl, 0, contexts, log_members=log_members, replay_threads=2, blocking_sync=blocking_sync)
    rng = np.random.default_rng(0x5EED)
    members = rng.integers(0, pr.sks.shape[0], size=M, dtype=np.uint64)
    batch_key = 77000
    t0 = time.perf_counter()
    leaves, proofs, per = plonk.semaphore_units(pr.sets, pr.sem, pr.nat, pr.sks, pr.topic, pr.aset.tree.digests, members, batch_key, want_proofs=True)
    t1 = time.perf_counter()
    out("%d contexts: %d units in %.2f s (%.1f units/s), units per context %s" % (contexts, M, t1 - t0, M / (t1 - t0), per))
    leaves1, proofs1, _ = plonk.semaphore_units(pr.sets[:1], pr.sem, pr.nat, pr.sks, pr.topic, pr.aset.tree.digests, members, batch_key, want_proofs=True)
    t2 = time.perf_counter()
    same = np.array_equal(proofs, proofs1) and np.array_equal(leaves, leaves1)
    out("1 context: %.2f s; recursive proofs and leaves identical to the concurrent run: %s" % (t2 - t1, same))
    out("sha256 of the %d recursive proofs: %s" % (M, hashlib.sha256(np.ascontiguousarray(proofs).tobytes()).hexdigest()))
    assert same
    if check_root:
        # BASELINE configs[4], one GPU's share: the (nullifier | topic) leaves of the step and the aggregation root over them (rank 0's
        # gl355_aggregation_root) against the oracle: nullifier = hash_no_pad(sk | topic) (signal.rs / circuit.rs:53), root = the
        # Poseidon Merkle cap of height 0 over the leaves padded to a power of two
        par = importlib.import_module("stark-verifier_amd.parallel")
        from oracle_lib import Oracle as _O
        o = _O()
        want_null = np.stack([o.hash_no_pad(np.concatenate([pr.sks[int(m)], pr.topic])) for m in members])
        assert np.array_equal(leaves[:, :4], want_null), "nullifiers differ from hash(sk | topic)"
        assert np.array_equal(leaves[:, 4:], np.tile(pr.topic, (M, 1)))
        root = par.aggregation_root(pr.sets[0], leaves)
        want_root = o.merkle_build(par.pad_pow2(leaves), 0)[1]
        assert np.array_equal(root, want_root), "aggregation root differs from the oracle"
        # every recursive proof of the step passes the product-side verifier (CircuitData::verify) with its own public inputs
        for j in range(M):
            opis = np.concatenate([pr.root, leaves[j]])
            pr.rc.data.verify(proofs[j], opis)
        out("aggregation root over %d leaves == oracle; %d recursive proofs verified" % (M, M))
    # ---- CPU restatement on K sampled units ----------------------------------------------------------------------------
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
        k_in, k_out = plonk.derive_key(batch_key, 2 * int(j)), plonk.derive_key(batch_key, 2 * int(j) + 1)
        flat_in = cpu_in.prove_sparse(idx, vals, pi, k_in)
        g_in, g_pis = pr.sem.semaphore_prove(pr.sets[0], pr.sks[m], pr.topic, m, pr.aset.tree.prove_host(m), k_in)
        wrows, wpis = pr.rc.witness([(flat_in, np.asarray(g_pis))])
        flat_out = cpu_out.prove_sparse(pr.rc.row_idx, wrows, wpis, k_out)
        e_in, e_out = bool(np.array_equal(g_in, flat_in)), bool(np.array_equal(proofs[j], flat_out))
        out("unit %4d (member %7d): Semaphore proof GPU == CPU: %s; recursive proof (concurrent run) == CPU: %s" % (j, m, e_in, e_out))
        ok &= e_in and e_out
    pr.close() if hasattr(pr, "close") else None
    assert ok
    out("PARITY SWEEP OK: %d units deterministic under %d-way concurrency, %d units byte-identical to the CPU prover" % (M, contexts, K))
    return hashlib.sha256(np.ascontiguousarray(proofs).tobytes()).hexdigest()


if __name__ == "__main__":
    import torch  # noqa: F401  (its libamdhip64 must be the one in the process)
    lib = importlib.import_module("stark-verifier_amd._lib").load(init_torch=False)
    assert lib.gl355_runtime_config(0, 22, 1) == 0
    run_sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 6)
