"""GPU: circuit artifacts (CircuitData.export_blob -> gl355_circuit_load) and the native per-proof entry points
gl355_semaphore_prove / gl355_circuit_prove_tape / gl355_circuit_prove_rows: byte-identical to the host-orchestrated path."""
import importlib
import threading

import numpy as np
import pytest

import plonk_verifier as pv
from oracle_lib import rand_field
from test_gpu_prover import make_access_set

pytestmark = pytest.mark.gpu


def test_native_semaphore_and_recursive_proofs(gl, ctx, orc):
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    rec = importlib.import_module("stark-verifier_amd.recursion")
    aset, sks, rng = make_access_set(gl, ctx, 4, 0x901)
    topic = rand_field(rng, 4)
    data, rows = aset.build(None)
    idx, vals, pi = aset.witness_rows(rows, sks[5], topic, 5)
    want = plonk.prove_sparse(ctx, data, idx, vals, pi, 33, flat_only=True)
    sem = plonk.NativeCircuit(ctx, data.export_blob(idx))
    assert sem.n_rows == idx.size and sem.degree_bits == data.degree_bits
    got, pis = sem.semaphore_prove(ctx, sks[5], topic, 5, aset.tree.prove(5), 33)
    assert np.array_equal(got, want) and np.array_equal(pis, pi)
    assert np.array_equal(sem.prove_rows(ctx, vals, pi, 33), want)
    # recursive circuit: artifact carries the tape
    inner = (want, pi)
    rc = rec.RecursiveCircuit(ctx, data.common(), k=1).build([inner], rng)
    nat = rc.native()
    flat_py, pis_py = rc.prove_flat([inner], seed=44)
    flat_nat, pis_nat = nat.prove_tape(ctx, np.concatenate([inner[0], inner[1]]), 44)
    assert np.array_equal(flat_nat, flat_py) and np.array_equal(pis_nat, pis_py)
    proof = plonk.parse_proof(rc.data, flat_nat)
    proof["public_inputs"] = pis_nat
    pv.verify(orc, rc.data.common(), proof)
    # an invalid inner proof is refused with GL355_E_WITNESS
    bad = inner[0].copy()
    bad[50] ^= np.uint64(1)
    with pytest.raises(gl.Gl355Error) as ei:
        nat.prove_tape(ctx, np.concatenate([bad, inner[1]]), 44)
    assert ei.value.code == -6
    # the handle is shared by other contexts of the device, concurrently
    ctxs = [gl.Context(0) for _ in range(3)]
    outs = [None] * 3

    def work(t):
        outs[t] = nat.prove_tape(ctxs[t], np.concatenate([inner[0], inner[1]]), 44)[0]
    ths = [threading.Thread(target=work, args=(t,)) for t in range(3)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert all(np.array_equal(o, flat_py) for o in outs)
    for c in ctxs:
        c.close()


def test_artifact_validation(gl, ctx):
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    aset, sks, rng = make_access_set(gl, ctx, 2, 0x902)
    data, rows = aset.build(None)
    idx, vals, pi = aset.witness_rows(rows, sks[1], rand_field(rng, 4), 1)
    blob = data.export_blob(idx)
    plonk.NativeCircuit(ctx, blob).close()
    for mutate in ("magic", "truncate", "table", "digest", "rowidx"):
        b = blob.copy()
        if mutate == "magic":
            b[0] ^= np.uint64(1)
        elif mutate == "truncate":
            b = b[:-1]
        elif mutate == "table":
            b[110 + 12345] ^= np.uint64(1)            # a selector / constant value: the digest no longer matches
        elif mutate == "digest":
            b[107] ^= np.uint64(1)
        elif mutate == "rowidx":
            b[b.size - 1] = np.uint64(1 << 40)                        # last row index (no pi positions, no tape in this artifact)
        with pytest.raises(gl.Gl355Error):
            plonk.NativeCircuit(ctx, b)
