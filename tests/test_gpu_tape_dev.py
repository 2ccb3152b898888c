"""GPU: witness generation on the device.  The tape interpreter kernel (csrc/witness_tape_dev.hip) must fill
exactly the rows the host replay fills -- every wire of the ~5 900 gate rows of the recursive verifier circuit, for several units
at once -- and refuse an invalid inner proof at the same tape entry; the batch runtime must give the same proofs with either."""
import importlib

import numpy as np
import pytest

from oracle_lib import rand_field
from test_gpu_prover import make_access_set

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def circuits(gl, ctx):
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    rec = importlib.import_module("stark-verifier_amd.recursion")
    aset, sks, rng = make_access_set(gl, ctx, 4, 0x7A9)
    topic = rand_field(rng, 4)
    data, rows = aset.build(None)
    idx, vals, pi = aset.witness_rows(rows, sks[0], topic, 0)
    sem = plonk.NativeCircuit(ctx, data.export_blob(idx))
    flat0, pis0 = sem.semaphore_prove(ctx, sks[0], topic, 0, aset.tree.prove_host(0), 1)
    rc = rec.RecursiveCircuit(ctx, data.common(), k=1).build([(flat0, pis0)], rng)
    return plonk, aset, sks, topic, sem, rc, rc.native()


def test_device_rows_equal_host_rows(gl, ctx, circuits):
    plonk, aset, sks, topic, sem, rc, nat = circuits
    inputs = []
    for j, m in enumerate((3, 9, 15, 0, 7)):
        f, p = sem.semaphore_prove(ctx, sks[m], topic, m, aset.tree.prove_host(m), 100 + j)
        inputs.append(np.concatenate([f, p]))
    inputs = np.stack(inputs)
    h_rows, h_pis = nat.witness_rows(ctx, inputs, on_device=False)
    d_rows, d_pis = nat.witness_rows(ctx, inputs, on_device=True)
    assert np.array_equal(h_pis, d_pis)
    if not np.array_equal(h_rows, d_rows):
        bad = np.argwhere(h_rows != d_rows)
        raise AssertionError("device rows differ from host rows at %d positions, first (unit, row, wire) = %s" % (len(bad), bad[0]))
    # and they are the rows of the eager gadget pass
    py_rows, py_pis = rc.witness([(inputs[0][:-12], inputs[0][-12:])])
    assert np.array_equal(py_rows, d_rows[0])
    # an invalid inner proof (one sibling word of unit 1 flipped) is refused at the same entry by both
    bad = inputs.copy()
    bad[1, -15] ^= np.uint64(1)
    entries = []
    for dev in (False, True):
        with pytest.raises(gl.Gl355Error) as ei:
            nat.witness_rows(ctx, bad, on_device=dev)
        assert ei.value.code == -6
        entries.append(ei.value.failed_entry)
    assert entries[0] == entries[1]


def test_batch_runtime_same_proofs_with_either_replay(gl, ctx, circuits):
    plonk, aset, sks, topic, sem, rc, nat = circuits
    members = np.array([1, 14, 6, 2, 11, 8, 5, 0, 13, 3], dtype=np.uint64)
    out = []
    for dev in (1, 0):
        ctxs = [gl.Context(0) for _ in range(2)]
        for c in ctxs:
            c.set_option(6, dev)          # GL355_OPT_DEVICE_REPLAY
            c.set_option(5, 4)            # GL355_OPT_BATCH_UNITS
        leaves, proofs, per = plonk.semaphore_units(ctxs, sem, nat, sks, topic, aset.tree.digests, members, 555, want_proofs=True)
        out.append((leaves, proofs))
        for c in ctxs:
            c.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
