"""GPU: lock-step units.  gl355_prove_sparse_units proves B independent witnesses of one circuit with a unit dimension in every
kernel (one forest of Merkle trees, one NTT batch, per-unit transcripts / challenges / keys); every unit's proof must be byte for
byte the proof of that unit alone (gl355_prove_sparse, B = 1) -- and therefore the CPU restatement's."""
import importlib

import numpy as np
import pytest

from oracle_lib import CpuProver, rand_field
from test_gpu_prover import make_access_set

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("units", [2, 5, 16])
def test_units_equal_single_proofs(gl, ctx, orc, units):
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    aset, sks, rng = make_access_set(gl, ctx, 4, 0x811 + units)
    data, rows = aset.build(rng)
    topics = rand_field(rng, (units, 4))
    members = rng.integers(0, 16, size=units)
    wit = [aset.witness_rows(rows, sks[int(m)], topics[j], int(m)) for j, m in enumerate(members)]
    idx = wit[0][0]
    vals = np.stack([w[1] for w in wit])
    pis = np.stack([w[2] for w in wit])
    seeds = [9000 + 7 * j for j in range(units)]
    got = plonk.prove_sparse_units(ctx, data, idx, vals, pis, seeds)
    for j in range(units):
        single = plonk.prove_sparse(ctx, data, idx, vals[j], pis[j], seeds[j], flat_only=True)
        assert np.array_equal(got[j], single), "unit %d of %d differs from its single proof" % (j, units)
    cpu = CpuProver.from_circuit_data(orc, data)
    j = units - 1
    assert np.array_equal(got[j], cpu.prove_sparse(idx, vals[j], pis[j], seeds[j]))
    # OS-random keys: proofs of the same witnesses differ from the keyed ones and from each other's salts
    rnd = plonk.prove_sparse_units(ctx, data, idx, vals[:2], pis[:2], None)
    assert not np.array_equal(rnd[0], got[0])


def test_units_argument_errors(gl, ctx):
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    aset, sks, rng = make_access_set(gl, ctx, 2, 0x812)
    data, rows = aset.build(rng)
    idx, vals, pi = aset.witness_rows(rows, sks[1], rand_field(rng, 4), 1)
    with pytest.raises(gl.Gl355Error):
        plonk.prove_sparse_units(ctx, data, idx, np.stack([vals] * 17), np.stack([pi] * 17), [1] * 17)


def test_gpu_proofs_pass_the_product_verifier(gl, ctx):
    """gl355_circuit_verify / CircuitData.verify (CircuitData::verify, access_set.rs:170-175) on proofs of the lock-step prover"""
    import ctypes as C
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    aset, sks, rng = make_access_set(gl, ctx, 3, 0x813)
    data, rows = aset.build(rng)
    wit = [aset.witness_rows(rows, sks[m], rand_field(rng, 4), m) for m in (1, 6, 3)]
    idx = wit[0][0]
    flats = plonk.prove_sparse_units(ctx, data, idx, np.stack([w[1] for w in wit]), np.stack([w[2] for w in wit]), None)   # OS-random blinding
    nat = plonk.NativeCircuit(ctx, data.export_blob(idx))
    for f, w in zip(flats, wit):
        assert data.verify(f, w[2])
        pi = np.ascontiguousarray(w[2], dtype=np.uint64)
        assert ctx.lib.gl355_circuit_verify(nat.h, f.ctypes.data, f.size, pi.ctypes.data, pi.size) == 0
    bad = flats[0].copy()
    bad[-7] ^= np.uint64(4)
    pi = np.ascontiguousarray(wit[0][2], dtype=np.uint64)
    assert ctx.lib.gl355_circuit_verify(nat.h, bad.ctypes.data, bad.size, pi.ctypes.data, pi.size) == -7
    assert b"verify" in ctx.lib.gl355_verify_last_error() or ctx.lib.gl355_verify_last_error()
    assert ctx.lib.gl355_circuit_verify(nat.h, flats[1].ctypes.data, flats[1].size, pi.ctypes.data, pi.size) == -7     # another unit's public inputs
