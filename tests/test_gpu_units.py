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
