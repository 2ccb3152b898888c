"""GPU: gadget builder + recursive verification (recursion.rs:25-185, wrapper.rs:35-56).  Every proof produced here
must pass the restatement of the reference's verifier (tests/plonk_verifier.py)."""
import ctypes as C
import importlib

import numpy as np
import pytest

import plonk_verifier as pv
import pymodel as pm
from oracle_lib import P, rand_field
from test_gpu_prover import make_access_set

pytestmark = pytest.mark.gpu


def test_gadget_circuit_proves_and_verifies(gl, ctx, orc):
    """a small circuit exercising every gate the builder emits; its proof verifies and the in-circuit values equal
    the big-integer model."""
    gad = importlib.import_module("stark-verifier_amd.gadgets")
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    rng = np.random.default_rng(0x601)
    b = gad.GadgetBuilder()
    inputs = rand_field(rng, 8 + 1 + 16 + 100)          # every free input is tagged with its position: the tape can replay them
    inputs[8] = 1
    tagged = [plonk.Src(v, i) for i, v in enumerate(inputs)]
    xs = b.add_virtual_targets(tagged[:8])
    # base arithmetic
    s = b.add(b.mul(xs[0], xs[1]), xs[2])
    assert s.v == (xs[0].v * xs[1].v + xs[2].v) % P
    sel = b.select(b.one(), xs[3], xs[4])
    assert sel.v == xs[3].v
    # extension arithmetic + inverse
    e1, e2 = (xs[0], xs[1]), (xs[2], xs[3])
    pr = b.ext_mul(e1, e2)
    assert (pr[0].v, pr[1].v) == pm.ext_mul((e1[0].v, e1[1].v), (e2[0].v, e2[1].v))
    q = b.ext_div(pr, e2)
    assert (q[0].v, q[1].v) == (e1[0].v, e1[1].v)
    assert tuple(t.v for t in b.ext_exp_const(e1, 11)) == pv.ext_pow((e1[0].v, e1[1].v), 11)
    # Poseidon sponge, swapped permutation
    h = b.hash_n_to_hash_no_pad(xs + xs[:3])
    assert [t.v for t in h] == [int(v) for v in orc.hash_no_pad(np.array([t.v for t in xs + xs[:3]], dtype=np.uint64))]
    bit = b.add_virtual_target(tagged[8])
    b.assert_bool(bit)
    sw = b.permute_swapped(xs + [b.zero()] * 4, swap=bit)
    st = np.array([t.v for t in xs[4:8] + xs[0:4]] + [0] * 4, dtype=np.uint64)
    assert [t.v for t in sw] == [int(v) for v in orc.permute(st)]
    # bits, random access, reductions, MDS
    bits = b.split_le_64(xs[5])
    assert sum(t.v << i for i, t in enumerate(bits)) == xs[5].v
    assert b.le_sum(bits[:12]).v == xs[5].v & 0xFFF
    items = b.add_virtual_targets(tagged[9:25])
    idx = b.le_sum(bits[:4])
    assert b.random_access(idx, items).v == items[idx.v].v
    alpha = (xs[6], xs[7])
    coeffs = b.add_virtual_targets(tagged[25:125])
    red = b.reduce_with_powers_base(coeffs, alpha)
    assert (red[0].v, red[1].v) == pv.reduce_with_powers([pv.base(c.v) for c in coeffs], (alpha[0].v, alpha[1].v))
    ecoeffs = [(coeffs[2 * i], coeffs[2 * i + 1]) for i in range(40)]
    red2 = b.reduce_with_powers_ext(ecoeffs, alpha)
    assert (red2[0].v, red2[1].v) == pv.reduce_with_powers([(a.v, c.v) for a, c in ecoeffs], (alpha[0].v, alpha[1].v))
    mds = b.mds_ext(ecoeffs[:12])
    b.register_public_inputs([s, h[0], red[1]])
    pi_vals = b.finalize_public_inputs()
    data = b.cb.build(ctx, rng)
    idx_rows, vals = b.sparse_witness()
    proof = plonk.prove_sparse(ctx, data, idx_rows, vals, np.array(pi_vals, dtype=np.uint64), 7)
    pv.verify(orc, data.common(), proof)
    # the recorded witness tape reproduces the witness (gl355_witness_replay), also for other inputs
    tape, ridx, pi_pos = b.witness_tape()
    assert np.array_equal(ridx, idx_rows) and getattr(b, "untagged_inputs", 0) == 0

    def replay(inp):
        rows = np.empty_like(vals)
        failed = C.c_uint64(0)
        rc = ctx.lib.gl355_witness_replay(tape.ctypes.data, tape.shape[0], inp.ctypes.data, inp.size, rows.ctypes.data, rows.size, 135, C.byref(failed))
        return rc, rows, failed.value
    rc, rows, _ = replay(inputs)
    assert rc == 0 and np.array_equal(rows, vals)
    assert [int(v) for v in rows.reshape(-1)[pi_pos]] == pi_vals
    other = rand_field(rng, inputs.size)
    other[8] = 0
    rc, rows2, _ = replay(other)
    assert rc == 0
    pv.verify(orc, data.common(), plonk.prove_sparse(ctx, data, ridx, rows2, rows2.reshape(-1)[pi_pos], 8))
    other[8] = 2                                        # not a bit: assert_bool's ASSERT_EQ entry fails
    rc, _, failed = replay(other)
    assert rc == -6 and tape[failed][0] == gad.TAPE_ASSERT_EQ
    assert ctx.lib.gl355_witness_replay(tape.ctypes.data, tape.shape[0], inputs.ctypes.data, 5, rows.ctypes.data, rows.size, 135, None) == -1
    # a violated gate (wrong product in an ArithmeticGate slot) must not verify
    bad = vals.copy()
    arith_rows = [k for k, r in enumerate(idx_rows) if data.gates[data.row_gate[r]][0] == 5]
    bad[arith_rows[0], 3] ^= np.uint64(1)
    with pytest.raises(pv.VerifyError):
        pv.verify(orc, data.common(), plonk.prove_sparse(ctx, data, idx_rows, bad, np.array(pi_vals, dtype=np.uint64), 7))


def test_recursive_proof_of_semaphore(gl, ctx, orc):
    """wrapper.rs:35-56 with PoseidonGoldilocksConfig outer: a proof that verifies a Semaphore proof, then a proof
    that verifies THAT proof (all gate evaluators in-circuit)."""
    rec = importlib.import_module("stark-verifier_amd.recursion")
    aset, sks, rng = make_access_set(gl, ctx, 4, 0x602)
    topic = rand_field(rng, 4)
    sig, data = aset.make_signal_fast(sks[3], topic, 3, 11)
    inner_cd = data.common()
    pv.verify(orc, inner_cd, sig.proof)
    rc1 = rec.RecursiveCircuit(ctx, inner_cd, k=1)
    p1 = rc1.prove([sig.proof], seed=21, rng=rng)
    cd1 = rc1.data.common()
    pv.verify(orc, cd1, p1)
    assert np.array_equal(p1["public_inputs"], sig.proof["public_inputs"])       # root | nullifier | topic re-exposed
    print("recursive circuit: degree 2^%d, gates %s" % (rc1.data.degree_bits, rc1.data.gates))
    # a second inner proof re-uses the layout
    sig2, _ = aset.make_signal_fast(sks[9], topic, 9, 12)
    p1b = rc1.prove([sig2.proof], seed=22)
    pv.verify(orc, cd1, p1b)
    # tampered inner proof: the in-circuit verifier's own consistency check (eager witness) refuses to build a witness
    badp = dict(sig2.proof)
    badp["public_inputs"] = sig2.proof["public_inputs"].copy()
    badp["public_inputs"][9] ^= np.uint64(1)
    with pytest.raises(AssertionError):
        rc1.prove([badp], seed=23)
    # recursion over the recursive proof
    rc2 = rec.RecursiveCircuit(ctx, cd1, k=1)
    p2 = rc2.prove([p1], seed=31, rng=rng)
    pv.verify(orc, rc2.data.common(), p2)
    assert np.array_equal(p2["public_inputs"], sig.proof["public_inputs"])
    print("2nd-level recursive circuit: degree 2^%d" % rc2.data.degree_bits)


def flat_signal(aset, sk, topic, index, seed):
    sig, data = aset.make_signal_fast(sk, topic, index, seed, flat_only=True)
    return (sig.proof, np.concatenate([aset.tree.cap[0], sig.nullifier[0], sig.topics[0]])), data


def test_witness_tape_equals_python_pass(gl, ctx, orc):
    """the recorded tape replayed in C on a new inner proof gives exactly the rows of the eager Python gadget pass;
    an invalid inner proof is refused with GL355_E_WITNESS"""
    rec = importlib.import_module("stark-verifier_amd.recursion")
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    aset, sks, rng = make_access_set(gl, ctx, 3, 0x603)
    topic = rand_field(rng, 4)
    s0, data = flat_signal(aset, sks[1], topic, 1, 5)
    s1, _ = flat_signal(aset, sks[6], rand_field(rng, 4), 6, 6)
    cd = data.common()
    rc = rec.RecursiveCircuit(ctx, cd, k=1).build([s0], rng)
    rows, pis = rc.witness([s1])
    parsed = plonk.parse_proof(cd, s1[0])
    parsed["public_inputs"] = s1[1]
    b, pi_vals = rc._run([parsed])
    assert b.structure_hash() == rc.structure
    idx, vals = b.sparse_witness()
    assert np.array_equal(idx, rc.row_idx) and np.array_equal(vals, rows)
    assert [int(v) for v in pis] == pi_vals == [int(v) for v in s1[1]]
    proof, pis2 = rc.prove_flat([s1], seed=9)
    outer = plonk.parse_proof(rc.data, proof)
    outer["public_inputs"] = pis2
    pv.verify(orc, rc.data.common(), outer)
    for word in (40, s1[0].size - 3):                   # a cap word / a Merkle sibling of the last query
        bad = s1[0].copy()
        bad[word] ^= np.uint64(1)
        with pytest.raises(AssertionError):
            rc.witness([(bad, s1[1])])
    with pytest.raises(AssertionError):
        badpi = s1[1].copy()
        badpi[5] ^= np.uint64(4)
        rc.witness([(s1[0], badpi)])


def same(a, b):
    """equality of common-data dicts (lists, tuples, integers, numpy arrays)"""
    if isinstance(a, dict):
        return isinstance(b, dict) and a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return isinstance(b, (list, tuple)) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return np.array_equal(np.asarray(a), np.asarray(b))
    return a == b


def test_aggregate_four_signals(gl, ctx, orc):
    """recursion.rs:187-247: 4 signals -> 2 level-1 proofs -> 1 level-2 proof; public inputs root | nullifiers | topics"""
    rec = importlib.import_module("stark-verifier_amd.recursion")
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    aset, sks, rng = make_access_set(gl, ctx, 3, 0x604)
    members = [2, 7, 0, 5]
    topics = [rand_field(rng, 4) for _ in members]
    sigs, data = [], None
    for k, (m, t) in enumerate(zip(members, topics)):
        s, data = flat_signal(aset, sks[m], t, m, 40 + k)
        sigs.append(s)
    agg = rec.Aggregator(ctx, data.common())
    proof, pis, cd = agg.aggregate(sigs, seed=100, rng=rng)
    ctx2 = gl.Context(0)                                   # the nodes of a level in parallel on two prover contexts: same proofs
    proof_p, pis_p, _ = agg.aggregate(sigs, seed=100, ctxs=[ctx, ctx2])
    assert np.array_equal(proof_p, proof) and np.array_equal(pis_p, pis)
    ctx2.close()
    # the default (seed=None): every proof of the tree is blinded under a fresh OS-random key -- two runs differ, both verify
    # (small public integers as keys, shared between levels and ranks, are gone; a seed now derives per-(domain, level, node) keys)
    fresh_a, pis_a, _ = agg.aggregate(sigs)
    fresh_b, pis_b, _ = agg.aggregate(sigs)
    assert not np.array_equal(fresh_a, fresh_b) and np.array_equal(pis_a, pis) and np.array_equal(pis_b, pis)
    for fresh in (fresh_a, fresh_b):
        o = plonk.parse_proof(cd, fresh)
        o["public_inputs"] = pis
        pv.verify(orc, cd, o)
    other_domain, _, _ = agg.aggregate(sigs, seed=100, key_domain=3)
    assert not np.array_equal(other_domain, proof)
    # the same tree through ONE native call (gl355_aggregate_units): byte-identical on the seeded run, with one and with two contexts, and
    # from artifacts persisted to disk and loaded into a fresh Aggregator (no circuit is built, no Python runs between the proofs)
    ctx3 = gl.Context(0)
    for cs in ([ctx], [ctx, ctx3]):
        n_proof, n_pis, n_cd, ms = agg.aggregate_native(sigs, seed=100, ctxs=cs, timed=True)
        assert np.array_equal(n_proof, proof) and np.array_equal(n_pis, pis) and same(n_cd, cd) and len(ms) == 2
    n_dom, _, _ = agg.aggregate_native(sigs, seed=100, key_domain=3)
    assert np.array_equal(n_dom, other_domain)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        agg.save(d)
        agg2 = rec.Aggregator.load(ctx, d)
        l_proof, l_pis, l_cd = agg2.aggregate_native(sigs, seed=100, ctxs=[ctx, ctx3])
        assert np.array_equal(l_proof, proof) and np.array_equal(l_pis, pis) and same(l_cd, cd)
        two, two_pis, _ = agg2.aggregate_native(sigs[:2], seed=100)          # a smaller tree uses the first level only
        assert two_pis.size == 4 + 8 + 8
    fresh_n, pis_n, _ = agg.aggregate_native(sigs)                        # OS-random keys
    assert not np.array_equal(fresh_n, proof) and np.array_equal(pis_n, pis)
    o = plonk.parse_proof(cd, fresh_n)
    o["public_inputs"] = pis
    pv.verify(orc, cd, o)
    bad = (sigs[1][0].copy(), sigs[1][1])
    bad[0][40] ^= np.uint64(1)
    with pytest.raises(gl.Gl355Error) as ei:
        agg.aggregate_native([sigs[0], bad, sigs[2], sigs[3]], seed=1)
    assert ei.value.code == -6                                            # GL355_E_WITNESS: an inner proof does not verify
    ctx3.close()
    outer = plonk.parse_proof(cd, proof)
    outer["public_inputs"] = pis
    pv.verify(orc, cd, outer)
    assert pis.size == 4 + 16 + 16
    assert np.array_equal(pis[:4], aset.tree.cap[0])
    assert np.array_equal(pis[4:20], np.concatenate([s[1][4:8] for s in sigs]))
    assert np.array_equal(pis[20:36], np.concatenate(topics))
    print("aggregation circuits: level-1 degree 2^%d, level-2 degree 2^%d" % (agg.levels[0].data.degree_bits, agg.levels[1].data.degree_bits))
    # a signal against another access set (different root) cannot be aggregated
    aset2, sks2, _ = make_access_set(gl, ctx, 3, 0x605)
    alien, _ = flat_signal(aset2, sks2[1], topics[0], 1, 77)
    with pytest.raises(AssertionError):
        agg.levels[0].witness([sigs[0], alien])
