"""GPU: SURVEY 8(f) N4 -- gl355_plonk_keygen / gl355_plonk_prove (the data-parallel stages of halo2's create_proof with SHPLONK and the Keccak256
transcript, chip/native_chip/test_utils.rs:57-95) against the oracle's restatement (oracle/halo2_model.py): the same commitments of the
verifying key, the same challenges, the same proof BYTES on the same (witness, seed) at k = 7 .. 10 over the reference's chip shape
(arithmetic chip + nine range lookups + BN254-Poseidon chip, halo2_chips.py); proofs accepted by the independent verifier restatement
(tests/halo2_verifier.py, opening check in the exponent under the known tau) incl. at k = 12 where the oracle prover is not run; errors."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import halo2_model as hm  # noqa: E402
import halo2_verifier as hv  # noqa: E402
from halo2_circuits import oracle_vk_digest, plonk_with_tuple_lookup, random_circuit  # noqa: E402

pytestmark = pytest.mark.gpu
h2 = importlib.import_module("stark-verifier_amd.halo2")
ch = importlib.import_module("stark-verifier_amd.halo2_chips")
TAU = 0x1234567890ABCDEF1234567890ABCDEF0123456789ABCDEF % hm.R


def pt(a):
    x, y = h2.from_limbs(a[:4])[0], h2.from_limbs(a[4:])[0]
    return None if (x, y) == (0, 0) else (x, y)


def build(ctx, k, tb, n_perm=1, seed=0x355):
    cs, cfg, w = ch.synthetic_circuit(k, table_bits=tb, n_permutations=n_perm, seed=seed)
    g, gl_ = h2.kzg_setup(ctx, k, TAU)
    prover = h2.PlonkProver(ctx, cs, k, g, gl_, w.fixed, w.assembly.mapping_array())
    return cs, cfg, w, prover


def test_keccak256_entry(ctx):
    import ctypes as C
    for m in (b"", b"abc", bytes(range(256)) * 5, b"z" * 135, b"z" * 136):
        out = C.create_string_buffer(32)
        assert ctx.lib.gl355_keccak256(m, len(m), out) == 0
        assert out.raw == hm.keccak256(m)


@pytest.mark.parametrize("k,tb,tables", [(7, 5, 0), (8, 6, 0), (9, 7, 0), (10, 7, 0), (12, 9, 0), (13, 10, 0), (12, 9, 1), (13, 10, 1)])
def test_proof_bytes_equal_the_oracle(ctx, k, tb, tables, monkeypatch):
    # tables = 1: the key holds the SRS's window multiples and every commitment of more than 20 bits runs on shared buckets (the default from k = 22 on)
    monkeypatch.setenv("GL355_PLONK_MSM_TABLES", str(tables))
    cs, cfg, w, prover = build(ctx, k, tb)
    params = hm.Params(k, TAU)
    pk = hm.keygen(params, cs, w.fixed_ints(), w.assembly)
    # the verifying key's commitments
    assert [pt(c) for c in prover.fixed_commitments] == pk.fixed_commitments
    assert [pt(c) for c in prover.sigma_commitments] == pk.sigma_commitments
    # the transcript's initial scalar, derived by gl355_plonk_keygen from the pinned key == recomputed from the oracle's commitments
    assert prover.digest == oracle_vk_digest(cs, k, pk)
    seed = bytes((7 * i + k) & 0xFF for i in range(32))
    tr = {}
    want = hm.create_proof(params, pk, w.advice_ints(), w.instance, seed, prover.digest, tr)
    got, trace = prover.prove(w.advice, w.instance, seed, want_trace=True)
    for name in ("theta", "beta", "gamma", "y", "x"):
        assert trace[name] == tr[name], name
    for name in ("y", "v", "u"):
        assert trace["shplonk_" + name] == tr["shplonk"][name], name
    assert len(got) == len(want) == prover.info["proof_bytes"]
    if got != want:
        first = next(i for i in range(len(want)) if got[i] != want[i])
        raise AssertionError("proof differs from the oracle's at byte %d of %d" % (first, len(want)))
    vk = dict(digest=prover.digest, fixed_commitments=pk.fixed_commitments, sigma_commitments=pk.sigma_commitments)
    assert hv.verify(k, cs, vk, w.instance, got, TAU)
    # deterministic in (witness, seed); another seed -> other blinding, still valid
    assert prover.prove(w.advice, w.instance, seed) == got
    other = prover.prove(w.advice, w.instance, bytes(32))
    assert other != got and hv.verify(k, cs, vk, w.instance, other, TAU)
    prover.close()


@pytest.mark.parametrize("k,tb", [(7, 5), (9, 6)])
def test_second_circuit_family_bytes_equal_the_oracle(ctx, k, tb):
    cs, w = plonk_with_tuple_lookup(k, tb)
    assert cs.degree() == 5 and len(cs.permutation) == 4 and cs.num_instance == 1 and len(cs.lookups[0][1]) == 2
    g, gl_ = h2.kzg_setup(ctx, k, TAU)
    prover = h2.PlonkProver(ctx, cs, k, g, gl_, w.fixed, w.assembly.mapping_array())
    params = hm.Params(k, TAU)
    pk = hm.keygen(params, cs, w.fixed_ints(), w.assembly)
    assert [pt(c) for c in prover.fixed_commitments] == pk.fixed_commitments
    assert [pt(c) for c in prover.sigma_commitments] == pk.sigma_commitments
    seed = bytes((3 * i + k) & 0xFF for i in range(32))
    want = hm.create_proof(params, pk, w.advice_ints(), w.instance, seed, prover.digest, {})
    got = prover.prove(w.advice, w.instance, seed)
    assert got == want
    vk = dict(digest=prover.digest, fixed_commitments=pk.fixed_commitments, sigma_commitments=pk.sigma_commitments)
    assert hv.verify(k, cs, vk, w.instance, got, TAU)
    with pytest.raises(hv.VerifyError):                              # another public input: refused
        hv.verify(k, cs, vk, [[w.instance[0][0] + 1, w.instance[0][1]]], got, TAU)
    prover.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_circuits_bytes_equal_the_oracle(ctx, seed):
    """circuits drawn at random (tests/halo2_circuits.random_circuit: random gate expressions with rotations -2 .. 2 on advice, fixed and
    instance queries, 0-2 lookups of 1-2 columns with expression inputs and scaled table expressions, copy constraints through advice,
    instance and fixed cells; degree 3 .. 7): verifying-key commitments, proof bytes == the oracle's, accepted by the verifier restatement"""
    k = 6 + seed % 3
    cs, w = random_circuit(k, seed)
    g, gl_ = h2.kzg_setup(ctx, k, TAU)
    prover = h2.PlonkProver(ctx, cs, k, g, gl_, w.fixed, w.assembly.mapping_array())
    params = hm.Params(k, TAU)
    pk = hm.keygen(params, cs, w.fixed_ints(), w.assembly)
    assert [pt(c) for c in prover.fixed_commitments] == pk.fixed_commitments
    assert [pt(c) for c in prover.sigma_commitments] == pk.sigma_commitments
    seed_bytes = bytes((5 * i + seed) & 0xFF for i in range(32))
    want = hm.create_proof(params, pk, w.advice_ints(), w.instance, seed_bytes, prover.digest, {})
    got = prover.prove(w.advice, w.instance, seed_bytes)
    if got != want:
        first = next(i for i in range(min(len(want), len(got))) if got[i] != want[i]) if len(got) == len(want) else -1
        raise AssertionError("seed %d (degree %d, %d lookups): proof differs from the oracle's at byte %d of %d / %d" % (seed, cs.degree(), len(cs.lookups), first, len(got), len(want)))
    vk = dict(digest=prover.digest, fixed_commitments=pk.fixed_commitments, sigma_commitments=pk.sigma_commitments)
    assert hv.verify(k, cs, vk, w.instance, got, TAU)
    prover.close()


def test_k12_proof_passes_the_verifier(ctx):
    k = 12
    cs, cfg, w, prover = build(ctx, k, 9, n_perm=8)
    vk = dict(digest=prover.digest, fixed_commitments=[pt(c) for c in prover.fixed_commitments], sigma_commitments=[pt(c) for c in prover.sigma_commitments])
    proof, ms = prover.prove(w.advice, w.instance, bytes(range(32)), timed=True)
    assert hv.verify(k, cs, vk, w.instance, proof, TAU)
    assert set(ms) == set(h2.STAGES) and all(v >= 0 for v in ms.values())
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 4
    with pytest.raises(hv.VerifyError):
        hv.verify(k, cs, vk, w.instance, bytes(bad), TAU)
    prover.close()


@pytest.mark.parametrize("tables", [0, 1])
def test_k17_reference_table_size(ctx, tables, monkeypatch):
    """k = 17: the smallest circuit that holds the reference's real 16-bit range table and the Goldilocks modulus (arithmetic_chip.rs:19,140-151);
    64 chained Poseidon permutations, every usable row an arithmetic row; the proof passes the verifier restatement -- with and without the
    SRS's window tables in the key"""
    monkeypatch.setenv("GL355_PLONK_MSM_TABLES", str(tables))
    k = 17
    cs, cfg, w, prover = build(ctx, k, 16, n_perm=64)
    assert cfg.arithmetic_config.modulus == ch.GOLDILOCKS_MODULUS and cs.degree() == 6 and len(cs.lookups) == 9 and cs.num_advice == 19
    vk = dict(digest=prover.digest, fixed_commitments=[pt(c) for c in prover.fixed_commitments], sigma_commitments=[pt(c) for c in prover.sigma_commitments])
    proof = prover.prove(w.advice, w.instance, bytes(range(32)))
    assert hv.verify(k, cs, vk, w.instance, proof, TAU)
    prover.close()


def test_k16_short_columns_commit_through_one_window(ctx):
    """k = 16 with a 12-bit range table: the range-limb and permuted-lookup columns are 12-bit values, shorter than the MSM's 14-bit window at 2^16 points, so
    their commitments take the one-window form (a window of 13 bits, no carry window, bn254_msm_bits) -- otherwise reached only at k = 23, where 16-bit columns
    meet 20-bit windows.  The proof passes the verifier restatement, which recomputes nothing from the prover: a wrong commitment fails the opening check."""
    k = 16
    cs, cfg, w, prover = build(ctx, k, 12, n_perm=8)
    vk = dict(digest=prover.digest, fixed_commitments=[pt(c) for c in prover.fixed_commitments], sigma_commitments=[pt(c) for c in prover.sigma_commitments])
    proof = prover.prove(w.advice, w.instance, bytes(range(32)))
    assert hv.verify(k, cs, vk, w.instance, proof, TAU)
    prover.close()


def test_a_lookup_input_outside_the_table_is_an_error(gl, ctx):
    k, tb = 8, 6
    cs, cfg, w, prover = build(ctx, k, tb)
    adv = w.advice.copy()
    adv[cfg.arithmetic_config.r_limbs[0].index, 5, 0] = (1 << tb) + 3          # not a tb-bit value
    with pytest.raises(gl.Gl355Error) as ei:
        prover.prove(adv, w.instance, bytes(32))
    assert ei.value.code == -1 and "lookup" in str(ei.value)
    prover.close()


def test_commit_columns_entry(ctx):
    """gl355_kzg_commit_columns == one gl355_kzg_commit per column"""
    k, cols = 10, 5
    g, gl_ = h2.kzg_setup(ctx, k, TAU)
    rng = np.random.default_rng(3)
    vals = rng.integers(0, 1 << 62, (cols, 1 << k, 4), dtype=np.uint64)
    vals[:, :, 3] >>= np.uint64(4)
    out = np.zeros((cols, 8), dtype=np.uint64)
    ctx.check(ctx.lib.gl355_kzg_commit_columns(ctx.h, gl_.ctypes.data, vals.ctypes.data, k, cols, out.ctypes.data))
    for c in range(cols):
        one = np.zeros(8, dtype=np.uint64)
        ctx.check(ctx.lib.gl355_kzg_commit(ctx.h, gl_.ctypes.data, vals[c].ctypes.data, k, 0, one.ctypes.data))
        assert np.array_equal(out[c], one)


def test_malformed_descriptors_are_refused(gl, ctx):
    """gl355_plonk_keygen on untrusted descriptors: wrong magic, truncation, a program operand out of range, a permutation column that is never
    queried, an implausible shape -- an error code every time, never a crash"""
    import ctypes as C
    k = 7
    cs, cfg, w = ch.synthetic_circuit(k, table_bits=5, n_permutations=1)
    g, gl_ = h2.kzg_setup(ctx, k, TAU)
    desc = h2.export_desc(cs, k, 1)
    fixed = np.ascontiguousarray(w.fixed)
    mapping = np.ascontiguousarray(w.assembly.mapping_array())

    def keygen(d):
        h = C.c_void_p()
        rc = ctx.lib.gl355_plonk_keygen(ctx.h, d.ctypes.data, d.size, g.ctypes.data, gl_.ctypes.data, fixed.ctypes.data, mapping.ctypes.data, C.byref(h))
        if rc == 0:
            ctx.lib.gl355_plonk_pk_destroy(h)
        return rc
    assert keygen(desc) == 0
    bad = desc.copy(); bad[0] ^= np.uint64(1)
    assert keygen(bad) == -1
    assert keygen(desc[:-3].copy()) == -1
    assert keygen(desc[:20].copy()) == -1
    bad = desc.copy(); bad[2] = 40                                   # k
    assert keygen(bad) == -1
    bad = desc.copy(); bad[8] = 2                                    # degree
    assert keygen(bad) == -1
    # the gate program starts after header (24) + permutation columns + queries + constants: poison one operand
    off = 24 + int(desc[6]) + int(desc[10]) + int(desc[11]) + int(desc[12]) + 4 * int(desc[13])
    bad = desc.copy()
    words = bad[off:off + 2].view(np.uint32)
    words[2] = (2 << 24) | 0xFFFF                                    # advice query 65535
    assert keygen(bad) == -1
    bad = desc.copy()
    bad[off:off + 2].view(np.uint32)[1] = 99                         # destination register 99
    assert keygen(bad) == -1
    bad = desc.copy(); bad[24] = (np.uint64(7) << np.uint64(32))     # permutation column of kind 7
    assert keygen(bad) == -1
    m2 = mapping.copy(); m2[0, 0, 1] = 1 << 20                       # a sigma pointing outside the domain
    h = C.c_void_p()
    assert ctx.lib.gl355_plonk_keygen(ctx.h, desc.ctypes.data, desc.size, g.ctypes.data, gl_.ctypes.data, fixed.ctypes.data, m2.ctypes.data, C.byref(h)) == -1
    # prove with a key of another context / too small a buffer
    prover = h2.PlonkProver(ctx, cs, k, g, gl_, w.fixed, mapping)
    ctx2 = gl.Context(0)
    buf = np.zeros(prover.info["proof_bytes"], dtype=np.uint8)
    n_out = C.c_uint64(0)
    lens = np.array([2, 0], dtype=np.uint32)
    inst = h2.to_limbs(w.instance[0])
    args = (w.advice.ctypes.data, inst.ctypes.data, lens.ctypes.data, bytes(32))
    assert ctx2.lib.gl355_plonk_prove(ctx2.h, prover.h, *args, buf.ctypes.data, buf.size, C.byref(n_out), None, None) == -1
    assert ctx.lib.gl355_plonk_prove(ctx.h, prover.h, *args, buf.ctypes.data, 100, C.byref(n_out), None, None) == -1
    assert n_out.value == prover.info["proof_bytes"]
    ctx2.close()
    prover.close()


def test_digest_binds_the_fixed_columns(ctx):
    """same shape, one fixed cell changed (a cell no constraint reads with the witness used: the proof still verifies) -> another
    verifying-key digest, hence another theta: the key is bound into every challenge"""
    k = 7
    cs, cfg, w = ch.synthetic_circuit(k, table_bits=5, n_permutations=1)
    g, gl_ = h2.kzg_setup(ctx, k, TAU)
    mapping = w.assembly.mapping_array()
    p0 = h2.PlonkProver(ctx, cs, k, g, gl_, w.fixed, mapping)
    fixed2 = w.fixed.copy()
    row = w.usable + 1                                  # a blinding row: outside every gate's active rows
    fixed2[0, row, 0] ^= np.uint64(1)
    p1 = h2.PlonkProver(ctx, cs, k, g, gl_, fixed2, mapping)
    assert p0.digest != p1.digest
    seed = bytes(range(32))
    _, t0 = p0.prove(w.advice, w.instance, seed, want_trace=True)
    _, t1 = p1.prove(w.advice, w.instance, seed, want_trace=True)
    assert t0["theta"] != t1["theta"]
    # an explicit digest in the descriptor is used as given, and gl355_plonk_pk_set_digest overrides
    p2 = h2.PlonkProver(ctx, cs, k, g, gl_, w.fixed, mapping, digest=12345)
    assert p2.digest == 12345
    for p in (p0, p1, p2):
        p.close()


def test_coset_quotient_equals_the_extended_domain_quotient(ctx):
    """The one algorithmic departure from halo2: evaluate_h runs on degree - 1 cosets of size 2^k and recovers the pieces h_p by a
    Vandermonde solve; halo2 (and oracle/halo2_model.py) divides on the extended domain of 2^extended_k points and splits the coefficient vector.
    gl355_plonk_pk_export_quotient hands out the prover's pieces: they equal the oracle's coefficient for coefficient (k = 10, 5 pieces)."""
    k, tb = 10, 7
    cs, cfg, w, prover = build(ctx, k, tb)
    params = hm.Params(k, TAU)
    pk = hm.keygen(params, cs, w.fixed_ints(), w.assembly)
    seed = bytes((11 * i + 5) & 0xFF for i in range(32))
    tr = {}
    want = hm.create_proof(params, pk, w.advice_ints(), w.instance, seed, prover.digest, tr)
    P, n = prover.info["n_pieces"], 1 << k
    assert P == cs.degree() - 1 == len(tr["h_pieces"]) and prover.info["extended_k"] > k
    out = np.zeros((P, n, 4), dtype=np.uint64)
    ctx.check(ctx.lib.gl355_plonk_pk_export_quotient(prover.h, out.ctypes.data))
    got = prover.prove(w.advice, w.instance, seed)
    assert got == want
    for p in range(P):
        mine = h2.from_limbs(out[p])
        assert mine == [int(v) for v in tr["h_pieces"][p]], "quotient piece %d differs from the extended-domain quotient" % p
    assert any(any(tr["h_pieces"][p]) for p in range(P))
    # the hook is one-shot: the next proof leaves the buffer alone
    out[:] = 0
    prover.prove(w.advice, w.instance, seed)
    assert not out.any()
    prover.close()


def test_reference_shape_k20_is_accepted_by_the_verifier_restatement(ctx):
    """k = 20 on the reference's chip shape (16-bit range table, 64 chained BN254-Poseidon permutations), SRS and witness resident on the device:
    three proofs under different blinding seeds pass tests/halo2_verifier.py (was tools/halo2_verify_many.py, outside pytest: VERDICT r4 #5)"""
    import torch
    k = 20
    cs, cfg, w = ch.synthetic_circuit(k, table_bits=16, n_permutations=64, seed=0x355 + k)
    n = 1 << k
    g = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    gl_ = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    tau = h2.to_limbs([TAU])[0]
    ctx.check(ctx.lib.gl355_kzg_setup(ctx.h, tau.ctypes.data, k, g.data_ptr(), gl_.data_ptr()))
    prover = h2.PlonkProver(ctx, cs, k, g.data_ptr(), gl_.data_ptr(), w.fixed, w.assembly.mapping_array())
    vk = dict(digest=prover.digest, fixed_commitments=[pt(c) for c in prover.fixed_commitments], sigma_commitments=[pt(c) for c in prover.sigma_commitments])
    adv = torch.from_numpy(w.advice.view(np.int64)).cuda()
    proofs = set()
    for s in range(3):
        proof = prover.prove(adv.data_ptr(), w.instance, bytes([17 * s + 3]) * 32)
        assert hv.verify(k, cs, vk, w.instance, proof, TAU % h2.R)
        proofs.add(proof)
    assert len(proofs) == 3
    bad = bytearray(proof)
    bad[100] ^= 4
    with pytest.raises(hv.VerifyError):
        hv.verify(k, cs, vk, w.instance, bytes(bad), TAU % h2.R)
    prover.close()
    del adv, g, gl_
    torch.cuda.empty_cache()
