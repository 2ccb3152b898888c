"""CPU: SURVEY 8(f) N4 -- the restatement of halo2's create_proof (oracle/halo2_model.py) against the independent restatement of its verify_proof
(tests/halo2_verifier.py, SHPLONK check in the exponent under a known tau) over the reference's chip shape
(chip/native_chip/arithmetic_chip.rs:44-160 + poseidon_bn254_chip.rs:27-123 through stark-verifier_amd/halo2_chips.py), tamper cases, Keccak-256
known answers, and the register programs of the descriptor against tree evaluation."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import halo2_model as hm  # noqa: E402
import halo2_verifier as hv  # noqa: E402
from halo2_circuits import oracle_vk_digest  # noqa: E402

h2 = importlib.import_module("stark-verifier_amd.halo2")
ch = importlib.import_module("stark-verifier_amd.halo2_chips")
TAU = 0x1234567890ABCDEF1234567890ABCDEF0123456789ABCDEF % hm.R


def test_keccak256_known_answers():
    for f in (hm.keccak256, hv.keccak256):
        assert f(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
        assert f(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    for n in (1, 55, 135, 136, 137, 272, 1000):
        m = bytes((7 * i + n) & 0xFF for i in range(n))
        assert hm.keccak256(m) == hv.keccak256(m)


def run_program(code, pool, query, fold):
    """reference interpreter of the descriptor's register programs (halo2.ProgramBuilder): -> the folded value"""
    R = h2.R
    regs = [0] * h2.MAX_REGS
    acc = 0

    def val(operand):
        kind, idx = operand >> 24, operand & 0xFFFFFF
        if kind == h2.K_REG:
            return regs[idx]
        if kind == h2.K_CONST:
            return pool[idx]
        return query({h2.K_ADVICE: h2.ADVICE, h2.K_FIXED: h2.FIXED, h2.K_INSTANCE: h2.INSTANCE}[kind], idx)
    for op, dst, a, b in code:
        if op == h2.OP_EMIT:
            acc = (acc * fold + val(a)) % R
        elif op == h2.OP_NEG:
            regs[dst] = (-val(a)) % R
        elif op == h2.OP_MOV:
            regs[dst] = val(a)
        else:
            x, y = val(a), val(b)
            regs[dst] = (x + y) % R if op == h2.OP_ADD else ((x - y) % R if op == h2.OP_SUB else x * y % R)
    return acc


def test_register_programs_equal_tree_evaluation():
    cs, cfg, w = ch.synthetic_circuit(7, table_bits=5, n_permutations=1)
    rng = np.random.default_rng(5)
    vals = {kind: [int.from_bytes(rng.bytes(31), "little") for _ in cs.queries[kind]] for kind in (h2.ADVICE, h2.FIXED, h2.INSTANCE)}
    q = lambda kind, qi: vals[kind][qi]          # noqa: E731
    y = 0x1357924680ACE
    pool = []
    pb = h2.ProgramBuilder(pool)
    for p in cs.all_gate_polys():
        pb.emit(p)
    pb.finish()
    want = 0
    for p in cs.all_gate_polys():
        want = (want * y + p.evaluate(q)) % h2.R
    assert run_program(pb.code, pool, q, y) == want
    assert pb.peak <= h2.MAX_REGS
    # common subexpressions are computed once: the full-round gate's five S-boxes cost 5 x 3 products, not 25 x 3
    n_mul = sum(1 for c in pb.code if c[0] == h2.OP_MUL)
    assert n_mul < 140, n_mul


def prove_and_verify(k, tb, n_perm=1, seed=bytes(range(32))):
    cs, cfg, w = ch.synthetic_circuit(k, table_bits=tb, n_permutations=n_perm)
    params = hm.Params(k, TAU)
    pk = hm.keygen(params, cs, w.fixed_ints(), w.assembly)
    digest = oracle_vk_digest(cs, k, pk)
    proof = hm.create_proof(params, pk, w.advice_ints(), w.instance, seed, digest)
    vk = dict(digest=digest, fixed_commitments=pk.fixed_commitments, sigma_commitments=pk.sigma_commitments)
    return cs, w, params, pk, vk, proof


def test_oracle_proof_verifies_and_tampering_is_rejected():
    k = 7
    cs, w, params, pk, vk, proof = prove_and_verify(k, 5)
    assert hv.verify(k, cs, vk, w.instance, proof, TAU)
    n_points = cs.num_advice + 3 * len(cs.lookups) + 3 + 1 + (cs.degree() - 1)
    rng = np.random.default_rng(9)
    # every region of the proof: commitments, evaluations, the two SHPLONK points
    for pos in [5, 64 * 3 + 40, 64 * n_points + 7, 64 * n_points + 32 * 20 + 31, len(proof) - 100, len(proof) - 1] + [int(v) for v in rng.integers(0, len(proof), 6)]:
        bad = bytearray(proof)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        with pytest.raises(hv.VerifyError):
            hv.verify(k, cs, vk, w.instance, bytes(bad), TAU)
    with pytest.raises(hv.VerifyError):
        hv.verify(k, cs, vk, [[w.instance[0][0] + 1, w.instance[0][1]]], proof, TAU)
    with pytest.raises(hv.VerifyError):
        hv.verify(k, cs, vk, w.instance, proof[:-64], TAU)
    with pytest.raises(hv.VerifyError):
        hv.verify(k, cs, vk, w.instance, proof, TAU + 1)
    # a different seed (other blinding) gives another valid proof; the same seed the same bytes
    cs2, w2, _, _, vk2, proof2 = prove_and_verify(k, 5, seed=bytes(range(1, 33)))
    assert proof2 != proof and hv.verify(k, cs2, vk2, w2.instance, proof2, TAU)


def test_second_circuit_family_verifies():
    """vanilla PLONK gates + a rotation + a two-column lookup with product inputs + an instance column in the permutation (tests/halo2_circuits.py):
    the oracle's proof passes the verifier restatement; other public inputs, a broken copy and a broken lookup do not give an accepted proof"""
    from halo2_circuits import plonk_with_tuple_lookup
    k = 7
    cs, w = plonk_with_tuple_lookup(k, 5)
    assert cs.degree() == 5 and cs.chunk_len() == 3 and len(cs.permutation) == 4 and cs.num_instance == 1
    params = hm.Params(k, TAU)
    pk = hm.keygen(params, cs, w.fixed_ints(), w.assembly)
    digest = oracle_vk_digest(cs, k, pk)
    vk = dict(digest=digest, fixed_commitments=pk.fixed_commitments, sigma_commitments=pk.sigma_commitments)
    proof = hm.create_proof(params, pk, w.advice_ints(), w.instance, bytes(32), digest)
    assert hv.verify(k, cs, vk, w.instance, proof, TAU)
    with pytest.raises(hv.VerifyError):
        hv.verify(k, cs, vk, [[w.instance[0][0] + 1, w.instance[0][1]]], proof, TAU)
    adv = w.advice_ints()
    adv[1][3] += 1                                     # b of row 3 is a copy of c of row 2 (and the gate of row 3 breaks with it)
    with pytest.raises((AssertionError, hv.VerifyError)):
        hv.verify(k, cs, vk, w.instance, hm.create_proof(params, pk, adv, w.instance, bytes(32), digest), TAU)
    adv = w.advice_ints()
    adv[0][4], adv[2][4] = 40, 3 * 40 + 1              # row 4 looks (a, c) up: the gate still holds, 40 is outside the 5-bit table
    with pytest.raises((AssertionError, hv.VerifyError)):
        hv.verify(k, cs, vk, w.instance, hm.create_proof(params, pk, adv, w.instance, bytes(32), digest), TAU)


@pytest.mark.parametrize("seed", range(6))
def test_random_circuits_verify(seed):
    """tests/halo2_circuits.random_circuit: the oracle's proof of a circuit drawn at random passes the verifier restatement; a changed
    result cell does not"""
    from halo2_circuits import random_circuit
    k = 6 + seed % 2
    cs, w = random_circuit(k, seed)
    params = hm.Params(k, TAU)
    pk = hm.keygen(params, cs, w.fixed_ints(), w.assembly)
    digest = oracle_vk_digest(cs, k, pk)
    vk = dict(digest=digest, fixed_commitments=pk.fixed_commitments, sigma_commitments=pk.sigma_commitments)
    proof = hm.create_proof(params, pk, w.advice_ints(), w.instance, bytes(32), digest)
    assert hv.verify(k, cs, vk, w.instance, proof, TAU)
    # the first gate's result column, on a row its selector switches on
    fixed = w.fixed_ints()
    q_col = next(i for i in range(cs.num_fixed) if any(fixed[i][r] == 1 for r in range(2, w.usable - 2)) and set(fixed[i]) <= {0, 1})
    row = next(r for r in range(2, w.usable - 2) if fixed[q_col][r] == 1)
    adv = w.advice_ints()
    noticed = 0
    for c in range(cs.num_advice):
        adv2 = [list(col) for col in adv]
        adv2[c][row] = (adv2[c][row] + 1) % hm.R
        try:
            hv.verify(k, cs, vk, w.instance, hm.create_proof(params, pk, adv2, w.instance, bytes(32), digest), TAU)
        except (AssertionError, hv.VerifyError):          # the quotient has a remainder, or the verifier refuses
            noticed += 1
    assert noticed >= 1, "no single-cell change on row %d was noticed" % row


def test_vk_digest_binds_fixed_values_and_copy_constraints():
    """the transcript's initial scalar is a hash of the PINNED key -- two circuits of one shape that differ only in a table / constant
    or only in their copy constraints start different transcripts (halo2: vk.transcript_repr covers fixed_commitments and the permutation's)"""
    k = 7
    cs, cfg, w = ch.synthetic_circuit(k, table_bits=5, n_permutations=1)
    params = hm.Params(k, TAU)
    fixed = w.fixed_ints()
    pk = hm.keygen(params, cs, fixed, w.assembly)
    d0 = oracle_vk_digest(cs, k, pk)
    assert d0 == oracle_vk_digest(cs, k, hm.keygen(params, cs, w.fixed_ints(), w.assembly))
    fixed2 = [list(col) for col in fixed]
    fixed2[0][5] = (fixed2[0][5] + 1) % hm.R
    assert oracle_vk_digest(cs, k, hm.keygen(params, cs, fixed2, w.assembly)) != d0
    cs3, cfg3, w3 = ch.synthetic_circuit(k, table_bits=5, n_permutations=1)
    cols = [c for c in cs3.permutation][:2]
    w3.assembly.copy(cols[0], 1, cols[1], 2)                       # one more copy constraint, same shape
    assert oracle_vk_digest(cs3, k, hm.keygen(params, cs3, w3.fixed_ints(), w3.assembly)) != d0


def test_unsatisfied_witness_has_no_quotient():
    k = 7
    cs, cfg, w = ch.synthetic_circuit(k, table_bits=5, n_permutations=1)
    params = hm.Params(k, TAU)
    pk = hm.keygen(params, cs, w.fixed_ints(), w.assembly)
    adv = w.advice_ints()
    adv[cfg.arithmetic_config.c.index][3] += 1                 # breaks a b + c = q p + r on row 3
    with pytest.raises(AssertionError):
        hm.create_proof(params, pk, adv, w.instance, bytes(32), oracle_vk_digest(cs, k, pk))


def test_permute_expression_pair_rules():
    A = [3, 1, 3, 3, 2, 1, 0, 0]
    S = [0, 1, 2, 3, 4, 5, 6, 7]
    ap, sp = hm.permute_expression_pair(A, S, 8)
    assert ap == sorted(A)
    assert sorted(sp) == sorted(S)
    for i in range(8):
        assert ap[i] == sp[i] or ap[i] == ap[i - 1]
    assert ap[0] == sp[0]
    with pytest.raises(AssertionError):
        hm.permute_expression_pair([9, 1], [0, 1], 2)
