"""CPU: the oracle's prove() (oracle/gl_prover.c) against the restatement of the REFERENCE's in-tree verifier
(tests/plonk_verifier.py, src/plonky2_verifier/chip/**) and against the committed golden proof digest.  The GPU suite checks
the product's proofs byte-for-byte against this same prover (tests/test_gpu_cpu_prover.py)."""
import json
import os

import numpy as np
import pytest

import cpu_semaphore as cs
import plonk_verifier as pv
from oracle_lib import rand_field

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "semaphore_proof.json")


def test_cpu_semaphore_proof_verifies_and_matches_golden(orc):
    case, topic, (idx, vals, pi), flat = cs.golden_proof(orc)
    plonk, data = case["plonk"], case["data"]
    # public inputs root | nullifier | topic (access_set.rs:33-41)
    member = cs.GOLDEN_CASE["member"]
    assert np.array_equal(pi[:4], case["root"]) and np.array_equal(pi[8:], topic)
    assert np.array_equal(pi[4:8], orc.hash_no_pad(np.concatenate([case["sks"][member], topic])))
    assert np.array_equal(data.circuit_digest, orc.hash_no_pad(np.concatenate([
        case["cpu"].cap().reshape(-1), np.array([data.degree_bits, len(data.gates), data.num_selectors] + [(t << 32) | p for t, p in data.gates], dtype=np.uint64)])))
    proof = plonk.parse_proof(data, flat)
    proof["public_inputs"] = pi
    ch = pv.verify(orc, data.common(), proof)
    assert len(ch["query_indices"]) == 28
    golden = json.load(open(GOLDEN))
    assert golden["case"] == cs.GOLDEN_CASE
    assert cs.digest_of(flat) == golden["sha256"] and int(flat.size) == golden["words"]
    assert [int(x) for x in pi] == [int(x, 16) for x in golden["public_inputs"]]
    # same (witness, seed) -> same bytes; another seed -> other blinding / salt, still valid
    assert np.array_equal(case["cpu"].prove_sparse(idx, vals, pi, cs.GOLDEN_CASE["proof_seed"]), flat)
    other = case["cpu"].prove_sparse(idx, vals, pi, 100)
    assert not np.array_equal(other, flat)
    p2 = plonk.parse_proof(data, other)
    p2["public_inputs"] = pi
    pv.verify(orc, data.common(), p2)


def test_cpu_prover_rejects_bad_witness(orc):
    case = cs.build_case(orc, 2, 0x7E58)
    topic = rand_field(case["rng"], 4)
    idx, vals, pi = cs.witness(orc, case, 1, topic)
    plonk, data = case["plonk"], case["data"]
    bad = vals.copy()
    bad[5, 70] ^= np.uint64(1)                    # an S-box wire of the first Merkle level
    proof = plonk.parse_proof(data, case["cpu"].prove_sparse(idx, bad, pi, 1))
    proof["public_inputs"] = pi
    with pytest.raises(pv.VerifyError):
        pv.verify(orc, data.common(), proof)
    # wrong public input (nullifier of another topic) with an otherwise valid witness
    proof = plonk.parse_proof(data, case["cpu"].prove_sparse(idx, vals, pi, 1))
    pi_bad = pi.copy()
    pi_bad[5] ^= np.uint64(2)
    proof["public_inputs"] = pi_bad
    with pytest.raises(pv.VerifyError):
        pv.verify(orc, data.common(), proof)
