"""A Semaphore proof entirely on the CPU side (test helper): the circuit tables come from the product's host-only
CircuitBuilder.layout(), the access-set tree, the preprocessed commitment and the proof from the oracle.  Used by the CPU
suite (oracle prover vs. the restated reference verifier) and to mint / check tests/golden/semaphore_proof.json."""
import ctypes as C
import hashlib
import importlib

import numpy as np

from oracle_lib import CpuProver, rand_field

GOLDEN_CASE = dict(log_members=2, seed=0x7E57, member=3, proof_seed=99)
# the largest group of the reference's own sweep (access_set.rs:193-215: groups 2^20 .. 2^25, signer index 12)
GROUP25_CASE = dict(log_members=25, seed=0x25D, member=12, proof_seed=0x25E)


def build_case(orc, log_members, seed, config=None, gate_order="own"):
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    sem = importlib.import_module("stark-verifier_amd.semaphore")
    rng = np.random.default_rng(seed)
    sks = rand_field(rng, (1 << log_members, 4))
    # public keys = hash_no_pad(sk | 0^4) (signal.rs:32-39): the all-cap "tree" over the 8-element leaves is exactly that batch of hashes
    keys = orc.merkle_build(np.concatenate([sks, np.zeros_like(sks)], axis=1), log_members)[1]
    digests, cap = orc.merkle_build(keys, 0)
    builder = plonk.CircuitBuilder(config or plonk.CircuitConfig(), gate_order=gate_order)
    rows = sem.semaphore_circuit(builder, log_members)
    data = builder.layout()
    cpu = CpuProver.from_circuit_data(orc, data)
    data.set_digest(cpu.cap())
    for i in range(4):
        cpu.pd.circuit_digest[i] = int(data.circuit_digest[i])
    return dict(sks=sks, keys=keys, digests=digests, root=cap[0], data=data, rows=rows, cpu=cpu, rng=rng, plonk=plonk)


def witness(orc, case, member, topic):
    """gl355_semaphore_witness (host-only C) on the member's Merkle path -> (row_idx, rows, public inputs)"""
    lib = importlib.import_module("stark-verifier_amd._lib").load()
    h = int(case["keys"].shape[0]).bit_length() - 1
    sib = np.ascontiguousarray(orc.merkle_prove(case["digests"], case["keys"].shape[0], 0, member), dtype=np.uint64)
    vals = np.empty((h + 7, 135), dtype=np.uint64)
    pi = np.empty(12, dtype=np.uint64)
    sk, tp = np.ascontiguousarray(case["sks"][member]), np.ascontiguousarray(topic, dtype=np.uint64)
    rc = lib.gl355_semaphore_witness(sk.ctypes.data, tp.ctypes.data, int(member), sib.ctypes.data, h, vals.ctypes.data, pi.ctypes.data)
    assert rc == 0
    r = case["rows"]
    idx = np.array([r["pi"], r["h1"], r["h2"], r["bits"], r["leaf"]] + list(r["m"]) + [r["null"], r["zero"]], dtype=np.uint32)
    return idx, vals, pi


def golden_proof(orc):
    g = GOLDEN_CASE
    case = build_case(orc, g["log_members"], g["seed"])
    topic = rand_field(case["rng"], 4)
    idx, vals, pi = witness(orc, case, g["member"], topic)
    flat = case["cpu"].prove_sparse(idx, vals, pi, g["proof_seed"])
    return case, topic, (idx, vals, pi), flat


def digest_of(flat):
    return hashlib.sha256(np.ascontiguousarray(flat, dtype="<u8").tobytes()).hexdigest()
