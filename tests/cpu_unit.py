"""One unit of BASELINE configs[3] entirely on the CPU side (test helper): a Semaphore signal of a 2^log_members access set and
the recursive proof that verifies it (wrapper.rs:35-56 over the Poseidon-Goldilocks config), circuit tables from the product's
host-only builders, every proof from the CPU restatement of prove() (oracle/gl_prover.c).  Used to mint / check
tests/golden/unit_depth20.json and by the GPU suite, which must reproduce both proofs byte for byte."""
import ctypes as C
import importlib

import numpy as np

import cpu_semaphore as cs
from oracle_lib import CpuProver, rand_field

UNIT_CASE = dict(log_members=20, seed=0x357, member=12, key_sem=0x358, key_rec=0x359)      # signal.rs:42: signer index 12


def recursive_cpu_circuit(orc, inner_common, flat, pi, k=1):
    """the verify_proof circuit of `k` inner proofs (first proof set fixes the layout), its CPU prover and its witness tape"""
    gad = importlib.import_module("stark-verifier_amd.gadgets")
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    rec = importlib.import_module("stark-verifier_amd.recursion")
    proofs = [(flat, pi)] if k == 1 else list(zip(flat, pi))
    b = gad.GadgetBuilder()
    tagged, off = [], 0
    for f, p in proofs:
        tagged.append(plonk.parse_proof_tagged(inner_common, f, p, off))
        off += len(f) + len(p)
    inner_pis = [rec.verify_proof(b, inner_common, t, register_pis=False) for t in tagged]
    (rec.wrap_public_inputs if k == 1 else rec.aggregate_public_inputs)(b, inner_pis)
    b.finalize_public_inputs()
    data = b.cb.layout()
    cpu = CpuProver.from_circuit_data(orc, data)
    data.set_digest(cpu.cap())
    for i in range(4):
        cpu.pd.circuit_digest[i] = int(data.circuit_digest[i])
    tape, row_idx, pi_pos = b.witness_tape()
    return dict(data=data, cpu=cpu, tape=tape, row_idx=row_idx, pi_pos=pi_pos, n_inputs=off)


def replay(rc, inputs):
    lib = importlib.import_module("stark-verifier_amd._lib").load()
    rows = np.empty((rc["row_idx"].size, 135), dtype=np.uint64)
    inputs = np.ascontiguousarray(inputs, dtype=np.uint64)
    failed = C.c_uint64(0)
    r = lib.gl355_witness_replay(rc["tape"].ctypes.data, rc["tape"].shape[0], inputs.ctypes.data, inputs.size, rows.ctypes.data, rows.size, 135,
                                 C.byref(failed))
    assert r == 0, (r, failed.value)
    return rows, rows.reshape(-1)[rc["pi_pos"]]


def cpu_unit(orc, u=UNIT_CASE):
    """-> (case, topic, semaphore proof, its public inputs, recursive-circuit dict, recursive proof, its public inputs)"""
    case = cs.build_case(orc, u["log_members"], u["seed"])
    topic = rand_field(case["rng"], 4)
    idx, vals, pi = cs.witness(orc, case, u["member"], topic)
    flat = case["cpu"].prove_sparse(idx, vals, pi, u["key_sem"])
    rc = recursive_cpu_circuit(orc, case["data"].common(), flat, pi)
    rows, opis = replay(rc, np.concatenate([flat, pi]))
    outer = rc["cpu"].prove_sparse(rc["row_idx"], rows, opis, u["key_rec"])
    return case, topic, flat, pi, rc, outer, opis
