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
    ctx.set_option(3, 4)                                  # GL355_OPT_REPLAY_THREADS: the query rounds of the tape on 4 host threads
    flat_mt, pis_mt = nat.prove_tape(ctx, np.concatenate([inner[0], inner[1]]), 44)
    ctx.set_option(3, 1)
    assert np.array_equal(flat_mt, flat_py) and np.array_equal(pis_mt, pis_py)
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
            b[112 + 12345] ^= np.uint64(1)            # a selector / constant value: the digest no longer matches
        elif mutate == "digest":
            b[107] ^= np.uint64(1)
        elif mutate == "rowidx":
            b[b.size - 1] = np.uint64(1 << 40)                        # last row index (no pi positions, no tape in this artifact)
        with pytest.raises(gl.Gl355Error):
            plonk.NativeCircuit(ctx, b)


def test_native_batch_runtime(gl, ctx, orc):
    """gl355_semaphore_units: every unit's proofs equal the ones made call by call (per-unit keys = gl355_derive_key of the batch key), the leaves are the
    re-exposed nullifier | topic, bad member indices are refused"""
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    rec = importlib.import_module("stark-verifier_amd.recursion")
    aset, sks, rng = make_access_set(gl, ctx, 4, 0x903)
    topic = rand_field(rng, 4)
    data, rows = aset.build(None)
    idx, vals, pi = aset.witness_rows(rows, sks[0], topic, 0)
    sem = plonk.NativeCircuit(ctx, data.export_blob(idx))
    flat0, pis0 = sem.semaphore_prove(ctx, sks[0], topic, 0, aset.tree.prove_host(0), 1)
    rc = rec.RecursiveCircuit(ctx, data.common(), k=1).build([(flat0, pis0)], rng)
    nat = rc.native()
    ctxs = [gl.Context(0) for _ in range(3)]
    members = np.array([3, 9, 0, 15, 7], dtype=np.uint64)
    leaves, proofs, per = plonk.semaphore_units(ctxs, sem, nat, sks, topic, aset.tree.digests, members, 1000, want_proofs=True)
    assert sum(per) == members.size and all(0 <= k <= members.size for k in per)      # units are handed out one at a time
    for j, m in enumerate(members):
        f, p = sem.semaphore_prove(ctx, sks[m], topic, int(m), aset.tree.prove_host(int(m)), plonk.derive_key(1000, 2 * j))
        o, op = nat.prove_tape(ctx, np.concatenate([f, p]), plonk.derive_key(1000, 2 * j + 1))
        assert np.array_equal(proofs[j], o) and np.array_equal(leaves[j], op[4:12])
        assert np.array_equal(leaves[j, :4], orc.hash_no_pad(np.concatenate([sks[m], topic]))) and np.array_equal(leaves[j, 4:], topic)
    # signals only (no verifier circuit)
    leaves2, proofs2, _ = plonk.semaphore_units(ctxs, sem, None, sks, topic, aset.tree.digests, members[:2], 1000, want_proofs=True)
    assert np.array_equal(leaves2, leaves[:2]) and proofs2.shape[1] == sem.proof_words
    # a key that is not in the tree still yields valid proofs -- of ANOTHER root (the verifier compares the public root with the
    # access set, access_set.rs:33-41); only that unit's leaf changes
    bad_keys = sks.copy()
    bad_keys[9] = rand_field(rng, 4)
    leaves3, _, _ = plonk.semaphore_units(ctxs, sem, nat, bad_keys, topic, aset.tree.digests, members, 1000)
    assert not np.array_equal(leaves3[1], leaves[1]) and np.array_equal(np.delete(leaves3, 1, 0), np.delete(leaves, 1, 0))
    with pytest.raises(gl.Gl355Error):
        plonk.semaphore_units(ctxs, sem, nat, sks, topic, aset.tree.digests, np.array([99], dtype=np.uint64), 1)
    for c in ctxs:
        c.close()


def test_cpp_host_program(tmp_path):
    """examples/native_units.cpp: a plain C++ host (g++, no scripting layer) that loads exported circuit artifacts and proves
    signals + recursive proofs through the C ABI alone"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    art = str(tmp_path / "art")
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "export_artifacts.py"), art, "3"], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    assert os.path.getsize(os.path.join(art, "recursive.gl355")) > 1 << 20
    subprocess.check_call(["make", "-C", os.path.join(root, "examples")], stdout=subprocess.DEVNULL)
    out = subprocess.check_output([os.path.join(root, "examples", "native_units"), art, "3", "2", "6"], text=True)
    assert "6 units" in out and "aggregation root" in out, out
    # a truncated artifact is refused
    blob = np.fromfile(os.path.join(art, "semaphore.gl355"), dtype=np.uint64)
    blob[:-5].tofile(os.path.join(art, "semaphore.gl355"))
    assert subprocess.call([os.path.join(root, "examples", "native_units"), art, "3", "2", "6"], stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL) != 0


def test_runtime_config_sleeping_waits():
    """gl355_runtime_config has to run before the HIP runtime initialises, so it is exercised in a fresh process: after it, a
    device wait costs (almost) no CPU time, and the results are the usual ones"""
    import subprocess
    import sys
    code = r'''
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, %r)
import numpy as np
import torch                                   # imported (its libamdhip64 is the process's HIP runtime) but not initialised
lib = importlib.import_module("stark-verifier_amd._lib").load(init_torch=False)
assert lib.gl355_runtime_config(0, 8, 1) == 0
getenv = C.CDLL(None).getenv; getenv.restype = C.c_char_p
assert getenv(b"GPU_MAX_HW_QUEUES") == b"16"      # two queues per context: proving stream + side stream       # set in the C environment (os.environ is Python's own copy)
gl = importlib.import_module("stark-verifier_amd")
ctx = gl.Context(0)
x = torch.arange(12 << 22, dtype=torch.int64, device="cuda").reshape(-1, 12)
ref = ctx.poseidon_permute(np.arange(24, dtype=np.uint64).reshape(2, 12))
for rep in range(2):
    w0, c0 = time.perf_counter(), time.process_time()
    for _ in range(4):
        ctx.check(lib.gl355_poseidon_permute(ctx.h, C.c_void_p(x.data_ptr()), x.shape[0]))
    ctx.sync()
    wall, cpu = time.perf_counter() - w0, time.process_time() - c0
print("RESULT", wall, cpu, int(ref[1, 0]))
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}      # tests/conftest.py sets it for this process
    out = subprocess.run([sys.executable, "-c", code % root], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    wall, cpu, v = out.stdout.split("RESULT")[1].split()
    assert float(cpu) < 0.5 * float(wall), (wall, cpu)          # a spinning wait would make them equal


def test_artifact_with_an_external_digest(gl, ctx, orc):
    """version-3 artifacts: the circuit digest comes from elsewhere (plonky2's own build), the loader ties the tables to it through
    the constants_sigmas cap; the transcript -- and therefore prover and verifier -- use the given digest"""
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    aset, sks, rng = make_access_set(gl, ctx, 3, 0x904)
    topic = rand_field(rng, 4)
    data, rows = aset.build(None)
    idx, vals, pi = aset.witness_rows(rows, sks[2], topic, 2)
    ext = rand_field(rng, 4)
    blob = data.export_blob(idx, external_digest=ext)
    nat = plonk.NativeCircuit(ctx, blob)
    flat, pis = nat.semaphore_prove(ctx, sks[2], topic, 2, aset.tree.prove_host(2), 5)
    p = np.ascontiguousarray(pis, dtype=np.uint64)
    assert ctx.lib.gl355_circuit_verify(nat.h, flat.ctypes.data, flat.size, p.ctypes.data, p.size) == 0
    own = plonk.NativeCircuit(ctx, data.export_blob(idx))           # the same circuit under this framework's digest: another transcript
    assert ctx.lib.gl355_circuit_verify(own.h, flat.ctypes.data, flat.size, p.ctypes.data, p.size) == -7
    bad = blob.copy()
    bad[200] ^= np.uint64(1)                                          # a table word: the commitment no longer matches the carried cap
    with pytest.raises(gl.Gl355Error):
        plonk.NativeCircuit(ctx, bad)
    bad = blob.copy()
    bad[-1] ^= np.uint64(1)                                           # the carried cap itself
    with pytest.raises(gl.Gl355Error):
        plonk.NativeCircuit(ctx, bad)


def test_artifact_in_plonky2_gate_order_with_an_external_digest(gl, ctx, orc):
    """What a plonky2-side exporter hands over: tables in upstream's gate order and selector grouping, and upstream's
    own circuit digest.  The loader ties the tables to the digest through the constants_sigmas cap, the GPU proves from the artifact alone,
    and the proof passes gl355_circuit_verify and the restated reference verifier run on the same verifier data"""
    import plonk_verifier as pv
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    aset, sks, rng = make_access_set(gl, ctx, 3, 0x905)
    topic = rand_field(rng, 4)
    data, rows = aset.build(None, gate_order="plonky2")
    degs = [plonk._GATE_DEGREE[t](p) for t, p in data.gates]
    assert degs == sorted(degs) and len(data.groups) >= 2
    idx, vals, pi = aset.witness_rows(rows, sks[5], topic, 5)
    ext = rand_field(rng, 4)                                             # stands for plonky2's circuit_digest (covers its domain separator too)
    nat = plonk.NativeCircuit(ctx, data.export_blob(idx, external_digest=ext))
    flat, pis = nat.semaphore_prove(ctx, sks[5], topic, 5, aset.tree.prove_host(5), 9)
    p = np.ascontiguousarray(pis, dtype=np.uint64)
    assert np.array_equal(p, pi)
    assert ctx.lib.gl355_circuit_verify(nat.h, flat.ctypes.data, flat.size, p.ctypes.data, p.size) == 0
    data.circuit_digest = ext                                            # the verifier data of the exported circuit
    proof = plonk.parse_proof(data, flat)
    proof["public_inputs"] = p
    pv.verify(orc, data.common(), proof)
    # the same circuit in this framework's own order is another artifact: its verifier refuses the proof
    aset2, _, _ = make_access_set(gl, ctx, 3, 0x905)
    d2, _ = aset2.build(None)
    own = plonk.NativeCircuit(ctx, d2.export_blob(idx))
    assert ctx.lib.gl355_circuit_verify(own.h, flat.ctypes.data, flat.size, p.ctypes.data, p.size) == -7
