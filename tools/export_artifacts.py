#!/usr/bin/env python3
"""Build the Semaphore circuit of a given tree height and the recursive verifier circuit over it (once per shape) and write them
as circuit artifacts: <dir>/semaphore.gl355, <dir>/recursive.gl355 (format: include/gl355.h).  Needs a GPU (the preprocessed
commitment enters the circuit digest)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
gl = importlib.import_module("stark-verifier_amd")
sem = importlib.import_module("stark-verifier_amd.semaphore")
rec = importlib.import_module("stark-verifier_amd.recursion")
out = sys.argv[1]
log_members = int(sys.argv[2]) if len(sys.argv) > 2 else 20
os.makedirs(out, exist_ok=True)
ctx = gl.Context(0)
rng = np.random.default_rng(1)
sks = rng.integers(0, 1 << 62, size=(1 << log_members, 4), dtype=np.uint64)
keys = ctx.hash_no_pad(np.concatenate([sks, np.zeros_like(sks)], axis=1))
aset = sem.AccessSet(ctx, keys)
data, rows = aset.build(rng)
topic = rng.integers(0, 1 << 62, size=4, dtype=np.uint64)
idx, vals, pi = aset.witness_rows(rows, sks[0], topic, 0)
blob = data.export_blob(idx)
blob.tofile(os.path.join(out, "semaphore.gl355"))
sig, _ = aset.make_signal_fast(sks[0], topic, 0, 1, flat_only=True)
rc = rec.RecursiveCircuit(ctx, data.common(), k=1).build([(sig.proof, pi)], rng)
rblob = rc.data.export_blob(rc.row_idx, rc.tape, rc.pi_pos, rc.n_inputs, rc.tape_layout)
rblob.tofile(os.path.join(out, "recursive.gl355"))
print("semaphore.gl355: %.1f MB (degree 2^%d), recursive.gl355: %.1f MB (degree 2^%d, %d tape entries)" % (
    blob.nbytes / 1e6, data.degree_bits, rblob.nbytes / 1e6, rc.data.degree_bits, rc.tape.shape[0]))
