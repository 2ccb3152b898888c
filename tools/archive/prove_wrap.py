#!/usr/bin/env python3
"""Timing of the final wrap proof (wrapper.rs:35-56: recursive circuit proven under the BN254-Poseidon hasher) on one GPU."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.init()
gl = importlib.import_module("stark-verifier_amd")
sem = importlib.import_module("stark-verifier_amd.semaphore")
rec = importlib.import_module("stark-verifier_amd.recursion")
plonk = importlib.import_module("stark-verifier_amd.plonk")
from oracle_lib import rand_field
log_members = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = gl.Context(0)
rng = np.random.default_rng(0x357)
sks = rand_field(rng, (1 << log_members, 4))
keys = ctx.hash_no_pad(np.concatenate([sks, np.zeros_like(sks)], axis=1))
aset = sem.AccessSet(ctx, keys)
topic = rand_field(rng, 4)
sig, data = aset.make_signal_fast(sks[0], topic, 0, 1, flat_only=True)
inner = (sig.proof, np.concatenate([aset.tree.cap[0], sig.nullifier[0], topic]))
t0 = time.perf_counter()
wc = rec.WrapperCircuit(ctx, data.common()).build([inner], rng)
print("wrap circuit build: %.2f s, degree 2^%d" % (time.perf_counter() - t0, wc.data.degree_bits))
rows, pis = wc.witness([inner])
for k in range(3):
    ctx.profile_enable(True); ctx.profile_read()
    t0 = time.perf_counter()
    flat = plonk.prove_sparse(ctx, wc.data, wc.row_idx, rows, pis, 5 + k, flat_only=True)
    dt = time.perf_counter() - t0
    prof = ctx.profile_read(); ctx.profile_enable(False)
    print("wrap proof %d: %.1f ms" % (k, dt * 1e3))
for name, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]:
    print("  %-28s launches %4d  %8.2f ms" % (name, v[0], v[1]))
