#!/usr/bin/env python3
"""A few Semaphore proofs on one context (target for rocprofv3 --kernel-trace --stats)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
gl = importlib.import_module("stark-verifier_amd")
sem = importlib.import_module("stark-verifier_amd.semaphore")
from oracle_lib import rand_field
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = gl.Context(0)
rng = np.random.default_rng(0x357)
sks = rand_field(rng, (1 << 20, 4))
keys = ctx.hash_no_pad(np.concatenate([sks, np.zeros_like(sks)], axis=1))
aset = sem.AccessSet(ctx, keys)
aset.build(np.random.default_rng(1))
topic = rand_field(rng, 4)
for k in range(n):
    aset.make_signal_fast(sks[k], topic, k, k, flat_only=True)
