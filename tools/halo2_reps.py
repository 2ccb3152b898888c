"""k = 23 proofs of the reference-shaped synthetic circuit, one line per proof with the stage times: run-to-run and seed-to-seed spread.
usage: python tools/halo2_reps.py [seed byte, repeated]  (default: seeds 0..7 once each)"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

torch.cuda.init()
gl = importlib.import_module("stark-verifier_amd")
import halo2_bench as hb

h2 = importlib.import_module("stark-verifier_amd.halo2")
ch = importlib.import_module("stark-verifier_amd.halo2_chips")
k = int(os.environ.get("GL355_H2_K", "23"))
ctx = gl.Context(0)
cs, cfg, w = ch.synthetic_circuit(k, table_bits=min(16, k - 1), n_permutations=64)
n = 1 << k
g = torch.empty((n, 8), dtype=torch.int64, device="cuda")
gl_ = torch.empty((n, 8), dtype=torch.int64, device="cuda")
tau = h2.to_limbs([hb.TAU])[0]
ctx.check(ctx.lib.gl355_kzg_setup(ctx.h, tau.ctypes.data, k, g.data_ptr(), gl_.data_ptr()))
prover = h2.PlonkProver(ctx, cs, k, g.data_ptr(), gl_.data_ptr(), w.fixed, w.assembly.mapping_array())
adv = torch.from_numpy(w.advice.view(np.int64)).cuda()
seeds = [int(sys.argv[1])] * 4 if len(sys.argv) > 1 else list(range(8))
for r in seeds:
    t0 = time.perf_counter()
    proof, ms = prover.prove(adv.data_ptr(), w.instance, bytes([r]) * 32, timed=True)
    print(r, round(time.perf_counter() - t0, 3), {k_: round(v, 1) for k_, v in ms.items()}, flush=True)
