#!/usr/bin/env python3
"""Latency of the witness generation of the depth-20 recursive circuit for one lock-step batch: host threads vs the device
tape interpreter (gl355_circuit_witness_rows).  python tools/replay_bench.py [units=8]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import bench
gl = importlib.import_module("stark-verifier_amd")
units = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pr = bench.RecursiveProvers(gl, 0, 1, replay_threads=2)
ctx = pr.sets[0]
inputs = []
for j in range(units):
    f, p = pr.sem.semaphore_prove(ctx, pr.sks[j], pr.topic, j, pr.aset.tree.prove_host(j), 7 + j)
    inputs.append(np.concatenate([f, p]))
inputs = np.stack(inputs)
for dev in (0, 1):
    pr.nat.witness_rows(ctx, inputs, dev)
    t0 = time.perf_counter()
    for _ in range(3):
        rows, pis = pr.nat.witness_rows(ctx, inputs, dev)
    dt = (time.perf_counter() - t0) / 3
    print("%s replay of %d units (n_rows %d): %.1f ms per batch (includes the %.0f-MB copy of the rows back to the host)" % (
        "device" if dev else "host (2 threads)", units, rows.shape[1], 1e3 * dt, rows.nbytes / 1e6))
ctx.profile_enable(True)
ctx.profile_read()
