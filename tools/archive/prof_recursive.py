#!/usr/bin/env python3
"""Per-kernel-group HIP-event times of one unit (Semaphore d=20 signal + recursive proof) on ONE context (no overlap between
streams, so the durations are the kernels' own), then units/s against the number of concurrent contexts."""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--contexts", default="1,2,4,8,12,16")
ap.add_argument("--units", type=int, default=48)
ap.add_argument("--lanes-log", default="14")
args = ap.parse_args()
gl = importlib.import_module("stark-verifier_amd")
counts = [int(x) for x in args.contexts.split(",")]
pr = bench.RecursiveProvers(gl, 0, max(counts))
sets = pr.sets
# ---- single context, per-kernel ----
pr.sets = sets[:1]
pr.prove_batch(100, 4)
pr.profile(True)
t0 = time.perf_counter()
pr.prove_batch(200, 8)
dt = time.perf_counter() - t0
prof, _ = pr.profile_read()
pr.profile(False)
tot = sum(v[1] for v in prof.values())
print("single context: %.2f ms wall per unit, %.2f ms inside kernel scopes" % (dt / 8 * 1e3, tot / 8))
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print("  %-26s launches/unit %6.1f  ms/unit %7.3f  %5.1f %%  alg %8.1f GB/s" % (k, v[0] / 8, v[1] / 8, 100 * v[1] / tot,
                                                                              v[2] / (v[1] * 1e-3) / 1e9 if v[1] > 0 else 0))
# host-side split of one unit
c = sets[0]
t = time.perf_counter(); flat, pis0 = pr.sem.semaphore_prove(c, pr.sks[3], pr.topic, 3, pr.aset.tree.prove_host(3), 5); t_in = time.perf_counter() - t
inp = np.concatenate([flat, pis0])
t = time.perf_counter(); rows, pis = pr.rc.witness([(flat, pis0)]); t_rep = time.perf_counter() - t
t = time.perf_counter(); pr.nat.prove_tape(c, inp, 1); t_out = time.perf_counter() - t
c.set_option(3, 4)
t = time.perf_counter(); pr.nat.prove_tape(c, inp, 1); t_out4 = time.perf_counter() - t
c.set_option(3, 1)
print("with 4 replay threads: replay + recursive proof %.2f ms" % (t_out4 * 1e3))
print("latency: signal %.2f ms, tape replay %.2f ms (inside the next figure), replay + recursive proof %.2f ms" % (t_in * 1e3, t_rep * 1e3, t_out * 1e3))
for ll in [int(x) for x in args.lanes_log.split(",")]:
  for a_ in sets:
    a_.set_option(1, ll)
  print("MERKLE_LANES_LOG", ll)
  for k in counts:
    pr.sets = sets[:k]
    pr.prove_batch(300, 2 * k)
    t0 = time.perf_counter()
    pr.prove_batch(400, args.units)
    dt = time.perf_counter() - t0
    print("  contexts %2d: %.1f units/s" % (k, args.units / dt))
