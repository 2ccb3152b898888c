#!/usr/bin/env python3
"""Host-memory growth per unit of the default workload's pieces (tools/soak.py shows ~0.19 GB steps of host RSS every ~25 k units):
   python tools/leak_probe.py [seconds per variant]
variants: the signal proofs alone, whole units with the witness tape replayed on the host, whole units with the device replay."""
import gc, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
lib = importlib.import_module("stark-verifier_amd._lib").load(init_torch=False)
assert lib.gl355_runtime_config(0, 8, 0) == 0
import psutil
import numpy as np
import bench
gl = importlib.import_module("stark-verifier_amd")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40
proc = psutil.Process()


def rss():
    return proc.memory_info().rss / 2**20


for name, replay in (("units, host tape replay", "0"), ("units, device tape replay", "1"), ("signals only", None)):
    if replay is not None:
        os.environ["GL355_BENCH_DEVICE_REPLAY"] = replay
    pr = bench.RecursiveProvers(gl, 0, 8, replay_threads=2, blocking_sync=2)
    plonk = pr.plonk
    n = pr.sks.shape[0]

    def step(first):
        members = (first + np.arange(128, dtype=np.uint64)) % np.uint64(n)
        if replay is None:
            plonk.semaphore_units(pr.sets, pr.sem, None, pr.sks, pr.topic, pr.aset.tree.digests, members, None)
        else:
            plonk.semaphore_units(pr.sets, pr.sem, pr.nat, pr.sks, pr.topic, pr.aset.tree.digests, members, None)
    for k in range(8):
        step(k * 128)
    gc.collect()
    r0, t0, units = rss(), time.time(), 0
    marks = []
    while time.time() - t0 < seconds:
        step(1000 + units); units += 128
        if units % 2560 == 0:
            marks.append((units, round(rss() - r0, 1)))
    print("%-28s %6d units  %.1f units/s  RSS growth %.1f MB = %.2f KB per unit   trace %s" % (
        name, units, units / (time.time() - t0), rss() - r0, (rss() - r0) * 1024 / units, marks[::2]), flush=True)
    for c in pr.sets:
        c.close()
    del pr
    gc.collect()
