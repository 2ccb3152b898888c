#!/usr/bin/env python3
"""Soak: the default bench workload (8 contexts x 8 lock-step units, polling waits; GL355_BENCH_DEVICE_REPLAY=0/1 picks the witness
replay) for a couple of minutes; device-free memory and host RSS must stay flat.  usage: python tools/soak.py [seconds]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
lib = importlib.import_module("stark-verifier_amd._lib").load(init_torch=False)
assert lib.gl355_runtime_config(0, 8, 0) == 0           # before the HIP runtime initialises
import psutil
import bench
gl = importlib.import_module("stark-verifier_amd")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120
pr = bench.RecursiveProvers(gl, 0, 8, replay_threads=2, blocking_sync=2)
proc = psutil.Process()
t0 = time.time(); units = 0; k = 0
while time.time() - t0 < seconds:
    pr.prove_batch(1000 + units, 128); units += 128; k += 1
    if k % 10 == 0:
        free, total = torch.cuda.mem_get_info()
        print("t=%5.0fs units=%6d  %.1f units/s  device used %.2f GB  host RSS %.2f GB" % (
            time.time() - t0, units, units / (time.time() - t0), (total - free) / 2**30, proc.memory_info().rss / 2**30), flush=True)
