#!/usr/bin/env python3
"""Semaphore (depth 20) proofs/s on one GPU with K concurrent prover contexts (one stream each).
  python tools/bench_proofs.py [threads=4] [proofs_per_thread=8] [log_members=20]"""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
gl = importlib.import_module("stark-verifier_amd")
sem = importlib.import_module("stark-verifier_amd.semaphore")
from oracle_lib import rand_field

def run(threads, per_thread, log_members):
    rng = np.random.default_rng(0x357)
    ctx0 = gl.Context(0)
    sks = rand_field(rng, (1 << log_members, 4))
    keys = ctx0.hash_no_pad(np.concatenate([sks, np.zeros_like(sks)], axis=1))
    topic = rand_field(rng, 4)
    sets = []
    for t in range(threads):
        c = gl.Context(0)
        a = sem.AccessSet(c, keys)
        a.build(np.random.default_rng(1))
        a.make_signal_fast(sks[t], topic, t, t)      # warm-up (tables, allocator)
        sets.append(a)
    done = [0] * threads
    def worker(t):
        for k in range(per_thread):
            i = 100 + t * per_thread + k
            sets[t].make_signal_fast(sks[i], topic, i, 0x358 + i, flat_only=True)
            done[t] += 1
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    dt = time.perf_counter() - t0
    total = sum(done)
    print("threads=%d proofs=%d  %.3f s  %.1f proofs/s  (%.1f ms/proof/thread)" % (threads, total, dt, total / dt, dt / per_thread * 1e3), flush=True)
    return total / dt

if __name__ == "__main__":
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    lm = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    ks = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8]
    for k in ks:
        run(k, per, lm)
