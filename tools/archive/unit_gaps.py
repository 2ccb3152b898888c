#!/usr/bin/env python3
"""Where one prover context's time goes: rocprofv3 --kernel-trace of ONE context proving lock-step batches; per batch the wall
time, the summed kernel time, and the idle gaps of the stream grouped by the kernel that precedes the gap (= the host round trip
that follows it).  Usage on the GPU box:
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/unit_gaps.py run [B] [batches] [sleep]
   python tools/unit_gaps.py report DIR"""
import collections, csv, glob, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(B, batches, sleep):
    import numpy as np
    import torch  # noqa
    lib = importlib.import_module("stark-verifier_amd._lib").load(init_torch=False)
    assert lib.gl355_runtime_config(0, 4, sleep) == 0
    import bench, time
    gl = importlib.import_module("stark-verifier_amd")
    os.environ["GL355_BENCH_BATCH_UNITS"] = str(B)
    pr = bench.RecursiveProvers(gl, 0, 1, replay_threads=4)
    pr.prove_batch(100, B)
    t0 = time.perf_counter()
    pr.prove_batch(1000, B * batches)
    dt = time.perf_counter() - t0
    print("one context, B=%d, %d batches, sleeping waits %d: %.2f ms per unit wall" % (B, batches, sleep, 1e3 * dt / (B * batches)))


def report(d):
    rows = []
    for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gl355::", "")))
    rows.sort()
    rows = rows[len(rows) // 3:]                       # skip set-up and warm-up
    busy = sum(e - s for s, e, _ in rows)
    wall = rows[-1][1] - rows[0][0]
    gaps = collections.defaultdict(lambda: [0, 0])
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = s1 - e0
        if g > 0:
            k = gaps[n0 + " -> " + n1]
            k[0] += 1; k[1] += g
    print("%d kernels over %.1f ms: kernels busy %.1f %%, idle %.1f ms" % (len(rows), wall / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6))
    for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %6.2f ms in %4d gaps (%6.1f us each) after %s" % (t / 1e6, c, t / c / 1e3, k))
    dur = collections.defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        dur[n][0] += 1; dur[n][1] += e - s
    print("kernel time:")
    for k, (c, t) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:22]:
        print("  %6.2f ms %5d x %8.1f us  %s" % (t / 1e6, c, t / c / 1e3, k))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 4, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    else:
        report(sys.argv[2])
