"""One unit (Semaphore signal + the recursive proof verifying it) alone on one prover context, for a kernel timeline: run under
rocprofv3 --kernel-trace; the unit to look at is the last cluster of launches (0.5 s of silence on either side).
`python tools/latency_trace.py analyse <kernel_trace.csv>` prints where the wall time of that cluster goes: kernels, idle gaps by size,
the largest gaps with the launches around them."""
import csv
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    torch.cuda.init()
    import bench
    gl = importlib.import_module("stark-verifier_amd")
    rt = int(os.environ.get("GL355_LAT_REPLAY_THREADS", "8"))
    pr = bench.RecursiveProvers(gl, 0, 1, 20, replay_threads=rt, blocking_sync=int(os.environ.get("GL355_LAT_SYNC", "2")))
    if os.environ.get("GL355_LAT_WHAT") == "wrap":        # the BN254-Poseidon wrap proof (wrapper.rs:35-56) instead of a unit
        import numpy as np
        rec = importlib.import_module("stark-verifier_amd.recursion")
        pr.unit(0, 9100)
        inner, c0 = pr.last[0], pr.sets[0]
        wc = rec.WrapperCircuit(c0, pr.inner_data.common()).build([inner], np.random.default_rng(3))
        rows_w, pis_w = wc.witness([inner])
        one = lambda k: pr.plonk.prove_sparse(c0, wc.data, wc.row_idx, rows_w, pis_w, k, flat_only=True)
    else:
        one = lambda k: pr.unit(0, k)
    for k in range(3):
        one(9100 + k)
    lat = []
    for k in range(5):
        time.sleep(0.5)
        t0 = time.perf_counter()
        one(9200 + k)
        lat.append(1e3 * (time.perf_counter() - t0))
    time.sleep(0.5)
    print("unit latencies ms", [round(x, 2) for x in lat])


def analyse(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("gl355::", "").replace("void ", "")))
    rows.sort()
    # clusters separated by > 0.3 s
    clusters, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if b[0] - a[1] > 300_000_000:
            clusters.append(cur)
            cur = []
        cur.append(b)
    clusters.append(cur)
    c = clusters[-1] if len(clusters[-1]) > 50 else clusters[-2]
    span = (c[-1][1] - c[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in c) / 1e6
    print("launches %d  span %.2f ms  kernels %.2f ms  idle %.2f ms" % (len(c), span, busy, span - busy))
    gaps = [(b[0] - a[1], a[2], b[2]) for a, b in zip(c, c[1:])]
    for lo, hi in ((0, 2_000), (2_000, 5_000), (5_000, 10_000), (10_000, 20_000), (20_000, 50_000), (50_000, 100_000), (100_000, 10**9)):
        g = [x[0] for x in gaps if lo <= x[0] < hi]
        print("gaps %6.0f-%-7.0f us: %4d  total %.2f ms" % (lo / 1e3, hi / 1e3, len(g), sum(g) / 1e6))
    print("largest gaps (us, after -> before):")
    for g, a, b in sorted(gaps, reverse=True)[:40]:
        print("  %7.1f  %s -> %s" % (g / 1e3, a[:50], b[:50]))
    agg = {}
    for s, e, n in c:
        k = agg.setdefault(n, [0, 0])
        k[0] += 1
        k[1] += e - s
    print("kernels in the unit:")
    for n, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:25]:
        print("  %-60s %4d  %.3f ms" % (n[:60], k, t / 1e6))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "analyse":
        analyse(sys.argv[2])
    else:
        run()
