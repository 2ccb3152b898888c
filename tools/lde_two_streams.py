"""A/B: the 135-column LDE 2^17 -> 2^20 as one call on one stream against column groups on 2 / 3 / 4 streams (one prover context each, host
threads): does running one group's column pass next to another group's row pass overlap their memory and arithmetic phases?"""
import ctypes as C
import importlib
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.cuda.init()
gl = importlib.import_module("stark-verifier_amd")
LOG_N, RB, B = 17, 3, 135
n, N = 1 << LOG_N, 1 << (LOG_N + RB)
g = torch.Generator(device="cuda")
g.manual_seed(1)
coeffs = torch.randint(0, (1 << 63) - 1, (B, n), dtype=torch.int64, device="cuda", generator=g)
out = torch.empty((B, N), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
alg = 8.0 * B * (n + N)


def run(groups, steps=40, warm=12):
    ctxs = [gl.Context(0) for _ in range(groups)]
    bounds = [round(i * B / groups) for i in range(groups + 1)]

    def one(t, reps):
        c0, c1 = bounds[t], bounds[t + 1]
        ctx = ctxs[t]
        for _ in range(reps):
            ctx.check(ctx.lib.gl355_lde_bitrev(ctx.h, C.c_void_p(coeffs[c0].data_ptr()), LOG_N, RB, 7, c1 - c0, C.c_void_p(out[c0].data_ptr())))
        ctx.sync()

    def all_(reps):
        ths = [threading.Thread(target=one, args=(t, reps)) for t in range(groups)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
    all_(warm)
    t0 = time.perf_counter()
    all_(steps)
    dt = (time.perf_counter() - t0) / steps
    for c in ctxs:
        c.close()
    return dt


for groups in (1, 2, 3, 4, 1):
    dt = run(groups)
    print("%d stream(s): %.3f ms per LDE of 135 columns, %.0f GB/s algorithmic" % (groups, dt * 1e3, alg / dt / 1e9), flush=True)
