#!/usr/bin/env python3
"""One bn256::G1 MSM of 2^k points (bases G / 2G at random, 240-bit scalars) for `rocprofv3 --kernel-trace --stats`:
  rocprofv3 --kernel-trace --stats -d gpurun_out/msm -- python tools/prof_msm.py [k]"""
import ctypes as C
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gl = importlib.import_module("stark-verifier_amd")

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << k
ctx = gl.Context(0)
g = torch.Generator(device="cuda")
g.manual_seed(0x254)
# distinct bases s_i * G from the fixed-base kernel (gl355_bn254_g1_fixed_base_mul): the bucket phase's gathers are real ones
gen = torch.tensor([1, 0, 0, 0, 2, 0, 0, 0], dtype=torch.int64, device="cuda")
s_i = torch.randint(0, (1 << 60) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
t = time.time()
ctx.check(ctx.lib.gl355_bn254_g1_fixed_base_mul(ctx.h, C.c_void_p(gen.data_ptr()), C.c_void_p(s_i.data_ptr()), n, C.c_void_p(pts.data_ptr())))
ctx.sync()
print("fixed-base mul 2^%d: %.2f ms" % (k, (time.time() - t) * 1e3))
sc = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)     # uniform 64-bit limbs ...
sc[:, 3] &= (1 << 61) - 1                                                                                   # ... below 2^253 < r
res = torch.zeros(8, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for it in range(3):
    t = time.time()
    ctx.check(ctx.lib.gl355_bn254_g1_msm(ctx.h, C.c_void_p(pts.data_ptr()), C.c_void_p(sc.data_ptr()), n, C.c_void_p(res.data_ptr())))
    ctx.sync()
    print("msm 2^%d: %.2f ms" % (k, (time.time() - t) * 1e3))
# several scalar sets over the same bases in one call (GL355_MSM_SETS=m)
m = int(os.environ.get("GL355_MSM_SETS", "0"))
if m:
    scb = torch.randint(-(1 << 63), (1 << 63) - 1, (m, n, 4), dtype=torch.int64, device="cuda", generator=g)
    scb[:, :, 3] &= (1 << 61) - 1
    resb = torch.zeros((m, 8), dtype=torch.int64, device="cuda")
    for it in range(2):
        t = time.time()
        ctx.check(ctx.lib.gl355_bn254_g1_msm_batch(ctx.h, C.c_void_p(pts.data_ptr()), C.c_void_p(scb.data_ptr()), n, m, C.c_void_p(resb.data_ptr())))
        ctx.sync()
        dt = (time.time() - t) * 1e3
        print("msm batch of %d x 2^%d: %.2f ms = %.2f ms per MSM" % (m, k, dt, dt / m))
