#!/usr/bin/env python3
"""One forward NTT (natural order in and out) of the bench shape 2^20 x 135, repeated a few times (target for rocprofv3 runs)."""
import ctypes as C, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gl = importlib.import_module("stark-verifier_amd")
log_n, batch = int(os.environ.get("LOGN", "20")), 135
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = gl.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randint(0, (1 << 63) - 1, (batch, 1 << log_n), dtype=torch.int64, device="cuda", generator=g)
torch.cuda.synchronize()
for _ in range(reps):
    ctx.check(ctx.lib.gl355_ntt(ctx.h, C.c_void_p(x.data_ptr()), log_n, batch, 1 << log_n, 0))
ctx.sync()
ctx.close()
