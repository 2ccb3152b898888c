#!/usr/bin/env python3
"""One LDE of the bench shape, repeated a few times (target for rocprofv3 runs)."""
import ctypes as C, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gl = importlib.import_module("stark-verifier_amd")
log_n, rb, batch = 17, 3, 135
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = gl.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(1)
c = torch.randint(0, (1 << 63) - 1, (batch, 1 << log_n), dtype=torch.int64, device="cuda", generator=g)
out = torch.empty((batch, 1 << (log_n + rb)), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for _ in range(reps):
    ctx.check(ctx.lib.gl355_lde_bitrev(ctx.h, C.c_void_p(c.data_ptr()), log_n, rb, 7, batch, C.c_void_p(out.data_ptr())))
ctx.sync()
ctx.close()
