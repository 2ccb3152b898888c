"""one bn256::Fr FFT of 2^k points and one batched MSM (8 columns) on resident operands: under `rocprofv3 --kernel-trace` the per-dispatch
durations show the three passes of the transform and the MSM's kernels one by one"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.cuda.init()
gl = importlib.import_module("stark-verifier_amd")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 23
ctx = gl.Context(0)
n = 1 << k
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.randint(0, (1 << 62), (n, 4), dtype=torch.int64, device="cuda", generator=g)
x[:, 3] >>= 3
torch.cuda.synchronize()
for _ in range(3):
    ctx.check(ctx.lib.gl355_bn254_fr_ntt(ctx.h, x.data_ptr(), k, 0))
ctx.sync()
if len(sys.argv) > 2:
    h2 = importlib.import_module("stark-verifier_amd.halo2")
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    tau = h2.to_limbs([12345678901234567890123])[0]
    ctx.check(ctx.lib.gl355_kzg_setup(ctx.h, tau.ctypes.data, k, pts.data_ptr(), None))
    sets = int(sys.argv[2])
    sc = torch.randint(0, (1 << 62), (sets, n, 4), dtype=torch.int64, device="cuda", generator=g)
    sc[:, :, 3] >>= 3
    out = np.zeros((sets, 8), dtype=np.uint64)
    torch.cuda.synchronize()
    for _ in range(2):
        ctx.check(ctx.lib.gl355_bn254_g1_msm_batch(ctx.h, pts.data_ptr(), sc.data_ptr(), n, sets, out.ctypes.data))
    ctx.sync()
ctx.close()
