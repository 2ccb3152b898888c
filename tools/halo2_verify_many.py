import importlib, os, sys, time
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tools")); sys.path.insert(0, os.path.join(ROOT,"tests"))
import torch; torch.cuda.init()
gl = importlib.import_module("stark-verifier_amd")
import halo2_bench as hb, halo2_verifier as hv
h2 = importlib.import_module("stark-verifier_amd.halo2"); ch = importlib.import_module("stark-verifier_amd.halo2_chips")
ctx = gl.Context(0)
for k, seeds in ((18, range(4)), (20, range(6)), (22, range(2))):
    cs, cfg, w = ch.synthetic_circuit(k, table_bits=16, n_permutations=64, seed=0x355 + k)
    n = 1 << k
    g = torch.empty((n, 8), dtype=torch.int64, device="cuda"); gl_ = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    tau = h2.to_limbs([hb.TAU])[0]
    ctx.check(ctx.lib.gl355_kzg_setup(ctx.h, tau.ctypes.data, k, g.data_ptr(), gl_.data_ptr()))
    prover = h2.PlonkProver(ctx, cs, k, g.data_ptr(), gl_.data_ptr(), w.fixed, w.assembly.mapping_array())
    pt = lambda a: (lambda x, y: None if (x, y) == (0, 0) else (x, y))(h2.from_limbs(a[:4])[0], h2.from_limbs(a[4:])[0])
    vk = dict(digest=prover.digest, fixed_commitments=[pt(c) for c in prover.fixed_commitments], sigma_commitments=[pt(c) for c in prover.sigma_commitments])
    adv = torch.from_numpy(w.advice.view(np.int64)).cuda()
    for s in seeds:
        t0 = time.perf_counter()
        proof = prover.prove(adv.data_ptr(), w.instance, bytes([17 * s + 3]) * 32)
        dt = time.perf_counter() - t0
        ok = hv.verify(k, cs, vk, w.instance, proof, hb.TAU % h2.R)
        print(k, s, round(dt, 3), ok, flush=True)
    prover.close(); del adv, g, gl_; torch.cuda.empty_cache()
