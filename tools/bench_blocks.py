"""The secondary figures of the bench line (`ntt_lde`, `merkle_2p22`, `bn254_finalisation_kernels`, `aggregate`, `halo2_create_proof_k23`, the
cfg-2 sweep) and the secondary workloads (`--workload lde | semaphore | exchange`), behind the same JSON keys as before.  Moved out of bench.py in
round 5: no behaviour change."""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

from bench_common import *          # noqa: F401,F403  (ROOT, the BASELINE shape constants, profile look-ups, VALU peak, open_comm)
from bench_common import _TorchComm  # noqa: F401


def cpu_baseline(seconds=4.0):
    """The CPU restatement (oracle, OpenMP over columns like plonky2's rayon) on the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle, rand_field
    orc = Oracle()
    orc.L.orc_set_num_threads(host_cores())
    threads = orc.L.orc_num_threads()
    rng = np.random.default_rng(0x355)
    n, N = 1 << LOG_N, 1 << (LOG_N + RATE_BITS)
    # bounded sample: as many columns as threads (at most the full batch), repeated for ~`seconds`
    cols = max(1, min(BATCH, threads))
    coeffs = rand_field(rng, (cols, n))
    out = np.empty((cols, N), dtype=np.uint64)
    u64p = C.POINTER(C.c_uint64)
    reps, t0 = 0, time.perf_counter()
    while True:
        orc.L.orc_lde(coeffs.ctypes.data_as(u64p), C.c_uint32(LOG_N), C.c_uint32(RATE_BITS), C.c_uint64(7),
                      C.c_uint32(cols), out.ctypes.data_as(u64p))
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or reps >= 50:
            break
    gbs = 8.0 * cols * (n + N) * reps / dt / 1e9
    return {"value": round(gbs, 3), "unit": "GB/s", "cores": int(threads), "kind": "port",
            "sample": "%d columns of the same 2^%d->2^%d LDE, %d repetitions, %.1f s wall (C restatement of "
                      "plonky2's lde+coset_fft, OpenMP over columns; not the Rust binary)" % (cols, LOG_N, LOG_N + RATE_BITS, reps, dt)}


class SemaphoreProvers:
    """K concurrent prover contexts (one HIP stream each) on one GPU proving depth-20 Semaphore signals
    (make_signal, access_set.rs:61-104), without the recursive proof."""

    def __init__(self, gl, device, threads, log_members=20, seed=0x357):
        rand_field = gl.api.rand_field
        sem = importlib.import_module("stark-verifier_amd.semaphore")
        rng = np.random.default_rng(seed)
        ctx0 = gl.Context(device)
        self.sks = rand_field(rng, (1 << log_members, 4))
        keys = ctx0.hash_no_pad(np.concatenate([self.sks, np.zeros_like(self.sks)], axis=1))
        self.topic = rand_field(rng, 4)
        self.sets = []
        for t in range(threads):
            a = sem.AccessSet(gl.Context(device), keys)
            a.build(np.random.default_rng(1))
            a.make_signal_fast(self.sks[t], self.topic, t, t)   # warm-up
            self.sets.append(a)
        self.root = self.sets[0].tree.cap[0].copy()

    def prove_batch(self, first, count):
        """proves members first..first+count-1, returns their (nullifier | topic) leaves [count][8]"""
        import threading
        k = len(self.sets)
        leaves = np.zeros((count, 8), dtype=np.uint64)

        def worker(t):
            for j in range(t, count, k):
                i = first + j
                sig, _ = self.sets[t].make_signal_fast(self.sks[i], self.topic, i, 0x358 + i, flat_only=True)
                leaves[j, :4] = sig.nullifier[0]
                leaves[j, 4:] = self.topic
        ths = [threading.Thread(target=worker, args=(t,)) for t in range(k)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        return leaves


def lde_figure(gl, device, steps=40, warm=12):
    """BASELINE configs[1] on this GPU: the metric's 'NTT HBM GB/s' half (full treatment: --workload lde).  The first ~10 launches
    after the device did something else run 15-20 % slower (memory-side clocks settling, profiles/r03_ubench_ntt_l24s.txt), so the
    figure is taken after `warm` untimed steps over `steps` steps (~50 ms in all)."""
    import torch
    ctx = gl.Context(device)
    lib = ctx.lib
    n, N = 1 << LOG_N, 1 << (LOG_N + RATE_BITS)
    dev = torch.device("cuda", device)
    g = torch.Generator(device=dev)
    g.manual_seed(0x355)
    coeffs = torch.randint(0, (1 << 63) - 1, (BATCH, n), dtype=torch.int64, device=dev, generator=g)
    out = torch.empty((BATCH, N), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def step():
        ctx.check(lib.gl355_lde_bitrev(ctx.h, C.c_void_p(coeffs.data_ptr()), LOG_N, RATE_BITS, 7, BATCH, C.c_void_p(out.data_ptr())))
    for _ in range(warm):
        step()
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_read()
    with ClockSampler(gl, device, period=0.004) as cs:
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.sync()
        dt = time.perf_counter() - t0
    clk = cs.summary()
    prof = {k: v for k, v in ctx.profile_read().items() if not k.startswith("host:")}
    ctx.profile_enable(False)
    alg = 8.0 * BATCH * (n + N)
    kern_ms = sum(v[1] / max(1, v[0]) for k, v in prof.items() if k.startswith("ntt_"))
    ach = alg / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    del coeffs, out
    ctx.close()
    traffic, tsrc, valu = lde_pmc()
    return {"value": round(alg * steps / dt / 1e9, 2), "unit": "GB/s", "steps": steps,
            "workload": "lde n=2^17 -> N=2^20, 135 columns, bit-reversed output, resident operands",
            "roofline": {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": tsrc,
                         "hbm_moved_GBps": round(traffic / (kern_ms * 1e-3) / 1e9, 1) if kern_ms > 0 and traffic else None,
                         "valu_insts_per_lde": valu,
                         "floors": lde_floors(alg, clk),
                         "kernels_ms_per_launch": {k: round(v[1] / max(1, v[0]), 4) for k, v in prof.items()}}}


def lde_floors(alg_bytes, clk):
    """What bounds the two-pass LDE from below, per pass: its HBM-side traffic at the copy rate the chip reaches (6.29 TB/s, MI355X_MICROARCH.md) and
    its VALU instructions (rocprofv3 --pmc SQ_INSTS_VALU) at the pair-aware issue floor of the pass's opcode forms (tools/isa_mix.py forms at the
    costs gl355_valu_probe_ops / _pairs measured in this run: bench_common.ValuModel) and the clock sampled during this run.  A pass cannot beat the larger of its two floors even with perfect overlap; the sum over the passes is the
    ceiling of THIS arithmetic (64-bit modular butterflies as 24-bit-limb integer work on a 32-bit VALU) in this two-pass structure, and
    `ceiling_frac_of_hbm_peak` is where the north star's ">= 50 % of HBM" target lands for it."""
    try:
        d = json.load(open(latest_profile("_lde_pmc.json")))
        isa = json.load(open(latest_profile("_isa_mix.json")))["kernels"]
    except Exception:
        return None
    if not clk:
        return None
    copy_peak = 6.29e12
    out = {"clock_mhz": clk["mean_mhz"], "copy_peak_TBps": 6.29, "passes": {}}
    total = 0.0
    for name, e in d["kernels"].items():
        import bench_common
        model = bench_common.VALU_MODEL
        n64, n32 = e.get("SQ_INSTS_VALU_INT64"), e.get("SQ_INSTS_VALU_INT32")
        forms = (model.dynamic_forms(name, 1.0, n64 / e["SQ_INSTS_VALU"], n32 / e["SQ_INSTS_VALU"]) if n64 is not None and n32 is not None
                 else model.dynamic_forms(name)) if model else None
        if not forms:
            continue
        add, fl, _ = model.cycles(forms)                   # per instruction (the forms are fractions; moved to the counters' dynamic class shares)
        cost = fl or add
        valu_ms = e["SQ_INSTS_VALU"] * cost / (N_SIMD * clk["mean_mhz"] * 1e6) * 1e3
        mem_ms = e["hbm_bytes_per_launch"] / copy_peak * 1e3
        out["passes"][name] = {"valu_floor_ms": round(valu_ms, 3), "mem_floor_ms": round(mem_ms, 3), "clk_per_inst_floor": round(cost, 3),
                               "clk_per_inst_additive": round(add, 3)}
        total += max(valu_ms, mem_ms)
    if total <= 0:
        return None
    out["valu_floor_ms"] = round(sum(p["valu_floor_ms"] for p in out["passes"].values()), 3)
    out["mem_floor_ms"] = round(sum(p["mem_floor_ms"] for p in out["passes"].values()), 3)
    out["floor_ms_perfect_overlap"] = round(total, 3)
    out["ceiling_GBps"] = round(alg_bytes / (total * 1e-3) / 1e9, 1)
    out["ceiling_frac_of_hbm_peak"] = round(alg_bytes / (total * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return out


def lde_pmc():
    """(HBM-side bytes per LDE, source, wave-level VALU instructions per LDE) from the latest committed --pmc passes over the LDE"""
    path = latest_profile("_lde_pmc.json")
    try:
        d = json.load(open(path))
        return int(d["hbm_bytes_per_lde"]), "profiles/%s: %s" % (os.path.basename(path), d["_source"]), d.get("valu_insts_per_lde")
    except Exception:
        return None, None, None


PSD_VALU_PER_PERM, PSD_MAD_PER_PERM = 15700, 10380      # static counts of the shipped ISA (hipcc -S): VALU 8 x 1 077 + 2 x 3 476 + 156; multiply-adds 8 x 588 + 2 x 2 838


def merkle_figures(gl, device):
    """BASELINE configs[2] / SURVEY cfg-3: MerkleTree::new over 2^22 leaves -- (L = 4, cap 4) the FRI-layer shape, (L = 135, cap 4)
    the wires-like shape (4.5 GB of leaves) -- and the Semaphore group tree (2^20 leaves, L = 4, cap 0, signal.rs:40); row-major leaves
    resident in HBM, gl355_merkle_build, HIP-event time per build.  The kernels are integer-VALU work (a permutation is 15.7 k VALU
    instructions, 10 380 of them v_mad_u64_u32): `valu` prices the build's permutations at that count against the chip's issue rate;
    `valu.multiply_adds_only` prices just the multiply-adds the formulation needs (the distance to a kernel that issued nothing else)."""
    import torch
    ctx = gl.Context(device)
    lib = ctx.lib
    g = torch.Generator(device="cuda")
    g.manual_seed(0x356)
    out = {"what": "gl355_merkle_build (leaf sponge + compression levels + cap, plonky2 digest layout), leaves resident, HIP events"}
    # the VALU roofline of these builds: PSD_VALU_PER_PERM instructions per permutation and lane (static count of the shipped ISA: 8 full rounds x 1 077 +
    # 2 blocks x 3 476 + the first constant layer; profiles/r05_poseidon_block_vs_dense.txt) against the issue rate for hash_leaves_kernel's
    # instruction mix (valu_peak) at the shader clock sampled while the builds run
    import bench_common
    model = bench_common.VALU_MODEL
    pe = model.pmc.get("probes", {}).get("vpc_permute_kernel") if model else None        # dynamic instructions per permutation (rocprofv3 --pmc over the probe)
    perm_forms = model.dynamic_forms("vpc_permute_kernel", pe["valu_insts_per_item"], pe["valu_int64_per_item"], pe["valu_int32_per_item"]) if pe else None
    perm_insts = pe["valu_insts_per_item"] if pe else PSD_VALU_PER_PERM
    mad_clk = min(v["clk"] for f, v in model.ops.items() if f.startswith("v_mad_u64_u32")) if model else 4.0
    for log_n, L, cap in ((22, 4, 4), (22, 135, 4), (20, 4, 0)):
        n = 1 << log_n
        leaves = torch.randint(0, (1 << 63) - 1, (n, L), dtype=torch.int64, device="cuda", generator=g)
        dig = torch.empty((2 * (n - (1 << cap)), 4), dtype=torch.int64, device="cuda")
        capb = torch.empty((1 << cap, 4), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()

        def build():
            ctx.check(lib.gl355_merkle_build(ctx.h, C.c_void_p(leaves.data_ptr()), n, L, cap, C.c_void_p(dig.data_ptr()), C.c_void_p(capb.data_ptr())))
        build()
        ctx.sync()
        reps = 6 if L > 8 else 40
        with ClockSampler(gl, device, period=0.01) as cs:
            ctx.timer_start()
            for _ in range(reps):
                build()
            ms = ctx.timer_stop() / reps
        clk = cs.summary()
        perms = n * ((L + 7) // 8 if L > 4 else 0) + (n - (1 << cap))
        alg = 8.0 * n * L + 64.0 * (n - (1 << cap)) + 32.0 * (1 << cap)
        out["N=2^%d L=%d cap=%d" % (log_n, L, cap)] = {
            "ms": round(ms, 3), "permutations": perms, "Gperm_per_s": round(perms / ms / 1e6, 3),
            "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4),
                         "algorithmic_bytes": int(alg)},
            "valu": None}
        if perm_forms and clk:
            pk = model.peak(perm_forms, clk["mean_mhz"])
            ach_v = perms * perm_insts / 64.0 / (ms * 1e-3) / 1e9
            ach_m = perms * 1077 * 4 / 64.0 / (ms * 1e-3) / 1e9
            peak_m = N_SIMD * clk["mean_mhz"] * 1e6 / mad_clk / 1e9
            out["N=2^%d L=%d cap=%d" % (log_n, L, cap)]["valu"] = {
                "bound": "valu", "achieved": round(ach_v, 1), "peak": pk["peak"], "unit": "G wave-instructions/s", "frac": round(ach_v / pk["peak"], 4),
                "clock_mhz": clk["mean_mhz"], "valu_insts_per_permutation": perm_insts, "clk_per_inst_floor": pk.get("clk_per_inst_floor"),
                "additive_model": {"peak": pk["peak_additive"], "frac": round(ach_v / pk["peak_additive"], 4)},
                "formula": "permutations x valu_insts_per_permutation (rocprofv3 --pmc over the permutation probe) / 64 / time against 1024 SIMDs x clock / "
                           "the pair-aware floor of the permutation's opcode forms (costs measured in this run)",
                "algorithmic_multiply_adds": {"achieved": round(ach_m, 1), "peak": round(peak_m, 1), "unit": "G v_mad_u64_u32/s (wave level)", "frac": round(ach_m / peak_m, 4),
                                              "formula": "SURVEY 8(d) cfg-3: permutations x 1 077 modular multiplications x 4 multiply-adds / 64 / time against "
                                                         "1024 SIMDs x clock / %.3f clk (the measured v_mad_u64_u32 cost)" % mad_clk}}
        del leaves, dig, capb
        torch.cuda.empty_cache()
    ctx.close()
    return out


def cfg2_sweep(ctx, dev):
    """BASELINE configs[1] / SURVEY cfg-2: (A) forward NTT N = 2^20, natural order in and out, batch 1 / 16 / 135; (B) LDE 2^17 -> 2^20
    (rate_bits 3, coset 7), same batches, bit-reversed (commitment) order; (C) forward NTT N = 2^16 .. 2^23, batch 16.  Operands resident;
    HIP-event time per call in steady state; algorithmic bytes 16 B N (A, C) and 8 B (n + N) (B)."""
    import torch
    lib = ctx.lib
    g = torch.Generator(device=dev)
    g.manual_seed(0x355)

    def timed(fn):
        # steady state (see lde_figure): warm up for ~15 ms, then time ~40 ms
        fn(); ctx.sync()
        ctx.timer_start(); fn(); one = max(ctx.timer_stop(), 1e-3)
        for _ in range(min(200, int(15.0 / one) + 2)):
            fn()
        ctx.sync()
        reps = min(400, int(40.0 / one) + 4)
        ctx.timer_start()
        for _ in range(reps):
            fn()
        return ctx.timer_stop() / reps
    out = {"A_forward_ntt_2p20": {}, "B_lde_2p17_to_2p20": {}, "C_forward_ntt_batch16": {}}
    for b in (1, 16, 135):
        x = torch.randint(0, (1 << 63) - 1, (b, 1 << 20), dtype=torch.int64, device=dev, generator=g)
        torch.cuda.synchronize()
        ms = timed(lambda: ctx.check(lib.gl355_ntt(ctx.h, C.c_void_p(x.data_ptr()), 20, b, 1 << 20, 0)))
        out["A_forward_ntt_2p20"]["batch_%d" % b] = {"ms": round(ms, 4), "GBps": round(16.0 * b * (1 << 20) / ms / 1e6, 1)}
        c = x[:, :1 << 17].contiguous()
        o = torch.empty((b, 1 << 20), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        ms = timed(lambda: ctx.check(lib.gl355_lde_bitrev(ctx.h, C.c_void_p(c.data_ptr()), 17, 3, 7, b, C.c_void_p(o.data_ptr()))))
        out["B_lde_2p17_to_2p20"]["batch_%d" % b] = {"ms": round(ms, 4), "GBps": round(8.0 * b * ((1 << 17) + (1 << 20)) / ms / 1e6, 1)}
        del x, c, o
    for lg in range(16, 24):
        x = torch.randint(0, (1 << 63) - 1, (16, 1 << lg), dtype=torch.int64, device=dev, generator=g)
        torch.cuda.synchronize()
        ms = timed(lambda: ctx.check(lib.gl355_ntt(ctx.h, C.c_void_p(x.data_ptr()), lg, 16, 1 << lg, 0)))
        out["C_forward_ntt_batch16"]["2p%d" % lg] = {"ms": round(ms, 4), "GBps": round(16.0 * 16 * (1 << lg) / ms / 1e6, 1)}
        del x
    return out


def bn254_figures(gl, device):
    """SURVEY 8(f) N4 first slice: bn256::Fr FFT (k = 20, 22) and bn256::G1 MSM (2^20 points) on resident operands"""
    import torch
    ctx = gl.Context(device)
    lib = ctx.lib
    g = torch.Generator(device="cuda")
    g.manual_seed(0x254)
    out = {"what": "halo2 best_fft over bn256::Fr and best_multiexp over bn256::G1 (verifier_api.rs:77-92), operands resident; MSM: signed 17-bit windows, buckets by decreasing size, recursive bucket reduction, windows combined on the host"}
    for k in (20, 22):
        x = torch.randint(0, (1 << 60) - 1, (1 << k, 4), dtype=torch.int64, device="cuda", generator=g)
        torch.cuda.synchronize()
        ctx.check(lib.gl355_bn254_fr_ntt(ctx.h, C.c_void_p(x.data_ptr()), k, 0))
        ctx.sync()
        ctx.timer_start()
        for _ in range(3):
            ctx.check(lib.gl355_bn254_fr_ntt(ctx.h, C.c_void_p(x.data_ptr()), k, 0))
        ms = ctx.timer_stop() / 3
        out["fr_ntt_k%d" % k] = {"ms": round(ms, 3), "butterflies_per_s": round((1 << (k - 1)) * k / ms * 1e3 / 1e9, 2), "unit": "G butterflies/s"}
        del x
    n = 1 << 20
    # 2^20 DISTINCT bases s_i * G from the fixed-base kernel (the powers-of-tau loop of ParamsKZG::setup): the bucket phase's point
    # gathers are real ones; parity of both entries: tests/test_gpu_bn254_curve.py
    gen = torch.tensor([1, 0, 0, 0, 2, 0, 0, 0], dtype=torch.int64, device="cuda")
    s_i = torch.randint(0, (1 << 60) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    pts = torch.empty((n, 8), dtype=torch.int64, device="cuda")
    ctx.check(lib.gl355_bn254_g1_fixed_base_mul(ctx.h, C.c_void_p(gen.data_ptr()), C.c_void_p(s_i.data_ptr()), n, C.c_void_p(pts.data_ptr())))
    ctx.sync()
    ctx.timer_start()
    ctx.check(lib.gl355_bn254_g1_fixed_base_mul(ctx.h, C.c_void_p(gen.data_ptr()), C.c_void_p(s_i.data_ptr()), n, C.c_void_p(pts.data_ptr())))
    ms = ctx.timer_stop()
    out["g1_fixed_base_mul_2p20"] = {"ms": round(ms, 2), "points_per_s": round(n / ms * 1e3 / 1e6, 2), "unit": "M points/s"}
    sc = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)     # uniform 64-bit limbs ...
    sc[:, 3] &= (1 << 61) - 1                                                                                   # ... below 2^253 < r
    res = torch.zeros(8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.check(lib.gl355_bn254_g1_msm(ctx.h, C.c_void_p(pts.data_ptr()), C.c_void_p(sc.data_ptr()), n, C.c_void_p(res.data_ptr())))
    ctx.sync()
    ctx.timer_start()
    for _ in range(3):
        ctx.check(lib.gl355_bn254_g1_msm(ctx.h, C.c_void_p(pts.data_ptr()), C.c_void_p(sc.data_ptr()), n, C.c_void_p(res.data_ptr())))
    ms = ctx.timer_stop() / 3
    out["g1_msm_2p20"] = {"ms": round(ms, 2), "points_per_s": round(n / ms * 1e3 / 1e6, 2), "unit": "M points/s", "bases": "distinct"}
    m_sets = 8                                                       # the commitments of 8 columns under one SRS in one call
    scb = torch.randint(-(1 << 63), (1 << 63) - 1, (m_sets, n, 4), dtype=torch.int64, device="cuda", generator=g)
    scb[:, :, 3] &= (1 << 61) - 1
    resb = torch.zeros((m_sets, 8), dtype=torch.int64, device="cuda")
    ctx.check(lib.gl355_bn254_g1_msm_batch(ctx.h, C.c_void_p(pts.data_ptr()), C.c_void_p(scb.data_ptr()), n, m_sets, C.c_void_p(resb.data_ptr())))
    ctx.sync()
    ctx.timer_start()
    ctx.check(lib.gl355_bn254_g1_msm_batch(ctx.h, C.c_void_p(pts.data_ptr()), C.c_void_p(scb.data_ptr()), n, m_sets, C.c_void_p(resb.data_ptr())))
    ms = ctx.timer_stop()
    out["g1_msm_batch_8x2p20"] = {"ms": round(ms, 2), "ms_per_msm": round(ms / m_sets, 2), "points_per_s": round(m_sets * n / ms * 1e3 / 1e6, 2), "unit": "M points/s"}
    # the reference's circuit size (README.md:171-177: k = 23): SRS on the device, FFT, extended-domain FFT, commit (= MSM over 2^23 distinct
    # bases), single-point opening (synthetic division + MSM); parity of these entries: tests/test_gpu_kzg.py
    try:
        k = 23
        n = 1 << k
        del pts, sc, scb, s_i
        torch.cuda.empty_cache()
        tau = np.array([0x5E3F50617283940A, 0x1B2C3D4E5F607182, 0x93A4B5C6D7E8F901, 0x0203040506070809], dtype=np.uint64)
        srs = torch.empty((n, 8), dtype=torch.int64, device="cuda")
        t0 = time.perf_counter()
        ctx.check(lib.gl355_kzg_setup(ctx.h, tau.ctypes.data, k, C.c_void_p(srs.data_ptr()), None))
        ctx.sync()
        kz = {"setup_powers_of_tau_ms": round(1e3 * (time.perf_counter() - t0), 1)}
        poly = torch.randint(-(1 << 63), (1 << 63) - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
        poly[:, 3] &= (1 << 61) - 1
        x = poly.clone()
        ctx.check(lib.gl355_bn254_fr_ntt(ctx.h, C.c_void_p(x.data_ptr()), k, 0))
        ctx.sync()
        ctx.timer_start()
        ctx.check(lib.gl355_bn254_fr_ntt(ctx.h, C.c_void_p(x.data_ptr()), k, 0))
        kz["fr_ntt_k23_ms"] = round(ctx.timer_stop(), 2)
        ext = torch.empty((1 << 25, 4), dtype=torch.int64, device="cuda")
        sh = np.array([7, 0, 0, 0], dtype=np.uint64)
        ctx.check(lib.gl355_bn254_fr_coset_ntt(ctx.h, C.c_void_p(poly.data_ptr()), k, 25, sh.ctypes.data, 0, C.c_void_p(ext.data_ptr())))
        ctx.sync()
        ctx.timer_start()
        ctx.check(lib.gl355_bn254_fr_coset_ntt(ctx.h, C.c_void_p(poly.data_ptr()), k, 25, sh.ctypes.data, 0, C.c_void_p(ext.data_ptr())))
        kz["coeff_to_extended_23_to_25_ms"] = round(ctx.timer_stop(), 2)
        del ext, x
        cm = np.zeros(8, dtype=np.uint64)
        ctx.check(lib.gl355_kzg_commit(ctx.h, C.c_void_p(srs.data_ptr()), C.c_void_p(poly.data_ptr()), k, 0, cm.ctypes.data))
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.check(lib.gl355_kzg_commit(ctx.h, C.c_void_p(srs.data_ptr()), C.c_void_p(poly.data_ptr()), k, 0, cm.ctypes.data))
        kz["commit_msm_2p23_ms"] = round(1e3 * (time.perf_counter() - t0) / 3, 2)
        kz["commit_points_per_s"] = round(n / kz["commit_msm_2p23_ms"] * 1e3 / 1e6, 1)
        ev, wit = np.zeros(4, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        z = np.array([0x8899AABBCCDDEEFF, 0x0011223344556677, 0x8796A5B4C3D2E1F0, 0x0F1E2D3C4B5A6978 >> 4], dtype=np.uint64)
        t0 = time.perf_counter()
        ctx.check(lib.gl355_kzg_open(ctx.h, C.c_void_p(srs.data_ptr()), C.c_void_p(poly.data_ptr()), k, z.ctypes.data, ev.ctypes.data, wit.ctypes.data, None))
        kz["open_division_plus_msm_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
        kz["what"] = "k = 23 (the reference's Halo2 circuit size): ParamsKZG::setup's powers of tau, best_fft, coeff_to_extended, commit, single-point opening; operands resident; host-side window combination included in the MSM figures"
        out["kzg_k23"] = kz
    except Exception as exc:
        out["kzg_k23"] = {"error": repr(exc)}
    ctx.close()
    return out


def aggregate_figure(gl, device, n_ctx=16, log_members=20, sizes=(2, 4, 8, 16, 32, 64, 128)):
    """The reference's own benchmark flow (README.md:167-177, recursion.rs:285-346 `semaphore_aggregation`): N depth-20 Semaphore signals ->
    pairwise aggregation tree of recursive proofs (recursion.rs:187-247) -> final wrap under the BN254-Poseidon config (wrapper.rs:35-56), each
    stage ONE native call (gl355_semaphore_units, gl355_aggregate_units, gl355_circuit_prove_tape).  The level circuits are built once by the
    Python builder (the reference rebuilds them inside every aggregate_signals call), persisted as artifacts, and the timed runs start from the
    artifacts: `cold` = a fresh process state loading them from disk, `warm` = loaded."""
    import shutil
    import tempfile
    sem = importlib.import_module("stark-verifier_amd.semaphore")
    rec = importlib.import_module("stark-verifier_amd.recursion")
    plonk = importlib.import_module("stark-verifier_amd.plonk")
    ctxs = [gl.Context(device) for _ in range(n_ctx)]
    ctx = ctxs[0]
    # witness generation of a node = replaying its circuit's tape over the two inner proofs (host threads inside gl355_circuit_prove_tape_units):
    # GL355_OPT_REPLAY_THREADS per context
    rt = int(os.environ.get("GL355_BENCH_AGG_REPLAY_THREADS", max(1, min(8, host_cores() // 2))))
    for c in ctxs:
        c.set_option(3, rt)
    tmp = tempfile.mkdtemp(prefix="gl355_agg_")
    try:
        rng = np.random.default_rng(0x357)
        sks = gl.api.rand_field(rng, (1 << log_members, 4))
        keys = ctx.hash_no_pad(np.concatenate([sks, np.zeros_like(sks)], axis=1))
        aset = sem.AccessSet(ctx, keys)
        topic = gl.api.rand_field(rng, 4)
        data, rows = aset.build(rng)
        idx, _, _ = aset.witness_rows(rows, sks[0], topic, 0)
        semc = plonk.NativeCircuit(ctx, data.export_blob(idx))
        n_max = max(sizes)

        def signals(n, first):
            leaves, proofs, _ = plonk.semaphore_units(ctxs, semc, None, sks, topic, aset.tree.digests, np.arange(first, first + n, dtype=np.uint64), 7000, want_proofs=True)
            return [(proofs[j], np.concatenate([aset.tree.cap[0], leaves[j]])) for j in range(n)]
        t0 = time.perf_counter()
        agg = rec.Aggregator(ctx, data.common())
        sig = signals(n_max, 0)
        proof, pis, cd = agg.aggregate(sig, seed=100, rng=rng, ctxs=ctxs)               # builds one circuit per level
        wrap = rec.WrapperCircuit(ctx, cd).build([(proof, pis)], rng)
        t_build = time.perf_counter() - t0
        agg.save(tmp)
        artifact_mb = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp)) / 1e6
        out = {"what": "N depth-20 signals -> aggregation tree (N - 1 recursive proofs) -> BN254-Poseidon wrap; seconds on one MI355X, %d prover contexts, "
                       "%d tape-replay threads per context; reference README.md:167-177 (AWS r5.4xlarge, 16 vCPU; its times include rebuilding every circuit)" % (n_ctx, rt),
               "one_off_circuit_build_s": round(t_build, 2), "artifacts_MB": round(artifact_mb, 1), "level_degree_bits": [l.data.degree_bits for l in agg.levels],
               "readme_s": {"2": 11, "4": 29, "8": 64, "16": 128, "32": 235, "64": 468, "128": 930}, "runs": {}}
        # cold: artifacts from disk into a fresh Aggregator, then the largest tree
        t0 = time.perf_counter()
        agg2 = rec.Aggregator.load(ctx, tmp)
        t_load = time.perf_counter() - t0
        t0 = time.perf_counter()
        sig = signals(n_max, 1000)
        t_sig = time.perf_counter() - t0
        t0 = time.perf_counter()
        p2, pi2, _ = agg2.aggregate_native(sig, seed=101, ctxs=ctxs)
        t_tree = time.perf_counter() - t0
        t0 = time.perf_counter()
        wrap.native().prove_tape(ctx, np.concatenate([p2, pi2]), 9)
        t_wrap = time.perf_counter() - t0
        out["cold_%d" % n_max] = {"artifact_load_s": round(t_load, 3), "signals_s": round(t_sig, 3), "tree_s": round(t_tree, 3), "wrap_s": round(t_wrap, 3),
                                   "total_s": round(t_load + t_sig + t_tree + t_wrap, 3)}
        for n in sizes:
            t0 = time.perf_counter()
            sig = signals(n, 2000)
            t1 = time.perf_counter()
            p2, pi2, _, ms = agg2.aggregate_native(sig, seed=102, ctxs=ctxs, timed=True)
            t2 = time.perf_counter()
            out["runs"][str(n)] = {"signals_s": round(t1 - t0, 3), "tree_s": round(t2 - t1, 3), "total_s": round(t2 - t0, 3), "level_ms": [round(v, 1) for v in ms]}
        t0 = time.perf_counter()
        wflat, wpis = wrap.native().prove_tape(ctx, np.concatenate([p2, pi2]), 10)
        out["runs"][str(n_max)]["wrap_s"] = round(time.perf_counter() - t0, 3)
        out["runs"][str(n_max)]["total_with_wrap_s"] = round(out["runs"][str(n_max)]["total_s"] + out["runs"][str(n_max)]["wrap_s"], 3)
        assert np.array_equal(wpis[:4], aset.tree.cap[0]) and wpis.size == 4 + 8 * n_max
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        for c in ctxs:
            c.close()


def halo2_valu(clock_mhz):
    """VALU roofline of the two kernels that carry the k = 23 proof (Fr transform passes, MSM bucket accumulation): dynamic wave instructions
    per launch and their 64-bit share from the committed --pmc pass (profiles/rNN_halo2_k23_pmc_sq.txt), launch time from the committed
    rocprofv3 kernel stats of the same tool, the rest of the mix from the shipped ISA -- the kernel's own body and the Montgomery product it
    calls, weighted so that their 64-bit share matches the counter -- against 1024 SIMDs x the clock sampled during this run's proof."""
    import csv
    import re
    pmc, st, isa = latest_profile("_halo2_k23_pmc_sq.txt"), latest_profile("_halo2_k23_kernel_stats.csv"), latest_profile("_isa_mix.json")
    if not (pmc and st and isa and clock_mhz):
        return None
    isa = json.load(open(isa))["kernels"]
    dur = {re.sub(r"^(void )?gl355::", "", r["Name"]).split("(")[0]: float(r["AverageNs"]) for r in csv.DictReader(open(st))}
    out = {"clock_mhz": clock_mhz, "source": "profiles/%s + %s + %s" % tuple(os.path.basename(x) for x in (pmc, st, latest_profile("_isa_mix.json"))),
           "formula": "insts_per_launch / avg_launch_s against 1024 SIMDs x clock / the pair-aware floor of the kernel's opcode forms (bench_common.ValuModel)", "kernels": {}}
    for line in open(pmc):
        name = re.sub(r"^gl355::", "", line.split("(")[0])
        if name not in ("fr_fft_pass_kernel", "msm_bucket_kernel") or name not in dur or name not in isa:
            continue
        # the transform pass calls the 8 x 32-bit asm product; the bucket loops inline their 29-bit-limb products (the kernel's own histogram)
        callee = {"fr_fft_pass_kernel": "u256 gl355::m_mul<0>"}.get(name, name)
        if callee not in isa:
            continue
        c = {m.group(1): float(m.group(2)) for m in re.finditer(r"(SQ_\w+)=([0-9.e+]+)", line)}
        n, f64 = c.get("SQ_INSTS_VALU"), c.get("SQ_INSTS_VALU_INT64", 0.0) / c.get("SQ_INSTS_VALU", 1.0)
        b, m = isa[name]["f"], isa[callee]["f"]
        al = min(1.0, max(0.0, (f64 - b["mad64"]) / (m["mad64"] - b["mad64"]))) if m["mad64"] != b["mad64"] else 1.0
        mix = {k2: al * m[k2] + (1 - al) * b[k2] for k2 in VALU_CLASSES}
        rest = mix["full32"] + mix["half32"]
        mix = {"mad64": round(f64, 4), "full32": round((1 - f64) * mix["full32"] / rest, 4), "half32": round((1 - f64) * mix["half32"] / rest, 4)}
        import bench_common
        model = bench_common.VALU_MODEL
        fb, fm = (model.static_forms(name), model.static_forms(callee)) if model else (None, None)
        if not (fb and fm):
            continue
        tb, tm = float(sum(fb.values())), float(sum(fm.values()))
        forms = {}
        for src_f, w, t in ((fb, 1 - al, tb), (fm, al, tm)):
            for f, c in src_f.items():
                forms[f] = forms.get(f, 0.0) + w * c / t
        pk = model.peak(forms, clock_mhz)
        ach = n / (dur[name] * 1e-9) / 1e9
        out["kernels"][name] = {"insts_per_launch": n, "avg_launch_ms": round(dur[name] * 1e-6, 4), "int64_share": round(f64, 4), "product_share_of_instructions": round(al, 3),
                                "achieved_ginst_s": round(ach, 1), "peak_ginst_s": pk["peak"], "frac": round(ach / pk["peak"], 4),
                                "clk_per_inst_floor": pk.get("clk_per_inst_floor"), "clk_per_inst_additive": pk["clk_per_inst_additive"]}
    return out


def halo2_figure(gl, device, k=23):
    """SURVEY 8(f) N4 at the reference's size: halo2's create_proof (SHPLONK, Keccak256 transcript; chip/native_chip/test_utils.rs:57-95) over a
    synthetic 2^23-row circuit with the reference's column / gate / lookup shape (tools/halo2_bench.py, stark-verifier_amd/halo2_chips.py), the
    witness resident, per-stage wall milliseconds, the proof checked by the restated verifier (tests/halo2_verifier.py: a checker, not measured)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import halo2_bench
    ctx = gl.Context(device)
    try:
        with ClockSampler(gl, device, period=0.05) as clk:
            out = halo2_bench.run(gl, ctx, int(os.environ.get("GL355_BENCH_HALO2_K", k)))
    finally:
        ctx.close()
    clock = clk.summary()
    out["valu"] = halo2_valu(clock["mean_mhz"] if clock else None)
    out["what"] = ("gl355_plonk_prove: advice commitments, lookup permutation, permutation / lookup grand products, evaluate_h on degree - 1 cosets, "
                   "quotient pieces, evaluations, SHPLONK multi-open; witness synthesis and the Halo2 verifier circuit itself out of scope")
    # no ratio is reported: the reference's 505-511 s (README.md:171-177, k = 23, AWS r5.4xlarge, 16 vCPU) time create_proof_checked
    # (verifier_api.rs:89-92), which also synthesises the whole plonky2-verifier circuit's witness inside the prover and runs verify_proof;
    # the figure here is gl355_plonk_prove on a synthetic witness of the same column shape, already resident in HBM
    out["reference"] = ("README.md:171-177: 505-511 s for create_proof_checked at k = 23 on 16 vCPU -- INCLUDES in-prover witness synthesis of the verifier "
                        "circuit and verify_proof, which this figure does not: different work, no ratio taken")
    return out


def main_exchange(args):
    """The N > 1 plumbing alone, no prover: launch, rendezvous, block partition, one gl355_gather_digests of 64 B per unit per step, rank
    order of the gathered leaves, rank 0 alone printing.  With GL355_BENCH_ONE_DEVICE=1 the communicator is the TCP one and no GPU is
    touched (tests/test_bench_launch.py runs it on CPU); otherwise RCCL, one rank per device.  Not a measurement of anything."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    rehearsal = os.environ.get("GL355_BENCH_ONE_DEVICE") == "1"
    lib = importlib.import_module("stark-verifier_amd._lib").load(init_torch=False)
    par = importlib.import_module("stark-verifier_amd.parallel")
    ctx, dev = None, None
    if not rehearsal:
        import torch
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        ctx = importlib.import_module("stark-verifier_amd").Context(local_rank)
    comm = open_comm(lib, par, ctx, rank, world, rehearsal, dev)
    per = args.proofs_per_step
    total = per * world
    lo, hi = par.shard_range(total, rank, world)
    ok = True
    t0 = time.perf_counter()
    for step in range(args.warmup + args.steps):
        local = (np.arange(lo, hi, dtype=np.uint64)[:, None] * np.uint64(8) + np.arange(8, dtype=np.uint64)[None, :]) + np.uint64(step << 32)
        allv = comm.gather(local) if comm is not None else local
        want = (np.arange(total, dtype=np.uint64)[:, None] * np.uint64(8) + np.arange(8, dtype=np.uint64)[None, :]) + np.uint64(step << 32)
        ok = ok and np.array_equal(allv, want)
    elapsed = time.perf_counter() - t0
    if comm is not None:
        comm.barrier()
        elapsed = comm.max(elapsed)
    if rank == 0:
        print(json.dumps({"metric": "exchange plumbing (no prover)", "value": round(total * (args.warmup + args.steps) / elapsed, 1), "unit": "leaves/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "leaves_in_rank_order": bool(ok),
                          "exchange": getattr(comm, "backend_name", "none (world 1)"),
                          "launcher": "bench.py itself" if os.environ.get("GL355_BENCH_SELF_LAUNCHED") == "1" else "external"}), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()
    if not ok:
        raise SystemExit(4)


def main_lde(args):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gl = importlib.import_module("stark-verifier_amd")
    par = importlib.import_module("stark-verifier_amd.parallel")
    ctx = gl.Context(local_rank)
    lib = ctx.lib
    comm = open_comm(lib, par, ctx, rank, world, False, dev)
    n, N = 1 << LOG_N, 1 << (LOG_N + RATE_BITS)

    # synthetic coefficients, uniform in [0, p) up to the negligible rejection tail (seeded per rank)
    g = torch.Generator(device=dev)
    g.manual_seed(0x355 + rank)
    coeffs = torch.randint(0, (1 << 63) - 1, (BATCH, n), dtype=torch.int64, device=dev, generator=g)
    out = torch.empty((BATCH, N), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    # one step = LDE_PER_STEP LDEs of 135 columns, back to back (the lock-step prover transforms the wires of 8 units per launch
    # sequence); with the default 3 warm-up steps the device's memory-side clocks have settled when the timed region starts
    # (the first ~10 LDEs after something else ran are 15-20 % slower, profiles/r03_ubench_ntt_l24s.txt)
    def step():
        for _ in range(LDE_PER_STEP):
            ctx.check(lib.gl355_lde_bitrev(ctx.h, C.c_void_p(coeffs.data_ptr()), LOG_N, RATE_BITS, 7, BATCH,
                                           C.c_void_p(out.data_ptr())))

    for _ in range(args.warmup):
        step()
    ctx.sync()

    def barrier():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides ----------------------
    ctx.profile_enable(True)
    ctx.profile_read()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    # aggregation root over one digest per rank (the only exchange step of the sharded job)
    digest = torch.empty(4, dtype=torch.int64, device=dev)
    ctx.check(lib.gl355_hash_no_pad(ctx.h, C.c_void_p(out.data_ptr()), 1, 8, C.c_void_p(digest.data_ptr())))
    ctx.sync()
    root = None
    if comm is not None:
        allv = comm.gather(digest.cpu().numpy().view(np.uint64).reshape(1, 4))          # gl355_gather_digests: one digest per rank
        if rank == 0:
            root = [int(x) for x in par.aggregation_root(ctx, allv)[0]]
    barrier()
    t1 = time.perf_counter()
    prof = {k: v for k, v in ctx.profile_read().items() if not k.startswith("host:")}
    ctx.profile_enable(False)

    elapsed = t1 - t0
    if comm is not None:
        elapsed = comm.max(elapsed)

    if rank == 0:
        alg_bytes_lde = 8.0 * BATCH * (n + N)
        alg_bytes_step = alg_bytes_lde * LDE_PER_STEP
        value = alg_bytes_step * args.steps * world / elapsed / 1e9
        # dominant kernel = the kernel group with the largest HIP-event time in the timed region
        dom_name, (dom_cnt, dom_ms, _) = max(prof.items(), key=lambda kv: kv[1][1]) if prof else ("none", (1, 0.0, 0))
        total_kernel_ms = sum(v[1] for v in prof.values())
        # algorithmic bytes of one launch of each pass (DESIGN.md "NTT"): pass 1 reads the n coefficients
        # once and owns the coset expansion; pass 2 turns them into the N evaluations.  A launch of either
        # pass is charged the FULL algorithmic traffic of the LDE it belongs to divided between the two
        # passes in proportion to what each must move at minimum: pass1 = 8*B*n, pass2 = 8*B*N.
        alg_by_kernel = {"ntt_cols_pass1": 8.0 * BATCH * n, "ntt_rows_pass2": 8.0 * BATCH * N,
                         "ntt_rows_single_pass": alg_bytes_lde}
        per_launch_ms = dom_ms / max(1, dom_cnt)
        # roofline of the whole LDE (both passes are needed to produce one unit of output): algorithmic
        # bytes of one LDE over the summed average launch durations of its kernels
        lde_ms = sum(v[1] / max(1, v[0]) for k, v in prof.items() if k.startswith("ntt_"))
        achieved = alg_bytes_lde / (lde_ms * 1e-3) / 1e9 if lde_ms > 0 else 0.0
        line = {
            "metric": "NTT HBM GB/s (2^20-point Goldilocks LDE, blowup 8, bit-exact)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
            "config": {"workload": "lde n=2^17 -> N=2^20 (rate_bits 3, coset 7), batch 135 columns per GPU, "
                                   "bit-reversed (commitment) output order, operands resident in HBM; one step = %d such LDEs back to back" % LDE_PER_STEP,
                       "ldes_per_step": LDE_PER_STEP, "algorithmic_bytes_per_lde": alg_bytes_lde,
                       "algorithmic_bytes_per_step_per_gpu": alg_bytes_step, "parallelism": "independent batches per GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": lde_pmc()[0],
                         "traffic_source": "%s; both kernels of one LDE summed -- the intermediate of the two-pass split is the excess over the algorithmic 1.27 GB" % lde_pmc()[1],
                         "kernel": "lde = ntt_cols_l24s_cosets_kernel<6> (pass 1: 32-point transforms over 64-column tiles on 24-bit limbs, all 8 cosets per block) + "
                                   "ntt_rows_l24s_kernel (pass 2: 4096-point rows as two radix-64 super-rounds on 24-bit limbs)",
                         "dominant_kernel": dom_name,
                         "dominant_avg_launch_ms": round(per_launch_ms, 4),
                         "dominant_alg_GBps": round(alg_by_kernel.get(dom_name, alg_bytes_lde) / (per_launch_ms * 1e-3) / 1e9, 2)
                         if per_launch_ms > 0 else None,
                         "kernels_ms_per_launch": {k: round(v[1] / max(1, v[0]), 4) for k, v in prof.items()},
                         "kernel_time_fraction_of_wall": round(total_kernel_ms * 1e-3 / elapsed, 3)},
        }
        # the other bound: wave-level VALU instructions of the two passes (rocprofv3 --pmc SQ_INSTS_VALU, profiles/r03_lde_pmc.json) against
        # the chip's issue rate for their instruction mix
        insts_lde = lde_pmc()[2] or 0.0
        line["roofline"]["valu_issue"] = {"unit": "G wave-instructions/s", "insts_per_lde": insts_lde,
                                          "achieved": round(insts_lde / (lde_ms * 1e-3) / 1e9, 1) if lde_ms > 0 else None,
                                          "peak": round(1024 * 2.05e9 / 2.95 / 1e9, 1),
                                          "note": "peak = 1024 SIMDs x 2.05 GHz (measured under load with an s_memtime probe) / 2.95 clk, the cost of the limb kernels' mix: "
                                                  "~60 % plain 32-bit add / sub / and / shift-right at ~2.3 clk and ~40 % carry / multiply / 64-bit instructions at ~3.9 clk per wave "
                                                  "instruction (tools/ubench/ubench_alu2.hip read with the real clock); 133 (rows) + 84 (columns) lane-instructions per output "
                                                  "element (162 + 99 on the radix-8 kernels of round 2, ~490 in round 1)"}
        if root is not None:
            line["aggregation_root"] = ["%016x" % x for x in root]
        if world == 1:
            try:
                del coeffs, out
                torch.cuda.empty_cache()
                line["cfg2_sweep"] = cfg2_sweep(ctx, dev)
            except Exception as exc:
                line["cfg2_sweep"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()


def main_semaphore(args):
    """proofs sharded over ranks (recursion.rs:300-308 maps to one block of members per GPU), one RCCL
    all_gather of the (nullifier | topic) leaves, aggregation root on rank 0."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gl = importlib.import_module("stark-verifier_amd")
    par = importlib.import_module("stark-verifier_amd.parallel")
    pr = SemaphoreProvers(gl, local_rank, args.threads)
    comm = open_comm(pr.sets[0].ctx.lib, par, pr.sets[0].ctx, rank, world, False, dev)
    per = args.proofs_per_step
    total = per * world
    lo, hi = par.shard_range(total, rank, world)
    for w in range(args.warmup):
        pr.prove_batch(1000 + lo, hi - lo)

    def barrier():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()
    barrier()
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    root = None
    for step in range(args.steps):
        leaves = pr.prove_batch(2000 + step * total + lo, hi - lo)
        allv = comm.gather(leaves) if comm is not None else leaves
        if rank == 0:
            root = par.aggregation_root(pr.sets[0].ctx, allv)
    barrier()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = comm.max(elapsed)
    if rank == 0:
        line = {"metric": "plonky2 proofs/sec (Semaphore d=20, no recursive wrap)", "value": round(total * args.steps / elapsed, 2),
                "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
                "config": {"workload": "make_signal: group of 2^20 members, %d proofs per GPU per step, %d prover contexts per GPU, "
                                       "all_gather of (nullifier|topic) leaves + Poseidon aggregation root per step" % (per, args.threads)},
                "aggregation_root": ["%016x" % int(x) for x in root[0]]}
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()

