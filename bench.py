#!/usr/bin/env python3
"""Benchmark of the MI355X prover hot path ("Measurement").

  python bench.py --gpus N --steps K --warmup W        (N > 1: under torch.distributed.run, or plain -- it then starts its N ranks itself)

Default workload `recursive` (BASELINE.json metric "recursive plonky2 proofs/sec (Semaphore d=20)", configs[3]; sharded
over ranks it is configs[4]): one UNIT = one depth-20 Semaphore signal (make_signal, access_set.rs:61-104: witness, proof at
n = 2^13, blowup 8, 28 queries, 16 PoW bits, zero-knowledge) PLUS the recursive proof that verifies it in-circuit
(wrapper.rs:35-56 with the Poseidon-Goldilocks config; witness by tape replay, proof at n = 2^14).  One step = U units per
GPU, proven by K concurrent prover contexts (one HIP stream + one host thread each), followed by the gather of the
(nullifier | topic) leaves and the Poseidon aggregation root -- the job's only exchange (RCCL all_gather, 64 B per unit).
value = units (= recursive proofs) per second of the whole job.  The roofline object is the kernel group with the largest
HIP-event time inside the timed region: algorithmic bytes of its launches over their summed duration.

`--workload lde` (configs[1]): 2^20-point Goldilocks LDE, blowup 8, 135 columns, value in algorithmic GB/s; also run for a
few steps after the default workload and reported as `ntt_lde` (the metric's "NTT HBM GB/s" half).
`--workload semaphore`: the signals alone, no recursive proof.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N, RATE_BITS, BATCH = 17, 3, 135
LDE_PER_STEP = 8      # --workload lde: LDEs of BATCH columns per step
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def host_cores():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a 256-thread host with a 128-CPU
    quota runs 256 OpenMP threads ten times slower than 128)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    return n


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


class RecursiveProvers:
    """K prover contexts (one HIP stream + one host thread each) on one GPU; a unit = a Semaphore signal + the recursive proof
    verifying it (wrapper.rs:35-56 over the Poseidon-Goldilocks config).  Both circuits are built once, exported as circuit
    artifacts and loaded into the library (gl355_circuit_load); per unit a host thread makes two native calls:
    gl355_semaphore_prove (witness + proof, n = 2^13) and gl355_circuit_prove_tape (tape replay + proof, n = 2^14); a whole step
    is one call into the native batch runtime (gl355_semaphore_units), which runs those host threads."""

    def __init__(self, gl, device, threads, log_members=20, seed=0x357, blocking_sync=False, replay_threads=1):
        rand_field = gl.api.rand_field
        sem = importlib.import_module("stark-verifier_amd.semaphore")
        rec = importlib.import_module("stark-verifier_amd.recursion")
        self.plonk = importlib.import_module("stark-verifier_amd.plonk")
        rng = np.random.default_rng(seed)
        self.sets = [gl.Context(device) for _ in range(threads)]      # one prover context per host thread
        if blocking_sync:
            for c in self.sets:
                c.set_option(2, int(blocking_sync))                  # GL355_OPT_BLOCKING_SYNC: 1 blocking event, 2 poll + back-off
        if replay_threads > 1:                                       # GL355_OPT_REPLAY_THREADS
            for c in self.sets:
                c.set_option(3, replay_threads)
        if os.environ.get("GL355_BENCH_NTT_SINGLE_MAX"):             # experiments: GL355_OPT_NTT_SINGLE_PASS_MAX_LOG
            for c in self.sets:
                c.set_option(4, int(os.environ["GL355_BENCH_NTT_SINGLE_MAX"]))
        if os.environ.get("GL355_BENCH_BATCH_UNITS"):                # GL355_OPT_BATCH_UNITS: units a context proves in lock-step
            for c in self.sets:
                c.set_option(5, int(os.environ["GL355_BENCH_BATCH_UNITS"]))
        if os.environ.get("GL355_BENCH_DEVICE_REPLAY"):              # GL355_OPT_DEVICE_REPLAY (default on): witness tape on the device
            for c in self.sets:
                c.set_option(6, int(os.environ["GL355_BENCH_DEVICE_REPLAY"]))
        if os.environ.get("GL355_BENCH_LANES_LOG"):                  # experiments: GL355_OPT_MERKLE_LANES_LOG
            for c in self.sets:
                c.set_option(1, int(os.environ["GL355_BENCH_LANES_LOG"]))
        c0 = self.sets[0]
        self.sks = rand_field(rng, (1 << log_members, 4))
        keys = c0.hash_no_pad(np.concatenate([self.sks, np.zeros_like(self.sks)], axis=1))
        self.topic = rand_field(rng, 4)
        self.aset = sem.AccessSet(c0, keys)
        self.root = self.aset.tree.cap[0].copy()
        self.height = self.aset.tree_height()
        data, rows = self.aset.build(np.random.default_rng(1))
        idx, vals, pi = self.aset.witness_rows(rows, self.sks[0], self.topic, 0)
        self.inner_data, self.inner_rows = data, (idx, vals, pi)
        self.sem = self.plonk.NativeCircuit(c0, data.export_blob(idx))
        flat, pis = self.sem.semaphore_prove(c0, self.sks[0], self.topic, 0, self.aset.tree.prove_host(0), 1)
        self.rc = rec.RecursiveCircuit(c0, data.common(), k=1).build([(flat, pis)], np.random.default_rng(2))
        self.nat = self.rc.native()
        self.last = None
        self.units_done = [0] * threads
        for t in range(threads):                              # warm every context's allocator / tables
            self.unit(t, t)

    def unit(self, t, i):
        ctx = self.sets[t]
        flat, pis = self.sem.semaphore_prove(ctx, self.sks[i], self.topic, i, self.aset.tree.prove_host(i), 0x358 + i)
        inputs = np.concatenate([flat, pis])
        outer, opis = self.nat.prove_tape(ctx, inputs, 0x359 + i)
        self.last = ((flat, pis), outer, opis)
        self.units_done[t] += 1
        return opis[4:12]                                     # nullifier | topic, re-exposed by the recursive proof

    def prove_batch(self, first, count):
        """`count` units starting at member `first` through the native batch runtime (gl355_semaphore_units: one host thread
        per prover context inside the library); returns the (nullifier | topic) leaves [count][8]"""
        n = self.sks.shape[0]
        members = (first + np.arange(count, dtype=np.uint64)) % np.uint64(n)
        # blinding key None = the production setting: every proof draws a fresh 256-bit key from the OS CSPRNG inside the library
        leaves, _, per = self.plonk.semaphore_units(self.sets, self.sem, self.nat, self.sks, self.topic, self.aset.tree.digests, members,
                                                    None)
        for t, k in enumerate(per):
            self.units_done[t] += k
        return leaves

    def profile(self, on, contexts=1):
        """HIP-event scopes on the first `contexts` prover contexts only: an event pair per launch on all 12 streams costs
        ~10 % throughput (extra barrier packets between back-to-back kernels); one stream's launches are a 1/12 sample of
        the same timed region."""
        self.prof_ctx = list(range(min(contexts, len(self.sets)))) if on else []
        for t, c in enumerate(self.sets):
            c.profile_enable(on and t in self.prof_ctx)
            c.profile_read()
        self.units_mark = list(self.units_done)

    def profile_read(self):
        """({kernel group: (launches, ms, algorithmic bytes)}, units proven by the profiled contexts since profile(True))"""
        agg = {}
        for t in self.prof_ctx:
            for name, (cnt, ms, nbytes) in self.sets[t].profile_read().items():
                c0, m0, b0 = agg.get(name, (0, 0.0, 0))
                agg[name] = (c0 + cnt, m0 + ms, b0 + nbytes)
        return agg, sum(self.units_done[t] - self.units_mark[t] for t in self.prof_ctx)


def cpu_baseline_recursive(pr, units=3):
    """The CPU restatement of prove() (oracle/gl_prover.c, OpenMP) on the same two circuits and witnesses: `units` signals +
    recursive proofs on every usable host core, proofs only (the witnesses are handed over ready-made); then the per-core figure
    BASELINE.md 3 asks for: one Semaphore proof (n = 2^13) on ONE thread and on all threads (a whole unit on one thread is ~1 min,
    outside the bench's budget)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import CpuProver, Oracle
    orc = Oracle()
    orc.L.orc_set_num_threads(host_cores())      # torchrun exports OMP_NUM_THREADS=1; the baseline uses every usable host core
    threads = orc.L.orc_num_threads()
    t_build = time.perf_counter()
    cpu_in = CpuProver.from_circuit_data(orc, pr.inner_data)
    cpu_out = CpuProver.from_circuit_data(orc, pr.rc.data)
    t_build = time.perf_counter() - t_build
    ctx = pr.sets[0]
    idx, vals, pi = pr.inner_rows
    inner = pr.last[0]
    wrows, wpis = pr.rc.witness([inner])
    t0 = time.perf_counter()
    t_inner_all = 0.0
    for u in range(units):
        ti = time.perf_counter()
        flat_in = cpu_in.prove_sparse(idx, vals, pi, 7 + u)
        t_inner_all += time.perf_counter() - ti
        flat_out = cpu_out.prove_sparse(pr.rc.row_idx, wrows, wpis, 9 + u)
    dt = time.perf_counter() - t0
    # bit-exactness of the product against this baseline on the very same inputs (seed of the last unit)
    g_in = pr.sem.prove_rows(ctx, vals, pi, 7 + units - 1)
    g_out, _ = pr.nat.prove_tape(ctx, np.concatenate([inner[0], inner[1]]), 9 + units - 1)
    same = bool(np.array_equal(g_in, flat_in) and np.array_equal(g_out, flat_out))
    # one thread: the Semaphore proof alone (bounded sample)
    orc.L.orc_set_num_threads(1)
    t1 = time.perf_counter()
    flat_1 = cpu_in.prove_sparse(idx, vals, pi, 7 + units - 1)
    t1 = time.perf_counter() - t1
    orc.L.orc_set_num_threads(threads)
    same = same and bool(np.array_equal(flat_1, flat_in))
    return {"value": round(units / dt, 4), "unit": "recursive proofs/s", "cores": int(threads), "kind": "port",
            "byte_identical_to_gpu_proofs": same,
            "one_thread": {"value": round(1.0 / t1, 4), "unit": "Semaphore proofs/s (n=2^13, no recursive proof)", "cores": 1,
                           "same_proof_all_threads_per_s": round(units / t_inner_all, 4),
                           "parallel_speedup": round(t1 / (t_inner_all / units), 2),
                           "sample": "1 Semaphore proof, %.1f s on one thread; the same proof on %d threads: %.2f s" % (t1, threads, t_inner_all / units)},
            "sample": "%d unit(s): Semaphore proof (n=2^13) + recursive proof (n=2^%d) by the C restatement of plonky2's prove() "
                      "(oracle/gl_prover.c, OpenMP, %d threads), witnesses given, preprocessed commitments prebuilt (%.1f s, untimed); "
                      "%.2f s wall.  Not the Rust binary (no Rust toolchain here); reference README: ~0.14 recursive proofs/s "
                      "on 16 vCPU" % (units, pr.rc.data.degree_bits, threads, t_build, dt)}


def latest_profile(suffix):
    """profiles/rNN<suffix> of the latest round that has one (bench.py cannot collect PMC counters itself: they come from the
    committed rocprofv3 --pmc passes)"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]" + suffix)))
    return c[-1] if c else None


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (bench.py cannot collect PMC counters
    itself); None when no pass covers the kernel."""
    path = latest_profile("_pmc_traffic.json")
    try:
        d = json.load(open(path))
        return d["kernels"][kernel]["bytes_per_launch_corrected"], "profiles/%s: %s" % (os.path.basename(path), d["_source"])
    except Exception:
        return None, None


N_SIMD = 1024                     # 256 CUs x 4 SIMDs
VALU_CLASSES = ("full32", "half32", "mad64")


def valu_probe(ctx):
    """gl355_valu_probe (csrc/valu_probe.hip) on this device, in this run: per instruction class the chip-wide issue rate of a kernel that
    only issues that class (G wave-instructions/s), the shader clock read inside that kernel, and the cost in shader cycles per wave
    instruction per SIMD that follows from the two (no assumed frequency anywhere)."""
    rates = (C.c_double * 3)()
    mhz = (C.c_double * 3)()
    ctx.check(ctx.lib.gl355_valu_probe(ctx.h, rates, mhz))
    return {c: {"rate_ginst_s": round(rates[i], 1), "shader_mhz": round(mhz[i]),
                "clk_per_wave_inst_per_simd": round(mhz[i] * 1e6 * N_SIMD / (rates[i] * 1e9), 3) if rates[i] > 0 else None}
            for i, c in enumerate(VALU_CLASSES)}


class ClockSampler:
    """shader clock during the timed region: gl355_clock_probe (one sleeping wave for 2 ms) on a context of its own every ~100 ms"""

    def __init__(self, gl, device, period=0.1):
        import threading
        self.ctx = gl.Context(device)
        self.period = period
        self.samples, self.stop = [], threading.Event()
        self.thread = threading.Thread(target=self.run, daemon=True)

    def run(self):
        v = C.c_double(0)
        while not self.stop.is_set():
            if self.ctx.lib.gl355_clock_probe(self.ctx.h, 2000, C.byref(v)) == 0 and v.value > 0:
                self.samples.append(v.value)
            self.stop.wait(self.period)

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, *a):
        self.stop.set()
        self.thread.join()
        self.ctx.close()

    def summary(self):
        if not self.samples:
            return None
        # a probe wave descheduled in mid-sleep (two processes on one device) reads a nonsense ratio: samples beyond 1.25x the
        # median are dropped and counted
        med = sorted(self.samples)[len(self.samples) // 2]
        kept = [s for s in self.samples if s <= 1.25 * med]
        return {"mean_mhz": round(sum(kept) / len(kept)), "min_mhz": round(min(kept)), "max_mhz": round(max(kept)),
                "samples": len(kept), "dropped": len(self.samples) - len(kept)}


def valu_mix(kernel=None):
    """Instruction-class fractions of `kernel` (None: of the whole unit, every kernel weighted by its dynamic instruction count).
    mad64 share: dynamic, SQ_INSTS_VALU_INT64 / SQ_INSTS_VALU of the committed --pmc pass when it carries those counters; the rest
    splits into full32 / half32 as the kernel's shipped ISA does (profiles/rNN_isa_mix.json, tools/isa_mix.py).  Without the INT64
    counters everything comes from the static histogram.  -> (fractions, dynamic VALU instructions per launch or per unit, source)"""
    try:
        pmc = json.load(open(latest_profile("_pmc_traffic.json")))
        isa = json.load(open(latest_profile("_isa_mix.json")))["kernels"]
    except Exception:
        return None, None, None
    src = "profiles/%s + profiles/%s" % (os.path.basename(latest_profile("_pmc_traffic.json")), os.path.basename(latest_profile("_isa_mix.json")))

    def static_f(name):
        k = isa.get(name)
        if k is None:      # template instances: name<...>
            cands = [v for n, v in isa.items() if n.split("<")[0] == name]
            if not cands:
                return None
            tot = sum(v["valu_static"] for v in cands)
            return {c: sum(v[c] for v in cands) / tot for c in VALU_CLASSES}
        return dict(k["f"])

    def one(name, e):
        f = static_f(name.split("<")[0]) or {"full32": 0.11, "half32": 0.36, "mad64": 0.53}
        n = e.get("valu_insts_per_launch")
        i64 = e.get("valu_int64_per_launch")
        if n and i64 is not None:
            rest = f["full32"] + f["half32"]
            m = i64 / n
            f = {"mad64": m, "full32": (1 - m) * f["full32"] / rest, "half32": (1 - m) * f["half32"] / rest}
        return f, n
    if kernel is not None:
        e = pmc["kernels"].get(kernel)
        if not e or not e.get("valu_insts_per_launch"):
            return None, None, None
        f, n = one(kernel, e)
        return {c: round(f[c], 4) for c in VALU_CLASSES}, n, src
    job = pmc.get("job")
    if not job:
        return None, None, None
    tot, acc = 0.0, {c: 0.0 for c in VALU_CLASSES}
    for name, e in pmc["kernels"].items():
        if not e.get("valu_insts_per_launch") or name.startswith("vp_"):
            continue
        f, n = one(name, e)
        w = n * e.get("sq_launches", e.get("launches", 0))
        tot += w
        for c in VALU_CLASSES:
            acc[c] += w * f[c]
    if tot <= 0:
        return None, None, None
    return {c: round(acc[c] / tot, 4) for c in VALU_CLASSES}, job["valu_insts_per_unit"], src


# Issue cost of a wave64 instruction on one SIMD, in shader cycles, by class: the hardware's own figures (MI355X_MICROARCH.md, "Wave scheduling":
# four SIMD-32 units per CU, a wave issues a VALU instruction over 2 cycles; the multiply / carry / 64-bit class goes at half that rate).  The
# probe kernels of csrc/valu_probe.hip measure the same classes on the device (2.8 / 4.7 / 4.7 in profiles/r04_valu_probe.json) and cannot beat
# these; the job itself issues FASTER than the probes reach, so the probes are a cross-check of the classes, not the ceiling.
NOMINAL_CLK = {"full32": 2.0, "half32": 4.0, "mad64": 4.0}


def valu_peak(mix, clock_mhz):
    """G wave-instructions/s the chip can issue at `clock_mhz` for instructions that split as `mix`: 1024 SIMDs x clock / the mix-weighted
    issue cost.  No kernel can exceed it (every class priced at the hardware's issue rate), so achieved / peak <= 1 by construction."""
    cost = sum(mix[c] * NOMINAL_CLK[c] for c in VALU_CLASSES)
    return N_SIMD * clock_mhz * 1e6 / cost / 1e9


def valu_peak_probe(mix, classes, clock_mhz=None):
    """the same with the class rates the probe kernels reached in this run (moved to `clock_mhz` if given)"""
    t = 0.0
    for c in VALU_CLASSES:
        r = classes[c]["rate_ginst_s"]
        if clock_mhz and classes[c]["shader_mhz"]:
            r = r * clock_mhz / classes[c]["shader_mhz"]
        if r <= 0:
            return None
        t += mix[c] / r
    return 1.0 / t if t > 0 else None


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
    its VALU instructions (rocprofv3 --pmc SQ_INSTS_VALU) at the issue cost of the pass's instruction mix (tools/isa_mix.py classes at 2 / 4 clk)
    and the clock sampled during this run.  A pass cannot beat the larger of its two floors even with perfect overlap; the sum over the passes is the
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
        f = isa.get(name, {}).get("f")
        if not f:
            continue
        cost = sum(f[c] * NOMINAL_CLK[c] for c in VALU_CLASSES)
        valu_ms = e["SQ_INSTS_VALU"] * cost / (N_SIMD * clk["mean_mhz"] * 1e6) * 1e3
        mem_ms = e["hbm_bytes_per_launch"] / copy_peak * 1e3
        out["passes"][name] = {"valu_floor_ms": round(valu_ms, 3), "mem_floor_ms": round(mem_ms, 3), "mix": f, "clk_per_inst": round(cost, 2)}
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


def merkle_figures(gl, device):
    """BASELINE configs[2] / SURVEY cfg-3: MerkleTree::new over 2^22 leaves -- (L = 4, cap 4) the FRI-layer shape, (L = 135, cap 4)
    the wires-like shape (4.5 GB of leaves) -- and the Semaphore group tree (2^20 leaves, L = 4, cap 0, signal.rs:40); row-major leaves
    resident in HBM, gl355_merkle_build, HIP-event time per build.  The kernels are integer-VALU work (a permutation is 17.6 k VALU
    instructions): `valu_frac` prices the build's permutations at that count against the chip's issue rate."""
    import torch
    ctx = gl.Context(device)
    lib = ctx.lib
    g = torch.Generator(device="cuda")
    g.manual_seed(0x356)
    out = {"what": "gl355_merkle_build (leaf sponge + compression levels + cap, plonky2 digest layout), leaves resident, HIP events"}
    # the VALU roofline of these builds: 17 600 instructions per permutation and lane against the issue rate for hash_leaves_kernel's
    # instruction mix (valu_peak) at the shader clock sampled while the builds run
    mix, _, mix_src = valu_mix("hash_leaves_kernel")
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
        if mix and clk:
            ach_v = perms * 17600 / 64.0 / (ms * 1e-3) / 1e9
            peak_v = valu_peak(mix, clk["mean_mhz"])
            out["N=2^%d L=%d cap=%d" % (log_n, L, cap)]["valu"] = {
                "bound": "valu", "achieved": round(ach_v, 1), "peak": round(peak_v, 1), "unit": "G wave-instructions/s", "frac": round(ach_v / peak_v, 4),
                "clock_mhz": clk["mean_mhz"], "mix": mix, "formula": "permutations x 17 600 / 64 / time against 1024 SIMDs x clock / sum_c mix[c] x nominal_clk[c]"}
        del leaves, dig, capb
        torch.cuda.empty_cache()
    ctx.close()
    return out


class _TorchComm:
    """stand-in with the Comm interface over torch.distributed -- used ONLY if some rank cannot bind librccl for the gl355
    communicator (reported in the JSON line as config.exchange); the product's exchange is gl355_gather_digests"""

    def __init__(self, dist, dev):
        self.dist, self.dev, self.backend_name = dist, dev, "torch.distributed (gl355 RCCL communicator unavailable on some rank)"

    def gather(self, local):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.uint64).view(np.int64)).to(self.dev)
        parts = [torch.empty_like(t) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(parts, t)
        return torch.cat(parts, dim=0).cpu().numpy().view(np.uint64)

    def barrier(self):
        self.dist.barrier()

    def max(self, v):
        import torch
        t = torch.tensor([v], dtype=torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        self.dist.destroy_process_group()


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
           "formula": "insts_per_launch / avg_launch_s against 1024 SIMDs x clock / sum_c mix[c] x nominal_clk[c]", "kernels": {}}
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
        peak = valu_peak(mix, clock_mhz)
        ach = n / (dur[name] * 1e-9) / 1e9
        out["kernels"][name] = {"insts_per_launch": n, "avg_launch_ms": round(dur[name] * 1e-6, 4), "mix": mix, "product_share_of_instructions": round(al, 3),
                                "achieved_ginst_s": round(ach, 1), "peak_ginst_s": round(peak, 1), "frac": round(ach / peak, 4)}
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


def thread_cpu_snapshot():
    """{tid: (comm, cpu seconds)} of the process's live threads (diagnostic: which threads burn host CPU; GL355_BENCH_THREAD_CPU=1)"""
    out = {}
    tck = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open("/proc/self/task/%s/stat" % tid).read()
            comm = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(f[11]) + int(f[12])) / tck)
        except Exception:
            pass
    return out


_STORE_KEEPALIVE = []


def open_comm(lib, par, ctx, rank, world, rehearsal, dev=None):
    """The exchange of the N > 1 job through the C ABI (gl355_comm_*): RCCL over xGMI, or TCP between the host processes in the
    one-device rehearsal.  The 128-byte communicator id travels through the launcher's key-value store (torchrun's TCPStore,
    MASTER_ADDR / MASTER_PORT) -- plumbing a Rust host would do with its own launcher; no torch collective is involved."""
    if world == 1:
        return None
    from datetime import timedelta
    from torch.distributed import PrefixStore, TCPStore
    addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    store = TCPStore(addr, port, world, (rank == 0 and not agent_store), timedelta(seconds=300), multi_tenant=True)
    # rank 0 may be the store's server (no agent store): the server must outlive every other rank's reads of the flags below, so the object is
    # kept for the life of the process (all ranks pass a communicator barrier before they exit)
    _STORE_KEEPALIVE.append(store)
    store = PrefixStore("gl355_bench/%s" % os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"), store)
    backend = par.COMM_HOST if rehearsal else par.COMM_RCCL
    # phase 1: can every rank bind its communicator library?  (a rank that cannot must not leave the others inside ncclCommInitRank)
    ok, cid, why = True, b"", ""
    try:
        if backend == par.COMM_HOST:
            import socket
            if rank == 0:
                s = socket.socket(); s.bind((addr if addr[0].isdigit() else "127.0.0.1", 0)); hp = s.getsockname()[1]; s.close()
                cid = par.Comm.unique_id(lib, backend, addr if addr[0].isdigit() else "127.0.0.1", hp)
        else:
            cid = par.Comm.unique_id(lib, backend)          # every rank: proves librccl binds here; rank 0's id is the one used
    except Exception as exc:
        ok, why = False, repr(exc)
    store.set("ok/%d" % rank, b"1" if ok else why.encode()[:200] or b"0")
    if rank == 0 and ok:
        store.set("id", cid)
    flags = [bytes(store.get("ok/%d" % r)) for r in range(world)]
    created_failed = False
    if all(f == b"1" for f in flags):
        # phase 2: the communicator itself (ncclCommInitRank is collective).  Every rank reports whether it came up; the job uses it only if
        # all did -- a communicator that exists on some ranks only would hang the first gather
        comm, err = None, ""
        try:
            if os.environ.get("GL355_BENCH_FORCE_COMM_FAIL") == "1":      # test hook for the fall-back below
                raise RuntimeError("forced failure (GL355_BENCH_FORCE_COMM_FAIL)")
            comm = par.Comm(ctx, backend, bytes(store.get("id")), rank, world, lib=lib)
        except Exception as exc:
            err = repr(exc)
        store.set("up/%d" % rank, b"1" if comm is not None else (err.encode()[:200] or b"0"))
        flags = [bytes(store.get("up/%d" % r)) for r in range(world)]
        if all(f == b"1" for f in flags):
            comm.backend_name = "gl355_gather_digests over RCCL (ncclAllGather)" if backend == par.COMM_RCCL else "gl355_gather_digests over TCP (one-device rehearsal)"
            return comm
        if comm is not None:
            comm.close()
        created_failed = True
    why = [f for f in flags if f != b"1"][:1]
    # --gpus N > 1 measures the RCCL exchange of SURVEY 8(e): when the RCCL communicator cannot be created on every rank the run FAILS
    # (every rank sees the same flags, so every rank exits) instead of quietly measuring something else.  GL355_BENCH_ALLOW_STANDIN=1
    # (never set by the driver) lets the 64-byte-per-unit exchange run over torch.distributed instead; the line then says so in
    # config.exchange.  The one-device rehearsal never had RCCL to begin with.
    if not rehearsal and os.environ.get("GL355_BENCH_ALLOW_STANDIN") != "1":
        sys.stderr.write("[bench] rank %d: the RCCL communicator (gl355_comm_create) is unavailable: %s -- refusing to substitute another "
                         "exchange (set GL355_BENCH_ALLOW_STANDIN=1 to allow the torch.distributed stand-in)\n" % (rank, why))
        sys.stderr.flush()
        os._exit(3)
    sys.stderr.write("[bench] rank %d: gl355 communicator unavailable (%s); using torch.distributed for the exchange\n" % (rank, why))
    import torch.distributed as dist
    # the exchange is 64 bytes per unit: when the RCCL communicator could not be CREATED (rather than librccl not binding), RCCL itself is
    # suspect, so the stand-in runs over gloo on host tensors
    if rehearsal or created_failed:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        c = _TorchComm(dist, "cpu")
        c.backend_name = "torch.distributed gloo on host tensors (gl355 communicator could not be created on some rank)" if created_failed else c.backend_name
        return c
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    return _TorchComm(dist, dev)


def main_recursive(args):
    """units sharded over ranks (recursion.rs:300-308: one block of members per GPU), no collective on the data path; one RCCL
    all_gather of the (nullifier | topic) leaves per step and the aggregation root on rank 0."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    # rehearsal of the N > 1 path on a box with fewer GPUs than ranks: GL355_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # uses gloo for the gather (NCCL refuses two ranks on one device); never set by the driver
    rehearsal = os.environ.get("GL355_BENCH_ONE_DEVICE") == "1"
    # one prover context = one HIP stream + one host thread.  With a core per context the threads spin in hipStreamSynchronize
    # (lowest latency); with fewer usable cores per rank (cgroup quota / ranks) than contexts every device wait sleeps instead,
    # so the contexts still keep the GPU fed (host work is ~10-12 ms of ~90 ms per unit and context)
    # ranks of THIS node share its cores (multi-node jobs: LOCAL_WORLD_SIZE ranks per node, not WORLD_SIZE)
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    cores_per_rank = max(1, host_cores() // local_world)
    # N > 1: every rank pins itself to its own contiguous slice of the CPUs the job may use (rank r of the node -> slice r), before any
    # thread exists, so the 8 x (prover contexts + replay threads + the HIP runtime's own thread) of a node do not migrate over each other's
    # cores; contiguous CPU numbers share a socket on the usual numbering, i.e. GPU r's slice sits on the socket of GPUs 4 (r // 4) .. + 3.
    # GL355_BENCH_NO_PIN=1 leaves the affinity alone.
    pinned = None
    if world > 1 and os.environ.get("GL355_BENCH_NO_PIN") != "1":
        cpus = sorted(os.sched_getaffinity(0))
        k = len(cpus) // local_world
        if k >= 1:
            mine = cpus[local_rank * k:(local_rank + 1) * k]
            try:
                os.sched_setaffinity(0, mine)
                pinned = "%d-%d" % (mine[0], mine[-1])
                cores_per_rank = min(cores_per_rank, len(mine))
            except OSError:
                pinned = None
    n_threads = max(1, int(os.environ.get("GL355_BENCH_CONTEXTS", args.threads)))
    # how a context's host thread waits for its stream: "spin" (hipStreamSynchronize, a core per context), "poll" (GL355_OPT_BLOCKING_SYNC
    # = 2: hipStreamQuery + 30-us sleeps, a few percent of a core per context), "sleep" (hipDeviceScheduleBlockingSync for the device)
    # (round 3, profiles/r03b_wait_modes.txt: host replay + poll 290 units/s at 13.1 ms of host CPU per unit; device replay + poll 282 at 9.2;
    # device replay + sleeping waits 282 at 6.1 -- the setting of ranks with fewer than 4 cores)
    wait_mode = os.environ.get("GL355_BENCH_WAIT", "sleep" if (os.environ.get("GL355_BENCH_SLEEP_WAITS") == "1" or cores_per_rank < 4) else
                               "poll")
    sleeping_waits = wait_mode == "sleep"
    # gl355_runtime_config: hardware queues per context (unless GPU_MAX_HW_QUEUES is already set) and, if asked, sleeping waits
    # (hipDeviceScheduleBlockingSync); it has to run before the device's HIP context exists, i.e. before torch touches the GPU
    lib = importlib.import_module("stark-verifier_amd._lib").load(init_torch=False)
    if lib.gl355_runtime_config(0 if rehearsal else local_rank, n_threads, 1 if sleeping_waits else 0) != 0:
        raise SystemExit("gl355_runtime_config failed (HIP runtime already initialised?)")
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gl = importlib.import_module("stark-verifier_amd")
    par = importlib.import_module("stark-verifier_amd.parallel")
    blocking = sleeping_waits
    # the witness tape of the recursive proof is host work inside each context's thread (7 ms, the context's stream idles
    # meanwhile); its FRI-query segments replay on 2 threads when the waits sleep and cores are to spare (191 -> 195 proofs/s)
    replay_threads = int(os.environ.get("GL355_BENCH_REPLAY_THREADS", 2 if cores_per_rank >= 8 else 1))
    # witness generation of the recursive circuit: on the device (tape interpreter on the context's side stream: 2 host cores per
    # rank are enough, 255 units/s) or on host threads (4 % more throughput when the rank has cores to spare: 270 vs 259 units/s)
    os.environ.setdefault("GL355_BENCH_DEVICE_REPLAY", "0" if cores_per_rank >= 8 else "1")
    pr = RecursiveProvers(gl, local_rank, n_threads, args.log_members, replay_threads=replay_threads, blocking_sync=2 if wait_mode == "poll" else 0)
    comm = open_comm(lib, par, pr.sets[0], rank, world, rehearsal, dev)
    per = args.proofs_per_step
    total = per * world
    lo, hi = par.shard_range(total, rank, world)
    for w in range(args.warmup):
        pr.prove_batch(1000 + w * total + lo, hi - lo)
        if comm is not None:
            comm.gather(np.zeros((hi - lo, 8), dtype=np.uint64))       # warm the communicator's first-use set-up as well

    def barrier():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()
    # the VALU roofline's peak: class issue rates measured on this device now (idle apart from the probe), the clock during the timed
    # region sampled by a one-wave probe on a context of its own (rank 0)
    classes = None
    if rank == 0:
        try:
            classes = valu_probe(pr.sets[0])
        except Exception as exc:
            sys.stderr.write("[bench] valu probe failed: %r\n" % (exc,))
    sampler = ClockSampler(gl, local_rank) if rank == 0 and os.environ.get("GL355_BENCH_NO_CLOCK_SAMPLER") != "1" else None
    pr.profile(True)
    barrier()
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    th0 = thread_cpu_snapshot() if os.environ.get("GL355_BENCH_THREAD_CPU") else None
    if sampler is not None:
        sampler.__enter__()
    t0 = time.perf_counter()
    root = None
    for step in range(args.steps):
        leaves = pr.prove_batch(5000 + step * total + lo, hi - lo)
        allv = comm.gather(leaves) if comm is not None else leaves          # gl355_gather_digests: 64 B per unit, rank order
        if rank == 0:
            root = par.aggregation_root(pr.sets[0], allv)                   # gl355_aggregation_root
    barrier()
    elapsed = time.perf_counter() - t0
    job_clock = None
    if sampler is not None:
        sampler.__exit__()
        job_clock = sampler.summary()
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    host_cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)   # rank 0's process, spinning waits included
    if th0 is not None:
        th1 = thread_cpu_snapshot()
        d = sorted(((th1[t][1] - th0.get(t, (None, 0.0))[1], th1[t][0], t) for t in th1), reverse=True)[:10]
        sys.stderr.write("[bench] process CPU %.2f s over %.2f s wall; long-lived threads: %s\n" % (host_cpu_s, elapsed, ["%s:%.2fs" % (c, v) for v, c, _ in d]))
    prof, local_units = pr.profile_read()
    pr.profile(False)
    if comm is not None:
        elapsed = comm.max(elapsed)                                         # gl355_comm_max_f64
    if rank == 0:
        units = total * args.steps
        # ---- roofline pass: the same units on ONE prover context right after the timed region.  With 12 streams sharing the
        # GPU a HIP-event pair around a launch also measures the time the launch waits behind other streams' kernels
        # (3-4x the kernel's own duration, and rocprofv3's per-kernel durations do not see that wait), so per-kernel
        # durations are taken where they are attributable: one stream, nothing else on the device.
        all_sets = pr.sets
        pr.sets = all_sets[:1]
        pr.profile(True)
        pr.prove_batch(9000, 8)
        t_iso = time.perf_counter()
        pr.prove_batch(9100, 8)
        t_iso = time.perf_counter() - t_iso
        iso, iso_units = pr.profile_read()
        iso_units //= 2
        iso = {k: (v[0] // 2, v[1] / 2, v[2] // 2) for k, v in iso.items()}      # two 8-unit batches, the second one timed on the host
        pr.profile(False)
        # BASELINE configs[3] as stated: ONE proof end to end.  One context, one unit per call (no lock-step partners, no pipelining):
        # gl355_semaphore_prove (witness + proof, n = 2^13) then gl355_circuit_prove_tape (tape replay + proof), host-visible wall time
        # (a lone unit has the rank's host cores to itself: its witness tape replays on up to 8 threads instead of the throughput setting)
        # and polls its stream without sleeping: GL355_OPT_BLOCKING_SYNC 3)
        lat_rt = max(1, min(8, cores_per_rank // 2))
        all_sets[0].set_option(3, lat_rt)
        all_sets[0].set_option(2, 3)
        lat = []
        for k in range(6):
            t_l = time.perf_counter()
            pr.unit(0, 9200 + k)
            lat.append(time.perf_counter() - t_l)
        lat = sorted(lat[1:])
        t_b1 = time.perf_counter()
        pr.prove_batch(9300, 1)                                                  # the same through the batch runtime (one unit)
        t_b1 = time.perf_counter() - t_b1
        all_sets[0].set_option(3, replay_threads)
        all_sets[0].set_option(2, 2 if wait_mode == "poll" else 0)
        latency = {"what": "configs[3]: one depth-20 Semaphore signal + the recursive proof verifying it, one prover context, one unit, "
                           "host-visible wall time (witness generation, transcript, downloads included)",
                   "median_ms": round(1e3 * lat[len(lat) // 2], 2), "min_ms": round(1e3 * lat[0], 2), "max_ms": round(1e3 * lat[-1], 2),
                   "runs": len(lat), "tape_replay_threads": lat_rt, "stream_wait": "poll, no sleep", "through_batch_runtime_ms": round(1e3 * t_b1, 2),
                   "lockstep_8_units_ms_per_unit": round(1e3 * t_iso / 8, 2)}
        pr.sets = all_sets
        # host/device split of one context: wall time in Ctx::wait() ("host:stream_wait" pseudo-scope) against the wall time per unit
        iso_wait = iso.pop("host:stream_wait", (0, 0.0, 0))
        tr_wait = prof.pop("host:stream_wait", (0, 0.0, 0))
        iso.pop("host:cpu_in_wait", None); iso.pop("host:cpu_in_prove", None)
        tr_cpu_wait = prof.pop("host:cpu_in_wait", (0, 0.0, 0))
        tr_cpu_prove = prof.pop("host:cpu_in_prove", (0, 0.0, 0))
        host_split = {"what": "wall ms per unit of one prover context's host thread: waiting for its stream (Ctx::wait) vs everything else "
                              "(witness generation, tape replay, transcript, copies, launches)",
                      "isolated_one_context": {"wall": round(1e3 * t_iso / 8, 2), "waiting": round(iso_wait[1] / 8, 2)},
                      "timed_region_%d_contexts" % n_threads: {"wall": round(1e3 * elapsed * 1 / max(1, local_units), 2) if local_units else None,
                                                                "waiting": round(tr_wait[1] / max(1, local_units), 2),
                                                                "cpu_ms_inside_prove_calls": round(tr_cpu_prove[1] / max(1, local_units), 2),
                                                                "cpu_ms_of_that_inside_waits": round(tr_cpu_wait[1] / max(1, local_units), 2)}}
        dname, (dcnt, dms, dbytes) = max(iso.items(), key=lambda kv: kv[1][1]) if iso else ("none", (1, 0.0, 0))
        ach = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0

        def groups(p, n_units, top=None):
            items = sorted(p.items(), key=lambda kv: -kv[1][1])[:top]
            return {k: {"launches_per_unit": round(v[0] / max(1, n_units), 1), "ms_per_unit": round(v[1] / max(1, n_units), 4),
                        "alg_GBps": round(v[2] / (v[1] * 1e-3) / 1e9, 1) if v[1] > 0 else None} for k, v in items}
        traffic, traffic_src = pmc_traffic(dname)
        # ---- the roofline block.  Bound: the integer VALU issue rate (Poseidon / Merkle / the constraint kernel are not HBM- or
        # MFMA-bound).  achieved = wave-level VALU instructions of ALL kernels per unit (SQ_INSTS_VALU of the committed --pmc pass, a property of the
        # shipped kernels) x the units/s of THIS timed region; peak = 1 / sum_c f_c / rate_c: class rates measured by gl355_valu_probe in this run,
        # moved from the clock each probe ran at to the clock sampled during the timed region; f_c = the job's dynamic instruction mix
        # (SQ_INSTS_VALU_INT64 share + the shipped ISA's split of the rest).  The dominant kernel's own launch figures and the HBM-side
        # figure SURVEY 8(d) also asks for follow as sub-blocks.
        # (the one-device rehearsal time-slices ONE GPU between the ranks: the whole job's rate is that device's rate)
        units_per_s_gpu = units / elapsed / (1 if rehearsal else max(1, world))
        job_mix, insts_per_unit, mix_src = valu_mix(None)
        k_mix, k_insts, _ = valu_mix(dname)
        clock_mhz = job_clock["mean_mhz"] if job_clock else None
        roofline = {"bound": "valu", "unit": "G wave-instructions/s", "kernel": "all kernels of a unit (dominant: %s)" % dname}
        if job_mix and insts_per_unit and clock_mhz:
            peak = valu_peak(job_mix, clock_mhz)
            ach_v = insts_per_unit * units_per_s_gpu / 1e9
            roofline.update({"achieved": round(ach_v, 1), "peak": round(peak, 1), "frac": round(ach_v / peak, 4),
                             "valu_insts_per_unit": insts_per_unit, "units_per_s_per_gpu": round(units_per_s_gpu, 2), "mix": job_mix, "mix_source": mix_src,
                             "nominal_clk_per_wave_inst_per_simd": NOMINAL_CLK, "clock_during_timed_region": job_clock,
                             "clk_per_valu_inst_per_simd_achieved": round(N_SIMD * clock_mhz * 1e6 / (ach_v * 1e9), 3),
                             "formula": "achieved = valu_insts_per_unit x units_per_s_per_gpu; peak = 1024 SIMDs x clock_during_timed_region.mean_mhz / "
                                        "sum_c mix[c] x nominal_clk[c] (full-rate classes 2 clk, multiply / carry / 64-bit classes 4 clk per wave64 "
                                        "instruction); valu_insts_per_unit and mix.mad64 from the committed rocprofv3 --pmc pass (SQ_INSTS_VALU, "
                                        "SQ_INSTS_VALU_INT64), the full32 / half32 split of the rest from the shipped ISA (tools/isa_mix.py)",
                             "probe": {"classes": classes, "peak_at_probe_rates": round(valu_peak_probe(job_mix, classes, clock_mhz), 1) if classes else None,
                                       "note": "gl355_valu_probe in this run: what kernels that ONLY issue one class reach (G wave-instructions/s, and the clock "
                                               "read inside each).  The job issues faster than these single-class loops, so they are not the ceiling"}})
        else:
            roofline.update({"achieved": None, "peak": None, "frac": None, "note": "no --pmc pass / ISA histogram under profiles/ or no clock samples"})
        dom = {"kernel": dname, "launches_per_unit": round(dcnt / max(1, iso_units), 1), "avg_launch_ms": round(dms / max(1, dcnt), 4),
               "how": "HIP events on the launching stream, ONE prover context (a lock-step batch of 8 units), %d units, straight after the timed region; "
                      "kernel = the scope group with the largest summed duration.  One context's launch is ~1 800 waves -- under two per SIMD -- so it "
                      "cannot fill the chip by itself (`fill`); the other contexts' kernels run in those slots, which the job-level figure above measures" % iso_units}
        if k_mix and k_insts and dms > 0 and clock_mhz:
            k_peak = valu_peak(k_mix, clock_mhz)
            k_ach = k_insts / (dms / max(1, dcnt) * 1e-3) / 1e9
            dom.update({"valu_insts_per_launch": k_insts, "mix": k_mix, "achieved": round(k_ach, 1), "peak": round(k_peak, 1), "fill": round(k_ach / k_peak, 4)})
        roofline["dominant_kernel"] = dom
        roofline["hbm"] = {"bound": "hbm", "kernel": dname, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                           "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(dbytes / max(1, dcnt)),
                           "note": "secondary figure (SURVEY 8(d) asks for it): the dominant kernel is not HBM-bound"}
        roofline["gpu_ms_per_unit_all_kernels"] = round(sum(v[1] for v in iso.values()) / max(1, iso_units), 3)
        roofline["kernel_groups"] = groups(iso, iso_units, 10)
        roofline["timed_region_events"] = {"what": "the same scopes on 1 of the %d concurrent streams during the timed region "
                                                   "(includes queueing behind the other streams)" % n_threads,
                                           "units": local_units, "kernel_groups": groups(prof, local_units, 6)}
        line = {
            "metric": "recursive plonky2 proofs/sec (Semaphore d=%d)" % args.log_members,
            "value": round(units / elapsed, 2), "unit": "recursive proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 (Goldilocks field, integer)", "data": "synthetic",
            "config": {"workload": "recursive: per unit one Semaphore signal (group 2^%d, n=2^13, blowup 8, 28 FRI queries, 16 PoW bits, "
                                   "zk) + the recursive proof verifying it (n=2^%d, same FRI parameters); %d units per GPU per step, %d "
                                   "prover contexts per GPU; all_gather of (nullifier|topic) + Poseidon aggregation root per step"
                                   % (args.log_members, pr.rc.data.degree_bits, per, n_threads),
                       "parallelism": "independent proofs sharded over ranks, no data-path collective",
                       "exchange": getattr(comm, "backend_name", "none (one rank): gl355_aggregation_root over the local leaves"),
                       "host": "%d usable host cores per rank, %s device waits, %d tape-replay thread(s) per context" % (
                           cores_per_rank, {"sleep": "sleeping (hipDeviceScheduleBlockingSync)", "poll": "polling (hipStreamQuery + 30-us sleeps)", "spin": "spinning"}[wait_mode], replay_threads) +
                               ("; witness tape replayed on the device" if os.environ.get("GL355_BENCH_DEVICE_REPLAY") == "1" else "; witness tape replayed on host threads") +
                               ("; rank pinned to CPUs %s (ranks take contiguous slices)" % pinned if pinned else ("; no CPU pinning (one rank)" if world == 1 else "; CPU pinning off / unavailable")),
                       "host_cpu_ms_per_unit": round(1e3 * host_cpu_s / max(1, (hi - lo) * args.steps), 2),
                       "units_proven_in_process": int(sum(pr.units_done)),
                       "host_split": host_split},
            "roofline": roofline,
            "aggregation_root": ["%016x" % int(x) for x in root[0]],
            "latency_single_unit_ms": latency,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline_recursive(pr)
            except Exception as exc:
                line["cpu_baseline"] = {"error": repr(exc)}
            try:    # the final wrap (wrapper.rs:35-56): the recursive circuit proven under the BN254-Poseidon hasher, one context
                rec = importlib.import_module("stark-verifier_amd.recursion")
                inner = pr.last[0]
                c0 = pr.sets[0]
                wc = rec.WrapperCircuit(c0, pr.inner_data.common()).build([inner], np.random.default_rng(3))
                rows_w, pis_w = wc.witness([inner])
                c0.set_option(2, 3)                                  # a lone proof: poll the stream without sleeping
                pr.plonk.prove_sparse(c0, wc.data, wc.row_idx, rows_w, pis_w, 1, flat_only=True)
                t_w = time.perf_counter()
                for k in range(3):
                    pr.plonk.prove_sparse(c0, wc.data, wc.row_idx, rows_w, pis_w, 2 + k, flat_only=True)
                t_w = time.perf_counter() - t_w
                c0.set_option(2, 2 if wait_mode == "poll" else 0)
                line["wrap_proof_bn254"] = {"ms_per_proof": round(t_w / 3 * 1e3, 2), "degree_bits": wc.data.degree_bits,
                                            "what": "WrapperCircuit: in-circuit verification of one Semaphore proof, outer proof with "
                                                    "Bn254PoseidonHash Merkle trees / transcript / PoW, cap_height 0, no blinding; "
                                                    "single context, latency"}
                del wc
            except Exception as exc:
                line["wrap_proof_bn254"] = {"error": repr(exc)}
            try:
                del pr
                line["ntt_lde"] = lde_figure(gl, local_rank)
            except Exception as exc:
                line["ntt_lde"] = {"error": repr(exc)}
            try:
                line["merkle_2p22"] = merkle_figures(gl, local_rank)
            except Exception as exc:
                line["merkle_2p22"] = {"error": repr(exc)}
            try:
                line["bn254_finalisation_kernels"] = bn254_figures(gl, local_rank)
            except Exception as exc:
                line["bn254_finalisation_kernels"] = {"error": repr(exc)}
            try:
                if os.environ.get("GL355_BENCH_NO_AGGREGATE") != "1":
                    line["aggregate"] = aggregate_figure(gl, local_rank)
            except Exception as exc:
                line["aggregate"] = {"error": repr(exc)}
            try:
                line["halo2_create_proof_k23"] = halo2_figure(gl, local_rank)
            except Exception as exc:
                line["halo2_create_proof_k23"] = {"error": repr(exc)}
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here (one process per GPU, the same command line, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment -- exactly what torch.distributed.run would set), stream their output through,
    and exit with the first non-zero code after ending the others.  Rank 0 alone prints the JSON line.  The ranks rendezvous through
    open_comm's TCPStore on 127.0.0.1 (rank 0 hosts it) and exchange through gl355_comm_* (RCCL), as under an external launcher."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GL355_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.2)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("[bench] rank %d exited with code %d: ending the other ranks\n" % (procs.index(p), code))
                    for q in live:          # the process groups started above, nothing else
                        try:
                            os.killpg(q.pid, signal.SIGTERM)
                        except OSError:
                            pass
    except KeyboardInterrupt:
        rc = 130
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, signal.SIGTERM)
                except OSError:
                    pass
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["recursive", "lde", "semaphore", "exchange"], default="recursive",
                    help="recursive = Semaphore d=20 signal + recursive proof per unit (default; BASELINE configs[3]/[4]); "
                         "lde = configs[1]; semaphore = the signals alone")
    ap.add_argument("--proofs-per-step", type=int, default=128,
                    help="units (recursive) / proofs (semaphore) per GPU per step; 128 = BASELINE configs[4] (1024 proofs over 8 GPUs)")
    ap.add_argument("--threads", type=int, default=0,
                    help="concurrent prover contexts per GPU (one HIP stream + one host thread each); every context proves "
                         "GL355_OPT_BATCH_UNITS = 8 units in lock-step.  Default: 10 with >= 12 usable host cores per rank, else 8 "
                         "(round 3, final kernels: 8 -> 289.6, 9 -> 291.7, 10 -> 291.4, 11 -> 290.5, 16 -> 249 units/s; profiles/r03b_contexts_sweep.txt)")
    ap.add_argument("--log-members", type=int, default=20, help="log2 of the access-set size (tree depth)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)        # no external launcher: this process becomes the launcher of its N ranks
    if args.threads <= 0:
        args.threads = 10 if host_cores() // max(1, int(os.environ.get("WORLD_SIZE", "1"))) >= 12 else 8
    # two hardware queues per prover context (proving stream + side stream): the HIP runtime's default is 4, streams then share queues and a latency-bound
    # Merkle-top kernel on one stream holds up the streams behind it (measured 164 -> 172 proofs/s at 16 contexts).  Read when
    # the HIP runtime initialises, i.e. before torch / libgl355 touch the device (they are imported by the main_* functions).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, 2 * int(os.environ.get("GL355_BENCH_CONTEXTS", args.threads)))))
    if args.workload == "exchange":
        return main_exchange(args)
    if args.workload == "semaphore":
        return main_semaphore(args)
    if args.workload == "recursive":
        return main_recursive(args)
    return main_lde(args)


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


if __name__ == "__main__":
    main()
