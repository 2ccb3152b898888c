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
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_common import *      # noqa: E402,F401,F403  (BASELINE shape constants, profile look-ups, VALU roofline helpers, open_comm)
from bench_blocks import (aggregate_figure, bn254_figures, halo2_figure, lde_figure, main_exchange, main_lde, main_semaphore,  # noqa: E402
                          merkle_figures)


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


def expected_rate(cores_per_rank):
    """units/s ONE MI355X reaches when its rank has this many usable host cores (bench.py picks the wait / replay mode from that number; measured with
    one rank confined by taskset on a 1-GPU box, tools/hostside_one_rank.sh -> profiles/r06_hostside_8rank.txt).  An 8-GPU node with 8 x k cores should
    see 8 x this figure: the ranks share nothing on the host side."""
    for k, r in ((16, 317.0), (12, 314.0), (8, 307.0), (3, 306.0), (2, 303.0), (1, 285.0)):
        if cores_per_rank >= k:
            return r
    return None


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
    replay_threads = int(os.environ.get("GL355_BENCH_REPLAY_THREADS", 2 if cores_per_rank >= 12 else 1))
    # witness generation of the recursive circuit: on the device (tape interpreter on the context's side stream: 2 host cores per
    # rank are enough, 255 units/s) or on host threads (4 % more throughput when the rank has cores to spare: 270 vs 259 units/s)
    # (round 6, profiles/r06_hostside_8rank.txt: one rank at 8 cores reaches 307 units/s with the tape on host threads at 12.4 ms of host CPU per unit and
    # 306 with the tape on the device at 4.8 ms -- the host tape only pays from 12 cores on, where ten contexts fit as well: 314-317)
    os.environ.setdefault("GL355_BENCH_DEVICE_REPLAY", "0" if cores_per_rank >= 12 else "1")
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
    model = None
    if rank == 0:
        try:
            import bench_common
            model = bench_common.VALU_MODEL = ValuModel(pr.sets[0])
        except Exception as exc:
            sys.stderr.write("[bench] valu probes failed: %r\n" % (exc,))
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
        lat_rt = int(os.environ.get("GL355_BENCH_LAT_REPLAY_THREADS", 28 if cores_per_rank >= 16 else max(1, min(14, cores_per_rank - 2))))      # 28 FRI-query segments: one thread each from 16 cores on (0.3 ms bursts; 11.06 vs 11.18 ms with 14 = two rounds, profiles/r06_latency_replay_threads.txt), 14 below
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
        # MFMA-bound).  achieved = wave-level VALU instructions of ALL kernels per unit in steady state (SQ_INSTS_VALU of the committed --pmc passes:
        # 3 steps minus 1 step, so the set-up kernels cancel) x the units/s of THIS timed region.  peak = 1024 SIMDs x the clock sampled during the
        # timed region / the pair-aware floor of the job's instruction multiset: every opcode form priced at the issue cost gl355_valu_probe_ops
        # measured in THIS run, every pair of forms that overlaps (gl355_valu_probe_pairs) allowed to -- the cheapest pairing, a linear program
        # (tools/bench_common.py valu_costs).  Next to it: the additive model (no overlaps: what round 5 priced, which real code beats by up to 10 %),
        # the hardware's absolute issue limit (2 clk per wave64 instruction whatever it is), and SURVEY 8(d)'s algorithmic line (multiply-adds only).
        # (the one-device rehearsal time-slices ONE GPU between the ranks: the whole job's rate is that device's rate)
        units_per_s_gpu = units / elapsed / (1 if rehearsal else max(1, world))
        clock_mhz = job_clock["mean_mhz"] if job_clock else None
        roofline = {"bound": "valu", "unit": "G wave-instructions/s", "kernel": "all kernels of a unit (dominant: %s)" % dname}
        job_forms, insts_per_unit = model.job_forms_per_unit() if model else (None, None)
        if job_forms and insts_per_unit and clock_mhz:
            pk = model.peak(job_forms, clock_mhz)
            ach_v = insts_per_unit * units_per_s_gpu / 1e9
            job = model.pmc.get("job", {})
            # SURVEY 8(d) cfg-3's algorithmic line: permutations x 1 077 modular multiplications x 4 v_mad_u64_u32 per 64 lanes against the measured multiply-add rate
            mad_clk = min(model.ops[f]["clk"] for f in model.ops if f.startswith("v_mad_u64_u32"))
            hash_share = sum(e.get("steady_valu_insts", 0.0) for k, e in model.pmc["kernels"].items()
                             if k.split("<")[0] in ("hash_leaves_kernel", "merkle_level_kernel", "pow_grind_units_kernel", "two_to_one_kernel")) / max(1.0, job.get("valu_insts_steady_total", 1.0))
            perm_insts = model.pmc.get("probes", {}).get("vpc_permute_kernel", {}).get("valu_insts_per_item")
            roofline.update({
                "achieved": round(ach_v, 1), "peak": pk["peak"], "frac": round(ach_v / pk["peak"], 4),
                "valu_insts_per_unit": insts_per_unit, "valu_insts_per_unit_including_setup": job.get("valu_insts_per_unit_including_setup"),
                "units_per_s_per_gpu": round(units_per_s_gpu, 2), "clock_during_timed_region": job_clock,
                "clk_per_valu_inst_per_simd": {"achieved": round(N_SIMD * clock_mhz * 1e6 / (ach_v * 1e9), 3), "pair_aware_floor": pk.get("clk_per_inst_floor"),
                                               "additive_model": pk["clk_per_inst_additive"], "hardware_issue_limit": 2.0},
                "other_lines": {
                    "additive_model": {"peak": pk["peak_additive"], "frac": round(ach_v / pk["peak_additive"], 4),
                                       "note": "sum of stand-alone opcode costs, no overlaps: NOT a ceiling -- the lock-step product issues up to 10 % faster (composite_checks)"},
                    "hardware_issue_limit_2clk": {"peak": round(N_SIMD * clock_mhz * 1e6 / 2.0 / 1e9, 1), "frac": round(ach_v * 1e9 * 2.0 / (N_SIMD * clock_mhz * 1e6), 4),
                                                  "note": "MI355X_MICROARCH.md, Wave scheduling: a wave64 VALU instruction occupies its SIMD-32 for at least 2 clk"},
                    "algorithmic_multiply_adds": None if not perm_insts else {
                        "what": "SURVEY 8(d) cfg-3: the job's permutations x 1 077 modular multiplications x 4 v_mad_u64_u32, per 64 lanes, against 1024 SIMDs x clock / the measured multiply-add cost",
                        "permutations_per_unit": round(hash_share * insts_per_unit * 64 / perm_insts),
                        "frac": round(hash_share * insts_per_unit / perm_insts * 1077 * 4 * units_per_s_gpu * mad_clk / (N_SIMD * clock_mhz * 1e6), 4),
                        "mad_clk": mad_clk}},
                "unprobed_share_of_instructions": pk["unprobed_share"],
                "formula": "achieved = valu_insts_per_unit x units_per_s_per_gpu; peak = achieved-independent: 1024 SIMDs x clock_during_timed_region.mean_mhz / "
                           "clk_per_valu_inst_per_simd.pair_aware_floor, the floor = min over pairings of sum(pair costs + stand-alone costs) / instructions, costs "
                           "measured in this run (probe), instruction multiset = sum over kernels of SQ_INSTS_VALU (steady state) split into opcode forms by the "
                           "shipped ISA inside each of the counters' classes (INT64 / INT32 / rest)",
                "probe": model.report()})
        else:
            roofline.update({"achieved": None, "peak": None, "frac": None, "note": "no --pmc pass / ISA histogram under profiles/ or no clock samples"})
        k_e = model.pmc.get("kernels", {}).get(dname) if model else None
        k_insts = k_e.get("valu_insts_per_launch") if k_e else None
        k_forms = model.dynamic_forms(dname, k_insts, k_e.get("valu_int64_per_launch"), k_e.get("valu_int32_per_launch")) if k_insts else None
        dom = {"kernel": dname, "launches_per_unit": round(dcnt / max(1, iso_units), 1), "avg_launch_ms": round(dms / max(1, dcnt), 4),
               "how": "HIP events on the launching stream, ONE prover context (a lock-step batch of 8 units), %d units, straight after the timed region; "
                      "kernel = the scope group with the largest summed duration.  One context's launch is ~1 800 waves -- under two per SIMD -- so it "
                      "cannot fill the chip by itself (`fill`); the other contexts' kernels run in those slots, which the job-level figure above measures" % iso_units}
        if k_forms and dms > 0 and clock_mhz:
            k_pk = model.peak(k_forms, clock_mhz)
            k_ach = k_insts / (dms / max(1, dcnt) * 1e-3) / 1e9
            dom.update({"valu_insts_per_launch": k_insts, "clk_per_inst_floor": k_pk.get("clk_per_inst_floor"), "clk_per_inst_additive": k_pk["clk_per_inst_additive"],
                        "achieved": round(k_ach, 1), "peak": k_pk["peak"], "fill": round(k_ach / k_pk["peak"], 4)})
        roofline["dominant_kernel"] = dom
        roofline["hbm"] = {"bound": "hbm", "kernel": dname, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                           "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(dbytes / max(1, dcnt)),
                           "note": "secondary figure (SURVEY 8(d) asks for it): the dominant kernel is not HBM-bound"}
        roofline["traffic"] = traffic          # HBM-side bytes per launch of the dominant kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected per the guide); details in `hbm`
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
                       # what this rank's host side was set to, and the rate measured for that setting on one MI355X (the 8-GPU job's rate
                       # depends on the node's cores per rank through this choice, so the line says which one it made)
                       "host_mode": {"cores_per_rank": cores_per_rank, "contexts": n_threads, "device_waits": wait_mode,
                                     "witness_tape": "device" if os.environ.get("GL355_BENCH_DEVICE_REPLAY") == "1" else "host threads",
                                     "tape_replay_threads": replay_threads, "pinned_cpus": pinned},
                       "expected_units_per_s_per_gpu": expected_rate(cores_per_rank),
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
    grace = float(os.environ.get("GL355_BENCH_KILL_GRACE_S", "10"))
    term_at = None                          # when the surviving ranks were asked to end (SIGTERM)

    def signal_all(which, sig):             # the process groups started above, nothing else
        for q in which:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, sig)
                except OSError:
                    pass
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
                    signal_all(live, signal.SIGTERM)
                    term_at = time.monotonic()
            # a rank stuck inside a native RCCL / HIP call (or one with a SIGTERM handler installed) never honours SIGTERM: after the grace
            # period the groups are killed outright, so the launcher always returns -- non-zero -- instead of hanging
            if term_at is not None and live and time.monotonic() - term_at > grace:
                sys.stderr.write("[bench] %d rank(s) still alive %.0f s after SIGTERM: SIGKILL\n" % (len(live), grace))
                signal_all(live, signal.SIGKILL)
                term_at = float("inf")
                deadline = time.monotonic() + 10
                while live and time.monotonic() < deadline:
                    live = [q for q in live if q.poll() is None]
                    time.sleep(0.1)
                break
    except KeyboardInterrupt:
        rc = 130
        signal_all(procs, signal.SIGTERM)
        time.sleep(min(grace, 2.0))
        signal_all(procs, signal.SIGKILL)
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



if __name__ == "__main__":
    main()
