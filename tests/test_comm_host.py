"""CPU: the communicator of the C ABI (gl355_comm_*, gl355_gather_digests) on its TCP back-end with three host processes:
rank order of the gathered leaves (results are placed by rank, not by arrival), barrier, max, error paths.
The RCCL back-end shares every line except the transport; it is exercised on the GPU box (tests/test_gpu_comm.py)."""
import ctypes as C
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, cid, total, q):
    sys.path.insert(0, ROOT)
    import time
    par = importlib.import_module("stark-verifier_amd.parallel")
    lib = importlib.import_module("stark-verifier_amd._lib").load()
    if rank == 0:
        time.sleep(0.3)                      # the other ranks retry until rank 0 listens
    comm = par.Comm(None, par.COMM_HOST, cid, rank, world, lib=lib)
    lo, hi = par.shard_range(total, rank, world)
    per = (total + world - 1) // world
    local = np.full((per, 8), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)     # padding rows of the last shard
    for k, i in enumerate(range(lo, hi)):
        local[k] = [i * 8 + j for j in range(8)]
    if rank == 1:
        time.sleep(0.2)                      # arrival order != rank order
    allv = comm.gather(local)
    comm.barrier()
    m = comm.max(10.0 + rank)
    tiny = comm.gather(np.array([rank], dtype=np.uint64))
    comm.close()
    q.put((rank, allv, m, tiny))


@pytest.mark.timeout(120)
def test_host_comm_three_ranks():
    import multiprocessing as mp
    total, world = 13, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    par = importlib.import_module("stark-verifier_amd.parallel")
    lib = importlib.import_module("stark-verifier_amd._lib").load()
    cid = par.Comm.unique_id(lib, par.COMM_HOST, "127.0.0.1", port)      # minted once (token inside), handed to every rank
    assert cid != par.Comm.unique_id(lib, par.COMM_HOST, "127.0.0.1", port)
    procs = [ctx.Process(target=_worker, args=(r, world, cid, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    # strangers on the port while the ranks assemble: a silent connection, garbage, and a well-formed rank claim with the
    # wrong token -- none of them may claim a rank or tear the communicator down
    import struct
    import threading
    import time
    strangers = []

    def intrude():
        for payload in (struct.pack("<i", 1) + b"\0" * 16, b"\xff" * 20, b""):
            for _ in range(200):                 # until rank 0 listens
                try:
                    s = socket.create_connection(("127.0.0.1", port), timeout=2)
                except OSError:
                    time.sleep(0.05)
                    continue
                if payload:
                    s.sendall(payload)
                strangers.append(s)
                break
    th = threading.Thread(target=intrude)
    th.start()
    res = {}
    for _ in range(world):
        r, allv, m, tiny = q.get(timeout=90)
        res[r] = (allv, m, tiny)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    th.join(30)
    assert len(strangers) >= 1                 # at least the wrong-token claim of rank 1 reached the listener
    for s in strangers:
        s.close()
    want = np.array([[i * 8 + j for j in range(8)] for i in range(total)], dtype=np.uint64)
    for r in range(world):
        allv, m, tiny = res[r]
        assert allv.shape == (15, 8)
        valid = allv[allv[:, 0] != np.uint64(0xFFFFFFFFFFFFFFFF)]
        assert np.array_equal(valid, want)            # every rank sees the units in rank (= unit) order
        assert m == 12.0 and list(tiny.reshape(-1)) == [0, 1, 2]


def _stale_worker(world, cid, q):
    sys.path.insert(0, ROOT)
    import time
    par = importlib.import_module("stark-verifier_amd.parallel")
    lib = importlib.import_module("stark-verifier_amd._lib").load()
    t0 = time.time()
    try:
        par.Comm(None, par.COMM_HOST, cid, 1, world, lib=lib)
        q.put(("stale", "accepted", time.time() - t0))
    except Exception as exc:
        q.put(("stale", repr(exc), time.time() - t0))


@pytest.mark.timeout(120)
def test_host_comm_rejected_rank_fails_at_create():
    """a rank presenting a stale token is turned away by rank 0 (connection closed, no acknowledgement byte) and its
    gl355_comm_create fails there and then -- not at its first gather -- while the real rank still gets in."""
    import multiprocessing as mp
    total, world = 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    par = importlib.import_module("stark-verifier_amd.parallel")
    lib = importlib.import_module("stark-verifier_amd._lib").load()
    cid = par.Comm.unique_id(lib, par.COMM_HOST, "127.0.0.1", port)
    stale = par.Comm.unique_id(lib, par.COMM_HOST, "127.0.0.1", port)         # same address, another token
    p0 = ctx.Process(target=_worker, args=(0, world, cid, total, q))
    pbad = ctx.Process(target=_stale_worker, args=(world, stale, q))
    p0.start()
    pbad.start()
    tag, what, dt = q.get(timeout=60)
    assert tag == "stale" and what != "accepted" and "rejected" in what and dt < 30, (what, dt)
    p1 = ctx.Process(target=_worker, args=(1, world, cid, total, q))
    p1.start()
    got = sorted(q.get(timeout=60)[0] for _ in range(2))
    assert got == [0, 1]
    for p in (p0, p1, pbad):
        p.join(30)
        assert p.exitcode == 0


@pytest.mark.timeout(180)
def test_host_comm_eight_ranks_cfg5_shape(orc):
    """BASELINE configs[4] without GPUs: 8 ranks (one process each, TCP back-end of the communicator) x 128 units -- the flow of
    examples/native_units --ranks 8 / bench.py --gpus 8 with stub leaves: block partition, one gl355_gather_digests of 64 B per unit,
    every rank sees the 1024 leaves in unit order, and the aggregation root over the gathered leaves == the single-process root"""
    import multiprocessing as mp
    total, world = 1024, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    par = importlib.import_module("stark-verifier_amd.parallel")
    lib = importlib.import_module("stark-verifier_amd._lib").load()
    cid = par.Comm.unique_id(lib, par.COMM_HOST, "127.0.0.1", port)
    procs = [ctx.Process(target=_worker, args=(r, world, cid, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, allv, m, tiny = q.get(timeout=150)
        res[r] = (allv, m, tiny)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want = np.array([[i * 8 + j for j in range(8)] for i in range(total)], dtype=np.uint64)
    single_root = orc.merkle_build(par.pad_pow2(want), 0)[1]
    for r in range(world):
        allv, m, tiny = res[r]
        assert np.array_equal(allv, want), r                    # 128 units per rank, rank (= unit) order, no padding rows
        assert m == 10.0 + world - 1 and list(tiny.reshape(-1)) == list(range(world))
    assert np.array_equal(orc.merkle_build(par.pad_pow2(res[0][0]), 0)[1], single_root)
    assert [par.shard_range(total, r, world) for r in (0, 7)] == [(0, 128), (896, 1024)]


def test_comm_argument_errors(gl):
    lib = gl._lib.load()
    par = importlib.import_module("stark-verifier_amd.parallel")
    buf = C.create_string_buffer(128)
    assert lib.gl355_comm_host_id(b"not-an-address", 1234, buf) == -1
    assert lib.gl355_comm_host_id(b"127.0.0.1", 0, buf) == -1
    assert lib.gl355_comm_host_id(b"127.0.0.1", 4242, buf) == 0
    h = C.c_void_p()
    assert lib.gl355_comm_create(None, par.COMM_HOST, buf, 3, 2, C.byref(h)) == -1          # rank >= world
    assert lib.gl355_comm_create(None, par.COMM_RCCL, buf, 0, 1, C.byref(h)) == -1          # RCCL needs a context
    assert lib.gl355_comm_create(None, par.COMM_HOST, b"\0" * 128, 0, 2, C.byref(h)) == -1  # not a host id
    assert b"host" in lib.gl355_comm_last_error(None)
    # a world of one needs no peer
    c = par.Comm(None, par.COMM_HOST, buf.raw, 0, 1, lib=lib)
    x = np.arange(8, dtype=np.uint64).reshape(1, 8)
    assert np.array_equal(c.gather(x), x) and c.max(3.5) == 3.5
    c.barrier()
    assert lib.gl355_gather_digests(c.h, None, 4, x.ctypes.data) == -1
    c.close()
