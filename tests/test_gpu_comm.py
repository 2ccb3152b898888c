"""GPU: the exchange behind the C ABI.  gl355_aggregation_root against the oracle; the RCCL back-end of gl355_comm_* as far as a
one-GPU box allows (a communicator of one rank: id, ncclCommInitRank, ncclAllGather / ncclAllReduce on the context's stream with
host and device operands); and the whole N = 2 flow of the plain C++ host program -- two processes, block-partitioned units,
gl355_gather_digests, aggregation root on rank 0 -- with the TCP back-end, both ranks on the one device (RCCL refuses two ranks
on one GPU), which must reproduce the root a single process computes for the same units."""
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle_lib import rand_field

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_aggregation_root_entry(gl, ctx, orc):
    par = importlib.import_module("stark-verifier_amd.parallel")
    rng = np.random.default_rng(0xA66)
    for n, w in ((1, 8), (2, 8), (5, 8), (128, 8), (1000, 8), (1024, 8), (37, 4), (9, 12)):
        leaves = rand_field(rng, (n, w))
        want = orc.merkle_build(par.pad_pow2(leaves), 0)[1]
        assert np.array_equal(par.aggregation_root(ctx, leaves), want), (n, w)
    import torch
    lv = rand_field(rng, (100, 8))
    t = torch.from_numpy(lv.view(np.int64)).cuda()
    root = np.empty(4, dtype=np.uint64)
    ctx.check(ctx.lib.gl355_aggregation_root(ctx.h, t.data_ptr(), 100, 8, root.ctypes.data))      # device operand
    assert np.array_equal(root, orc.merkle_build(par.pad_pow2(lv), 0)[1][0])
    assert ctx.lib.gl355_aggregation_root(ctx.h, None, 4, 8, root.ctypes.data) == -1


def test_rccl_communicator_of_one_rank(gl, ctx):
    import torch
    par = importlib.import_module("stark-verifier_amd.parallel")
    cid = par.Comm.unique_id(ctx.lib, par.COMM_RCCL)
    assert len(cid) == 128 and any(cid)
    comm = par.Comm(ctx, par.COMM_RCCL, cid, 0, 1)
    x = np.arange(64 * 8, dtype=np.uint64).reshape(64, 8)
    assert np.array_equal(comm.gather(x), x)                                   # ncclAllGather, host operands staged through HBM
    t = torch.from_numpy(x.view(np.int64)).cuda()
    out = torch.zeros_like(t)
    torch.cuda.synchronize()
    assert ctx.lib.gl355_gather_digests(comm.h, t.data_ptr(), x.size, out.data_ptr()) == 0     # device operands in place
    ctx.sync()
    assert torch.equal(out, t)
    comm.barrier()
    assert comm.max(2.25) == 2.25
    comm.close()


def test_rccl_two_devices(gl, ctx, orc, tmp_path):
    """world > 1 over RCCL / xGMI: one process per device, gl355_comm_create(RCCL) + gl355_gather_digests (host and device operands) +
    barrier + max, and the aggregation root rank 0 computes over the gathered leaves == the root of the same leaves in one process
    (the oracle's).  Needs two devices: SKIPPED (never passed) on a one-GPU box -- RCCL refuses two ranks on one device."""
    import ctypes as C
    n = C.c_int32(0)
    assert ctx.lib.gl355_device_count(C.byref(n)) == 0
    if n.value < 2:
        pytest.skip("needs >= 2 devices (gl355_device_count = %d)" % n.value)
    par = importlib.import_module("stark-verifier_amd.parallel")
    world = 8 if n.value >= 8 else (4 if n.value >= 4 else 2)
    per = 16
    idf = str(tmp_path / "rccl.id")
    worker = os.path.join(ROOT, "tests", "rccl_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), idf, str(per)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rccl_worker
    leaves = np.concatenate([rccl_worker.block(r, per) for r in range(world)])
    want = orc.merkle_build(par.pad_pow2(leaves), 0)[1][0]
    got = re.search(r"ROOT ([0-9a-f ]+)", outs[0][0]).group(1).split()
    assert [int(v, 16) for v in got] == [int(v) for v in want]


def test_two_rank_cpp_host_flow(tmp_path):
    art = str(tmp_path / "art")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "export_artifacts.py"), art, "3"], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "examples", "native_units")
    idf = str(tmp_path / "comm.id")
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    procs = [subprocess.Popen([exe, art, "3", "2", "4", "--ranks", "2", str(r), idf, "--host-comm", str(port)], stdout=subprocess.PIPE, text=True, env=env)
             for r in (1, 0)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    two = outs[1]
    assert "8 units" in two
    one = subprocess.check_output([exe, art, "3", "2", "8"], text=True, env=env)       # the same 8 units (members 12..19 mod 8) in one process
    root = lambda o: re.search(r"aggregation root ([0-9a-f ]+)", o).group(1).strip()
    assert root(two) == root(one)


@pytest.mark.timeout(900)
def test_bench_self_launched_two_ranks_one_device():
    """`python bench.py --gpus 2` with no launcher around it -- the default (recursive) workload, both ranks on cuda:0
    (GL355_BENCH_ONE_DEVICE=1: the exchange then runs over the C ABI's TCP communicator, RCCL refuses two ranks on one device).  The
    launcher starts the ranks, they rendezvous, shard the units, gather the leaves, and rank 0 alone prints the line."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                "GPU_MAX_HW_QUEUES")}
    env.update(GL355_BENCH_ONE_DEVICE="1", GL355_BENCH_CONTEXTS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--proofs-per-step", "16",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["unit"] == "recursive proofs/s"
