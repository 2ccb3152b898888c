"""CPU, world_size 2 over gloo: the sharding + gather logic of the multi-GPU path (no GPU compute).
The aggregation root itself is a GPU Merkle build (covered in test_gpu_parity); here the gathered
leaves are folded with the oracle only to check that the gather order reproduces the single-process
result."""
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


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    par = importlib.import_module("stark-verifier_amd.parallel")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = par.shard_range(total, rank, world)
    # unit i's leaf = 8 deterministic words (stand-in for nullifier || topic)
    local = torch.tensor([[i * 8 + j for j in range(8)] for i in range(lo, hi)], dtype=torch.int64)
    per = (total + world - 1) // world
    if local.shape[0] < per:                     # equal shapes for all_gather: pad the last shard
        local = torch.cat([local, torch.full((per - local.shape[0], 8), -1, dtype=torch.int64)])
    allv = par.gather_leaves(local, dist)
    dist.barrier()
    if rank == 0:
        q.put(allv.numpy())
    dist.destroy_process_group()


def test_shard_range():
    par = importlib.import_module("stark-verifier_amd.parallel")
    for total in (0, 1, 7, 8, 1024, 1000):
        for world in (1, 2, 4, 8):
            spans = [par.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert par.shard_range(1024, 3, 8) == (384, 512)     # 128 proofs per GPU (BASELINE cfg-5)


@pytest.mark.timeout(120)
def test_gather_two_ranks_gloo(orc):
    import torch.multiprocessing as mp
    par = importlib.import_module("stark-verifier_amd.parallel")
    total, world = 13, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=90)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    valid = got[got[:, 0] >= 0]
    want = np.array([[i * 8 + j for j in range(8)] for i in range(total)], dtype=np.int64)
    assert np.array_equal(valid, want)           # rank order == unit order
    # the root over the gathered leaves equals the root a single process would build
    a = par.pad_pow2(valid.astype(np.uint64))
    b = par.pad_pow2(want.astype(np.uint64))
    assert a.shape == (16, 8)
    assert np.array_equal(orc.merkle_build(a, 0)[1], orc.merkle_build(b, 0)[1])
