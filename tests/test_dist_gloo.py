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


class _FakeAggregator:
    """stand-in with the Aggregator interface: 'aggregating' concatenates the inputs' tags, so the test sees which proofs reached
    which call in which order (the real tree needs a GPU: tools/aggregate_distributed.py, tests/test_gpu_recursion.py)"""

    def __init__(self):
        self.calls = []

    def aggregate(self, signals, seed=None, rng=None, ctxs=None, start_level=0, key_domain=0):
        self.calls.append((len(signals), start_level, key_domain))
        proof = np.concatenate([np.asarray(s[0], dtype=np.uint64) for s in signals])[:6]
        proof = np.concatenate([proof, np.zeros(6 - proof.size, dtype=np.uint64)])          # fixed "proof" size
        pis = np.concatenate([np.asarray(s[1], dtype=np.uint64) for s in signals])
        return proof, pis, {"level": start_level + len(signals).bit_length() - 1}


def _agg_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    par = importlib.import_module("stark-verifier_amd.parallel")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    agg = _FakeAggregator()
    local = [(np.array([100 * rank + j], dtype=np.uint64), np.array([1000 * rank + j], dtype=np.uint64)) for j in range(4)]
    out = par.aggregate_distributed(agg, local, dist)
    dist.barrier()
    q.put((rank, agg.calls, None if out is None else (out[0].tolist(), out[1].tolist(), out[2])))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_aggregate_distributed_exchange_gloo():
    """the lower levels stay on the ranks, one all_gather carries the per-rank proofs (words | public inputs) in rank order, and
    only rank 0 runs the upper levels, starting at the right tree level"""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(world):
        rank, calls, out = q.get(timeout=100)
        res[rank] = (calls, out)
    for p in procs:
        p.join(timeout=30)
    assert res[1] == ([(4, 0, 2)], None)                   # blinding-key domain 1 + rank: ranks never share a key stream
    calls0, out0 = res[0]
    assert calls0 == [(4, 0, 1), (2, 2, 0)]                # 4 local signals = 2 levels, then the 2 rank proofs from level 2 (domain 0)
    proof, pis, cd = out0
    assert pis == [0, 1, 2, 3, 1000, 1001, 1002, 1003] and cd == {"level": 3}
    assert proof[:4] == [0, 1, 2, 3]                       # rank 0's proof words first, then rank 1's
