"""Multi-GPU sharding of independent units (proofs / commitment batches), one process per GPU.

The reference parallelises over independent proofs with rayon (src/plonky2_semaphore/recursion.rs:300-308
`(0..num_proofs).into_par_iter()`, :211-227 `par_chunks_exact(2)`); here the same units are block-partitioned
over ranks with NO data-path collective.  The only exchange is the gather of one small leaf per unit
(nullifier || topic = 8 u64, recursion.rs:110-165, or a 4-u64 digest) for the aggregation root --
latency-bound, so a single all_gather over RCCL (backend "nccl") on GPUs, gloo in the CPU tests.
"""
import numpy as np


def shard_range(total, rank, world):
    """Static block partition: unit i -> rank i // ceil(total/world)."""
    per = (total + world - 1) // world
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def gather_leaves(local, dist=None, group=None):
    """all_gather of equally shaped [k, w] int64 tensors -> [world*k, w] in rank order on every rank."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    parts = [torch.empty_like(local) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, local.contiguous(), group=group)
    return torch.cat(parts, dim=0)


def pad_pow2(leaves):
    """Merkle trees need a power-of-two leaf count: pad with all-zero leaves (numpy [n, w])."""
    n = leaves.shape[0]
    m = 1
    while m < n:
        m *= 2
    if m == n:
        return leaves
    return np.concatenate([leaves, np.zeros((m - n, leaves.shape[1]), dtype=leaves.dtype)])


def aggregation_root(ctx, leaves, cap_height=0):
    """Poseidon-Goldilocks Merkle cap over the gathered leaves, on the GPU (rank 0)."""
    from .api import MerkleTree
    lv = pad_pow2(np.ascontiguousarray(leaves, dtype=np.uint64))
    return MerkleTree(ctx, lv, cap_height).cap


def aggregate_distributed(aggregator, local_signals, dist=None, device=None, ctxs=None, seed=1, rng=None):
    """recursion.rs:187-247 across GPUs: every rank aggregates its own block of signals into one proof (the lower
    log2(len(local_signals)) levels of the tree, no communication), the per-rank proofs -- flat words | public inputs, the wire
    format of SURVEY N3 -- are exchanged with ONE all_gather (about 0.2 MB per rank over RCCL / xGMI), and rank 0 aggregates them
    through the upper log2(world) levels.  Every rank builds the same level circuits (deterministic builder), so a proof made on
    one GPU is an input of a circuit loaded on another.  Returns (proof, public inputs, common data) on rank 0, None elsewhere."""
    import torch
    proof, pis, cd = aggregator.aggregate(local_signals, seed=seed, rng=rng, ctxs=ctxs)
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    if world == 1:
        return proof, pis, cd
    assert world & (world - 1) == 0, "the aggregation tree is binary: a power-of-two number of ranks"
    local_levels = len(local_signals).bit_length() - 1
    packed = np.concatenate([np.ascontiguousarray(proof, dtype=np.uint64), np.ascontiguousarray(pis, dtype=np.uint64)])
    t = torch.from_numpy(packed.view(np.int64)).reshape(1, -1)
    if device is not None:
        t = t.to(device)
    allp = gather_leaves(t, dist).cpu().numpy().view(np.uint64)
    if dist.get_rank() != 0:
        return None
    n_words = proof.size
    signals = [(allp[r, :n_words].copy(), allp[r, n_words:].copy()) for r in range(world)]
    return aggregator.aggregate(signals, seed=seed + 1000003, rng=rng, ctxs=ctxs, start_level=local_levels)
