"""Multi-GPU sharding of independent units (proofs / commitment batches), one process per GPU.

The reference parallelises over independent proofs with rayon (src/plonky2_semaphore/recursion.rs:300-308
`(0..num_proofs).into_par_iter()`, :211-227 `par_chunks_exact(2)`); here the same units are block-partitioned
over ranks with NO data-path collective.  The only exchange is the gather of one small leaf per unit
(nullifier || topic = 8 u64, recursion.rs:110-165, or a 4-u64 digest) for the aggregation root --
latency-bound, so a single all_gather over RCCL (backend "nccl") on GPUs, gloo in the CPU tests.
"""
import ctypes as C

import numpy as np

COMM_RCCL, COMM_HOST, COMM_ID_BYTES = 0, 1, 128


class Comm:
    """gl355_comm (include/gl355.h): the multi-GPU exchange behind the C ABI -- all-gather of per-unit leaves / per-rank proofs,
    barrier, max -- over RCCL (xGMI) or, explicitly chosen, over TCP between the host processes.  The id travels by the caller's
    own means: `exchange_id(store)` uses any key-value object with set/get (torch.distributed.TCPStore under torchrun)."""

    def __init__(self, ctx, backend, comm_id, rank, world, lib=None):
        from . import _lib
        self.lib = lib or (ctx.lib if ctx is not None else _lib.load())
        self.ctx, self.rank, self.world, self.backend = ctx, rank, world, backend
        self.h = C.c_void_p()
        rc = self.lib.gl355_comm_create(ctx.h if ctx is not None else None, backend, comm_id, rank, world, C.byref(self.h))
        if rc != 0:
            raise _lib.Gl355Error(rc, (self.lib.gl355_comm_last_error(None) or b"").decode())

    @staticmethod
    def unique_id(lib, backend=COMM_RCCL, addr="127.0.0.1", port=0):
        buf = C.create_string_buffer(COMM_ID_BYTES)
        rc = lib.gl355_comm_host_id(addr.encode(), port, buf) if backend == COMM_HOST else lib.gl355_comm_unique_id(backend, buf)
        if rc != 0:
            from . import _lib
            raise _lib.Gl355Error(rc, (lib.gl355_comm_last_error(None) or b"").decode())
        return buf.raw

    def _check(self, rc):
        if rc != 0:
            from . import _lib
            raise _lib.Gl355Error(rc, (self.lib.gl355_comm_last_error(self.h) or b"").decode())

    def gather(self, local):
        """all-gather of equally shaped uint64 arrays [k, w] -> [world * k, w] in rank order (gl355_gather_digests)"""
        loc = np.ascontiguousarray(local, dtype=np.uint64)
        out = np.empty((self.world,) + loc.shape, dtype=np.uint64)
        self._check(self.lib.gl355_gather_digests(self.h, loc.ctypes.data, loc.size, out.ctypes.data))
        return out.reshape((self.world * loc.shape[0],) + loc.shape[1:]) if loc.ndim else out

    def barrier(self):
        self._check(self.lib.gl355_comm_barrier(self.h))

    def max(self, value):
        v = C.c_double(float(value))
        self._check(self.lib.gl355_comm_max_f64(self.h, C.byref(v)))
        return v.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.gl355_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
    """Poseidon-Goldilocks Merkle cap over the gathered leaves, on the GPU (rank 0): gl355_aggregation_root for the root
    (cap height 0), the general MerkleTree for a wider cap."""
    lv = np.ascontiguousarray(leaves, dtype=np.uint64)
    if cap_height == 0:
        root = np.empty((1, 4), dtype=np.uint64)
        ctx.check(ctx.lib.gl355_aggregation_root(ctx.h, lv.ctypes.data, lv.shape[0], lv.shape[1], root.ctypes.data))
        return root
    from .api import MerkleTree
    return MerkleTree(ctx, pad_pow2(lv), cap_height).cap


def aggregate_distributed(aggregator, local_signals, dist=None, device=None, ctxs=None, seed=None, rng=None, comm=None):
    """recursion.rs:187-247 across GPUs: every rank aggregates its own block of signals into one proof (the lower
    log2(len(local_signals)) levels of the tree, no communication), the per-rank proofs -- flat words | public inputs, the wire
    format of SURVEY N3 -- are exchanged with ONE all-gather (about 0.2 MB per rank: gl355_gather_digests over RCCL / xGMI when
    `comm` is a Comm; a torch.distributed group `dist` is accepted for the gloo tests), and rank 0 aggregates them through the
    upper log2(world) levels.  Every rank builds the same level circuits (deterministic builder), so a proof made on one GPU is an
    input of a circuit loaded on another.  Returns (proof, public inputs, common data) on rank 0, None elsewhere.
    seed: None = a fresh OS-random blinding key per proof; otherwise every proof's key is derived from (seed, rank, level, node)
    -- ranks never share a blinding stream (Aggregator.aggregate, `key_domain`)."""
    if comm is not None:
        world, rank = comm.world, comm.rank
    else:
        world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
    proof, pis, cd = aggregator.aggregate(local_signals, seed=seed, rng=rng, ctxs=ctxs, key_domain=1 + rank)
    if world == 1:
        return proof, pis, cd
    assert world & (world - 1) == 0, "the aggregation tree is binary: a power-of-two number of ranks"
    local_levels = len(local_signals).bit_length() - 1
    packed = np.concatenate([np.ascontiguousarray(proof, dtype=np.uint64), np.ascontiguousarray(pis, dtype=np.uint64)])
    if comm is not None:
        allp = comm.gather(packed.reshape(1, -1))
    else:
        import torch
        t = torch.from_numpy(packed.view(np.int64)).reshape(1, -1)
        if device is not None:
            t = t.to(device)
        allp = gather_leaves(t, dist).cpu().numpy().view(np.uint64)
    if rank != 0:
        return None
    n_words = proof.size
    signals = [(allp[r, :n_words].copy(), allp[r, n_words:].copy()) for r in range(world)]
    return aggregator.aggregate(signals, seed=seed, rng=rng, ctxs=ctxs, start_level=local_levels, key_domain=0)
