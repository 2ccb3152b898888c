"""Host-side mirror of the reference's interface for the hot path, on top of the C ABI.

The reference (DoHoonKim8/stark-verifier) reaches this path through plonky2's Rust API; the names
below keep plonky2's names and argument meaning so the parity tests read like the reference's own
tests (which draw random inputs and compare two implementations):

    PoseidonHash.hash_no_pad            <- src/plonky2_semaphore/access_set.rs:67, signal.rs:35
    MerkleTree(leaves, cap_height)      <- signal.rs:40, access_set.rs:205, recursion.rs:360
    MerkleTree.prove(i)                 <- circuit.rs:91
    PolynomialBatch.from_values/from_coeffs, .get_lde_values, prove_openings pieces, fri layers
                                        <- inside CircuitBuilder::build / CircuitData::prove
                                           (access_set.rs:91,94; recursion.rs:167-168; wrapper.rs:41,55)

Arrays are numpy uint64 (host; staged by the library) or torch uint64/int64 CUDA tensors (device;
used in place).  Nothing here computes field arithmetic on the CPU: every operation is a call into
libgl355.so, and importing this module without the built library raises.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Gl355Error, PolyRef

P = (1 << 64) - (1 << 32) + 1
COSET_SHIFT = 7
SALT_SIZE = 4


def _ptr(x):
    """void* of a numpy array, a torch tensor, a raw int pointer or None."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        assert x.dtype == np.uint64 and x.flags["C_CONTIGUOUS"], "need C-contiguous uint64"
        return x.ctypes.data_as(C.c_void_p)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        assert x.is_contiguous()
        return C.c_void_p(x.data_ptr())
    raise TypeError("unsupported buffer type %r" % type(x))


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


HASH_POSEIDON, HASH_BN254_POSEIDON = 0, 1   # GL355_HASH_*


class Context:
    """One (device, stream) binding: gl355_ctx."""

    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        if stream is None:
            rc = self.lib.gl355_ctx_create(device, C.byref(h))
        else:
            rc = self.lib.gl355_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h))
        if rc != 0:
            raise Gl355Error(rc, (self.lib.gl355_last_error(None) or b"").decode())
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.gl355_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise Gl355Error(rc, (self.lib.gl355_last_error(self.h) or b"").decode())

    def sync(self):
        self.check(self.lib.gl355_ctx_sync(self.h))

    def timer_start(self):
        self.check(self.lib.gl355_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self.check(self.lib.gl355_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def set_option(self, option, value):
        self.check(self.lib.gl355_ctx_set_option(self.h, int(option), int(value)))

    def profile_enable(self, on=True):
        self.check(self.lib.gl355_profile_enable(self.h, int(on)))

    def profile_read(self):
        """{name: (count, total_ms, algorithmic_bytes)} of the kernel groups run since the last read."""
        buf = C.create_string_buffer(1 << 16)
        self.check(self.lib.gl355_profile_read(self.h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms, nbytes = line.split()
            out[name] = (int(cnt), float(ms), int(nbytes))
        return out

    # ---- a1 --------------------------------------------------------------------------------
    def field_batch(self, op, a, b=None):
        a = _u64(a)
        out = np.empty_like(a)
        n = a.size // (2 if op >= 4 else 1)
        self.check(self.lib.gl355_field_batch(self.h, op, _ptr(a), _ptr(_u64(b)) if b is not None else None, _ptr(out), n))
        return out

    # ---- a2 / a3: plonky2_field::fft ---------------------------------------------------------
    def fft(self, values, inverse=False, shift=None):
        """fft_with_options / ifft_with_options (+ coset variants) on columns: values[batch][n]."""
        v = _u64(values).copy()
        v2 = v.reshape(1, -1) if v.ndim == 1 else v
        batch, n = v2.shape
        log_n = int(n).bit_length() - 1
        assert 1 << log_n == n
        if shift is None:
            self.check(self.lib.gl355_ntt(self.h, _ptr(v2), log_n, batch, n, int(inverse)))
        else:
            self.check(self.lib.gl355_coset_ntt(self.h, _ptr(v2), log_n, batch, n, shift, int(inverse)))
        return v

    def ifft(self, values):
        return self.fft(values, inverse=True)

    def coset_fft(self, coeffs, shift=COSET_SHIFT):
        return self.fft(coeffs, inverse=False, shift=shift)

    def coset_ifft(self, values, shift=COSET_SHIFT):
        return self.fft(values, inverse=True, shift=shift)

    def lde(self, coeffs, rate_bits, shift=COSET_SHIFT, bitrev=False):
        """PolynomialCoeffs::lde(rate_bits).coset_fft(shift): [batch][n] -> [batch][n << rate_bits]."""
        c = _u64(coeffs)
        c2 = c.reshape(1, -1) if c.ndim == 1 else c
        batch, n = c2.shape
        log_n = int(n).bit_length() - 1
        out = np.empty((batch, n << rate_bits), dtype=np.uint64)
        fn = self.lib.gl355_lde_bitrev if bitrev else self.lib.gl355_lde
        self.check(fn(self.h, _ptr(c2), log_n, rate_bits, shift, batch, _ptr(out)))
        return out.reshape(-1) if c.ndim == 1 else out

    def lde_ext(self, coeffs_ext, rate_bits, shift=COSET_SHIFT):
        c = _u64(coeffs_ext)
        n = c.size // 2
        out = np.empty(2 * (n << rate_bits), dtype=np.uint64)
        self.check(self.lib.gl355_lde_ext(self.h, _ptr(c), int(n).bit_length() - 1, rate_bits, shift, _ptr(out)))
        return out

    # ---- a5: plonky2_util ----------------------------------------------------------------------
    def transpose(self, m):
        m = _u64(m)
        rows, cols = m.shape
        out = np.empty((cols, rows), dtype=np.uint64)
        self.check(self.lib.gl355_transpose(self.h, _ptr(m), rows, cols, _ptr(out)))
        return out

    def reverse_index_bits(self, rows):
        r = _u64(rows).copy()
        r2 = r.reshape(-1, 1) if r.ndim == 1 else r
        self.check(self.lib.gl355_reverse_index_bits(self.h, _ptr(r2), r2.shape[0], r2.shape[1]))
        return r

    # ---- a6 / a7: plonky2::hash ------------------------------------------------------------------
    def poseidon_permute(self, states):
        s = _u64(states).copy()
        self.check(self.lib.gl355_poseidon_permute(self.h, _ptr(s), s.size // 12))
        return s

    def hash_no_pad(self, inputs):
        """PoseidonHash::hash_no_pad on each row of inputs[n][len] -> [n][4]."""
        x = _u64(inputs)
        x2 = x.reshape(1, -1) if x.ndim == 1 else x
        out = np.empty((x2.shape[0], 4), dtype=np.uint64)
        self.check(self.lib.gl355_hash_no_pad(self.h, _ptr(x2), x2.shape[0], x2.shape[1], _ptr(out)))
        return out[0] if x.ndim == 1 else out

    def hash_leaves(self, leaves):
        x = _u64(leaves)
        out = np.empty((x.shape[0], 4), dtype=np.uint64)
        self.check(self.lib.gl355_hash_leaves(self.h, _ptr(x), x.shape[0], x.shape[1], _ptr(out)))
        return out

    def two_to_one(self, left, right):
        l, r = _u64(left).reshape(-1, 4), _u64(right).reshape(-1, 4)
        out = np.empty_like(l)
        self.check(self.lib.gl355_two_to_one(self.h, _ptr(l), _ptr(r), l.shape[0], _ptr(out)))
        return out

    # ---- a12 / a13 ---------------------------------------------------------------------------------
    def fri_fold(self, coeffs_ext, beta):
        c = _u64(coeffs_ext)
        n = c.size // 2
        out = np.empty(n, dtype=np.uint64)  # n/2 ext = n u64
        b = _u64(beta)
        self.check(self.lib.gl355_fri_fold(self.h, _ptr(c), n, _ptr(b), _ptr(out)))
        return out

    def fri_layer_commit(self, values_ext, cap_height):
        v = _u64(values_ext)
        n = v.size // 2
        leaves = np.empty((n // 2, 4), dtype=np.uint64)
        digests = np.empty((2 * (n // 2 - (1 << cap_height)), 4), dtype=np.uint64)
        cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
        self.check(self.lib.gl355_fri_layer_commit(self.h, _ptr(v), n, cap_height, _ptr(leaves), _ptr(digests), _ptr(cap)))
        return MerkleTree._from_parts(self, leaves, digests, cap, cap_height)

    def pow_grind(self, state, pos, bits, start=0):
        s = _u64(state)
        w = C.c_uint64()
        self.check(self.lib.gl355_pow_grind(self.h, _ptr(s), pos, bits, start, C.byref(w)))
        return w.value

    # ---- a9 ------------------------------------------------------------------------------------------
    def zs_partial_products(self, wires, sigmas, k_is, max_degree, beta, gamma):
        w, s, k = _u64(wires), _u64(sigmas), _u64(k_is)
        n_routed, n = w.shape
        n_chunks = (n_routed + max_degree - 1) // max_degree
        z = np.empty(n, dtype=np.uint64)
        pp = np.empty((n_chunks - 1, n), dtype=np.uint64)
        self.check(self.lib.gl355_zs_partial_products(self.h, _ptr(w), _ptr(s), _ptr(k), int(n).bit_length() - 1,
                                                      n_routed, max_degree, beta, gamma, _ptr(z), _ptr(pp)))
        return z, pp


class _Hasher:
    """plonky2 `Hasher<F>`: hash_no_pad, two_to_one, the permutation of the 12-element sponge state"""
    ID = HASH_POSEIDON

    def __init__(self, ctx):
        self.ctx = ctx

    def permute(self, states):
        st = _u64(states).copy()
        s2 = st.reshape(-1, 12)
        self.ctx.check(self.ctx.lib.gl355_permute_h(self.ctx.h, self.ID, _ptr(s2), s2.shape[0]))
        return st

    def hash_no_pad(self, inputs):
        x = _u64(inputs)
        x2 = x.reshape(1, -1) if x.ndim == 1 else x
        out = np.empty((x2.shape[0], 4), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.gl355_hash_no_pad_h(self.ctx.h, self.ID, _ptr(x2), x2.shape[0], x2.shape[1], _ptr(out)))
        return out[0] if x.ndim == 1 else out

    def hash_leaves(self, leaves):
        x = _u64(leaves)
        out = np.empty((x.shape[0], 4), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.gl355_hash_leaves_h(self.ctx.h, self.ID, _ptr(x), x.shape[0], x.shape[1], _ptr(out)))
        return out

    def two_to_one(self, left, right):
        l, r = _u64(left).reshape(-1, 4), _u64(right).reshape(-1, 4)
        out = np.empty_like(l)
        self.ctx.check(self.ctx.lib.gl355_two_to_one_h(self.ctx.h, self.ID, _ptr(l), _ptr(r), l.shape[0], _ptr(out)))
        return out


class PoseidonHash(_Hasher):
    """plonky2::hash::poseidon::PoseidonHash (the reference's `C::Hasher`)."""
    ID = HASH_POSEIDON


class Bn254PoseidonHash(_Hasher):
    """the reference's Bn254PoseidonHash (src/plonky2_verifier/bn245_poseidon/plonky2_config.rs:57-75), the hasher of
    `Bn254PoseidonGoldilocksConfig` = OuterC (access_set.rs:48-49, recursion.rs:333-335, wrapper.rs:35-56)"""
    ID = HASH_BN254_POSEIDON


class MerkleTree:
    """plonky2::hash::merkle_tree::MerkleTree { leaves, digests, cap } built on the GPU."""

    def __init__(self, ctx, leaves, cap_height, hasher=HASH_POSEIDON):
        """MerkleTree::new::<F, H>(leaves, cap_height); hasher = HASH_POSEIDON | HASH_BN254_POSEIDON (or a _Hasher class)"""
        lv = _u64(leaves)
        n, leaf_len = lv.shape
        self.hasher = getattr(hasher, "ID", hasher)
        self.ctx, self.leaves, self.cap_height = ctx, lv, cap_height
        self.digests = np.empty((max(0, 2 * (n - (1 << cap_height))), 4), dtype=np.uint64)
        self.cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
        ctx.check(ctx.lib.gl355_merkle_build_h(ctx.h, self.hasher, _ptr(lv), n, leaf_len, cap_height, _ptr(self.digests), _ptr(self.cap)))

    @classmethod
    def new(cls, ctx, leaves, cap_height, hasher=HASH_POSEIDON):
        return cls(ctx, leaves, cap_height, hasher)

    @classmethod
    def _from_parts(cls, ctx, leaves, digests, cap, cap_height):
        t = cls.__new__(cls)
        t.ctx, t.leaves, t.digests, t.cap, t.cap_height = ctx, leaves, digests, cap, cap_height
        return t

    def get(self, i):
        return self.leaves[i]

    def prove_host(self, leaf_index):
        """the same siblings read straight from the host copy of `digests` (plonky2's MerkleTree::prove indexing); no
        library call, so many threads may use one tree"""
        n = self.leaves.shape[0]
        layers = (int(n).bit_length() - 1) - self.cap_height
        tree_len = 2 * ((n >> self.cap_height) - 1)
        base = (leaf_index >> layers) * tree_len
        pair = leaf_index & ((1 << layers) - 1)
        sib = np.empty((layers, 4), dtype=np.uint64)
        for i in range(layers):
            parity = pair & 1
            pair >>= 1
            slot = (pair << (i + 1)) + (1 << i) - 1
            sib[i] = self.digests[base + 2 * slot + (1 - parity)]
        return sib

    def prove(self, leaf_index):
        n = self.leaves.shape[0]
        layers = (int(n).bit_length() - 1) - self.cap_height
        sib = np.empty((layers, 4), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.gl355_merkle_prove(self.ctx.h, _ptr(self.digests), n, self.cap_height, leaf_index, _ptr(sib)))
        return sib


class PolynomialBatch:
    """plonky2::fri::oracle::PolynomialBatch, resident in HBM (gl355_oracle)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        vals = [C.c_uint32() for _ in range(5)]
        ctx.check(ctx.lib.gl355_oracle_info(handle, *[C.byref(v) for v in vals]))
        self.degree_log, self.rate_bits, self.batch, self.leaf_len, self.cap_height = [v.value for v in vals]
        self.blinding = self.leaf_len != self.batch

    @classmethod
    def _commit(cls, ctx, data, rate_bits, salt, cap_height, is_coeffs, hasher=HASH_POSEIDON):
        d = data if hasattr(data, "data_ptr") else _u64(data)
        batch, n = d.shape
        s = None if salt is None else (salt if hasattr(salt, "data_ptr") else _u64(salt))
        h = C.c_void_p()
        ctx.check(ctx.lib.gl355_commit_h(ctx.h, getattr(hasher, "ID", hasher), _ptr(d), int(n).bit_length() - 1, batch, rate_bits,
                                         int(is_coeffs), _ptr(s), cap_height, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_values(cls, ctx, values, rate_bits, cap_height, salt=None, hasher=HASH_POSEIDON):
        return cls._commit(ctx, values, rate_bits, salt, cap_height, False, hasher)

    @classmethod
    def from_coeffs(cls, ctx, coeffs, rate_bits, cap_height, salt=None, hasher=HASH_POSEIDON):
        return cls._commit(ctx, coeffs, rate_bits, salt, cap_height, True, hasher)

    def close(self):
        if getattr(self, "h", None):
            # the oracle's memory belongs to its context's pool: after the context is gone (e.g. an object kept alive by a
            # reference cycle until interpreter shutdown) there is nothing left to release, and touching it would crash
            if getattr(self.ctx, "h", None):
                self.ctx.lib.gl355_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def lde_size(self):
        return 1 << (self.degree_log + self.rate_bits)

    @property
    def cap(self):
        out = np.empty((1 << self.cap_height, 4), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.gl355_oracle_cap(self.h, _ptr(out)))
        return out

    @property
    def polynomials(self):
        out = np.empty((self.batch, 1 << self.degree_log), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.gl355_oracle_coeffs(self.h, _ptr(out)))
        return out

    def leaves(self):
        out = np.empty((self.lde_size, self.leaf_len), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.gl355_oracle_leaves(self.h, _ptr(out)))
        return out

    def digests(self):
        out = np.empty((2 * (self.lde_size - (1 << self.cap_height)), 4), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.gl355_oracle_digests(self.h, _ptr(out)))
        return out

    def open(self, index):
        """(leaf, siblings) = (MerkleTree::get(index), MerkleTree::prove(index))."""
        leaf = np.empty(self.leaf_len, dtype=np.uint64)
        sib = np.empty((self.degree_log + self.rate_bits - self.cap_height, 4), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.gl355_oracle_open(self.h, index, _ptr(leaf), _ptr(sib)))
        return leaf, sib

    def open_batch(self, indices):
        """[(leaf, siblings)] for every index, one kernel + one copy (fri_prover_query_rounds)."""
        idx = _u64(indices)
        layers = self.degree_log + self.rate_bits - self.cap_height
        leaves = np.empty((idx.size, self.leaf_len), dtype=np.uint64)
        sib = np.empty((idx.size, layers, 4), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.gl355_oracle_open_batch(self.h, _ptr(idx), idx.size, _ptr(leaves), _ptr(sib)))
        return [(leaves[i], sib[i]) for i in range(idx.size)]

    def get_lde_values(self, index, step=1):
        """PolynomialBatch::get_lde_values: leaf bitrev(index*step) without the salt."""
        bits = self.degree_log + self.rate_bits
        i = int(format(index * step, "0%db" % bits)[::-1], 2)
        leaf, _ = self.open(i)
        return leaf[: self.batch]


def _poly_refs(polys):
    arr = (PolyRef * len(polys))()
    for i, (batch, col) in enumerate(polys):
        arr[i].oracle = batch.h
        arr[i].column = col
    return arr


def deep_batch(ctx, polys, alpha, z, acc):
    """One batch of PolynomialBatch::prove_openings: acc <- acc*alpha^k + (C - C(z))/(X - z)."""
    acc = _u64(acc).copy()
    a, zz = _u64(alpha), _u64(z)
    ctx.check(ctx.lib.gl355_deep_batch(ctx.h, _poly_refs(polys), len(polys), _ptr(a), _ptr(zz), _ptr(acc)))
    return acc


def eval_polys(ctx, polys, z):
    out = np.empty((len(polys), 2), dtype=np.uint64)
    zz = _u64(z)
    ctx.check(ctx.lib.gl355_eval_polys(ctx.h, _poly_refs(polys), len(polys), _ptr(zz), _ptr(out)))
    return out


def rand_field(rng, shape):
    """synthetic field elements, uniform in [0, p) by rejection (numpy Generator `rng`)"""
    a = rng.integers(0, 1 << 64, size=shape, dtype=np.uint64, endpoint=False)
    bad = a >= np.uint64(P)
    while bad.any():
        a[bad] = rng.integers(0, 1 << 64, size=int(bad.sum()), dtype=np.uint64, endpoint=False)
        bad = a >= np.uint64(P)
    return a
