"""GPU parity: every C-ABI entry point of libgl355 against the CPU oracle, bit-exact (integer field
arithmetic => tolerance is zero).  All calls go through include/gl355.h via ctypes.

Methodology follows the reference's own tests (random inputs, differential check of two
implementations: chip/plonk/gates/gate_test.rs:154-176, chip/hasher_chip.rs:263), with seeded RNG.
"""
import numpy as np
import pytest

from oracle_lib import P, rand_field

pytestmark = pytest.mark.gpu


def eq(a, b):
    a, b = np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        raise AssertionError("mismatch at %d/%d positions, first %s: got %x want %x" % (
            len(bad), a.size, bad[0], int(a[tuple(bad[0])]), int(b[tuple(bad[0])])))


# ---- a1 ------------------------------------------------------------------------------------------
def test_field_ops(ctx, orc):
    rng = np.random.default_rng(0x351)
    edge = np.array([0, 1, 2, P - 1, P - 2, P, P + 1, (1 << 64) - 1, 1 << 32, (1 << 32) - 1, 0xFFFFFFFF00000000],
                    dtype=np.uint64)
    a = np.concatenate([edge, np.repeat(edge, len(edge)), rng.integers(0, 1 << 64, 4000, dtype=np.uint64)])
    b = np.concatenate([edge[::-1], np.tile(edge, len(edge)), rng.integers(0, 1 << 64, 4000, dtype=np.uint64)])
    for op, f in ((0, orc.add), (1, orc.sub), (2, orc.mul)):
        got = ctx.field_batch(op, a, b)
        want = np.array([f(int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint64)
        eq(got, want)
    nz = a[(a % np.uint64(P)) != 0][:500]
    eq(ctx.field_batch(3, nz), np.array([orc.inv(int(x)) for x in nz], dtype=np.uint64))
    # extension field
    ea, eb = a[:2000].copy(), b[:2000].copy()
    got = ctx.field_batch(4, ea, eb).reshape(-1, 2)
    want = np.array([orc.ext_mul(ea[2 * i:2 * i + 2], eb[2 * i:2 * i + 2]) for i in range(1000)])
    eq(got, want)
    en = rand_field(rng, 400)
    got = ctx.field_batch(5, en).reshape(-1, 2)
    want = np.array([orc.ext_inv(en[2 * i:2 * i + 2]) for i in range(200)])
    eq(got, want)


def test_field_mul_structured_operands(ctx):
    """The device product takes its carries from the hardware (v_mad_u64_u32 carry-out, v_sub_co borrow; gl_field.cuh): the
    borrow of `lo - hi_hi` needs a 128-bit product whose low 64 bits are < 2^32 and fires for ~2^-32 of random operands, so
    it is driven here by structured ones -- all pairs of 2^k, 2^k - 1, 2^k + 1 and the edges -- against Python integers."""
    vals = {0, 1, P - 1, P, P + 1, (1 << 64) - 1, 0xFFFFFFFF00000000, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFE00000001}
    for k in range(64):
        vals.update({(1 << k) % (1 << 64), ((1 << k) - 1) % (1 << 64), ((1 << k) + 1) % (1 << 64), ((1 << 64) - (1 << k)) % (1 << 64)})
    vals = np.array(sorted(vals), dtype=np.uint64)
    a, b = np.repeat(vals, len(vals)), np.tile(vals, len(vals))
    borrow_cases = sum(1 for x, y in zip(a.tolist(), b.tolist()) if ((x * y) & ((1 << 64) - 1)) < ((x * y) >> 96))
    assert borrow_cases > 1000          # the rare path is exercised
    for op, f in ((2, lambda x, y: x * y % P), (0, lambda x, y: (x + y) % P), (1, lambda x, y: (x - y) % P)):
        got = ctx.field_batch(op, a, b)
        want = np.array([f(x, y) for x, y in zip(a.tolist(), b.tolist())], dtype=np.uint64)
        eq(got, want)


# ---- a2 ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n", list(range(0, 15)) + [15, 16, 17, 18, 20])
def test_ntt_forward_inverse(ctx, orc, log_n):
    rng = np.random.default_rng(0x355 + log_n)
    batch = 3 if log_n < 18 else 2
    x = rand_field(rng, (batch, 1 << log_n))
    fwd = ctx.fft(x)
    eq(fwd, orc.ntt(x))
    eq(ctx.ifft(fwd), x)           # round trip
    eq(ctx.ifft(x), orc.ntt(x, inverse=True))


@pytest.mark.parametrize("log_n", [1, 5, 12, 13, 14, 16])
def test_coset_ntt(ctx, orc, log_n):
    rng = np.random.default_rng(0x365 + log_n)
    x = rand_field(rng, (2, 1 << log_n))
    for shift in (7, 49, 0x123456789ABCDEF):
        eq(ctx.coset_fft(x, shift), orc.ntt(x, shift=shift))
        eq(ctx.coset_ifft(x, shift), orc.ntt(x, inverse=True, shift=shift))


def test_ntt_edge_vectors(ctx, orc):
    for log_n in (3, 12, 14, 16):
        n = 1 << log_n
        vecs = np.zeros((5, n), dtype=np.uint64)
        vecs[1, :] = P - 1
        vecs[2, 0] = 1
        vecs[3, 1] = 1
        vecs[4, n - 1] = 1
        eq(ctx.fft(vecs), orc.ntt(vecs))
        # non-canonical inputs are accepted and reduced
        nc = np.full((1, n), (1 << 64) - 1, dtype=np.uint64)
        eq(ctx.fft(nc), orc.ntt(nc))


def test_ntt_linearity_large(ctx):
    """size-independent property at BASELINE size 2^20: NTT(a + b) = NTT(a) + NTT(b)."""
    rng = np.random.default_rng(0x375)
    a, b = rand_field(rng, (1, 1 << 20)), rand_field(rng, (1, 1 << 20))
    s = ctx.field_batch(0, a.reshape(-1), b.reshape(-1)).reshape(1, -1)
    lhs = ctx.fft(s)
    rhs = ctx.field_batch(0, ctx.fft(a).reshape(-1), ctx.fft(b).reshape(-1)).reshape(1, -1)
    eq(lhs, rhs)


# ---- a3 ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n,rate_bits", [(1, 1), (2, 1), (4, 3), (10, 3), (12, 3), (13, 3), (14, 2), (15, 3), (17, 3), (9, 4), (13, 0)])
def test_lde(ctx, orc, log_n, rate_bits):
    rng = np.random.default_rng(0x385 + log_n)
    c = rand_field(rng, (3, 1 << log_n))
    want = orc.lde(c, rate_bits)
    eq(ctx.lde(c, rate_bits), want)
    eq(ctx.lde(c, rate_bits, bitrev=True), orc.reverse_index_bits(want.T.copy()).T)


@pytest.mark.parametrize("log_n,max_log", [(13, 12), (14, 12), (13, 14), (14, 14), (14, 13)])
def test_lde_small_sizes_in_one_and_two_passes(gl, orc, log_n, max_log):
    """GL355_OPT_NTT_SINGLE_PASS_MAX_LOG: the commit-path shape (natural coefficients in, bit-reversed cosets out) of 2^13 / 2^14
    points runs as two passes (12, the default since round 3: the streaming 2- / 4-row column kernel + 4096-point limb rows) or as
    one CU-filling pass (14: the radix-8 single-pass kernels); same values either way"""
    c2 = gl.Context(0)
    c2.set_option(4, max_log)
    rng = np.random.default_rng(0x386 + log_n)
    c = rand_field(rng, (5, 1 << log_n))
    for rate_bits in (1, 2, 3):
        want = orc.lde(c, rate_bits)
        eq(c2.lde(c, rate_bits, bitrev=True), orc.reverse_index_bits(want.T.copy()).T)
        eq(c2.lde(c, rate_bits), want)
    with pytest.raises(gl.Gl355Error):
        c2.set_option(4, 15)
    c2.close()


# ---- a5 ------------------------------------------------------------------------------------------
def test_transpose_and_bitrev(ctx, orc):
    rng = np.random.default_rng(0x395)
    for rows, cols in ((1, 1), (7, 5), (64, 135), (1000, 33), (4096, 20)):
        m = rng.integers(0, 1 << 64, (rows, cols), dtype=np.uint64)
        eq(ctx.transpose(m), m.T)
    for n, w in ((1, 3), (2, 1), (8, 5), (1024, 139), (1 << 14, 4)):
        m = rng.integers(0, 1 << 64, (n, w), dtype=np.uint64)
        eq(ctx.reverse_index_bits(m), orc.reverse_index_bits(m))


# ---- a6 / a7 -------------------------------------------------------------------------------------
def test_poseidon_permute(ctx, orc):
    rng = np.random.default_rng(0x3A5)
    st = np.concatenate([np.zeros((1, 12), np.uint64), np.arange(12, dtype=np.uint64)[None], np.full((1, 12), P - 1, np.uint64),
                         np.full((1, 12), (1 << 64) - 1, np.uint64), rng.integers(0, 1 << 64, (700, 12), dtype=np.uint64)])
    eq(ctx.poseidon_permute(st), orc.permute(st))


def test_poseidon_permute_many_states(ctx, orc):
    """65 536 states over the whole u64 range (non-canonical representatives included) and sparse / high-bit patterns through the block-form
    partial rounds (poseidon.cuh, round 5): every lane of many waves, both blocks, against the oracle's naive permutation"""
    rng = np.random.default_rng(0x3A6)
    st = rng.integers(0, 1 << 64, (1 << 16, 12), dtype=np.uint64)
    st[:4096] &= np.uint64(0xFFFFFFFF00000000)                     # empty low halves
    st[4096:8192] |= np.uint64(0x00000000FFFFFFFF)                 # saturated low halves
    st[8192:12288, 1:] = 0                                         # only lane 0 populated
    st[12288:16384, 0] = 0                                         # lane 0 empty
    eq(ctx.poseidon_permute(st), orc.permute(st))


@pytest.mark.parametrize("length", [0, 1, 3, 4, 5, 8, 9, 16, 17, 85, 135, 139])
def test_hash_no_pad_and_leaves(ctx, orc, length):
    rng = np.random.default_rng(0x3B5 + length)
    x = rand_field(rng, (130, length)) if length else np.zeros((130, 0), np.uint64)
    if length:
        eq(ctx.hash_no_pad(x), np.array([orc.hash_no_pad(r) for r in x]))
    eq(ctx.hash_leaves(x), np.array([orc.hash_or_noop(r) for r in x]))


def test_two_to_one(ctx, orc):
    rng = np.random.default_rng(0x3C5)
    l, r = rand_field(rng, (300, 4)), rand_field(rng, (300, 4))
    eq(ctx.two_to_one(l, r), np.array([orc.two_to_one(a, b) for a, b in zip(l, r)]))


# ---- a8 ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n,leaf_len,cap", [(0, 4, 0), (1, 4, 0), (1, 4, 1), (3, 4, 0), (3, 9, 1), (4, 135, 4), (6, 20, 4),
                                                (10, 4, 0), (10, 85, 4), (12, 7, 3), (13, 4, 4)])
def test_merkle_build_and_prove(gl, ctx, orc, log_n, leaf_len, cap):
    rng = np.random.default_rng(0x3D5 + log_n * 31 + leaf_len)
    n = 1 << log_n
    leaves = rand_field(rng, (n, leaf_len))
    t = gl.MerkleTree(ctx, leaves, cap)
    dig, capw = orc.merkle_build(leaves, cap)
    eq(t.cap, capw)
    eq(t.digests, dig)
    for idx in sorted(set([0, n - 1, n // 2, int(rng.integers(0, n))])):
        sib = t.prove(idx)
        eq(sib, orc.merkle_prove(dig, n, cap, idx))
        assert orc.merkle_verify(leaves[idx], idx, sib, capw, cap)


def test_merkle_large_root_property(gl, ctx, orc):
    """2^18 leaves of 4 (the Semaphore group-tree shape, signal.rs:40): cap equals the oracle's, and a
    random opening verifies against it."""
    rng = np.random.default_rng(0x3E5)
    leaves = rand_field(rng, (1 << 18, 4))
    t = gl.MerkleTree(ctx, leaves, 0)
    dig, cap = orc.merkle_build(leaves, 0)
    eq(t.cap, cap)
    idx = 12
    assert orc.merkle_verify(leaves[idx], idx, t.prove(idx), t.cap, 0)


# ---- a4 ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n,batch,rate_bits,cap,salted,is_coeffs", [
    (3, 2, 1, 0, False, False), (5, 7, 3, 2, True, False), (8, 20, 3, 4, True, False),
    (10, 135, 3, 4, True, False), (10, 16, 3, 4, True, True), (13, 9, 3, 4, False, False), (15, 3, 3, 4, True, False)])
def test_commit(gl, ctx, orc, log_n, batch, rate_bits, cap, salted, is_coeffs):
    rng = np.random.default_rng(0x3F5 + log_n + batch)
    n, N = 1 << log_n, 1 << (log_n + rate_bits)
    vals = rand_field(rng, (batch, n))
    salt = rand_field(rng, (4, N)) if salted else None
    fn = gl.PolynomialBatch.from_coeffs if is_coeffs else gl.PolynomialBatch.from_values
    pb = fn(ctx, vals, rate_bits, cap, salt=salt)
    coeffs, leaves, dig, capw = orc.commit(vals, rate_bits, cap, salt=salt, is_coeffs=is_coeffs)
    eq(pb.cap, capw)
    eq(pb.polynomials, coeffs)
    eq(pb.leaves(), leaves)
    eq(pb.digests(), dig)
    for idx in (0, N - 1, int(rng.integers(0, N))):
        leaf, sib = pb.open(idx)
        eq(leaf, leaves[idx])
        assert orc.merkle_verify(leaf, idx, sib, capw, cap)
    eq(pb.get_lde_values(5 % N), orc.reverse_index_bits(leaves)[5 % N][:batch])
    pb.close()


# ---- a11 -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n", [1, 4, 8, 9, 13])
def test_deep_quotient_and_openings(gl, ctx, orc, log_n):
    rng = np.random.default_rng(0x405 + log_n)
    n = 1 << log_n
    a = rand_field(rng, (11, n))
    b = rand_field(rng, (5, n))
    pa = gl.PolynomialBatch.from_coeffs(ctx, a, 1, 0)
    pb = gl.PolynomialBatch.from_coeffs(ctx, b, 1, 0)
    alpha, zeta = rand_field(rng, 2), rand_field(rng, 2)
    refs = [(pa, i) for i in range(11)] + [(pb, i) for i in (4, 0, 2)]
    flat = np.concatenate([a, b[[4, 0, 2]]])
    eq(gl.eval_polys(ctx, refs, zeta), orc.eval_polys_ext(flat, zeta))
    acc0 = np.zeros(2 * n, np.uint64)
    acc1 = gl.deep_batch(ctx, refs, alpha, zeta, acc0)
    want1 = orc.deep_batch(flat, alpha, zeta, acc0)
    eq(acc1, want1)
    # second batch accumulates on top of the first with the alpha^k shift
    gz = rand_field(rng, 2)
    refs2 = [(pb, 1), (pb, 3)]
    eq(gl.deep_batch(ctx, refs2, alpha, gz, acc1), orc.deep_batch(b[[1, 3]], alpha, gz, want1))
    eq(ctx.lde_ext(want1, 3), orc.lde_ext(want1, 3))


# ---- a12 / a13 -----------------------------------------------------------------------------------
def test_fri_fold_and_layer(ctx, orc):
    rng = np.random.default_rng(0x415)
    for log_n in (1, 2, 6, 12, 16):
        n = 1 << log_n
        c = rand_field(rng, 2 * n)
        beta = rand_field(rng, 2)
        eq(ctx.fri_fold(c, beta), orc.fri_fold(c, beta))
    for log_n, cap in ((1, 0), (5, 4), (9, 4), (12, 2)):
        n = 1 << log_n
        v = rand_field(rng, 2 * n)
        t = ctx.fri_layer_commit(v, cap)
        leaves = orc.fri_layer_leaves(v)
        dig, capw = orc.merkle_build(leaves, cap)
        eq(t.leaves, leaves)
        eq(t.cap, capw)
        eq(t.digests, dig)


def test_pow_grind(ctx, orc):
    rng = np.random.default_rng(0x425)
    for bits, pos in ((0, 0), (4, 3), (10, 7), (16, 5)):
        st = rand_field(rng, 12)
        w = ctx.pow_grind(st, pos, bits)
        assert w == orc.pow_grind(st, pos, bits)
        s2 = st.copy()
        s2[pos] = w
        assert bits == 0 or int(orc.permute(s2)[7]) >> (64 - bits) == 0
    st = rand_field(rng, 12)
    assert ctx.pow_grind(st, 2, 8, start=1000) == orc.pow_grind(st, 2, 8, start=1000)


# ---- a9 ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n,n_routed,max_degree", [(3, 8, 8), (5, 80, 8), (11, 80, 8), (6, 10, 3)])
def test_zs_partial_products(ctx, orc, log_n, n_routed, max_degree):
    rng = np.random.default_rng(0x435 + log_n)
    n = 1 << log_n
    wires, sigmas = rand_field(rng, (n_routed, n)), rand_field(rng, (n_routed, n))
    k_is = rand_field(rng, n_routed)
    beta, gamma = int(rand_field(rng, 1)[0]), int(rand_field(rng, 1)[0])
    z, pp = ctx.zs_partial_products(wires, sigmas, k_is, max_degree, beta, gamma)
    zw, ppw = orc.zs_partial_products(wires, sigmas, k_is, max_degree, beta, gamma)
    eq(z, zw)
    eq(pp, ppw)


# ---- error behaviour: codes, never aborts ----------------------------------------------------------
def test_error_codes(gl, ctx):
    with pytest.raises(gl.Gl355Error) as e:
        gl.MerkleTree(ctx, np.zeros((6, 4), np.uint64), 0)   # not a power of two
    assert e.value.code == -1
    with pytest.raises(gl.Gl355Error):
        gl.MerkleTree(ctx, np.zeros((4, 4), np.uint64), 3)   # cap higher than the tree
    with pytest.raises(gl.Gl355Error):
        ctx.pow_grind(np.zeros(12, np.uint64), 9, 4)         # witness outside the rate part
    # the context stays usable after an error
    assert ctx.hash_no_pad(np.arange(1, 9, dtype=np.uint64))[0] == 0xD110AA6A46373941


# ---- committed golden vectors through the C ABI ------------------------------------------------------
def test_golden_vectors_on_gpu(gl, ctx):
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))

    def ux(lst):
        return np.array([int(x, 16) for x in lst], dtype=np.uint64)
    kat = json.load(open(os.path.join(here, "golden", "poseidon_kat.json")))
    for v in kat["permute"]:
        eq(ctx.poseidon_permute(ux(v["input"])), ux(v["output"]))
    for v in kat["hash_no_pad"]:
        eq(ctx.hash_no_pad(ux(v["input"])), ux(v["output"]))
    conv = json.load(open(os.path.join(here, "golden", "conventions.json")))
    for v in conv["ntt"]:
        eq(ctx.fft(ux(v["input"])), ux(v["forward"]))
        eq(ctx.ifft(ux(v["input"])), ux(v["inverse"]))
    for v in conv["lde"]:
        if len(v["coeffs"]) > 1:
            eq(ctx.lde(ux(v["coeffs"]), v["rate_bits"], v["shift"]), ux(v["natural"]))
    for v in conv["merkle"]:
        leaves = np.array([[int(x, 16) for x in l] for l in v["leaves"]], dtype=np.uint64)
        t = gl.MerkleTree(ctx, leaves, v["cap_height"])
        eq(t.cap, np.array([[int(x, 16) for x in d] for d in v["cap"]], dtype=np.uint64))
        eq(t.digests, np.array([[int(x, 16) for x in d] for d in v["digests"]], dtype=np.uint64).reshape(-1, 4))


def test_aggregation_root_on_gpu(gl, ctx, orc):
    import importlib
    par = importlib.import_module("stark-verifier_amd.parallel")
    rng = np.random.default_rng(0x445)
    leaves = rand_field(rng, (1000, 8))          # 1000 proofs x (nullifier || topic), padded to 1024
    root = par.aggregation_root(ctx, leaves)
    eq(root, orc.merkle_build(par.pad_pow2(leaves), 0)[1])


def test_device_resident_buffers(gl, ctx, orc):
    """device pointers (torch tensors) are used in place, no host staging: same results."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(0x455)
    x = rand_field(rng, (4, 1 << 13))
    t = torch.from_numpy(x.view(np.int64)).cuda()
    out = torch.empty((4, 1 << 16), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ctx.check(ctx.lib.gl355_lde_bitrev(ctx.h, C.c_void_p(t.data_ptr()), 13, 3, 7, 4, C.c_void_p(out.data_ptr())))
    ctx.sync()
    want = orc.reverse_index_bits(orc.lde(x, 3).T.copy()).T
    eq(out.cpu().numpy().view(np.uint64), want)
    eq(t.cpu().numpy().view(np.uint64), x)       # input untouched


def test_oracle_open_batch(gl, ctx, orc):
    rng = np.random.default_rng(0x465)
    vals = rand_field(rng, (7, 1 << 6))
    pb = gl.PolynomialBatch.from_values(ctx, vals, 3, 2, salt=rand_field(rng, (4, 1 << 9)))
    idx = [0, 511, 77, 77, 256]
    for i, (leaf, sib) in zip(idx, pb.open_batch(idx)):
        l2, s2 = pb.open(i)
        eq(leaf, l2)
        eq(sib, s2)
        assert orc.merkle_verify(leaf, i, sib, pb.cap, 2)
    pb.close()
