"""GPU: SURVEY 8(f) N4 first slice -- bn256::Fr FFT and bn256::G1 MSM kernels against the oracle (pinned by halo2curves'
ROOT_OF_UNITY and the EIP-196 2*G vector, tests/test_bn254_curve_oracle.py), bit-exact; at the reference's sizes (k = 20, MSM of
2^20 points) through properties that need no slow oracle pass: round trip / linearity / known spectrum for the FFT, and for the
MSM points that are known multiples of G, so the result is one scalar multiplication."""
import numpy as np
import pytest

import pymodel_bn254_curve as pm
from oracle_lib import Bn254Curve

pytestmark = pytest.mark.gpu


def rand_scalars(rng, n, below_r=True):
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64, endpoint=False)
    if below_r:
        a[:, 3] &= np.uint64((1 << 60) - 1)              # < 2^252 < r
    return a


def gpu_ntt(ctx, a, inverse=False):
    d = np.ascontiguousarray(a, dtype=np.uint64).copy()
    ctx.check(ctx.lib.gl355_bn254_fr_ntt(ctx.h, d.ctypes.data, int(d.shape[0]).bit_length() - 1, int(inverse)))
    return d


def gpu_msm(ctx, pts, sc):
    pts, sc = np.ascontiguousarray(pts, dtype=np.uint64), np.ascontiguousarray(sc, dtype=np.uint64)
    out = np.zeros(8, dtype=np.uint64)
    ctx.check(ctx.lib.gl355_bn254_g1_msm(ctx.h, pts.ctypes.data, sc.ctypes.data, pts.shape[0], out.ctypes.data))
    return out


@pytest.mark.parametrize("log_n", [1, 2, 5, 9, 12, 16])
def test_fr_ntt_vs_oracle(ctx, orc, log_n):
    cv = Bn254Curve(orc)
    rng = np.random.default_rng(0x4E0 + log_n)
    a = rand_scalars(rng, 1 << log_n, below_r=False)          # any 256-bit value is accepted and reduced
    a[0] = 0
    a[-1] = cv.scalars([pm.R - 1])[0]
    assert np.array_equal(gpu_ntt(ctx, a), cv.ntt_array(a))
    assert np.array_equal(gpu_ntt(ctx, a, inverse=True), cv.ntt_array(a, inverse=True))


def test_fr_ntt_k20_properties(ctx, orc):
    cv = Bn254Curve(orc)
    n = 1 << 20
    rng = np.random.default_rng(0x4E1)
    a = rand_scalars(rng, n)
    fa = gpu_ntt(ctx, a)
    assert np.array_equal(gpu_ntt(ctx, fa, inverse=True), a)                      # round trip
    imp = np.zeros((n, 4), dtype=np.uint64)
    imp[1, 0] = 1
    spec = gpu_ntt(ctx, imp)                                                     # spectrum of the shifted impulse = powers of omega
    w = pm.omega(20)
    for k in (0, 1, 2, 3, 12345, n // 2, n - 1):
        assert cv.ints(spec[k])[0] == pow(w, k, pm.R)
    # linearity on sampled outputs: F(a + imp)[k] = F(a)[k] + w^k
    b = a.copy()
    s = cv.ints(b[1])[0] + 1
    b[1] = cv.scalars([s])[0]
    fb = gpu_ntt(ctx, b)
    for k in (0, 7, 99999, n - 1):
        assert cv.ints(fb[k])[0] == (cv.ints(fa[k])[0] + pow(w, k, pm.R)) % pm.R


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 300, 1025, 4096])
def test_g1_msm_vs_oracle(ctx, orc, n):
    cv = Bn254Curve(orc)
    rng = np.random.default_rng(0x4E2 + n)
    pts = cv.multiples_array(int(rng.integers(1, 1 << 40)), int(rng.integers(1, 1 << 40)), n)
    sc = rand_scalars(rng, n, below_r=False)
    if n > 3:
        pts[2] = 0                        # the identity among the bases
        sc[1] = 0                         # a zero scalar
        pts[3] = pts[0]                   # a repeated base
    assert np.array_equal(gpu_msm(ctx, pts, sc), cv.msm_arrays(pts, sc))


def test_g1_msm_edge_cases(ctx, orc):
    cv = Bn254Curve(orc)
    g = cv._pt(pm.G)
    one = cv.scalars([1])
    assert cv._unpt(gpu_msm(ctx, g.reshape(1, 8), one)) == pm.G
    assert cv._unpt(gpu_msm(ctx, g.reshape(1, 8), cv.scalars([2]))) == pm.EIP196_2G          # published vector
    assert cv._unpt(gpu_msm(ctx, g.reshape(1, 8), cv.scalars([pm.R]))) is None                 # r * G = identity
    # P + (-P) and P + P inside one bucket
    neg = cv._pt((1, pm.Q - 2))
    assert cv._unpt(gpu_msm(ctx, np.stack([g, neg]), cv.scalars([5, 5]))) is None
    assert cv._unpt(gpu_msm(ctx, np.stack([g, g]), cv.scalars([5, 5]))) == pm.mul(pm.G, 10)
    assert ctx.lib.gl355_bn254_g1_msm(ctx.h, None, None, 4, g.ctypes.data) == -1


def test_g1_msm_2p20_known_multiples(ctx, orc):
    """2^20 bases (i + 1) * 7 G and random scalars: the MSM must equal (sum_i s_i * 7 (i + 1) mod r) * G"""
    cv = Bn254Curve(orc)
    n = 1 << 20
    rng = np.random.default_rng(0x4E3)
    pts = cv.multiples_array(7, 7, n)
    sc = rand_scalars(rng, n)
    vals = np.zeros(n, dtype=object)
    for limb in range(4):
        vals += sc[:, limb].astype(object) << (64 * limb)
    k = int(sum(int(v) * (7 * (i + 1)) for i, v in enumerate(vals)) % pm.R)
    assert cv._unpt(gpu_msm(ctx, pts, sc)) == cv.mul(pm.G, k)


@pytest.mark.parametrize("n", [8193, 12345, 16384, 16385, 100003, (1 << 18) + 1])
def test_g1_msm_two_level_sort_sizes(ctx, orc, n):
    """sizes around the switch to the two-level sort (more than 2^10 buckets per window: n > 2^13) and away from powers of two, with an
    identity, a zero scalar, a repeated and an opposite base: bases (i + 1) * 11 G, so the answer is one scalar multiplication"""
    cv = Bn254Curve(orc)
    rng = np.random.default_rng(0x4E9 + n)
    pts = cv.multiples_array(11, 11, n)
    sc = rand_scalars(rng, n)
    vals = np.zeros(n, dtype=object)
    for limb in range(4):
        vals += sc[:, limb].astype(object) << (64 * limb)
    mult = [11 * (i + 1) for i in range(n)]
    pts[2] = 0; mult[2] = 0                                   # the identity among the bases
    sc[5] = 0; vals[5] = 0                                    # a zero scalar
    pts[7] = pts[0]; mult[7] = mult[0]                        # a repeated base
    pts[9, 4:] = cv.scalars([pm.Q - cv.ints(pts[1:2, 4:])[0]])[0]; pts[9, :4] = pts[1, :4]; mult[9] = -mult[1]    # the opposite of base 1
    # ... and pairs that meet in the SAME bucket of every window: a base twice under one scalar (the sum doubles), a base and its opposite
    # under one scalar (the sum is the identity, and the bucket goes on from there)
    pts[11] = pts[10]; mult[11] = mult[10]; sc[11] = sc[10]; vals[11] = vals[10]
    pts[13, 4:] = cv.scalars([pm.Q - cv.ints(pts[12:13, 4:])[0]])[0]; pts[13, :4] = pts[12, :4]; mult[13] = -mult[12]; sc[13] = sc[12]; vals[13] = vals[12]
    k = int(sum(int(v) * m for v, m in zip(vals, mult)) % pm.R)
    assert cv._unpt(gpu_msm(ctx, pts, sc)) == cv.mul(pm.G, k)


def gpu_fixed_base(ctx, base, sc):
    base, sc = np.ascontiguousarray(base, dtype=np.uint64), np.ascontiguousarray(sc, dtype=np.uint64)
    out = np.full((sc.shape[0], 8), 0xAA, dtype=np.uint64)
    ctx.check(ctx.lib.gl355_bn254_g1_fixed_base_mul(ctx.h, base.ctypes.data, sc.ctypes.data, sc.shape[0], out.ctypes.data))
    return out


def test_g1_fixed_base_mul_vs_oracle(ctx, orc):
    """out[i] = scalars[i] * base (the powers-of-tau loop of ParamsKZG::setup) against the oracle's double-and-add, edge scalars included"""
    cv = Bn254Curve(orc)
    rng = np.random.default_rng(0x4E4)
    for base in (pm.G, cv.mul(pm.G, 0xC0FFEE1234567)):
        vals = [0, 1, 2, 255, 256, pm.R - 1, pm.R, pm.R + 1, (1 << 256) - 1, 1 << 248, 0xFF << 120]
        sc = np.concatenate([cv.scalars(vals), rand_scalars(rng, 200, below_r=False)])
        out = gpu_fixed_base(ctx, cv._pt(base), sc)
        ints = cv.ints(sc)
        for i in list(range(len(vals))) + [len(vals) + 3, len(vals) + 77, len(sc) - 1]:
            assert cv._unpt(out[i]) == cv.mul(base, ints[i] % pm.R), i
        assert cv._unpt(out[2]) == (pm.EIP196_2G if base == pm.G else cv.add(base, base))       # published vector for 2 G
    ident = gpu_fixed_base(ctx, np.zeros(8, dtype=np.uint64), rand_scalars(rng, 5))
    assert not ident.any()                                                                       # k * identity = identity
    assert ctx.lib.gl355_bn254_g1_fixed_base_mul(ctx.h, None, None, 4, None) == -1


def test_g1_msm_2p20_distinct_bases_linearity(ctx, orc):
    """2^20 distinct bases s_i * G made by the fixed-base kernel (so the gathers of the bucket phase are not cache hits) and random
    scalars a, b: MSM(a) + MSM(b) = MSM(a + b mod r), and MSM(a) = (sum a_i s_i) * G"""
    cv = Bn254Curve(orc)
    n = 1 << 20
    rng = np.random.default_rng(0x4E5)
    s = rand_scalars(rng, n)
    pts = gpu_fixed_base(ctx, cv._pt(pm.G), s)
    a, b = rand_scalars(rng, n), rand_scalars(rng, n)

    def to_obj(x):
        v = np.zeros(n, dtype=object)
        for limb in range(4):
            v += x[:, limb].astype(object) << (64 * limb)
        return v

    sv, av, bv = to_obj(s), to_obj(a), to_obj(b)
    ma, mb = cv._unpt(gpu_msm(ctx, pts, a)), cv._unpt(gpu_msm(ctx, pts, b))
    assert ma == cv.mul(pm.G, int(sum(int(x) * int(y) for x, y in zip(av, sv)) % pm.R))
    ab = cv.scalars([(int(x) + int(y)) % pm.R for x, y in zip(av, bv)])
    assert cv._unpt(gpu_msm(ctx, pts, ab)) == cv.add(ma, mb)


@pytest.mark.parametrize("kind", ["all_equal", "zero_one", "small", "two_values_2p18", "mid_buckets", "mid_and_big"])
def test_g1_msm_skewed_scalars(ctx, orc, kind):
    """scalars that put most points into a handful of buckets (selector columns of 0 / 1, constants, small values): those buckets are
    summed by whole workgroups (one lane per bucket would walk 10^5 dependent additions); bases (i + 1) * 3 G, so the answer is one
    scalar multiplication"""
    cv = Bn254Curve(orc)
    n = 1 << (18 if kind == "two_values_2p18" else 16)
    rng = np.random.default_rng(0x4E6)
    pts = cv.multiples_array(3, 3, n)
    if kind == "all_equal":
        vals = [0x1234567890ABCDEF1122334455667788990011223344556677] * n
    elif kind == "zero_one":
        vals = [int(v) for v in rng.integers(0, 2, size=n)]
    elif kind == "small":
        vals = [int(v) for v in rng.integers(0, 5, size=n)]
    elif kind == "mid_buckets":          # 64 buckets of ~1024 points in the lowest window: the lane-item path of mid-size buckets (257 .. 2048 points)
        vals = [int(v) for v in rng.integers(1, 65, size=n)]
    elif kind == "mid_and_big":          # bucket sizes from a few points up to ~16 000 in one window, negative digits included
        vals = [int(v) if v % 3 else pm.R - int(v) for v in np.minimum(rng.geometric(0.25, size=n), 40)]
    else:
        vals = [pm.R - 1 if v else 7 for v in rng.integers(0, 2, size=n)]
    sc = cv.scalars(vals)
    k = sum(v * 3 * (i + 1) for i, v in enumerate(vals)) % pm.R
    assert cv._unpt(gpu_msm(ctx, pts, sc)) == cv.mul(pm.G, k)


def test_g1_msm_batch_equals_single_msms(ctx, orc):
    """several scalar sets over the same bases in one call (commitments of several columns under one SRS) == the MSMs one by one;
    small case against the oracle, 2^18 bases against single calls"""
    cv = Bn254Curve(orc)
    rng = np.random.default_rng(0x4E7)
    for n, m in ((300, 3), (1 << 18, 4)):
        pts = cv.multiples_array(5, 11, n)
        sc = np.stack([rand_scalars(rng, n, below_r=False) for _ in range(m)])
        if n == 300:
            sc[1, :, :] = 0                                              # an all-zero column
            sc[2, 7] = cv.scalars([pm.R - 1])[0]
        out = np.zeros((m, 8), dtype=np.uint64)
        ctx.check(ctx.lib.gl355_bn254_g1_msm_batch(ctx.h, pts.ctypes.data, np.ascontiguousarray(sc).ctypes.data, n, m, out.ctypes.data))
        for j in range(m):
            ref = cv.msm_arrays(pts, sc[j]) if n == 300 else gpu_msm(ctx, pts, sc[j])
            assert np.array_equal(out[j], ref), (n, j)
    assert ctx.lib.gl355_bn254_g1_msm_batch(ctx.h, pts.ctypes.data, sc.ctypes.data, 4, 65, out.ctypes.data) == -5       # GL355_E_UNSUPPORTED


class PreparedBases:
    """gl355_bn254_g1_msm_prepare / _msm_prepared / _msm_bases_free around one base set"""
    def __init__(self, ctx, pts):
        import ctypes
        self.ctx, self.n = ctx, pts.shape[0]
        self.h = ctypes.c_void_p()
        pts = np.ascontiguousarray(pts, dtype=np.uint64)
        ctx.check(ctx.lib.gl355_bn254_g1_msm_prepare(ctx.h, pts.ctypes.data, self.n, ctypes.byref(self.h)))

    def msm(self, sc):
        sc = np.ascontiguousarray(sc, dtype=np.uint64)
        m = 1 if sc.ndim == 2 else sc.shape[0]
        out = np.zeros((m, 8), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.gl355_bn254_g1_msm_prepared(self.ctx.h, self.h, sc.ctypes.data, m, out.ctypes.data))
        return out[0] if sc.ndim == 2 else out

    def close(self):
        self.ctx.check(self.ctx.lib.gl355_bn254_g1_msm_bases_free(self.ctx.h, self.h))


@pytest.mark.parametrize("n", [1, 3, 64, 300, 4096])
def test_g1_msm_prepared_vs_oracle(ctx, orc, n):
    """prepared bases (window multiples tabulated once, all windows into one bucket set) == the oracle's MSM, with an identity, a zero scalar,
    a repeated base, full-width scalars (>= r included) and several scalar sets per call"""
    cv = Bn254Curve(orc)
    rng = np.random.default_rng(0x4F2 + n)
    pts = cv.multiples_array(int(rng.integers(1, 1 << 40)), int(rng.integers(1, 1 << 40)), n)
    sc = np.stack([rand_scalars(rng, n, below_r=False) for _ in range(3)])
    if n > 3:
        pts[2] = 0
        sc[0, 1] = 0
        pts[3] = pts[0]
        sc[1, :, :] = 0                                    # an all-zero set
        sc[2, :, 1:] = 0; sc[2, :, 0] &= np.uint64(0xFFFF)     # 16-bit values (a range-check column)
    pb = PreparedBases(ctx, pts)
    try:
        assert np.array_equal(pb.msm(sc[0]), cv.msm_arrays(pts, sc[0]))
        out = pb.msm(sc)
        for j in range(3):
            assert np.array_equal(out[j], cv.msm_arrays(pts, sc[j])), j
    finally:
        pb.close()


@pytest.mark.parametrize("n", [16385, 100003, 1 << 18])
def test_g1_msm_prepared_equals_plain(ctx, orc, n):
    """at sizes past the oracle's reach: prepared == the per-window MSM on the same inputs (uniform, skewed and short scalars; the identity,
    equal and opposite bases under one scalar), and bases (i + 1) * 11 G so that one of them is also checked as a single scalar multiplication"""
    cv = Bn254Curve(orc)
    rng = np.random.default_rng(0x4F3 + n)
    pts = cv.multiples_array(11, 11, n)
    mult = [11 * (i + 1) for i in range(n)]
    pts[2] = 0; mult[2] = 0
    pts[7] = pts[0]; mult[7] = mult[0]
    pts[13, 4:] = cv.scalars([pm.Q - cv.ints(pts[12:13, 4:])[0]])[0]; pts[13, :4] = pts[12, :4]; mult[13] = -mult[12]
    uni = rand_scalars(rng, n)
    uni[13] = uni[12]; uni[7] = uni[0]
    vals = np.zeros(n, dtype=object)
    for limb in range(4):
        vals += uni[:, limb].astype(object) << (64 * limb)
    skew = cv.scalars([pm.R - 1 if v else 7 for v in rng.integers(0, 2, size=n)])
    short = np.zeros((n, 4), dtype=np.uint64); short[:, 0] = rng.integers(0, 1 << 16, size=n, dtype=np.uint64)
    const = cv.scalars([0x1234567890ABCDEF1122334455667788990011223344556677] * n)
    sets = np.stack([uni, skew, short, const])
    pb = PreparedBases(ctx, pts)
    try:
        out = pb.msm(sets)
        for j in range(sets.shape[0]):
            assert np.array_equal(out[j], gpu_msm(ctx, pts, sets[j])), j
        k = int(sum(int(v) * m for v, m in zip(vals, mult)) % pm.R)
        assert cv._unpt(out[0]) == cv.mul(pm.G, k)
    finally:
        pb.close()


def test_g1_msm_prepared_errors(ctx, orc):
    import ctypes
    cv = Bn254Curve(orc)
    pts = cv.multiples_array(3, 3, 8)
    h = ctypes.c_void_p()
    assert ctx.lib.gl355_bn254_g1_msm_prepare(ctx.h, None, 8, ctypes.byref(h)) == -1
    assert ctx.lib.gl355_bn254_g1_msm_prepare(ctx.h, pts.ctypes.data, 0, ctypes.byref(h)) == -1
    out = np.zeros(8, dtype=np.uint64)
    assert ctx.lib.gl355_bn254_g1_msm_prepared(ctx.h, None, pts.ctypes.data, 1, out.ctypes.data) == -1
    assert ctx.lib.gl355_bn254_g1_msm_bases_free(ctx.h, None) == 0


@pytest.mark.parametrize("log_small,log_n", [(3, 3), (6, 9), (9, 11), (10, 13), (12, 14)])
def test_fr_coset_ntt_vs_oracle(ctx, orc, log_small, log_n):
    """coeff_to_extended / extended_to_coeff of halo2's EvaluationDomain: c_i shift^i zero-padded, transformed -- against the oracle's FFT of
    the shifted coefficients (Python integers for the powers); the inverse form gives the coefficients back"""
    cv = Bn254Curve(orc)
    rng = np.random.default_rng(0x4E8 + log_n)
    ns, n = 1 << log_small, 1 << log_n
    shift = 0x1D4C7A2B5E6F8091A2B3C4D5E6F708192A3B4C5D6E7F8091A2B3C4D5E6F7081 % pm.R
    c = rand_scalars(rng, ns, below_r=False)
    ci = [v % pm.R for v in cv.ints(c)]
    padded = cv.scalars([ci[i] * pow(shift, i, pm.R) % pm.R if i < ns else 0 for i in range(n)])
    want = cv.ntt_array(padded)
    sh = cv.scalars([shift])[0]
    out = np.full((n, 4), 0xAA, dtype=np.uint64)
    ctx.check(ctx.lib.gl355_bn254_fr_coset_ntt(ctx.h, np.ascontiguousarray(c).ctypes.data, log_small, log_n, sh.ctypes.data, 0, out.ctypes.data))
    assert np.array_equal(out, want)
    back = np.full((ns, 4), 0xAA, dtype=np.uint64)
    ctx.check(ctx.lib.gl355_bn254_fr_coset_ntt(ctx.h, out.ctypes.data, log_small, log_n, sh.ctypes.data, 1, back.ctypes.data))
    assert cv.ints(back) == ci
    # the inverse form on arbitrary evaluations: ifft, divide by shift^i, keep the first 2^log_small
    ev = rand_scalars(rng, n)
    inv = cv.ints(cv.ntt_array(ev, inverse=True))
    sinv = pow(shift, pm.R - 2, pm.R)
    ctx.check(ctx.lib.gl355_bn254_fr_coset_ntt(ctx.h, ev.ctypes.data, log_small, log_n, sh.ctypes.data, 1, back.ctypes.data))
    assert cv.ints(back) == [inv[i] * pow(sinv, i, pm.R) % pm.R for i in range(ns)]
    zero = np.zeros(4, dtype=np.uint64)
    assert ctx.lib.gl355_bn254_fr_coset_ntt(ctx.h, ev.ctypes.data, log_small, log_n, zero.ctypes.data, 0, out.ctypes.data) == -1
