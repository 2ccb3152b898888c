"""CPU: the oracle against the committed golden vectors and against the independent big-integer
model (tests/pymodel.py).  This is what pins the oracle before it is trusted as the GPU checker."""
import json
import os
import random

import numpy as np
import pytest

import pymodel as pm
from oracle_lib import P, rand_field

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    return json.load(open(os.path.join(HERE, "golden", name)))


def ux(lst):
    return np.array([int(x, 16) for x in lst], dtype=np.uint64)


def test_poseidon_upstream_kats(orc):
    kat = load("poseidon_kat.json")
    for v in kat["permute"]:
        assert np.array_equal(orc.permute(ux(v["input"])), ux(v["output"])), v["name"]
    for v in kat["hash_no_pad"]:
        assert np.array_equal(orc.hash_no_pad(ux(v["input"])), ux(v["output"]))
    # two_to_one == hash_no_pad of the concatenation
    assert np.array_equal(orc.two_to_one(ux(["1", "2", "3", "4"]), ux(["5", "6", "7", "8"])), ux(kat["hash_no_pad"][2]["output"]))


def test_convention_vectors(orc):
    c = load("conventions.json")
    assert orc.root_of_unity(3) == int(c["omega_8"], 16)
    assert orc.root_of_unity(16) == int(c["omega_2_16"], 16)
    assert orc.root_of_unity(32) == int(c["omega_2_32"], 16) == 1753635133440165772
    for v in c["ntt"]:
        x = ux(v["input"])
        assert np.array_equal(orc.ntt(x), ux(v["forward"]))
        assert np.array_equal(orc.ntt(x, inverse=True), ux(v["inverse"]))
    for v in c["lde"]:
        assert np.array_equal(orc.lde(ux(v["coeffs"]), v["rate_bits"], v["shift"]), ux(v["natural"]))
    for v in c["merkle"]:
        leaves = np.array([[int(x, 16) for x in l] for l in v["leaves"]], dtype=np.uint64)
        for variant in ("", "_recursive", "_layered"):
            dig, cap = orc.merkle_build(leaves, v["cap_height"], variant)
            assert np.array_equal(cap, np.array([[int(x, 16) for x in d] for d in v["cap"]], dtype=np.uint64))
            want = np.array([[int(x, 16) for x in d] for d in v["digests"]], dtype=np.uint64).reshape(-1, 4)
            assert np.array_equal(dig, want), variant
    e = c["ext_mul"]
    assert np.array_equal(orc.ext_mul(ux(e["a"]), ux(e["b"])), ux(e["out"]))


def test_field_against_bigint(orc):
    rnd = random.Random(1)
    edge = [0, 1, P - 1, P, P + 1, (1 << 64) - 1, 1 << 32, (1 << 32) - 1, 0xFFFFFFFF00000000]
    vals = edge + [rnd.randrange(1 << 64) for _ in range(300)]
    for a in vals:
        for b in vals[:40]:
            assert orc.add(a, b) == (a + b) % P
            assert orc.sub(a, b) == (a - b) % P
            assert orc.mul(a, b) == (a * b) % P == orc.mul_ref(a, b)
    for a in vals:
        if a % P:
            assert orc.inv(a) == pow(a, P - 2, P)
    for _ in range(50):
        a, b = (rnd.randrange(P), rnd.randrange(P)), (rnd.randrange(P), rnd.randrange(P))
        assert tuple(int(x) for x in orc.ext_mul(a, b)) == pm.ext_mul(a, b)
        assert tuple(int(x) for x in orc.ext_inv(a)) == pm.ext_inv(a)


def test_ntt_lde_definition_random(orc):
    rnd = random.Random(2)
    for lg in range(0, 8):
        c = [rnd.randrange(P) for _ in range(1 << lg)]
        x = np.array(c, dtype=np.uint64)
        assert [int(v) for v in orc.ntt(x)] == pm.dft(c)
        assert [int(v) for v in orc.ntt(x, inverse=True)] == pm.dft(c, inverse=True)
        if lg:
            sh = rnd.randrange(1, P)
            cs = [a * pow(sh, i, P) % P for i, a in enumerate(c)]
            assert [int(v) for v in orc.ntt(x, shift=sh)] == pm.dft(cs)
            back = orc.ntt(orc.ntt(x, shift=sh), inverse=True, shift=sh)
            assert np.array_equal(back, x)
    for lg, rb in ((1, 1), (3, 2), (4, 3), (6, 3)):
        c = [rnd.randrange(P) for _ in range(1 << lg)]
        assert [int(v) for v in orc.lde(np.array(c, dtype=np.uint64), rb)] == pm.lde(c, rb)


def test_ntt_large_roundtrip_and_linearity(orc):
    rng = np.random.default_rng(3)
    a, b = rand_field(rng, 1 << 16), rand_field(rng, 1 << 16)
    assert np.array_equal(orc.ntt(orc.ntt(a), inverse=True), a)
    s = np.array([orc.add(int(x), int(y)) for x, y in zip(a[:4096], b[:4096])], dtype=np.uint64)
    fa, fb, fs = orc.ntt(a[:4096]), orc.ntt(b[:4096]), orc.ntt(s)
    assert all(orc.add(int(x), int(y)) == int(z) for x, y, z in zip(fa, fb, fs))


def test_transpose_bitrev(orc):
    rng = np.random.default_rng(4)
    m = rng.integers(0, 1 << 62, (16, 5), dtype=np.uint64)
    assert np.array_equal(orc.transpose(m), m.T)
    r = orc.reverse_index_bits(m)
    for i in range(16):
        assert np.array_equal(r[i], m[pm.bitrev(i, 4)])


def test_merkle_prove_verify_and_layouts(orc):
    rng = np.random.default_rng(5)
    for lg, ll, cap in ((5, 4, 0), (6, 11, 2), (13, 4, 4), (13, 9, 0)):
        n = 1 << lg
        leaves = rand_field(rng, (n, ll))
        d1, c1 = orc.merkle_build(leaves, cap, "_recursive")
        d2, c2 = orc.merkle_build(leaves, cap, "_layered")
        assert np.array_equal(d1, d2) and np.array_equal(c1, c2)
        for idx in (0, 1, n - 1, n // 3):
            sib = orc.merkle_prove(d1, n, cap, idx)
            assert orc.merkle_verify(leaves[idx], idx, sib, c1, cap)
            bad = leaves[idx].copy()
            bad[0] ^= np.uint64(1)
            assert not orc.merkle_verify(bad, idx, sib, c1, cap)
            # model: fold the path by hand (merkle_proof_chip.rs:58-70)
            st = pm.hash_or_noop([int(x) for x in leaves[idx]])
            k = idx
            for s in sib:
                s = [int(x) for x in s]
                st = pm.two_to_one(s, st) if k & 1 else pm.two_to_one(st, s)
                k >>= 1
            assert st == [int(x) for x in c1[k]]


def test_commit_matches_composition(orc):
    """orc_commit == iNTT, LDE, salt, transpose, bit-reverse, Merkle composed by hand."""
    rng = np.random.default_rng(6)
    vals = rand_field(rng, (5, 1 << 4))
    salt = rand_field(rng, (4, 1 << 7))
    coeffs, leaves, dig, cap = orc.commit(vals, 3, 2, salt=salt)
    assert np.array_equal(coeffs, orc.ntt(vals, inverse=True))
    cols = np.concatenate([orc.lde(coeffs, 3), salt])
    want = orc.reverse_index_bits(np.ascontiguousarray(cols.T))
    assert np.array_equal(leaves, want)
    d2, c2 = orc.merkle_build(want, 2)
    assert np.array_equal(dig, d2) and np.array_equal(cap, c2)
    # leaf i = evaluations at 7 * w^bitrev(i)  (fri_chip.rs:245-264)
    w = pm.root_of_unity(7)
    for i in (0, 1, 77):
        x = 7 * pow(w, pm.bitrev(i, 7), P) % P
        assert [int(v) for v in leaves[i][:5]] == [pm.poly_eval([int(a) for a in coeffs[c]], x) for c in range(5)]


def test_deep_quotient_identity(orc):
    """acc' = acc*alpha^k + (C - C(z))/(X - z): check (X - z) * Q + C(z) == C coefficient-wise."""
    rnd = random.Random(7)
    rng = np.random.default_rng(7)
    n, k = 16, 5
    polys = rand_field(rng, (k, n))
    alpha, z = (rnd.randrange(P), rnd.randrange(P)), (rnd.randrange(P), rnd.randrange(P))
    acc0 = rand_field(rng, 2 * n)
    got = orc.deep_batch(polys, alpha, z, acc0).reshape(n, 2)
    comp = [(0, 0)] * n
    ap = (1, 0)
    for i in range(k):
        comp = [pm.ext_add(c, pm.ext_mul(ap, (int(polys[i][j]), 0))) for j, c in enumerate(comp)]
        ap = pm.ext_mul(ap, alpha)
    cz = pm.ext_poly_eval(comp, z)
    q = [pm.ext_sub((int(g[0]), int(g[1])), pm.ext_mul((int(acc0[2 * j]), int(acc0[2 * j + 1])), ap)) for j, g in enumerate(got)]
    assert q[n - 1] == (0, 0)
    # (X - z) * Q + C(z) == C
    for j in range(n):
        lhs = pm.ext_sub(q[j - 1] if j else (0, 0), pm.ext_mul(z, q[j]))
        if j == 0:
            lhs = pm.ext_add(lhs, cz)
        assert lhs == comp[j]
    ev = orc.eval_polys_ext(polys, z)
    for i in range(k):
        assert (int(ev[i][0]), int(ev[i][1])) == pm.ext_poly_eval([int(x) for x in polys[i]], z)


def test_fri_fold_matches_verifier_formula(orc):
    """fold in coefficient form == the verifier's interpolation (fri_chip.rs:168-226)."""
    rnd = random.Random(8)
    n = 16
    c = [(rnd.randrange(P), rnd.randrange(P)) for _ in range(n)]
    beta = (rnd.randrange(P), rnd.randrange(P))
    flat = np.array([v for pair in c for v in pair], dtype=np.uint64)
    folded = orc.fri_fold(flat, beta).reshape(-1, 2)
    fc = [(int(a), int(b)) for a, b in folded]
    w = pm.root_of_unity(4)
    for j in (0, 3, 5):
        x = 7 * pow(w, j, P) % P
        a0, b0 = (x, 0), ((P - x) % P, 0)
        a1, b1 = pm.ext_poly_eval(c, a0), pm.ext_poly_eval(c, b0)
        num = pm.ext_mul(pm.ext_sub(beta, a0), pm.ext_sub(b1, a1))
        want = pm.ext_add(a1, pm.ext_mul(num, pm.ext_inv(pm.ext_sub(b0, a0))))
        assert pm.ext_poly_eval(fc, (x * x % P, 0)) == want
    # layer leaves: pairs (v[br(2i)], v[br(2i+1)]) are the evaluations at x and -x
    vals = rand_field(np.random.default_rng(8), 2 * n)
    lv = orc.fri_layer_leaves(vals)
    for i in range(n // 2):
        assert np.array_equal(lv[i][:2], vals[2 * pm.bitrev(2 * i, 4):][:2])
        assert np.array_equal(lv[i][2:], vals[2 * pm.bitrev(2 * i + 1, 4):][:2])
        assert pm.bitrev(2 * i + 1, 4) == pm.bitrev(2 * i, 4) + n // 2


def test_pow_and_challenger(orc):
    rng = np.random.default_rng(9)
    st = rand_field(rng, 12)
    w = orc.pow_grind(st, 3, 8)
    for cand in range(w + 1):
        s2 = [int(x) for x in st]
        s2[3] = cand
        ok = pm.permute(s2)[7] >> 56 == 0
        assert ok == (cand == w)
    # challenger: observe < 8, squeeze pops state[7], state[6], ... (hasher_chip.rs:73-89)
    ch = orc.challenger()
    orc.observe(ch, [1, 2, 3])
    st = pm.permute([1, 2, 3] + [0] * 9)
    assert [orc.squeeze(ch) for _ in range(3)] == [st[7], st[6], st[5]]
    orc.observe(ch, [9])                     # new input invalidates the buffer, overwrites state[0]
    st2 = pm.permute([9] + st[1:])
    assert orc.squeeze(ch) == st2[7]
    ch = orc.challenger()
    orc.observe(ch, list(range(1, 11)))      # 8 absorbed eagerly, 2 pending
    s1 = pm.permute(list(range(1, 9)) + [0] * 4)
    s2 = pm.permute([9, 10] + s1[2:])
    assert orc.squeeze(ch) == s2[7]


def test_zs_partial_products_definition(orc):
    rnd = random.Random(10)
    rng = np.random.default_rng(10)
    lg, nr, md = 3, 6, 2
    n = 1 << lg
    wires, sig = rand_field(rng, (nr, n)), rand_field(rng, (nr, n))
    k_is = rand_field(rng, nr)
    beta, gamma = rnd.randrange(P), rnd.randrange(P)
    z, pp = orc.zs_partial_products(wires, sig, k_is, md, beta, gamma)
    g = pm.root_of_unity(lg)
    zz = 1
    for i in range(n):
        assert int(z[i]) == zz
        x = pow(g, i, P)
        acc = zz
        for ch in range(nr // md):
            for j in range(ch * md, (ch + 1) * md):
                num = (int(wires[j][i]) + beta * int(k_is[j]) * x + gamma) % P
                den = (int(wires[j][i]) + beta * int(sig[j][i]) + gamma) % P
                acc = acc * num * pow(den, P - 2, P) % P
            if ch + 1 < nr // md:
                assert int(pp[ch][i]) == acc
        zz = acc
