#!/usr/bin/env python3
"""Derive the Poseidon-Goldilocks constant tables the HIP kernels and the oracle compile in.

Inputs (data, not code):
  * stark-verifier_amd/data/poseidon_goldilocks_round_constants.txt -- the 360 published round
    constants (reference pins them at src/plonky2_verifier/chip/plonk/gates/poseidon.rs:26-124)
  * the circulant MDS first row + diagonal (poseidon.rs:321-322)

Outputs:
  * oracle/poseidon_rc.h                         round constants + MDS only (the oracle runs the
                                                 NAIVE permutation, so it never sees derived tables)
  * stark-verifier_amd/csrc/poseidon_tables.h    round constants + the "fast partial round" tables
                                                 (first-round constant vector, per-round scalar
                                                 constants, pre-matrix, sparse v / w_hat vectors)

The fast tables are DERIVED here by linear algebra over F_p (factor every partial-round MDS as
sparse * blockdiag(1, M_hat) and push the block-diagonal factor and all but lane 0 of each constant
vector backwards through the partial S-box, which only touches lane 0).  The semantics they must
satisfy are the ones the reference's Poseidon gate evaluates (poseidon.rs:504-589, 652-673):
    state += FIRST; state = blockdiag(1, INIT^T) state;
    for r: state[0] = sbox(state[0]); if r < 21: state[0] += RC[r];
           d = state[0]*M00 + sum_i W_HAT[r][i-1]*state[i]; state[i] += VS[r][i-1]*state[0]; state[0] = d
`--check-reference` (only usable where /root/reference exists) compares every derived table with
the reference's literal tables.
"""
import argparse
import os
import re
import sys

P = (1 << 64) - (1 << 32) + 1
T = 12
HALF_F = 4
R_P = 22
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
DIAG = [8] + [0] * 11

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_rc():
    path = os.path.join(ROOT, "stark-verifier_amd", "data", "poseidon_goldilocks_round_constants.txt")
    vals = [int(l, 16) for l in open(path).read().split("\n") if l and not l.startswith("#")]
    assert len(vals) == T * (2 * HALF_F + R_P)
    return [vals[T * r: T * (r + 1)] for r in range(2 * HALF_F + R_P)]


def mds_matrix():
    # new[r] = sum_i old[(i + r) % 12] * CIRC[i] + old[r] * DIAG[r]   (poseidon.rs:450-486)
    m = [[0] * T for _ in range(T)]
    for r in range(T):
        for i in range(T):
            m[r][(i + r) % T] = (m[r][(i + r) % T] + CIRC[i]) % P
        m[r][r] = (m[r][r] + DIAG[r]) % P
    return m


def mat_mul(a, b):
    n, k, m = len(a), len(b), len(b[0])
    return [[sum(a[i][x] * b[x][j] for x in range(k)) % P for j in range(m)] for i in range(n)]


def mat_vec(a, v):
    return [sum(a[i][j] * v[j] for j in range(len(v))) % P for i in range(len(a))]


def mat_inv(a):
    n = len(a)
    aug = [list(row) + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(a)]
    for c in range(n):
        piv = next(r for r in range(c, n) if aug[r][c] % P)
        aug[c], aug[piv] = aug[piv], aug[c]
        inv = pow(aug[c][c], P - 2, P)
        aug[c] = [x * inv % P for x in aug[c]]
        for r in range(n):
            if r != c and aug[r][c]:
                f = aug[r][c]
                aug[r] = [(x - f * y) % P for x, y in zip(aug[r], aug[c])]
    return [row[n:] for row in aug]


def derive(rc):
    m = mds_matrix()
    minv = mat_inv(m)
    part = rc[HALF_F: HALF_F + R_P]
    # --- constants: push each round's vector back through the previous round's MDS -----------
    acc = list(part[R_P - 1])
    post = [0] * R_P
    for r in range(R_P - 2, -1, -1):
        back = mat_vec(minv, acc)
        post[r] = back[0]
        back[0] = 0
        acc = [(x + y) % P for x, y in zip(part[r], back)]
    first = acc
    # --- matrices: A = S * blockdiag(1, A_hat); the block-diagonal factor commutes with the
    #     lane-0 S-box and is absorbed into the previous round's MDS ------------------------------
    a = [row[:] for row in m]
    vs = [None] * R_P
    w_hats = [None] * R_P
    for r in range(R_P - 1, -1, -1):
        a_hat = [row[1:] for row in a[1:]]
        a_hat_inv = mat_inv(a_hat)
        vs[r] = [a[i][0] for i in range(1, T)]
        row = [a[0][1:]]
        w_hats[r] = mat_mul(row, a_hat_inv)[0]
        d = [[1] + [0] * (T - 1)] + [[0] + a_hat[i] for i in range(T - 1)]
        a = mat_mul(d, m)
        last_d = d
    # plonky2 stores the pre-matrix transposed: result[c] += INIT[r-1][c-1] * state[r]
    init = [[last_d[c][r] for c in range(1, T)] for r in range(1, T)]
    return dict(first=first, post=post, vs=vs, w_hats=w_hats, init=init, m00=m[0][0])


# ----------------------------------------------------------------------------------------------
def sbox(x):
    return pow(x, 7, P)


def permute_naive(state, rc):
    m = mds_matrix()
    s = [x % P for x in state]
    for r in range(2 * HALF_F + R_P):
        s = [(x + c) % P for x, c in zip(s, rc[r])]
        if r < HALF_F or r >= HALF_F + R_P:
            s = [sbox(x) for x in s]
        else:
            s[0] = sbox(s[0])
        s = mat_vec(m, s)
    return s


def permute_fast(state, rc, tb):
    m = mds_matrix()
    s = [x % P for x in state]
    for r in range(HALF_F):
        s = mat_vec(m, [sbox((x + c) % P) for x, c in zip(s, rc[r])])
    s = [(x + c) % P for x, c in zip(s, tb["first"])]
    s = [s[0]] + [sum(tb["init"][r - 1][c - 1] * s[r] for r in range(1, T)) % P for c in range(1, T)]
    for r in range(R_P):
        s[0] = sbox(s[0])
        if r < R_P - 1:
            s[0] = (s[0] + tb["post"][r]) % P
        d = (s[0] * tb["m00"] + sum(tb["w_hats"][r][i - 1] * s[i] for i in range(1, T))) % P
        s = [d] + [(s[i] + tb["vs"][r][i - 1] * s[0]) % P for i in range(1, T)]
    for r in range(HALF_F + R_P, 2 * HALF_F + R_P):
        s = mat_vec(m, [sbox((x + c) % P) for x, c in zip(s, rc[r])])
    return s


def c_array(name, vals, per_line=3, ctype="uint64_t", qual="static const"):
    out = ["%s %s %s[%d] = {" % (qual, ctype, name, len(vals))]
    for i in range(0, len(vals), per_line):
        out.append("  " + " ".join("UINT64_C(0x%016x)," % v for v in vals[i:i + per_line]))
    out.append("};")
    return "\n".join(out)


def emit(rc, tb):
    flat_rc = [x for row in rc for x in row]
    hdr = ("// GENERATED by tools/gen_poseidon_tables.py -- do not edit.\n"
           "// Poseidon over Goldilocks, width 12, S-box x^7, 4 + 22 + 4 rounds.\n")
    oracle = hdr + "#pragma once\n#include <stdint.h>\n" + \
        c_array("ORC_POSEIDON_RC", flat_rc) + "\n" + \
        c_array("ORC_MDS_CIRC", CIRC, 12) + "\n" + c_array("ORC_MDS_DIAG", DIAG, 12) + "\n"
    with open(os.path.join(ROOT, "oracle", "poseidon_rc.h"), "w") as f:
        f.write(oracle)
    full = [x for r in list(range(HALF_F)) + list(range(HALF_F + R_P, 2 * HALF_F + R_P)) for x in rc[r]]
    prod = hdr + ("// Layout: PSD_ALL_RC[31][12] (all rounds, naive form, then a zero row), PSD_FULL_RC[8][12] (first 4 = opening full rounds, last 4 = closing),\n"
                  "// PSD_PART_FIRST[12], PSD_PART_INIT[11][11] (row r-1, col c-1: out[c] += M*in[r]),\n"
                  "// PSD_PART_RC[22] (entry 21 unused = 0), PSD_PART_VS[22][11], PSD_PART_WHAT[22][11].\n"
                  "#pragma once\n#include <stdint.h>\n#ifndef PSD_TABLE_QUAL\n#define PSD_TABLE_QUAL static const\n#endif\n")
    q = "PSD_TABLE_QUAL"
    prod += c_array("PSD_FULL_RC", full, qual=q) + "\n"
    prod += c_array("PSD_ALL_RC", flat_rc + [0] * 12, qual=q) + "\n"       # row 30 = zeros: "the constants of the round after the last"
    prod += c_array("PSD_PART_FIRST", tb["first"], qual=q) + "\n"
    prod += c_array("PSD_PART_INIT", [x for row in tb["init"] for x in row], qual=q) + "\n"
    prod += c_array("PSD_PART_RC", tb["post"], qual=q) + "\n"
    prod += c_array("PSD_PART_VS", [x for row in tb["vs"] for x in row], qual=q) + "\n"
    prod += c_array("PSD_PART_WHAT", [x for row in tb["w_hats"] for x in row], qual=q) + "\n"
    with open(os.path.join(ROOT, "stark-verifier_amd", "csrc", "poseidon_tables.h"), "w") as f:
        f.write(prod)


def check_reference(tb):
    path = "/root/reference/src/plonky2_verifier/chip/plonk/gates/poseidon.rs"
    src = open(path).read()

    def grab(name):
        m = re.search(r"const %s.*?=\s*\[(.*?)\];" % name, src, re.S)
        return [int(x, 16) for x in re.findall(r"0x[0-9a-f]+", m.group(1))]
    ok = True
    ok &= grab("FAST_PARTIAL_FIRST_ROUND_CONSTANT") == tb["first"]
    ok &= grab("FAST_PARTIAL_ROUND_CONSTANTS") == tb["post"]
    ok &= grab("FAST_PARTIAL_ROUND_VS") == [x for row in tb["vs"] for x in row]
    ok &= grab("FAST_PARTIAL_ROUND_W_HATS") == [x for row in tb["w_hats"] for x in row]
    ok &= grab("FAST_PARTIAL_ROUND_INITIAL_MATRIX") == [x for row in tb["init"] for x in row]
    return ok


# ----------------------------------------------------------------------------------------------
# The "block" form of the partial rounds (round 5; stark-verifier_amd/csrc/poseidon.cuh, psd_partial_rounds_block).
# Only lane 0 meets the S-box in a partial round, so over a block of B rounds the other eleven lanes are a LINEAR function of the state
# u that entered the block and of the S-box outputs x_0 .. x_{B-1}.  With A = M * diag(0,1,..,1) and m0 = M e0 (a round is
# s <- A s + x m0 + C):
#     y_j  = <alpha_j, u> + sum_{i<j} x_i kappa_{j-1-i} + gamma_j        alpha_j = e0^T A^j,  kappa_k = e0^T A^k m0      (j = 1 .. B-1; y_0 = u_0)
#     s'_r = <(A^B)_r, u> + sum_{i<B} x_i beta_{B-1-i}[r] + Gamma_r       beta_k = A^k m0                                   (r = 0 .. 11)
# alpha_j, kappa, beta, A^B do not depend on the block; gamma / Gamma collect the round constants of the block's rounds (the last block's Gamma
# includes the constants of the first closing full round, as the dense form's "MDS adds the next round's constants" did).  Every
# output is ONE dot product with constants, reduced once -- 429 multiply-accumulates and 22 reductions per block of 11 rounds against
# 11 x 144 small-constant multiply-accumulates and 11 x 12 reductions of the dense form.
# A multiply-accumulate x * c: x is split once into limbs of 22 / 22 / 20 bits and multiplied with the 32-bit halves of c, 2^22 c, 2^44 c
# (mod p): six 32 x 32 + 64 multiply-adds into two accumulators (low halves, high halves) that cannot overflow (66 products < 2^54).
K_BLOCK = 11


def kform_tables(rc):
    m = mds_matrix()
    a = [[0 if j == 0 else m[i][j] for j in range(T)] for i in range(T)]          # A = M P
    m0 = [m[i][0] for i in range(T)]
    B = K_BLOCK
    apow = [[[1 if i == j else 0 for j in range(T)] for i in range(T)]]
    for _ in range(B):
        apow.append(mat_mul(a, apow[-1]))
    alpha = [apow[j][0] for j in range(B)]                                        # row 0 of A^j
    beta = [mat_vec(apow[k], m0) for k in range(B)]
    kappa = [beta[k][0] for k in range(B)]
    mac = []                                                                      # consumption order
    for j in range(1, B):
        assert alpha[j][0] == 0
        mac += alpha[j][1:]
        mac += [kappa[j - 1 - i] for i in range(j)]
    for r in range(T):
        assert apow[B][r][0] == 0
        mac += apow[B][r][1:]
        mac += [beta[B - 1 - i][r] for i in range(B)]
    assert len(mac) == sum(11 + j for j in range(1, B)) + T * (11 + B)
    add = []
    for blk in range(R_P // B):
        q0 = blk * B
        cs = [rc[HALF_F + q0 + i + 1] if HALF_F + q0 + i + 1 < 2 * HALF_F + R_P else [0] * T for i in range(B)]       # C_{q0+i+1}
        for j in range(1, B):
            add.append(sum(mat_vec(apow[j - 1 - i], cs[i])[0] for i in range(j)) % P)
        gam = [0] * T
        for i in range(B):
            gam = [(x + y) % P for x, y in zip(gam, mat_vec(apow[B - 1 - i], cs[i]))]
        add += gam
    return dict(mac=mac, add=add, B=B)


def k_split(x):
    return [x & 0x3FFFFF, (x >> 22) & 0x3FFFFF, x >> 44]


def k_triple(c):
    return [c, (c << 22) % P, (c << 44) % P]


def k_recombine(al, ah):
    """al + ah 2^32 for al, ah < 2^61 -> some u64 congruent to it (psd_recombine)"""
    assert al < (1 << 61) and ah < (1 << 61)
    mid = (al >> 32) + (ah & 0xFFFFFFFF)
    top = (ah >> 32) + (mid >> 32)
    x = (al & 0xFFFFFFFF) | ((mid & 0xFFFFFFFF) << 32)
    r = x + top * 0xFFFFFFFF
    if r >> 64:
        r = (r & ((1 << 64) - 1)) + 0xFFFFFFFF
        assert r < (1 << 64)
    return r


def k_mul64(a, b):
    """the kernels' product of two arbitrary u64: a u64 congruent to a b, not necessarily canonical (modelled as canonical + p when that fits)"""
    r = a * b % P
    return r + P if (a ^ b) & 1 and r + P < (1 << 64) else r


def permute_kform(state, rc, kt, noncanonical=False):
    """the device algorithm, limb for limb: dense full rounds, block-form partial rounds"""
    m = mds_matrix()
    mul = k_mul64 if noncanonical else (lambda a, b: a * b % P)

    def sbox64(x):
        x2 = mul(x, x); x4 = mul(x2, x2); x3 = mul(x, x2)
        return mul(x3, x4)
    s = [(x + c) % P for x, c in zip(state, rc[0])]
    for r in range(HALF_F):
        s = [(v + c) % P for v, c in zip(mat_vec(m, [sbox64(x) for x in s]), rc[r + 1])]
    if noncanonical:
        s = [v + P if v + P < (1 << 64) and v & 1 else v for v in s]
    B = kt["B"]
    for blk in range(R_P // B):
        pos = 0
        addc = kt["add"][blk * (B - 1 + T): (blk + 1) * (B - 1 + T)]
        ul = [k_split(v) for v in s[1:]]
        xl = []
        y = s[0]
        out = []
        for j in range(B + T):
            if j < B:
                if j > 0:
                    c = addc[j - 1]
                else:
                    xl.append(k_split(sbox64(y)))
                    continue
                terms = ul + xl[:j]
            else:
                c = addc[B - 1 + (j - B)]
                terms = ul + xl
            lo, hi = c & 0xFFFFFFFF, c >> 32
            for lim in terms:
                tr = k_triple(kt["mac"][pos]); pos += 1
                for k in range(3):
                    lo += lim[k] * (tr[k] & 0xFFFFFFFF)
                    hi += lim[k] * (tr[k] >> 32)
            v = k_recombine(lo, hi)
            if j < B:
                xl.append(k_split(sbox64(v)))
            else:
                out.append(v)
        assert pos == len(kt["mac"])
        s = out
    s = [v % P for v in s]
    for r in range(HALF_F + R_P, 2 * HALF_F + R_P):
        nxt = rc[r + 1] if r + 1 < 2 * HALF_F + R_P else [0] * T
        s = [(v + c) % P for v, c in zip(mat_vec(m, [sbox64(x) % P for x in s]), nxt)]
    return s


def emit_kform(kt):
    hdr = ("// GENERATED by tools/gen_poseidon_tables.py -- do not edit.\n"
           "// Block form of Poseidon-Goldilocks' 22 partial rounds (two blocks of %d): see the generator for the algebra.\n"
           "// PSD_K_MAC[%d][3]: per multiply-accumulate, in consumption order, (c, 2^22 c, 2^44 c) mod p; per block 10 lane-0 inputs y_1..y_10\n"
           "//   (11 state terms + j S-box terms each) then the 12 lanes of the outgoing state (11 + 11 terms each).\n"
           "// PSD_K_ADD[2][22][2]: per block the additive constants of y_1..y_10 and of the 12 outgoing lanes, as (low 32 bits, high 32 bits).\n"
           "#pragma once\n#include <stdint.h>\n#ifndef PSD_TABLE_QUAL\n#define PSD_TABLE_QUAL static const\n#endif\n") % (kt["B"], len(kt["mac"]))
    q = "PSD_TABLE_QUAL"
    macs = [v for c in kt["mac"] for v in k_triple(c)]
    adds = [v for c in kt["add"] for v in (c & 0xFFFFFFFF, c >> 32)]
    body = "#define PSD_K_BLOCK %d\n#define PSD_K_MACS %d\n" % (kt["B"], len(kt["mac"]))
    body += c_array("PSD_K_MAC", macs, qual=q) + "\n" + c_array("PSD_K_ADD", adds, 4, qual=q) + "\n"
    with open(os.path.join(ROOT, "stark-verifier_amd", "csrc", "poseidon_ktables.h"), "w") as f:
        f.write(hdr + body)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-reference", action="store_true")
    args = ap.parse_args()
    rc = load_rc()
    tb = derive(rc)
    import random
    rnd = random.Random(355)
    for _ in range(8):
        st = [rnd.randrange(P) for _ in range(T)]
        assert permute_naive(st, rc) == permute_fast(st, rc, tb), "fast != naive"
    kat0 = permute_naive([0] * 12, rc)
    assert kat0[0] == 0x3c18a9786cb0b359, hex(kat0[0])
    kt = kform_tables(rc)
    edge = [[0] * 12, [P - 1] * 12, [1] + [0] * 11, [(1 << 32) - 1] * 12, [P - (1 << 32)] * 12]
    for st in edge + [[rnd.randrange(P) for _ in range(T)] for _ in range(24)]:
        want = permute_naive(st, rc)
        assert permute_kform(st, rc, kt) == want, "block form != naive"
        assert permute_kform(st, rc, kt, noncanonical=True) == want, "block form != naive on non-canonical intermediates"
    emit(rc, tb)
    emit_kform(kt)
    print("tables written; fast == naive on 8 random states, block form == naive on 29 states (limb model, accumulator bounds asserted); "
          "permute(0)[0] = %016x" % kat0[0])
    if args.check_reference:
        ok = check_reference(tb)
        print("derived tables == reference literal tables:", ok)
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
