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
    emit(rc, tb)
    print("tables written; fast == naive on 8 random states; permute(0)[0] = %016x" % kat0[0])
    if args.check_reference:
        ok = check_reference(tb)
        print("derived tables == reference literal tables:", ok)
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
