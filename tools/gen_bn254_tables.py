#!/usr/bin/env python3
"""BN254-Poseidon (t = 5, R_F = 8, R_P = 60, x^5) parameter tables for the second hash back-end
(reference: src/plonky2_verifier/bn245_poseidon/{constants.rs,native.rs,plonky2_config.rs}).

Input (data, not code): stark-verifier_amd/data/poseidon_bn254_t5.txt -- 340 round constants followed by the 25 MDS
entries (row major), hexadecimal, one per line: the published circomlib Poseidon parameters for t = 5 which the
reference pins at constants.rs:5-379.
Outputs: oracle/bn254_tables.h (canonical form, 4 x u64 little-endian limbs) and
         stark-verifier_amd/csrc/bn254_tables.h (Montgomery form R = 2^256, 8 x u32 little-endian limbs).
`--extract` (re)creates the data file from the reference's literals; `--check-reference` compares (both need /root/reference).
"""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "stark-verifier_amd", "data", "poseidon_bn254_t5.txt")
REF = "/root/reference/src/plonky2_verifier/bn245_poseidon/constants.rs"
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # BN254 scalar field
T, RF, RP = 5, 8, 60


def reference_values():
    txt = open(REF).read()
    vals = [int(h, 16) for h in re.findall(r'"0x([0-9a-fA-F]{64})"', txt)]
    assert len(vals) == T * (RF + RP) + T * T, len(vals)
    return vals


def load():
    vals = [int(l, 16) for l in open(DATA).read().split("\n") if l and not l.startswith("#")]
    assert len(vals) == T * (RF + RP) + T * T
    assert all(v < R for v in vals)
    return vals[:T * (RF + RP)], [vals[T * (RF + RP) + T * i: T * (RF + RP) + T * (i + 1)] for i in range(T)]


# ---- "fast partial rounds": the same factorisation tools/gen_poseidon_tables.py derives for Poseidon-Goldilocks, over Fr ----
def mat_mul(a, b):
    return [[sum(a[i][x] * b[x][j] for x in range(len(b))) % R for j in range(len(b[0]))] for i in range(len(a))]


def mat_vec(a, v):
    return [sum(a[i][j] * v[j] for j in range(len(v))) % R for i in range(len(a))]


def mat_inv(a):
    n = len(a)
    aug = [list(row) + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(a)]
    for c in range(n):
        piv = next(r for r in range(c, n) if aug[r][c] % R)
        aug[c], aug[piv] = aug[piv], aug[c]
        inv = pow(aug[c][c], R - 2, R)
        aug[c] = [x * inv % R for x in aug[c]]
        for r in range(n):
            if r != c and aug[r][c]:
                f = aug[r][c]
                aug[r] = [(x - f * y) % R for x, y in zip(aug[r], aug[c])]
    return [row[n:] for row in aug]


def derive_fast(rc_flat, m):
    """partial rounds as: state += FIRST; state[1..] = INIT^T state[1..]; then per round r:
         s0 = sbox(s0) (+ POST[r] for r < RP-1); d = M00 s0 + sum_i WHAT[r][i-1] s_i; s_i += VS[r][i-1] s0; s0 = d"""
    rc = [rc_flat[T * r: T * (r + 1)] for r in range(RF + RP)]
    half = RF // 2
    minv = mat_inv(m)
    part = rc[half: half + RP]
    acc = list(part[RP - 1])
    post = [0] * RP
    for r in range(RP - 2, -1, -1):
        back = mat_vec(minv, acc)
        post[r] = back[0]
        back[0] = 0
        acc = [(x + y) % R for x, y in zip(part[r], back)]
    first = acc
    a = [row[:] for row in m]
    vs, w_hats, last_d = [None] * RP, [None] * RP, None
    for r in range(RP - 1, -1, -1):
        a_hat = [row[1:] for row in a[1:]]
        a_hat_inv = mat_inv(a_hat)
        vs[r] = [a[i][0] for i in range(1, T)]
        w_hats[r] = mat_mul([a[0][1:]], a_hat_inv)[0]
        d = [[1] + [0] * (T - 1)] + [[0] + a_hat[i] for i in range(T - 1)]
        a = mat_mul(d, m)
        last_d = d
    init = [[last_d[c][r] for c in range(1, T)] for r in range(1, T)]
    return dict(first=first, post=post, vs=vs, w_hats=w_hats, init=init, m00=m[0][0])


def permute_naive(state, rc_flat, m):
    s, k = [x % R for x in state], 0
    for rnd in range(RF + RP):
        s = [(x + rc_flat[k + i]) % R for i, x in enumerate(s)]
        k += T
        if rnd < RF // 2 or rnd >= RF // 2 + RP:
            s = [pow(x, 5, R) for x in s]
        else:
            s[0] = pow(s[0], 5, R)
        s = mat_vec(m, s)
    return s


def permute_fast(state, rc_flat, m, tb):
    half = RF // 2
    s = [x % R for x in state]
    for r in range(half):
        s = mat_vec(m, [pow((x + rc_flat[T * r + i]) % R, 5, R) for i, x in enumerate(s)])
    s = [(x + c) % R for x, c in zip(s, tb["first"])]
    s = [s[0]] + [sum(tb["init"][r - 1][c - 1] * s[r] for r in range(1, T)) % R for c in range(1, T)]
    for r in range(RP):
        s[0] = pow(s[0], 5, R)
        if r < RP - 1:
            s[0] = (s[0] + tb["post"][r]) % R
        d = (s[0] * tb["m00"] + sum(tb["w_hats"][r][i - 1] * s[i] for i in range(1, T))) % R
        s = [d] + [(s[i] + tb["vs"][r][i - 1] * s[0]) % R for i in range(1, T)]
    for r in range(half + RP, RF + RP):
        s = mat_vec(m, [pow((x + rc_flat[T * r + i]) % R, 5, R) for i, x in enumerate(s)])
    return s


def limbs(v, bits, n):
    return [(v >> (bits * i)) & ((1 << bits) - 1) for i in range(n)]


def emit(path, qual, name_prefix, vals_rc, mds, bits, n, mont, fast=None):
    conv = (lambda v: v * (1 << 256) % R) if mont else (lambda v: v)
    ctype = "uint32_t" if bits == 32 else "uint64_t"
    fmt = "0x%08xu" if bits == 32 else "UINT64_C(0x%016x)"
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_bn254_tables.py -- do not edit.\n")
        f.write("// BN254-Poseidon t = 5, R_F = 8, R_P = 60, S-box x^5; %s form, %d x %d-bit little-endian limbs.\n" % (
            "Montgomery (R = 2^256)" if mont else "canonical", n, bits))
        f.write("#pragma once\n#include <stdint.h>\n")
        f.write("%s %s %s_RC[%d][%d] = {\n" % (qual, ctype, name_prefix, len(vals_rc), n))
        for v in vals_rc:
            f.write("  {" + ", ".join(fmt % x for x in limbs(conv(v), bits, n)) + "},\n")
        f.write("};\n%s %s %s_MDS[%d][%d] = {\n" % (qual, ctype, name_prefix, T * T, n))
        for row in mds:
            for v in row:
                f.write("  {" + ", ".join(fmt % x for x in limbs(conv(v), bits, n)) + "},\n")
        f.write("};\n")
        # Montgomery parameters: modulus, -r^-1 mod 2^bits, R^2 mod r (to enter the domain), p_g powers
        word = 1 << bits
        n0 = (-pow(R, -1, word)) % word
        f.write("%s %s %s_MOD[%d] = {%s};\n" % (qual, ctype, name_prefix, n, ", ".join(fmt % x for x in limbs(R, bits, n))))
        f.write("%s %s %s_N0INV = %s;\n" % (qual, ctype, name_prefix, fmt % n0))
        f.write("%s %s %s_R2[%d] = {%s};\n" % (qual, ctype, name_prefix, n, ", ".join(fmt % x for x in limbs(pow(1 << 256, 2, R), bits, n))))
        f.write("%s %s %s_ONE_MONT[%d] = {%s};\n" % (qual, ctype, name_prefix, n, ", ".join(fmt % x for x in limbs((1 << 256) % R, bits, n))))
        pg = (1 << 64) - (1 << 32) + 1
        f.write("// p_g^-1 mod 2^256 (exact division by the Goldilocks prime when splitting an Fr into base-p_g digits)\n")
        f.write("%s %s %s_PG_INV[%d] = {%s};\n" % (qual, ctype, name_prefix, n, ", ".join(fmt % x for x in limbs(pow(pg, -1, 1 << 256), bits, n))))
        f.write("%s %s %s_TWO_MOD[%d] = {%s};\n" % (qual, ctype, name_prefix, n, ", ".join(fmt % x for x in limbs(2 * R, bits, n))))
        if fast is not None:
            f.write("// fast partial rounds (derived, tools/gen_bn254_tables.py derive_fast): FIRST[5], INIT[4][4] (row r-1, col c-1:\n"
                    "// out[c] += INIT * in[r]), POST[60] (entry 59 unused = 0), VS[60][4], WHAT[60][4], M00\n")
            def arr(name, vals):
                f.write("%s %s %s_%s[%d][%d] = {\n" % (qual, ctype, name_prefix, name, len(vals), n))
                for v in vals:
                    f.write("  {" + ", ".join(fmt % x for x in limbs(conv(v), bits, n)) + "},\n")
                f.write("};\n")
            arr("PART_FIRST", fast["first"])
            arr("PART_INIT", [x for row in fast["init"] for x in row])
            arr("PART_POST", fast["post"])
            arr("PART_VS", [x for row in fast["vs"] for x in row])
            arr("PART_WHAT", [x for row in fast["w_hats"] for x in row])
            arr("PART_M00", [fast["m00"]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--extract", action="store_true")
    ap.add_argument("--check-reference", action="store_true")
    args = ap.parse_args()
    if args.extract:
        vals = reference_values()
        with open(DATA, "w") as f:
            f.write("# BN254-Poseidon parameters, t = 5 (circomlib): 340 round constants, then the 5x5 MDS matrix row by row\n")
            for v in vals:
                f.write("%064x\n" % v)
    rc, mds = load()
    if args.check_reference:
        ref = reference_values()
        assert ref[:len(rc)] == rc and ref[len(rc):] == [v for row in mds for v in row]
        print("data file equals the reference's literals (340 + 25 values)")
    fast = derive_fast(rc, mds)
    import random
    rnd = random.Random(5)
    for _ in range(4):
        st = [rnd.randrange(R) for _ in range(T)]
        assert permute_fast(st, rc, mds, fast) == permute_naive(st, rc, mds), "fast partial rounds disagree with the definition"
    emit(os.path.join(ROOT, "oracle", "bn254_tables.h"), "static const", "ORC_BN254", rc, mds, 64, 4, False)     # the oracle stays naive
    emit(os.path.join(ROOT, "stark-verifier_amd", "csrc", "bn254_tables.h"), "BN254_TABLE_QUAL", "BN254", rc, mds, 32, 8, True, fast)
    print("tables written")


if __name__ == "__main__":
    main()
