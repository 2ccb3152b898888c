/*
 * bn254_oracle.c -- CPU restatement of the reference's BN254-Poseidon hasher over Goldilocks elements
 * (SURVEY 8(f) N1).  TEST INFRASTRUCTURE ONLY (see gl_oracle.h).
 *
 * Follows src/plonky2_verifier/bn245_poseidon/native.rs:16-62 (constant / S-box / MDS layers, t = 5, 8 full + 60
 * partial rounds, x^5), native.rs:64-77 + chip/native_chip/utils.rs:25-36 (three Goldilocks elements <-> one Fr in base
 * p_g) and plonky2_config.rs:38-75 (Bn254PoseidonPermutation::permute over the 12-element sponge state, hash_no_pad,
 * two_to_one).  PINNED: with the reference's parameters (constants.rs:5-379) the Fr permutation reproduces the published
 * circomlib known answer poseidon([1,2,3,4]) (tests/golden/poseidon_bn254_kat.json).
 */
#include "gl_oracle.h"
#include "gl_inline.h"
#include "bn254_tables.h"

#include <string.h>

typedef struct { uint64_t l[4]; } fr;   /* little-endian limbs; Montgomery form inside the permutation */

static int fr_geq_mod(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > ORC_BN254_MOD[i]) return 1;
        if (a[i] < ORC_BN254_MOD[i]) return 0;
    }
    return 1;
}
static void fr_sub_mod(uint64_t a[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - ORC_BN254_MOD[i] - (uint64_t)br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static fr fr_add(fr a, fr b) {
    fr r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || fr_geq_mod(r.l)) fr_sub_mod(r.l);       /* a, b < r < 2^254: no carry out of 256 bits */
    return r;
}
/* Montgomery product a b R^-1 mod r, R = 2^256 (CIOS) */
static fr fr_mul(fr a, fr b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * ORC_BN254_N0INV;
        c = (u128)m * ORC_BN254_MOD[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * ORC_BN254_MOD[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fr r;
    memcpy(r.l, t, 32);
    if (t[4] || fr_geq_mod(r.l)) fr_sub_mod(r.l);
    return r;
}
static fr fr_from_limbs(const uint64_t l[4]) { fr r; memcpy(r.l, l, 32); return r; }
static fr fr_to_mont(fr a) { return fr_mul(a, fr_from_limbs(ORC_BN254_R2)); }
static fr fr_from_mont(fr a) { fr one = {{1, 0, 0, 0}}; return fr_mul(a, one); }
static fr fr_pow5(fr a) { fr a2 = fr_mul(a, a), a4 = fr_mul(a2, a2); return fr_mul(a4, a); }

/* the Fr permutation on canonical (non-Montgomery) values: state[5][4] limbs */
void orc_bn254_permute_fr(uint64_t state[5][4]) {
    static fr rc[340], mds[25];
    static volatile int init = 0;
    if (!init) {
#pragma omp critical(orc_bn254_tables)
        if (!init) {
            for (int i = 0; i < 340; i++) rc[i] = fr_to_mont(fr_from_limbs(ORC_BN254_RC[i]));
            for (int i = 0; i < 25; i++) mds[i] = fr_to_mont(fr_from_limbs(ORC_BN254_MDS[i]));
            __sync_synchronize();
            init = 1;
        }
    }
    fr s[5];
    for (int i = 0; i < 5; i++) s[i] = fr_to_mont(fr_from_limbs(state[i]));
    int k = 0;
    for (int rnd = 0; rnd < 68; rnd++) {
        for (int i = 0; i < 5; i++) s[i] = fr_add(s[i], rc[k++]);                  /* native.rs:16-21 */
        if (rnd < 4 || rnd >= 64) { for (int i = 0; i < 5; i++) s[i] = fr_pow5(s[i]); }   /* :23-27 */
        else s[0] = fr_pow5(s[0]);                                                  /* :29-31 */
        fr n[5];
        for (int i = 0; i < 5; i++) {                                               /* :33-41: new[i] = sum_j M[i][j] s[j] */
            fr acc = {{0, 0, 0, 0}};
            for (int j = 0; j < 5; j++) acc = fr_add(acc, fr_mul(s[j], mds[5 * i + j]));
            n[i] = acc;
        }
        memcpy(s, n, sizeof n);
    }
    for (int i = 0; i < 5; i++) { fr c = fr_from_mont(s[i]); memcpy(state[i], c.l, 32); }
}

/* x0 + x1 p + x2 p^2 < 2^192: no reduction needed (native.rs:64-69) */
static void encode3(const uint64_t x[3], uint64_t out[4]) {
    uint64_t a[4] = {canon(x[2]), 0, 0, 0};
    for (int step = 1; step >= 0; step--) {
        u128 c = canon(x[step]);                                 /* a = a * p + x[step] */
        for (int i = 0; i < 4; i++) { c += (u128)a[i] * P; a[i] = (uint64_t)c; c >>= 64; }
    }
    memcpy(out, a, 32);
}
/* the three low base-p digits (native.rs:71-77, utils.rs:25-36) */
static void decode3(const uint64_t x[4], uint64_t out[3]) {
    uint64_t a[4];
    memcpy(a, x, 32);
    for (int d = 0; d < 3; d++) {
        u128 rem = 0;
        for (int i = 3; i >= 0; i--) {
            u128 cur = (rem << 64) | a[i];
            a[i] = (uint64_t)(cur / P);
            rem = cur % P;
        }
        out[d] = (uint64_t)rem;
    }
}

/* Bn254PoseidonPermutation::permute (plonky2_config.rs:38-55) */
void orc_bn254_permute(uint64_t s[12]) {
    uint64_t st[5][4];
    memset(st, 0, sizeof st);
    for (int i = 0; i < 4; i++) encode3(s + 3 * i, st[i]);
    orc_bn254_permute_fr(st);
    uint64_t flat[15];
    for (int i = 0; i < 5; i++) decode3(st[i], flat + 3 * i);
    memcpy(s, flat, 12 * 8);
}
void orc_bn254_hash_no_pad(const uint64_t *in, size_t len, uint64_t out[4]) {
    uint64_t st[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t m = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < m; i++) st[i] = canon(in[off + i]);
        orc_bn254_permute(st);
    }
    memcpy(out, st, 32);
}
void orc_bn254_hash_or_noop(const uint64_t *in, size_t len, uint64_t out[4]) {
    if (len <= 4) { for (size_t i = 0; i < 4; i++) out[i] = i < len ? canon(in[i]) : 0; }
    else orc_bn254_hash_no_pad(in, len, out);
}
void orc_bn254_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
    uint64_t st[12] = {0};
    for (int i = 0; i < 4; i++) { st[i] = canon(l[i]); st[4 + i] = canon(r[i]); }
    orc_bn254_permute(st);
    memcpy(out, st, 32);
}
