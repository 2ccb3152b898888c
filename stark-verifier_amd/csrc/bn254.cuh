// BN254-Poseidon over Goldilocks sponge states for gfx950: the reference's second hasher
// (src/plonky2_verifier/bn245_poseidon/native.rs:16-77, plonky2_config.rs:38-75; parameters constants.rs:5-404),
// the hash of the final wrap proof's Merkle trees and transcript (wrapper.rs:35-56 with OuterC).  SURVEY 8(f) N1.
//
// One sponge state (12 Goldilocks elements) per lane.  The permutation packs 3 elements per BN254 scalar
// (x0 + x1 p + x2 p^2, p = the Goldilocks prime), runs Poseidon t = 5 (8 full + 60 partial rounds, x^5, dense 5x5 MDS)
// and splits each of the 5 results back into its three low base-p digits.
//
// Fr arithmetic: 8 x 32-bit limbs, Montgomery form with R = 2^256, CIOS with v_mad_u64_u32.  4r < R, so products of
// operands < 2r stay < 2r WITHOUT the final conditional subtraction; sums are brought back under 2r by one conditional
// subtraction of 2r, and values are canonicalised only when they leave the permutation.  fr_mul is deliberately not
// inlined: a round calls it 28-40 times and the inlined body (~450 instructions) would not fit the instruction cache.
//
// Round 4: GL355_BN254_HASH_F29 (default 1) runs the same permutation on nine 29-bit limbs with Montgomery radix 2^261 (bn254_f29.cuh, tables
// bn254_tables29.h from the same generator): products as carry-free column sums, INLINED, sums without carries or conditional subtractions
// (a value may grow to ~85 r over the 60 partial rounds; anything below 2^261 ~ 169 r is a legal operand of the next product).  These kernels
// run at one wave per SIMD or less -- the small Merkle levels of a wrap proof are chains of dependent products -- and there the form is
// twice as fast per product as the 8 x 32-bit asm one (tools/ubench/ubench_mont29.hip with 256 blocks: 943 against 1 850 clocks).
#pragma once
#include "gl_field.cuh"

#ifndef GL355_BN254_HASH_F29
#define GL355_BN254_HASH_F29 1
#endif
#define BN254_TABLE_QUAL __device__ __constant__ const
#include "bn254_tables.h"
#if GL355_BN254_HASH_F29
#include "bn254_f29.cuh"
#include "bn254_tables29.h"
#define BNT(NAME) BN254F_##NAME
#define FR_W 9
#else
#define BNT(NAME) BN254_##NAME
#define FR_W 8
#endif
#ifndef GL355_BN254_MMUL_ASM
#define GL355_BN254_MMUL_ASM 1
#endif
#include "bn254_mmul_asm.inc"
#include "bn254_addsub_asm.cuh"

namespace gl355 {

#if GL355_BN254_HASH_F29
typedef f29 fr8;                                      // (the name stays: "an Fr element of the hasher")
GL_DEV fr8 fr_mul(const fr8& a, const fr8& b) { return f29_mul_fr(a, b); }        // a's limbs < 2^30.6, b's < 2^29
GL_DEV fr8 fr_add(const fr8& a, const fr8& b) { return f29_norm(f29_add(a, b)); }  // limbs back under 2^29; the VALUE is left to grow
GL_DEV fr8 fr_pow5(const fr8& a) {
    const fr8 a2 = fr_mul(a, a), a4 = fr_mul(a2, a2);
    return fr_mul(a4, a);
}
GL_DEV fr8 fr_const(const uint32_t* p) { return f29_const(p); }
GL_DEV fr8 fr_zero() { fr8 r; for (int j = 0; j < 9; j++) r.l[j] = 0; return r; }
// a plain integer below 2^256 (four 64-bit words) into the Montgomery form, and a value (< 169 r) back to its canonical integer
GL_DEV fr8 fr_enter(const uint64_t a[4]) {
    u256 x;
#pragma unroll
    for (int i = 0; i < 4; i++) { x.l[2 * i] = (uint32_t)a[i]; x.l[2 * i + 1] = (uint32_t)(a[i] >> 32); }
    return fr_mul(f29_from_u256(x), fr_const(BN254F_R2));
}
GL_DEV void fr_leave(const fr8& xm, uint64_t a[4]) {
    fr8 one = fr_zero();
    one.l[0] = 1;
    fr8 x = fr_mul(xm, one);                          // x 2^-261: < r + 1, i.e. canonical or exactly r (for zero)
    uint32_t e = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) e |= x.l[j] ^ FR29_R[j];
    if (e == 0) x = fr_zero();
    const u256 w = f29_to_u256(x);
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = (uint64_t)w.l[2 * i] | ((uint64_t)w.l[2 * i + 1] << 32);
}
#else
struct fr8 { uint32_t l[8]; };

// a * b * R^-1 (mod r), result < 2r for a, b < 2r
__device__ __noinline__ fr8 fr_mul(fr8 a, fr8 b) {
    constexpr uint32_t M[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    constexpr uint32_t N0INV = 0xefffffffu;
#if GL355_BN254_MMUL_ASM && defined(__HIP_DEVICE_COMPILE__)
    // the hand-scheduled product of bn254_field.cuh (tools/gen_mmul_asm.py): 128 multiply-adds + 128 carry adds, no moves
    fr8 q;
    uint32_t u8, mm;
    uint64_t sd;
    asm(GL355_MMUL_ASM_TEXT
        : [r0] "=v"(q.l[0]), [r1] "=v"(q.l[1]), [r2] "=v"(q.l[2]), [r3] "=v"(q.l[3]), [r4] "=v"(q.l[4]), [r5] "=v"(q.l[5]), [r6] "=v"(q.l[6]),
          [r7] "=v"(q.l[7]), [u8] "=&v"(u8), [mm] "=&v"(mm), [sd] "=&s"(sd)
        : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]), [a5] "v"(a.l[5]), [a6] "v"(a.l[6]),
          [a7] "v"(a.l[7]), [b0] "v"(b.l[0]), [b1] "v"(b.l[1]), [b2] "v"(b.l[2]), [b3] "v"(b.l[3]), [b4] "v"(b.l[4]), [b5] "v"(b.l[5]),
          [b6] "v"(b.l[6]), [b7] "v"(b.l[7]), [M0] "s"(M[0]), [M1] "s"(M[1]), [M2] "s"(M[2]), [M3] "s"(M[3]), [M4] "s"(M[4]), [M5] "s"(M[5]),
          [M6] "s"(M[6]), [M7] "s"(M[7]), [n0] "s"(N0INV)
        : GL355_MMUL_ASM_CLOBBERS);
    return q;
#else
    uint32_t t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t t9 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c = (uint64_t)a.l[j] * b.l[i] + ((uint64_t)t[j] + c);      // <= (2^32-1)^2 + 2(2^32-1) = 2^64 - 1
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        t9 = (uint32_t)(c >> 32);
        const uint32_t m = t[0] * N0INV;
        c = ((uint64_t)m * M[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c = (uint64_t)m * M[j] + ((uint64_t)t[j] + c);
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = t9 + (uint32_t)(c >> 32);
    }
    fr8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = t[j];     // t[8] == 0 here: the result is < 2r < 2^256
    return r;
#endif
}

// r = a - m if a >= m else a   (m: 8-limb constant table)
GL_DEV fr8 fr_cond_sub(fr8 a, const uint32_t* m) {
    uint32_t d[8];
    uint64_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint64_t v = (uint64_t)a.l[j] - m[j] - br;
        d[j] = (uint32_t)v;
        br = (v >> 32) & 1;
    }
    fr8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = br ? a.l[j] : d[j];
    return r;
}
// (a + b) brought back under 2r; a, b < 2r so the sum fits 256 bits (4r < 2^256)
GL_DEV fr8 fr_add(fr8 a, fr8 b) {
    constexpr uint32_t TWO_R[8] = {0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u, 0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};
#if GL355_BN254_MMUL_ASM && defined(__HIP_DEVICE_COMPILE__)
    const bn_limbs q = bn254_add_asm(a.l, b.l, TWO_R);
    fr8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = q.l[j];
    return r;
#endif
    fr8 s;
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        c += (uint64_t)a.l[j] + b.l[j];
        s.l[j] = (uint32_t)c;
        c >>= 32;
    }
    return fr_cond_sub(s, TWO_R);
}
GL_DEV fr8 fr_pow5(fr8 a) {
    const fr8 a2 = fr_mul(a, a), a4 = fr_mul(a2, a2);
    return fr_mul(a4, a);
}
GL_DEV fr8 fr_const(const uint32_t* p) {
    fr8 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = p[j];
    return r;
}

GL_DEV fr8 fr_zero() { fr8 r; for (int j = 0; j < 8; j++) r.l[j] = 0; return r; }
GL_DEV fr8 fr_enter(const uint64_t a[4]) {
    fr8 r;
#pragma unroll
    for (int i = 0; i < 4; i++) { r.l[2 * i] = (uint32_t)a[i]; r.l[2 * i + 1] = (uint32_t)(a[i] >> 32); }
    return fr_mul(r, fr_const(BN254_R2));
}
GL_DEV void fr_leave(fr8 xm, uint64_t a[4]) {
    fr8 one;
#pragma unroll
    for (int j = 0; j < 8; j++) one.l[j] = j == 0 ? 1u : 0u;
    fr8 x = fr_mul(xm, one);                       // x * R^-1: out of Montgomery form, < 2r
    x = fr_cond_sub(x, BN254_MOD);
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = (uint64_t)x.l[2 * i] | ((uint64_t)x.l[2 * i + 1] << 32);
}
#endif

// three Goldilocks elements -> x0 + x1 p + x2 p^2 (< 2^192, no reduction needed), then into Montgomery form
GL_DEV fr8 fr_encode3(uint64_t x0, uint64_t x1, uint64_t x2) {
    // a = x2; a = a * p + x1; a = a * p + x0 with p = 2^64 - 2^32 + 1: a * p = (a << 64) - (a << 32) + a
    uint64_t a[4] = {gl_canon(x2), 0, 0, 0};
    const uint64_t add[2] = {gl_canon(x1), gl_canon(x0)};
#pragma unroll
    for (int step = 0; step < 2; step++) {
        unsigned __int128 c = add[step];
        uint64_t o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c += (unsigned __int128)a[i] * GL_P;
            o[i] = (uint64_t)c;
            c >>= 64;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = o[i];
    }
    return fr_enter(a);
}

// canonical Fr value (leaves Montgomery form) -> its three low base-p digits
GL_DEV void fr_decode3(const fr8& xm, uint64_t out[3]) {
    uint64_t a[4];
    fr_leave(xm, a);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        // remainder: Horner over the 64-bit limbs with 2^64 = 2^32 - 1 (mod p)
        uint64_t r = gl_canon(a[3]);
#pragma unroll
        for (int i = 2; i >= 0; i--) r = gl_canon(gl_reduce128(a[i], r));
        out[d] = r;
        if (d == 2) break;
        // exact quotient (a - r) / p = (a - r) * p^-1 mod 2^256
        uint64_t s[4];
        unsigned __int128 br = (unsigned __int128)a[0] - r;
        s[0] = (uint64_t)br;
        uint64_t borrow = (uint64_t)(br >> 64) & 1;
#pragma unroll
        for (int i = 1; i < 4; i++) {
            br = (unsigned __int128)a[i] - borrow;
            s[i] = (uint64_t)br;
            borrow = (uint64_t)(br >> 64) & 1;
        }
        // p^-1 mod 2^256 as 64-bit limbs
        constexpr uint64_t PINV[4] = {0x0000000100000001ull, 0xffffffff00000000ull, 0xfffffffffffffffeull, 0x0000000100000000ull};
        uint64_t q[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            unsigned __int128 c = 0;
#pragma unroll
            for (int j = 0; i + j < 4; j++) {
                c += (unsigned __int128)s[i] * PINV[j] + q[i + j];
                q[i + j] = (uint64_t)c;
                c >>= 64;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = q[i];
    }
}

// the Fr permutation (native.rs:45-62), values in Montgomery form and < 2r throughout.  The 8 full rounds follow the
// definition (constants, x^5 on every element, dense 5x5 MDS: 40 products each).  The 60 partial rounds use the derived
// sparse form (tools/gen_bn254_tables.py derive_fast, checked there against the definition): one pre-matrix (16 products),
// then per round x^5 on element 0 and 9 products instead of 25 -- 1 056 products per permutation instead of 2 000.
GL_DEV void bn254_full_round(fr8 (&s)[5], int rnd) {
#pragma unroll
    for (int i = 0; i < 5; i++) s[i] = fr_pow5(fr_add(s[i], fr_const(BNT(RC)[5 * rnd + i])));
    fr8 n[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        fr8 acc = fr_mul(s[0], fr_const(BNT(MDS)[5 * i]));
#pragma unroll
        for (int j = 1; j < 5; j++) acc = fr_add(acc, fr_mul(s[j], fr_const(BNT(MDS)[5 * i + j])));
        n[i] = acc;
    }
#pragma unroll
    for (int i = 0; i < 5; i++) s[i] = n[i];
}
GL_DEV void bn254_permute_fr(fr8 (&s)[5]) {
#pragma unroll 1
    for (int rnd = 0; rnd < 4; rnd++) bn254_full_round(s, rnd);
    // ---- 60 partial rounds ----
#pragma unroll
    for (int i = 0; i < 5; i++) s[i] = fr_add(s[i], fr_const(BNT(PART_FIRST)[i]));
    {
        fr8 n[5];
        n[0] = s[0];
#pragma unroll
        for (int c = 1; c < 5; c++) {
            fr8 acc = fr_mul(s[1], fr_const(BNT(PART_INIT)[c - 1]));
#pragma unroll
            for (int r = 2; r < 5; r++) acc = fr_add(acc, fr_mul(s[r], fr_const(BNT(PART_INIT)[(r - 1) * 4 + (c - 1)])));
            n[c] = acc;
        }
#pragma unroll
        for (int i = 0; i < 5; i++) s[i] = n[i];
    }
#pragma unroll 1
    for (int r = 0; r < 60; r++) {
        fr8 s0 = fr_pow5(s[0]);
        if (r < 59) s0 = fr_add(s0, fr_const(BNT(PART_POST)[r]));
        fr8 d = fr_mul(s0, fr_const(BNT(PART_M00)[0]));
#pragma unroll
        for (int i = 1; i < 5; i++) {
            d = fr_add(d, fr_mul(s[i], fr_const(BNT(PART_WHAT)[4 * r + (i - 1)])));
            s[i] = fr_add(s[i], fr_mul(s0, fr_const(BNT(PART_VS)[4 * r + (i - 1)])));
        }
        s[0] = d;
    }
#pragma unroll 1
    for (int rnd = 64; rnd < 68; rnd++) bn254_full_round(s, rnd);
}

// Bn254PoseidonPermutation::permute on the 12-element sponge state (plonky2_config.rs:38-55)
GL_DEV void bn254_permute(uint64_t (&s)[12]) {
    fr8 st[5];
#pragma unroll
    for (int i = 0; i < 4; i++) st[i] = fr_encode3(s[3 * i], s[3 * i + 1], s[3 * i + 2]);
    st[4] = fr_zero();
    bn254_permute_fr(st);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t d[3];
        fr_decode3(st[i], d);
        s[3 * i] = d[0]; s[3 * i + 1] = d[1]; s[3 * i + 2] = d[2];
    }
}

}  // namespace gl355
