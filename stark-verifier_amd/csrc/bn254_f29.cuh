// BN254 base field (Fq) on nine 29-bit limbs for the MSM's bucket accumulation (bn254_curve.hip): the mixed Jacobian addition with
//   * products as carry-free column sums: every limb product lands in a 64-bit accumulator, 162 v_mad_u64_u32 + 58 shifts / adds / masks per
//     product and no moves (plain C: the compiler's accumulate-in-place form), Montgomery radix R' = 2^261 -- 1.26x the 8 x 32-bit asm product
//     as a bare product (tools/ubench/ubench_mont29.hip, profiles/r04_ubench_mont29.txt), and it is INLINED (its 9-word operands would
//     travel through scratch memory as call arguments);
//   * sums and differences lazily: limb by limb, no carries, no conditional subtraction; a difference a - b is a + (K q - b) with K q in a
//     representation whose low limbs carry 2^29 (or 2^30) extra, so that no limb goes negative; values are bounded by the formulas'
//     structure (every result that grows is the operand of a product next, and a product brings anything below 2^261 ~ 169 q back under
//     1.3 q) and limbs are re-normalised only where a product needs them (< 2^29 for the second operand, < 2^30.6 for the first).
// Values live in the R' = 2^261 Montgomery domain while they are in this form: the point table of an MSM is written in it
// (msm_digits_kernel), a finished bucket sum goes back to the 8 x 32-bit R = 2^256 form the reduction kernels use (three products by 2^256 mod q).
#pragma once
#include "bn254_field.cuh"

namespace gl355 {

struct f29 { uint32_t l[9]; };
#define F29_MASK 0x1fffffffu
#define F29_N0 0x04866389u                  /* -q^-1 mod 2^29 */
#define F29_DEF(NAME, ...) __device__ __constant__ const uint32_t NAME[9] = {__VA_ARGS__};
F29_DEF(FQ29_Q, 0x187cfd47, 0x010460b6, 0x1c72a34f, 0x02d522d0, 0x1585d978, 0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e)
F29_DEF(FQ29_ONE, 0x157ccc21, 0x141c2758, 0x185230d3, 0x014c0419, 0x0aa36fb9, 0x1d4240ce, 0x11d54c07, 0x052ac7a8, 0x000dc836)      // 2^261 mod q
F29_DEF(FQ29_R256, 0x058f0d9d, 0x1aea1c6e, 0x11c2cf74, 0x11d651eb, 0x1462c0a7, 0x11b7bc3c, 0x1cbd99ba, 0x183340fb, 0x000e0a77)     // 2^256 mod q
// K q with 2^29 lent to every low limb (subtrahends with limbs < 2^29 and value < K q) ...
F29_DEF(FQ29_C2, 0x30f9fa8e, 0x2208c16c, 0x38e5469d, 0x25aa45a0, 0x2b0bb2ef, 0x25b68180, 0x214dc281, 0x3cb84c67, 0x0060c89b)
F29_DEF(FQ29_C4, 0x21f3f51c, 0x241182da, 0x31ca8d3b, 0x2b548b42, 0x361765df, 0x2b6d0301, 0x229b8503, 0x397098cf, 0x00c19138)
F29_DEF(FQ29_C8, 0x23e7ea38, 0x282305b5, 0x23951a77, 0x36a91686, 0x2c2ecbbf, 0x36da0604, 0x25370a07, 0x32e1319f, 0x01832272)
F29_DEF(FQ29_C16, 0x27cfd470, 0x30460b6b, 0x272a34ef, 0x2d522d0d, 0x385d9780, 0x2db40c09, 0x2a6e1410, 0x25c2633f, 0x030644e6)
F29_DEF(FQ29_LIFT, 0x13349ca1, 0x1a5d84a8, 0x0a3e5cac, 0x100249e0, 0x12b951e8, 0x0e92d304, 0x14cb95b3, 0x041b9d3d, 0x00058003)     // 2^266 mod q
// 32 * 2^256 mod q as 8 x 32-bit words: the Montgomery form (R = 2^256) of 32, which lifts an R-domain value into the R' domain
__device__ __constant__ const uint32_t FQ_C32[8] = {0x157ccc21, 0x4e8384eb, 0x0ce148c3, 0xfb90a602, 0x819caa36, 0x5301fa84, 0x563d4475, 0x0dc83629};

GL_DEV f29 f29_const(const uint32_t* p) {
    f29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) r.l[j] = p[j];
    return r;
}
// the same integer in 29-bit slices (a < 2^256)
GL_DEV f29 f29_from_u256(const u256& a) {
    uint64_t w[5];
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
    w[4] = 0;
    f29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int bit = 29 * j, k = bit >> 6, o = bit & 63;
        uint64_t v = w[k] >> o;
        if (o > 35) v |= w[k + 1] << (64 - o);
        r.l[j] = (uint32_t)v & F29_MASK;
    }
    return r;
}
// limbs normalised (< 2^29, the top one free), value < 2^256
GL_DEV u256 f29_to_u256(const f29& a) {
    uint64_t w[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int bit = 29 * j, k = bit >> 6, o = bit & 63;
        w[k] |= (uint64_t)a.l[j] << o;
        if (o > 35) w[k + 1] |= (uint64_t)a.l[j] >> (64 - o);
    }
    u256 r;
#pragma unroll
    for (int i = 0; i < 4; i++) { r.l[2 * i] = (uint32_t)w[i]; r.l[2 * i + 1] = (uint32_t)(w[i] >> 32); }
    return r;
}
GL_DEV f29 f29_norm(f29 a) {                   // limbs < 2^32 -> < 2^29 (the top limb takes what is left)
#pragma unroll
    for (int j = 0; j < 8; j++) { a.l[j + 1] += a.l[j] >> 29; a.l[j] &= F29_MASK; }
    return a;
}
GL_DEV f29 f29_add(const f29& a, const f29& b) {
    f29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) r.l[j] = a.l[j] + b.l[j];
    return r;
}
// a + (K q - b): C = the lent representation of K q; b's limbs below what C lends, b < K q
GL_DEV f29 f29_sub(const f29& a, const f29& b, const uint32_t* C) {
    f29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) r.l[j] = a.l[j] + (C[j] - b.l[j]);
    return r;
}
GL_DEV f29 f29_neg(const f29& b, const uint32_t* C) {
    f29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) r.l[j] = C[j] - b.l[j];
    return r;
}
// a * b * 2^-261 mod m (m, n0 = -m^-1 mod 2^29: compile-time constants): a's limbs < 2^30.6, b's < 2^29.  Result: limbs < 2^29 (the top one takes
// what is left), value < a b / 2^261 + m -- anything whose product stays below 2^261 m is a legal operand, so sums may pile up between products.
GL_DEV f29 f29_mul_mod(const f29& a, const f29& b, const uint32_t* M, uint32_t n0) {
    uint64_t t[10];
#pragma unroll
    for (int j = 0; j < 10; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)a.l[j] * b.l[i];
        const uint32_t m = ((uint32_t)t[0] * n0) & F29_MASK;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)m * M[j];
        const uint64_t c = t[0] >> 29;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] = t[j + 1];
        t[9] = 0;
        t[0] += c;
    }
    f29 r;
#pragma unroll
    for (int j = 0; j < 8; j++) { r.l[j] = (uint32_t)t[j] & F29_MASK; t[j + 1] += t[j] >> 29; }
    r.l[8] = (uint32_t)t[8];
    return r;
}
GL_DEV f29 f29_mul(const f29& a, const f29& b) { return f29_mul_mod(a, b, FQ29_Q, F29_N0); }               // the base field (the MSM)
// the scalar field (the BN254-Poseidon hasher, bn254.cuh): r on 29-bit limbs, -r^-1 mod 2^29
F29_DEF(FR29_R, 0x10000001, 0x1f0fac9f, 0x0e5c2450, 0x07d090f3, 0x1585d283, 0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e)
#define FR29_N0 0x0fffffffu
GL_DEV f29 f29_mul_fr(const f29& a, const f29& b) { return f29_mul_mod(a, b, FR29_R, FR29_N0); }
// a product's result (normalised, < 1.3 q, = 0 mod q)  <=>  it is 0 or q
GL_DEV bool f29_is_zero_mod(const f29& a) {
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) { z |= a.l[j]; e |= a.l[j] ^ FQ29_Q[j]; }
    return z == 0 || e == 0;
}

// an R-domain 8 x 32-bit value (< 2q) in the R' domain on 29-bit limbs, and back (normalised limbs, any value below 2^261 -> < 1.3 q)
GL_DEV f29 f29_lift(const u256& a) { return f29_from_u256(m_mul<F_Q>(a, u_const(FQ_C32))); }
GL_DEV u256 f29_lower(const f29& a) { return f29_to_u256(f29_mul(a, f29_const(FQ29_R256))); }
// the same lift with a 29-bit product instead of the called 8 x 32-bit one: x 2^256 (sliced) times 2^266 / 2^261
GL_DEV f29 f29_lift_inl(const u256& a) { return f29_mul(f29_from_u256(a), f29_const(FQ29_LIFT)); }

}  // namespace gl355
