// BN254 (halo2curves bn256) field arithmetic for the device: Fr and Fq elements as 8 x 32-bit limbs in Montgomery form (R = 2^256), CIOS on
// v_mad_u64_u32 (4 m < R for both primes, so products of operands < 2m stay < 2m without a final subtraction; sums and differences take one
// conditional subtraction of 2m).  Shared by the curve / FFT kernels (bn254_curve.hip) and the PLONK prover kernels (plonk_bn254.hip).
#pragma once
#include "gl355_internal.h"

#ifndef BN254C_QUAL
#define BN254C_QUAL __device__ __constant__ const
#endif
#include "bn254_curve_tables.h"
#include "bn254_addsub_asm.cuh"

namespace gl355 {

struct u256 { uint32_t l[8]; };
enum { F_R = 0, F_Q = 1 };

template <int F> GL_DEV const uint32_t* f_mod() { return F == F_Q ? BN254C_FQ_MOD : BN254C_FR_MOD; }
template <int F> GL_DEV const uint32_t* f_two_mod() { return F == F_Q ? BN254C_FQ_TWO_MOD : BN254C_FR_TWO_MOD; }
template <int F> GL_DEV const uint32_t* f_r2() { return F == F_Q ? BN254C_FQ_R2 : BN254C_FR_R2; }
template <int F> GL_DEV const uint32_t* f_one() { return F == F_Q ? BN254C_FQ_ONE : BN254C_FR_ONE; }

GL_DEV u256 u_const(const uint32_t* p) {
    u256 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = p[j];
    return r;
}
GL_DEV u256 u_zero() {
    u256 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = 0;
    return r;
}
GL_DEV bool u_is_zero(const u256& a) {
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) o |= a.l[j];
    return o == 0;
}
GL_DEV bool u_eq(const u256& a, const u256& b) {
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) o |= a.l[j] ^ b.l[j];
    return o == 0;
}
// a - m if a >= m else a
GL_DEV u256 u_cond_sub(const u256& a, const uint32_t* m) {
    uint32_t d[8];
    uint64_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint64_t v = (uint64_t)a.l[j] - m[j] - br;
        d[j] = (uint32_t)v;
        br = (v >> 32) & 1;
    }
    u256 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = br ? a.l[j] : d[j];
    return r;
}
// a * b * R^-1 (mod m), result < 2m for a, b < 2m
//
// Device build: one asm statement (bn254_mmul_asm.inc, written by tools/gen_mmul_asm.py, which explains the scheme): independent
// multiply-adds against a bank of {t_j, 0} pairs + one carry chain per half-row, 128 v_mad_u64_u32 + 128 carry adds + 8 v_mul_lo, where
// the compiled C below spends 128 + 118 64-bit adds + 268 moves.  GL355_BN254_MMUL_ASM=0 builds the C form (the A/B of DESIGN 4.8).
#ifndef GL355_BN254_MMUL_ASM
#define GL355_BN254_MMUL_ASM 1
#endif
#include "bn254_mmul_asm.inc"
template <int F>
__device__ __noinline__ u256 m_mul(u256 a, u256 b) {
    const uint32_t* M = f_mod<F>();
    const uint32_t n0 = F == F_Q ? BN254C_FQ_N0INV : BN254C_FR_N0INV;
#if GL355_BN254_MMUL_ASM && defined(__HIP_DEVICE_COMPILE__)
    u256 r;
    uint32_t u8, mm;
    uint64_t sd;
    asm(GL355_MMUL_ASM_TEXT
        : [r0] "=v"(r.l[0]), [r1] "=v"(r.l[1]), [r2] "=v"(r.l[2]), [r3] "=v"(r.l[3]), [r4] "=v"(r.l[4]), [r5] "=v"(r.l[5]), [r6] "=v"(r.l[6]),
          [r7] "=v"(r.l[7]), [u8] "=&v"(u8), [mm] "=&v"(mm), [sd] "=&s"(sd)
        : [a0] "v"(a.l[0]), [a1] "v"(a.l[1]), [a2] "v"(a.l[2]), [a3] "v"(a.l[3]), [a4] "v"(a.l[4]), [a5] "v"(a.l[5]), [a6] "v"(a.l[6]),
          [a7] "v"(a.l[7]), [b0] "v"(b.l[0]), [b1] "v"(b.l[1]), [b2] "v"(b.l[2]), [b3] "v"(b.l[3]), [b4] "v"(b.l[4]), [b5] "v"(b.l[5]),
          [b6] "v"(b.l[6]), [b7] "v"(b.l[7]), [M0] "s"(M[0]), [M1] "s"(M[1]), [M2] "s"(M[2]), [M3] "s"(M[3]), [M4] "s"(M[4]), [M5] "s"(M[5]),
          [M6] "s"(M[6]), [M7] "s"(M[7]), [n0] "s"(n0)
        : GL355_MMUL_ASM_CLOBBERS);
    return r;
#else
    uint32_t t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t t9 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c = (uint64_t)a.l[j] * b.l[i] + ((uint64_t)t[j] + c);
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        t9 = (uint32_t)(c >> 32);
        const uint32_t m = t[0] * n0;
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
    u256 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = t[j];
    return r;
#endif
}
// Sums and differences (results < 2m for operands < 2m): bn254_addsub_asm.cuh on the device, the plain C below with GL355_BN254_MMUL_ASM=0
template <int F> GL_DEV u256 m_add(const u256& a, const u256& b) {
#if GL355_BN254_MMUL_ASM && defined(__HIP_DEVICE_COMPILE__)
    const bn_limbs q = bn254_add_asm(a.l, b.l, f_two_mod<F>());
    u256 r;
#pragma unroll
    for (int j = 0; j < 8; j++) r.l[j] = q.l[j];
    return r;
#else
    u256 s;
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        c += (uint64_t)a.l[j] + b.l[j];
        s.l[j] = (uint32_t)c;
        c >>= 32;
    }
    return u_cond_sub(s, f_two_mod<F>());
#endif
}
template <int F> GL_DEV u256 m_sub(const u256& a, const u256& b) {
    const uint32_t* tm = f_two_mod<F>();
#if GL355_BN254_MMUL_ASM && defined(__HIP_DEVICE_COMPILE__)
    // a - b, plus 2m where that borrowed: a representative in [0, 2m) as the C form gives (not always the same one; every consumer
    // either multiplies on or canonicalises)
    const bn_limbs q = bn254_sub_asm(a.l, b.l, tm);
    u256 d;
#pragma unroll
    for (int j = 0; j < 8; j++) d.l[j] = q.l[j];
    return d;
#else
    u256 nb;
    uint64_t br = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint64_t v = (uint64_t)tm[j] - b.l[j] - br;
        nb.l[j] = (uint32_t)v;
        br = (v >> 32) & 1;
    }
    u256 s;
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        c += (uint64_t)a.l[j] + nb.l[j];
        s.l[j] = (uint32_t)c;
        c >>= 32;
    }
    return u_cond_sub(s, f_two_mod<F>());
#endif
}
template <int F> GL_DEV u256 m_canon(const u256& a) { return u_cond_sub(a, f_mod<F>()); }       // < 2m -> < m
template <int F> GL_DEV bool m_is_zero(const u256& a) { return u_is_zero(m_canon<F>(a)); }
template <int F> GL_DEV bool m_eq(const u256& a, const u256& b) { return u_eq(m_canon<F>(a), m_canon<F>(b)); }
// any 256-bit integer -> Montgomery form (< 2m): 2^256 < 6m, so five conditional subtractions bring the input below m first
template <int F> GL_DEV u256 m_from_int(u256 a) {
#pragma unroll 1
    for (int k = 0; k < 5; k++) a = u_cond_sub(a, f_mod<F>());
    return m_mul<F>(a, u_const(f_r2<F>()));
}
template <int F> GL_DEV u256 m_to_int(const u256& a) {
    u256 one = u_zero();
    one.l[0] = 1;
    return m_canon<F>(m_mul<F>(a, one));
}
GL_DEV u256 load256(const uint64_t* p) {
    u256 r;
#pragma unroll
    for (int i = 0; i < 4; i++) { r.l[2 * i] = (uint32_t)p[i]; r.l[2 * i + 1] = (uint32_t)(p[i] >> 32); }
    return r;
}
GL_DEV void store256(uint64_t* p, const u256& a) {
#pragma unroll
    for (int i = 0; i < 4; i++) p[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
}
template <int F> GL_DEV u256 m_pow_u64(u256 a, uint64_t e) {
    u256 r = u_const(f_one<F>());
    while (e) {
        if (e & 1) r = m_mul<F>(r, a);
        a = m_mul<F>(a, a);
        e >>= 1;
    }
    return r;
}
// a^(m-2): the inverse
template <int F> GL_DEV u256 m_inv(const u256& a) {
    const uint32_t* M = f_mod<F>();
    u256 r = u_const(f_one<F>());
#pragma unroll 1
    for (int i = 255; i >= 0; i--) {
        r = m_mul<F>(r, r);
        uint32_t w = M[i >> 5];
        if ((i >> 5) == 0) w -= 2;               // low limb of both primes is > 2: no borrow
        if ((w >> (i & 31)) & 1) r = m_mul<F>(r, a);
    }
    return r;
}


// host-side Fr helpers for the few constants a call needs (omega_n, n^-1): plain 256-bit integers with __int128
typedef unsigned __int128 u128;
struct H256 { uint64_t l[4]; };
static const uint64_t HR[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};   // r
inline bool h_geq(const H256& a) {
    for (int i = 3; i >= 0; i--) { if (a.l[i] > HR[i]) return true; if (a.l[i] < HR[i]) return false; }
    return true;
}
inline H256 h_addmod(const H256& a, const H256& b) {
    H256 r; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || h_geq(r)) { u128 br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)r.l[i] - HR[i] - (uint64_t)br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; } }
    return r;
}
inline H256 h_mulmod(const H256& a, const H256& b) {          // double-and-add: called a few dozen times per API call
    H256 r = {{0, 0, 0, 0}};
    for (int i = 255; i >= 0; i--) {
        r = h_addmod(r, r);
        if ((b.l[i >> 6] >> (i & 63)) & 1) r = h_addmod(r, a);
    }
    return r;
}
inline H256 h_powmod(H256 a, const H256& e) {
    H256 r = {{1, 0, 0, 0}};
    for (int i = 255; i >= 0; i--) {
        r = h_mulmod(r, r);
        if ((e.l[i >> 6] >> (i & 63)) & 1) r = h_mulmod(r, a);
    }
    return r;
}
inline u256 to_u256(const H256& a) {
    u256 r;
    for (int i = 0; i < 4; i++) { r.l[2 * i] = (uint32_t)a.l[i]; r.l[2 * i + 1] = (uint32_t)(a.l[i] >> 32); }
    return r;
}


inline H256 h_from_words(const uint64_t w[4]) {
    H256 a = {{w[0], w[1], w[2], w[3]}};
    while (h_geq(a)) { u128 br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)a.l[i] - HR[i] - (uint64_t)br; a.l[i] = (uint64_t)d; br = (d >> 64) & 1; } }
    return a;
}
inline H256 h_submod(const H256& a, const H256& b) {
    H256 nb; u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)HR[i] - b.l[i] - (uint64_t)br; nb.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if ((b.l[0] | b.l[1] | b.l[2] | b.l[3]) == 0) nb = H256{{0, 0, 0, 0}};
    return h_addmod(a, nb);
}
inline u256 h_to_mont(const H256& a) {
    const H256 Rm = {{BN254C_FR_ONE_64[0], BN254C_FR_ONE_64[1], BN254C_FR_ONE_64[2], BN254C_FR_ONE_64[3]}};      // R mod r
    return to_u256(h_mulmod(a, Rm));
}
inline H256 h_root_of_unity(uint32_t log_n) {
    H256 w = {{BN254C_FR_ROOT_64[0], BN254C_FR_ROOT_64[1], BN254C_FR_ROOT_64[2], BN254C_FR_ROOT_64[3]}};
    for (uint32_t k = log_n; k < BN254C_FR_S; k++) w = h_mulmod(w, w);
    return w;
}



// ---- bn254_curve.hip: resident building blocks for the PLONK prover (plonk_bn254.hip) -------------------------------------------
int32_t bn254_fr_twiddles(Ctx* ctx, uint32_t log_n, bool inverse, uint64_t* tw /* n / 2 + 1 elements */);
int32_t bn254_fr_power_table(Ctx* ctx, const uint64_t base[4], const uint64_t f[4], uint64_t count, uint64_t* tab);
int32_t bn254_fr_ntt_mont(Ctx* ctx, const uint64_t* in, uint64_t n_in, uint64_t* out, uint64_t n_out, uint32_t log_n, const uint64_t* tw,
                          const uint64_t* pre, const uint64_t* post, const uint64_t scale_plain[4], uint64_t* work);
// `bases` (gl355_bn254_g1_msm_prepare over the same points, or null): the table of the points' window multiples -- every window's digits then fall into ONE
// set of buckets per scalar set (bn254_curve.hip, "shared buckets")
int32_t bn254_msm_bits(gl355_ctx* h, const uint64_t* points, const uint64_t* scalars, uint64_t n, uint32_t m, uint32_t max_bits, uint64_t* result,
                       const gl355_msm_bases* bases = nullptr);
int32_t bn254_fr_ntt_mont_dif(Ctx* ctx, const uint64_t* in, uint64_t n_in, uint64_t* out, uint32_t log_n, const uint64_t* tw, const uint64_t* pre);
int32_t bn254_fr_ntt_mont_coset_dif(Ctx* ctx, const uint64_t* in, uint64_t n_in, uint64_t* out, uint32_t log_n, const uint64_t* tw, const uint64_t shift_plain[4],
                                    uint64_t* btw /* n elements */, bool fill);
int32_t bn254_fr_ntt_mont_from_bitrev(Ctx* ctx, const uint64_t* in, uint64_t* out, uint64_t n_out, uint32_t log_n, const uint64_t* tw, const uint64_t* post,
                                      const uint64_t scale_plain[4]);
// Q[i] = sum_{j > i} A[j] z^(j - i - 1), E = sum_j A[j] z^j on device arrays (see the division kernels)
int32_t kzg_divide(Ctx* ctx, const uint64_t* A, uint64_t m, const H256& z, int a_is_mont, uint64_t* Q, int q_plain, uint64_t* E_mont);
}  // namespace gl355
