// Poseidon permutation over Goldilocks (width 12, x^7, 4 + 22 + 4 rounds) for gfx950: one state per
// lane, 12 x u64 in VGPRs, round constants read through the scalar cache (uniform index).
//
// Replaces plonky2::hash::poseidon::{Poseidon::poseidon, PoseidonPermutation::permute}, reached from
// the reference at src/plonky2_semaphore/access_set.rs:67, signal.rs:35 and inside every Merkle
// build.  Parameters and round structure: src/plonky2_verifier/chip/plonk/gates/poseidon.rs:26-322
// (constants), :450-486 (MDS row), :504-589 and :634-686 (fast partial rounds).  The fast-partial
// tables are derived by tools/gen_poseidon_tables.py and equal the reference's literals.
#pragma once
#include "gl_field.cuh"

#define PSD_TABLE_QUAL __device__ __constant__ const
#include "poseidon_tables.h"

namespace gl355 {

// x^7 = (x x^2)(x^2 x^2).  PSD_SBOX_INTERLEAVE 1 (default, round 5): x^3 and x^4 of one S-box, and everything of the two S-boxes of psd_sbox2, go
// through gl_mul_multi, whose products fill each other's carry wait states; 0: one product after the other (A/B).
#ifndef PSD_SBOX_INTERLEAVE
#define PSD_SBOX_INTERLEAVE 1
#endif
#ifndef PSD_SBOX_WIDTH
#define PSD_SBOX_WIDTH 4          // S-boxes of a full round per lock-step group: 2 or 4 (4: no wait states at all; -4 % at one wave per SIMD, nothing at full occupancy)
#endif
GL_DEV uint64_t psd_sbox(uint64_t x) {
#if PSD_SBOX_INTERLEAVE && defined(__HIP_DEVICE_COMPILE__) && GL_MUL_VARIANT == 1
    const uint64_t x2 = gl_sqr(x);
    const uint64_t a[2] = {x, x2}, b[2] = {x2, x2};
    uint64_t r[2];
    gl_mul_multi<2>(a, b, r);
    return gl_mul(r[0], r[1]);
#else
    uint64_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
#endif
}
GL_DEV void psd_sbox2(uint64_t& x, uint64_t& y) {
#if PSD_SBOX_INTERLEAVE && defined(__HIP_DEVICE_COMPILE__) && GL_MUL_VARIANT == 1
    uint64_t sq[2];
    { const uint64_t a[2] = {x, y}; gl_mul_multi<2>(a, a, sq); }
    uint64_t r[4];
    { const uint64_t a[4] = {x, sq[0], y, sq[1]}, b[4] = {sq[0], sq[0], sq[1], sq[1]}; gl_mul_multi<4>(a, b, r); }
    { const uint64_t a[2] = {r[0], r[2]}, b[2] = {r[1], r[3]}; uint64_t o[2]; gl_mul_multi<2>(a, b, o); x = o[0]; y = o[1]; }
#else
    x = psd_sbox(x); y = psd_sbox(y);
#endif
}

// al + ah * 2^32  ->  u64 representative, for accumulators whose top word ah1 = ah >> 32 satisfies ah1 + 1 < 2^32: the two chains of an MDS row
// (al, ah < 2^44) and the block form's dot products (66 products < 2^54 plus a 32-bit constant: al, ah < 2^61, ah1 < 2^29; the bound is asserted on
// the generated tables by tools/gen_poseidon_tables.py and tests/test_tables.py).
// = al0 + (al1 + ah0) 2^32 + ah1 2^64: one 32-bit add with carry into the top word, then top * EPS with the carry-out repaid.
GL_DEV uint64_t psd_recombine(uint64_t al, uint64_t ah) {
#if GL_MUL_VARIANT == 1 && defined(__HIP_DEVICE_COMPILE__)
    uint32_t t1 = (uint32_t)(al >> 32), top;     // t1 is updated in place: (al0, t1) stays the register pair of al
    asm("v_add_co_u32_e32 %0, vcc, %0, %2\n\t" GL_HAZARD_NOP "v_addc_co_u32_e32 %1, vcc, 0, %3, vcc"
        : "+v"(t1), "=v"(top) : "v"((uint32_t)ah), "v"((uint32_t)(ah >> 32)) : "vcc");
    return gl_dev_add_mul_eps(((uint64_t)t1 << 32) | (uint32_t)al, top);
#else
    const uint64_t mid = ah << 32;
    const uint32_t top = (uint32_t)(ah >> 32);
    const uint64_t r0 = al + mid;
    const uint64_t carry = r0 < mid ? 1u : 0u;
    const uint64_t t = (uint64_t)(top + carry) * GL_EPS;  // top + carry < 2^32 (see above): the product fits 64 bits
    uint64_t r1 = r0 + t;
    if (r1 < t) r1 += GL_EPS;
    return r1;
#endif
}

// four S-boxes at once: every phase is four products wide, no wait states at all
GL_DEV void psd_sbox4(uint64_t& x0, uint64_t& x1, uint64_t& x2, uint64_t& x3) {
#if PSD_SBOX_INTERLEAVE && defined(__HIP_DEVICE_COMPILE__) && GL_MUL_VARIANT == 1
    const uint64_t a[4] = {x0, x1, x2, x3};
    uint64_t sq[4], c3[4], c4[4], o[4];
    gl_mul_multi<4>(a, a, sq);
    gl_mul_multi<4>(a, sq, c3);
    gl_mul_multi<4>(sq, sq, c4);
    gl_mul_multi<4>(c3, c4, o);
    x0 = o[0]; x1 = o[1]; x2 = o[2]; x3 = o[3];
#else
    x0 = psd_sbox(x0); x1 = psd_sbox(x1); x2 = psd_sbox(x2); x3 = psd_sbox(x3);
#endif
}
// N recombinations in lock-step (the two carry steps of one fill the wait states of the others: see gl_mul_multi)
template <int N>
GL_DEV void psd_recombine_multi(const uint64_t (&al)[N], const uint64_t (&ah)[N], uint64_t (&out)[N]) {
#if GL_MUL_VARIANT == 1 && defined(__HIP_DEVICE_COMPILE__)
    static_assert(N >= 3, "three or more: no wait states of their own");
    uint32_t t1[N], top[N], m[N];
    uint64_t c[N], r[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(t1[j]), "=s"(c[j]) : "v"((uint32_t)(al[j] >> 32)), "v"((uint32_t)ah[j]));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        asm("v_addc_co_u32_e64 %0, %1, 0, %2, %1" : "=v"(top[j]), "+s"(c[j]) : "v"((uint32_t)(ah[j] >> 32)));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        const uint64_t x = ((uint64_t)t1[j] << 32) | (uint32_t)al[j];
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(r[j]), "=s"(c[j]) : "v"(top[j]), "v"(x));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(m[j]) : "s"(c[j]));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < N; j++) out[j] = r[j] + (uint64_t)m[j];
#else
#pragma unroll
    for (int j = 0; j < N; j++) out[j] = psd_recombine(al[j], ah[j]);
#endif
}

// Full MDS layer, plus the NEXT round's constants.  Row r: sum_i s[(i+r)%12] * CIRC[i] + 8*s[0] (r == 0) + rc[r].  The entries
// are < 64, so the 32-bit halves are accumulated separately in u64 (12 * 41 * 2^32 + 2^32 < 2^43: no overflow) with
// v_mad_u64_u32 and recombined once per row; the two halves of the (canonical) round constant are the initial addends of
// the two chains (scalar operands), so adding the constants costs no vector instruction.
//
// Each chain is written out as 12 multiply-adds: left to itself the compiler re-associates the sums (chains started at 0
// and joined with extra 64-bit adds) and turns the entries 16 and 2 into shift-adds that need the 32-bit half zero-extended
// into a register pair first: 382-445 vector instructions per layer instead of 24 * 12 + 5 * 12 = 348.
#define PSD_MDS_CHAIN(C0)                                                                                  \
    "v_mad_u64_u32 %0, %1, %2, " #C0 ", %14\n\tv_mad_u64_u32 %0, %1, %3, 15, %0\n\tv_mad_u64_u32 %0, %1, %4, 41, %0\n\t" \
    "v_mad_u64_u32 %0, %1, %5, 16, %0\n\tv_mad_u64_u32 %0, %1, %6, 2, %0\n\tv_mad_u64_u32 %0, %1, %7, 28, %0\n\t"      \
    "v_mad_u64_u32 %0, %1, %8, 13, %0\n\tv_mad_u64_u32 %0, %1, %9, 13, %0\n\tv_mad_u64_u32 %0, %1, %10, 39, %0\n\t"    \
    "v_mad_u64_u32 %0, %1, %11, 18, %0\n\tv_mad_u64_u32 %0, %1, %12, 34, %0\n\tv_mad_u64_u32 %0, %1, %13, 20, %0"
template <int R>
GL_DEV uint64_t psd_mds_chain(const uint32_t (&x)[12], uint64_t addend) {
    uint64_t acc, unused;
#define PSD_X(i) "v"(x[((i) + R) % 12])
    if constexpr (R == 0)
        asm(PSD_MDS_CHAIN(25) : "=&v"(acc), "=&s"(unused) : PSD_X(0), PSD_X(1), PSD_X(2), PSD_X(3), PSD_X(4), PSD_X(5), PSD_X(6),
            PSD_X(7), PSD_X(8), PSD_X(9), PSD_X(10), PSD_X(11), "s"(addend));
    else
        asm(PSD_MDS_CHAIN(17) : "=&v"(acc), "=&s"(unused) : PSD_X(0), PSD_X(1), PSD_X(2), PSD_X(3), PSD_X(4), PSD_X(5), PSD_X(6),
            PSD_X(7), PSD_X(8), PSD_X(9), PSD_X(10), PSD_X(11), "s"(addend));
#undef PSD_X
    return acc;
}
template <int R>
GL_DEV void psd_mds_rows(uint64_t (&s)[12], const uint32_t (&lo)[12], const uint32_t (&hi)[12], const uint64_t* rc) {
    if constexpr (R < 12) {
#if PSD_SBOX_INTERLEAVE
        // four rows at a time: eight chains, then their recombinations in lock-step
        const uint64_t c0 = rc[R], c1 = rc[R + 1], c2 = rc[R + 2], c3 = rc[R + 3];
        const uint64_t al[4] = {psd_mds_chain<R>(lo, (uint32_t)c0), psd_mds_chain<R + 1>(lo, (uint32_t)c1), psd_mds_chain<R + 2>(lo, (uint32_t)c2),
                                psd_mds_chain<R + 3>(lo, (uint32_t)c3)};
        const uint64_t ah[4] = {psd_mds_chain<R>(hi, c0 >> 32), psd_mds_chain<R + 1>(hi, c1 >> 32), psd_mds_chain<R + 2>(hi, c2 >> 32),
                                psd_mds_chain<R + 3>(hi, c3 >> 32)};
        uint64_t o[4];
        __builtin_amdgcn_sched_barrier(0);
        psd_recombine_multi<4>(al, ah, o);
        s[R] = o[0]; s[R + 1] = o[1]; s[R + 2] = o[2]; s[R + 3] = o[3];
        psd_mds_rows<R + 4>(s, lo, hi, rc);
#else
        const uint64_t c = rc[R];
        s[R] = psd_recombine(psd_mds_chain<R>(lo, (uint32_t)c), psd_mds_chain<R>(hi, c >> 32));
        psd_mds_rows<R + 1>(s, lo, hi, rc);
#endif
    }
}
// rc: 12 constants at a wave-uniform address (scalar loads)
GL_DEV void psd_mds(uint64_t (&s)[12], const uint64_t* rc) {
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (uint32_t)s[i]; hi[i] = (uint32_t)(s[i] >> 32); }
    psd_mds_rows<0>(s, lo, hi, rc);
}


// ---- the 22 partial rounds in BLOCK form (round 5) ------------------------------------------------------------------------------
// Only lane 0 meets the S-box in a partial round, so over a block of B = 11 rounds lanes 1..11 are a linear function of the state u that
// entered the block and of the block's S-box outputs x_0..x_{B-1} (gates/poseidon.rs:504-589 factors the same linearity round by round;
// here it is unrolled over a block, tools/gen_poseidon_tables.py has the algebra and a limb-exact model):
//     y_j  = <alpha_j, u> + sum_{i<j} x_i kappa_{j-1-i} + gamma_j     (lane 0 entering round j of the block),   x_j = y_j^7
//     s'_r = <(A^B)_r, u> + sum_{i<B} x_i beta_{B-1-i}[r] + Gamma_r   (the state leaving the block)
// Every output is ONE dot product with 64-bit constants, reduced once: 429 multiply-accumulates + 22 reductions per block where the dense form
// spends 11 x 144 small-constant multiply-accumulates + 11 x 12 reductions -- 6 864 instead of 8 976 vector instructions for the 22 rounds.
// A multiply-accumulate x * c costs six v_mad_u64_u32 and nothing else: x is split ONCE into limbs of 22 / 22 / 20 bits (each x and u is used by
// 11..22 outputs), the table holds (c, 2^22 c, 2^44 c) mod p, limb k times the low / high half of the k-th constant goes into the low / high
// accumulator; 66 products < 2^54 and a 32-bit constant cannot overflow 64 bits, and psd_recombine folds al + ah 2^32 as for an MDS row.
// (the tables are NOT const-qualified on the device: with the values visible the compiler strength-reduces the small ones into 64-bit shifts and
// adds, keeps every limb as a zero-extended register PAIR for them -- 174 VGPRs -- and materialises the rest as 32-bit literals with a wait state each)
#undef PSD_TABLE_QUAL
#define PSD_TABLE_QUAL __device__ __constant__
#include "poseidon_ktables.h"

GL_DEV void psd_split(uint64_t x, uint32_t (&l)[3]) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    l[0] = lo & 0x3FFFFFu;
    l[1] = __builtin_amdgcn_alignbit(hi, lo, 22) & 0x3FFFFFu;
    l[2] = hi >> 12;
}
// c: the three table words of this multiply-accumulate, wave-uniform (SGPRs: the multiplier of every multiply-add)
typedef const __attribute__((address_space(4))) uint64_t* psd_ktab;      // constant address space: wave-uniform reads become scalar loads
GL_DEV void psd_mac(uint64_t& lo, uint64_t& hi, const uint32_t (&l)[3], const uint64_t (&c)[3]) {
    const uint64_t c0 = c[0], c1 = c[1], c2 = c[2];
#if defined(__HIP_DEVICE_COMPILE__)
    // written out: in C the compiler keeps one zero-extended 64-bit copy of every limb (a register PAIR each, 132 VGPRs of limbs) and truncates
    // it at each use.  src0 = the constant (SGPR, the one constant-bus operand), src1 = the limb; the carry-out pair is never read.
    uint64_t unused;
    asm("v_mad_u64_u32 %0, %2, %3, %9, %0\n\tv_mad_u64_u32 %1, %2, %4, %9, %1\n\t"
        "v_mad_u64_u32 %0, %2, %5, %10, %0\n\tv_mad_u64_u32 %1, %2, %6, %10, %1\n\t"
        "v_mad_u64_u32 %0, %2, %7, %11, %0\n\tv_mad_u64_u32 %1, %2, %8, %11, %1"
        : "+v"(lo), "+v"(hi), "=&s"(unused)
        : "s"((uint32_t)c0), "s"((uint32_t)(c0 >> 32)), "s"((uint32_t)c1), "s"((uint32_t)(c1 >> 32)), "s"((uint32_t)c2), "s"((uint32_t)(c2 >> 32)),
          "v"(l[0]), "v"(l[1]), "v"(l[2]));
#else
    lo += (uint64_t)l[0] * (uint32_t)c0 + (uint64_t)l[1] * (uint32_t)c1 + (uint64_t)l[2] * (uint32_t)c2;
    hi += (uint64_t)l[0] * (uint32_t)(c0 >> 32) + (uint64_t)l[1] * (uint32_t)(c1 >> 32) + (uint64_t)l[2] * (uint32_t)(c2 >> 32);
#endif
}
// the first multiply-accumulate of an output: the accumulators start from the additive constant's halves.  Inside the statement, so that the
// constant is read where the group's other constants are (a C initialisation is copied to VGPRs right behind its scalar load: a wait of its own).
GL_DEV void psd_mac_first(uint64_t& lo, uint64_t& hi, const uint32_t (&l)[3], const uint64_t (&c)[3], const uint64_t (&add)[2]) {
    const uint64_t c0 = c[0], c1 = c[1], c2 = c[2];
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t unused;
    asm("v_mov_b64 %0, %12\n\tv_mov_b64 %1, %13\n\t"
        "v_mad_u64_u32 %0, %2, %3, %9, %0\n\tv_mad_u64_u32 %1, %2, %4, %9, %1\n\t"
        "v_mad_u64_u32 %0, %2, %5, %10, %0\n\tv_mad_u64_u32 %1, %2, %6, %10, %1\n\t"
        "v_mad_u64_u32 %0, %2, %7, %11, %0\n\tv_mad_u64_u32 %1, %2, %8, %11, %1"
        : "=&v"(lo), "=&v"(hi), "=&s"(unused)
        : "s"((uint32_t)c0), "s"((uint32_t)(c0 >> 32)), "s"((uint32_t)c1), "s"((uint32_t)(c1 >> 32)), "s"((uint32_t)c2), "s"((uint32_t)(c2 >> 32)),
          "v"(l[0]), "v"(l[1]), "v"(l[2]), "s"(add[0]), "s"(add[1]));
#else
    lo = add[0]; hi = add[1];
    psd_mac(lo, hi, l, c);
#endif
}
#ifndef PSD_K_GROUP
#define PSD_K_GROUP 5      // multiply-accumulates per group of scalar loads: 6 SGPRs of constants each, two groups in flight
#endif
// The block's table is consumed front to back.  Scalar loads return out of order, so the only wait is "all of them": a group's constants are
// therefore requested one group AHEAD -- the first multiply-accumulate of a group waits for the group's constants (requested while the previous
// group was being consumed), then the next group's loads are issued, then the remaining multiply-accumulates run under their latency.  The
// scheduling barriers pin that order; without them the scheduler either issues the loads of a whole dot product (132 SGPRs, spilled into VGPR
// lanes) or waits for each group right after requesting it (a lonely wave then sits through the scalar-cache latency 90 times per block).
struct PsdK {
    uint64_t w[PSD_K_GROUP][3];     // (c, 2^22 c, 2^44 c) of the group's multiply-accumulates
    uint64_t a[2];                  // when the group opens an output: the output's additive constant (low half, high half)
};
template <int N, bool WITH_ADD>
GL_DEV void psd_kload(PsdK& k, psd_ktab c, psd_ktab add) {
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) k.w[i][j] = c[3 * i + j];
    if constexpr (WITH_ADD) { k.a[0] = add[0]; k.a[1] = add[1]; }
}
// -> (lo, hi): lo + hi 2^32 = add + sum_{t < 11} ul[t] c[t] + sum_{t < NX} xl[t] c[11 + t] (the caller folds it: psd_recombine).  k: this output's first group (requested
// earlier); on return the NEXT output's first group (table position c + 3 (11 + NX), additive constant add_next), requested, if HAS_NEXT.
template <int NX, bool HAS_NEXT, int XCAP>
GL_DEV void psd_dot(const uint32_t (&ul)[11][3], const uint32_t (&xl)[XCAP][3], psd_ktab c, PsdK& k, psd_ktab add_next, uint64_t& lo, uint64_t& hi) {
    constexpr int G = PSD_K_GROUP, NT = 11 + NX, NG = (NT + G - 1) / G;
    PsdK cur = k;
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int n = NT - g * G < G ? NT - g * G : G;
        PsdK nxt;
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int t = g * G + i;
            if (t == 0) psd_mac_first(lo, hi, ul[0], cur.w[0], cur.a);
            else if (i < n) {
                if (t < 11) psd_mac(lo, hi, ul[t], cur.w[i]);
                else psd_mac(lo, hi, xl[t - 11], cur.w[i]);
            }
            if (i == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (g + 1 < NG) {
                    if (NT - (g + 1) * G >= G) psd_kload<G, false>(nxt, c + 3 * (g + 1) * G, add_next);
                    else psd_kload<(NT % G ? NT % G : G), false>(nxt, c + 3 * (g + 1) * G, add_next);
                } else if constexpr (HAS_NEXT) {
                    psd_kload<G, true>(nxt, c + 3 * NT, add_next);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < NG || HAS_NEXT) cur = nxt;          // (the last group of the last output requested nothing: nxt is not initialised then)
    }
    if constexpr (HAS_NEXT) k = cur;
}
template <int J>
GL_DEV void psd_block_lane0(const uint32_t (&ul)[11][3], uint32_t (&xl)[PSD_K_BLOCK][3], psd_ktab mac, psd_ktab add, PsdK& k) {
    if constexpr (J < PSD_K_BLOCK) {
        uint64_t lo, hi;
        psd_dot<J, true>(ul, xl, mac + 3 * (11 * (J - 1) + (J - 1) * J / 2), k, add + 2 * J, lo, hi);
        psd_split(psd_sbox(psd_recombine(lo, hi)), xl[J]);
        __builtin_amdgcn_sched_barrier(0);
        psd_block_lane0<J + 1>(ul, xl, mac, add, k);
    }
}
template <int R>
GL_DEV void psd_block_out(uint64_t (&s)[12], const uint32_t (&ul)[11][3], const uint32_t (&xl)[PSD_K_BLOCK][3], psd_ktab mac, psd_ktab add, PsdK& k) {
    constexpr int B = PSD_K_BLOCK;
    if constexpr (R < 12) {
        // (grouping four outgoing lanes for a lock-step recombination costs 12 live VGPRs: 102 instead of 96, a wave per SIMD less)
        uint64_t lo, hi;
        psd_dot<B, (R < 11)>(ul, xl, mac + 3 * (11 * (B - 1) + (B - 1) * B / 2 + (11 + B) * R), k, add + 2 * (B + R), lo, hi);
        s[R] = psd_recombine(lo, hi);
        __builtin_amdgcn_sched_barrier(0);
        psd_block_out<R + 1>(s, ul, xl, mac, add, k);
    }
}
GL_DEV void psd_partial_rounds_block(uint64_t (&s)[12]) {
    constexpr int B = PSD_K_BLOCK;
#pragma unroll 1
    for (int blk = 0; blk < 22 / B; blk++) {
        psd_ktab add = (psd_ktab)PSD_K_ADD + blk * 2 * (B - 1 + 12);
        psd_ktab mac = (psd_ktab)PSD_K_MAC;
        asm volatile("" : "+s"(mac));         // re-read per block: hoisted out of this loop the table would sit in ~2 600 SGPRs
        PsdK k;
        psd_kload<PSD_K_GROUP, true>(k, mac, add);        // y_1's first group, under the first S-box
        __builtin_amdgcn_sched_barrier(0);
        uint32_t ul[11][3], xl[B][3];
#pragma unroll
        for (int i = 0; i < 11; i++) psd_split(s[i + 1], ul[i]);
        psd_split(psd_sbox(s[0]), xl[0]);
        __builtin_amdgcn_sched_barrier(0);
        psd_block_lane0<1>(ul, xl, mac, add, k);
        psd_block_out<0>(s, ul, xl, mac, add, k);
    }
}

// 30 rounds of (constants, S-box on every element [4 + 4 full rounds] or on element 0 [22 partial rounds], MDS); round r's MDS adds round
// r+1's constants (PSD_ALL_RC row 30 is zero).  PSD_PARTIAL_FORM 1 (default): the partial rounds in the block form above; 0: the dense form
// for all 30 rounds (rounds 1-4: a v_mad_u64_u32 by a 6-bit MDS entry beat the 64x64 modular products of the round-by-round factored form,
// profiles/r01_poseidon_dense_vs_sparse.txt; the block form needs neither -- A/B in profiles/r05_poseidon_block_vs_dense.txt).
#ifndef PSD_PARTIAL_FORM
#define PSD_PARTIAL_FORM 1
#endif
#ifndef PSD_ROUNDS_PER_ITER
#define PSD_ROUNDS_PER_ITER 2
#endif
template <bool FULL>
GL_DEV void psd_dense_rounds(uint64_t (&s)[12], int r0, int r1) {
    // two rounds per iteration (4 + 22 + 4: a pair is never mixed), so the state ping-pongs between two register sets
    // instead of being copied back at the loop edge
#pragma unroll 1
    for (int r = r0; r < r1; r += PSD_ROUNDS_PER_ITER) {
#pragma unroll
        for (int h = 0; h < PSD_ROUNDS_PER_ITER; h++) {
            if (FULL) {
#pragma unroll
#if PSD_SBOX_WIDTH == 4
                for (int i = 0; i < 12; i += 4) psd_sbox4(s[i], s[i + 1], s[i + 2], s[i + 3]);
#else
                for (int i = 0; i < 12; i += 2) psd_sbox2(s[i], s[i + 1]);
#endif
            } else {
                s[0] = psd_sbox(s[0]);
            }
            psd_mds(s, &PSD_ALL_RC[12 * (r + h + 1)]);
        }
    }
}
GL_DEV void psd_permute(uint64_t (&s)[12]) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_canonical(s[i], PSD_ALL_RC[i]);
    psd_dense_rounds<true>(s, 0, 4);
#if PSD_PARTIAL_FORM == 1
    psd_partial_rounds_block(s);
#else
    psd_dense_rounds<false>(s, 4, 26);
#endif
    psd_dense_rounds<true>(s, 26, 30);
}

// digest of <= 4 elements is the elements themselves, zero padded (chip/merkle_proof_chip.rs:52-57);
// longer inputs go through the overwrite-mode sponge (chip/hasher_chip.rs:122-148).
struct PsdSponge {
    uint64_t s[12];
    GL_DEV void init() {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = 0;
    }
};

}  // namespace gl355
