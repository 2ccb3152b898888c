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

GL_DEV uint64_t psd_sbox(uint64_t x) {
    uint64_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
}

// al + ah * 2^32 for al, ah < 2^44 (the two accumulators of an MDS row)  ->  u64 representative.
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
    const uint64_t t = (uint64_t)(top + carry) * GL_EPS;  // (top + carry) * (2^32 - 1) < 2^44
    uint64_t r1 = r0 + t;
    if (r1 < t) r1 += GL_EPS;
    return r1;
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
        const uint64_t c = rc[R];
        s[R] = psd_recombine(psd_mds_chain<R>(lo, (uint32_t)c), psd_mds_chain<R>(hi, c >> 32));
        psd_mds_rows<R + 1>(s, lo, hi, rc);
    }
}
// rc: 12 constants at a wave-uniform address (scalar loads)
GL_DEV void psd_mds(uint64_t (&s)[12], const uint64_t* rc) {
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (uint32_t)s[i]; hi[i] = (uint32_t)(s[i] >> 32); }
    psd_mds_rows<0>(s, lo, hi, rc);
}


// 30 rounds of (constants, S-box on every element [4 + 4 full rounds] or on element 0 [22 partial rounds], MDS).  The partial
// rounds run in the naive form as well: a v_mad_u64_u32 by a 6-bit MDS entry is far cheaper than the 64x64 modular products
// of the factored form (measured, profiles/r01_poseidon_dense_vs_sparse.txt).  Round r's MDS adds round r+1's constants
// (PSD_ALL_RC row 30 is zero).
GL_DEV void psd_permute(uint64_t (&s)[12]) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_canonical(s[i], PSD_ALL_RC[i]);
    // two rounds per iteration (4 + 22 + 4: a pair is never mixed), so the state ping-pongs between two register sets
    // instead of being copied back at the loop edge
#ifndef PSD_ROUNDS_PER_ITER
#define PSD_ROUNDS_PER_ITER 2
#endif
#pragma unroll 1
    for (int r = 0; r < 30; r += PSD_ROUNDS_PER_ITER) {
        const bool full = r < 4 || r >= 26;
#pragma unroll
        for (int h = 0; h < PSD_ROUNDS_PER_ITER; h++) {
            if (full) {
#pragma unroll
                for (int i = 0; i < 12; i++) s[i] = psd_sbox(s[i]);
            } else {
                s[0] = psd_sbox(s[0]);
            }
            psd_mds(s, &PSD_ALL_RC[12 * (r + h + 1)]);
        }
    }
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
