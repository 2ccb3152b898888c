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
    uint32_t t1, top;
    asm("v_add_co_u32_e32 %0, vcc, %2, %3\n\t" GL_HAZARD_NOP "v_addc_co_u32_e32 %1, vcc, 0, %4, vcc"
        : "=&v"(t1), "=v"(top) : "v"((uint32_t)(al >> 32)), "v"((uint32_t)ah), "v"((uint32_t)(ah >> 32)) : "vcc");
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
GL_DEV void psd_mds(uint64_t (&s)[12], const uint64_t* rc) {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (uint32_t)s[i]; hi[i] = (uint32_t)(s[i] >> 32); }
    // the entries 16 and 2 are kept opaque (scalar registers): as literals the compiler turns those products into 64-bit
    // shift-adds, which need the 32-bit half zero-extended into a register pair first -- one v_mov more than the multiply-add
    uint32_t c16 = 16, c2 = 2;
    asm("" : "+s"(c16), "+s"(c2));
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t al = rc ? (uint64_t)(uint32_t)rc[r] : 0, ah = rc ? rc[r] >> 32 : 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const uint32_t c = CIRC[i] == 16 ? c16 : CIRC[i] == 2 ? c2 : CIRC[i] + ((r == 0 && i == 0) ? 8u : 0u);
            al += (uint64_t)lo[(i + r) % 12] * c;
            ah += (uint64_t)hi[(i + r) % 12] * c;
        }
        s[r] = psd_recombine(al, ah);
    }
}
GL_DEV void psd_mds(uint64_t (&s)[12]) { psd_mds(s, nullptr); }

// 30 rounds of (constants, S-box on every element [4 + 4 full rounds] or on element 0 [22 partial rounds], MDS).  The partial
// rounds run in the naive form as well: a v_mad_u64_u32 by a 6-bit MDS entry is far cheaper than the 64x64 modular products
// of the factored form (measured, profiles/r01_poseidon_dense_vs_sparse.txt).  Round r's MDS adds round r+1's constants
// (PSD_ALL_RC row 30 is zero).
GL_DEV void psd_permute(uint64_t (&s)[12]) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_canonical(s[i], PSD_ALL_RC[i]);
#pragma unroll 1
    for (int r = 0; r < 30; r++) {
        if (r < 4 || r >= 26) {
#pragma unroll
            for (int i = 0; i < 12; i++) s[i] = psd_sbox(s[i]);
        } else {
            s[0] = psd_sbox(s[0]);
        }
        psd_mds(s, &PSD_ALL_RC[12 * (r + 1)]);
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
