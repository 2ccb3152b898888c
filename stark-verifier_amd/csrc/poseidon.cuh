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

// Full MDS layer.  Row r: sum_i s[(i+r)%12] * CIRC[i] + 8*s[0] (r == 0).  The entries are < 64, so
// the 32-bit halves are accumulated separately in u64 (12 * 41 * 2^32 < 2^42: no overflow) with
// v_mad_u64_u32 and recombined once per row.
GL_DEV void psd_mds(uint64_t (&s)[12]) {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { lo[i] = (uint32_t)s[i]; hi[i] = (uint32_t)(s[i] >> 32); }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const uint32_t c = CIRC[i] + ((r == 0 && i == 0) ? 8u : 0u);
            al += (uint64_t)lo[(i + r) % 12] * c;
            ah += (uint64_t)hi[(i + r) % 12] * c;
        }
        // value = al + ah * 2^32, ah < 2^42: split ah*2^32 = (ah >> 32) * 2^64 + (ah << 32)
        const uint64_t mid = ah << 32;
        const uint32_t top = (uint32_t)(ah >> 32);
        uint64_t r0 = al + mid;
        uint64_t carry = r0 < mid ? 1u : 0u;
        const uint64_t t = (uint64_t)(top + carry) * GL_EPS;  // (top + carry) * (2^32 - 1) < 2^43
        uint64_t r1 = r0 + t;
        if (r1 < t) r1 += GL_EPS;
        s[r] = r1;
    }
}

template <bool FIRST_HALF>
GL_DEV void psd_full_rounds(uint64_t (&s)[12]) {
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        const int base = (FIRST_HALF ? 0 : 48) + 12 * r;
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = psd_sbox(gl_add_canonical(s[i], PSD_FULL_RC[base + i]));
        psd_mds(s);
    }
}

GL_DEV void psd_partial_rounds(uint64_t (&s)[12]) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_canonical(s[i], PSD_PART_FIRST[i]);
    // pre-matrix: out[c] = sum_{r>=1} INIT[r-1][c-1] * s[r]  (lane 0 passes through)
    {
        uint64_t t[12];
        t[0] = s[0];
#pragma unroll
        for (int c = 1; c < 12; c++) t[c] = 0;
#pragma unroll 1
        for (int r = 1; r < 12; r++) {
            const uint64_t sr = s[r];
#pragma unroll
            for (int c = 1; c < 12; c++) t[c] = gl_add(t[c], gl_mul(sr, PSD_PART_INIT[(r - 1) * 11 + (c - 1)]));
        }
#pragma unroll
        for (int c = 0; c < 12; c++) s[c] = t[c];
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        uint64_t s0 = psd_sbox(s[0]);
        s0 = gl_add_canonical(s0, PSD_PART_RC[r]);  // entry 21 is 0
        uint64_t d = gl_mul_small(s0, 25);  // MDS[0][0] = CIRC[0] + DIAG[0]
#pragma unroll
        for (int i = 1; i < 12; i++) {
            d = gl_add(d, gl_mul(s[i], PSD_PART_WHAT[r * 11 + (i - 1)]));
            s[i] = gl_add(s[i], gl_mul(s0, PSD_PART_VS[r * 11 + (i - 1)]));
        }
        s[0] = d;
    }
}

// partial rounds in the naive form (12 constants, lane-0 S-box, dense small-constant MDS): no 64-bit
// constant multiplications at all; which form is faster is a measured choice (tools/kbench.py poseidon)
GL_DEV void psd_partial_rounds_dense(uint64_t (&s)[12]) {
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl_add_canonical(s[i], PSD_ALL_RC[12 * (4 + r) + i]);
        s[0] = psd_sbox(s[0]);
        psd_mds(s);
    }
}

// measured on MI355X (profiles/r01_poseidon_dense_vs_sparse.txt): dense partial rounds 1.56 G perm/s vs
// 1.34 G perm/s for the sparse form -- v_mad_u64_u32 by a 6-bit constant is far cheaper than a 64x64 modmul
#ifndef PSD_DENSE_PARTIAL
#define PSD_DENSE_PARTIAL 1
#endif
GL_DEV void psd_permute(uint64_t (&s)[12]) {
    psd_full_rounds<true>(s);
    if (PSD_DENSE_PARTIAL) psd_partial_rounds_dense(s);
    else psd_partial_rounds(s);
    psd_full_rounds<false>(s);
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
