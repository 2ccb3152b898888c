// Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic extension F_p[X]/(X^2 - 7) for gfx950.
//
// Replaces plonky2_field::goldilocks_field::GoldilocksField / QuadraticExtension, reached from the
// reference at src/plonky2_semaphore/signal.rs:1,5; modulus pinned at
// src/plonky2_verifier/chip/native_chip/arithmetic_chip.rs:19, non-residue W = 7 at :122-125.
//
// Representation: a field element is ANY uint64_t (values >= p are legal, "non-canonical"); every
// routine accepts any u64 and returns some u64 congruent to the result.  gl_canon() maps to [0, p)
// and is applied wherever a value leaves the device or is compared.
//
// gfx950 notes (tools/ubench/ubench_alu.hip, measured): v_mad_u64_u32 issues at half rate
// (~4.9 cyc/wave), a carry-chained add_co/addc pair costs ~9 cyc while the single v_lshl_add_u64
// costs ~4.6, so 64-bit adds are written as plain u64 arithmetic with compare-based carry detection.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GL_P 0xFFFFFFFF00000001ull
#define GL_EPS 0xFFFFFFFFull  // 2^64 mod p

#define GL_DEV __device__ __forceinline__
#define GL_HD __host__ __device__ __forceinline__

// data-dependent corrections fire for about half of all random operands: on the host (Challenger, witness tape) ask for
// conditional moves instead of branches; on the device the compiler predicates anyway
#if defined(__HIP_DEVICE_COMPILE__)
#define GL_UNPRED(c) (c)
#else
#define GL_UNPRED(c) __builtin_unpredictable(c)
#endif

GL_HD uint64_t gl_canon(uint64_t a) { return a >= GL_P ? a - GL_P : a; }

GL_HD uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (GL_UNPRED(s < a)) {      // true sum = s + 2^64 == s + EPS
        s += GL_EPS;
        if (GL_UNPRED(s < GL_EPS)) s += GL_EPS;
    }
    return s;
}

// a + c for a CANONICAL c (< p), e.g. a round constant: a + c < 2^64 + p, so after a wrapped sum the single
// correction s + EPS < p + EPS = 2^64 cannot wrap again -- one correction instead of two
GL_HD uint64_t gl_add_canonical(uint64_t a, uint64_t c) {
    uint64_t s = a + c;
    if (GL_UNPRED(s < a)) s += GL_EPS;
    return s;
}

GL_HD uint64_t gl_sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    if (GL_UNPRED(a < b)) {      // true diff = d - 2^64 == d - EPS
        uint64_t e = d - GL_EPS;
        if (GL_UNPRED(d < GL_EPS)) e -= GL_EPS;
        d = e;
    }
    return d;
}

GL_HD uint64_t gl_neg(uint64_t a) { return gl_sub(0, a); }

// ---- device multiplication / 128-bit reduction ----
// The compiler cannot be made to use the carry-out of v_mad_u64_u32 (it re-derives carries with 64-bit compares and
// selects every correction with two v_cndmask: 26 VALU instructions per product).  Two device formulations:
//   GL_MUL_VARIANT 1: the carry-out / borrow steps in inline asm: 15 VALU instructions, but gfx950 needs two wait states
//     between a VALU write of an SGPR/VCC and a VALU read of it, the hazard recogniser does not look inside inline asm, and
//     an asm block cannot be interleaved with its neighbours: 8 s_nop per product.  Best where many waves hide them (Poseidon).
//   GL_MUL_VARIANT 2: carries re-derived in C the way the compiler digests best: 22 VALU instructions, no forced s_nop (latency-bound
//     kernels with many independent products in flight: the NTT butterflies).
// A translation unit picks with -DGL_MUL_VARIANT / #define before this header; the host always uses the 128-bit C product.
#ifndef GL_MUL_VARIANT
#define GL_MUL_VARIANT 1
#endif
// two wait states between a VALU write of an SGPR / VCC and the VALU instruction that reads it (carry-in, select mask)
#ifdef GL_EXPERIMENT_NO_HAZARD_NOP   // timing experiments only: results are undefined without the wait states
#define GL_HAZARD_NOP ""
#else
#define GL_HAZARD_NOP "s_nop 1\n\t"
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#if GL_MUL_VARIANT == 1
// x = lo - sub32: a borrow is repaid with -EPS (the wrapped x is >= p then, so no second borrow)
GL_DEV uint64_t gl_dev_sub32(uint32_t lo0, uint32_t lo1, uint32_t sub32) {
    uint32_t x0, x1, m;
    asm("v_sub_co_u32_e32 %0, vcc, %3, %5\n\t" GL_HAZARD_NOP ""
        "v_subbrev_co_u32_e32 %1, vcc, 0, %4, vcc\n\t" GL_HAZARD_NOP ""
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t" GL_HAZARD_NOP ""
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "=&v"(x0), "=&v"(x1), "=&v"(m) : "v"(lo0), "v"(lo1), "v"(sub32) : "vcc");
    return ((uint64_t)x1 << 32) | x0;
}
// x + k * EPS for k < 2^32: one multiply-add, its carry-out repaid with +EPS (the wrapped sum is < (2^32-1)^2, no second carry)
GL_DEV uint64_t gl_dev_add_mul_eps(uint64_t x, uint32_t k) {
    uint64_t r, carry;
    uint32_t m;
    asm("v_mad_u64_u32 %0, %1, %3, -1, %4\n\t" GL_HAZARD_NOP "v_cndmask_b32_e64 %2, 0, -1, %1"
        : "=v"(r), "=s"(carry), "=v"(m) : "v"(k), "v"(x));
    return r + (uint64_t)m;
}
// (a1*b0 + u) >> 32 as a 33-bit value: the sum can overflow 64 bits once, its carry-out becomes bit 32
GL_DEV uint64_t gl_dev_mid(uint32_t a1, uint32_t b0, uint64_t u, uint32_t* v0) {
    uint64_t v, carry;
    uint32_t ch;
    asm("v_mad_u64_u32 %0, %1, %3, %4, %5\n\t" GL_HAZARD_NOP "v_cndmask_b32_e64 %2, 0, 1, %1"
        : "=v"(v), "=s"(carry), "=v"(ch) : "v"(a1), "v"(b0), "v"(u));
    *v0 = (uint32_t)v;
    return ((uint64_t)ch << 32) | (v >> 32);
}
#else
GL_DEV uint64_t gl_dev_sub32(uint32_t lo0, uint32_t lo1, uint32_t sub32) {
    unsigned long x, y;
    const bool b = __builtin_usubl_overflow(((unsigned long)lo1 << 32) | lo0, (unsigned long)sub32, &x);
    __builtin_usubl_overflow(x, (unsigned long)(b ? 0xFFFFFFFFu : 0u), &y);
    return y;
}
GL_DEV uint64_t gl_dev_add_mul_eps(uint64_t x, uint32_t k) {
    const uint64_t r = (uint64_t)k * 0xFFFFFFFFu + x;
    return r + (r < x ? 0xFFFFFFFFu : 0u);
}
GL_DEV uint64_t gl_dev_mid(uint32_t a1, uint32_t b0, uint64_t u, uint32_t* v0) {
    const uint64_t v = (uint64_t)a1 * b0 + u;
    *v0 = (uint32_t)v;
    return ((uint64_t)(v < u ? 1u : 0u) << 32) | (v >> 32);
}
#endif
#endif

// (lo, hi) = hi*2^64 + lo  ->  lo - hi_hi + hi_lo*(2^32 - 1)      [2^64 = 2^32-1, 2^96 = -1 mod p]
GL_HD uint64_t gl_reduce128(uint64_t lo, uint64_t hi) {
#if defined(__HIP_DEVICE_COMPILE__) && GL_MUL_VARIANT != 0
    return gl_dev_add_mul_eps(gl_dev_sub32((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)(hi >> 32)), (uint32_t)hi);
#else
    uint32_t hi_hi = (uint32_t)(hi >> 32), hi_lo = (uint32_t)hi;
    uint64_t t0 = lo - hi_hi;
    if (GL_UNPRED(lo < hi_hi)) t0 -= GL_EPS;
    uint64_t t1 = ((uint64_t)hi_lo << 32) - hi_lo;
    uint64_t r = t0 + t1;
    if (GL_UNPRED(r < t1)) r += GL_EPS;
    return r;
#endif
}

GL_HD uint64_t gl_mul(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && GL_MUL_VARIANT != 0
    // four 32x32+64 multiply-adds (v_mad_u64_u32): t and u cannot overflow, the third sum is taken with its carry
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t = (uint64_t)a0 * b0;
    const uint64_t u = (uint64_t)a0 * b1 + (t >> 32);
    uint32_t v0;
    const uint64_t mid = gl_dev_mid(a1, b0, u, &v0);
    const uint64_t w = (uint64_t)a1 * b1 + mid;
    return gl_dev_add_mul_eps(gl_dev_sub32((uint32_t)t, v0, (uint32_t)(w >> 32)), (uint32_t)w);
#elif defined(__HIP_DEVICE_COMPILE__)
    uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint64_t t = (uint64_t)a0 * b0;
    uint64_t u = (uint64_t)a0 * b1 + (t >> 32);
    uint64_t v = (uint64_t)a1 * b0 + (uint32_t)u;
    uint64_t w = (uint64_t)a1 * b1 + (u >> 32) + (v >> 32);
    uint64_t lo = (v << 32) | (uint32_t)t;
    return gl_reduce128(lo, w);
#else
    unsigned __int128 pr = (unsigned __int128)a * b;
    return gl_reduce128((uint64_t)pr, (uint64_t)(pr >> 64));
#endif
}

GL_HD uint64_t gl_sqr(uint64_t a) { return gl_mul(a, a); }

// N independent products in lock-step, stage by stage (round 5).  The five carry steps of a product each need two wait states between the VALU
// instruction that writes the carry (an SGPR pair) and the one that reads it; inside ONE product nothing else can go there (s_nop: 1.7 % of a
// full-occupancy hash kernel, 18 % of one at one wave per SIMD, profiles/r05_poseidon_block_vs_dense.txt).  With N products issued round-robin the
// partners' instructions ARE the wait states: every stage is its own asm statement (so the compiler still names the 32-bit halves of the 64-bit
// intermediates for free), pinned in this order by scheduling barriers; N = 2 needs one extra wait state in front of the first product's readers
// (its writer is one instruction back; the second product then has the first one's nop + instruction behind its writer), N >= 3 none.  Anything
// the compiler adds between two statements only adds wait states.  Each product keeps its carry in its own SGPR pair.
#if defined(__HIP_DEVICE_COMPILE__) && GL_MUL_VARIANT == 1
template <int N>
GL_DEV void gl_mul_multi(const uint64_t (&a)[N], const uint64_t (&b)[N], uint64_t (&r)[N]) {
    static_assert(N >= 1 && N <= 4, "1 .. 4 products");
#define GL_MM_SB() __builtin_amdgcn_sched_barrier(0)
    // the nop in front of a carry READER of product j
#define GL_MM_NOP(j) (N == 1 ? "s_nop 1\n\t" : (N == 2 && (j) == 0 ? "s_nop 0\n\t" : ""))
    uint32_t a0[N], a1[N], b0[N], b1[N], ch[N], x0[N], x1[N], m[N];
    uint64_t t[N], u[N], v[N], w[N], c[N];
#pragma unroll
    for (int j = 0; j < N; j++) { a0[j] = (uint32_t)a[j]; a1[j] = (uint32_t)(a[j] >> 32); b0[j] = (uint32_t)b[j]; b1[j] = (uint32_t)(b[j] >> 32); }
#pragma unroll
    for (int j = 0; j < N; j++) { t[j] = (uint64_t)a0[j] * b0[j]; u[j] = (uint64_t)a0[j] * b1[j] + (t[j] >> 32); }
    GL_MM_SB();
#pragma unroll
    for (int j = 0; j < N; j++) { asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(v[j]), "=s"(c[j]) : "v"(a1[j]), "v"(b0[j]), "v"(u[j])); GL_MM_SB(); }
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (N == 1) asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(ch[j]) : "s"(c[j]));
        else if (N == 2 && j == 0) asm("s_nop 0\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(ch[j]) : "s"(c[j]));
        else asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(ch[j]) : "s"(c[j]));
        GL_MM_SB();
    }
#pragma unroll
    for (int j = 0; j < N; j++) w[j] = (uint64_t)a1[j] * b1[j] + (((uint64_t)ch[j] << 32) | (v[j] >> 32));
    GL_MM_SB();
    // x = (t_lo, v_lo) - w_hi, a borrow repaid with -EPS
#pragma unroll
    for (int j = 0; j < N; j++) { asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(x0[j]), "=s"(c[j]) : "v"((uint32_t)t[j]), "v"((uint32_t)(w[j] >> 32))); GL_MM_SB(); }
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (N == 1) asm("s_nop 1\n\tv_subb_co_u32_e64 %0, %1, %2, 0, %1" : "=v"(x1[j]), "+s"(c[j]) : "v"((uint32_t)v[j]));
        else if (N == 2 && j == 0) asm("s_nop 0\n\tv_subb_co_u32_e64 %0, %1, %2, 0, %1" : "=v"(x1[j]), "+s"(c[j]) : "v"((uint32_t)v[j]));
        else asm("v_subb_co_u32_e64 %0, %1, %2, 0, %1" : "=v"(x1[j]), "+s"(c[j]) : "v"((uint32_t)v[j]));
        GL_MM_SB();
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (N == 1) asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(m[j]) : "s"(c[j]));
        else if (N == 2 && j == 0) asm("s_nop 0\n\tv_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(m[j]) : "s"(c[j]));
        else asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(m[j]) : "s"(c[j]));
        GL_MM_SB();
    }
#pragma unroll
    for (int j = 0; j < N; j++) { asm("v_sub_co_u32_e64 %0, %1, %0, %2" : "+v"(x0[j]), "=s"(c[j]) : "v"(m[j])); GL_MM_SB(); }
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (N == 1) asm("s_nop 1\n\tv_subb_co_u32_e64 %0, %1, %0, 0, %1" : "+v"(x1[j]), "+s"(c[j]));
        else if (N == 2 && j == 0) asm("s_nop 0\n\tv_subb_co_u32_e64 %0, %1, %0, 0, %1" : "+v"(x1[j]), "+s"(c[j]));
        else asm("v_subb_co_u32_e64 %0, %1, %0, 0, %1" : "+v"(x1[j]), "+s"(c[j]));
        GL_MM_SB();
    }
    // + w_lo * EPS, the carry-out repaid with +EPS.  The repayment is a second multiply-add (carry in {0, 1} times EPS): as a 64-bit add every product
    // in flight needs a zero-extended register pair of its own for the addend, i.e. a register move each
#pragma unroll
    for (int j = 0; j < N; j++) {
        const uint64_t x = ((uint64_t)x1[j] << 32) | x0[j];
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(r[j]), "=s"(c[j]) : "v"((uint32_t)w[j]), "v"(x));
        GL_MM_SB();
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        if (N == 1) asm("s_nop 1\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(m[j]) : "s"(c[j]));
        else if (N == 2 && j == 0) asm("s_nop 0\n\tv_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(m[j]) : "s"(c[j]));
        else asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(m[j]) : "s"(c[j]));
        GL_MM_SB();
    }
#pragma unroll
    for (int j = 0; j < N; j++) asm("v_mad_u64_u32 %0, %1, %2, -1, %0" : "+v"(r[j]), "=s"(c[j]) : "v"(m[j]));
#undef GL_MM_NOP
#undef GL_MM_SB
}
// ONE product (and a lock-step PAIR) whose carry wait states are filled by the CALLER's instructions (round 6): fill(k), k = 0 .. 4, is invoked between the
// VALU instruction that writes a carry pair and the one that reads it and has to issue at least two (pair form: one) VALU instructions of its own.  For a lone
// dependent chain with free issue slots -- the 16-lane permutation's partial rounds, whose S-box is on the critical path while the MDS row's multiply-adds
// are independent of it -- this hides the fillers behind the chain instead of queueing them in front of it.  Same arithmetic as gl_mul_multi<1> / <2>;
// tools/check_hazards.py checks the distances on the emitted ISA.
template <class F>
GL_DEV uint64_t gl_mul_fill(uint64_t a, uint64_t b, F&& fill) {
#define GL_MF_SB() __builtin_amdgcn_sched_barrier(0)
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t = (uint64_t)a0 * b0;
    const uint64_t u = (uint64_t)a0 * b1 + (t >> 32);
    uint64_t v, c, r;
    uint32_t ch, x0, x1, m;
    GL_MF_SB();
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(v), "=s"(c) : "v"(a1), "v"(b0), "v"(u)); GL_MF_SB();
    fill(0); GL_MF_SB();
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(ch) : "s"(c)); GL_MF_SB();
    const uint64_t w = (uint64_t)a1 * b1 + (((uint64_t)ch << 32) | (v >> 32)); GL_MF_SB();
    asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(x0), "=s"(c) : "v"((uint32_t)t), "v"((uint32_t)(w >> 32))); GL_MF_SB();
    fill(1); GL_MF_SB();
    asm("v_subb_co_u32_e64 %0, %1, %2, 0, %1" : "=v"(x1), "+s"(c) : "v"((uint32_t)v)); GL_MF_SB();
    fill(2); GL_MF_SB();
    asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(m) : "s"(c)); GL_MF_SB();
    asm("v_sub_co_u32_e64 %0, %1, %0, %2" : "+v"(x0), "=s"(c) : "v"(m)); GL_MF_SB();
    fill(3); GL_MF_SB();
    asm("v_subb_co_u32_e64 %0, %1, %0, 0, %1" : "+v"(x1), "+s"(c)); GL_MF_SB();
    const uint64_t x = ((uint64_t)x1 << 32) | x0;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(r), "=s"(c) : "v"((uint32_t)w), "v"(x)); GL_MF_SB();
    fill(4); GL_MF_SB();
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(m) : "s"(c)); GL_MF_SB();
    asm("v_mad_u64_u32 %0, %1, %2, -1, %0" : "+v"(r), "=s"(c) : "v"(m));
    return r;
}
template <class F>
GL_DEV void gl_mul2_fill(const uint64_t (&a)[2], const uint64_t (&b)[2], uint64_t (&r)[2], F&& fill) {
    uint32_t a0[2], a1[2], b0[2], b1[2], ch[2], x0[2], x1[2], m[2];
    uint64_t t[2], u[2], v[2], w[2], c[2];
#pragma unroll
    for (int j = 0; j < 2; j++) { a0[j] = (uint32_t)a[j]; a1[j] = (uint32_t)(a[j] >> 32); b0[j] = (uint32_t)b[j]; b1[j] = (uint32_t)(b[j] >> 32); }
#pragma unroll
    for (int j = 0; j < 2; j++) { t[j] = (uint64_t)a0[j] * b0[j]; u[j] = (uint64_t)a0[j] * b1[j] + (t[j] >> 32); }
    GL_MF_SB();
#pragma unroll
    for (int j = 0; j < 2; j++) { asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(v[j]), "=s"(c[j]) : "v"(a1[j]), "v"(b0[j]), "v"(u[j])); GL_MF_SB(); }
    fill(0); GL_MF_SB();
#pragma unroll
    for (int j = 0; j < 2; j++) { asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(ch[j]) : "s"(c[j])); GL_MF_SB(); }
#pragma unroll
    for (int j = 0; j < 2; j++) w[j] = (uint64_t)a1[j] * b1[j] + (((uint64_t)ch[j] << 32) | (v[j] >> 32));
    GL_MF_SB();
#pragma unroll
    for (int j = 0; j < 2; j++) { asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(x0[j]), "=s"(c[j]) : "v"((uint32_t)t[j]), "v"((uint32_t)(w[j] >> 32))); GL_MF_SB(); }
    fill(1); GL_MF_SB();
#pragma unroll
    for (int j = 0; j < 2; j++) { asm("v_subb_co_u32_e64 %0, %1, %2, 0, %1" : "=v"(x1[j]), "+s"(c[j]) : "v"((uint32_t)v[j])); GL_MF_SB(); }
    fill(2); GL_MF_SB();
#pragma unroll
    for (int j = 0; j < 2; j++) { asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(m[j]) : "s"(c[j])); GL_MF_SB(); }
#pragma unroll
    for (int j = 0; j < 2; j++) { asm("v_sub_co_u32_e64 %0, %1, %0, %2" : "+v"(x0[j]), "=s"(c[j]) : "v"(m[j])); GL_MF_SB(); }
    fill(3); GL_MF_SB();
#pragma unroll
    for (int j = 0; j < 2; j++) { asm("v_subb_co_u32_e64 %0, %1, %0, 0, %1" : "+v"(x1[j]), "+s"(c[j])); GL_MF_SB(); }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const uint64_t x = ((uint64_t)x1[j] << 32) | x0[j];
        asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(r[j]), "=s"(c[j]) : "v"((uint32_t)w[j]), "v"(x));
        GL_MF_SB();
    }
    fill(4); GL_MF_SB();
#pragma unroll
    for (int j = 0; j < 2; j++) { asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(m[j]) : "s"(c[j])); GL_MF_SB(); }
#pragma unroll
    for (int j = 0; j < 2; j++) asm("v_mad_u64_u32 %0, %1, %2, -1, %0" : "+v"(r[j]), "=s"(c[j]) : "v"(m[j]));
#undef GL_MF_SB
}
#else
// host pass / compiler-scheduled product: one after the other
template <int N>
GL_HD void gl_mul_multi(const uint64_t (&a)[N], const uint64_t (&b)[N], uint64_t (&r)[N]) {
    for (int j = 0; j < N; j++) r[j] = gl_mul(a[j], b[j]);
}
#endif


// a * c for a small constant c < 2^32 (MDS entries, W = 7, ...): two mads instead of four
GL_HD uint64_t gl_mul_small(uint64_t a, uint32_t c) {
    uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32);
    uint64_t t = (uint64_t)a0 * c;
    uint64_t u = (uint64_t)a1 * c + (t >> 32);   // < 2^64
    uint64_t lo = (u << 32) | (uint32_t)t;
    uint32_t hi = (uint32_t)(u >> 32);           // product = hi*2^64 + lo, hi < 2^32
#if defined(__HIP_DEVICE_COMPILE__) && GL_MUL_VARIANT != 0
    return gl_dev_add_mul_eps(lo, hi);
#endif
    uint64_t t1 = ((uint64_t)hi << 32) - hi;
    uint64_t r = lo + t1;
    if (GL_UNPRED(r < t1)) r += GL_EPS;
    return r;
}

// a * 2^S mod p for a compile-time 0 <= S < 96: powers of two are roots of unity in Goldilocks
// (2^96 = -1), so the radix-16 butterflies' internal twiddles cost shifts instead of a 64x64 multiply.
template <int S>
GL_HD uint64_t gl_mul_2exp(uint64_t x) {
    static_assert(S >= 0 && S < 96, "shift out of range");
    if constexpr (S == 0) {
        return x;
    } else if constexpr (S <= 32) {
        const uint32_t hi = (uint32_t)(x >> (64 - S));       // < 2^S <= 2^32
        const uint64_t lo = x << S;
#if defined(__HIP_DEVICE_COMPILE__) && GL_MUL_VARIANT != 0
        return gl_dev_add_mul_eps(lo, hi);
#endif
        const uint64_t t = (uint64_t)hi * 0xFFFFFFFFu;        // hi * 2^64 == hi * EPS
        uint64_t r = lo + t;
        if (r < t) r += GL_EPS;
        return r;
    } else if constexpr (S < 64) {
        return gl_reduce128(x << S, x >> (64 - S));
    } else {
        const uint64_t y = gl_mul_2exp<S - 64>(x);            // then * 2^64
        return gl_reduce128(0, y);
    }
}

GL_HD uint64_t gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, a);
        a = gl_mul(a, a);
        e >>= 1;
    }
    return r;
}
GL_HD uint64_t gl_inv(uint64_t a) { return gl_pow(a, GL_P - 2); }

// omega_N = 7^((p-1)/N): src/plonky2_verifier/chip/fri_chip.rs:162-163
GL_HD uint64_t gl_root_of_unity(uint32_t log_n) { return gl_pow(7, (GL_P - 1) >> log_n); }

// ---- quadratic extension, X^2 = 7 (arithmetic_chip.rs:109-132) -------------------------------
struct gl2 {
    uint64_t c0, c1;
};
GL_HD gl2 gl2_make(uint64_t a, uint64_t b) { gl2 r; r.c0 = a; r.c1 = b; return r; }
GL_HD gl2 gl2_add(gl2 a, gl2 b) { return gl2_make(gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)); }
GL_HD gl2 gl2_sub(gl2 a, gl2 b) { return gl2_make(gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)); }
GL_HD gl2 gl2_mul(gl2 a, gl2 b) {
#if defined(__HIP_DEVICE_COMPILE__) && GL_MUL_VARIANT == 1 && !defined(GL2_MUL_SEQUENTIAL)
    // the four base-field products in lock-step: they fill each other's carry wait states (gl_mul_multi)
    const uint64_t x[4] = {a.c0, a.c1, a.c0, a.c1}, y[4] = {b.c0, b.c1, b.c1, b.c0};
    uint64_t p[4];
    gl_mul_multi<4>(x, y, p);
    return gl2_make(gl_add(p[0], gl_mul_small(p[1], 7)), gl_add(p[2], p[3]));
#else
    uint64_t c0 = gl_add(gl_mul(a.c0, b.c0), gl_mul_small(gl_mul(a.c1, b.c1), 7));
    uint64_t c1 = gl_add(gl_mul(a.c0, b.c1), gl_mul(a.c1, b.c0));
    return gl2_make(c0, c1);
#endif
}
GL_HD gl2 gl2_mul_base(gl2 a, uint64_t b) { return gl2_make(gl_mul(a.c0, b), gl_mul(a.c1, b)); }
GL_HD gl2 gl2_canon(gl2 a) { return gl2_make(gl_canon(a.c0), gl_canon(a.c1)); }
GL_HD gl2 gl2_inv(gl2 a) {
    uint64_t norm = gl_sub(gl_mul(a.c0, a.c0), gl_mul_small(gl_mul(a.c1, a.c1), 7));
    uint64_t ni = gl_inv(norm);
    return gl2_make(gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni));
}
GL_HD gl2 gl2_pow(gl2 a, uint64_t e) {
    gl2 r = gl2_make(1, 0);
    while (e) {
        if (e & 1) r = gl2_mul(r, a);
        a = gl2_mul(a, a);
        e >>= 1;
    }
    return r;
}

// sum over k = t (mod 256), k < n, of p[k] z^k for base-field coefficients p and an extension point z: the share of lane t of a
// 256-lane workgroup in p(z).  Lane t strides through the coefficients, so the 64 lanes of a wave read 64 CONSECUTIVE words per step
// (one 512-byte segment) -- with a contiguous chunk per lane every wave load touched 64 cache lines (round 2: 6x the algorithmic
// bytes through the fabric).  Two interleaved Horner chains in z^512 give the lane some ILP.
GL_HD gl2 gl2_horner_strided256(const uint64_t* __restrict__ p, uint64_t n, gl2 z, uint32_t t) {
    if (t >= n) return gl2_make(0, 0);
    gl2 z256 = z;
    for (int i = 0; i < 8; i++) z256 = gl2_mul(z256, z256);
    const gl2 z512 = gl2_mul(z256, z256);
    uint64_t j = (n - t + 255) >> 8;                 // terms of this lane
    gl2 a0 = gl2_make(0, 0), a1 = gl2_make(0, 0);
    if (j & 1) { j--; a0.c0 = p[t + (j << 8)]; }     // odd count: the top term has an even index
    while (j) {
        j -= 2;
        a0 = gl2_mul(a0, z512); a0.c0 = gl_add(a0.c0, p[t + (j << 8)]);
        a1 = gl2_mul(a1, z512); a1.c0 = gl_add(a1.c0, p[t + ((j + 1) << 8)]);
    }
    return gl2_mul(gl2_add(a0, gl2_mul(a1, z256)), gl2_pow(z, t));
}
