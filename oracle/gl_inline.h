/*
 * gl_inline.h -- inline Goldilocks helpers shared by the oracle's translation units
 * (gl_oracle.c, gl_prover.c).  TEST INFRASTRUCTURE ONLY (see gl_oracle.h).
 */
#ifndef GL_INLINE_H
#define GL_INLINE_H
#include <stddef.h>
#include <stdint.h>
#include "gl_oracle.h"

typedef unsigned __int128 u128;
#define P ORC_P
#define EPS UINT64_C(0xFFFFFFFF) /* 2^64 mod p */

/* a1 -- Goldilocks field, p = 2^64 - 2^32 + 1 (chip/native_chip/arithmetic_chip.rs:19).
 * Values are kept canonical everywhere in the oracle: simplest possible model. */
static inline uint64_t canon(uint64_t a) { return a - (P & (uint64_t)(-(int64_t)(a >= P))); }

static inline uint64_t f_add(uint64_t a, uint64_t b) {
    a = canon(a); b = canon(b);
    uint64_t s = a + b;
    if (s < a || s >= P) s -= P;
    return s;
}
static inline uint64_t f_sub(uint64_t a, uint64_t b) {
    a = canon(a); b = canon(b);
    return a >= b ? a - b : a + (P - b);
}
/* 2^64 = 2^32 - 1, 2^96 = -1 (mod p): x = lo + 2^64*hi_lo + 2^96*hi_hi = lo - hi_hi + EPS*hi_lo */
static inline uint64_t reduce128(u128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & EPS;
    uint64_t t0 = lo - hi_hi;
    t0 -= EPS & (uint64_t)(-(int64_t)(lo < hi_hi));        /* branch-free: data is random */
    uint64_t t1 = hi_lo * EPS;
    uint64_t r = t0 + t1;
    r += EPS & (uint64_t)(-(int64_t)(r < t1));
    return canon(r);
}
static inline uint64_t f_mul(uint64_t a, uint64_t b) { return reduce128((u128)a * b); }
static inline size_t bitrev(size_t x, uint32_t bits) {
    size_t r = 0;
    for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
static inline uint32_t log2_exact(size_t n) { uint32_t l = 0; while (((size_t)1 << l) < n) l++; return l; }
#endif
