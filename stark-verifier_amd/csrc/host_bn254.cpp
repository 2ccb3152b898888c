// Host-side (CPU, sequential) BN254-Poseidon permutation of the 12-element sponge state: what the Fiat-Shamir
// Challenger and the circuit digest need when the proof's hasher is the reference's Bn254PoseidonHash
// (src/plonky2_verifier/bn245_poseidon/plonky2_config.rs:38-75, native.rs:16-77).  About 50 permutations per proof, so it
// runs on the calling thread like the Poseidon-Goldilocks one in host_transcript.cpp; the data-parallel work is in
// merkle_bn254.hip.  4 x 64-bit limbs with unsigned __int128, Montgomery form (tables: bn254_tables.h, 32-bit limbs).
#include "gl355_internal.h"

#define BN254_TABLE_QUAL static const
#include "bn254_tables.h"

namespace gl355 {
namespace {

typedef unsigned __int128 u128;
struct Fr { uint64_t l[4]; };

inline Fr from32(const uint32_t* p) {
    Fr r;
    for (int i = 0; i < 4; i++) r.l[i] = (uint64_t)p[2 * i] | ((uint64_t)p[2 * i + 1] << 32);
    return r;
}
const Fr MOD = from32(BN254_MOD);
const uint64_t N0INV64 = [] {            // -r^-1 mod 2^64 by Newton iteration from the 32-bit value
    uint64_t m0 = MOD.l[0], inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - m0 * inv;
    return (uint64_t)(0 - inv);
}();

bool geq_mod(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > MOD.l[i]) return true;
        if (a[i] < MOD.l[i]) return false;
    }
    return true;
}
void sub_mod(uint64_t a[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        const u128 d = (u128)a[i] - MOD.l[i] - (uint64_t)br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
Fr add(Fr a, Fr b) {
    Fr r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || geq_mod(r.l)) sub_mod(r.l);
    return r;
}
Fr mul(Fr a, Fr b) {   // a b 2^-256 mod r, canonical result
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * N0INV64;
        c = ((u128)m * MOD.l[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * MOD.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fr r;
    memcpy(r.l, t, 32);
    if (t[4] || geq_mod(r.l)) sub_mod(r.l);
    return r;
}
Fr pow5(Fr a) { const Fr a2 = mul(a, a), a4 = mul(a2, a2); return mul(a4, a); }

Fr encode3(const uint64_t* x) {          // x0 + x1 p + x2 p^2 (< 2^192), into Montgomery form
    uint64_t a[4] = {gl_canon(x[2]), 0, 0, 0};
    for (int step = 1; step >= 0; step--) {
        u128 c = gl_canon(x[step]);
        for (int i = 0; i < 4; i++) { c += (u128)a[i] * GL_P; a[i] = (uint64_t)c; c >>= 64; }
    }
    Fr r;
    memcpy(r.l, a, 32);
    return mul(r, from32(BN254_R2));
}
void decode3(Fr xm, uint64_t* out) {      // the three low base-p digits of the canonical value
    const Fr one = {{1, 0, 0, 0}};
    Fr x = mul(xm, one);
    uint64_t a[4];
    memcpy(a, x.l, 32);
    for (int d = 0; d < 3; d++) {
        u128 rem = 0;
        for (int i = 3; i >= 0; i--) {
            const u128 cur = (rem << 64) | a[i];
            a[i] = (uint64_t)(cur / GL_P);
            rem = cur % GL_P;
        }
        out[d] = (uint64_t)rem;
    }
}

}  // namespace

void host_bn254_permute(uint64_t s[12]) {
    static const struct Tables {
        Fr rc[340], mds[25];
        Tables() {
            for (int i = 0; i < 340; i++) rc[i] = from32(BN254_RC[i]);
            for (int i = 0; i < 25; i++) mds[i] = from32(BN254_MDS[i]);
        }
    } T;   // C++11 magic static: initialised once, thread-safe
    Fr st[5];
    for (int i = 0; i < 4; i++) st[i] = encode3(s + 3 * i);
    memset(&st[4], 0, sizeof(Fr));
    int k = 0;
    for (int rnd = 0; rnd < 68; rnd++) {
        for (int i = 0; i < 5; i++) st[i] = add(st[i], T.rc[k++]);
        if (rnd < 4 || rnd >= 64) { for (int i = 0; i < 5; i++) st[i] = pow5(st[i]); }
        else st[0] = pow5(st[0]);
        Fr n[5];
        for (int i = 0; i < 5; i++) {
            Fr acc = mul(st[0], T.mds[5 * i]);
            for (int j = 1; j < 5; j++) acc = add(acc, mul(st[j], T.mds[5 * i + j]));
            n[i] = acc;
        }
        memcpy(st, n, sizeof n);
    }
    for (int i = 0; i < 4; i++) decode3(st[i], s + 3 * i);
}

}  // namespace gl355
