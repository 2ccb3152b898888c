// Host end of the bn256::G1 multi-scalar multiplication (halo2 `best_multiexp`, reached from the reference at
// src/plonky2_verifier/verifier_api.rs:77-92): the combination of the per-window sums
//     result = sum_w 2^(c w) W_w          (Horner from the top window: 254 dependent doublings)
// and the conversion to an affine point.  It is strictly sequential -- 270 group operations of ~10 k VALU instructions each are
// ~4 ms on one GPU lane (what the first version did) and ~0.1 ms here on one host core with 64-bit limbs.
// Fq arithmetic: 4 x 64-bit limbs, Montgomery form with R = 2^256 (the representation the kernels store), CIOS on unsigned __int128.
#include <stdint.h>
#include <string.h>

#include "bn254_curve_tables.h"

namespace gl355 {
namespace {
typedef unsigned __int128 u128;
struct Fq { uint64_t l[4]; };
const uint64_t* Q = BN254C_FQ_MOD_64;

bool geq_q(const Fq& a) {
    for (int i = 3; i >= 0; i--) { if (a.l[i] > Q[i]) return true; if (a.l[i] < Q[i]) return false; }
    return true;
}
void sub_q(Fq& a) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) { const u128 d = (u128)a.l[i] - Q[i] - (uint64_t)br; a.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
Fq fq_canon(Fq a) { while (geq_q(a)) sub_q(a); return a; }      // the kernels keep values lazily below 2q
Fq fq_add(const Fq& a, const Fq& b) {
    Fq r; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || geq_q(r)) sub_q(r);          // q < 2^254: the sum of two canonical values never carries out of 256 bits
    return r;
}
Fq fq_sub(const Fq& a, const Fq& b) {
    Fq r; u128 br = 0;
    for (int i = 0; i < 4; i++) { const u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + Q[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
Fq fq_mul(const Fq& a, const Fq& b) {       // a b R^-1 mod q, canonical operands and result
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * BN254C_FQ_N0INV_64;
        c = ((u128)m * Q[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * Q[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fq r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_q(r)) sub_q(r);
    return r;
}
bool fq_is_zero(const Fq& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
Fq fq_one() { Fq r; memcpy(r.l, BN254C_FQ_ONE_64, 32); return r; }
Fq fq_inv(const Fq& a) {                    // a^(q-2)
    uint64_t e[4] = {Q[0] - 2, Q[1], Q[2], Q[3]};
    Fq r = fq_one();
    for (int i = 255; i >= 0; i--) {
        r = fq_mul(r, r);
        if ((e[i >> 6] >> (i & 63)) & 1) r = fq_mul(r, a);
    }
    return r;
}
struct Jac { Fq x, y, z; };
bool j_is_identity(const Jac& p) { return fq_is_zero(p.z); }
Jac j_double(const Jac& p) {                // y^2 = x^3 + 3 (a = 0)
    if (j_is_identity(p)) return p;
    const Fq a = fq_mul(p.x, p.x), b = fq_mul(p.y, p.y), c = fq_mul(b, b);
    const Fq xb = fq_add(p.x, b);
    Fq d = fq_sub(fq_sub(fq_mul(xb, xb), a), c);
    d = fq_add(d, d);
    const Fq e = fq_add(fq_add(a, a), a), f = fq_mul(e, e);
    Jac r;
    r.x = fq_sub(f, fq_add(d, d));
    Fq c8 = fq_add(c, c); c8 = fq_add(c8, c8); c8 = fq_add(c8, c8);
    r.y = fq_sub(fq_mul(e, fq_sub(d, r.x)), c8);
    const Fq yz = fq_mul(p.y, p.z);
    r.z = fq_add(yz, yz);
    return r;
}
Jac j_add(const Jac& p, const Jac& q) {
    if (j_is_identity(p)) return q;
    if (j_is_identity(q)) return p;
    const Fq z1z1 = fq_mul(p.z, p.z), z2z2 = fq_mul(q.z, q.z);
    const Fq u1 = fq_mul(p.x, z2z2), u2 = fq_mul(q.x, z1z1);
    const Fq s1 = fq_mul(fq_mul(p.y, q.z), z2z2), s2 = fq_mul(fq_mul(q.y, p.z), z1z1);
    const Fq h = fq_sub(u2, u1), r = fq_sub(s2, s1);
    if (fq_is_zero(h)) {
        if (fq_is_zero(r)) return j_double(p);
        Jac id; id.x = fq_one(); id.y = id.x; memset(id.z.l, 0, 32);
        return id;
    }
    const Fq h2 = fq_mul(h, h), h3 = fq_mul(h2, h), v = fq_mul(u1, h2);
    Jac o;
    o.x = fq_sub(fq_sub(fq_mul(r, r), h3), fq_add(v, v));
    o.y = fq_sub(fq_mul(r, fq_sub(v, o.x)), fq_mul(s1, h3));
    o.z = fq_mul(fq_mul(p.z, q.z), h);
    return o;
}
Fq load_fq(const uint32_t* p) {
    Fq r;
    for (int i = 0; i < 4; i++) r.l[i] = (uint64_t)p[2 * i] | ((uint64_t)p[2 * i + 1] << 32);
    return fq_canon(r);
}
}  // namespace

// s, wt: n_windows Jacobian points each as the kernels store them (x | y | z, 8 x 32-bit limbs each, Montgomery form, < 2q); the sum
// of window w is s[w] + wt[w] (bucket j weighs j + 1); wt may be null (a window of one bucket)
void bn254_g1_horner_host(const uint32_t* s, const uint32_t* wt, uint32_t n_windows, uint32_t c, uint64_t result[8]) {
    Jac r; r.x = fq_one(); r.y = r.x; memset(r.z.l, 0, 32);
    for (uint32_t w = n_windows; w-- > 0;) {
        for (uint32_t k = 0; k < c; k++) r = j_double(r);
        Jac t;
        t.x = load_fq(s + 24 * w); t.y = load_fq(s + 24 * w + 8); t.z = load_fq(s + 24 * w + 16);
        r = j_add(r, t);
        if (wt) {
            t.x = load_fq(wt + 24 * w); t.y = load_fq(wt + 24 * w + 8); t.z = load_fq(wt + 24 * w + 16);
            r = j_add(r, t);
        }
    }
    if (j_is_identity(r)) { memset(result, 0, 64); return; }
    const Fq zi = fq_inv(r.z), zi2 = fq_mul(zi, zi);
    Fq one_int; memset(one_int.l, 0, 32); one_int.l[0] = 1;                 // x R * 1 * R^-1 = x: out of Montgomery form
    const Fq x = fq_mul(fq_mul(r.x, zi2), one_int), y = fq_mul(fq_mul(r.y, fq_mul(zi2, zi)), one_int);
    memcpy(result, x.l, 32);
    memcpy(result + 4, y.l, 32);
}

// out = a + b for affine points as canonical integers (x | y, zeros = the identity): the host end of a commitment made in two parts
void bn254_g1_add_host(const uint64_t a[8], const uint64_t b[8], uint64_t out[8]) {
    auto lift = [](const uint64_t p[8]) {
        Jac r;
        bool ident = true;
        for (int i = 0; i < 8; i++) ident = ident && p[i] == 0;
        if (ident) { r.x = fq_one(); r.y = r.x; memset(r.z.l, 0, 32); return r; }
        Fq x, y, r2;
        memcpy(x.l, p, 32); memcpy(y.l, p + 4, 32); memcpy(r2.l, BN254C_FQ_R2_64, 32);
        r.x = fq_mul(fq_canon(x), r2); r.y = fq_mul(fq_canon(y), r2); r.z = fq_one();
        return r;
    };
    const Jac r = j_add(lift(a), lift(b));
    if (j_is_identity(r)) { memset(out, 0, 64); return; }
    const Fq zi = fq_inv(r.z), zi2 = fq_mul(zi, zi);
    Fq one_int; memset(one_int.l, 0, 32); one_int.l[0] = 1;
    const Fq x = fq_mul(fq_mul(r.x, zi2), one_int), y = fq_mul(fq_mul(r.y, fq_mul(zi2, zi)), one_int);
    memcpy(out, x.l, 32);
    memcpy(out + 4, y.l, 32);
}

}  // namespace gl355
