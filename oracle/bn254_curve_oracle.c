/*
 * bn254_curve_oracle.c -- CPU restatement of the two halo2 kernels behind the reference's SNARK finalisation:
 * the radix-2 FFT over bn256::Fr (`halo2_proofs::arithmetic::best_fft`) and the multi-scalar multiplication over bn256::G1
 * (`best_multiexp`), which ParamsKZG::setup / keygen_vk / keygen_pk / create_proof run at k = 20..23
 * (src/plonky2_verifier/verifier_api.rs:77-92, chip/native_chip/test_utils.rs:57-95).  TEST INFRASTRUCTURE ONLY (see gl_oracle.h).
 *
 * halo2_proofs / halo2curves are un-vendored dependencies (Cargo.lock), so this follows their PUBLISHED definitions:
 *   Fr = integers mod r, Fq mod q (the alt_bn128 primes), G1: y^2 = x^3 + 3 over Fq, generator (1, 2);
 *   best_fft(a, omega, log_n): a[k] <- sum_i a[i] omega^(i k), natural order in and out; omega_n = ROOT^(2^(28 - log_n)) with
 *   ROOT = 7^((r-1)/2^28); the inverse uses omega^-1 and scales by n^-1 (EvaluationDomain::ifft);
 *   best_multiexp(coeffs, bases) = sum_i coeffs[i] * bases[i].
 * PINNED by published constants: ROOT equals halo2curves' bn256::Fr::ROOT_OF_UNITY (0x03ddb9f5...c37c9c) and 2 * (1, 2)
 * equals the EIP-196 test-vector point (0x030644e7...cfd3, 0x15ed738c...a2c4) (tests/test_bn254_curve_oracle.py, which also
 * checks this file against the big-integer model tests/pymodel_bn254_curve.py).
 * Written for clarity, not speed: 4 x 64-bit Montgomery, naive double-and-add, O(n log n) FFT.
 */
#include "gl_oracle.h"
#include "gl_inline.h"
#include "bn254_curve_tables.h"

#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t l[4]; } fe;                 /* Montgomery form */
typedef struct { const uint64_t *mod, *r2, *one; uint64_t n0inv; } field;
static const field FR = {BN254C_FR_MOD_64, BN254C_FR_R2_64, BN254C_FR_ONE_64, UINT64_C(0)};
static const field FQ = {BN254C_FQ_MOD_64, BN254C_FQ_R2_64, BN254C_FQ_ONE_64, UINT64_C(0)};
static uint64_t n0inv_of(const field *f) { return f == &FR ? BN254C_FR_N0INV_64 : BN254C_FQ_N0INV_64; }

static int geq(const uint64_t a[4], const uint64_t m[4]) {
    for (int i = 3; i >= 0; i--) { if (a[i] > m[i]) return 1; if (a[i] < m[i]) return 0; }
    return 1;
}
static void sub_in_place(uint64_t a[4], const uint64_t m[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - m[i] - (uint64_t)br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
static fe bf_add(const field *f, fe a, fe b) {
    fe r; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || geq(r.l, f->mod)) sub_in_place(r.l, f->mod);
    return r;
}
static fe bf_sub(const field *f, fe a, fe b) {
    fe r; u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)br; r.l[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + f->mod[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
static fe bf_mul(const field *f, fe a, fe b) {
    const uint64_t n0 = n0inv_of(f);
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * n0;
        c = (u128)m * f->mod[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * f->mod[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe r; memcpy(r.l, t, 32);
    if (t[4] || geq(r.l, f->mod)) sub_in_place(r.l, f->mod);
    return r;
}
static fe from_limbs(const uint64_t l[4]) { fe r; memcpy(r.l, l, 32); return r; }
static fe to_mont(const field *f, const uint64_t l[4]) {
    fe a = from_limbs(l);
    while (geq(a.l, f->mod)) sub_in_place(a.l, f->mod);      /* any 256-bit input is accepted */
    return bf_mul(f, a, from_limbs(f->r2));
}
static void from_mont(const field *f, fe a, uint64_t out[4]) { fe one = {{1, 0, 0, 0}}; fe r = bf_mul(f, a, one); memcpy(out, r.l, 32); }
static int is_zero(fe a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static int bf_eq(fe a, fe b) { return memcmp(a.l, b.l, 32) == 0; }
static fe bf_pow(const field *f, fe a, const uint64_t e[4]) {
    fe r = from_limbs(f->one);
    for (int i = 255; i >= 0; i--) { r = bf_mul(f, r, r); if ((e[i >> 6] >> (i & 63)) & 1) r = bf_mul(f, r, a); }
    return r;
}
static fe bf_inv(const field *f, fe a) {
    uint64_t e[4]; memcpy(e, f->mod, 32);
    e[0] -= 2;                                   /* the low limb of both primes is > 2 */
    return bf_pow(f, a, e);
}

/* ---- Fr FFT ------------------------------------------------------------------------------------------------ */
/* data: n x 4 limbs, canonical values in and out; forward: a[k] = sum_i a[i] w^(ik); inverse: with w^-1 and 1/n */
void orc_bn254_fr_ntt(uint64_t *data, uint32_t log_n, int inverse) {
    const size_t n = (size_t)1 << log_n;
    fe *a = (fe *)malloc(n * sizeof(fe));
    for (size_t i = 0; i < n; i++) a[i] = to_mont(&FR, data + 4 * i);
    fe w_n = to_mont(&FR, inverse ? BN254C_FR_ROOT_INV_64 : BN254C_FR_ROOT_64);
    for (uint32_t k = log_n; k < BN254C_FR_S; k++) w_n = bf_mul(&FR, w_n, w_n);
    for (size_t i = 0; i < n; i++) {                 /* bit-reversal, then decimation in time */
        size_t j = 0;
        for (uint32_t b = 0; b < log_n; b++) j |= ((i >> b) & 1) << (log_n - 1 - b);
        if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (uint32_t s = 1; s <= log_n; s++) {
        const size_t m = (size_t)1 << s;
        fe w_m = w_n;
        for (uint32_t k = s; k < log_n; k++) w_m = bf_mul(&FR, w_m, w_m);
        for (size_t base = 0; base < n; base += m) {
            fe w = from_limbs(FR.one);
            for (size_t j = 0; j < m / 2; j++) {
                const fe t = bf_mul(&FR, w, a[base + j + m / 2]), u = a[base + j];
                a[base + j] = bf_add(&FR, u, t);
                a[base + j + m / 2] = bf_sub(&FR, u, t);
                w = bf_mul(&FR, w, w_m);
            }
        }
    }
    if (inverse) {
        uint64_t nl[4] = {(uint64_t)n, 0, 0, 0};
        const fe ninv = bf_inv(&FR, to_mont(&FR, nl));
        for (size_t i = 0; i < n; i++) a[i] = bf_mul(&FR, a[i], ninv);
    }
    for (size_t i = 0; i < n; i++) from_mont(&FR, a[i], data + 4 * i);
    free(a);
}
/* element-wise a*b mod r and a+b mod r on canonical limbs (for the linearity / convolution property tests) */
void orc_bn254_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { from_mont(&FR, bf_mul(&FR, to_mont(&FR, a), to_mont(&FR, b)), out); }
void orc_bn254_fr_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) { from_mont(&FR, bf_add(&FR, to_mont(&FR, a), to_mont(&FR, b)), out); }

/* ---- G1 --------------------------------------------------------------------------------------------------------- */
typedef struct { fe x, y, z; } jac;                    /* z == 0: the identity */
static jac j_identity(void) { jac p; memset(&p, 0, sizeof p); p.x = p.y = from_limbs(FQ.one); return p; }
static jac j_double(jac p) {
    if (is_zero(p.z)) return p;
    const field *f = &FQ;
    fe a = bf_mul(f, p.x, p.x), b = bf_mul(f, p.y, p.y), c = bf_mul(f, b, b);
    fe xb = bf_add(f, p.x, b);
    fe d = bf_sub(f, bf_sub(f, bf_mul(f, xb, xb), a), c); d = bf_add(f, d, d);
    fe e = bf_add(f, bf_add(f, a, a), a), ff = bf_mul(f, e, e);
    jac r;
    r.x = bf_sub(f, ff, bf_add(f, d, d));
    fe c8 = bf_add(f, c, c); c8 = bf_add(f, c8, c8); c8 = bf_add(f, c8, c8);
    r.y = bf_sub(f, bf_mul(f, e, bf_sub(f, d, r.x)), c8);
    r.z = bf_mul(f, p.y, p.z); r.z = bf_add(f, r.z, r.z);
    return r;
}
static jac j_add(jac p, jac q) {
    if (is_zero(p.z)) return q;
    if (is_zero(q.z)) return p;
    const field *f = &FQ;
    fe z1z1 = bf_mul(f, p.z, p.z), z2z2 = bf_mul(f, q.z, q.z);
    fe u1 = bf_mul(f, p.x, z2z2), u2 = bf_mul(f, q.x, z1z1);
    fe s1 = bf_mul(f, bf_mul(f, p.y, q.z), z2z2), s2 = bf_mul(f, bf_mul(f, q.y, p.z), z1z1);
    if (bf_eq(u1, u2)) return bf_eq(s1, s2) ? j_double(p) : j_identity();
    fe h = bf_sub(f, u2, u1), r = bf_sub(f, s2, s1);
    fe h2 = bf_mul(f, h, h), h3 = bf_mul(f, h2, h), v = bf_mul(f, u1, h2);
    jac o;
    o.x = bf_sub(f, bf_sub(f, bf_mul(f, r, r), h3), bf_add(f, v, v));
    o.y = bf_sub(f, bf_mul(f, r, bf_sub(f, v, o.x)), bf_mul(f, s1, h3));
    o.z = bf_mul(f, bf_mul(f, p.z, q.z), h);
    return o;
}
static jac j_from_affine(const uint64_t xy[8]) {      /* (0, 0) encodes the identity */
    int zero = 1;
    for (int i = 0; i < 8; i++) zero &= xy[i] == 0;
    if (zero) return j_identity();
    jac p; p.x = to_mont(&FQ, xy); p.y = to_mont(&FQ, xy + 4); p.z = from_limbs(FQ.one);
    return p;
}
static void j_to_affine(jac p, uint64_t xy[8]) {
    if (is_zero(p.z)) { memset(xy, 0, 64); return; }
    const field *f = &FQ;
    fe zi = bf_inv(f, p.z), zi2 = bf_mul(f, zi, zi);
    from_mont(f, bf_mul(f, p.x, zi2), xy);
    from_mont(f, bf_mul(f, p.y, bf_mul(f, zi2, zi)), xy + 4);
}
static jac j_mul(jac p, const uint64_t k[4]) {
    jac r = j_identity();
    for (int i = 255; i >= 0; i--) { r = j_double(r); if ((k[i >> 6] >> (i & 63)) & 1) r = j_add(r, p); }
    return r;
}
int orc_bn254_g1_on_curve(const uint64_t xy[8]) {
    const field *f = &FQ;
    fe x = to_mont(f, xy), y = to_mont(f, xy + 4);
    uint64_t three[4] = {3, 0, 0, 0};
    return bf_eq(bf_mul(f, y, y), bf_add(f, bf_mul(f, bf_mul(f, x, x), x), to_mont(f, three)));
}
/* out = k * P (affine in, affine out; scalars are plain 256-bit integers, reduced mod r implicitly by the group) */
void orc_bn254_g1_mul(const uint64_t p[8], const uint64_t k[4], uint64_t out[8]) { j_to_affine(j_mul(j_from_affine(p), k), out); }
void orc_bn254_g1_add(const uint64_t p[8], const uint64_t q[8], uint64_t out[8]) { j_to_affine(j_add(j_from_affine(p), j_from_affine(q)), out); }
/* best_multiexp by its definition: sum_i scalars[i] * points[i] */
void orc_bn254_g1_msm(const uint64_t *points, const uint64_t *scalars, size_t n, uint64_t out[8]) {
    jac acc = j_identity();
    jac *part = NULL;
    int nt = 1;
#ifdef _OPENMP
    nt = orc_num_threads();
#endif
    part = (jac *)malloc((size_t)nt * sizeof(jac));
    for (int t = 0; t < nt; t++) part[t] = j_identity();
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int t = 0; t < nt; t++)
        for (size_t i = (size_t)t; i < n; i += (size_t)nt) part[t] = j_add(part[t], j_mul(j_from_affine(points + 8 * i), scalars + 4 * i));
    for (int t = 0; t < nt; t++) acc = j_add(acc, part[t]);
    free(part);
    j_to_affine(acc, out);
}
/* points[i] = (first + i * step) * G, affine, for large structured test inputs: one Jacobian chain + batch normalisation */
void orc_bn254_g1_multiples(uint64_t first, uint64_t step, size_t n, uint64_t *points) {
    const uint64_t g[8] = {1, 0, 0, 0, 2, 0, 0, 0};
    const uint64_t kf[4] = {first, 0, 0, 0}, ks[4] = {step, 0, 0, 0};
    jac cur = j_mul(j_from_affine(g), kf);
    const jac stp = j_mul(j_from_affine(g), ks);
    jac *pts = (jac *)malloc(n * sizeof(jac));
    fe *pre = (fe *)malloc(n * sizeof(fe));
    const field *f = &FQ;
    fe run = from_limbs(f->one);
    for (size_t i = 0; i < n; i++) {
        pts[i] = cur;
        pre[i] = run;
        if (!is_zero(cur.z)) run = bf_mul(f, run, cur.z);
        cur = j_add(cur, stp);
    }
    fe inv = bf_inv(f, run);
    for (size_t i = n; i-- > 0;) {
        if (is_zero(pts[i].z)) { memset(points + 8 * i, 0, 64); continue; }
        const fe zi = bf_mul(f, inv, pre[i]);
        inv = bf_mul(f, inv, pts[i].z);
        const fe zi2 = bf_mul(f, zi, zi);
        from_mont(f, bf_mul(f, pts[i].x, zi2), points + 8 * i);
        from_mont(f, bf_mul(f, pts[i].y, bf_mul(f, zi2, zi)), points + 8 * i + 4);
    }
    free(pts); free(pre);
}
