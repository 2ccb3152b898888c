/*
 * gl_oracle.c -- CPU restatement of the plonky2 prover hot path.  TEST INFRASTRUCTURE ONLY:
 * see gl_oracle.h for who may load this and for the parity-pinning status.
 *
 * Written from the algorithm descriptions, not from plonky2 source (absent from /root/reference).
 * Every function cites the reference location that pins its semantics.
 */
#include "gl_oracle.h"
#include "gl_inline.h"
#include "poseidon_rc.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif


void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * a1 -- Goldilocks field, p = 2^64 - 2^32 + 1 (chip/native_chip/arithmetic_chip.rs:19).
 * Values are kept canonical everywhere in the oracle: simplest possible model.
 * ---------------------------------------------------------------------------------------- */
uint64_t orc_add(uint64_t a, uint64_t b) { return f_add(a, b); }
uint64_t orc_sub(uint64_t a, uint64_t b) { return f_sub(a, b); }
/* textbook model, kept so tests can pin the fast reduction below against plain `% p` */
uint64_t orc_mul_ref(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % P); }
uint64_t orc_mul(uint64_t a, uint64_t b) { return f_mul(a, b); }
uint64_t orc_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1; a = canon(a);
    while (e) { if (e & 1) r = f_mul(r, a); a = f_mul(a, a); e >>= 1; }
    return r;
}
uint64_t orc_inv(uint64_t a) { return orc_pow(a, P - 2); }

/* omega_N = g^((p-1)/N) with g = 7 the multiplicative generator (chip/fri_chip.rs:162-163,
 * chip/plonk/plonk_verifier_chip.rs:219-222). */
uint64_t orc_root_of_unity(uint32_t log_n) { return orc_pow(7, (P - 1) >> log_n); }

/* F_p^2 = F_p[X]/(X^2 - 7) (arithmetic_chip.rs:109-132: left_x = a_x b_x + 7 a_y b_y,
 * left_y = a_x b_y + a_y b_x). */
void orc_ext_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
    uint64_t c0 = f_add(f_mul(a[0], b[0]), f_mul(7, f_mul(a[1], b[1])));
    uint64_t c1 = f_add(f_mul(a[0], b[1]), f_mul(a[1], b[0]));
    out[0] = c0; out[1] = c1;
}
static void ext_add(const uint64_t a[2], const uint64_t b[2], uint64_t o[2]) {
    o[0] = f_add(a[0], b[0]); o[1] = f_add(a[1], b[1]);
}
/* 1/(a0 + a1 X) = (a0 - a1 X) / (a0^2 - 7 a1^2) */
void orc_ext_inv(const uint64_t a[2], uint64_t out[2]) {
    uint64_t norm = f_sub(f_mul(a[0], a[0]), f_mul(7, f_mul(a[1], a[1])));
    uint64_t ni = orc_inv(norm);
    out[0] = f_mul(canon(a[0]), ni);
    out[1] = f_mul(f_sub(0, a[1]), ni);
}

/* ------------------------------------------------------------------------------------------
 * a5 -- bit reversal / transpose (plonky2_util; the reference calls reverse_index_bits_in_place
 * itself at chip/fri_chip.rs:6,189).
 * ---------------------------------------------------------------------------------------- */

void orc_reverse_index_bits(uint64_t *data, size_t n_rows, size_t row_len) {
    uint32_t bits = log2_exact(n_rows);
    uint64_t *tmp = (uint64_t *)malloc(row_len * sizeof(uint64_t));
    for (size_t i = 0; i < n_rows; i++) {
        size_t j = bitrev(i, bits);
        if (i < j) {
            memcpy(tmp, data + i * row_len, row_len * 8);
            memcpy(data + i * row_len, data + j * row_len, row_len * 8);
            memcpy(data + j * row_len, tmp, row_len * 8);
        }
    }
    free(tmp);
}
void orc_transpose(const uint64_t *in, size_t rows, size_t cols, uint64_t *out) {
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < cols; c++)
        for (size_t r = 0; r < rows; r++) out[c * rows + r] = in[r * cols + c];
}

/* ------------------------------------------------------------------------------------------
 * a2 -- radix-2 NTT as plonky2_field::fft does it: bit-reverse the input, then log2(n)
 * decimation-in-time layers, natural-order output; out[j] = sum_i in[i] * omega_n^(i*j).
 * Inverse = forward, then index reversal (i <-> n-i) and scaling by n^-1.
 * ---------------------------------------------------------------------------------------- */
/* per-layer contiguous root tables, as plonky2's fft_root_table: layer s (block size 2^s) uses
 * omega_{2^s}^j for j < 2^(s-1), stored back to back (total n-1 entries). */
static void ntt_one(uint64_t *a, uint32_t log_n, const uint64_t *roots) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        a[i] = canon(a[i]);
        size_t j = bitrev(i, log_n);
        if (i < j) { uint64_t t = a[i]; a[i] = canon(a[j]); a[j] = t; }
    }
    const uint64_t *layer = roots;
    for (uint32_t s = 1; s <= log_n; s++) {
        size_t m = (size_t)1 << s, half = m >> 1;
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half; j++) {
                uint64_t u = a[k + j], v = f_mul(a[k + j + half], layer[j]);
                uint64_t x = u + v;                      /* u, v canonical: one conditional subtract */
                x -= P & (uint64_t)(-(int64_t)((x < u) | (x >= P)));
                a[k + j] = x;
                a[k + j + half] = (u - v) + (P & (uint64_t)(-(int64_t)(u < v)));
            }
        layer += half;
    }
}
static uint64_t *root_table(uint32_t log_n) {
    size_t n = (size_t)1 << log_n;
    uint64_t *r = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    uint64_t *layer = r;
    for (uint32_t s = 1; s <= log_n; s++) {
        size_t half = (size_t)1 << (s - 1);
        uint64_t w = orc_root_of_unity(s), x = 1;
        for (size_t i = 0; i < half; i++) { layer[i] = x; x = f_mul(x, w); }
        layer += half;
    }
    return r;
}
void orc_ntt(uint64_t *data, uint32_t log_n, uint32_t batch, size_t stride) {
    uint64_t *roots = root_table(log_n);
#pragma omp parallel for schedule(dynamic)
    for (uint32_t c = 0; c < batch; c++) ntt_one(data + (size_t)c * stride, log_n, roots);
    free(roots);
}
void orc_intt(uint64_t *data, uint32_t log_n, uint32_t batch, size_t stride) {
    size_t n = (size_t)1 << log_n;
    uint64_t n_inv = orc_inv((uint64_t)n % P);
    uint64_t *roots = root_table(log_n);
#pragma omp parallel for schedule(dynamic)
    for (uint32_t c = 0; c < batch; c++) {
        uint64_t *a = data + (size_t)c * stride;
        ntt_one(a, log_n, roots);
        a[0] = f_mul(a[0], n_inv);
        if (n > 1) a[n / 2] = f_mul(a[n / 2], n_inv);
        for (size_t i = 1; i < n / 2; i++) {
            size_t j = n - i;
            uint64_t ci = f_mul(a[j], n_inv), cj = f_mul(a[i], n_inv);
            a[i] = ci; a[j] = cj;
        }
    }
    free(roots);
}
/* coset variants: scale coefficient i by shift^i before the forward transform / by shift^-i after
 * the inverse one (coset generator 7: plonk_verifier_chip.rs:225-227, fri_chip.rs:264). */
void orc_coset_ntt(uint64_t *data, uint32_t log_n, uint64_t shift, uint32_t batch, size_t stride) {
    size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(dynamic)
    for (uint32_t c = 0; c < batch; c++) {
        uint64_t *a = data + (size_t)c * stride, s = 1;
        for (size_t i = 0; i < n; i++) { a[i] = f_mul(canon(a[i]), s); s = f_mul(s, shift); }
    }
    orc_ntt(data, log_n, batch, stride);
}
void orc_coset_intt(uint64_t *data, uint32_t log_n, uint64_t shift, uint32_t batch, size_t stride) {
    size_t n = (size_t)1 << log_n;
    uint64_t si = orc_inv(shift);
    orc_intt(data, log_n, batch, stride);
#pragma omp parallel for schedule(dynamic)
    for (uint32_t c = 0; c < batch; c++) {
        uint64_t *a = data + (size_t)c * stride, s = 1;
        for (size_t i = 0; i < n; i++) { a[i] = f_mul(a[i], s); s = f_mul(s, si); }
    }
}
/* a3 -- PolynomialCoeffs::lde + coset_fft: zero-pad to N = n << rate_bits, scale by shift^i, NTT. */
void orc_lde(const uint64_t *coeffs, uint32_t log_n, uint32_t rate_bits, uint64_t shift,
             uint32_t batch, uint64_t *out) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
#pragma omp parallel for schedule(static)
    for (uint32_t c = 0; c < batch; c++) {
        memcpy(out + (size_t)c * N, coeffs + (size_t)c * n, n * 8);
        memset(out + (size_t)c * N + n, 0, (N - n) * 8);
    }
    orc_coset_ntt(out, log_n + rate_bits, shift, batch, N);
}

/* ------------------------------------------------------------------------------------------
 * a6 -- Poseidon permutation, NAIVE form: 30 x (add round constants, S-box x^7 on all lanes in
 * the 4+4 full rounds / lane 0 only in the 22 partial rounds, MDS).  MDS row r:
 * sum_i s[(i+r)%12]*CIRC[i] + s[r]*DIAG[r]  (chip/plonk/gates/poseidon.rs:450-486, :321-322;
 * round structure :634-686).  The product uses the fast partial-round form instead, so comparing
 * the two also cross-checks the derived tables.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t sbox7(uint64_t x) {
    uint64_t x2 = f_mul(x, x), x4 = f_mul(x2, x2), x3 = f_mul(x, x2);
    return f_mul(x3, x4);
}
void orc_poseidon_permute(uint64_t s[12]) {
    for (int i = 0; i < 12; i++) s[i] = canon(s[i]);
    for (int r = 0; r < 30; r++) {
        for (int i = 0; i < 12; i++) s[i] = f_add(s[i], ORC_POSEIDON_RC[12 * r + i]);
        if (r < 4 || r >= 26) { for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]); }
        else s[0] = sbox7(s[0]);
        uint64_t t[12];
        for (int row = 0; row < 12; row++) {
            u128 acc = 0;
            for (int i = 0; i < 12; i++) acc += (u128)s[(i + row) % 12] * ORC_MDS_CIRC[i];
            acc += (u128)s[row] * ORC_MDS_DIAG[row];
            t[row] = reduce128(acc);
        }
        memcpy(s, t, sizeof t);
    }
}
/* a7 -- sponge: rate 8, capacity 4, OVERWRITE absorb, no padding (chip/hasher_chip.rs:122-148). */
void orc_hash_no_pad(const uint64_t *in, size_t len, uint64_t out[4]) {
    uint64_t st[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t m = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < m; i++) st[i] = canon(in[off + i]);
        orc_poseidon_permute(st);
    }
    memcpy(out, st, 32);
}
/* leaf of <= 4 elements is its own digest, zero padded (chip/merkle_proof_chip.rs:52-57). */
void orc_hash_or_noop(const uint64_t *in, size_t len, uint64_t out[4]) {
    if (len <= 4) {
        for (size_t i = 0; i < 4; i++) out[i] = i < len ? canon(in[i]) : 0;
    } else orc_hash_no_pad(in, len, out);
}
/* permute(l || r || 0^4)[0..4] (merkle_proof_chip.rs:58-70). */
void orc_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
    uint64_t st[12] = {0};
    for (int i = 0; i < 4; i++) { st[i] = l[i]; st[4 + i] = r[i]; }
    orc_poseidon_permute(st);
    memcpy(out, st, 32);
}

/* ------------------------------------------------------------------------------------------
 * a8 -- Merkle tree with cap.  digests layout: per cap subtree of m leaves a
 * region of 2(m-1) digests = left-subtree || left-child || right-child || right-subtree,
 * recursively; i.e. the pair p of layer i (layer 0 = leaf digests) sits at pair slot
 * (p << (i+1)) + (1<<i) - 1.  Path direction/cap index: merkle_proof_chip.rs:58-84,
 * fri_chip.rs:72-82.
 * ---------------------------------------------------------------------------------------- */
static void fill_subtree(uint64_t *dig, size_t n_dig, const uint64_t *leaves, size_t m,
                         uint32_t leaf_len, uint64_t out[4]) {
    if (n_dig == 0) { orc_hash_or_noop(leaves, leaf_len, out); return; }
    size_t half = n_dig / 2; /* digests in each half, including the child digest slot */
    uint64_t l[4], r[4];
    fill_subtree(dig, half - 1, leaves, m / 2, leaf_len, l);
    fill_subtree(dig + (half + 1) * 4, half - 1, leaves + (m / 2) * leaf_len, m / 2, leaf_len, r);
    memcpy(dig + (half - 1) * 4, l, 32);
    memcpy(dig + half * 4, r, 32);
    orc_two_to_one(l, r, out);
}
/* the literal restatement: one recursive fill per cap subtree */
void orc_merkle_build_recursive(const uint64_t *leaves, size_t n_leaves, uint32_t leaf_len,
                                uint32_t cap_height, uint64_t *digests, uint64_t *cap) {
    size_t n_cap = (size_t)1 << cap_height, sub_leaves = n_leaves >> cap_height;
    size_t sub_dig = 2 * (sub_leaves - 1);
    for (size_t t = 0; t < n_cap; t++)
        fill_subtree(digests + t * sub_dig * 4, sub_dig, leaves + t * sub_leaves * leaf_len,
                     sub_leaves, leaf_len, cap + t * 4);
}
/* same tree, level by level (parallel over nodes) using the closed-form slot of the layout:
 * node k of layer i (layer 0 = leaf digests) of a subtree lives at digest index
 * 2*(((k>>1) << (i+1)) + (1<<i) - 1) + (k&1).  Used for the multi-threaded CPU baseline. */
typedef void (*leaf_hash_fn)(const uint64_t *, size_t, uint64_t *);
typedef void (*pair_hash_fn)(const uint64_t *, const uint64_t *, uint64_t *);
static void merkle_layered(leaf_hash_fn leaf_hash, pair_hash_fn pair_hash, const uint64_t *leaves, size_t n_leaves,
                           uint32_t leaf_len, uint32_t cap_height, uint64_t *digests, uint64_t *cap) {
    size_t n_cap = (size_t)1 << cap_height, sub_leaves = n_leaves >> cap_height;
    size_t sub_dig = 2 * (sub_leaves - 1);
    uint32_t sub_bits = log2_exact(sub_leaves);
    if (sub_bits == 0) {
#pragma omp parallel for schedule(static)
        for (size_t t = 0; t < n_cap; t++) leaf_hash(leaves + t * leaf_len, leaf_len, cap + t * 4);
        return;
    }
#pragma omp parallel for schedule(static)
    for (size_t g = 0; g < n_leaves; g++) {
        size_t t = g >> sub_bits, k = g & (sub_leaves - 1);
        size_t pos = 2 * (((k >> 1) << 1) + 0) + (k & 1);
        leaf_hash(leaves + g * leaf_len, leaf_len, digests + (t * sub_dig + pos) * 4);
    }
    for (uint32_t i = 1; i <= sub_bits; i++) {
        size_t per = sub_leaves >> i; /* nodes of layer i per subtree */
#pragma omp parallel for schedule(static)
        for (size_t g = 0; g < per * n_cap; g++) {
            size_t t = g / per, k = g % per;
            uint64_t *tree = digests + t * sub_dig * 4;
            size_t child_slot = (k << i) + ((size_t)1 << (i - 1)) - 1; /* pair k of layer i-1 */
            const uint64_t *l = tree + (2 * child_slot) * 4, *r = tree + (2 * child_slot + 1) * 4;
            uint64_t *dst = (i == sub_bits)
                                ? cap + t * 4
                                : tree + (2 * (((k >> 1) << (i + 1)) + ((size_t)1 << i) - 1) + (k & 1)) * 4;
            pair_hash(l, r, dst);
        }
    }
}
void orc_merkle_build_layered(const uint64_t *leaves, size_t n_leaves, uint32_t leaf_len,
                              uint32_t cap_height, uint64_t *digests, uint64_t *cap) {
    merkle_layered(orc_hash_or_noop, orc_two_to_one, leaves, n_leaves, leaf_len, cap_height, digests, cap);
}
/* MerkleTree::new::<F, H> for H = PoseidonHash (ORC_HASH_POSEIDON) or the reference's Bn254PoseidonHash
 * (ORC_HASH_BN254_POSEIDON, bn245_poseidon/plonky2_config.rs:57-75); same digest layout */
void orc_merkle_build_h(int hasher, const uint64_t *leaves, size_t n_leaves, uint32_t leaf_len, uint32_t cap_height,
                        uint64_t *digests, uint64_t *cap) {
    if (hasher == ORC_HASH_BN254_POSEIDON)
        merkle_layered(orc_bn254_hash_or_noop, orc_bn254_two_to_one, leaves, n_leaves, leaf_len, cap_height, digests, cap);
    else orc_merkle_build(leaves, n_leaves, leaf_len, cap_height, digests, cap);
}
void orc_merkle_build(const uint64_t *leaves, size_t n_leaves, uint32_t leaf_len,
                      uint32_t cap_height, uint64_t *digests, uint64_t *cap) {
    if (n_leaves <= 4096) orc_merkle_build_recursive(leaves, n_leaves, leaf_len, cap_height, digests, cap);
    else orc_merkle_build_layered(leaves, n_leaves, leaf_len, cap_height, digests, cap);
}
void orc_merkle_prove(const uint64_t *digests, size_t n_leaves, uint32_t cap_height,
                      size_t leaf_index, uint64_t *siblings) {
    uint32_t num_layers = log2_exact(n_leaves) - cap_height;
    size_t tree_len = 2 * ((n_leaves >> cap_height) - 1);
    const uint64_t *tree = digests + (leaf_index >> num_layers) * tree_len * 4;
    size_t pair_index = leaf_index & (((size_t)1 << num_layers) - 1);
    for (uint32_t i = 0; i < num_layers; i++) {
        size_t parity = pair_index & 1;
        pair_index >>= 1;
        size_t slot = (pair_index << (i + 1)) + ((size_t)1 << i) - 1;
        memcpy(siblings + i * 4, tree + (2 * slot + (1 - parity)) * 4, 32);
    }
}
int orc_merkle_verify(const uint64_t *leaf, uint32_t leaf_len, size_t leaf_index,
                      const uint64_t *siblings, uint32_t n_siblings, const uint64_t *cap,
                      uint32_t cap_height) {
    return orc_merkle_verify_h(ORC_HASH_POSEIDON, leaf, leaf_len, leaf_index, siblings, n_siblings, cap, cap_height);
}
int orc_merkle_verify_h(int hasher, const uint64_t *leaf, uint32_t leaf_len, size_t leaf_index,
                        const uint64_t *siblings, uint32_t n_siblings, const uint64_t *cap, uint32_t cap_height) {
    (void)cap_height;
    leaf_hash_fn leaf_hash = hasher == ORC_HASH_BN254_POSEIDON ? orc_bn254_hash_or_noop : orc_hash_or_noop;
    pair_hash_fn pair_hash = hasher == ORC_HASH_BN254_POSEIDON ? orc_bn254_two_to_one : orc_two_to_one;
    uint64_t st[4], o[4];
    leaf_hash(leaf, leaf_len, st);
    size_t idx = leaf_index;
    for (uint32_t i = 0; i < n_siblings; i++) {
        if (idx & 1) pair_hash(siblings + i * 4, st, o);
        else pair_hash(st, siblings + i * 4, o);
        memcpy(st, o, 32);
        idx >>= 1;
    }
    return memcmp(st, cap + idx * 4, 32) == 0;
}

/* ------------------------------------------------------------------------------------------
 * a4 -- PolynomialBatch::from_values / from_coeffs: iNTT -> LDE (coset 7) -> append salt columns
 * -> transpose -> bit-reverse rows -> Merkle.  Leaf order fri_chip.rs:245-264; salt = last 4 leaf
 * elements (types/assigned.rs:67-71).
 * ---------------------------------------------------------------------------------------- */
void orc_commit(const uint64_t *values, uint32_t log_n, uint32_t batch, uint32_t rate_bits,
                int is_coeffs, const uint64_t *salt, uint32_t cap_height, uint64_t *coeffs_out,
                uint64_t *leaves, uint64_t *digests, uint64_t *cap) {
    orc_commit_h(ORC_HASH_POSEIDON, values, log_n, batch, rate_bits, is_coeffs, salt, cap_height, coeffs_out, leaves, digests, cap);
}
void orc_commit_h(int hasher, const uint64_t *values, uint32_t log_n, uint32_t batch, uint32_t rate_bits,
                  int is_coeffs, const uint64_t *salt, uint32_t cap_height, uint64_t *coeffs_out,
                  uint64_t *leaves, uint64_t *digests, uint64_t *cap) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    uint32_t width = batch + (salt ? 4 : 0);
    uint64_t *coeffs = (uint64_t *)malloc((size_t)batch * n * 8);
    memcpy(coeffs, values, (size_t)batch * n * 8);
    if (!is_coeffs) orc_intt(coeffs, log_n, batch, n);
    else for (size_t i = 0; i < (size_t)batch * n; i++) coeffs[i] = canon(coeffs[i]);
    if (coeffs_out) memcpy(coeffs_out, coeffs, (size_t)batch * n * 8);
    uint64_t *lde = (uint64_t *)malloc((size_t)width * N * 8);
    orc_lde(coeffs, log_n, rate_bits, 7, batch, lde);
    if (salt) for (size_t i = 0; i < 4 * N; i++) lde[(size_t)batch * N + i] = canon(salt[i]);
    orc_transpose(lde, width, N, leaves); /* lde is [width][N] -> leaves [N][width] */
    orc_reverse_index_bits(leaves, N, width);
    orc_merkle_build_h(hasher, leaves, N, width, cap_height, digests, cap);
    free(lde); free(coeffs);
}

/* ------------------------------------------------------------------------------------------
 * a11 -- DEEP quotient (PolynomialBatch::prove_openings), pinned by the verifier's batch combine
 * chip/fri_chip.rs:112-149: sum = sum*alpha^|batch| + (sum_i alpha^i p_i(x) - sum_i alpha^i p_i(z))
 * / (x - z); reduce order chip/goldilocks_extension_chip.rs:331-342.
 * ---------------------------------------------------------------------------------------- */
void orc_deep_batch(const uint64_t *polys, uint32_t log_n, uint32_t n_polys, size_t stride,
                    const uint64_t alpha[2], const uint64_t z[2], uint64_t *acc) {
    size_t n = (size_t)1 << log_n;
    uint64_t *comp = (uint64_t *)calloc(2 * n, 8);
    uint64_t ap[2] = {1, 0};
    for (uint32_t i = 0; i < n_polys; i++) {
        const uint64_t *p = polys + (size_t)i * stride;
        for (size_t k = 0; k < n; k++) {
            uint64_t c = canon(p[k]);
            comp[2 * k] = f_add(comp[2 * k], f_mul(ap[0], c));
            comp[2 * k + 1] = f_add(comp[2 * k + 1], f_mul(ap[1], c));
        }
        orc_ext_mul(ap, alpha, ap);
    }
    /* synthetic division by (X - z): b_{n-1} = c_{n-1}; b_k = c_k + z b_{k+1}; quotient = b_1..b_{n-1} */
    uint64_t *q = (uint64_t *)calloc(2 * n, 8);
    uint64_t b[2] = {0, 0};
    for (size_t k = n; k-- > 0;) {
        uint64_t t[2];
        orc_ext_mul(b, z, t);
        ext_add(t, comp + 2 * k, b);
        if (k >= 1) { q[2 * (k - 1)] = b[0]; q[2 * (k - 1) + 1] = b[1]; }
    }
    /* acc = acc * alpha^n_polys + q   (ap now holds alpha^n_polys) */
    for (size_t k = 0; k < n; k++) {
        uint64_t t[2];
        orc_ext_mul(acc + 2 * k, ap, t);
        ext_add(t, q + 2 * k, acc + 2 * k);
    }
    free(q); free(comp);
}
void orc_eval_polys_ext(const uint64_t *polys, uint32_t log_n, uint32_t n_polys, size_t stride,
                        const uint64_t z[2], uint64_t *out) {
    size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(dynamic)
    for (uint32_t i = 0; i < n_polys; i++) {
        const uint64_t *p = polys + (size_t)i * stride;
        uint64_t acc[2] = {0, 0};
        for (size_t k = n; k-- > 0;) {
            uint64_t t[2];
            orc_ext_mul(acc, z, t);
            acc[0] = f_add(t[0], canon(p[k])); acc[1] = t[1];
        }
        out[2 * i] = acc[0]; out[2 * i + 1] = acc[1];
    }
}
/* F_p^2 LDE: the two limbs are independent base-field LDEs (omega and the shift are in F_p). */
void orc_lde_ext(const uint64_t *coeffs, uint32_t log_n, uint32_t rate_bits, uint64_t shift,
                 uint64_t *out) {
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    uint64_t *cols = (uint64_t *)malloc(2 * n * 8), *res = (uint64_t *)malloc(2 * N * 8);
    for (size_t k = 0; k < n; k++) { cols[k] = coeffs[2 * k]; cols[n + k] = coeffs[2 * k + 1]; }
    orc_lde(cols, log_n, rate_bits, shift, 2, res);
    for (size_t j = 0; j < N; j++) { out[2 * j] = res[j]; out[2 * j + 1] = res[N + j]; }
    free(cols); free(res);
}

/* ------------------------------------------------------------------------------------------
 * a12 -- FRI arity-2 fold in coefficient form: c'[k] = c[2k] + beta c[2k+1]  (== P_even + beta
 * P_odd, the verifier's interpolation formula at chip/fri_chip.rs:168-226); layer leaves are the
 * two evaluations on {x, -x}, bit-reversed order, flattened (fri_chip.rs:275-311).
 * ---------------------------------------------------------------------------------------- */
void orc_fri_fold(const uint64_t *coeffs, size_t n, const uint64_t beta[2], uint64_t *out) {
    for (size_t k = 0; k < n / 2; k++) {
        uint64_t t[2];
        orc_ext_mul(coeffs + 2 * (2 * k + 1), beta, t);
        uint64_t e[2] = {canon(coeffs[4 * k]), canon(coeffs[4 * k + 1])};
        ext_add(e, t, out + 2 * k);
    }
}
void orc_fri_layer_leaves(const uint64_t *values, size_t n, uint64_t *leaves) {
    uint32_t bits = log2_exact(n);
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev(i, bits);
        leaves[2 * i] = canon(values[2 * j]); leaves[2 * i + 1] = canon(values[2 * j + 1]);
    }
}

/* ------------------------------------------------------------------------------------------
 * a13 -- proof of work (chip/fri_chip.rs:364-376; response = last rate element popped,
 * chip/hasher_chip.rs:82-87; observe(witness) then squeeze: plonk_verifier_chip.rs:136-137).
 * Deterministic rule: the SMALLEST witness >= start.
 * ---------------------------------------------------------------------------------------- */
uint64_t orc_pow_grind(const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t start) {
    for (uint64_t w = start;; w++) {
        uint64_t st[12];
        memcpy(st, state, sizeof st);
        st[pos] = w;
        orc_poseidon_permute(st);
        if (bits == 0 || (st[7] >> (64 - bits)) == 0) return w;
    }
}

/* ------------------------------------------------------------------------------------------
 * a15 -- Challenger: inputs buffered and absorbed lazily in chunks of 8 by overwriting
 * state[0..len] then permuting; squeeze pops from the END of the rate part; a new input clears the
 * squeeze buffer (chip/hasher_chip.rs:48-89,107-120).
 * ---------------------------------------------------------------------------------------- */
void orc_challenger_init(orc_challenger *c) { memset(c, 0, sizeof *c); }
static void challenger_duplex(orc_challenger *c) {
    for (uint32_t i = 0; i < c->in_len; i++) c->state[i] = c->in_buf[i];
    c->in_len = 0;
    if (c->hasher == ORC_HASH_BN254_POSEIDON) orc_bn254_permute(c->state);
    else orc_poseidon_permute(c->state);
    memcpy(c->out_buf, c->state, 64);
    c->out_len = 8;
}
void orc_challenger_observe(orc_challenger *c, const uint64_t *elems, size_t n) {
    for (size_t i = 0; i < n; i++) {
        c->out_len = 0;
        c->in_buf[c->in_len++] = canon(elems[i]);
        if (c->in_len == 8) challenger_duplex(c);
    }
}
uint64_t orc_challenger_squeeze(orc_challenger *c) {
    if (c->in_len > 0 || c->out_len == 0) challenger_duplex(c);
    return c->out_buf[--c->out_len];
}

/* ------------------------------------------------------------------------------------------
 * a9 -- permutation argument (wires_permutation_partial_products_and_zs), pinned by
 * chip/plonk/vanishing_poly.rs:54-108,183-218: per row i, x = g^i:
 *   q_j = (w_j + beta*k_j*x + gamma) / (w_j + beta*sigma_j + gamma),  chunk products of max_degree
 *   running product:  acc = Z(x); for each chunk: acc *= chunk -> partial products; last = Z(gx).
 * Outputs: z_out[n]; pp_out[(n_chunks-1)][n].
 * ---------------------------------------------------------------------------------------- */
void orc_zs_partial_products(const uint64_t *wires, const uint64_t *sigmas, const uint64_t *k_is,
                             uint32_t log_n, uint32_t n_routed, uint32_t max_degree,
                             uint64_t beta, uint64_t gamma, uint64_t *z_out, uint64_t *pp_out) {
    size_t n = (size_t)1 << log_n;
    uint32_t n_chunks = (n_routed + max_degree - 1) / max_degree;
    uint64_t g = orc_root_of_unity(log_n), x = 1, z = 1;
    for (size_t i = 0; i < n; i++) {
        z_out[i] = z;
        uint64_t acc = z;
        for (uint32_t ch = 0; ch < n_chunks; ch++) {
            uint64_t num = 1, den = 1;
            for (uint32_t j = ch * max_degree; j < (ch + 1) * max_degree && j < n_routed; j++) {
                uint64_t w = canon(wires[(size_t)j * n + i]);
                uint64_t s_id = f_mul(k_is[j], x);
                num = f_mul(num, f_add(f_add(w, f_mul(beta, s_id)), gamma));
                den = f_mul(den, f_add(f_add(w, f_mul(beta, canon(sigmas[(size_t)j * n + i]))), gamma));
            }
            acc = f_mul(acc, f_mul(num, orc_inv(den)));
            if (ch + 1 < n_chunks) pp_out[(size_t)ch * n + i] = acc;
        }
        z = acc;
        x = f_mul(x, g);
    }
}
