/*
 * gl_prover.c -- CPU restatement of plonky2's CircuitData::prove for the gate set the reference uses.
 * TEST INFRASTRUCTURE ONLY (see gl_oracle.h): the checker for gl355_prove / gl355_prove_sparse (the flat proof
 * must be byte-identical) and the CPU baseline of bench.py.  Nothing in the product links or calls it.
 *
 * What it restates (plonky2 @ 72229c47, absent from /root/reference; semantics pinned by the reference's
 * in-tree verifier, file:line on each step):
 *   prover.rs prove(): commit wires -> betas, gammas -> Z / partial products -> alphas -> quotient -> zeta ->
 *   openings -> FRI (transcript order src/plonky2_verifier/chip/plonk/plonk_verifier_chip.rs:55-154)
 *   vanishing_poly.rs eval_vanishing_poly_base (chip/plonk/vanishing_poly.rs:18-153, term order :110-123)
 *   gate evaluators (chip/plonk/gates/{noop,constant,public_input,base_sum,poseidon,arithmetic,
 *   arithmetic_extension,multiplication_extension,poseidon_mds,random_access,reducing,reducing_extension}.rs),
 *   selector filters (chip/plonk/gates/mod.rs:87-132), FRI commit / PoW / queries (chip/fri_chip.rs:168-376).
 *
 * Written independently of the HIP implementation: per point of the quotient coset it materialises plonky2's
 * `vanishing_terms` vector [L0 (Z - 1)] ++ [partial-product checks] ++ [sum_g filter_g constraint_g,k] and then
 * reduces it with powers of alpha, where the HIP kernel streams the terms through running accumulators.
 * The blinding / salt values follow the product's documented key -> stream convention (include/gl355.h): ChaCha20
 * (RFC 8439 2.3 block function, restated below and pinned by the RFC's own test vector, tests/test_oracle_golden.py)
 * under the proof's 256-bit key, block counter = block index, nonce = (stream, 0, 0); streams 1/2/3 = salt of the
 * wires / Z / quotient oracle, 4 = witness blinding rows; element k = key-stream bytes [16k, 16k+16) as a little-endian
 * 128-bit number mod p.  So (witness, key) fixes the proof on both sides.
 */
#include "gl_oracle.h"
#include "gl_inline.h"
#include "poseidon_rc.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SALT 4

typedef struct { uint64_t c0, c1; } e2;
static inline e2 e2_mk(uint64_t a, uint64_t b) { e2 r = {a, b}; return r; }
static inline e2 e2_add(e2 a, e2 b) { return e2_mk(f_add(a.c0, b.c0), f_add(a.c1, b.c1)); }
static inline e2 e2_sub(e2 a, e2 b) { return e2_mk(f_sub(a.c0, b.c0), f_sub(a.c1, b.c1)); }
static inline e2 e2_mul(e2 a, e2 b) {
    return e2_mk(f_add(f_mul(a.c0, b.c0), f_mul(7, f_mul(a.c1, b.c1))), f_add(f_mul(a.c0, b.c1), f_mul(a.c1, b.c0)));
}
static inline e2 e2_scale(e2 a, uint64_t s) { return e2_mk(f_mul(a.c0, s), f_mul(a.c1, s)); }

/* ---- ChaCha20 block function, RFC 8439 section 2.3 (written from the RFC text) -------------------------------- */
static inline uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
static void quarter_round(uint32_t *st, int a, int b, int c, int d) {
    st[a] += st[b]; st[d] ^= st[a]; st[d] = rotl32(st[d], 16);
    st[c] += st[d]; st[b] ^= st[c]; st[b] = rotl32(st[b], 12);
    st[a] += st[b]; st[d] ^= st[a]; st[d] = rotl32(st[d], 8);
    st[c] += st[d]; st[b] ^= st[c]; st[b] = rotl32(st[b], 7);
}
static uint32_t le32(const uint8_t *b) { return (uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24; }
/* key 32 bytes, nonce 3 words, -> 64 key-stream bytes (serialised little-endian, RFC 8439 2.3) */
void orc_chacha20_block(const uint8_t key[32], uint32_t counter, const uint32_t nonce[3], uint8_t out[64]) {
    uint32_t init[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u}, w[16];
    for (int i = 0; i < 8; i++) init[4 + i] = le32(key + 4 * i);
    init[12] = counter; init[13] = nonce[0]; init[14] = nonce[1]; init[15] = nonce[2];
    memcpy(w, init, sizeof w);
    for (int r = 0; r < 10; r++) {
        quarter_round(w, 0, 4, 8, 12); quarter_round(w, 1, 5, 9, 13); quarter_round(w, 2, 6, 10, 14); quarter_round(w, 3, 7, 11, 15);
        quarter_round(w, 0, 5, 10, 15); quarter_round(w, 1, 6, 11, 12); quarter_round(w, 2, 7, 8, 13); quarter_round(w, 3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) {
        const uint32_t v = w[i] + init[i];
        out[4 * i] = (uint8_t)v; out[4 * i + 1] = (uint8_t)(v >> 8); out[4 * i + 2] = (uint8_t)(v >> 16); out[4 * i + 3] = (uint8_t)(v >> 24);
    }
}
/* elements [0, count) of blinding stream `stream` under `key` (the convention of include/gl355.h) */
void orc_blinding_elements(const uint8_t key[32], uint32_t stream, uint64_t count, uint64_t *out) {
    const uint32_t nonce[3] = {stream, 0, 0};
    uint8_t blk[64];
    for (uint64_t k = 0; k < count; k++) {
        if ((k & 3) == 0) orc_chacha20_block(key, (uint32_t)(k >> 2), nonce, blk);
        const uint8_t *b = blk + 16 * (k & 3);
        unsigned __int128 v = 0;
        for (int i = 15; i >= 0; i--) v = (v << 8) | b[i];
        out[k] = (uint64_t)(v % (unsigned __int128)P);
    }
}
/* per-unit key of a batch: first 32 bytes of block 0 under the batch key with nonce ("key", index_lo, index_hi) */
void orc_derive_key(const uint8_t base[32], uint64_t index, uint8_t out[32]) {
    const uint32_t nonce[3] = {0x0079656bu, (uint32_t)index, (uint32_t)(index >> 32)};
    uint8_t blk[64];
    orc_chacha20_block(base, 0, nonce, blk);
    memcpy(out, blk, 32);
}

/* ---- committed batch (PolynomialBatch) ------------------------------------------------------------------ */
orc_batch *orc_batch_commit(const uint64_t *values, uint32_t log_n, uint32_t batch, uint32_t rate_bits, int is_coeffs,
                            const uint64_t *salt, uint32_t cap_height) {
    return orc_batch_commit_h(ORC_HASH_POSEIDON, values, log_n, batch, rate_bits, is_coeffs, salt, cap_height);
}
orc_batch *orc_batch_commit_h(int hasher, const uint64_t *values, uint32_t log_n, uint32_t batch, uint32_t rate_bits, int is_coeffs,
                              const uint64_t *salt, uint32_t cap_height) {
    orc_batch *b = (orc_batch *)calloc(1, sizeof *b);
    size_t n = (size_t)1 << log_n, N = n << rate_bits;
    b->log_n = log_n; b->rate_bits = rate_bits; b->batch = batch; b->cap_height = cap_height;
    b->leaf_len = batch + (salt ? SALT : 0);
    b->coeffs = (uint64_t *)malloc((size_t)batch * n * 8);
    b->leaves = (uint64_t *)malloc((size_t)b->leaf_len * N * 8);
    b->digests = (uint64_t *)malloc(2 * (N - ((size_t)1 << cap_height)) * 32 + 32);
    b->cap = (uint64_t *)malloc(((size_t)1 << cap_height) * 32);
    orc_commit_h(hasher, values, log_n, batch, rate_bits, is_coeffs, salt, cap_height, b->coeffs, b->leaves, b->digests, b->cap);
    return b;
}
void orc_batch_free(orc_batch *b) {
    if (!b) return;
    free(b->coeffs); free(b->leaves); free(b->digests); free(b->cap); free(b);
}

/* ---- gate constraint evaluators at one point: w = wire values, k = selectors | gate constants ----------- */
static uint32_t gate_num_constraints(const orc_gate *g) {
    switch (g->type) {
    case ORC_GATE_CONSTANT: return g->param;
    case ORC_GATE_PUBLIC_INPUT: return 4;
    case ORC_GATE_BASE_SUM: return 1 + g->param;
    case ORC_GATE_POSEIDON: return 123;
    case ORC_GATE_ARITHMETIC: return g->param;
    case ORC_GATE_ARITHMETIC_EXT: return 2 * g->param;
    case ORC_GATE_MUL_EXT: return 2 * g->param;
    case ORC_GATE_POSEIDON_MDS: return 24;
    case ORC_GATE_RANDOM_ACCESS: {
        uint32_t bits = g->param & 0xFF, copies = (g->param >> 8) & 0xFF, extra = (g->param >> 16) & 0xFF;
        return copies * (bits + 2) + extra;
    }
    case ORC_GATE_REDUCING: case ORC_GATE_REDUCING_EXT: return 2 * g->param;
    default: return 0;
    }
}

static inline uint64_t sbox7(uint64_t x) {
    uint64_t x2 = f_mul(x, x), x4 = f_mul(x2, x2), x3 = f_mul(x, x2);
    return f_mul(x3, x4);
}
static void mds_layer(uint64_t s[12]) {
    uint64_t t[12];
    for (int row = 0; row < 12; row++) {
        u128 acc = 0;
        for (int i = 0; i < 12; i++) acc += (u128)s[(i + row) % 12] * ORC_MDS_CIRC[i];
        acc += (u128)s[row] * ORC_MDS_DIAG[row];
        t[row] = reduce128(acc);
    }
    memcpy(s, t, sizeof t);
}

/* PoseidonGate (gates/poseidon.rs:592-698; wires :329-380): swap bit, 4 deltas, the S-box inputs of full rounds
 * 1..3, of the 22 partial rounds and of full rounds 4..7, then the 12 outputs */
static uint32_t eval_poseidon(const uint64_t *w, uint64_t *out) {
    uint32_t k = 0;
    uint64_t swap = w[24], s[12];
    out[k++] = f_mul(swap, f_sub(swap, 1));
    for (int i = 0; i < 4; i++) {
        uint64_t lhs = w[i], rhs = w[4 + i], delta = w[25 + i];
        out[k++] = f_sub(f_mul(swap, f_sub(rhs, lhs)), delta);
        s[i] = f_add(lhs, delta);
        s[4 + i] = f_sub(rhs, delta);
    }
    for (int i = 8; i < 12; i++) s[i] = w[i];
    for (int r = 0; r < 30; r++) {
        for (int i = 0; i < 12; i++) s[i] = f_add(s[i], ORC_POSEIDON_RC[12 * r + i]);
        if (r < 4 || r >= 26) {
            if (r != 0) {
                const uint64_t *sin = r < 4 ? w + 29 + 12 * (r - 1) : w + 87 + 12 * (r - 26);
                for (int i = 0; i < 12; i++) { out[k++] = f_sub(s[i], sin[i]); s[i] = sin[i]; }
            }
            for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
        } else {
            uint64_t sin = w[65 + (r - 4)];
            out[k++] = f_sub(s[0], sin);
            s[0] = sbox7(sin);
        }
        mds_layer(s);
    }
    for (int i = 0; i < 12; i++) out[k++] = f_sub(s[i], w[12 + i]);
    return k;
}

static uint32_t eval_gate(const orc_circuit *c, const orc_gate *g, const uint64_t *w, const uint64_t *kk,
                          const uint64_t pi_hash[4], uint64_t *out) {
    const uint64_t *gc = kk + c->num_selectors;          /* this row's gate constants */
    uint32_t k = 0;
    switch (g->type) {
    case ORC_GATE_CONSTANT:                               /* gates/constant.rs:31-36 */
        for (uint32_t i = 0; i < g->param; i++) out[k++] = f_sub(gc[i], w[i]);
        break;
    case ORC_GATE_PUBLIC_INPUT:                           /* gates/public_input.rs:32-39 */
        for (uint32_t i = 0; i < 4; i++) out[k++] = f_sub(w[i], pi_hash[i]);
        break;
    case ORC_GATE_BASE_SUM: {                             /* gates/base_sum.rs:37-60, base 2 */
        uint64_t sum = 0;
        for (uint32_t i = g->param; i-- > 0;) sum = f_add(f_add(sum, sum), w[1 + i]);
        out[k++] = f_sub(sum, w[0]);
        for (uint32_t i = 0; i < g->param; i++) out[k++] = f_mul(w[1 + i], f_sub(w[1 + i], 1));
        break;
    }
    case ORC_GATE_POSEIDON:
        k = eval_poseidon(w, out);
        break;
    case ORC_GATE_ARITHMETIC:                             /* gates/arithmetic.rs:47-68 */
        for (uint32_t i = 0; i < g->param; i++) {
            uint64_t computed = f_add(f_mul(f_mul(w[4 * i], w[4 * i + 1]), gc[0]), f_mul(w[4 * i + 2], gc[1]));
            out[k++] = f_sub(w[4 * i + 3], computed);
        }
        break;
    case ORC_GATE_ARITHMETIC_EXT:                         /* gates/arithmetic_extension.rs:22-80 */
        for (uint32_t i = 0; i < g->param; i++) {
            const uint64_t *q = w + 8 * i;
            e2 computed = e2_add(e2_scale(e2_mul(e2_mk(q[0], q[1]), e2_mk(q[2], q[3])), gc[0]), e2_scale(e2_mk(q[4], q[5]), gc[1]));
            e2 d = e2_sub(e2_mk(q[6], q[7]), computed);
            out[k++] = d.c0; out[k++] = d.c1;
        }
        break;
    case ORC_GATE_MUL_EXT:                                /* gates/multiplication_extension.rs:22-68 */
        for (uint32_t i = 0; i < g->param; i++) {
            const uint64_t *q = w + 6 * i;
            e2 d = e2_sub(e2_mk(q[4], q[5]), e2_scale(e2_mul(e2_mk(q[0], q[1]), e2_mk(q[2], q[3])), gc[0]));
            out[k++] = d.c0; out[k++] = d.c1;
        }
        break;
    case ORC_GATE_POSEIDON_MDS:                           /* gates/poseidon_mds.rs:26-126 */
        for (uint32_t r = 0; r < 12; r++)
            for (uint32_t half = 0; half < 2; half++) {
                u128 acc = 0;
                for (uint32_t i = 0; i < 12; i++) acc += (u128)w[2 * ((i + r) % 12) + half] * ORC_MDS_CIRC[i];
                acc += (u128)w[2 * r + half] * ORC_MDS_DIAG[r];
                out[2 * r + half] = f_sub(w[2 * (12 + r) + half], reduce128(acc));
            }
        k = 24;
        break;
    case ORC_GATE_RANDOM_ACCESS: {                        /* gates/random_access.rs:27-147 */
        uint32_t bits = g->param & 0xFF, copies = (g->param >> 8) & 0xFF, extra = (g->param >> 16) & 0xFF;
        uint32_t vec = 1u << bits, routed = (2 + vec) * copies + extra;
        for (uint32_t cp = 0; cp < copies; cp++) {
            const uint64_t *q = w + (2 + vec) * cp, *b = w + routed + cp * bits;
            for (uint32_t i = 0; i < bits; i++) out[k++] = f_mul(b[i], f_sub(b[i], 1));
            uint64_t recon = 0;
            for (uint32_t i = bits; i-- > 0;) recon = f_add(f_add(recon, recon), b[i]);
            out[k++] = f_sub(recon, q[0]);
            uint64_t items[256];
            for (uint32_t i = 0; i < vec; i++) items[i] = q[2 + i];
            uint32_t len = vec;
            for (uint32_t lvl = 0; lvl < bits; lvl++) {   /* fold pairs: x + b (y - x) */
                for (uint32_t j = 0; j < len / 2; j++)
                    items[j] = f_add(items[2 * j], f_mul(b[lvl], f_sub(items[2 * j + 1], items[2 * j])));
                len >>= 1;
            }
            out[k++] = f_sub(items[0], q[1]);
        }
        for (uint32_t i = 0; i < extra; i++) out[k++] = f_sub(gc[i], w[(2 + vec) * copies + i]);
        break;
    }
    case ORC_GATE_REDUCING: case ORC_GATE_REDUCING_EXT: { /* gates/reducing.rs:20-85, reducing_extension.rs:20-87 */
        int ext = g->type == ORC_GATE_REDUCING_EXT;
        uint32_t n = g->param, start_accs = 6 + (ext ? 2 * n : n);
        e2 alpha = e2_mk(w[2], w[3]), acc = e2_mk(w[4], w[5]);
        for (uint32_t i = 0; i < n; i++) {
            e2 coeff = ext ? e2_mk(w[6 + 2 * i], w[7 + 2 * i]) : e2_mk(w[6 + i], 0);
            e2 acc_i = i == n - 1 ? e2_mk(w[0], w[1]) : e2_mk(w[start_accs + 2 * i], w[start_accs + 2 * i + 1]);
            e2 d = e2_sub(e2_add(e2_mul(acc, alpha), coeff), acc_i);
            out[k++] = d.c0; out[k++] = d.c1;
            acc = acc_i;
        }
        break;
    }
    default: break;
    }
    return k;
}

/* gates/mod.rs:87-132: prod_{j in group, j != i} (j - s) * (UNUSED - s) */
static uint64_t gate_filter(const orc_circuit *c, uint32_t gi, const uint64_t *kk) {
    const orc_gate *g = &c->gates[gi];
    uint64_t s = kk[g->selector_index], f = 1;
    for (uint32_t j = g->group_start; j < g->group_end; j++)
        if (j != gi) f = f_mul(f, f_sub(j, s));
    if (c->num_selectors > 1) f = f_mul(f, f_sub(UINT64_C(0xFFFFFFFF), s));
    return f;
}

/* Combined vanishing polynomial / Z_H on the quotient coset {7 w^i}, |coset| = n * max_degree, for every challenge:
 * out[c][i], natural order.  Leaves are the committed batches' rows (row of LDE index j at bitrev(j)). */
void orc_vanishing_values(const orc_circuit *c, const orc_batch *cs, const orc_batch *wires, const orc_batch *zs,
                          const uint64_t *k_is, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                          const uint64_t pi_hash[4], uint64_t *out) {
    const uint32_t nch = c->num_challenges, npp = c->num_partial_products, routed = c->num_routed_wires;
    const uint32_t chunk = c->max_degree, n_chunks = (routed + chunk - 1) / chunk;
    const uint32_t qdb = log2_exact(c->max_degree), qbits = c->degree_bits + qdb, lde_bits = c->degree_bits + c->rate_bits;
    const size_t nq = (size_t)1 << qbits, n = (size_t)1 << c->degree_bits;
    const uint32_t n_sc = c->num_selectors + c->num_constants;
    uint32_t max_gc = 0;
    for (uint32_t g = 0; g < c->num_gates; g++) {
        uint32_t k = gate_num_constraints(&c->gates[g]);
        if (k > max_gc) max_gc = k;
    }
    const uint64_t wq = orc_root_of_unity(qbits), n_f = canon((uint64_t)n);
    uint64_t zh[64], zh_inv[64];                 /* x^n - 1 depends on i mod 2^qdb only */
    {
        uint64_t sn = orc_pow(7, n), wk = orc_root_of_unity(qdb), t = 1;
        for (uint32_t k = 0; k < (1u << qdb); k++) { zh[k] = f_sub(f_mul(sn, t), 1); zh_inv[k] = orc_inv(zh[k]); t = f_mul(t, wk); }
    }
    const size_t BLK = 64;
#pragma omp parallel
    {
        uint64_t *terms = (uint64_t *)malloc((size_t)(nch + nch * n_chunks + max_gc) * 8);
        uint64_t *gcs = (uint64_t *)malloc((size_t)(max_gc + 1) * 8);
        uint64_t xs[64], dinv[64], pref[64];
#pragma omp for schedule(dynamic, 4)
        for (size_t blk = 0; blk < nq / BLK + (nq % BLK != 0); blk++) {
            size_t i0 = blk * BLK, cnt = nq - i0 < BLK ? nq - i0 : BLK;
            /* x_i and 1 / (n (x_i - 1)) by one batched inversion */
            uint64_t x = f_mul(7, orc_pow(wq, i0)), run = 1;
            for (size_t j = 0; j < cnt; j++) {
                xs[j] = x; x = f_mul(x, wq);
                pref[j] = run; run = f_mul(run, f_mul(n_f, f_sub(xs[j], 1)));
            }
            uint64_t inv = orc_inv(run);
            for (size_t j = cnt; j-- > 0;) { dinv[j] = f_mul(inv, pref[j]); inv = f_mul(inv, f_mul(n_f, f_sub(xs[j], 1))); }
            for (size_t j = 0; j < cnt; j++) {
                const size_t iq = i0 + j, inext = (iq + ((size_t)1 << qdb)) & (nq - 1);
                const size_t row = bitrev(iq << (c->rate_bits - qdb), lde_bits), row_next = bitrev(inext << (c->rate_bits - qdb), lde_bits);
                const uint64_t *kk = cs->leaves + row * cs->leaf_len;      /* selectors | constants | sigmas */
                const uint64_t *w = wires->leaves + row * wires->leaf_len;
                const uint64_t *z = zs->leaves + row * zs->leaf_len, *z_next = zs->leaves + row_next * zs->leaf_len;
                const uint64_t xv = xs[j], zhv = zh[iq & ((1u << qdb) - 1)];
                uint32_t nt = 0;
                /* L0(x) (Z(x) - 1), L0 = (x^n - 1) / (n (x - 1))  (vanishing_poly.rs:155-178) */
                const uint64_t l0 = f_mul(zhv, dinv[j]);
                for (uint32_t ch = 0; ch < nch; ch++) terms[nt++] = f_mul(l0, f_sub(z[ch], 1));
                /* partial products (vanishing_poly.rs:54-108,183-218) */
                for (uint32_t ch = 0; ch < nch; ch++) {
                    uint64_t prev = z[ch];
                    for (uint32_t q = 0; q < n_chunks; q++) {
                        uint64_t num = 1, den = 1;
                        for (uint32_t r = q * chunk; r < (q + 1) * chunk && r < routed; r++) {
                            num = f_mul(num, f_add(f_add(w[r], f_mul(betas[ch], f_mul(k_is[r], xv))), gammas[ch]));
                            den = f_mul(den, f_add(f_add(w[r], f_mul(betas[ch], kk[n_sc + r])), gammas[ch]));
                        }
                        uint64_t next = q + 1 < n_chunks ? z[nch + ch * npp + q] : z_next[ch];
                        terms[nt++] = f_sub(f_mul(prev, num), f_mul(next, den));
                        prev = next;
                    }
                }
                /* gate constraints: constraint k = sum over gates of filter * its k-th constraint */
                for (uint32_t k = 0; k < max_gc; k++) terms[nt + k] = 0;
                for (uint32_t gi = 0; gi < c->num_gates; gi++) {
                    if (c->gates[gi].type == ORC_GATE_NOOP) continue;
                    uint64_t f = gate_filter(c, gi, kk);
                    if (f == 0) continue;                                   /* the filter vanishes: gate not active here */
                    uint32_t cnt_g = eval_gate(c, &c->gates[gi], w, kk, pi_hash, gcs);
                    for (uint32_t k = 0; k < cnt_g; k++) terms[nt + k] = f_add(terms[nt + k], f_mul(f, gcs[k]));
                }
                nt += max_gc;
                for (uint32_t ch = 0; ch < nch; ch++) {
                    uint64_t acc = 0;
                    for (uint32_t k = nt; k-- > 0;) acc = f_add(f_mul(acc, alphas[ch]), terms[k]);
                    out[(size_t)ch * nq + iq] = f_mul(acc, zh_inv[iq & ((1u << qdb) - 1)]);
                }
            }
        }
        free(terms); free(gcs);
    }
}

/* ---- proof size (layout of include/gl355.h) ---------------------------------------------------------------- */
uint64_t orc_proof_words(const orc_prover_data *pd) {
    const orc_circuit *c = pd->circuit;
    const uint64_t n_cap = (uint64_t)1 << pd->cap_height;
    const uint32_t nch = c->num_challenges, lde_bits = c->degree_bits + c->rate_bits;
    const uint32_t widths[4] = {c->num_selectors + c->num_constants + c->num_routed_wires, c->num_wires,
                                nch * (1 + c->num_partial_products), nch * c->max_degree};
    uint64_t w = 8 + 3 * n_cap * 4, n_open = 0;
    for (int o = 0; o < 4; o++) n_open += widths[o];
    w += 2 * (n_open + nch) + (uint64_t)pd->n_fri_layers * n_cap * 4 + 2 * (((uint64_t)1 << c->degree_bits) >> pd->n_fri_layers) + 1;
    uint64_t per_q = 1;
    for (int o = 0; o < 4; o++) per_q += widths[o] + ((pd->zero_knowledge && o > 0) ? SALT : 0) + (uint64_t)(lde_bits - pd->cap_height) * 4;
    for (uint32_t l = 0; l < pd->n_fri_layers; l++) per_q += 4 + (uint64_t)(lde_bits - 1 - l - pd->cap_height) * 4;
    return w + per_q * pd->num_queries;
}

static void squeeze_n(orc_challenger *ch, uint64_t *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = orc_challenger_squeeze(ch); }

/* smallest witness, searched in parallel blocks (plonky2 grinds with rayon; any valid witness verifies, the
 * smallest makes the proof deterministic) */
static uint64_t grind(const orc_challenger *ch, uint32_t bits) {
    uint64_t st[12];
    memcpy(st, ch->state, sizeof st);
    for (uint32_t i = 0; i < ch->in_len; i++) st[i] = ch->in_buf[i];
    const uint32_t pos = ch->in_len;
    if (bits == 0) return 0;
    const uint64_t BLK = 1 << 14;
    for (uint64_t base = 0;; base += BLK) {
        uint64_t best = UINT64_MAX;
#pragma omp parallel for reduction(min : best) schedule(static)
        for (uint64_t w = base; w < base + BLK; w++) {
            if (w > best) continue;
            uint64_t t[12];
            memcpy(t, st, sizeof t);
            t[pos] = w;
            if (ch->hasher == ORC_HASH_BN254_POSEIDON) orc_bn254_permute(t);
            else orc_poseidon_permute(t);
            if ((t[7] >> (64 - bits)) == 0 && w < best) best = w;
        }
        if (best != UINT64_MAX) return best;
    }
}

/* wires: the full witness [num_wires][n] (including blinding rows).  proof: orc_proof_words(pd) words. */
int orc_prove(const orc_prover_data *pd, const uint64_t *wires, const uint64_t *public_inputs, uint32_t n_pi, const uint8_t key[32],
              uint64_t *proof) {
    const orc_circuit *c = pd->circuit;
    const orc_batch *cs = pd->constants_sigmas;
    const uint32_t nch = c->num_challenges, qdf = c->max_degree, npp = c->num_partial_products, routed = c->num_routed_wires;
    const uint32_t lde_bits = c->degree_bits + c->rate_bits, cap_h = pd->cap_height, nl = pd->n_fri_layers;
    const size_t n = (size_t)1 << c->degree_bits, N = (size_t)1 << lde_bits, n_cap = (size_t)1 << cap_h;
    const int zk = pd->zero_knowledge != 0;
    uint64_t *out = proof;
    const uint64_t need = orc_proof_words(pd);
    out[0] = need; out[1] = c->degree_bits; out[2] = nl; out[3] = pd->num_queries; out[4] = n_pi; out[5] = zk; out[6] = cap_h; out[7] = nch;
    out += 8;

    orc_challenger ch;
    orc_challenger_init(&ch);
    ch.hasher = pd->hasher;
    const int hs = pd->hasher;
    uint64_t pi_hash[4];
    orc_hash_no_pad(public_inputs, n_pi, pi_hash);
    orc_challenger_observe(&ch, pd->circuit_digest, 4);
    orc_challenger_observe(&ch, pi_hash, 4);

    uint64_t *salt = zk ? (uint64_t *)malloc((size_t)SALT * N * 8) : NULL;
#define FRESH_SALT(id) \
    if (zk) orc_blinding_elements(key, (uint32_t)(id), (uint64_t)SALT * N, salt);
    /* ---- wires ---- */
    FRESH_SALT(1)
    orc_batch *b_w = orc_batch_commit_h(hs, wires, c->degree_bits, c->num_wires, c->rate_bits, 0, salt, cap_h);
    memcpy(out, b_w->cap, n_cap * 32); orc_challenger_observe(&ch, out, n_cap * 4); out += n_cap * 4;
    uint64_t betas[4], gammas[4], alphas[4];
    squeeze_n(&ch, betas, nch);
    squeeze_n(&ch, gammas, nch);
    /* ---- Z / partial products: columns [Z_c]_c | [pp_{c,k}]_{c,k} ---- */
    const uint32_t z_width = nch * (1 + npp);
    uint64_t *zbuf = (uint64_t *)malloc((size_t)z_width * n * 8);
    for (uint32_t k = 0; k < nch; k++)
        orc_zs_partial_products(wires, pd->sigmas, pd->k_is, c->degree_bits, routed, qdf, betas[k], gammas[k], zbuf + (size_t)k * n,
                                zbuf + ((size_t)nch + (size_t)k * npp) * n);
    FRESH_SALT(2)
    orc_batch *b_z = orc_batch_commit_h(hs, zbuf, c->degree_bits, z_width, c->rate_bits, 0, salt, cap_h);
    memcpy(out, b_z->cap, n_cap * 32); orc_challenger_observe(&ch, out, n_cap * 4); out += n_cap * 4;
    squeeze_n(&ch, alphas, nch);
    /* ---- quotient: values on the coset -> coefficients -> max_degree chunks of degree < n ---- */
    const uint32_t qdb = log2_exact(qdf);
    const size_t nq = n << qdb;
    uint64_t *qv = (uint64_t *)malloc((size_t)nch * nq * 8);
    orc_vanishing_values(c, cs, b_w, b_z, pd->k_is, betas, gammas, alphas, pi_hash, qv);
    orc_coset_intt(qv, c->degree_bits + qdb, 7, nch, nq);
    FRESH_SALT(3)
    orc_batch *b_q = orc_batch_commit_h(hs, qv, c->degree_bits, nch * qdf, c->rate_bits, 1, salt, cap_h);
    memcpy(out, b_q->cap, n_cap * 32); orc_challenger_observe(&ch, out, n_cap * 4); out += n_cap * 4;
    uint64_t zeta[2], zeta_next[2];
    squeeze_n(&ch, zeta, 2);
    const uint64_t g = orc_root_of_unity(c->degree_bits);
    zeta_next[0] = f_mul(zeta[0], g); zeta_next[1] = f_mul(zeta[1], g);
    /* ---- openings: every polynomial at zeta, the Z polynomials at g zeta ---- */
    const orc_batch *oracles[4] = {cs, b_w, b_z, b_q};
    size_t n_open = 0;
    for (int o = 0; o < 4; o++) n_open += oracles[o]->batch;
    uint64_t *all = (uint64_t *)malloc(n_open * n * 8);
    {
        size_t off = 0;
        for (int o = 0; o < 4; o++) { memcpy(all + off * n, oracles[o]->coeffs, (size_t)oracles[o]->batch * n * 8); off += oracles[o]->batch; }
    }
    uint64_t *p_open = out; out += 2 * (n_open + nch);
    orc_eval_polys_ext(all, c->degree_bits, (uint32_t)n_open, n, zeta, p_open);
    orc_eval_polys_ext(b_z->coeffs, c->degree_bits, nch, n, zeta_next, p_open + 2 * n_open);
    orc_challenger_observe(&ch, p_open, 2 * (n_open + nch));
    uint64_t fri_alpha[2];
    squeeze_n(&ch, fri_alpha, 2);
    /* ---- DEEP quotient: batch 0 = all at zeta, batch 1 = Z at g zeta (fri_chip.rs:112-149) ---- */
    uint64_t *acc = (uint64_t *)calloc(2 * N, 8);     /* zero padded to N for the FRI folds */
    orc_deep_batch(all, c->degree_bits, (uint32_t)n_open, n, fri_alpha, zeta, acc);
    orc_deep_batch(b_z->coeffs, c->degree_bits, nch, n, fri_alpha, zeta_next, acc);
    free(all);
    /* ---- FRI commit phase (fri_chip.rs:168-226,275-316) ---- */
    uint64_t *p_caps = out; out += (uint64_t)nl * n_cap * 4;
    uint64_t *values = (uint64_t *)malloc(2 * N * 8), *coeffs2 = (uint64_t *)malloc(2 * N * 8), *coeffs = acc;
    orc_lde_ext(coeffs, c->degree_bits, c->rate_bits, 7, values);
    uint64_t **lv = (uint64_t **)calloc(nl + 1, sizeof *lv), **dg = (uint64_t **)calloc(nl + 1, sizeof *dg);
    uint64_t shift = 7;
    size_t len = N;
    for (uint32_t l = 0; l < nl; l++) {
        lv[l] = (uint64_t *)malloc(len / 2 * 32);
        dg[l] = (uint64_t *)malloc(2 * (len / 2 - n_cap) * 32 + 32);
        orc_fri_layer_leaves(values, len, lv[l]);
        orc_merkle_build_h(hs, lv[l], len / 2, 4, cap_h, dg[l], p_caps + (size_t)l * n_cap * 4);
        orc_challenger_observe(&ch, p_caps + (size_t)l * n_cap * 4, n_cap * 4);
        uint64_t beta[2];
        squeeze_n(&ch, beta, 2);
        orc_fri_fold(coeffs, len, beta, coeffs2);
        { uint64_t *t = coeffs; coeffs = coeffs2; coeffs2 = t; }
        len >>= 1;
        shift = f_mul(shift, shift);
        if (l + 1 < nl) orc_lde_ext(coeffs, log2_exact(len), 0, shift, values);
    }
    const size_t final_len = len >> c->rate_bits;
    memcpy(out, coeffs, final_len * 16);
    orc_challenger_observe(&ch, out, final_len * 2); out += 2 * final_len;
    /* ---- proof of work, query indices ---- */
    const uint64_t pow_witness = grind(&ch, pd->pow_bits);
    *out++ = pow_witness;
    orc_challenger_observe(&ch, &pow_witness, 1);
    (void)orc_challenger_squeeze(&ch);                 /* the PoW response */
    uint64_t *q_idx = (uint64_t *)malloc((size_t)pd->num_queries * 8);
    squeeze_n(&ch, q_idx, pd->num_queries);
    /* ---- query rounds: initial trees, then every layer at x >> (l + 1) ---- */
    const uint32_t depth0 = lde_bits - cap_h;
    for (uint32_t q = 0; q < pd->num_queries; q++) {
        const size_t x = q_idx[q] & (N - 1);
        *out++ = x;
        for (int o = 0; o < 4; o++) {
            const uint32_t ll = oracles[o]->leaf_len;
            memcpy(out, oracles[o]->leaves + x * ll, (size_t)ll * 8); out += ll;
            orc_merkle_prove(oracles[o]->digests, N, cap_h, x, out); out += (size_t)depth0 * 4;
        }
        for (uint32_t l = 0; l < nl; l++) {
            const size_t idx = x >> (l + 1);
            memcpy(out, lv[l] + idx * 4, 32); out += 4;
            orc_merkle_prove(dg[l], N >> (l + 1), cap_h, idx, out); out += (size_t)(lde_bits - 1 - l - cap_h) * 4;
        }
    }
    for (uint32_t l = 0; l < nl; l++) { free(lv[l]); free(dg[l]); }
    free(lv); free(dg); free(q_idx); free(values); free(coeffs); free(coeffs2); free(qv); free(zbuf); free(salt);
    orc_batch_free(b_w); orc_batch_free(b_z); orc_batch_free(b_q);
    return (uint64_t)(out - proof) == need ? 0 : -1;
}

/* the sparse-witness entry: scatter the given rows, fill the blinding rows from the key exactly as the
 * product documents (include/gl355.h gl355_prove_sparse), then prove */
int orc_prove_sparse(const orc_prover_data *pd, const uint32_t *row_idx, const uint64_t *rows, uint32_t n_rows, uint32_t blind_start,
                     uint32_t n_blind, uint32_t z_start, uint32_t n_z_pairs, const uint64_t *public_inputs, uint32_t n_pi,
                     const uint8_t key[32], uint64_t *proof) {
    const uint32_t nw = pd->circuit->num_wires;
    const size_t n = (size_t)1 << pd->circuit->degree_bits;
    uint64_t *wires = (uint64_t *)calloc((size_t)nw * n, 8);
    for (uint32_t r = 0; r < n_rows; r++)
        for (uint32_t cidx = 0; cidx < nw; cidx++) wires[(size_t)cidx * n + row_idx[r]] = canon(rows[(size_t)r * nw + cidx]);
    const size_t n_a = (size_t)n_blind * nw;
    /* plonky2 `blind`: every routed column of a Z-blinding pair has its own random value, the same on both rows of the pair
     * (stream order: column-major over the pairs, after the n_blind x num_wires block) */
    const uint32_t routed = pd->circuit->num_routed_wires;
    const size_t n_z = (size_t)n_z_pairs * routed;
    uint64_t *bl = (uint64_t *)malloc((n_a + n_z + 1) * 8);
    orc_blinding_elements(key, 4, n_a + n_z, bl);
    for (size_t gidx = 0; gidx < n_a; gidx++) wires[(gidx / n_blind) * n + blind_start + gidx % n_blind] = bl[gidx];
    for (uint32_t cidx = 0; cidx < routed; cidx++)
        for (size_t k = 0; k < n_z_pairs; k++)
            wires[(size_t)cidx * n + z_start + 2 * k] = wires[(size_t)cidx * n + z_start + 2 * k + 1] = bl[n_a + (size_t)cidx * n_z_pairs + k];
    free(bl);
    int rc = orc_prove(pd, wires, public_inputs, n_pi, key, proof);
    free(wires);
    return rc;
}
