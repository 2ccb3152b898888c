// extern "C" entry points of libgl355 (include/gl355.h) and the resident FRI-commit pipeline (a4).
//
// gl355_commit replaces plonky2::fri::oracle::PolynomialBatch::{from_values, from_coeffs,
// lde_values} + MerkleTree::new, which the reference reaches 4x per proof through
// CircuitBuilder::build / CircuitData::prove (src/plonky2_semaphore/access_set.rs:91,94;
// recursion.rs:167-168; wrapper.rs:41,55).  plonky2 does iFFT -> LDE -> append salt -> transpose ->
// bit-reverse rows -> hash; here the LDE is written once, column-major and already in bit-reversed
// row order (DIF output order), the leaf hash reads it coalesced, and the row-major `leaves`
// matrix only exists if the caller asks for the plonky2-layout export.
#include "gl355_internal.h"

using namespace gl355;

namespace gl355 {

// forward / inverse (coset) NTT in place, natural order in and out
int32_t ntt_dev(Ctx* ctx, uint64_t* data, uint32_t log_n, uint32_t batch, uint64_t stride, bool inverse,
                uint64_t coset_shift) {
    if (batch == 0) return GL355_OK;
    const uint64_t n = 1ull << log_n;
    if (log_n > 24) return ctx->fail(GL355_E_UNSUPPORTED, "ntt: log_n > 24 unsupported");
    NttPlan p;
    p.in = data; p.out = data; p.in_col_stride = stride; p.out_col_stride = stride;
    p.log_n = log_n; p.batch = batch; p.inverse = inverse;
    const bool coset = coset_shift != 0 && gl_canon(coset_shift) != 1;
    if (!inverse) {
        // natural -> (DIF) -> natural; coset scaling on the natural-order input
        if (coset) GL355_TRY(ctx->pow_tables(coset_shift, &p.pre_lo, &p.pre_hi));
        p.out_bitrev = false;
        return ntt_run(ctx, p);
    }
    // inverse: x[i] = n^-1 * sum_k X[k] omega^-ik, then (coset) * shift^-i on the natural-order output
    p.scale = gl_canon(gl_inv(n % GL_P));
    if (log_n <= 14) {
        if (coset) GL355_TRY(ctx->pow_tables(gl_inv(coset_shift), &p.post_lo, &p.post_hi));
        return ntt_run(ctx, p);
    }
    // two-pass sizes: bring the input to bit-reversed order first so the natural-output flow (which
    // supports the post multiplier) applies
    GL355_TRY(bitrev_permute(ctx, data, data, log_n, 1, stride, stride, batch));
    p.in_bitrev = true; p.out_bitrev = false;
    if (coset) GL355_TRY(ctx->pow_tables(gl_inv(coset_shift), &p.post_lo, &p.post_hi));
    return ntt_run(ctx, p);
}

// inverse NTT from bit-reversed input to natural-order output (used on quotient evaluations)
int32_t intt_from_bitrev_dev(Ctx* ctx, const uint64_t* in, uint64_t in_stride, uint64_t* out, uint64_t out_stride,
                             uint32_t log_n, uint32_t batch, uint64_t coset_shift) {
    if (batch == 0) return GL355_OK;
    NttPlan p;
    p.in = in; p.out = out; p.in_col_stride = in_stride; p.out_col_stride = out_stride;
    p.log_n = log_n; p.batch = batch; p.inverse = true; p.in_bitrev = true; p.out_bitrev = false;
    p.scale = gl_canon(gl_inv((1ull << log_n) % GL_P));
    if (coset_shift != 0 && gl_canon(coset_shift) != 1) GL355_TRY(ctx->pow_tables(gl_inv(coset_shift), &p.post_lo, &p.post_hi));
    return ntt_run(ctx, p);
}

int32_t lde_dev(Ctx* ctx, const uint64_t* coeffs, uint64_t in_stride, uint32_t log_n, uint32_t rate_bits, uint64_t shift,
                uint32_t batch, uint64_t* out, uint64_t out_stride, bool out_bitrev) {
    if (batch == 0) return GL355_OK;
    if (rate_bits > 4) return ctx->fail(GL355_E_UNSUPPORTED, "lde: rate_bits > 4 unsupported");
    if (log_n + rate_bits > 28) return ctx->fail(GL355_E_UNSUPPORTED, "lde: N > 2^28 unsupported");
    const uint32_t n_cosets = 1u << rate_bits;
    const uint64_t n = 1ull << log_n;
    // coset c evaluates on (shift * omega_N^c) * <omega_n>; in the bit-reversed result it is the
    // contiguous block bitrev(c)
    std::vector<uint64_t> bases(n_cosets);
    const uint64_t wN = gl_root_of_unity(log_n + rate_bits);
    uint64_t g = gl_canon(shift);
    for (uint32_t c = 0; c < n_cosets; c++) { bases[c] = gl_canon(g); g = gl_mul(g, wN); }
    NttPlan p;
    p.in = coeffs; p.out = out; p.in_col_stride = in_stride; p.out_col_stride = out_stride;
    p.log_n = log_n; p.batch = batch; p.inverse = false; p.in_bitrev = false; p.out_bitrev = true;
    p.n_cosets = n_cosets; p.coset_out_stride = n; p.coset_ratio = gl_canon(wN);
    for (uint32_t c = 0; c < n_cosets; c++) p.coset_slot[c] = (uint8_t)host_brev(c, rate_bits);
    GL355_TRY(ctx->pow_tables_multi(bases, &p.pre_lo, &p.pre_hi));
    if (log_n == 0) {
        // constants: every evaluation equals the coefficient
        return ctx->fail(GL355_E_UNSUPPORTED, "lde: log_n == 0 unsupported");
    }
    GL355_TRY(ntt_run(ctx, p));
    if (!out_bitrev) GL355_TRY(bitrev_permute(ctx, out, out, log_n + rate_bits, 1, out_stride, out_stride, batch));
    return GL355_OK;
}

// the same on an explicit context (stream + scratch): a read-only oracle such as a circuit's preprocessed
// constants_sigmas may be opened concurrently by several prover contexts of one device
int32_t oracle_open_batch_on(Ctx* ctx, const gl355_oracle* o, const uint64_t* indices, uint32_t n_idx, uint64_t* leaves,
                                    uint64_t* siblings) {
    const uint32_t bits = o->log_n + o->rate_bits, layers = bits - o->cap_height;
    const uint64_t N = 1ull << bits;
    for (uint32_t i = 0; i < n_idx; i++)
        if (indices[i] >= N) return ctx->fail(GL355_E_INVALID_ARG, "oracle_open_batch: index out of range");
    Scratch sc(ctx);
    const uint64_t n_leaf = (uint64_t)n_idx * o->leaf_len, n_sib = (uint64_t)n_idx * layers * 4;
    GL355_TRY(sc.get((n_idx + n_leaf + n_sib) * 8));
    uint64_t* d_idx = sc.as<uint64_t>();
    uint64_t* d_leaf = d_idx + n_idx;
    uint64_t* d_sib = d_leaf + n_leaf;
    GL355_HIP(ctx, hipMemcpyAsync(d_idx, indices, (uint64_t)n_idx * 8, hipMemcpyHostToDevice, ctx->stream));
    GL355_TRY(open_batch_dev(ctx, o->lde, N, o->leaf_len, o->digests, bits, o->cap_height, d_idx, n_idx, d_leaf, d_sib));
    GL355_HIP(ctx, ctx->d2h(leaves, d_leaf, n_leaf * 8));
    if (n_sib) GL355_HIP(ctx, ctx->d2h(siblings, d_sib, n_sib * 8));
    GL355_HIP(ctx, ctx->wait());
    return GL355_OK;
}


}  // namespace gl355

// ---- oracle (PolynomialBatch) --------------------------------------------------------------

#define CTX_OR_FAIL(h)                      \
    Ctx* ctx = ctx_of(h);                   \
    if (!ctx) return GL355_E_INVALID_ARG;   \
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed")

extern "C" {

int32_t gl355_field_batch(gl355_ctx* h, int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n) {
    CTX_OR_FAIL(h);
    if (op < 0 || op > GL355_OP_EXT_INV) return ctx->fail(GL355_E_INVALID_ARG, "field_batch: bad op");
    const uint64_t w = (op >= GL355_OP_EXT_MUL) ? 2 : 1;
    Staged sa(ctx), sb(ctx), so(ctx);
    GL355_TRY(sa.open(a, n * w * 8, 1));
    GL355_TRY(sb.open(b, b ? n * w * 8 : 0, 1));
    GL355_TRY(so.open(out, n * w * 8, 2));
    GL355_TRY(field_batch_dev(ctx, op, sa.as<uint64_t>(), sb.as<uint64_t>(), so.as<uint64_t>(), n));
    return so.finish();
}

static int32_t ntt_common(gl355_ctx* h, uint64_t* data, uint32_t log_n, uint32_t batch, uint64_t stride, uint64_t shift,
                          int32_t inverse) {
    CTX_OR_FAIL(h);
    const uint64_t n = 1ull << log_n;
    if (log_n > 24) return ctx->fail(GL355_E_UNSUPPORTED, "ntt: log_n > 24 unsupported");
    if (batch == 0) return GL355_OK;
    if (!data) return ctx->fail(GL355_E_INVALID_ARG, "ntt: null data");
    if (stride < n) return ctx->fail(GL355_E_INVALID_ARG, "ntt: stride < n");
    Staged s(ctx);
    GL355_TRY(s.open(data, ((uint64_t)(batch - 1) * stride + n) * 8, 3));
    GL355_TRY(ntt_dev(ctx, s.as<uint64_t>(), log_n, batch, stride, inverse != 0, shift));
    return s.finish();
}
int32_t gl355_ntt(gl355_ctx* h, uint64_t* data, uint32_t log_n, uint32_t batch, uint64_t stride, int32_t inverse) {
    return ntt_common(h, data, log_n, batch, stride, 0, inverse);
}
int32_t gl355_coset_ntt(gl355_ctx* h, uint64_t* data, uint32_t log_n, uint32_t batch, uint64_t stride, uint64_t shift,
                        int32_t inverse) {
    return ntt_common(h, data, log_n, batch, stride, shift, inverse);
}

static int32_t lde_common(gl355_ctx* h, const uint64_t* coeffs, uint32_t log_n, uint32_t rate_bits, uint64_t shift,
                          uint32_t batch, uint64_t* out, bool bitrev) {
    CTX_OR_FAIL(h);
    if (batch == 0) return GL355_OK;
    if (!coeffs || !out) return ctx->fail(GL355_E_INVALID_ARG, "lde: null buffer");
    if (log_n == 0 || log_n + rate_bits > 28) return ctx->fail(GL355_E_UNSUPPORTED, "lde: unsupported size");
    const uint64_t n = 1ull << log_n, N = n << rate_bits;
    Staged si(ctx), so(ctx);
    GL355_TRY(si.open(coeffs, (uint64_t)batch * n * 8, 1));
    GL355_TRY(so.open(out, (uint64_t)batch * N * 8, 2));
    GL355_TRY(lde_dev(ctx, si.as<uint64_t>(), n, log_n, rate_bits, shift, batch, so.as<uint64_t>(), N, bitrev));
    return so.finish();
}
int32_t gl355_lde(gl355_ctx* h, const uint64_t* coeffs, uint32_t log_n, uint32_t rate_bits, uint64_t shift, uint32_t batch,
                  uint64_t* out) {
    return lde_common(h, coeffs, log_n, rate_bits, shift, batch, out, false);
}
int32_t gl355_lde_bitrev(gl355_ctx* h, const uint64_t* coeffs, uint32_t log_n, uint32_t rate_bits, uint64_t shift,
                         uint32_t batch, uint64_t* out) {
    return lde_common(h, coeffs, log_n, rate_bits, shift, batch, out, true);
}

int32_t gl355_transpose(gl355_ctx* h, const uint64_t* in, uint64_t rows, uint64_t cols, uint64_t* out) {
    CTX_OR_FAIL(h);
    if (rows == 0 || cols == 0) return GL355_OK;
    if (cols > 0xFFFFFFFFull || rows > 0xFFFFFFFFull) return ctx->fail(GL355_E_UNSUPPORTED, "transpose: dimension too large");
    Staged si(ctx), so(ctx);
    GL355_TRY(si.open(in, rows * cols * 8, 1));
    GL355_TRY(so.open(out, rows * cols * 8, 2));
    // `in` is rows x cols row-major == column-major with `rows` columns of length `cols`
    GL355_TRY(transpose_cols_to_rows(ctx, si.as<uint64_t>(), so.as<uint64_t>(), cols, (uint32_t)rows, cols, (uint32_t)rows, 0));
    return so.finish();
}
int32_t gl355_reverse_index_bits(gl355_ctx* h, uint64_t* data, uint64_t n_rows, uint32_t row_len) {
    CTX_OR_FAIL(h);
    if (n_rows <= 1 || row_len == 0) return GL355_OK;
    const uint32_t lg = log2_u64(n_rows);
    if ((1ull << lg) != n_rows) return ctx->fail(GL355_E_INVALID_ARG, "reverse_index_bits: n_rows must be a power of two");
    Staged s(ctx);
    GL355_TRY(s.open(data, n_rows * row_len * 8, 3));
    GL355_TRY(bitrev_permute(ctx, s.as<uint64_t>(), s.as<uint64_t>(), lg, row_len, 0, 0, 1));
    return s.finish();
}

#define HASHER_OR_FAIL(hs) \
    if ((hs) != GL355_HASH_POSEIDON && (hs) != GL355_HASH_BN254_POSEIDON) return ctx->fail(GL355_E_INVALID_ARG, "unknown hasher")

int32_t gl355_permute_h(gl355_ctx* h, int32_t hasher, uint64_t* states, uint64_t count) {
    CTX_OR_FAIL(h);
    HASHER_OR_FAIL(hasher);
    if (count == 0) return GL355_OK;
    Staged s(ctx);
    GL355_TRY(s.open(states, count * 96, 3));
    if (hasher == GL355_HASH_BN254_POSEIDON) GL355_TRY(bn254_permute_dev(ctx, s.as<uint64_t>(), count));
    else GL355_TRY(poseidon_permute_dev(ctx, s.as<uint64_t>(), count));
    return s.finish();
}
int32_t gl355_poseidon_permute(gl355_ctx* h, uint64_t* states, uint64_t count) { return gl355_permute_h(h, GL355_HASH_POSEIDON, states, count); }

int32_t gl355_hash_no_pad_h(gl355_ctx* h, int32_t hasher, const uint64_t* inputs, uint64_t n, uint32_t len, uint64_t* digests) {
    CTX_OR_FAIL(h);
    HASHER_OR_FAIL(hasher);
    if (n == 0) return GL355_OK;
    Staged si(ctx), so(ctx);
    GL355_TRY(si.open(inputs, n * len * 8, 1));
    GL355_TRY(so.open(digests, n * 32, 2));
    if (hasher == GL355_HASH_BN254_POSEIDON) GL355_TRY(bn254_hash_leaves_dev(ctx, si.as<uint64_t>(), n, len, false, 0, so.as<uint64_t>(), true));
    else GL355_TRY(hash_no_pad_dev(ctx, si.as<uint64_t>(), n, len, so.as<uint64_t>()));
    return so.finish();
}
int32_t gl355_hash_no_pad(gl355_ctx* h, const uint64_t* inputs, uint64_t n, uint32_t len, uint64_t* digests) {
    return gl355_hash_no_pad_h(h, GL355_HASH_POSEIDON, inputs, n, len, digests);
}
int32_t gl355_hash_leaves_h(gl355_ctx* h, int32_t hasher, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, uint64_t* digests) {
    CTX_OR_FAIL(h);
    HASHER_OR_FAIL(hasher);
    if (n_leaves == 0) return GL355_OK;
    Staged si(ctx), so(ctx);
    GL355_TRY(si.open(leaves, n_leaves * leaf_len * 8, 1));
    GL355_TRY(so.open(digests, n_leaves * 32, 2));
    if (hasher == GL355_HASH_BN254_POSEIDON) GL355_TRY(bn254_hash_leaves_dev(ctx, si.as<uint64_t>(), n_leaves, leaf_len, false, 0, so.as<uint64_t>(), false));
    else GL355_TRY(hash_leaves_dev(ctx, si.as<uint64_t>(), n_leaves, leaf_len, false, 0, so.as<uint64_t>()));
    return so.finish();
}
int32_t gl355_hash_leaves(gl355_ctx* h, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, uint64_t* digests) {
    return gl355_hash_leaves_h(h, GL355_HASH_POSEIDON, leaves, n_leaves, leaf_len, digests);
}
int32_t gl355_two_to_one_h(gl355_ctx* h, int32_t hasher, const uint64_t* left, const uint64_t* right, uint64_t n, uint64_t* out) {
    CTX_OR_FAIL(h);
    HASHER_OR_FAIL(hasher);
    if (n == 0) return GL355_OK;
    Staged sl(ctx), sr(ctx), so(ctx);
    GL355_TRY(sl.open(left, n * 32, 1));
    GL355_TRY(sr.open(right, n * 32, 1));
    GL355_TRY(so.open(out, n * 32, 2));
    if (hasher == GL355_HASH_BN254_POSEIDON) GL355_TRY(bn254_two_to_one_dev(ctx, sl.as<uint64_t>(), sr.as<uint64_t>(), n, so.as<uint64_t>()));
    else GL355_TRY(two_to_one_dev(ctx, sl.as<uint64_t>(), sr.as<uint64_t>(), n, so.as<uint64_t>()));
    return so.finish();
}
int32_t gl355_two_to_one(gl355_ctx* h, const uint64_t* left, const uint64_t* right, uint64_t n, uint64_t* out) {
    return gl355_two_to_one_h(h, GL355_HASH_POSEIDON, left, right, n, out);
}

int32_t gl355_merkle_build(gl355_ctx* h, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, uint32_t cap_height,
                           uint64_t* digests, uint64_t* cap) {
    return gl355_merkle_build_h(h, GL355_HASH_POSEIDON, leaves, n_leaves, leaf_len, cap_height, digests, cap);
}
int32_t gl355_merkle_build_h(gl355_ctx* h, int32_t hasher, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, uint32_t cap_height,
                             uint64_t* digests, uint64_t* cap) {
    CTX_OR_FAIL(h);
    HASHER_OR_FAIL(hasher);
    if (n_leaves == 0) return ctx->fail(GL355_E_INVALID_ARG, "merkle: empty tree");
    const uint32_t lg = log2_u64(n_leaves);
    if ((1ull << lg) != n_leaves) return ctx->fail(GL355_E_INVALID_ARG, "merkle: n_leaves must be a power of two");
    if (cap_height > lg) return ctx->fail(GL355_E_INVALID_ARG, "merkle: cap_height > log2(n_leaves)");
    const uint64_t n_cap = 1ull << cap_height, n_dig = 2 * (n_leaves - n_cap);
    Staged sl(ctx), sd(ctx), sc(ctx);
    GL355_TRY(sl.open(leaves, n_leaves * leaf_len * 8, 1));
    GL355_TRY(sd.open(digests, n_dig * 32, 2));
    GL355_TRY(sc.open(cap, n_cap * 32, 2));
    if (hasher == GL355_HASH_BN254_POSEIDON)
        GL355_TRY(bn254_merkle_build_dev(ctx, sl.as<uint64_t>(), n_leaves, leaf_len, false, 0, cap_height, sd.as<uint64_t>(), sc.as<uint64_t>()));
    else
        GL355_TRY(merkle_build_dev(ctx, sl.as<uint64_t>(), n_leaves, leaf_len, false, 0, cap_height, sd.as<uint64_t>(), sc.as<uint64_t>()));
    GL355_TRY(sd.finish());
    return sc.finish();
}

static int32_t prove_from(Ctx* ctx, const uint64_t* digests_dev, uint64_t n_leaves, uint32_t cap_height, uint64_t leaf_index,
                          uint64_t* siblings_host) {
    const uint32_t lg = log2_u64(n_leaves);
    if (leaf_index >= n_leaves) return ctx->fail(GL355_E_INVALID_ARG, "merkle_prove: index out of range");
    const uint32_t num_layers = lg - cap_height;
    const uint64_t tree_len = 2 * ((n_leaves >> cap_height) - 1);
    const uint64_t* tree = digests_dev + (leaf_index >> num_layers) * tree_len * 4;
    uint64_t pair_index = leaf_index & ((1ull << num_layers) - 1);
    for (uint32_t i = 0; i < num_layers; i++) {
        const uint64_t parity = pair_index & 1;
        pair_index >>= 1;
        const uint64_t slot = (pair_index << (i + 1)) + (1ull << i) - 1;
        GL355_HIP(ctx, ctx->d2h(siblings_host + 4 * i, tree + (2 * slot + (1 - parity)) * 4, 32));
    }
    GL355_HIP(ctx, ctx->wait());
    return GL355_OK;
}
int32_t gl355_merkle_prove(gl355_ctx* h, const uint64_t* digests, uint64_t n_leaves, uint32_t cap_height, uint64_t leaf_index,
                           uint64_t* siblings) {
    CTX_OR_FAIL(h);
    const uint32_t lg = log2_u64(n_leaves);
    if (n_leaves == 0 || (1ull << lg) != n_leaves || cap_height > lg) return ctx->fail(GL355_E_INVALID_ARG, "merkle_prove: bad shape");
    if (leaf_index >= n_leaves) return ctx->fail(GL355_E_INVALID_ARG, "merkle_prove: index out of range");
    if (ptr_is_device(digests)) return prove_from(ctx, digests, n_leaves, cap_height, leaf_index, siblings);
    // host digests: pure index arithmetic (MerkleTree::prove)
    const uint32_t num_layers = lg - cap_height;
    const uint64_t tree_len = 2 * ((n_leaves >> cap_height) - 1);
    const uint64_t* tree = digests + (leaf_index >> num_layers) * tree_len * 4;
    uint64_t pair_index = leaf_index & ((1ull << num_layers) - 1);
    for (uint32_t i = 0; i < num_layers; i++) {
        const uint64_t parity = pair_index & 1;
        pair_index >>= 1;
        const uint64_t slot = (pair_index << (i + 1)) + (1ull << i) - 1;
        memcpy(siblings + 4 * i, tree + (2 * slot + (1 - parity)) * 4, 32);
    }
    return GL355_OK;
}

// ---- a4: commit -------------------------------------------------------------------------------
int32_t gl355_commit(gl355_ctx* h, const uint64_t* values, uint32_t log_n, uint32_t batch, uint32_t rate_bits,
                     int32_t is_coeffs, const uint64_t* salt, uint32_t cap_height, gl355_oracle** out) {
    return gl355_commit_h(h, GL355_HASH_POSEIDON, values, log_n, batch, rate_bits, is_coeffs, salt, cap_height, out);
}
int32_t gl355_commit_h(gl355_ctx* h, int32_t hasher, const uint64_t* values, uint32_t log_n, uint32_t batch, uint32_t rate_bits,
                       int32_t is_coeffs, const uint64_t* salt, uint32_t cap_height, gl355_oracle** out) {
    CTX_OR_FAIL(h);
    HASHER_OR_FAIL(hasher);
    if (!out) return ctx->fail(GL355_E_INVALID_ARG, "commit: null out");
    *out = nullptr;
    if (!values || batch == 0) return ctx->fail(GL355_E_INVALID_ARG, "commit: empty batch");
    if (log_n == 0 || log_n > 24 || rate_bits > 4 || log_n + rate_bits > 28) return ctx->fail(GL355_E_UNSUPPORTED, "commit: unsupported size");
    if (cap_height > log_n + rate_bits) return ctx->fail(GL355_E_INVALID_ARG, "commit: cap_height too large");
    const uint64_t n = 1ull << log_n, N = n << rate_bits;
    const uint32_t leaf_len = batch + (salt ? GL355_SALT_SIZE : 0);
    const uint64_t n_cap = 1ull << cap_height, n_dig = 2 * (N - n_cap);
    gl355_oracle* o = new (std::nothrow) gl355_oracle();
    if (!o) return GL355_E_OOM;
    memset(o, 0, sizeof *o);
    o->ctx = ctx; o->log_n = log_n; o->rate_bits = rate_bits; o->batch = batch; o->leaf_len = leaf_len;
    o->cap_height = cap_height; o->n_digests = n_dig; o->hasher = hasher;
    // one allocation: coeffs | lde | digests | cap
    const uint64_t total = (uint64_t)batch * n + (uint64_t)leaf_len * N + n_dig * 4 + n_cap * 4;
    uint64_t* base = nullptr;
    hipError_t e = hipSuccess;
    {
        // from the context's caching allocator: repeated proofs reuse the same blocks (no hipMalloc/hipFree per commit)
        void* p = nullptr;
        int32_t arc = ctx->alloc(total * 8, &p);
        if (arc != GL355_OK) { delete o; return arc; }
        base = reinterpret_cast<uint64_t*>(p);
    }
    o->coeffs = base; o->lde = base + (uint64_t)batch * n; o->digests = o->lde + (uint64_t)leaf_len * N; o->cap = o->digests + n_dig * 4;
    int32_t rc = GL355_OK;
    do {
        // values -> coefficients (kept: the openings and the DEEP quotient read them)
        hipMemcpyKind kind = ptr_is_device(values) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        e = hipMemcpyAsync(o->coeffs, values, (uint64_t)batch * n * 8, kind, ctx->stream);
        if (e != hipSuccess) { rc = ctx->fail_hip(e, "hipMemcpyAsync(values)", __FILE__, __LINE__); break; }
        if (!is_coeffs) { rc = ntt_dev(ctx, o->coeffs, log_n, batch, n, true, 0); if (rc) break; }
        else { rc = canon_dev(ctx, o->coeffs, (uint64_t)batch * n); if (rc) break; }
        // LDE on the coset 7<omega_N>, written once in leaf (bit-reversed) order
        rc = lde_dev(ctx, o->coeffs, n, log_n, rate_bits, GL355_COSET_SHIFT, batch, o->lde, N, true);
        if (rc) break;
        if (salt) {
            // salt columns join the leaves after the LDE: row i of the tree holds salt[c][bitrev(i)]
            uint64_t* dst = o->lde + (uint64_t)batch * N;
            Staged ss(ctx);
            rc = ss.open(salt, (uint64_t)GL355_SALT_SIZE * N * 8, 1); if (rc) break;
            rc = bitrev_permute(ctx, ss.as<uint64_t>(), dst, log_n + rate_bits, 1, N, N, GL355_SALT_SIZE); if (rc) break;
            rc = ss.finish(); if (rc) break;
        }
        rc = merkle_build_any(ctx, hasher, o->lde, N, leaf_len, true, N, cap_height, o->digests, o->cap);
        if (rc) break;
        e = ctx->wait();
        if (e != hipSuccess) { rc = ctx->fail_hip(e, "hipStreamSynchronize(commit)", __FILE__, __LINE__); break; }
    } while (0);
    if (rc != GL355_OK) { ctx->release(base); delete o; return rc; }
    *out = o;
    return GL355_OK;
}
int32_t gl355_oracle_destroy(gl355_oracle* o) {
    if (!o) return GL355_OK;
    (void)hipSetDevice(o->ctx->device);
    (void)o->ctx->wait();
    o->ctx->release(o->coeffs);
    delete o;
    return GL355_OK;
}
int32_t gl355_oracle_info(const gl355_oracle* o, uint32_t* log_n, uint32_t* rate_bits, uint32_t* batch, uint32_t* leaf_len,
                          uint32_t* cap_height) {
    if (!o) return GL355_E_INVALID_ARG;
    if (log_n) *log_n = o->log_n;
    if (rate_bits) *rate_bits = o->rate_bits;
    if (batch) *batch = o->batch;
    if (leaf_len) *leaf_len = o->leaf_len;
    if (cap_height) *cap_height = o->cap_height;
    return GL355_OK;
}
static int32_t oracle_copy_out(const gl355_oracle* o, void* dst, const void* src_dev, size_t bytes) {
    Ctx* ctx = o->ctx;
    hipMemcpyKind kind = ptr_is_device(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    GL355_HIP(ctx, hipMemcpyAsync(dst, src_dev, bytes, kind, ctx->stream));
    GL355_HIP(ctx, ctx->wait());
    return GL355_OK;
}
int32_t gl355_oracle_cap(const gl355_oracle* o, uint64_t* cap) {
    if (!o || !cap) return GL355_E_INVALID_ARG;
    return oracle_copy_out(o, cap, o->cap, (32ull << o->cap_height));
}
int32_t gl355_oracle_coeffs(const gl355_oracle* o, uint64_t* coeffs) {
    if (!o || !coeffs) return GL355_E_INVALID_ARG;
    return oracle_copy_out(o, coeffs, o->coeffs, ((uint64_t)o->batch << o->log_n) * 8);
}
int32_t gl355_oracle_digests(const gl355_oracle* o, uint64_t* digests) {
    if (!o || !digests) return GL355_E_INVALID_ARG;
    return oracle_copy_out(o, digests, o->digests, o->n_digests * 32);
}
int32_t gl355_oracle_leaves(const gl355_oracle* o, uint64_t* leaves) {
    if (!o || !leaves) return GL355_E_INVALID_ARG;
    Ctx* ctx = o->ctx;
    const uint64_t N = 1ull << (o->log_n + o->rate_bits);
    Staged so(ctx);
    GL355_TRY(so.open(leaves, N * o->leaf_len * 8, 2));
    GL355_TRY(transpose_cols_to_rows(ctx, o->lde, so.as<uint64_t>(), N, o->leaf_len, N, o->leaf_len, 0));
    return so.finish();
}
const uint64_t* gl355_oracle_lde_ptr(const gl355_oracle* o) { return o ? o->lde : nullptr; }
const uint64_t* gl355_oracle_coeffs_ptr(const gl355_oracle* o) { return o ? o->coeffs : nullptr; }

int32_t gl355_oracle_open(const gl355_oracle* o, uint64_t index, uint64_t* leaf, uint64_t* siblings) {
    if (!o || !leaf || !siblings) return GL355_E_INVALID_ARG;
    Ctx* ctx = o->ctx;
    const uint64_t N = 1ull << (o->log_n + o->rate_bits);
    if (index >= N) return ctx->fail(GL355_E_INVALID_ARG, "oracle_open: index out of range");
    // row `index` of the column-major LDE: one strided 2-D copy
    GL355_HIP(ctx, hipMemcpy2DAsync(leaf, 8, o->lde + index, N * 8, 8, o->leaf_len, hipMemcpyDeviceToHost, ctx->stream));
    return prove_from(ctx, o->digests, N, o->cap_height, index, siblings);
}

int32_t gl355_oracle_open_batch(const gl355_oracle* o, const uint64_t* indices, uint32_t n_idx, uint64_t* leaves,
                                uint64_t* siblings) {
    if (!o || !indices || !leaves || !siblings) return GL355_E_INVALID_ARG;
    return gl355::oracle_open_batch_on(o->ctx, o, indices, n_idx, leaves, siblings);
}

// ---- a10 -------------------------------------------------------------------------------------
static int32_t quotient_common(gl355_ctx* h, const gl355_circuit* c, const gl355_oracle* cs, const gl355_oracle* wires,
                               const gl355_oracle* zs, const uint64_t* k_is, const uint64_t* betas, const uint64_t* gammas,
                               const uint64_t* alphas, const uint64_t pi_hash[4], uint64_t* out, bool interpolate) {
    CTX_OR_FAIL(h);
    if (!c || !cs || !wires || !zs || !k_is || !betas || !gammas || !alphas || !pi_hash || !out)
        return ctx->fail(GL355_E_INVALID_ARG, "quotient: null argument");
    if (cs->log_n != c->degree_bits || wires->log_n != c->degree_bits || zs->log_n != c->degree_bits ||
        cs->rate_bits != c->rate_bits || wires->rate_bits != c->rate_bits || zs->rate_bits != c->rate_bits)
        return ctx->fail(GL355_E_INVALID_ARG, "quotient: oracle shapes do not match the circuit");
    if (cs->batch != c->num_selectors + c->num_constants + c->num_routed_wires || wires->batch != c->num_wires ||
        zs->batch != c->num_challenges * (1 + c->num_partial_products))
        return ctx->fail(GL355_E_INVALID_ARG, "quotient: oracle widths do not match the circuit");
    if (c->num_routed_wires > c->num_wires || c->max_degree == 0 ||
        c->num_partial_products + 1 != (c->num_routed_wires + c->max_degree - 1) / c->max_degree)
        return ctx->fail(GL355_E_INVALID_ARG, "quotient: inconsistent routed-wire / partial-product counts");
    for (uint32_t g = 0; g < c->num_gates && g < GL355_MAX_GATES; g++) {
        if (c->gates[g].type > GL355_GATE_TYPE_MAX) return ctx->fail(GL355_E_UNSUPPORTED, "quotient: unknown gate type");
        if (c->gates[g].selector_index >= c->num_selectors || c->gates[g].group_end > c->num_gates)
            return ctx->fail(GL355_E_INVALID_ARG, "quotient: bad selector group");
    }
    uint32_t qdb = 0;
    while ((1u << qdb) < c->max_degree) qdb++;
    const uint64_t N = 1ull << (c->degree_bits + c->rate_bits), nq = 1ull << (c->degree_bits + qdb);
    const uint32_t nch = c->num_challenges;
    Staged sk(ctx), so(ctx);
    GL355_TRY(sk.open(k_is, (uint64_t)c->num_routed_wires * 8, 1));
    GL355_TRY(so.open(out, (uint64_t)nch * nq * 8, 2));
    if (!interpolate) {
        GL355_TRY(quotient_dev(ctx, c, cs->lde, wires->lde, zs->lde, N, sk.as<uint64_t>(), betas, gammas, alphas, pi_hash, so.as<uint64_t>()));
        return so.finish();
    }
    Scratch vals(ctx);
    GL355_TRY(vals.get((uint64_t)nch * nq * 8));
    GL355_TRY(quotient_dev(ctx, c, cs->lde, wires->lde, zs->lde, N, sk.as<uint64_t>(), betas, gammas, alphas, pi_hash, vals.as<uint64_t>()));
    GL355_TRY(intt_from_bitrev_dev(ctx, vals.as<uint64_t>(), nq, so.as<uint64_t>(), nq, c->degree_bits + qdb, nch, GL355_COSET_SHIFT));
    return so.finish();
}
int32_t gl355_quotient(gl355_ctx* h, const gl355_circuit* c, const gl355_oracle* cs, const gl355_oracle* wires,
                       const gl355_oracle* zs, const uint64_t* k_is, const uint64_t* betas, const uint64_t* gammas,
                       const uint64_t* alphas, const uint64_t pi_hash[4], uint64_t* quotient_coeffs) {
    return quotient_common(h, c, cs, wires, zs, k_is, betas, gammas, alphas, pi_hash, quotient_coeffs, true);
}
int32_t gl355_quotient_values(gl355_ctx* h, const gl355_circuit* c, const gl355_oracle* cs, const gl355_oracle* wires,
                              const gl355_oracle* zs, const uint64_t* k_is, const uint64_t* betas, const uint64_t* gammas,
                              const uint64_t* alphas, const uint64_t pi_hash[4], uint64_t* values) {
    return quotient_common(h, c, cs, wires, zs, k_is, betas, gammas, alphas, pi_hash, values, false);
}

// ---- a11 -------------------------------------------------------------------------------------
static int32_t collect_polys(Ctx* ctx, const gl355_poly_ref* polys, uint32_t n_polys, std::vector<const uint64_t*>& ptrs,
                             uint32_t* log_n) {
    if (!polys || n_polys == 0) return ctx->fail(GL355_E_INVALID_ARG, "empty polynomial list");
    ptrs.resize(n_polys);
    *log_n = polys[0].oracle ? polys[0].oracle->log_n : 0;
    for (uint32_t i = 0; i < n_polys; i++) {
        const gl355_oracle* o = polys[i].oracle;
        if (!o || polys[i].column >= o->batch) return ctx->fail(GL355_E_INVALID_ARG, "polynomial reference out of range");
        if (o->log_n != *log_n) return ctx->fail(GL355_E_INVALID_ARG, "polynomial degrees inconsistent");
        ptrs[i] = o->coeffs + ((uint64_t)polys[i].column << o->log_n);
    }
    return GL355_OK;
}
int32_t gl355_deep_batch(gl355_ctx* h, const gl355_poly_ref* polys, uint32_t n_polys, const uint64_t alpha[2],
                         const uint64_t z[2], uint64_t* acc) {
    CTX_OR_FAIL(h);
    std::vector<const uint64_t*> ptrs;
    uint32_t log_n;
    GL355_TRY(collect_polys(ctx, polys, n_polys, ptrs, &log_n));
    Staged sa(ctx);
    GL355_TRY(sa.open(acc, (16ull << log_n), 3));
    GL355_TRY(deep_batch_dev(ctx, ptrs.data(), n_polys, log_n, alpha, z, sa.as<uint64_t>()));
    return sa.finish();
}
int32_t gl355_eval_polys(gl355_ctx* h, const gl355_poly_ref* polys, uint32_t n_polys, const uint64_t z[2], uint64_t* out) {
    CTX_OR_FAIL(h);
    std::vector<const uint64_t*> ptrs;
    uint32_t log_n;
    GL355_TRY(collect_polys(ctx, polys, n_polys, ptrs, &log_n));
    Staged so(ctx);
    GL355_TRY(so.open(out, 16ull * n_polys, 2));
    GL355_TRY(eval_polys_ext_dev(ctx, ptrs.data(), n_polys, log_n, z, so.as<uint64_t>()));
    return so.finish();
}
int32_t gl355_lde_ext(gl355_ctx* h, const uint64_t* coeffs, uint32_t log_n, uint32_t rate_bits, uint64_t shift, uint64_t* out) {
    CTX_OR_FAIL(h);
    if (log_n == 0 || log_n + rate_bits > 27 || rate_bits > 4) return ctx->fail(GL355_E_UNSUPPORTED, "lde_ext: unsupported size");
    const uint64_t n = 1ull << log_n, N = n << rate_bits;
    Staged si(ctx), so(ctx);
    GL355_TRY(si.open(coeffs, n * 16, 1));
    GL355_TRY(so.open(out, N * 16, 2));
    GL355_TRY(lde_ext_dev(ctx, si.as<uint64_t>(), log_n, rate_bits, shift, so.as<uint64_t>(), false));
    return so.finish();
}

// ---- a12 / a13 -------------------------------------------------------------------------------
int32_t gl355_fri_fold(gl355_ctx* h, const uint64_t* coeffs, uint64_t n, const uint64_t beta[2], uint64_t* out) {
    CTX_OR_FAIL(h);
    if (n < 2 || (n & (n - 1))) return ctx->fail(GL355_E_INVALID_ARG, "fri_fold: n must be a power of two >= 2");
    Staged si(ctx), so(ctx);
    GL355_TRY(si.open(coeffs, n * 16, 1));
    GL355_TRY(so.open(out, n * 8, 2));
    GL355_TRY(fri_fold_dev(ctx, si.as<uint64_t>(), n, beta, so.as<uint64_t>()));
    return so.finish();
}
int32_t gl355_fri_layer_commit(gl355_ctx* h, const uint64_t* values, uint64_t n, uint32_t cap_height, uint64_t* leaves,
                               uint64_t* digests, uint64_t* cap) {
    return gl355_fri_layer_commit_h(h, GL355_HASH_POSEIDON, values, n, cap_height, leaves, digests, cap);
}
int32_t gl355_fri_layer_commit_h(gl355_ctx* h, int32_t hasher, const uint64_t* values, uint64_t n, uint32_t cap_height, uint64_t* leaves,
                                 uint64_t* digests, uint64_t* cap) {
    CTX_OR_FAIL(h);
    HASHER_OR_FAIL(hasher);
    if (n < 2 || (n & (n - 1))) return ctx->fail(GL355_E_INVALID_ARG, "fri_layer_commit: n must be a power of two >= 2");
    const uint64_t n_leaves = n / 2;
    const uint32_t lg = log2_u64(n_leaves);
    if (cap_height > lg) return ctx->fail(GL355_E_INVALID_ARG, "fri_layer_commit: cap_height too large");
    const uint64_t n_cap = 1ull << cap_height, n_dig = 2 * (n_leaves - n_cap);
    Staged sv(ctx), sl(ctx), sd(ctx), sc(ctx);
    GL355_TRY(sv.open(values, n * 16, 1));
    GL355_TRY(sl.open(leaves, n * 16, 2));
    GL355_TRY(sd.open(digests, n_dig * 32, 2));
    GL355_TRY(sc.open(cap, n_cap * 32, 2));
    GL355_TRY(fri_layer_leaves_dev(ctx, sv.as<uint64_t>(), n, sl.as<uint64_t>()));
    GL355_TRY(merkle_build_any(ctx, hasher, sl.as<uint64_t>(), n_leaves, 4, false, 0, cap_height, sd.as<uint64_t>(), sc.as<uint64_t>()));
    GL355_TRY(sl.finish());
    GL355_TRY(sd.finish());
    return sc.finish();
}
int32_t gl355_pow_grind(gl355_ctx* h, const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t start, uint64_t* witness) {
    return gl355_pow_grind_h(h, GL355_HASH_POSEIDON, state, pos, bits, start, witness);
}
int32_t gl355_pow_grind_h(gl355_ctx* h, int32_t hasher, const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t start,
                          uint64_t* witness) {
    CTX_OR_FAIL(h);
    HASHER_OR_FAIL(hasher);
    if (!state || !witness || pos >= 8) return ctx->fail(GL355_E_INVALID_ARG, "pow_grind: bad argument");
    return pow_grind_any(ctx, hasher, state, pos, bits, start, witness);
}

// ---- a9 --------------------------------------------------------------------------------------
int32_t gl355_zs_partial_products(gl355_ctx* h, const uint64_t* wires, const uint64_t* sigmas, const uint64_t* k_is,
                                  uint32_t log_n, uint32_t n_routed, uint32_t max_degree, uint64_t beta, uint64_t gamma,
                                  uint64_t* z_out, uint64_t* pp_out) {
    CTX_OR_FAIL(h);
    if (n_routed == 0 || max_degree == 0 || log_n > 24) return ctx->fail(GL355_E_INVALID_ARG, "zs_partial_products: bad shape");
    const uint64_t n = 1ull << log_n;
    const uint32_t n_chunks = (n_routed + max_degree - 1) / max_degree;
    Staged sw(ctx), ss(ctx), sk(ctx), sz(ctx), sp(ctx);
    GL355_TRY(sw.open(wires, (uint64_t)n_routed * n * 8, 1));
    GL355_TRY(ss.open(sigmas, (uint64_t)n_routed * n * 8, 1));
    GL355_TRY(sk.open(k_is, (uint64_t)n_routed * 8, 1));
    GL355_TRY(sz.open(z_out, n * 8, 2));
    GL355_TRY(sp.open(pp_out, (uint64_t)(n_chunks - 1) * n * 8, 2));
    GL355_TRY(zs_partial_products_dev(ctx, sw.as<uint64_t>(), ss.as<uint64_t>(), sk.as<uint64_t>(), log_n, n_routed, max_degree,
                                      beta, gamma, sz.as<uint64_t>(), sp.as<uint64_t>()));
    GL355_TRY(sz.finish());
    return sp.finish();
}

}  // extern "C"
