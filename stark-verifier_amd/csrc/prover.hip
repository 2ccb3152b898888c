// FRI prover loop in one call (a12 + a13 + a14 of SURVEY.md 8): commit phase, proof of work and the
// layer openings of every query, with the polynomial resident on the device throughout.
//
// Replaces plonky2::fri::prover::{fri_proof, fri_committed_trees, fri_proof_of_work,
// fri_prover_query_rounds} (reached through CircuitData::prove, src/plonky2_semaphore/access_set.rs:94).
// Per layer: bit-reverse the evaluations into leaves of two extension elements (no leaf hash), Merkle
// tree with cap, observe cap, squeeze beta, fold the coefficients, coset-NTT on shift^2
// (src/plonky2_verifier/chip/fri_chip.rs:168-226,275-316; transcript order
// chip/plonk/plonk_verifier_chip.rs:120-140).  Only the cap (<= 16 digests) crosses PCIe per layer.
#include "gl355_internal.h"
#include "blinding.cuh"

#include <errno.h>
#include <new>
#include <sys/random.h>

using namespace gl355;

extern "C" int32_t gl355_fri_prove(gl355_ctx* h, const uint64_t* final_coeffs, uint32_t log_n, uint32_t rate_bits,
                                   uint32_t cap_height, const uint32_t* arity_bits, uint32_t n_layers, uint32_t pow_bits,
                                   uint32_t num_queries, gl355_challenger* ch, uint64_t* caps_out, uint64_t* final_poly_out,
                                   uint64_t* pow_witness, uint64_t* query_indices, uint64_t* step_evals,
                                   uint64_t* step_siblings) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!final_coeffs || !ch || !caps_out || !final_poly_out || !pow_witness || !query_indices || !step_evals || !step_siblings)
        return ctx->fail(GL355_E_INVALID_ARG, "fri_prove: null argument");
    const uint32_t lde_bits = log_n + rate_bits;
    if (log_n == 0 || lde_bits > 27 || n_layers > 32 || n_layers > log_n) return ctx->fail(GL355_E_UNSUPPORTED, "fri_prove: unsupported size");
    for (uint32_t l = 0; l < n_layers; l++)
        if (arity_bits[l] != 1) return ctx->fail(GL355_E_UNSUPPORTED, "fri_prove: arity-2 folding only (fri_chip.rs:211)");
    if (lde_bits - n_layers < cap_height + 0u) return ctx->fail(GL355_E_INVALID_ARG, "fri_prove: too many layers for the cap height");
    const uint64_t n = 1ull << log_n, N = 1ull << lde_bits, n_cap = 1ull << cap_height;

    // device buffers: coefficients (N ext, zero padded), values (N ext), per-layer leaves + digests
    Staged sc(ctx);
    GL355_TRY(sc.open(final_coeffs, n * 16, 1));
    Scratch work(ctx), trees(ctx);
    GL355_TRY(work.get((2 * N + 2 * N + 2 * N) * 8 + 64));
    uint64_t* coeffs = work.as<uint64_t>();       // 2N u64
    uint64_t* values = coeffs + 2 * N;           // 2N u64
    uint64_t* coeffs2 = values + 2 * N;          // fold target
    // tree storage: layer l has N/2^(l+1) leaves of 4 u64 and 2*(leaves - n_cap) digests of 4 u64
    std::vector<uint64_t> leaf_off(n_layers), dig_off(n_layers);
    uint64_t total = 0;
    for (uint32_t l = 0; l < n_layers; l++) {
        const uint64_t nl = N >> (l + 1);
        if (nl < n_cap) return ctx->fail(GL355_E_INVALID_ARG, "fri_prove: layer smaller than the cap");
        leaf_off[l] = total; total += nl * 4;
        dig_off[l] = total; total += 2 * (nl - n_cap) * 4;
    }
    GL355_TRY(trees.get((total + n_cap * 4 + 16) * 8));
    uint64_t* tree_buf = trees.as<uint64_t>();
    uint64_t* d_cap = tree_buf + total;

    GL355_HIP(ctx, hipMemcpyAsync(coeffs, sc.as<uint64_t>(), n * 16, hipMemcpyDeviceToDevice, ctx->stream));
    GL355_HIP(ctx, hipMemsetAsync(coeffs + 2 * n, 0, (N - n) * 16, ctx->stream));
    GL355_TRY(lde_ext_dev(ctx, coeffs, log_n, rate_bits, GL355_COSET_SHIFT, values, false));

    uint64_t shift = GL355_COSET_SHIFT;
    uint64_t len = N;  // current number of ext coefficients / values
    std::vector<uint64_t> cap_host(n_cap * 4);
    for (uint32_t l = 0; l < n_layers; l++) {
        uint64_t* lv = tree_buf + leaf_off[l];
        uint64_t* dg = tree_buf + dig_off[l];
        GL355_TRY(fri_layer_leaves_dev(ctx, values, len, lv));
        GL355_TRY(merkle_build_any(ctx, ch->hasher, lv, len / 2, 4, false, 0, cap_height, dg, d_cap));
        GL355_HIP(ctx, ctx->d2h(cap_host.data(), d_cap, n_cap * 32));
        GL355_HIP(ctx, ctx->wait());
        memcpy(caps_out + (uint64_t)l * n_cap * 4, cap_host.data(), n_cap * 32);
        gl355_challenger_observe(ch, cap_host.data(), n_cap * 4);
        uint64_t beta[2];
        gl355_challenger_squeeze(ch, beta, 2);
        GL355_TRY(fri_fold_dev(ctx, coeffs, len, beta, coeffs2));
        std::swap(coeffs, coeffs2);
        len >>= 1;
        shift = gl_mul(shift, shift);
        if (l + 1 < n_layers) {
            const uint32_t lg = log2_u64(len);
            GL355_TRY(lde_ext_dev(ctx, coeffs, lg, 0, gl_canon(shift), values, false));
        }
    }
    // final polynomial: the upper (1 - 2^-rate_bits) of the coefficients is zero by construction
    const uint64_t final_len = len >> rate_bits;
    GL355_HIP(ctx, ctx->d2h(final_poly_out, coeffs, final_len * 16));
    GL355_HIP(ctx, ctx->wait());
    gl355_challenger_observe(ch, final_poly_out, final_len * 2);
    // proof of work
    uint64_t st[12];
    uint32_t pos;
    if (gl355_challenger_pow_state(ch, st, &pos) != GL355_OK) return ctx->fail(GL355_E_INVALID_ARG, "fri_prove: challenger state");
    GL355_TRY(pow_grind_any(ctx, ch->hasher, st, pos, pow_bits, 0, pow_witness));
    gl355_challenger_observe(ch, pow_witness, 1);
    uint64_t resp;
    gl355_challenger_squeeze(ch, &resp, 1);
    if (pow_bits && (resp >> (64 - pow_bits)) != 0) return ctx->fail(GL355_E_HIP, "fri_prove: proof-of-work response check failed");
    // queries: x_index = challenge mod N; layer l opens index x_index >> (l + 1)
    gl355_challenger_squeeze(ch, query_indices, num_queries);
    for (uint32_t q = 0; q < num_queries; q++) query_indices[q] &= (N - 1);
    uint64_t sib_total = 0;
    std::vector<uint64_t> sib_off(n_layers);
    for (uint32_t l = 0; l < n_layers; l++) { sib_off[l] = sib_total; sib_total += (uint64_t)(lde_bits - 1 - l - cap_height) * 4; }
    Scratch outb(ctx);
    GL355_TRY(outb.get(((uint64_t)num_queries * (1 + n_layers * 4 + sib_total) + 16) * 8));
    uint64_t* d_idx = outb.as<uint64_t>();
    uint64_t* d_ev = d_idx + num_queries;
    uint64_t* d_sib = d_ev + (uint64_t)num_queries * n_layers * 4;
    GL355_HIP(ctx, hipMemcpyAsync(d_idx, query_indices, (uint64_t)num_queries * 8, hipMemcpyHostToDevice, ctx->stream));
    for (uint32_t l = 0; l < n_layers; l++) {
        GL355_TRY(open_batch_ex_dev(ctx, tree_buf + leaf_off[l], 0, 4, tree_buf + dig_off[l], lde_bits - 1 - l, cap_height, d_idx,
                                    l + 1, num_queries, d_ev + (uint64_t)l * 4, (uint64_t)n_layers * 4, d_sib + sib_off[l], sib_total));
    }
    GL355_HIP(ctx, ctx->d2h(step_evals, d_ev, (uint64_t)num_queries * n_layers * 32));
    if (sib_total) GL355_HIP(ctx, ctx->d2h(step_siblings, d_sib, (uint64_t)num_queries * sib_total * 8));
    GL355_HIP(ctx, ctx->wait());
    return GL355_OK;
}

// ================================================================================================
// gl355_prove: CircuitData::prove in one call (src/plonky2_semaphore/access_set.rs:94, recursion.rs:168,
// wrapper.rs:55).  Stage order and transcript: chip/plonk/plonk_verifier_chip.rs:55-154; oracle and
// opening order: types/common_data.rs:100-222, types/assigned.rs:26-44.  Everything between the
// witness upload and the final proof download stays resident in HBM; the host only runs the
// Challenger (about 50 permutations).  The proof is returned as one flat u64 buffer (layout in
// include/gl355.h), which doubles as the wire format between aggregation levels.
// ================================================================================================
namespace gl355 {

// salt columns of a blinded oracle: thread b writes the four elements of ChaCha20 block b of `stream` (blinding.cuh)
__global__ void salt_kernel(uint64_t* out, uint64_t n, BlindKey key, uint32_t stream) {
    const uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (4 * b >= n) return;
    uint64_t e[4];
    blind_block_elements(key, stream, (uint32_t)b, e);
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (4 * b + j < n) out[4 * b + j] = e[j];
}

// key == NULL: a fresh 256-bit key from the OS CSPRNG for this proof (what plonky2's OsRng does); else the caller's key
int32_t resolve_blinding_key(Ctx* ctx, const uint8_t* key, BlindKey* out) {
    uint8_t buf[32];
    if (!key) {
        size_t got = 0;
        while (got < sizeof buf) {
            const ssize_t r = getrandom(buf + got, sizeof buf - got, 0);
            if (r < 0) { if (errno == EINTR) continue; return ctx->fail(GL355_E_UNSUPPORTED, "getrandom failed: no blinding key"); }
            got += (size_t)r;
        }
        key = buf;
    }
    *out = blind_key_from_bytes(key);
    return GL355_OK;
}

int32_t resolve_blinding_key_words(Ctx* ctx, const uint8_t* key, uint32_t out[8]) {
    BlindKey k;
    GL355_TRY(resolve_blinding_key(ctx, key, &k));
    memcpy(out, k.w, 32);
    return GL355_OK;
}

}  // namespace gl355

extern "C" uint64_t gl355_proof_words(const gl355_prover_data* pd) {
    if (!pd || !pd->circuit) return 0;
    const gl355_circuit& c = *pd->circuit;
    const uint64_t n_cap = 1ull << pd->cap_height;
    const uint32_t nch = c.num_challenges, qdf = c.max_degree;
    const uint32_t lde_bits = c.degree_bits + c.rate_bits;
    const uint32_t widths[4] = {c.num_selectors + c.num_constants + c.num_routed_wires, c.num_wires,
                                nch * (1 + c.num_partial_products), nch * qdf};
    uint64_t w = 8;                                            // header
    w += 3 * n_cap * 4;                                        // wires / zs / quotient caps
    uint64_t n_open = 0;
    for (int o = 0; o < 4; o++) n_open += widths[o];
    w += 2 * (n_open + nch);                                   // openings at zeta, Z at g*zeta (ext)
    w += (uint64_t)pd->n_fri_layers * n_cap * 4;               // commit-phase caps
    w += 2 * ((1ull << c.degree_bits) >> pd->n_fri_layers);    // final polynomial (ext)
    w += 1;                                                    // pow witness
    uint64_t per_q = 1;
    for (int o = 0; o < 4; o++) {
        const uint32_t leaf = widths[o] + ((pd->zero_knowledge && o > 0) ? GL355_SALT_SIZE : 0);
        per_q += leaf + (uint64_t)(lde_bits - pd->cap_height) * 4;
    }
    for (uint32_t l = 0; l < pd->n_fri_layers; l++) per_q += 4 + (uint64_t)(lde_bits - 1 - l - pd->cap_height) * 4;
    w += per_q * pd->num_queries;
    return w;
}

// ---- single-proof entries: one unit through the lock-step prover (prover_batch.hip) ---------------------------------------
extern "C" int32_t gl355_prove(gl355_ctx* h, const gl355_prover_data* pd, const uint64_t* wires, const uint64_t* public_inputs,
                               uint32_t n_public_inputs, const uint8_t* blinding_key, uint64_t* proof, uint64_t proof_capacity_words) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!pd || !pd->circuit || !wires) return ctx->fail(GL355_E_INVALID_ARG, "prove: null argument");
    Staged s_wires(ctx);
    GL355_TRY(s_wires.open(wires, ((uint64_t)pd->circuit->num_wires << pd->circuit->degree_bits) * 8, 1));
    ProveUnit io{public_inputs, n_public_inputs, {0}, proof, proof_capacity_words};
    GL355_TRY(resolve_blinding_key_words(ctx, blinding_key, io.key));
    try {
        return prove_units(ctx, pd, 1, s_wires.as<uint64_t>(), nullptr, nullptr, 0, 0, 0, 0, 0, &io);
    } catch (const std::bad_alloc&) {
        return ctx->fail(GL355_E_OOM, "prove: out of host memory");
    } catch (...) {
        return ctx->fail(GL355_E_HIP, "prove: unexpected host failure");
    }
}

// Witness given as its non-zero rows only (the rest of the 2^degree_bits rows are Noop rows): rows[r] lists
// all num_wires values of circuit row row_idx[r].  The zero-knowledge blinding rows are filled on the device:
// rows [blind_start, blind_start+n_blind) get random values on every wire, and n_z_pairs consecutive row
// pairs starting at z_start carry, on every routed wire, one random value shared by the two rows of the pair (the builder
// copy-constrains each column between them: plonky2 `blind`).
extern "C" int32_t gl355_prove_sparse(gl355_ctx* h, const gl355_prover_data* pd, const uint32_t* row_idx, const uint64_t* rows,
                                      uint32_t n_rows, uint32_t blind_start, uint32_t n_blind, uint32_t z_start, uint32_t n_z_pairs,
                                      const uint64_t* public_inputs, uint32_t n_public_inputs, const uint8_t* blinding_key, uint64_t* proof,
                                      uint64_t proof_capacity_words) {
    return gl355_prove_sparse_units(h, pd, 1, row_idx, rows, n_rows, blind_start, n_blind, z_start, n_z_pairs, public_inputs, n_public_inputs,
                                    blinding_key, proof, proof_capacity_words);
}

// the same for n_units independent witnesses of ONE circuit, proven in lock-step on this context (every stage's kernels carry
// a unit dimension; per-unit transcripts, keys and proofs): rows = [n_units][n_rows][num_wires], public_inputs = [n_units][n_public_inputs],
// proofs = [n_units][proof_capacity_words], blinding_keys = [n_units][32] or NULL (a fresh OS-random key for every unit)
extern "C" int32_t gl355_prove_sparse_units(gl355_ctx* h, const gl355_prover_data* pd, uint32_t n_units, const uint32_t* row_idx, const uint64_t* rows,
                                            uint32_t n_rows, uint32_t blind_start, uint32_t n_blind, uint32_t z_start, uint32_t n_z_pairs,
                                            const uint64_t* public_inputs, uint32_t n_public_inputs, const uint8_t* blinding_keys,
                                            uint64_t* proofs, uint64_t proof_capacity_words) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!pd || !pd->circuit || (!row_idx && n_rows) || (!rows && n_rows) || !proofs || (!public_inputs && n_public_inputs))
        return ctx->fail(GL355_E_INVALID_ARG, "prove_sparse: null argument");
    if (n_units == 0 || n_units > GL355_MAX_UNITS) return ctx->fail(GL355_E_INVALID_ARG, "prove_sparse: 1..GL355_MAX_UNITS units per call");
    const uint64_t n = 1ull << pd->circuit->degree_bits;
    if ((uint64_t)blind_start + n_blind > n || (uint64_t)z_start + 2ull * n_z_pairs > n) return ctx->fail(GL355_E_INVALID_ARG, "prove_sparse: blinding rows out of range");
    for (uint32_t r = 0; r < n_rows; r++)
        if (row_idx[r] >= n) return ctx->fail(GL355_E_INVALID_ARG, "prove_sparse: row index out of range");
    ProveUnit io[GL355_MAX_UNITS];
    for (uint32_t u = 0; u < n_units; u++) {
        io[u] = ProveUnit{public_inputs ? public_inputs + (uint64_t)u * n_public_inputs : nullptr, n_public_inputs, {0},
                          proofs + (uint64_t)u * proof_capacity_words, proof_capacity_words};
        GL355_TRY(resolve_blinding_key_words(ctx, blinding_keys ? blinding_keys + 32ull * u : nullptr, io[u].key));
    }
    try {
        return prove_units(ctx, pd, n_units, nullptr, row_idx, rows, n_rows, blind_start, n_blind, z_start, n_z_pairs, io);
    } catch (const std::bad_alloc&) {
        return ctx->fail(GL355_E_OOM, "prove: out of host memory");
    } catch (...) {
        return ctx->fail(GL355_E_HIP, "prove: unexpected host failure");      // nothing is thrown across the C boundary
    }
}

// ---- blinding-stream surface (blinding.cuh) ----------------------------------------------------------------------------
// per-unit key of a batch: first 32 bytes of ChaCha20 block 0 under the batch key with nonce ("key", index_lo, index_hi)
extern "C" int32_t gl355_derive_key(const uint8_t base_key[32], uint64_t index, uint8_t out[32]) {
    if (!base_key || !out) return GL355_E_INVALID_ARG;
    uint32_t o[16];
    chacha20_block(blind_key_from_bytes(base_key), 0, GL355_BLIND_NONCE_KEY, (uint32_t)index, (uint32_t)(index >> 32), o);
    for (int i = 0; i < 8; i++)
        for (int b = 0; b < 4; b++) out[4 * i + b] = (uint8_t)(o[i] >> (8 * b));
    return GL355_OK;
}
// the first `count` field elements of blinding stream `stream` under `key`, produced by the device kernel the prover uses
extern "C" int32_t gl355_blinding_elements(gl355_ctx* h, const uint8_t key[32], uint32_t stream, uint64_t count, uint64_t* out) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!key || (!out && count)) return ctx->fail(GL355_E_INVALID_ARG, "blinding_elements: null argument");
    if (count == 0) return GL355_OK;
    if (count > (1ull << 34)) return ctx->fail(GL355_E_UNSUPPORTED, "blinding_elements: a stream holds 2^34 elements");
    Staged so(ctx);
    GL355_TRY(so.open(out, count * 8, 2));
    hipLaunchKernelGGL(salt_kernel, dim3((uint32_t)((count / 4 + 256) / 256)), dim3(256), 0, ctx->stream, so.as<uint64_t>(), count,
                       blind_key_from_bytes(key), stream);
    GL355_HIP(ctx, hipGetLastError());
    return so.finish();
}
