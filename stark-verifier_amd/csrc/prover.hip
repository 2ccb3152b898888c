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
        GL355_TRY(merkle_build_dev(ctx, lv, len / 2, 4, false, 0, cap_height, dg, d_cap));
        GL355_HIP(ctx, hipMemcpyAsync(cap_host.data(), d_cap, n_cap * 32, hipMemcpyDeviceToHost, ctx->stream));
        GL355_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
    GL355_HIP(ctx, hipMemcpyAsync(final_poly_out, coeffs, final_len * 16, hipMemcpyDeviceToHost, ctx->stream));
    GL355_HIP(ctx, hipStreamSynchronize(ctx->stream));
    gl355_challenger_observe(ch, final_poly_out, final_len * 2);
    // proof of work
    uint64_t st[12];
    uint32_t pos;
    if (gl355_challenger_pow_state(ch, st, &pos) != GL355_OK) return ctx->fail(GL355_E_INVALID_ARG, "fri_prove: challenger state");
    GL355_TRY(pow_grind_dev(ctx, st, pos, pow_bits, 0, pow_witness));
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
    GL355_HIP(ctx, hipMemcpyAsync(step_evals, d_ev, (uint64_t)num_queries * n_layers * 32, hipMemcpyDeviceToHost, ctx->stream));
    if (sib_total) GL355_HIP(ctx, hipMemcpyAsync(step_siblings, d_sib, (uint64_t)num_queries * sib_total * 8, hipMemcpyDeviceToHost, ctx->stream));
    GL355_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GL355_OK;
}
