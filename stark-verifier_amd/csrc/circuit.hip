// Circuit artifacts: a built circuit (shape, selector / constant / sigma tables, FRI parameters, blinding rows, the sparse
// witness-row map and, for the recursive verifier, the witness tape) serialised once by the host-side builder and loaded
// here, so that the per-proof path is native end to end:
//   gl355_semaphore_prove      = fill_semaphore_targets + data.prove      (src/plonky2_semaphore/access_set.rs:61-104)
//   gl355_circuit_prove_tape   = set_proof_with_pis_target + data.prove   (recursion.rs:72-86,167-168; wrapper.rs:49-55)
// Building a circuit (plonky2's CircuitBuilder::build, access_set.rs:91, recursion.rs:167, wrapper.rs:41) happens once per
// circuit shape and stays on the host-side builder; proving happens per signal and needs nothing but this file's calls.
// A loaded circuit is read-only and may be used concurrently by every context of its device.
#include "gl355_internal.h"

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

using namespace gl355;

struct gl355_circuit_handle {
    Ctx* owner = nullptr;
    gl355_circuit c;
    gl355_prover_data pd;
    gl355_oracle* cs = nullptr;
    uint64_t* d_sigmas = nullptr;
    uint64_t* d_kis = nullptr;
    uint64_t* d_tape = nullptr;        // device copy of the tape | segment starts | public-input positions (witness_tape_dev.hip)
    uint64_t* d_seg_start = nullptr;
    uint64_t* d_pi_pos = nullptr;
    std::vector<uint32_t> row_idx;
    std::vector<uint64_t> pi_pos, tape, seg_lens;
    std::vector<uint64_t> cs_cap, k_is_host;     // verifier data (gl355_circuit_verify)
    uint64_t n_seq = 0;
    uint32_t blind_start = 0, n_blind = 0, z_start = 0, n_z_pairs = 0, n_pi = 0;
    uint64_t n_inputs = 0;
};

namespace {
constexpr uint64_t MAGIC = 0x5249433535334c47ull;   // "GL355CIR" little endian
constexpr uint64_t HDR = 112;
}

extern "C" {

int32_t gl355_circuit_load(gl355_ctx* h, const uint64_t* blob, uint64_t words, gl355_circuit_handle** out) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!blob || !out || words < HDR) return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: null or truncated artifact");
    *out = nullptr;
    if (blob[0] != MAGIC || (blob[1] != 2 && blob[1] != 3)) return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: not a version-2/3 gl355 circuit artifact");
    const bool external_digest = blob[1] == 3;      // the digest was computed elsewhere; the artifact carries the expected cap instead
    gl355_circuit c;
    memset(&c, 0, sizeof c);
    c.degree_bits = (uint32_t)blob[2]; c.rate_bits = (uint32_t)blob[3]; c.num_wires = (uint32_t)blob[4];
    c.num_routed_wires = (uint32_t)blob[5]; c.num_constants = (uint32_t)blob[6]; c.num_selectors = (uint32_t)blob[7];
    c.num_challenges = (uint32_t)blob[8]; c.max_degree = (uint32_t)blob[9]; c.num_partial_products = (uint32_t)blob[10];
    c.num_gates = (uint32_t)blob[11];
    if (c.num_gates > GL355_MAX_GATES || c.degree_bits == 0 || c.degree_bits > 24 || c.num_wires == 0 || c.num_wires > 1024 ||
        c.num_routed_wires > c.num_wires || c.num_selectors + c.num_constants > 64)
        return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: implausible circuit shape");
    for (uint32_t g = 0; g < GL355_MAX_GATES; g++) {
        const uint64_t* p = blob + 12 + 5 * g;
        c.gates[g].type = (uint32_t)p[0]; c.gates[g].param = (uint32_t)p[1]; c.gates[g].selector_index = (uint32_t)p[2];
        c.gates[g].group_start = (uint32_t)p[3]; c.gates[g].group_end = (uint32_t)p[4];
        if (g < c.num_gates && (c.gates[g].type > GL355_GATE_TYPE_MAX || c.gates[g].selector_index >= c.num_selectors ||
                                c.gates[g].group_end > c.num_gates || c.gates[g].group_start > c.gates[g].group_end))
            return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: bad gate table");
        // gate parameters are counts of per-gate operations (RANDOM_ACCESS: three packed bytes): more of them than wires cannot be laid
        // out, and a value >= 2^30 would wrap the 32-bit products (4 p, 8 p ...) evaluators form from it
        if (g < c.num_gates && (p[1] > 0xFFFFFFFFull || (c.gates[g].type != GL355_GATE_RANDOM_ACCESS && p[1] > c.num_wires) ||
                                (c.gates[g].type == GL355_GATE_RANDOM_ACCESS && p[1] > 0xFFFFFFull)))
            return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: gate parameter out of range");
    }
    const uint64_t n = 1ull << c.degree_bits;
    const uint64_t n_sc = c.num_selectors + c.num_constants, routed = c.num_routed_wires;
    const uint64_t n_rows = blob[102], n_ops = blob[103], n_inputs = blob[104], n_pi = blob[105];
    if (n_rows > n || n_ops > (1ull << 28) || n_pi > (1u << 20)) return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: implausible sizes");
    const uint64_t n_seq = blob[110], n_segs = blob[111];
    if (n_seq > n_ops || n_segs > n_ops) return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: implausible tape segmentation");
    if (blob[92] > 20) return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: implausible cap height");
    const uint64_t need = HDR + (n_sc + routed) * n + routed + n_rows + n_pi + 5 * n_ops + n_segs + (external_digest ? (4ull << blob[92]) : 0);
    if (words != need) return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: artifact length does not match its header");
    const int32_t hasher = (int32_t)blob[97];
    if (hasher != GL355_HASH_POSEIDON && hasher != GL355_HASH_BN254_POSEIDON) return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: unknown hasher");

    gl355_circuit_handle* ch = new (std::nothrow) gl355_circuit_handle();
    if (!ch) return GL355_E_OOM;
    ch->owner = ctx;
    ch->c = c;
    ch->blind_start = (uint32_t)blob[98]; ch->n_blind = (uint32_t)blob[99]; ch->z_start = (uint32_t)blob[100];
    ch->n_z_pairs = (uint32_t)blob[101]; ch->n_inputs = n_inputs; ch->n_pi = (uint32_t)n_pi;
    const uint64_t* p = blob + HDR;
    const uint64_t* cs_values = p; p += (n_sc + routed) * n;
    const uint64_t* sigmas = cs_values + n_sc * n;
    const uint64_t* k_is = p; p += routed;
    ch->row_idx.resize(n_rows);
    for (uint64_t i = 0; i < n_rows; i++) {
        if (p[i] >= n) { delete ch; return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: witness row index out of range"); }
        ch->row_idx[i] = (uint32_t)p[i];
    }
    p += n_rows;
    ch->pi_pos.assign(p, p + n_pi); p += n_pi;
    for (uint64_t v : ch->pi_pos)
        if (v >= n_rows * c.num_wires) { delete ch; return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: public-input position out of range"); }
    ch->tape.assign(p, p + 5 * n_ops); p += 5 * n_ops;
    ch->seg_lens.assign(p, p + n_segs);
    ch->n_seq = n_seq;
    {
        // no entry may exceed n_ops, so the running sum (checked after every step) cannot wrap around to n_ops
        uint64_t tot = n_seq;
        bool fits = n_seq <= n_ops;
        for (uint64_t v : ch->seg_lens) {
            if (v > n_ops || tot + v > n_ops) { fits = false; break; }
            tot += v;
        }
        if (!fits || tot != n_ops) { delete ch; return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: tape segments do not add up"); }
    }
    if ((uint64_t)ch->blind_start + ch->n_blind > n || (uint64_t)ch->z_start + 2ull * ch->n_z_pairs > n) {
        delete ch; return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: blinding rows out of range");
    }
    // every offset of the tape is checked here, once: the replays (host and device) of a loaded circuit cannot leave the rows
    if (n_ops) {
        const uint64_t bad = tape_validate(ch->tape.data(), n_ops, n_inputs, n_rows * c.num_wires, c.num_wires);
        if (bad != ~0ull) { delete ch; return ctx->fail(GL355_E_INVALID_ARG, "circuit_load: malformed witness tape"); }
    }
    // device copies of the sigma values / k_is (plain allocations: shared by every context of the device)
    int32_t rc = GL355_OK;
    do {
        if (n_ops) {
            std::vector<uint64_t> seg_start(n_segs + 1, n_seq);
            for (uint64_t k = 0; k < n_segs; k++) seg_start[k + 1] = seg_start[k] + ch->seg_lens[k];
            const uint64_t words = 5 * n_ops + (n_segs + 1) + n_pi;
            if (hipMalloc((void**)&ch->d_tape, words * 8 + 8) != hipSuccess) { rc = ctx->fail(GL355_E_OOM, "circuit_load: hipMalloc"); break; }
            ch->d_seg_start = ch->d_tape + 5 * n_ops;
            ch->d_pi_pos = ch->d_seg_start + (n_segs + 1);
            if (hipMemcpy(ch->d_tape, ch->tape.data(), 5 * n_ops * 8, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(ch->d_seg_start, seg_start.data(), (n_segs + 1) * 8, hipMemcpyHostToDevice) != hipSuccess ||
                (n_pi && hipMemcpy(ch->d_pi_pos, ch->pi_pos.data(), n_pi * 8, hipMemcpyHostToDevice) != hipSuccess)) { rc = ctx->fail(GL355_E_HIP, "circuit_load: upload"); break; }
        }
        if (hipMalloc(&ch->d_sigmas, routed * n * 8) != hipSuccess || hipMalloc(&ch->d_kis, routed * 8 + 8) != hipSuccess) { rc = ctx->fail(GL355_E_OOM, "circuit_load: hipMalloc"); break; }
        if (hipMemcpyAsync(ch->d_sigmas, sigmas, routed * n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(ch->d_kis, k_is, routed * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = ctx->fail(GL355_E_HIP, "circuit_load: upload"); break; }
        rc = gl355_commit_h(h, hasher, cs_values, c.degree_bits, (uint32_t)(n_sc + routed), c.rate_bits, 0, nullptr, (uint32_t)blob[92], &ch->cs);
        if (rc) break;
        // the digest covers the preprocessed commitment and the shape: recompute it, a stale or foreign artifact is refused
        const uint64_t n_cap = 1ull << blob[92];
        std::vector<uint64_t> pre(n_cap * 4 + 3 + c.num_gates);
        rc = gl355_oracle_cap(ch->cs, pre.data());
        if (rc) break;
        ch->cs_cap.assign(pre.begin(), pre.begin() + n_cap * 4);
        ch->k_is_host.assign(k_is, k_is + routed);
        uint64_t* s = pre.data() + n_cap * 4;
        s[0] = c.degree_bits; s[1] = c.num_gates; s[2] = c.num_selectors;
        for (uint32_t g = 0; g < c.num_gates; g++) s[3 + g] = ((uint64_t)c.gates[g].type << 32) | c.gates[g].param;   // no aliasing between (type, param) pairs
        if (external_digest) {
            // version 3: the transcript uses the digest as given (e.g. plonky2's own); what ties the tables to it is the commitment
            if (memcmp(pre.data(), blob + words - n_cap * 4, n_cap * 32) != 0) { rc = ctx->fail(GL355_E_INVALID_ARG, "circuit_load: the artifact's constants_sigmas cap does not match its tables"); break; }
        } else {
            uint64_t dg[4];
            gl355_host_hash_no_pad_h(hasher, pre.data(), pre.size(), dg);
            if (memcmp(dg, blob + 106, 32) != 0) { rc = ctx->fail(GL355_E_INVALID_ARG, "circuit_load: circuit digest of the artifact does not match its tables"); break; }
        }
    } while (0);
    if (rc != GL355_OK) {
        if (ch->cs) gl355_oracle_destroy(ch->cs);
        if (ch->d_sigmas) (void)hipFree(ch->d_sigmas);
        if (ch->d_kis) (void)hipFree(ch->d_kis);
        if (ch->d_tape) (void)hipFree(ch->d_tape);
        delete ch;
        return rc;
    }
    memset(&ch->pd, 0, sizeof ch->pd);
    ch->pd.circuit = &ch->c;
    ch->pd.constants_sigmas = ch->cs;
    ch->pd.sigmas = ch->d_sigmas;
    ch->pd.k_is = ch->d_kis;
    memcpy(ch->pd.circuit_digest, blob + 106, 32);
    ch->pd.cap_height = (uint32_t)blob[92]; ch->pd.pow_bits = (uint32_t)blob[93]; ch->pd.num_queries = (uint32_t)blob[94];
    ch->pd.n_fri_layers = (uint32_t)blob[95]; ch->pd.zero_knowledge = (int32_t)blob[96]; ch->pd.hasher = hasher;
    *out = ch;
    return GL355_OK;
}

int32_t gl355_circuit_destroy(gl355_circuit_handle* ch) {
    if (!ch) return GL355_OK;
    if (ch->cs) gl355_oracle_destroy(ch->cs);
    (void)hipSetDevice(ch->owner->device);
    if (ch->d_sigmas) (void)hipFree(ch->d_sigmas);
    if (ch->d_kis) (void)hipFree(ch->d_kis);
    if (ch->d_tape) (void)hipFree(ch->d_tape);
    delete ch;
    return GL355_OK;
}

int32_t gl355_circuit_info(const gl355_circuit_handle* ch, uint64_t* proof_words, uint32_t* n_public_inputs, uint32_t* n_rows,
                           uint64_t* n_inputs, uint32_t* degree_bits) {
    if (!ch) return GL355_E_INVALID_ARG;
    if (proof_words) *proof_words = gl355_proof_words(&ch->pd);
    if (n_public_inputs) *n_public_inputs = ch->n_pi;
    if (n_rows) *n_rows = (uint32_t)ch->row_idx.size();
    if (n_inputs) *n_inputs = ch->n_inputs;
    if (degree_bits) *degree_bits = ch->c.degree_bits;
    return GL355_OK;
}
const uint64_t* gl355_circuit_digest(const gl355_circuit_handle* ch) { return ch ? ch->pd.circuit_digest : nullptr; }

int32_t gl355_circuit_verify(const gl355_circuit_handle* ch, const uint64_t* proof, uint64_t proof_words, const uint64_t* public_inputs,
                             uint32_t n_public_inputs) {
    if (!ch) return GL355_E_INVALID_ARG;
    gl355_verifier_data vd;
    memset(&vd, 0, sizeof vd);
    vd.circuit = &ch->c; vd.constants_sigmas_cap = ch->cs_cap.data(); vd.k_is = ch->k_is_host.data();
    memcpy(vd.circuit_digest, ch->pd.circuit_digest, 32);
    vd.cap_height = ch->pd.cap_height; vd.pow_bits = ch->pd.pow_bits; vd.num_queries = ch->pd.num_queries; vd.n_fri_layers = ch->pd.n_fri_layers;
    vd.zero_knowledge = ch->pd.zero_knowledge; vd.hasher = ch->pd.hasher;
    return gl355_verify(&vd, proof, proof_words, public_inputs, n_public_inputs);
}

int32_t gl355_circuit_prove_rows(gl355_ctx* h, const gl355_circuit_handle* ch, const uint64_t* rows, const uint64_t* public_inputs,
                                 uint32_t n_public_inputs, const uint8_t* blinding_key, uint64_t* proof, uint64_t proof_capacity_words) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (!ch || !rows) return ctx->fail(GL355_E_INVALID_ARG, "circuit_prove_rows: null argument");
    if (ctx->device != ch->owner->device) return ctx->fail(GL355_E_INVALID_ARG, "circuit_prove_rows: circuit was loaded on another device");
    return gl355_prove_sparse(h, &ch->pd, ch->row_idx.data(), rows, (uint32_t)ch->row_idx.size(), ch->blind_start, ch->n_blind, ch->z_start,
                              ch->n_z_pairs, public_inputs, n_public_inputs, blinding_key, proof, proof_capacity_words);
}
int32_t gl355_circuit_prove_rows_units(gl355_ctx* h, const gl355_circuit_handle* ch, uint32_t n_units, const uint64_t* rows, const uint64_t* public_inputs,
                                       uint32_t n_public_inputs, const uint8_t* blinding_keys, uint64_t* proofs) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (!ch || !rows || !proofs) return ctx->fail(GL355_E_INVALID_ARG, "circuit_prove_rows_units: null argument");
    if (ctx->device != ch->owner->device) return ctx->fail(GL355_E_INVALID_ARG, "circuit_prove_rows_units: circuit was loaded on another device");
    return gl355_prove_sparse_units(h, &ch->pd, n_units, ch->row_idx.data(), rows, (uint32_t)ch->row_idx.size(), ch->blind_start, ch->n_blind, ch->z_start,
                                    ch->n_z_pairs, public_inputs, n_public_inputs, blinding_keys, proofs, gl355_proof_words(&ch->pd));
}

// witness generation of n_units units: tape replays spread over the context's replay threads (one unit per thread at a time;
// a single unit still uses its segments in parallel)
static int32_t replay_units(uint32_t threads, const gl355_circuit_handle* ch, uint32_t n_units, const uint64_t* inputs, uint64_t* rows, uint64_t n_words,
                            uint64_t* failed_unit, uint64_t* failed_op) {
    const uint32_t nt = std::max<uint32_t>(1, std::min<uint32_t>(threads, n_units));
    if (n_units == 1 || nt == 1) {
        for (uint32_t u = 0; u < n_units; u++) {
            uint64_t f = 0;
            const int32_t rc = gl355_witness_replay_segmented(ch->tape.data(), ch->tape.size() / 5, ch->n_seq, ch->seg_lens.data(), (uint32_t)ch->seg_lens.size(),
                                                              n_units == 1 ? threads : 1, inputs + (uint64_t)u * ch->n_inputs, ch->n_inputs,
                                                              rows + (uint64_t)u * n_words, n_words, ch->c.num_wires, &f);
            if (rc != GL355_OK) { *failed_unit = u; *failed_op = f; return rc; }
        }
        return GL355_OK;
    }
    std::atomic<uint32_t> next{0};
    std::vector<int32_t> rcs(n_units, GL355_OK);
    std::vector<uint64_t> fails(n_units, 0);
    auto worker = [&]() {
        for (uint32_t u; (u = next.fetch_add(1)) < n_units;)
            rcs[u] = gl355_witness_replay_segmented(ch->tape.data(), ch->tape.size() / 5, ch->n_seq, ch->seg_lens.data(), (uint32_t)ch->seg_lens.size(), 1,
                                                    inputs + (uint64_t)u * ch->n_inputs, ch->n_inputs, rows + (uint64_t)u * n_words, n_words, ch->c.num_wires,
                                                    &fails[u]);
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < nt; t++) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
    for (uint32_t u = 0; u < n_units; u++)
        if (rcs[u] != GL355_OK) { *failed_unit = u; *failed_op = fails[u]; return rcs[u]; }
    return GL355_OK;
}

}  // extern "C"
// for the batch runtime (batch.cpp), which overlaps the witness generation of one batch with the proving of the previous one
namespace gl355 {
uint64_t circuit_rows_words(const gl355_circuit_handle* ch) { return (uint64_t)ch->row_idx.size() * ch->c.num_wires; }
// device scratch of one device replay: inputs | status | public inputs
uint64_t circuit_replay_aux_bytes(const gl355_circuit_handle* ch, uint32_t n_units) { return (uint64_t)n_units * (ch->n_inputs + 1 + ch->n_pi) * 8 + 64; }
int32_t circuit_replay_units_dev(const gl355_circuit_handle* ch, int device, hipStream_t stream, uint32_t n_units, const uint64_t* inputs, uint64_t* d_rows,
                                 void* d_aux, uint64_t* pis_out, uint64_t* failed_unit, uint64_t* failed_op, void* h_aux) {
    if (!ch->d_tape) return GL355_E_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return GL355_E_HIP;
    uint64_t* d_inputs = reinterpret_cast<uint64_t*>(d_aux);
    uint64_t* d_status = d_inputs + (uint64_t)n_units * ch->n_inputs;
    uint64_t* d_pis = d_status + n_units;
    // h_aux: a pinned host mirror of d_aux (circuit_replay_aux_bytes) -- the batch runtime's side stream; nullptr: the caller's pageable memory
    const uint64_t* src = inputs;
    std::vector<uint64_t> back_pageable;
    uint64_t* back;
    if (h_aux) {
        uint64_t* h_inputs = reinterpret_cast<uint64_t*>(h_aux);
        memcpy(h_inputs, inputs, (size_t)n_units * ch->n_inputs * 8);
        src = h_inputs;
        back = h_inputs + (uint64_t)n_units * ch->n_inputs;
    } else {
        back_pageable.resize((size_t)n_units * (1 + ch->n_pi));
        back = back_pageable.data();
    }
    if (hipMemcpyAsync(d_inputs, src, (size_t)n_units * ch->n_inputs * 8, hipMemcpyHostToDevice, stream) != hipSuccess) return GL355_E_HIP;
    GL355_TRY(tape_replay_dev(stream, ch->d_tape, ch->d_seg_start, ch->n_seq, (uint32_t)ch->seg_lens.size(), n_units, d_inputs, ch->n_inputs, d_rows,
                              circuit_rows_words(ch), ch->d_pi_pos, ch->n_pi, d_status, d_pis));
    if (hipMemcpyAsync(back, d_status, (size_t)n_units * (1 + ch->n_pi) * 8, hipMemcpyDeviceToHost, stream) != hipSuccess) return GL355_E_HIP;
    for (;;) {                        // wait for the side stream without holding a core
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return GL355_E_HIP;
        (void)hipGetLastError();
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    for (uint32_t u = 0; u < n_units; u++)
        if (back[u] != ~0ull) { *failed_unit = u; *failed_op = back[u]; return GL355_E_WITNESS; }
    memcpy(pis_out, back + n_units, (size_t)n_units * ch->n_pi * 8);
    return GL355_OK;
}
int32_t circuit_replay_units(const gl355_circuit_handle* ch, uint32_t threads, uint32_t n_units, const uint64_t* inputs, uint64_t* rows, uint64_t* pis_out,
                             uint64_t* failed_unit, uint64_t* failed_op) {
    const uint64_t n_words = circuit_rows_words(ch);
    GL355_TRY(replay_units(threads, ch, n_units, inputs, rows, n_words, failed_unit, failed_op));
    for (uint32_t u = 0; u < n_units; u++)
        for (uint32_t i = 0; i < ch->n_pi; i++) pis_out[(size_t)u * ch->n_pi + i] = rows[(uint64_t)u * n_words + ch->pi_pos[i]];
    return GL355_OK;
}
}  // namespace gl355
extern "C" {

int32_t gl355_circuit_witness_rows(gl355_ctx* h, const gl355_circuit_handle* ch, uint32_t n_units, const uint64_t* inputs, uint64_t n_inputs,
                                   int32_t on_device, uint64_t* rows, uint64_t* public_inputs_out, uint64_t* failed_entry) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (!ch || !inputs || !rows) return ctx->fail(GL355_E_INVALID_ARG, "circuit_witness_rows: null argument");
    if (ch->tape.empty()) return ctx->fail(GL355_E_INVALID_ARG, "circuit_witness_rows: this artifact carries no witness tape");
    if (n_inputs != ch->n_inputs || n_units == 0 || n_units > GL355_MAX_UNITS) return ctx->fail(GL355_E_INVALID_ARG, "circuit_witness_rows: bad input shape");
    const uint64_t n_words = circuit_rows_words(ch);
    std::vector<uint64_t> pis((size_t)n_units * ch->n_pi + 1);
    uint64_t fu = 0, fo = ~0ull;
    int32_t rc;
    if (!on_device) {
        rc = circuit_replay_units(ch, ctx->replay_threads, n_units, inputs, rows, pis.data(), &fu, &fo);
    } else {
        if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
        Scratch d_rows(ctx), aux(ctx);
        GL355_TRY(d_rows.get(n_units * n_words * 8));
        GL355_TRY(aux.get(circuit_replay_aux_bytes(ch, n_units)));
        rc = circuit_replay_units_dev(ch, ctx->device, ctx->stream, n_units, inputs, d_rows.as<uint64_t>(), aux.p, pis.data(), &fu, &fo);
        if (rc == GL355_OK || rc == GL355_E_WITNESS) {
            GL355_HIP(ctx, ctx->d2h(rows, d_rows.p, n_units * n_words * 8));
            GL355_HIP(ctx, ctx->wait());
        }
    }
    if (failed_entry) *failed_entry = fo;
    if (rc != GL355_OK) return ctx->fail(rc, "circuit_witness_rows: the inputs do not satisfy the circuit");
    if (public_inputs_out) memcpy(public_inputs_out, pis.data(), (size_t)n_units * ch->n_pi * 8);
    return GL355_OK;
}

int32_t gl355_circuit_prove_tape_units(gl355_ctx* h, const gl355_circuit_handle* ch, uint32_t n_units, const uint64_t* inputs, uint64_t n_inputs,
                                       const uint8_t* blinding_keys, uint64_t* proofs, uint64_t* public_inputs_out) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (!ch || !inputs || !proofs) return ctx->fail(GL355_E_INVALID_ARG, "circuit_prove_tape: null argument");
    if (ch->tape.empty()) return ctx->fail(GL355_E_INVALID_ARG, "circuit_prove_tape: this artifact carries no witness tape");
    if (n_inputs != ch->n_inputs) return ctx->fail(GL355_E_INVALID_ARG, "circuit_prove_tape: wrong number of input words");
    if (n_units == 0 || n_units > GL355_MAX_UNITS) return ctx->fail(GL355_E_INVALID_ARG, "circuit_prove_tape: 1..GL355_MAX_UNITS units per call");
    const uint64_t n_words = (uint64_t)ch->row_idx.size() * ch->c.num_wires;
    std::vector<uint64_t> pis((size_t)ch->n_pi * n_units);
    uint64_t failed_unit = 0, failed_op = 0;
    auto witness_error = [&](int32_t rc) {
        char msg[128];
        snprintf(msg, sizeof msg, "circuit_prove_tape: witness generation of unit %llu failed at tape entry %llu", (unsigned long long)failed_unit,
                 (unsigned long long)failed_op);
        return ctx->fail(rc, msg);
    };
    if (ctx->device_replay && ch->d_tape && n_units > 1) {
        // witness rows generated where the prover reads them (witness_tape_dev.hip): the host uploads the inputs only.  A single unit
        // keeps the host replay: its latency (2-7 ms) beats the interpreter's (~20 ms, one lane for the sequential part)
        if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
        Scratch d_rows(ctx), aux(ctx);
        GL355_TRY(d_rows.get(n_units * n_words * 8));
        GL355_TRY(aux.get(circuit_replay_aux_bytes(ch, n_units)));
        const int32_t rc = circuit_replay_units_dev(ch, ctx->device, ctx->stream, n_units, inputs, d_rows.as<uint64_t>(), aux.p, pis.data(), &failed_unit, &failed_op);
        if (rc != GL355_OK) return rc == GL355_E_WITNESS ? witness_error(rc) : ctx->fail(rc, "circuit_prove_tape: device witness generation failed");
        if (public_inputs_out) memcpy(public_inputs_out, pis.data(), pis.size() * 8);
        return gl355_circuit_prove_rows_units(h, ch, n_units, d_rows.as<uint64_t>(), pis.data(), ch->n_pi, blinding_keys, proofs);
    }
    static thread_local std::vector<uint64_t> rows;      // one prover thread per context: reuse the row buffer
    rows.resize(n_words * n_units);
    const int32_t rc = circuit_replay_units(ch, ctx->replay_threads, n_units, inputs, rows.data(), pis.data(), &failed_unit, &failed_op);
    if (rc != GL355_OK) return witness_error(rc);
    if (public_inputs_out) memcpy(public_inputs_out, pis.data(), pis.size() * 8);
    return gl355_circuit_prove_rows_units(h, ch, n_units, rows.data(), pis.data(), ch->n_pi, blinding_keys, proofs);
}
int32_t gl355_circuit_prove_tape(gl355_ctx* h, const gl355_circuit_handle* ch, const uint64_t* inputs, uint64_t n_inputs, const uint8_t* blinding_key,
                                 uint64_t* proof, uint64_t proof_capacity_words, uint64_t* public_inputs_out) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (!ch || !proof) return ctx->fail(GL355_E_INVALID_ARG, "circuit_prove_tape: null argument");
    if (proof_capacity_words < gl355_proof_words(&ch->pd)) return ctx->fail(GL355_E_INVALID_ARG, "prove: proof buffer too small (see gl355_proof_words)");
    return gl355_circuit_prove_tape_units(h, ch, 1, inputs, n_inputs, blinding_key, proof, public_inputs_out);
}

int32_t gl355_semaphore_prove_units(gl355_ctx* h, const gl355_circuit_handle* ch, uint32_t n_units, const uint64_t* private_keys, const uint64_t* topics,
                                    const uint64_t* indices, const uint64_t* siblings, uint32_t height, const uint8_t* blinding_keys, uint64_t* proofs,
                                    uint64_t* public_inputs_out) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (!ch || !private_keys || !topics || !indices || !proofs || (!siblings && height)) return ctx->fail(GL355_E_INVALID_ARG, "semaphore_prove: null argument");
    if (ch->row_idx.size() != (size_t)height + 7 || ch->c.num_wires != 135)
        return ctx->fail(GL355_E_INVALID_ARG, "semaphore_prove: the artifact is not the Semaphore circuit of this tree height");
    if (n_units == 0 || n_units > GL355_MAX_UNITS) return ctx->fail(GL355_E_INVALID_ARG, "semaphore_prove: 1..GL355_MAX_UNITS units per call");
    const size_t per = (size_t)(height + 7) * 135;
    std::vector<uint64_t> rows(per * n_units), pis(12 * (size_t)n_units);
    for (uint32_t u = 0; u < n_units; u++)
        GL355_TRY(gl355_semaphore_witness(private_keys + 4 * u, topics + 4 * u, indices[u], siblings + (size_t)u * height * 4, height, rows.data() + u * per,
                                          pis.data() + 12 * u));
    if (public_inputs_out) memcpy(public_inputs_out, pis.data(), pis.size() * 8);
    return gl355_circuit_prove_rows_units(h, ch, n_units, rows.data(), pis.data(), 12, blinding_keys, proofs);
}
int32_t gl355_semaphore_prove(gl355_ctx* h, const gl355_circuit_handle* ch, const uint64_t private_key[4], const uint64_t topic[4],
                              uint64_t index, const uint64_t* siblings, uint32_t height, const uint8_t* blinding_key, uint64_t* proof,
                              uint64_t proof_capacity_words, uint64_t public_inputs_out[12]) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (!ch || !proof) return ctx->fail(GL355_E_INVALID_ARG, "semaphore_prove: null argument");
    if (proof_capacity_words < gl355_proof_words(&ch->pd)) return ctx->fail(GL355_E_INVALID_ARG, "prove: proof buffer too small (see gl355_proof_words)");
    return gl355_semaphore_prove_units(h, ch, 1, private_key, topic, &index, siblings, height, blinding_key, proof, public_inputs_out);
}

}  // extern "C"
