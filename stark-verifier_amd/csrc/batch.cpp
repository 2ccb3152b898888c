// Native batch runtime: the reference proves its signals from a rayon `par_iter` (src/plonky2_semaphore/recursion.rs:300-308:
// one `make_signal` per member) and verifies / aggregates them from `par_chunks_exact` (recursion.rs:211-227).  Here one host
// thread per prover context takes the next GL355_OPT_BATCH_UNITS units and proves them in lock-step: Merkle path of the member from the access-set tree, gl355_semaphore_prove,
// and -- when a verifier circuit is given -- gl355_circuit_prove_tape on (proof | public inputs).  No host-language code runs
// between the units; the caller gets the (nullifier | topic) leaf of every unit for the aggregation root.
#include "gl355_internal.h"

#include <atomic>
#include <chrono>
#include <future>
#include <thread>
#include <vector>

using namespace gl355;

extern "C" int32_t gl355_semaphore_units(gl355_ctx* const* ctxs, uint32_t n_ctx, const gl355_circuit_handle* sem, const gl355_circuit_handle* rec,
                                         const uint64_t* private_keys, uint64_t n_members, const uint64_t topic[4], const uint64_t* tree_digests,
                                         const uint64_t* member_indices, uint32_t count, const uint8_t* key_base, uint64_t* leaves_out,
                                         uint64_t* proofs_out, uint32_t* units_per_ctx) {
    if (!ctxs || n_ctx == 0 || !sem || !private_keys || !topic || (!tree_digests && n_members > 1) || (!member_indices && count) || !leaves_out)
        return GL355_E_INVALID_ARG;
    for (uint32_t t = 0; t < n_ctx; t++)
        if (!ctxs[t]) return GL355_E_INVALID_ARG;
    if (n_members == 0 || (n_members & (n_members - 1))) return ctx_of(ctxs[0])->fail(GL355_E_INVALID_ARG, "semaphore_units: the access set must hold a power-of-two number of members");
    uint32_t height = 0;
    while ((1ull << height) < n_members) height++;
    uint64_t sem_words = 0, rec_words = 0, rec_inputs = 0;
    uint32_t rec_pi = 0;
    GL355_TRY(gl355_circuit_info(sem, &sem_words, nullptr, nullptr, nullptr, nullptr));
    if (rec) {
        GL355_TRY(gl355_circuit_info(rec, &rec_words, &rec_pi, nullptr, &rec_inputs, nullptr));
        if (rec_inputs != sem_words + 12 || rec_pi != 12)
            return ctx_of(ctxs[0])->fail(GL355_E_INVALID_ARG, "semaphore_units: the verifier circuit does not take one Semaphore proof");
    }
    for (uint32_t j = 0; j < count; j++)
        if (member_indices[j] >= n_members) return ctx_of(ctxs[0])->fail(GL355_E_INVALID_ARG, "semaphore_units: member index out of range");
    const uint64_t out_words = rec ? rec_words : sem_words;
    if (units_per_ctx) for (uint32_t t = 0; t < n_ctx; t++) units_per_ctx[t] = 0;       // written on every exit path from here on
    std::atomic<int32_t> first_error{GL355_OK};
    std::atomic<uint32_t> next_unit{0};     // units are handed out a batch at a time: a context that finishes early takes the next batch
    // One context's loop is software-pipelined over its batches: while the tapes of batch k are replayed on host threads (the
    // recursive circuit's witness generation, ~7 ms of one core per unit), the context's stream proves the recursive proofs of
    // batch k-1 -- the stream never waits for the host phase of its own batch.
    // The witness rows (~6.4 MB per unit) are replayed into PINNED host memory and uploaded on the context's copy stream by the
    // same helper thread, so the proving stream receives them already resident (a pageable upload is staged in chunks on the
    // proving stream itself: ~5 ms per batch of 8 during which it computes nothing).
    struct Slot {
        uint32_t j0 = 0, nb = 0;
        std::vector<uint64_t> flat, pis, inputs, rpis, outer;
        uint64_t* rows = nullptr;      // pinned host: the witness rows (host replay) or the mirror of d_aux (device replay)
        uint64_t* d_rows = nullptr;    // device
        void* d_aux = nullptr;         // device replay: inputs | status | public inputs
        std::vector<uint8_t> k_rec;
        std::future<int32_t> replay;
        uint64_t failed_unit = 0, failed_op = 0;
    };
    auto worker = [&](uint32_t t) {
        Ctx* cx = ctx_of(ctxs[t]);
        const uint32_t B = std::max<uint32_t>(1, std::min<uint32_t>(cx->batch_units, GL355_MAX_UNITS));
        const uint64_t rec_row_words = rec ? circuit_rows_words(rec) : 0;
        std::vector<uint64_t> sib((size_t)B * height * 4 + 4), sks((size_t)B * 4), topics((size_t)B * 4), idxs(B);
        std::vector<uint8_t> k_sem((size_t)B * 32);
        Slot slots[2];
        hipStream_t copy_stream = nullptr;
        for (auto& s : slots) {
            s.flat.resize((size_t)B * sem_words); s.pis.resize((size_t)B * 12); s.k_rec.resize((size_t)B * 32);
            if (rec) { s.inputs.resize((size_t)B * (sem_words + 12)); s.rpis.resize((size_t)B * 12); s.outer.resize((size_t)B * rec_words); }
        }
        if (rec) {
            uint64_t *rows[2], *drows[2];
            void* aux[2];
            const int32_t rc = hipSetDevice(cx->device) == hipSuccess
                                   ? cx->runtime_buffers((size_t)B * rec_row_words * 8, cx->device_replay ? circuit_replay_aux_bytes(rec, B) : 0, rows, drows, aux, &copy_stream)
                                   : GL355_E_HIP;
            if (rc != GL355_OK) {
                int32_t expected = GL355_OK;
                first_error.compare_exchange_strong(expected, rc);
                if (units_per_ctx) units_per_ctx[t] = 0;
                return;
            }
            for (int i = 0; i < 2; i++) { slots[i].rows = rows[i]; slots[i].d_rows = drows[i]; slots[i].d_aux = aux[i]; }
        }
        uint32_t done = 0;
        auto fail = [&](int32_t rc) { int32_t expected = GL355_OK; first_error.compare_exchange_strong(expected, rc); };
        // second half of a batch: wait for its witness rows, prove the recursive proofs, hand out the results
        auto finish = [&](Slot& s) -> bool {
            const uint64_t* result = s.flat.data();
            const uint64_t* opis = s.pis.data();
            if (rec) {
                // the future is consumed here: on every failure below the slot is emptied (nb = 0), so neither a later iteration nor
                // the drain path can call get() a second time on an invalid future
                int32_t rc = s.replay.valid() ? s.replay.get() : GL355_E_INVALID_ARG;
                if (rc != GL355_OK) {
                    char msg[128];
                    snprintf(msg, sizeof msg, "semaphore_units: witness generation of unit %llu failed at tape entry %llu",
                             (unsigned long long)(s.j0 + s.failed_unit), (unsigned long long)s.failed_op);
                    cx->fail(rc, msg);
                    fail(rc);
                    s.nb = 0;
                    return false;
                }
                rc = gl355_circuit_prove_rows_units(ctxs[t], rec, s.nb, s.d_rows, s.rpis.data(), 12, key_base ? s.k_rec.data() : nullptr, s.outer.data());
                if (rc != GL355_OK) { fail(rc); s.nb = 0; return false; }
                result = s.outer.data(); opis = s.rpis.data();
            }
            for (uint32_t b = 0; b < s.nb; b++) {
                memcpy(leaves_out + 8ull * (s.j0 + b), opis + (size_t)b * 12 + 4, 64);            // nullifier | topic
                if (proofs_out) memcpy(proofs_out + (uint64_t)(s.j0 + b) * out_words, result + (uint64_t)b * out_words, out_words * 8);
            }
            done += s.nb;
            if (units_per_ctx) units_per_ctx[t] = done;       // kept current, so it is written on every exit path (incl. an exception)
            s.nb = 0;
            return true;
        };
        int cur = 0;
        bool ok = true;
        for (;;) {
            Slot& s = slots[cur];
            Slot& prev = slots[cur ^ 1];
            const uint32_t j0 = (ok && first_error.load() == GL355_OK) ? next_unit.fetch_add(B) : count;
            if (j0 < count) {
                const uint32_t nb = std::min<uint32_t>(B, count - j0);
                s.j0 = j0; s.nb = nb;
                for (uint32_t b = 0; b < nb; b++) {
                    const uint32_t j = j0 + b;
                    const uint64_t idx = member_indices[j];
                    idxs[b] = idx;
                    memcpy(&sks[4 * b], private_keys + 4 * idx, 32);
                    memcpy(&topics[4 * b], topic, 32);
                    // MerkleTree::prove on the plonky2 digest layout (cap height 0: one tree)
                    uint64_t pair = idx;
                    for (uint32_t i = 0; i < height; i++) {
                        const uint64_t parity = pair & 1;
                        pair >>= 1;
                        const uint64_t slot = (pair << (i + 1)) + (1ull << i) - 1;
                        memcpy(&sib[((size_t)b * height + i) * 4], tree_digests + (2 * slot + (1 - parity)) * 4, 32);
                    }
                    // per-proof blinding keys: derived from the batch key (reproducible batches), or NULL = fresh OS randomness per proof
                    if (key_base) { gl355_derive_key(key_base, 2ull * j, &k_sem[32 * b]); gl355_derive_key(key_base, 2ull * j + 1, &s.k_rec[32 * b]); }
                }
                const int32_t rc = gl355_semaphore_prove_units(ctxs[t], sem, nb, sks.data(), topics.data(), idxs.data(), sib.data(), height,
                                                               key_base ? k_sem.data() : nullptr, s.flat.data(), s.pis.data());
                if (rc != GL355_OK) { fail(rc); s.nb = 0; ok = false; }
                else if (rec) {
                    for (uint32_t b = 0; b < nb; b++) {
                        memcpy(&s.inputs[(size_t)b * (sem_words + 12)], &s.flat[(size_t)b * sem_words], sem_words * 8);
                        memcpy(&s.inputs[(size_t)b * (sem_words + 12) + sem_words], &s.pis[(size_t)b * 12], 96);
                    }
                    const uint32_t threads = cx->replay_threads;
                    Slot* sp = &s;
                    const int device = cx->device;
                    const bool on_device = cx->device_replay;
                    s.replay = std::async(std::launch::async, [sp, rec, threads, device, copy_stream, rec_row_words, on_device]() -> int32_t {
                        // witness generation: the tape interpreter on the side stream (default), or host threads + upload
                        if (on_device)
                            return circuit_replay_units_dev(rec, device, copy_stream, sp->nb, sp->inputs.data(), sp->d_rows, sp->d_aux, sp->rpis.data(),
                                                            &sp->failed_unit, &sp->failed_op, sp->rows /* pinned mirror of d_aux in this mode */);
                        const int32_t rc = circuit_replay_units(rec, threads, sp->nb, sp->inputs.data(), sp->rows, sp->rpis.data(), &sp->failed_unit, &sp->failed_op);
                        if (rc != GL355_OK) return rc;
                        if (hipSetDevice(device) != hipSuccess ||
                            hipMemcpyAsync(sp->d_rows, sp->rows, (size_t)sp->nb * rec_row_words * 8, hipMemcpyHostToDevice, copy_stream) != hipSuccess) return GL355_E_HIP;
                        for (;;) {                        // wait for the upload without holding a core
                            const hipError_t q = hipStreamQuery(copy_stream);
                            if (q == hipSuccess) return GL355_OK;
                            if (q != hipErrorNotReady) return GL355_E_HIP;
                            (void)hipGetLastError();
                            std::this_thread::sleep_for(std::chrono::microseconds(50));
                        }
                    });
                }
            }
            if (prev.nb) { if (!finish(prev)) ok = false; }
            if (j0 >= count) {                       // nothing new was started: drain the current slot and stop
                if (s.nb) { if (rec && !ok) { if (s.replay.valid()) (void)s.replay.get(); s.nb = 0; } else finish(s); }
                break;
            }
            cur ^= 1;
        }
        if (units_per_ctx) units_per_ctx[t] = done;
    };
    // nothing is thrown across the C boundary: a failed allocation / thread creation inside a worker becomes GL355_E_OOM
    auto guarded = [&](uint32_t t) {
        try {
            worker(t);
        } catch (...) {
            int32_t expected = GL355_OK;
            first_error.compare_exchange_strong(expected, GL355_E_OOM);
        }
    };
    std::vector<std::thread> pool;
    try {
        for (uint32_t t = 1; t < n_ctx; t++) pool.emplace_back(guarded, t);
    } catch (...) {
        int32_t expected = GL355_OK;
        first_error.compare_exchange_strong(expected, GL355_E_OOM);
    }
    guarded(0);
    for (auto& th : pool) th.join();
    return first_error.load();
}

// recursion.rs:187-247 `aggregate` as one native call: the binary aggregation tree over n_leaves = 2^n_levels proofs of one circuit.
// levels[l] is the circuit that verifies two proofs of tree level l (level 0 = the leaves, e.g. Semaphore signals) -- one artifact per
// level, loaded once with gl355_circuit_load (the reference rebuilds the circuit inside every aggregate_signals call,
// recursion.rs:25-185).  The nodes of a level are independent (`par_chunks_exact(2)`, recursion.rs:211-227): every context takes the nodes
// t, t + n_ctx, ... and proves them in lock-step batches (gl355_circuit_prove_tape_units); levels are separated by a join.  Node j of level l
// (Scheduling nodes by readiness instead -- a parent starts when its two children are done -- was built and measured: 128 leaves on 8
// contexts 1.07 s against 0.84 s level by level: ready parents trickle in one or two at a time and are proven in lock-step batches of one or
// two, at 28 ms per single proof against ~5 ms per proof in batches of eight.  Level-by-level keeps the batches full.)
// has the blinding key gl355_derive_key(key_base, key_domain << 48 | l << 32 | j) (NULL key_base: fresh OS randomness per proof), so the
// result does not depend on the number of contexts or on the batch size.
extern "C" int32_t gl355_aggregate_units(gl355_ctx* const* ctxs, uint32_t n_ctx, const gl355_circuit_handle* const* levels, uint32_t n_levels,
                                         const uint64_t* leaf_proofs, const uint64_t* leaf_public_inputs, uint32_t n_leaves, uint64_t leaf_words, uint32_t leaf_n_pi,
                                         const uint8_t* key_base, uint64_t key_domain, uint64_t* proof_out, uint64_t proof_capacity_words,
                                         uint64_t* public_inputs_out, uint32_t public_inputs_capacity, double* level_ms) {
    if (!ctxs || n_ctx == 0 || !levels || n_levels == 0 || n_levels > 30 || !leaf_proofs || !leaf_public_inputs || !proof_out || !public_inputs_out) return GL355_E_INVALID_ARG;
    for (uint32_t t = 0; t < n_ctx; t++)
        if (!ctxs[t]) return GL355_E_INVALID_ARG;
    Ctx* c0 = ctx_of(ctxs[0]);
    if (n_leaves != (1u << n_levels)) return c0->fail(GL355_E_INVALID_ARG, "aggregate_units: 2^levels leaves expected");
    // shapes: level l takes two proofs of level l - 1
    std::vector<uint64_t> words(n_levels + 1), n_pi(n_levels + 1);
    words[0] = leaf_words; n_pi[0] = leaf_n_pi;
    for (uint32_t l = 0; l < n_levels; l++) {
        if (!levels[l]) return c0->fail(GL355_E_INVALID_ARG, "aggregate_units: null level circuit");
        uint64_t w = 0, inputs = 0;
        uint32_t pi = 0;
        GL355_TRY(gl355_circuit_info(levels[l], &w, &pi, nullptr, &inputs, nullptr));
        if (inputs != 2 * (words[l] + n_pi[l])) return c0->fail(GL355_E_INVALID_ARG, "aggregate_units: a level circuit does not take two proofs of the level below");
        words[l + 1] = w; n_pi[l + 1] = pi;
    }
    if (words[n_levels] > proof_capacity_words || n_pi[n_levels] > public_inputs_capacity) return c0->fail(GL355_E_INVALID_ARG, "aggregate_units: output buffers too small");
    std::vector<uint64_t> cur_p(leaf_proofs, leaf_proofs + (size_t)n_leaves * leaf_words), cur_pi(leaf_public_inputs, leaf_public_inputs + (size_t)n_leaves * leaf_n_pi);
    std::vector<uint64_t> nxt_p, nxt_pi;
    uint32_t n_nodes = n_leaves;
    for (uint32_t l = 0; l < n_levels; l++) {
        const auto t0 = std::chrono::steady_clock::now();
        n_nodes >>= 1;
        const uint64_t wi = words[l], pii = n_pi[l], wo = words[l + 1], pio = n_pi[l + 1], n_in = 2 * (wi + pii);
        nxt_p.assign((size_t)n_nodes * wo, 0);
        nxt_pi.assign((size_t)n_nodes * pio, 0);
        // Levels of 32 nodes and more keep the device busy: lock-step batches on at most eight contexts (more contexts with smaller batches
        // measured slower: 32 nodes on 16 x 2 units 369 ms against 202-256 ms on 8 x 4; 64 nodes on 16 x 4 447 ms against 328 ms on 8 x 8).
        // Below that a level is latency-bound and every context takes one node (16 nodes on 16 x 1: 108 ms against 132-186 ms on 8 x 2).
        const uint32_t full = std::min<uint32_t>(8, GL355_MAX_UNITS);
        const uint32_t workers = n_nodes >= 32 ? std::min<uint32_t>(std::min<uint32_t>(n_ctx, 8), n_nodes) : std::min(n_ctx, n_nodes);
        const uint32_t units = std::max<uint32_t>(1, std::min<uint32_t>(full, (n_nodes + workers - 1) / workers));
        std::atomic<int32_t> first_error{GL355_OK};
        auto worker = [&](uint32_t t) {
            try {
                std::vector<uint32_t> mine;
                for (uint32_t j = t; j < n_nodes; j += workers) mine.push_back(j);
                std::vector<uint64_t> inputs((size_t)units * n_in), flats((size_t)units * wo), pis((size_t)units * pio);
                std::vector<uint8_t> keys((size_t)units * 32);
                for (size_t b = 0; b < mine.size() && first_error.load() == GL355_OK; b += units) {
                    const uint32_t nb = (uint32_t)std::min<size_t>(units, mine.size() - b);
                    for (uint32_t k = 0; k < nb; k++) {
                        const uint32_t j = mine[b + k];
                        uint64_t* d = &inputs[(size_t)k * n_in];
                        for (uint32_t side = 0; side < 2; side++) {
                            memcpy(d, &cur_p[(size_t)(2 * j + side) * wi], wi * 8); d += wi;
                            memcpy(d, &cur_pi[(size_t)(2 * j + side) * pii], pii * 8); d += pii;
                        }
                        if (key_base) gl355_derive_key(key_base, (key_domain << 48) | ((uint64_t)l << 32) | j, &keys[32 * k]);
                    }
                    const int32_t rc = gl355_circuit_prove_tape_units(ctxs[t], levels[l], nb, inputs.data(), n_in, key_base ? keys.data() : nullptr, flats.data(), pis.data());
                    if (rc != GL355_OK) { int32_t e = GL355_OK; first_error.compare_exchange_strong(e, rc); return; }
                    for (uint32_t k = 0; k < nb; k++) {
                        memcpy(&nxt_p[(size_t)mine[b + k] * wo], &flats[(size_t)k * wo], wo * 8);
                        memcpy(&nxt_pi[(size_t)mine[b + k] * pio], &pis[(size_t)k * pio], pio * 8);
                    }
                }
            } catch (...) {
                int32_t e = GL355_OK;
                first_error.compare_exchange_strong(e, GL355_E_OOM);
            }
        };
        std::vector<std::thread> pool;
        try {
            for (uint32_t t = 1; t < workers; t++) pool.emplace_back(worker, t);
        } catch (...) {
            int32_t e = GL355_OK;
            first_error.compare_exchange_strong(e, GL355_E_OOM);
        }
        worker(0);
        for (auto& th : pool) th.join();
        if (first_error.load() != GL355_OK) return first_error.load();
        cur_p.swap(nxt_p);
        cur_pi.swap(nxt_pi);
        if (level_ms) level_ms[l] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    memcpy(proof_out, cur_p.data(), words[n_levels] * 8);
    memcpy(public_inputs_out, cur_pi.data(), n_pi[n_levels] * 8);
    return GL355_OK;
}
