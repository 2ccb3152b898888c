// Native batch runtime: the reference proves its signals from a rayon `par_iter` (src/plonky2_semaphore/recursion.rs:300-308:
// one `make_signal` per member) and verifies / aggregates them from `par_chunks_exact` (recursion.rs:211-227).  Here one host
// thread per prover context takes the next GL355_OPT_BATCH_UNITS units and proves them in lock-step: Merkle path of the member from the access-set tree, gl355_semaphore_prove,
// and -- when a verifier circuit is given -- gl355_circuit_prove_tape on (proof | public inputs).  No host-language code runs
// between the units; the caller gets the (nullifier | topic) leaf of every unit for the aggregation root.
#include "gl355_internal.h"

#include <atomic>
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
    std::atomic<int32_t> first_error{GL355_OK};
    std::atomic<uint32_t> next_unit{0};     // units are handed out a batch at a time: a context that finishes early takes the next batch
    auto worker = [&](uint32_t t) {
        const uint32_t B = std::max<uint32_t>(1, std::min<uint32_t>(ctx_of(ctxs[t])->batch_units, GL355_MAX_UNITS));
        std::vector<uint64_t> sib((size_t)B * height * 4 + 4), flat((size_t)B * sem_words), pis((size_t)B * 12), inputs(rec ? (size_t)B * (sem_words + 12) : 0),
            outer(rec ? (size_t)B * rec_words : 0), opis((size_t)B * 12), sks((size_t)B * 4), topics((size_t)B * 4), idxs(B);
        std::vector<uint8_t> k_sem((size_t)B * 32), k_rec((size_t)B * 32);
        uint32_t done = 0;
        for (;;) {
            if (first_error.load() != GL355_OK) break;
            const uint32_t j0 = next_unit.fetch_add(B);
            if (j0 >= count) break;
            const uint32_t nb = std::min<uint32_t>(B, count - j0);
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
                if (key_base) { gl355_derive_key(key_base, 2ull * j, &k_sem[32 * b]); gl355_derive_key(key_base, 2ull * j + 1, &k_rec[32 * b]); }
            }
            int32_t rc = gl355_semaphore_prove_units(ctxs[t], sem, nb, sks.data(), topics.data(), idxs.data(), sib.data(), height,
                                                     key_base ? k_sem.data() : nullptr, flat.data(), pis.data());
            const uint64_t* result = flat.data();
            if (rc == GL355_OK && rec) {
                for (uint32_t b = 0; b < nb; b++) {
                    memcpy(&inputs[(size_t)b * (sem_words + 12)], &flat[(size_t)b * sem_words], sem_words * 8);
                    memcpy(&inputs[(size_t)b * (sem_words + 12) + sem_words], &pis[(size_t)b * 12], 96);
                }
                rc = gl355_circuit_prove_tape_units(ctxs[t], rec, nb, inputs.data(), sem_words + 12, key_base ? k_rec.data() : nullptr, outer.data(), opis.data());
                result = outer.data();
            } else if (rc == GL355_OK) {
                memcpy(opis.data(), pis.data(), (size_t)nb * 96);
            }
            if (rc != GL355_OK) {
                int32_t expected = GL355_OK;
                first_error.compare_exchange_strong(expected, rc);
                break;
            }
            for (uint32_t b = 0; b < nb; b++) {
                memcpy(leaves_out + 8ull * (j0 + b), &opis[(size_t)b * 12 + 4], 64);            // nullifier | topic
                if (proofs_out) memcpy(proofs_out + (uint64_t)(j0 + b) * out_words, result + (uint64_t)b * out_words, out_words * 8);
            }
            done += nb;
        }
        if (units_per_ctx) units_per_ctx[t] = done;
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < n_ctx; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    return first_error.load();
}
