// Native batch runtime: the reference proves its signals from a rayon `par_iter` (src/plonky2_semaphore/recursion.rs:300-308:
// one `make_signal` per member) and verifies / aggregates them from `par_chunks_exact` (recursion.rs:211-227).  Here one host
// thread per prover context takes every n_ctx-th unit: Merkle path of the member from the access-set tree, gl355_semaphore_prove,
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
    std::atomic<uint32_t> next_unit{0};     // units are handed out one at a time: a context that finishes early takes the next one
    auto worker = [&](uint32_t t) {
        std::vector<uint64_t> sib((size_t)height * 4 + 4), flat(sem_words + 12), outer(rec ? rec_words : 0);
        uint32_t done = 0;
        for (uint32_t j; (j = next_unit.fetch_add(1)) < count && first_error.load() == GL355_OK;) {
            const uint64_t idx = member_indices[j];
            // MerkleTree::prove on the plonky2 digest layout (cap height 0: one tree)
            uint64_t pair = idx;
            for (uint32_t i = 0; i < height; i++) {
                const uint64_t parity = pair & 1;
                pair >>= 1;
                const uint64_t slot = (pair << (i + 1)) + (1ull << i) - 1;
                memcpy(&sib[4 * i], tree_digests + (2 * slot + (1 - parity)) * 4, 32);
            }
            uint64_t* pis = flat.data() + sem_words;
            // per-proof blinding keys: derived from the batch key (reproducible batches), or NULL = fresh OS randomness per proof
            uint8_t k_sem[32], k_rec[32];
            if (key_base) { gl355_derive_key(key_base, 2ull * j, k_sem); gl355_derive_key(key_base, 2ull * j + 1, k_rec); }
            int32_t rc = gl355_semaphore_prove(ctxs[t], sem, private_keys + 4 * idx, topic, idx, sib.data(), height, key_base ? k_sem : nullptr,
                                               flat.data(), sem_words, pis);
            uint64_t opis[12];
            const uint64_t* result = flat.data();
            if (rc == GL355_OK && rec) {
                rc = gl355_circuit_prove_tape(ctxs[t], rec, flat.data(), sem_words + 12, key_base ? k_rec : nullptr, outer.data(), rec_words, opis);
                result = outer.data();
            } else if (rc == GL355_OK) {
                memcpy(opis, pis, sizeof opis);
            }
            if (rc != GL355_OK) {
                int32_t expected = GL355_OK;
                first_error.compare_exchange_strong(expected, rc);
                break;
            }
            memcpy(leaves_out + 8ull * j, opis + 4, 64);            // nullifier | topic
            if (proofs_out) memcpy(proofs_out + (uint64_t)j * out_words, result, out_words * 8);
            done++;
        }
        if (units_per_ctx) units_per_ctx[t] = done;
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < n_ctx; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    return first_error.load();
}
