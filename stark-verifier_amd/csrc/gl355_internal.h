// Internal C++ declarations shared by the HIP translation units of libgl355.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gl355.h"
#include "gl_field.cuh"
#include "merkle_common.cuh"

namespace gl355 { struct Ctx; }

// a committed polynomial batch resident in HBM (PolynomialBatch); one allocation from the context pool
struct gl355_oracle {
    struct gl355::Ctx* ctx;
    uint32_t log_n, rate_bits, batch, leaf_len, cap_height;
    uint64_t* coeffs;   // [batch][n]
    uint64_t* lde;      // [leaf_len][N], rows in bit-reversed order (salt columns last)
    uint64_t* digests;  // plonky2 layout
    uint64_t* cap;
    uint64_t n_digests;
    int32_t hasher;     // GL355_HASH_*: the hash of this batch's Merkle tree
};

namespace gl355 {

struct Ctx;

// evaluate a HIP call, record the error on the context and return GL355_E_HIP from the caller
#define GL355_HIP(ctx, expr)                                                          \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) return (ctx)->fail_hip(_e, #expr, __FILE__, __LINE__);  \
    } while (0)
#define GL355_TRY(expr)          \
    do {                         \
        int32_t _rc = (expr);    \
        if (_rc != GL355_OK) return _rc; \
    } while (0)

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool external_stream = false;
    std::string err;
    uint64_t* tw_fwd = nullptr;  // omega_{2^14}^e, e < 2^14
    uint64_t* tw_inv = nullptr;  // omega_{2^14}^-e
    // round-major copy for the LDS rounds of radix 2^rho (rho = 3, 4), stored behind the two tables above
    const uint64_t* twr(uint32_t rho, bool inv) const { return tw_fwd + 32768 + ((rho - 3) * 2 + (inv ? 1 : 0)) * 32768; }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // cached two-level power tables, keyed by the list of bases
    struct PowTab { uint64_t* lo; uint64_t* hi; };
    std::map<std::vector<uint64_t>, PowTab> pow_cache;
    std::map<std::vector<uint64_t>, uint64_t*> full_cache;   // full-size multiplier tables (ntt.hip)
    int32_t full_pow_table(const uint64_t* lo, const uint64_t* hi, uint32_t n_cosets, uint32_t log_n, const uint64_t** out);
    int32_t full_step_table(const uint64_t* lo, const uint64_t* hi, uint32_t l1, uint32_t l2, bool inv, const uint64_t** out);
    int32_t nat_step_table(const uint64_t* lo, const uint64_t* hi, uint32_t l1, uint32_t l2, const uint64_t** out);   // natural -> natural flow
    // table of the 24-bit-limb row pass (ntt_l24.hip): the twiddles between its two radix-64 super-rounds
    int32_t l24_mid_table(const uint64_t** out);

    // trivial caching device allocator (per context => per stream, so reuse is stream-ordered)
    struct Block { void* p; size_t size; bool used; uint64_t serial; };
    std::vector<Block> blocks;
    uint64_t block_serial = 0;        // blocks are numbered in creation order: trim_since(mark) only touches what a phase created itself

    // optional per-kernel timing (gl355_profile_enable): HIP events around every launch group
    struct ProfRec { const char* name; hipEvent_t e0, e1; uint64_t bytes; };
    // host wait for the stream: spinning hipStreamSynchronize (lowest latency) or, with GL355_OPT_BLOCKING_SYNC, a blocking
    // event wait that leaves the CPU to other prover threads (more host threads than cores)
    uint32_t replay_threads = 1;      // GL355_OPT_REPLAY_THREADS
    uint32_t batch_units = 8;         // GL355_OPT_BATCH_UNITS
    uint32_t ntt_single_pass_max_log = 12;   // GL355_OPT_NTT_SINGLE_PASS_MAX_LOG (12..14); 12: +1.8 % units/s over 14 (profiles/r03b_single_pass_ab.txt)
    int blocking_sync = 0;            // GL355_OPT_BLOCKING_SYNC: 0 runtime wait, 1 blocking event, 2 poll + back-off
    hipEvent_t sync_ev = nullptr;
    hipError_t wait();
    hipError_t wait_impl();
    // device -> host copy on the stream; into pageable memory the call itself waits for the stream, so it is accounted like wait()
    hipError_t d2h(void* dst, const void* src, size_t bytes);
    uint64_t wait_ns = 0, wait_calls = 0;   // accumulated by wait() while prof_on
    uint64_t wait_cpu_ns = 0, prove_cpu_ns = 0, prove_calls = 0;   // thread CPU time inside wait() / inside prove_units (waits included)
    uint32_t merkle_lanes_log = 14;   // Merkle levels with <= 2^this nodes use the 16-lanes-per-node kernel (GL355_OPT_MERKLE_LANES_LOG)
    bool prof_on = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t prof_event();

    int32_t fail(int32_t code, const char* msg);
    int32_t fail_hip(hipError_t e, const char* expr, const char* file, int line);
    int32_t alloc(size_t bytes, void** out);
    void release(void* p);
    // pinned host staging area of the context (grow-only): device -> host copies into it are truly asynchronous
    void* pinned_buf = nullptr;
    size_t pinned_size = 0;
    int32_t pinned(size_t bytes, void** out);
    // batch runtime (batch.cpp): two witness-row slots (pinned host + device) and a copy stream, kept across calls
    void* rt_rows[2] = {nullptr, nullptr};
    void* rt_drows[2] = {nullptr, nullptr};
    void* rt_aux[2] = {nullptr, nullptr};
    size_t rt_bytes = 0, rt_aux_bytes = 0;
    bool device_replay = true;        // GL355_OPT_DEVICE_REPLAY
    hipStream_t rt_copy_stream = nullptr;
    // second compute stream of the context (the MSM runs its window chunks on two streams so that one chunk's sort -- memory-bound -- and
    // reduction -- latency-bound -- overlap the neighbour's bucket accumulation -- VALU-bound); created on first use
    hipStream_t aux_stream = nullptr;
    std::vector<hipEvent_t> order_ev;       // untimed events for cross-stream ordering, grown on demand
    int32_t aux_stream_get(hipStream_t* out);
    int32_t order_event(size_t i, hipEvent_t* out);
    int32_t runtime_buffers(size_t bytes, size_t aux_bytes, uint64_t* rows[2], uint64_t* drows[2], void* aux[2], hipStream_t* copy_stream);
    void runtime_buffers_free();
    void release_all();
    void trim_since(uint64_t mark);   // hipFree every cached block CREATED at or after `mark` (= block_serial at the start of a one-off phase with large
                                      // temporaries, e.g. a keygen) that is not in use; older blocks -- the warmed cache of proofs this context serves -- stay
    // lo[c*4096 + j] = bases[c]^j, hi[c*4096 + j] = bases[c]^(4096 j)
    int32_t pow_tables_multi(const std::vector<uint64_t>& bases, const uint64_t** lo, const uint64_t** hi);
    int32_t pow_tables(uint64_t base, const uint64_t** lo, const uint64_t** hi) {
        return pow_tables_multi(std::vector<uint64_t>{gl_canon(base)}, lo, hi);
    }
};

// RAII: times everything enqueued on the context stream during its lifetime under `name`
struct ProfScope {
    Ctx* ctx; int idx = -1;
    // alg_bytes: the ALGORITHMIC HBM bytes of the launches in this scope (compulsory reads + writes)
    ProfScope(Ctx* c, const char* name, uint64_t alg_bytes = 0) : ctx(c) {
        if (!c->prof_on) return;
        Ctx::ProfRec r{name, c->prof_event(), c->prof_event(), alg_bytes};
        if (!r.e0 || !r.e1) return;
        (void)hipEventRecord(r.e0, c->stream);
        idx = (int)c->prof.size();
        c->prof.push_back(r);
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(ctx->prof[idx].e1, ctx->stream); }
};

// RAII scratch buffer from the context allocator
struct Scratch {
    Ctx* ctx;
    void* p = nullptr;
    explicit Scratch(Ctx* c) : ctx(c) {}
    ~Scratch() { if (p) ctx->release(p); }
    int32_t get(size_t bytes) { return ctx->alloc(bytes, &p); }
    void reset() { if (p) ctx->release(p); p = nullptr; }      // give the block back early (reuse is stream-ordered)
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
};

// A caller buffer that may live on the host or on the device.  Host buffers are staged through a
// scratch allocation (H2D before, D2H after); device buffers are used in place.
struct Staged {
    Ctx* ctx;
    void* user = nullptr;
    void* dev = nullptr;
    size_t bytes = 0;
    bool is_host = false, copy_back = false;
    explicit Staged(Ctx* c) : ctx(c) {}
    ~Staged() { if (is_host && dev) ctx->release(dev); }
    // dir: bit0 = read by the kernels (copy in), bit1 = written (copy out at finish())
    int32_t open(const void* ptr, size_t nbytes, int dir);
    int32_t finish();  // enqueue D2H if needed, then sync
    template <typename T> T* as() const { return reinterpret_cast<T*>(dev); }
    Staged(const Staged&) = delete;
    Staged& operator=(const Staged&) = delete;
};
bool ptr_is_device(const void* p);

// ---- ntt.hip -------------------------------------------------------------------------------
struct NttPlan {
    const uint64_t* in = nullptr;
    uint64_t* out = nullptr;
    uint64_t in_col_stride = 0, out_col_stride = 0;
    uint32_t log_n = 0, batch = 1;
    bool inverse = false;      // use omega^-1 (no scaling unless `scale` says so)
    bool in_bitrev = false;    // input is in bit-reversed order
    bool out_bitrev = false;   // leave output in bit-reversed order
    uint32_t n_cosets = 1;     // > 1: LDE-style, coset c reads tables c and writes slot coset_slot[c]
    uint64_t coset_ratio = 0;  // != 0: the cosets' bases are base[0] * coset_ratio^c (lets one block run all cosets of a tile)
    uint8_t coset_slot[16] = {0};
    uint64_t coset_out_stride = 0;
    const uint64_t* pre_lo = nullptr;  // multiplier tables on natural-order input index
    const uint64_t* pre_hi = nullptr;
    const uint64_t* post_lo = nullptr; // multiplier tables on natural-order output index
    const uint64_t* post_hi = nullptr;
    uint64_t scale = 1;
};
int32_t ntt_init_constants(Ctx* ctx);
int32_t ntt_run(Ctx* ctx, const NttPlan& p);
int32_t bitrev_permute(Ctx* ctx, const uint64_t* in, uint64_t* out, uint32_t log_n, uint32_t width,
                       uint64_t in_col_stride, uint64_t out_col_stride, uint32_t batch);
int32_t transpose_cols_to_rows(Ctx* ctx, const uint64_t* in, uint64_t* out, uint64_t rows, uint32_t cols,
                               uint64_t in_col_stride, uint32_t out_row_stride, uint32_t log_rows_brev);
// device-resident building blocks used by the C ABI layer and the commit pipeline
int32_t ntt_dev(Ctx* ctx, uint64_t* data, uint32_t log_n, uint32_t batch, uint64_t stride, bool inverse,
                uint64_t coset_shift /* 0 = none */);
int32_t lde_dev(Ctx* ctx, const uint64_t* coeffs, uint64_t in_stride, uint32_t log_n, uint32_t rate_bits,
                uint64_t shift, uint32_t batch, uint64_t* out, uint64_t out_stride, bool out_bitrev);

// ---- merkle.hip ----------------------------------------------------------------------------
int32_t poseidon_permute_dev(Ctx* ctx, uint64_t* states, uint64_t count);
// leaves: row-major [n][leaf_len] (row_stride = leaf_len) when col_major == false, else
// column-major: element (leaf i, column c) at leaves[c * col_stride + i].
int32_t hash_leaves_dev(Ctx* ctx, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, bool col_major,
                        uint64_t col_stride, uint64_t* digests /* n_leaves * 4, linear order */);
int32_t merkle_build_dev(Ctx* ctx, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, bool col_major,
                         uint64_t col_stride, uint32_t cap_height, uint64_t* digests, uint64_t* cap);
int32_t merkle_build_args(Ctx* ctx, LeafArgs a, uint32_t sub_bits, uint64_t* digests, uint64_t* cap);
int32_t bn254_merkle_build_args(Ctx* ctx, LeafArgs a, uint32_t sub_bits, uint64_t* digests, uint64_t* cap);
int32_t merkle_build_args_any(Ctx* ctx, int32_t hasher, const LeafArgs& a, uint32_t sub_bits, uint64_t* digests, uint64_t* cap);
int32_t hash_no_pad_dev(Ctx* ctx, const uint64_t* inputs, uint64_t n, uint32_t len, uint64_t* digests);
int32_t two_to_one_dev(Ctx* ctx, const uint64_t* l, const uint64_t* r, uint64_t n, uint64_t* out);
int32_t open_batch_dev(Ctx* ctx, const uint64_t* lde, uint64_t stride, uint32_t leaf_len, const uint64_t* digests,
                       uint32_t log_n, uint32_t cap_height, const uint64_t* idx_dev, uint32_t n_idx, uint64_t* leaves_out,
                       uint64_t* sib_out);
int32_t open_batch_ex_dev(Ctx* ctx, const uint64_t* leaves, uint64_t stride, uint32_t leaf_len, const uint64_t* digests,
                          uint32_t log_n, uint32_t cap_height, const uint64_t* idx_dev, uint32_t idx_shift, uint32_t n_idx,
                          uint64_t* leaves_out, uint64_t leaf_out_stride, uint64_t* sib_out, uint64_t sib_out_stride);
// second hash back-end (merkle_bn254.hip, host_bn254.cpp)
void host_bn254_permute(uint64_t state[12]);
// Merkle tree / PoW with the hasher chosen at run time
int32_t merkle_build_any(Ctx* ctx, int32_t hasher, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, bool col_major,
                         uint64_t col_stride, uint32_t cap_height, uint64_t* digests, uint64_t* cap);
int32_t pow_grind_any(Ctx* ctx, int32_t hasher, const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t start,
                      uint64_t* witness_host);
int32_t bn254_pow_grind_dev(Ctx* ctx, const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t start, uint64_t* witness_host);
int32_t bn254_permute_dev(Ctx* ctx, uint64_t* states, uint64_t count);
int32_t bn254_hash_leaves_dev(Ctx* ctx, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, bool col_major, uint64_t col_stride,
                              uint64_t* digests, bool always_hash);
int32_t bn254_two_to_one_dev(Ctx* ctx, const uint64_t* l, const uint64_t* r, uint64_t n, uint64_t* out);
int32_t bn254_merkle_build_dev(Ctx* ctx, const uint64_t* leaves, uint64_t n_leaves, uint32_t leaf_len, bool col_major, uint64_t col_stride,
                               uint32_t cap_height, uint64_t* digests, uint64_t* cap);
int32_t oracle_open_batch_on(Ctx* ctx, const gl355_oracle* o, const uint64_t* indices, uint32_t n_idx, uint64_t* leaves,
                             uint64_t* siblings);
int32_t pow_grind_dev(Ctx* ctx, const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t start,
                      uint64_t* witness_host);

// ---- fri.hip -------------------------------------------------------------------------------
int32_t deep_batch_dev(Ctx* ctx, const uint64_t* const* poly_ptrs_host, uint32_t n_polys, uint32_t log_n,
                       const uint64_t alpha[2], const uint64_t z[2], uint64_t* acc /* n ext */);
int32_t eval_polys_ext_dev(Ctx* ctx, const uint64_t* const* poly_ptrs_host, uint32_t n_polys, uint32_t log_n,
                           const uint64_t z[2], uint64_t* out_host_or_dev);
int32_t fri_fold_dev(Ctx* ctx, const uint64_t* coeffs, uint64_t n, const uint64_t beta[2], uint64_t* out);
int32_t fri_layer_leaves_dev(Ctx* ctx, const uint64_t* values, uint64_t n, uint64_t* leaves);
int32_t lde_ext_dev(Ctx* ctx, const uint64_t* coeffs, uint32_t log_n, uint32_t rate_bits, uint64_t shift, uint64_t* out,
                    bool out_bitrev);
int32_t field_batch_dev(Ctx* ctx, int32_t op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n);
int32_t zs_partial_products_dev(Ctx* ctx, const uint64_t* wires, const uint64_t* sigmas, const uint64_t* k_is,
                                uint32_t log_n, uint32_t n_routed, uint32_t max_degree, uint64_t beta, uint64_t gamma,
                                uint64_t* z_out, uint64_t* pp_out);
int32_t canon_dev(Ctx* ctx, uint64_t* a, uint64_t n);
int32_t intt_from_bitrev_dev(Ctx* ctx, const uint64_t* in, uint64_t in_stride, uint64_t* out, uint64_t out_stride,
                             uint32_t log_n, uint32_t batch, uint64_t coset_shift);
int32_t quotient_dev(Ctx* ctx, const gl355_circuit* c, const uint64_t* cs_lde, const uint64_t* wires_lde,
                     const uint64_t* zs_lde, uint64_t lde_stride, const uint64_t* k_is_dev, const uint64_t* betas,
                     const uint64_t* gammas, const uint64_t* alphas, const uint64_t pi_hash[4], uint64_t* out_values);
Ctx* ctx_of(gl355_ctx* h);
uint64_t thread_cpu_ns();

// ---- prover_batch.hip: CircuitData::prove for B lock-step units of one circuit ---------------------------------------------
struct BlindKey;
struct ProveUnit {
    const uint64_t* public_inputs; uint32_t n_public_inputs;
    uint32_t key[8];                       // the unit's blinding key (blinding.cuh BlindKey words)
    uint64_t* proof; uint64_t proof_capacity_words;
};
// dense witness: d_wires_dense = [B][num_wires][n] on the device; sparse: rows_host = [B][n_rows][num_wires] (host) of circuit rows
// row_idx[r], blinding rows generated on the device from each unit's key
int32_t prove_units(Ctx* ctx, const gl355_prover_data* pd, uint32_t B, const uint64_t* d_wires_dense, const uint32_t* row_idx,
                    const uint64_t* rows_host, uint32_t n_rows, uint32_t blind_start, uint32_t n_blind, uint32_t z_start, uint32_t n_z_pairs,
                    const ProveUnit* io);
uint64_t tape_validate(const uint64_t* tape, uint64_t n_ops, uint64_t n_inputs, uint64_t n_words, uint32_t num_wires);
int32_t tape_replay_dev(hipStream_t stream, const uint64_t* d_tape, const uint64_t* d_seg_start, uint64_t n_seq, uint32_t n_segs, uint32_t n_units,
                        const uint64_t* d_inputs, uint64_t n_inputs, uint64_t* d_rows, uint64_t n_words, const uint64_t* d_pi_pos, uint32_t n_pi,
                        uint64_t* d_status, uint64_t* d_pis);
// witness rows of n_units units generated on the device on `stream` (inputs: host [units][n_inputs]); d_aux: device scratch of
// circuit_replay_aux_bytes(); the call returns after the stream has finished (polling, no spinning)
uint64_t circuit_replay_aux_bytes(const gl355_circuit_handle* ch, uint32_t n_units);
int32_t circuit_replay_units_dev(const gl355_circuit_handle* ch, int device, hipStream_t stream, uint32_t n_units, const uint64_t* inputs, uint64_t* d_rows,
                                 void* d_aux, uint64_t* pis_out, uint64_t* failed_unit, uint64_t* failed_op, void* h_aux = nullptr);
uint64_t circuit_rows_words(const gl355_circuit_handle* ch);
int32_t circuit_replay_units(const gl355_circuit_handle* ch, uint32_t threads, uint32_t n_units, const uint64_t* inputs, uint64_t* rows, uint64_t* pis_out,
                             uint64_t* failed_unit, uint64_t* failed_op);
int32_t resolve_blinding_key_words(Ctx* ctx, const uint8_t* key, uint32_t out[8]);
int32_t quotient_units_dev(Ctx* ctx, const gl355_circuit* c, uint32_t B, const uint64_t* cs_lde, const uint64_t* wires_lde, uint64_t wires_us,
                           const uint64_t* zs_lde, uint64_t zs_us, uint64_t lde_stride, const uint64_t* k_is_dev, const uint64_t* betas /* [B][4] */,
                           const uint64_t* gammas, const uint64_t* alphas, const uint64_t* pi_hashes /* [B][4] */, uint64_t* out_values /* [B][nch][nq] */);

static inline uint32_t log2_u64(uint64_t n) { uint32_t l = 0; while ((1ull << l) < n) l++; return l; }
static inline uint32_t host_brev(uint32_t x, uint32_t bits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

}  // namespace gl355
