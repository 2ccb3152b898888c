// SURVEY 8(f) N4, the data-parallel stages of the reference's SNARK finalisation on the GPU: halo2_proofs'
//     create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<_>, _, _, Keccak256Transcript, _>
// as src/plonky2_verifier/chip/native_chip/test_utils.rs:57-95 calls it (verifier_api.rs:77-92; README.md:171-177: 505-511 s at k = 23 on 16
// vCPUs), driven by a plain circuit descriptor (columns, queries, gates and lookup expressions as register programs, the permutation's
// columns -- stark-verifier_amd/halo2.py writes it; the reference's own chips: halo2_chips.py).  The Halo2 verifier CIRCUIT that the reference
// proves with it (the layout of a plonky2 verification over those chips) and witness synthesis stay out of scope: the
// advice columns are an input.
//
// Stages, names as in halo2 (plonk/prover.rs and the argument provers), all on resident Montgomery-form columns:
//   advice       blind the unusable rows, commit every column in Lagrange form (batched MSMs over the resident SRS), Lagrange -> coefficients
//   lookups      compress the input / table expressions with theta (expression evaluator), permute_expression_pair (radix sort of the 256-bit
//                values, rocPRIM, + flag / scan / gather kernels), commit
//   permutation  per set of `degree - 2` columns: row products, batched inversion, running product, blinding, commit
//   lookup z     the same for (A + beta)(S + gamma) / ((A' + beta)(S' + gamma))
//   vanishing    random polynomial, commit
//   evaluate_h   per coset of the extended domain: coset FFT of every polynomial, gate program + permutation + lookup constraints folded with
//                y, division by X^n - 1; one inverse FFT over the extended domain; split into `degree - 1` pieces, commit
//   evaluations  every queried polynomial at x omega^rot (block Horner + reduction)
//   SHPLONK      rotation sets, sum_j y^j (p_ij - r_ij) / Z_i combined with v, commit; linearisation at u, division by (X - u), commit
// The transcript (Keccak-256, host_keccak.cpp) and a few dozen field operations per stage run on the host.  Byte-for-byte the same proof as
// oracle/halo2_model.py on the same (witness, seed) -- tests/test_gpu_halo2.py -- and accepted by the restated verifier.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <chrono>
#include <memory>

#include "host_fr.h"
#include "plonk_kernels.cuh"

using namespace gl355;

namespace gl355 { void bn254_g1_add_host(const uint64_t a[8], const uint64_t b[8], uint64_t out[8]); }      // host_bn254_curve.cpp

namespace {

constexpr uint64_t PLK_MAGIC = 0x4B4C503535334C47ull;       // "GL355PLK"
constexpr uint32_t PLK_HDR = 24;

struct Lookup { std::vector<uint32_t> in_code, tab_code; };

inline uint32_t blocks(uint64_t n, uint32_t per = 256) { return (uint32_t)((n + per - 1) / per); }
inline u256 to_dev(const Fr& a) { u256 r; for (int i = 0; i < 4; i++) { r.l[2 * i] = (uint32_t)a.l[i]; r.l[2 * i + 1] = (uint32_t)(a.l[i] >> 32); } return r; }

}  // namespace

struct gl355_plonk_pk {
    Ctx* ctx = nullptr;
    gl355_ctx* handle = nullptr;
    uint32_t k = 0, n_advice = 0, n_fixed = 0, n_instance = 0, n_perm = 0, n_lookups = 0, degree = 0, bf = 0, n_gate_polys = 0;
    uint32_t ext_k = 0, n_pieces = 0, n_sets = 0, chunk_len = 0;
    uint64_t n = 0, usable = 0;
    Fr digest;
    std::vector<std::pair<uint32_t, uint32_t>> perm_cols;                 // (kind, index)
    std::vector<std::pair<int32_t, int32_t>> queries[3];                  // (column, rotation)
    std::vector<Fr> consts;
    std::vector<uint32_t> gate_code;
    std::vector<Lookup> lookups;
    std::vector<void*> owned;                                             // device allocations of the key
    // device, Montgomery: values and coefficient forms
    uint64_t *fixed_vals = nullptr, *fixed_polys = nullptr, *sigma_vals = nullptr, *sigma_polys = nullptr, *l_polys = nullptr /* l0 | l_last | l_active */;
    uint64_t *omega_pows = nullptr, *delta_pows = nullptr, *d_consts = nullptr;
    uint64_t *tw_fwd = nullptr, *tw_inv = nullptr;
    // the circuit's own polynomials on every coset of the extended domain, computed once at keygen (halo2's ProvingKey keeps fixed_cosets,
    // the permutation's cosets and l0 / l_last / l_active the same way): [coset][fixed | sigma | l0 l_last l_active][n].  At the reference's
    // k = 23 that is 8 x 28 x 256 MB = 57 GB of the 288 GB -- and 36 % of evaluate_h's transforms gone.  nullptr: recomputed per proof
    uint64_t* fixed_cos = nullptr;
    uint32_t n_fix_cos = 0;
    uint64_t* export_quotient = nullptr;                                  // host buffer the next proof's quotient pieces are copied to (inspection hook)
    const uint64_t *g = nullptr, *g_lagrange = nullptr;                   // the caller's resident SRS (device) or owned copies
    // the SRS's window multiples (gl355_bn254_g1_msm_prepare): every commitment of a proof is an MSM over one of the two base sets, and with the
    // tables its windows share one bucket set.  2 x 6.4 GB at k = 23; null below k = 22, when the device is short of memory or with GL355_PLONK_MSM_TABLES=0
    gl355_msm_bases *tab_g = nullptr, *tab_gl = nullptr;
    uint32_t* d_gate_code = nullptr;
    std::vector<uint32_t*> d_lk_code;                                     // per lookup: input program, table program
    int32_t* d_q[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    std::vector<uint64_t> fixed_commitments, sigma_commitments;           // host, affine
    int32_t dalloc(size_t bytes, void** out) { GL355_TRY(ctx->alloc(bytes, out)); owned.push_back(*out); return GL355_OK; }
};

namespace {

// ---- small helpers over the context ------------------------------------------------------------------------------------------------
// GL355_PLONK_MEM_TRACE=1: at the end of every stage, the scratch allocator's live and cached bytes and the device's free memory (stderr)
static void mem_trace(Ctx* ctx, const char* what) {
    static const bool on = getenv("GL355_PLONK_MEM_TRACE") != nullptr;
    if (!on) return;
    size_t live = 0, cached = 0, fr = 0, tot = 0;
    for (auto& b : ctx->blocks) (b.used ? live : cached) += b.size;
    (void)hipMemGetInfo(&fr, &tot);
    fprintf(stderr, "[plonk mem] %-18s scratch live %7.2f GB  cached %7.2f GB  device used %7.2f GB\n", what, live / 1e9, cached / 1e9, (tot - fr) / 1e9);
}
struct Timer {
    Ctx* ctx; double* slot; std::chrono::steady_clock::time_point t0; const char* name;
    Timer(Ctx* c, double* s, const char* nm = "stage") : ctx(c), slot(s), t0(std::chrono::steady_clock::now()), name(nm) {}
    ~Timer() {
        if (slot) { (void)ctx->wait(); *slot += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
        mem_trace(ctx, name);
    }
};

int32_t upload(Ctx* ctx, void* dst, const void* src, size_t bytes) {
    GL355_HIP(ctx, hipMemcpyAsync(dst, src, bytes, ptr_is_device(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    if (!ptr_is_device(src)) GL355_HIP(ctx, ctx->wait());          // pageable source: the buffer may go away
    return GL355_OK;
}
int32_t fr_to_device(Ctx* ctx, const std::vector<Fr>& v, uint64_t* d) {
    if (v.empty()) return GL355_OK;
    GL355_HIP(ctx, hipMemcpyAsync(d, v.data(), v.size() * 32, hipMemcpyHostToDevice, ctx->stream));
    GL355_HIP(ctx, ctx->wait());
    return GL355_OK;
}
int32_t ptrs_to_device(Ctx* ctx, const std::vector<const uint64_t*>& v, const uint64_t** d) {
    if (v.empty()) return GL355_OK;
    GL355_HIP(ctx, hipMemcpyAsync((void*)d, v.data(), v.size() * sizeof(void*), hipMemcpyHostToDevice, ctx->stream));
    GL355_HIP(ctx, ctx->wait());
    return GL355_OK;
}
void words_of(const Fr& a, uint64_t w[4]) { a.to_words(w); }

// values (n) -> coefficients: inverse transform with the 1 / n
int32_t lagrange_to_coeff(gl355_plonk_pk* pk, const uint64_t* vals, uint64_t* coeffs, uint64_t* work) {
    uint64_t ninv[4];
    Fr::from_u64(pk->n).inv().to_words(ninv);
    return bn254_fr_ntt_mont(pk->ctx, vals, pk->n, coeffs, pk->n, pk->k, pk->tw_inv, nullptr, nullptr, ninv, work);
}

// commitments of `sets` columns (Montgomery scalars, [sets][n]) over n bases: plain copies of the scalars, their bit lengths (one OR
// reduction per column), then batched MSMs over runs of neighbouring columns of the same length class -- windows above a column's bit
// length are never built (range-check columns hold 16-bit values, the arithmetic chip's operands 64-bit ones, selectors 0 / 1).  Rows >= tail
// (the blinding rows: full-size random scalars in every column) are committed by a second MSM over those few bases and added on the host.
int32_t commit_columns(gl355_plonk_pk* pk, const uint64_t* bases, const uint64_t* cols_mont, uint32_t sets, uint64_t* out_host /* sets x 8 */, uint64_t tail = ~0ull) {
    Ctx* ctx = pk->ctx;
    const uint64_t n = pk->n;
    if (!sets) return GL355_OK;
    tail = std::min(tail, n);
    const uint64_t nt = n - tail;
    const gl355_msm_bases* tab = bases == pk->g ? pk->tab_g : (bases == pk->g_lagrange ? pk->tab_gl : nullptr);
    // <= 2^27 scalars and <= 64 sets per batched MSM (bn254_curve.hip)
    const uint32_t per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(16, (1ull << 27) / n));
    Scratch plain(ctx);
    GL355_TRY(plain.get((size_t)sets * n * 32 + (size_t)sets * nt * 32 + (size_t)sets * 32));
    uint64_t* d_plain = plain.as<uint64_t>();
    uint64_t* d_tail = d_plain + 4ull * sets * n;
    unsigned long long* d_or = reinterpret_cast<unsigned long long*>(d_tail + 4ull * sets * nt);
    GL355_HIP(ctx, hipMemsetAsync(d_or, 0, (size_t)sets * 32, ctx->stream));
    hipLaunchKernelGGL(plk_from_mont_body_kernel, dim3(blocks((uint64_t)sets * n)), dim3(256), 0, ctx->stream, cols_mont, d_plain, n, tail, (uint64_t)sets * n);
    if (nt) hipLaunchKernelGGL(plk_from_mont_tail_kernel, dim3(blocks((uint64_t)sets * nt)), dim3(256), 0, ctx->stream, cols_mont, d_tail, n, tail, sets);
    hipLaunchKernelGGL(plk_column_or_kernel, dim3((uint32_t)std::min<uint64_t>(256, blocks(n)), sets), dim3(256), 0, ctx->stream, (const uint64_t*)d_plain, n, d_or);
    GL355_HIP(ctx, hipGetLastError());
    std::vector<unsigned long long> ors(4ull * sets);
    GL355_HIP(ctx, ctx->d2h(ors.data(), d_or, (size_t)sets * 32));
    GL355_HIP(ctx, ctx->wait());
    std::vector<uint32_t> cls(sets), nbits(sets);
    for (uint32_t s = 0; s < sets; s++) {
        uint32_t bits = 0;
        for (int l = 3; l >= 0 && !bits; l--) if (ors[4 * s + l]) bits = 64 * l + 64 - (uint32_t)__builtin_clzll(ors[4 * s + l]);
        nbits[s] = std::max(1u, bits);
        cls[s] = (nbits[s] + 19) / 20;                           // classes of 20 bits (the MSM's windows are 17 .. 20 bits wide)
    }
    for (uint32_t s0 = 0; s0 < sets;) {
        // ... and <= 72 windows per call: the sort's scratch is ~230 MB per window at k = 23 (nine full-size columns in one call held 29 GB of the
        // context's cache for the rest of the proof; five + four cost the same time)
        const uint32_t max_m = std::max(1u, 72u / (cls[s0] + 1));
        uint32_t m = 1;
        while (s0 + m < sets && m < per && m < max_m && cls[s0 + m] == cls[s0]) m++;
        // (columns of at most 20 bits are one window either way: the per-window form's buckets are cheaper to size-sort and reduce than the tables' 2^21 --
        // and it gets the run's exact bit length: a 16-bit column is ONE window of 17 bits, 2^16 buckets, and no carry window)
        uint32_t run_bits = 0;
        for (uint32_t j = 0; j < m; j++) run_bits = std::max(run_bits, nbits[s0 + j]);
        GL355_TRY(bn254_msm_bits(pk->handle, bases, d_plain + 4ull * s0 * n, n, m, cls[s0] >= 2 ? std::min(256u, 20 * cls[s0]) : run_bits, out_host + 8ull * s0,
                                 cls[s0] >= 2 ? tab : nullptr));
        s0 += m;
    }
    if (nt) {
        std::vector<uint64_t> tails(8ull * sets);
        for (uint32_t s0 = 0; s0 < sets; s0 += 64) {
            const uint32_t m = std::min(64u, sets - s0);
            GL355_TRY(bn254_msm_bits(pk->handle, bases + 8 * tail, d_tail + 4ull * s0 * nt, nt, m, 256, tails.data() + 8ull * s0));
        }
        for (uint32_t s = 0; s < sets; s++) bn254_g1_add_host(out_host + 8ull * s, tails.data() + 8ull * s, out_host + 8ull * s);
    }
    return GL355_OK;
}

int32_t run_program(gl355_plonk_pk* pk, const uint32_t* d_code, uint32_t n_instr, const uint64_t* const* const d_cols[3], const Fr& fold, const uint64_t* acc_in,
                    uint64_t* acc_out, bool bitrev) {
    PlkEvalArgs a;
    memset(&a, 0, sizeof a);
    a.code = d_code; a.n_instr = n_instr; a.consts = pk->d_consts; a.n = pk->n; a.fold = to_dev(fold); a.acc_in = acc_in; a.acc_out = acc_out;
    a.log_n = pk->k; a.bitrev = bitrev ? 1 : 0;
    for (int kd = 0; kd < 3; kd++) { a.cols[kd] = d_cols[kd]; a.q_col[kd] = pk->d_q[kd][0]; a.q_rot[kd] = pk->d_q[kd][1]; }
    hipLaunchKernelGGL(plk_eval_kernel, dim3(blocks(pk->n)), dim3(256), 0, pk->ctx->stream, a);
    GL355_HIP(pk->ctx, hipGetLastError());
    return GL355_OK;
}

// a lookup expression list that is one column at the current rotation (the reference's nine range checks, arithmetic_chip.rs:140-151):
// the "compressed" column is the column itself, no program run, no copy.  -> (kind, column) or kind = 3
std::pair<uint32_t, uint32_t> single_query(const gl355_plonk_pk* pk, const std::vector<uint32_t>& code) {
    if (code.size() == 4 && code[0] == PLK_OP_EMIT) {
        const uint32_t kind = code[2] >> 24, idx = code[2] & 0xFFFFFFu;
        if (kind >= PLK_K_ADVICE && kind <= PLK_K_INSTANCE) {
            const auto& q = pk->queries[kind - PLK_K_ADVICE][idx];
            if (q.second == 0) return {kind - PLK_K_ADVICE, (uint32_t)q.first};
        }
    }
    return {3u, 0u};
}

// z[0] = *start, z[i] = z[i - 1] r[i - 1]
int32_t running_product(Ctx* ctx, const uint64_t* r, uint64_t n, const uint64_t* d_start, uint64_t* z) {
    if (n <= PLK_SCAN_CHUNK) {
        hipLaunchKernelGGL(plk_scan_walk_kernel, dim3(1), dim3(64), 0, ctx->stream, r, n, (const uint64_t*)nullptr, d_start, z);
        GL355_HIP(ctx, hipGetLastError());
        return GL355_OK;
    }
    const uint64_t chunks = (n + PLK_SCAN_CHUNK - 1) / PLK_SCAN_CHUNK;
    Scratch sc(ctx);
    GL355_TRY(sc.get(chunks * 64));
    uint64_t* P = sc.as<uint64_t>();
    uint64_t* S = P + 4 * chunks;
    hipLaunchKernelGGL(plk_scan_chunk_kernel, dim3(blocks(chunks, 64)), dim3(64), 0, ctx->stream, r, n, P);
    GL355_HIP(ctx, hipGetLastError());
    GL355_TRY(running_product(ctx, P, chunks, d_start, S));
    hipLaunchKernelGGL(plk_scan_walk_kernel, dim3(blocks(chunks, 64)), dim3(64), 0, ctx->stream, r, n, (const uint64_t*)S, d_start, z);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

// ascending order of `count` plain 256-bit integers: out = v sorted.  LSD over the four 64-bit limbs, a stable radix sort of (limb, index) per
// limb; limbs that are zero everywhere (range-check columns hold 16-bit values) are skipped
int32_t sort256(Ctx* ctx, const uint64_t* v, uint64_t count, uint64_t* out) {
    Scratch sc(ctx);
    size_t tmp_bytes = 0;
    GL355_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, count, 0, 64, ctx->stream));
    const size_t keys_b = (count * 8 + 255) & ~size_t(255), idx_b = (count * 4 + 255) & ~size_t(255);
    GL355_TRY(sc.get(2 * keys_b + 2 * idx_b + tmp_bytes + 256));
    uint8_t* p = sc.as<uint8_t>();
    uint64_t* keys_in = (uint64_t*)p; p += keys_b;
    uint64_t* keys_out = (uint64_t*)p; p += keys_b;
    uint32_t* idx_a = (uint32_t*)p; p += idx_b;
    uint32_t* idx_b_ = (uint32_t*)p; p += idx_b;
    unsigned long long* d_or = (unsigned long long*)p; p += 256;
    void* tmp = p;
    GL355_HIP(ctx, hipMemsetAsync(d_or, 0, 32, ctx->stream));
    hipLaunchKernelGGL(plk_limb_or_kernel, dim3(blocks(count)), dim3(256), 0, ctx->stream, v, count, d_or);
    hipLaunchKernelGGL(plk_iota_kernel, dim3(blocks(count)), dim3(256), 0, ctx->stream, idx_a, count);
    GL355_HIP(ctx, hipGetLastError());
    unsigned long long ors[4];
    GL355_HIP(ctx, ctx->d2h(ors, d_or, 32));
    GL355_HIP(ctx, ctx->wait());
    for (uint32_t limb = 0; limb < 4; limb++) {
        if (limb && !ors[limb]) continue;
        hipLaunchKernelGGL(plk_gather_limb_kernel, dim3(blocks(count)), dim3(256), 0, ctx->stream, v, (const uint32_t*)idx_a, limb, keys_in, count);
        GL355_HIP(ctx, hipGetLastError());
        GL355_HIP(ctx, rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, idx_a, idx_b_, count, 0, 64, ctx->stream));
        std::swap(idx_a, idx_b_);
    }
    hipLaunchKernelGGL(plk_gather_rows_kernel, dim3(blocks(count)), dim3(256), 0, ctx->stream, v, (const uint32_t*)idx_a, out, count);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

// lookup::prover::permute_expression_pair on the first `u` rows of A and S (Montgomery); writes A' and S' (Montgomery) rows [0, u)
int32_t permute_pair(Ctx* ctx, const uint64_t* A, const uint64_t* S, uint64_t u, uint64_t* Ap, uint64_t* Sp) {
    Scratch sc(ctx);
    size_t scan_bytes = 0;
    GL355_HIP(ctx, rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, u, rocprim::plus<uint32_t>(), ctx->stream));
    const size_t vb = u * 32, fb = (u * 4 + 255) & ~size_t(255);
    GL355_TRY(sc.get(5 * vb + 4 * fb + scan_bytes + 512));
    uint8_t* p = sc.as<uint8_t>();
    uint64_t* plain = (uint64_t*)p; p += vb;
    uint64_t* a_sorted = (uint64_t*)p; p += vb;
    uint64_t* t_sorted = (uint64_t*)p; p += vb;
    uint64_t* list = (uint64_t*)p; p += vb;
    uint64_t* table = (uint64_t*)p; p += vb;
    uint32_t* rep = (uint32_t*)p; p += fb;
    uint32_t* left = (uint32_t*)p; p += fb;
    uint32_t* rep_pos = (uint32_t*)p; p += fb;
    uint32_t* left_pos = (uint32_t*)p; p += fb;
    void* scan_tmp = p;
    hipLaunchKernelGGL(plk_from_mont_kernel, dim3(blocks(u)), dim3(256), 0, ctx->stream, A, plain, u);
    GL355_TRY(sort256(ctx, plain, u, a_sorted));
    hipLaunchKernelGGL(plk_from_mont_kernel, dim3(blocks(u)), dim3(256), 0, ctx->stream, S, plain, u);
    GL355_TRY(sort256(ctx, plain, u, t_sorted));
    hipLaunchKernelGGL(plk_lookup_flags_kernel, dim3(blocks(u)), dim3(256), 0, ctx->stream, (const uint64_t*)a_sorted, (const uint64_t*)t_sorted, u, rep, left);
    GL355_HIP(ctx, hipGetLastError());
    GL355_HIP(ctx, rocprim::exclusive_scan(scan_tmp, scan_bytes, rep, rep_pos, 0u, u, rocprim::plus<uint32_t>(), ctx->stream));
    GL355_HIP(ctx, rocprim::exclusive_scan(scan_tmp, scan_bytes, left, left_pos, 0u, u, rocprim::plus<uint32_t>(), ctx->stream));
    uint32_t last[4];
    GL355_HIP(ctx, ctx->d2h(&last[0], rep + (u - 1), 4));
    GL355_HIP(ctx, ctx->d2h(&last[1], rep_pos + (u - 1), 4));
    GL355_HIP(ctx, ctx->d2h(&last[2], left + (u - 1), 4));
    GL355_HIP(ctx, ctx->d2h(&last[3], left_pos + (u - 1), 4));
    GL355_HIP(ctx, ctx->wait());
    const uint32_t n_rep = last[0] + last[1], n_left = last[2] + last[3];
    if (n_rep != n_left) return ctx->fail(GL355_E_INVALID_ARG, "plonk_prove: a lookup input value does not occur in its table");
    hipLaunchKernelGGL(plk_lookup_compact_kernel, dim3(blocks(u)), dim3(256), 0, ctx->stream, (const uint64_t*)t_sorted, (const uint32_t*)left, (const uint32_t*)left_pos, u, list);
    hipLaunchKernelGGL(plk_lookup_table_kernel, dim3(blocks(u)), dim3(256), 0, ctx->stream, (const uint64_t*)a_sorted, (const uint32_t*)rep, (const uint32_t*)rep_pos,
                       (const uint64_t*)list, n_rep, u, table);
    hipLaunchKernelGGL(plk_to_mont_kernel, dim3(blocks(u)), dim3(256), 0, ctx->stream, (const uint64_t*)a_sorted, Ap, u);
    hipLaunchKernelGGL(plk_to_mont_kernel, dim3(blocks(u)), dim3(256), 0, ctx->stream, (const uint64_t*)table, Sp, u);
    GL355_HIP(ctx, hipGetLastError());
    GL355_HIP(ctx, ctx->wait());          // the scratch block goes back to the pool when this returns
    return GL355_OK;
}

int32_t random_rows(Ctx* ctx, const BlindKey& key, uint32_t stream, uint32_t a, uint64_t first, uint64_t count, uint64_t* out) {
    if (!count) return GL355_OK;
    hipLaunchKernelGGL(plk_random_kernel, dim3(blocks(count)), dim3(256), 0, ctx->stream, key, stream, a, first, count, out);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

// values of `polys[q]` at `points[q]`: -> host Fr (canonical Montgomery)
int32_t eval_polys(gl355_plonk_pk* pk, const std::vector<const uint64_t*>& polys, const std::vector<Fr>& points, std::vector<Fr>& out) {
    Ctx* ctx = pk->ctx;
    const uint32_t nq = (uint32_t)polys.size();
    out.resize(nq);
    if (!nq) return GL355_OK;
    const uint32_t bpq = blocks(pk->n, 256 * PLK_EVAL_PER);
    Scratch sc(ctx);
    GL355_TRY(sc.get((size_t)nq * (8 + 32 + 32 + 32) + (size_t)nq * bpq * 32 + 256));
    uint8_t* p = sc.as<uint8_t>();
    const uint64_t** d_polys = (const uint64_t**)p; p += ((size_t)nq * 8 + 31) & ~size_t(31);
    uint64_t* d_pts = (uint64_t*)p; p += (size_t)nq * 32;
    uint64_t* d_pts256 = (uint64_t*)p; p += (size_t)nq * 32;
    uint64_t* d_out = (uint64_t*)p; p += (size_t)nq * 32;
    uint64_t* d_partial = (uint64_t*)p;
    std::vector<Fr> p256(nq);
    for (uint32_t q = 0; q < nq; q++) p256[q] = points[q].pow_u64(256);
    GL355_TRY(ptrs_to_device(ctx, polys, d_polys));
    GL355_TRY(fr_to_device(ctx, points, d_pts));
    GL355_TRY(fr_to_device(ctx, p256, d_pts256));
    PlkEvalPolyArgs a;
    a.polys = d_polys; a.points = d_pts; a.points256 = d_pts256; a.n = pk->n; a.blocks_per_q = bpq; a.partial = d_partial;
    hipLaunchKernelGGL(plk_eval_poly_kernel, dim3(bpq, nq), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(plk_eval_sum_kernel, dim3(blocks(nq, 64)), dim3(64), 0, ctx->stream, (const uint64_t*)d_partial, nq, bpq, d_out);
    GL355_HIP(ctx, hipGetLastError());
    GL355_HIP(ctx, ctx->d2h(out.data(), d_out, (size_t)nq * 32));
    GL355_HIP(ctx, ctx->wait());
    return GL355_OK;
}

// out = (acc_in) + sum_j coeffs[j] polys[j] - low (the first low.size() coefficients)
int32_t lincomb(gl355_plonk_pk* pk, const std::vector<const uint64_t*>& polys, const std::vector<Fr>& coeffs, const std::vector<Fr>& low, const uint64_t* acc_in,
                uint64_t* out) {
    Ctx* ctx = pk->ctx;
    const uint32_t cnt = (uint32_t)polys.size();
    Scratch sc(ctx);
    GL355_TRY(sc.get((size_t)cnt * 40 + 256));
    const uint64_t** d_polys = sc.as<const uint64_t*>();
    uint64_t* d_co = (uint64_t*)(sc.as<uint8_t>() + (((size_t)cnt * 8 + 31) & ~size_t(31)));
    GL355_TRY(ptrs_to_device(ctx, polys, d_polys));
    GL355_TRY(fr_to_device(ctx, coeffs, d_co));
    PlkLincombArgs a;
    memset(&a, 0, sizeof a);
    a.polys = d_polys; a.coeffs = d_co; a.count = cnt; a.n = pk->n; a.acc_in = acc_in; a.out = out;
    if (low.size() > 4) return ctx->fail(GL355_E_UNSUPPORTED, "plonk_prove: a polynomial is opened at more than four points");
    a.n_low = (uint32_t)low.size();
    for (size_t i = 0; i < low.size(); i++) a.low[i] = to_dev(low[i]);
    hipLaunchKernelGGL(plk_lincomb_kernel, dim3(blocks(pk->n)), dim3(256), 0, ctx->stream, a);
    GL355_HIP(ctx, hipGetLastError());
    GL355_HIP(ctx, ctx->wait());          // the pointer / coefficient tables are scratch
    return GL355_OK;
}

Fr plonk_zeta() {
    const uint64_t zw[4] = {0xb8ca0b2d36636f23ull, 0xcc37a73fec2bc5e9ull, 0x048b6e193fd84104ull, 0x30644e72e131a029ull};      // Fr::ZETA
    return Fr::from_words(zw);
}
// the key's polynomials (fixed | sigma | l0 l_last l_active) on cosets [c0, c1) of the extended domain: out[(c - c0)][poly][n]
int32_t plonk_fixed_cosets(gl355_plonk_pk* pk, uint32_t c0, uint32_t c1, uint64_t* out, uint64_t* pre, uint64_t* work) {
    const uint64_t n = pk->n;
    const Fr ext_omega = Fr::root_of_unity(pk->ext_k);

    Fr base = plonk_zeta() * ext_omega.pow_u64(c0);
    for (uint32_t c = c0; c < c1; c++) {
        uint64_t bw[4];
        base.to_words(bw);
        uint64_t* dst = out + 4ull * (uint64_t)(c - c0) * pk->n_fix_cos * n;
        for (uint32_t i = 0; i < pk->n_fix_cos; i++) {
            const uint64_t* src = i < pk->n_fixed ? pk->fixed_polys + 4ull * i * n
                                  : (i < pk->n_fixed + pk->n_perm ? pk->sigma_polys + 4ull * (i - pk->n_fixed) * n : pk->l_polys + 4ull * (i - pk->n_fixed - pk->n_perm) * n);
            GL355_TRY(bn254_fr_ntt_mont_coset_dif(pk->ctx, src, n, dst + 4ull * i * n, pk->k, pk->tw_fwd, bw, pre, i == 0));     // `pre`: the coset's block constants
        }
        base = base * ext_omega;
    }
    return GL355_OK;
}

// coefficients (ascending) of the polynomial of degree < m through (pts[i], evs[i])
std::vector<Fr> interpolate(const std::vector<Fr>& pts, const std::vector<Fr>& evs) {
    const size_t m = pts.size();
    std::vector<Fr> out(m, Fr::zero());
    for (size_t i = 0; i < m; i++) {
        std::vector<Fr> num(1, Fr::one());
        Fr den = Fr::one();
        for (size_t j = 0; j < m; j++) {
            if (j == i) continue;
            std::vector<Fr> nx(num.size() + 1, Fr::zero());
            for (size_t t = 0; t < num.size(); t++) { nx[t + 1] = nx[t + 1] + num[t]; nx[t] = nx[t] - pts[j] * num[t]; }
            num.swap(nx);
            den = den * (pts[i] - pts[j]);
        }
        const Fr c = evs[i] * den.inv();
        for (size_t t = 0; t < num.size(); t++) out[t] = out[t] + c * num[t];
    }
    return out;
}
Fr horner(const std::vector<Fr>& c, const Fr& x) {
    Fr acc = Fr::zero();
    for (size_t i = c.size(); i-- > 0;) acc = acc * x + c[i];
    return acc;
}

}  // namespace

extern "C" {

int32_t gl355_kzg_commit_columns(gl355_ctx* h, const uint64_t* g, const uint64_t* columns, uint32_t log_n, uint32_t n_cols, uint64_t* results) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!g || !columns || !results) return ctx->fail(GL355_E_INVALID_ARG, "kzg_commit_columns: null argument");
    if (log_n > 26) return ctx->fail(GL355_E_UNSUPPORTED, "kzg_commit_columns: log_n > 26");
    const uint64_t n = 1ull << log_n;
    const bool dev_res = ptr_is_device(results);
    std::vector<uint64_t> host(8ull * n_cols);
    Staged sg(ctx), sc(ctx);
    GL355_TRY(sg.open(g, n * 64, 1));                       // host SRS: one upload for all the columns
    const uint32_t per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(8, (1ull << 27) / n));
    for (uint32_t c0 = 0; c0 < n_cols; c0 += per) {
        const uint32_t m = std::min(per, n_cols - c0);
        GL355_TRY(gl355_bn254_g1_msm_batch(h, sg.as<uint64_t>(), columns + 4ull * c0 * n, n, m, host.data() + 8ull * c0));
    }
    if (dev_res) { GL355_HIP(ctx, hipMemcpyAsync(results, host.data(), host.size() * 8, hipMemcpyHostToDevice, ctx->stream)); GL355_HIP(ctx, ctx->wait()); }
    else memcpy(results, host.data(), host.size() * 8);
    return GL355_OK;
}

int32_t gl355_plonk_pk_destroy(gl355_plonk_pk* pk) {
    if (!pk) return GL355_OK;
    if (pk->ctx && hipSetDevice(pk->ctx->device) == hipSuccess) {
        (void)pk->ctx->wait();
        for (void* p : pk->owned) pk->ctx->release(p);
        (void)gl355_bn254_g1_msm_bases_free(pk->handle, pk->tab_g);
        (void)gl355_bn254_g1_msm_bases_free(pk->handle, pk->tab_gl);
    }
    delete pk;
    return GL355_OK;
}

int32_t gl355_plonk_keygen(gl355_ctx* h, const uint64_t* desc, uint64_t words, const uint64_t* g, const uint64_t* g_lagrange, const uint64_t* fixed_values,
                           const uint32_t* mapping, gl355_plonk_pk** out) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!desc || !g || !g_lagrange || !out || words < PLK_HDR) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: null or truncated argument");
    *out = nullptr;
    if (desc[0] != PLK_MAGIC || desc[1] != 1) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: not a version-1 gl355 PLONK descriptor");
    // keygen's temporaries (staged fixed values and mapping, the MSM scratch of 25 commitments, transform scratch: 66 GB at k = 23) are of no use to a
    // proof: back to the device, not into the context's cache -- on every return path (declared first, so it runs after the key's own destructor on
    // a failure), and only the blocks this call created: a context that already serves proofs keeps its warmed cache
    struct TrimGuard { Ctx* c; uint64_t mark; ~TrimGuard() { c->trim_since(mark); } } trim_guard{ctx, ctx->block_serial};
    std::unique_ptr<gl355_plonk_pk, int32_t (*)(gl355_plonk_pk*)> pk(new (std::nothrow) gl355_plonk_pk(), gl355_plonk_pk_destroy);
    if (!pk) return GL355_E_OOM;
    pk->ctx = ctx; pk->handle = h;
    pk->k = (uint32_t)desc[2]; pk->n_advice = (uint32_t)desc[3]; pk->n_fixed = (uint32_t)desc[4]; pk->n_instance = (uint32_t)desc[5];
    pk->n_perm = (uint32_t)desc[6]; pk->n_lookups = (uint32_t)desc[7]; pk->degree = (uint32_t)desc[8]; pk->bf = (uint32_t)desc[9];
    const uint64_t nq[3] = {desc[10], desc[11], desc[12]}, n_consts = desc[13], gate_len = desc[14];
    pk->n_gate_polys = (uint32_t)desc[15];
    if (pk->k < 3 || pk->k > 24 || pk->n_advice > 256 || pk->n_fixed > 256 || pk->n_instance > 16 || pk->n_perm > 256 || pk->n_lookups > 64 || pk->degree < 3 ||
        pk->degree > 10 || pk->bf < 3 || pk->bf > 64 || nq[0] > 1024 || nq[1] > 1024 || nq[2] > 64 || n_consts > 4096 || gate_len > (1u << 20))
        return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: implausible circuit shape");
    pk->n = 1ull << pk->k;
    if (pk->n < pk->bf + 3ull) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: fewer rows than the blinding needs");
    pk->usable = pk->n - (pk->bf + 1);
    pk->n_pieces = pk->degree - 1;
    pk->ext_k = pk->k;
    while ((1ull << pk->ext_k) < pk->n * pk->n_pieces) pk->ext_k++;
    pk->chunk_len = pk->degree - 2;
    pk->n_sets = pk->n_perm ? (pk->n_perm + pk->chunk_len - 1) / pk->chunk_len : 0;
    pk->digest = Fr::from_words(desc + 16);
    if ((pk->n_fixed && !fixed_values) || (pk->n_perm && !mapping)) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: fixed values / permutation mapping missing");
    // ---- the rest of the descriptor
    const uint64_t* p = desc + PLK_HDR;
    const uint64_t* end = desc + words;
    auto need = [&](uint64_t w) { return (uint64_t)(end - p) >= w; };
    if (!need(pk->n_perm)) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: truncated descriptor");
    const uint32_t kind_cols[3] = {pk->n_advice, pk->n_fixed, pk->n_instance};
    for (uint32_t j = 0; j < pk->n_perm; j++, p++) {
        const uint32_t kind = (uint32_t)(*p >> 32), idx = (uint32_t)*p;
        if (kind > 2 || idx >= kind_cols[kind]) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: bad permutation column");
        pk->perm_cols.push_back({kind, idx});
    }
    for (int kd = 0; kd < 3; kd++) {
        if (!need(nq[kd])) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: truncated descriptor");
        for (uint64_t q = 0; q < nq[kd]; q++, p++) {
            const int32_t col = (int32_t)(*p >> 32), rot = (int32_t)(uint32_t)*p;
            if (col < 0 || (uint32_t)col >= kind_cols[kd] || rot < -(int32_t)pk->bf - 1 || rot > (int32_t)pk->bf + 1) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: bad query");
            pk->queries[kd].push_back({col, rot});
        }
    }
    if (!need(4 * n_consts)) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: truncated descriptor");
    for (uint64_t c = 0; c < n_consts; c++, p += 4) pk->consts.push_back(Fr::from_words(p));
    auto read_code = [&](uint64_t len, std::vector<uint32_t>& code) -> bool {
        if (!need(2 * len)) return false;
        code.assign(reinterpret_cast<const uint32_t*>(p), reinterpret_cast<const uint32_t*>(p) + 4 * len);
        p += 2 * len;
        for (uint64_t i = 0; i < len; i++) {                  // every operand in range: the evaluator trusts its program
            const uint32_t op = code[4 * i], dst = code[4 * i + 1];
            if (op > PLK_OP_MOV || dst >= PLK_MAX_REGS) return false;
            for (int o = 0; o < (op == PLK_OP_ADD || op == PLK_OP_SUB || op == PLK_OP_MUL ? 2 : 1); o++) {
                const uint32_t v = code[4 * i + 2 + o], kind = v >> 24, idx = v & 0xFFFFFFu;
                if (kind == PLK_K_REG ? idx >= PLK_MAX_REGS : (kind == PLK_K_CONST ? idx >= n_consts : (kind > PLK_K_INSTANCE || idx >= nq[kind - PLK_K_ADVICE]))) return false;
            }
        }
        return true;
    };
    if (!read_code(gate_len, pk->gate_code)) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: bad gate program");
    for (uint32_t l = 0; l < pk->n_lookups; l++) {
        if (!need(2)) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: truncated descriptor");
        const uint64_t li = p[0], lt = p[1];
        p += 2;
        Lookup lk;
        if (li > (1u << 16) || lt > (1u << 16) || !read_code(li, lk.in_code) || !read_code(lt, lk.tab_code)) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: bad lookup program");
        pk->lookups.push_back(std::move(lk));
    }
    if (p != end) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: descriptor length does not match its header");
    // every permutation column must be queried at rotation 0 (halo2's enable_equality does that): the verifier reads it at x
    for (auto& pc : pk->perm_cols) {
        bool ok = false;
        for (auto& q : pk->queries[pc.first]) ok = ok || (q.first == (int32_t)pc.second && q.second == 0);
        if (!ok) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: a permutation column is not queried at the current rotation");
    }
    // SHPLONK's linear combination carries at most four low-order correction terms per polynomial (PlkLincombArgs::low): a column queried at
    // five or more distinct rotations is refused HERE, not at the end of every proof.  The arguments' own polynomials are opened
    // at <= 3 points (z: x, wx, w^last x; permuted input: x, w^-1 x).
    for (int kd = 0; kd < 2; kd++) {
        std::vector<std::vector<int32_t>> rots(kind_cols[kd]);
        for (auto& q : pk->queries[kd]) {
            auto& r = rots[q.first];
            if (std::find(r.begin(), r.end(), q.second) == r.end()) r.push_back(q.second);
            if (r.size() > 4) return ctx->fail(GL355_E_UNSUPPORTED, "plonk_keygen: a column is queried at more than four distinct rotations");
        }
    }
    // ---- device side
    const uint64_t n = pk->n;
    gl355_plonk_pk* k_ = pk.get();
    auto D = [&](size_t bytes, auto** ptr) { return k_->dalloc(bytes, reinterpret_cast<void**>(ptr)); };
    if (ptr_is_device(g)) pk->g = g; else { uint64_t* d; GL355_TRY(D(n * 64, &d)); GL355_TRY(upload(ctx, d, g, n * 64)); pk->g = d; }
    if (ptr_is_device(g_lagrange)) pk->g_lagrange = g_lagrange; else { uint64_t* d; GL355_TRY(D(n * 64, &d)); GL355_TRY(upload(ctx, d, g_lagrange, n * 64)); pk->g_lagrange = d; }
    {
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        // worth their memory from k = 22 on: below that a column's shared bucket set (2^18 buckets at k = 20) is too few lanes for the bucket kernel --
        // k = 20 proof 0.176 s without, 0.184 s with; k = 23: 1.010 / 0.962 s.  GL355_PLONK_MSM_TABLES=0 / 1: never / from k = 12 (A/B, small-memory runs, tests)
        const char* e = getenv("GL355_PLONK_MSM_TABLES");
        const bool want = e ? (atoi(e) != 0 && pk->k >= 12) : pk->k >= 22;
        if (want && (size_t)n * 64 * 13 * 2 < free_b / 4) {
            GL355_TRY(gl355_bn254_g1_msm_prepare(h, pk->g, n, &pk->tab_g));
            GL355_TRY(gl355_bn254_g1_msm_prepare(h, pk->g_lagrange, n, &pk->tab_gl));
        }
    }
    GL355_TRY(D((n / 2 + 1) * 32, &pk->tw_fwd));
    GL355_TRY(D((n / 2 + 1) * 32, &pk->tw_inv));
    GL355_TRY(bn254_fr_twiddles(ctx, pk->k, false, pk->tw_fwd));
    GL355_TRY(bn254_fr_twiddles(ctx, pk->k, true, pk->tw_inv));
    const uint64_t one_w[4] = {1, 0, 0, 0};
    uint64_t w_w[4], d_w[4];
    Fr::root_of_unity(pk->k).to_words(w_w);
    GL355_TRY(D(n * 32, &pk->omega_pows));
    GL355_TRY(bn254_fr_power_table(ctx, w_w, one_w, n, pk->omega_pows));
    const Fr delta = Fr::from_u64(7).pow_u64(1ull << 28);            // Fr::DELTA = GENERATOR^(2^S)
    delta.to_words(d_w);
    GL355_TRY(D(std::max<uint32_t>(1, pk->n_perm) * 32, &pk->delta_pows));
    if (pk->n_perm) GL355_TRY(bn254_fr_power_table(ctx, d_w, one_w, pk->n_perm, pk->delta_pows));
    GL355_TRY(D(std::max<size_t>(1, pk->consts.size()) * 32, &pk->d_consts));
    GL355_TRY(fr_to_device(ctx, pk->consts, pk->d_consts));
    auto code_to_dev = [&](const std::vector<uint32_t>& code, uint32_t** d) -> int32_t {
        GL355_TRY(D(std::max<size_t>(16, code.size() * 4), d));
        if (!code.empty()) { GL355_HIP(ctx, hipMemcpyAsync(*d, code.data(), code.size() * 4, hipMemcpyHostToDevice, ctx->stream)); GL355_HIP(ctx, ctx->wait()); }
        return GL355_OK;
    };
    GL355_TRY(code_to_dev(pk->gate_code, &pk->d_gate_code));
    for (auto& lk : pk->lookups) {
        uint32_t *a = nullptr, *b = nullptr;
        GL355_TRY(code_to_dev(lk.in_code, &a));
        GL355_TRY(code_to_dev(lk.tab_code, &b));
        pk->d_lk_code.push_back(a);
        pk->d_lk_code.push_back(b);
    }
    for (int kd = 0; kd < 3; kd++) {
        std::vector<int32_t> cols, rots;
        for (auto& q : pk->queries[kd]) { cols.push_back(q.first); rots.push_back(q.second); }
        for (int w = 0; w < 2; w++) {
            GL355_TRY(D(std::max<size_t>(16, cols.size() * 4), &pk->d_q[kd][w]));
            if (!cols.empty()) { GL355_HIP(ctx, hipMemcpyAsync(pk->d_q[kd][w], (w ? rots : cols).data(), cols.size() * 4, hipMemcpyHostToDevice, ctx->stream)); GL355_HIP(ctx, ctx->wait()); }
        }
    }
    Scratch work(ctx);
    GL355_TRY(work.get(n * 32));
    // fixed columns: values, coefficients, commitments (Lagrange form)
    GL355_TRY(D(std::max<uint32_t>(1, pk->n_fixed) * n * 32, &pk->fixed_vals));
    GL355_TRY(D(std::max<uint32_t>(1, pk->n_fixed) * n * 32, &pk->fixed_polys));
    if (pk->n_fixed) {
        Staged sf(ctx);
        GL355_TRY(sf.open(fixed_values, (size_t)pk->n_fixed * n * 32, 1));
        hipLaunchKernelGGL(plk_to_mont_kernel, dim3(blocks((uint64_t)pk->n_fixed * n)), dim3(256), 0, ctx->stream, sf.as<uint64_t>(), pk->fixed_vals, (uint64_t)pk->n_fixed * n);
        GL355_HIP(ctx, hipGetLastError());
        GL355_HIP(ctx, ctx->wait());
    }
    pk->fixed_commitments.assign(8ull * pk->n_fixed, 0);
    GL355_TRY(commit_columns(k_, pk->g_lagrange, pk->fixed_vals, pk->n_fixed, pk->fixed_commitments.data()));
    for (uint32_t c = 0; c < pk->n_fixed; c++) GL355_TRY(lagrange_to_coeff(k_, pk->fixed_vals + 4ull * c * n, pk->fixed_polys + 4ull * c * n, work.as<uint64_t>()));
    // permutation: sigma values from the mapping, coefficients, commitments
    GL355_TRY(D(std::max<uint32_t>(1, pk->n_perm) * n * 32, &pk->sigma_vals));
    GL355_TRY(D(std::max<uint32_t>(1, pk->n_perm) * n * 32, &pk->sigma_polys));
    if (pk->n_perm) {
        Staged sm(ctx);
        GL355_TRY(sm.open(mapping, (size_t)pk->n_perm * n * 8, 1));
        // the range of every entry is checked by the kernel itself (host and device mappings alike): an entry outside [0, n_perm) x [0, n)
        // raises the flag and reads nothing
        Scratch flag(ctx);
        GL355_TRY(flag.get(32));
        GL355_HIP(ctx, hipMemsetAsync(flag.as<uint32_t>(), 0, 4, ctx->stream));
        hipLaunchKernelGGL(plk_sigma_kernel, dim3(blocks((uint64_t)pk->n_perm * n)), dim3(256), 0, ctx->stream, sm.as<uint32_t>(), n, pk->n_perm, (const uint64_t*)pk->delta_pows,
                           (const uint64_t*)pk->omega_pows, pk->sigma_vals, flag.as<uint32_t>());
        GL355_HIP(ctx, hipGetLastError());
        uint32_t bad = 0;
        GL355_HIP(ctx, ctx->d2h(&bad, flag.as<uint32_t>(), 4));
        GL355_HIP(ctx, ctx->wait());
        if (bad) return ctx->fail(GL355_E_INVALID_ARG, "plonk_keygen: permutation mapping out of range");
    }
    pk->sigma_commitments.assign(8ull * pk->n_perm, 0);
    GL355_TRY(commit_columns(k_, pk->g_lagrange, pk->sigma_vals, pk->n_perm, pk->sigma_commitments.data()));
    for (uint32_t c = 0; c < pk->n_perm; c++) GL355_TRY(lagrange_to_coeff(k_, pk->sigma_vals + 4ull * c * n, pk->sigma_polys + 4ull * c * n, work.as<uint64_t>()));
    // the transcript's initial scalar (halo2: vk.transcript_repr, a hash of the PINNED verifying key -- shape, fixed commitments, permutation
    // commitments): a descriptor whose digest field is zero gets Keccak-256 over (the whole descriptor | fixed commitments | sigma
    // commitments), as a big-endian integer mod r, so two circuits that differ only in fixed values or copy constraints never share a
    // Fiat-Shamir prefix and a C caller cannot forget the step.  A non-zero field is taken as given (a host that computed
    // halo2's own transcript_repr); gl355_plonk_pk_set_digest still overrides.
    if (!(desc[16] | desc[17] | desc[18] | desc[19])) {
        std::vector<uint8_t> pre((size_t)words * 8 + (pk->fixed_commitments.size() + pk->sigma_commitments.size()) * 8);
        memcpy(pre.data(), desc, (size_t)words * 8);
        if (!pk->fixed_commitments.empty()) memcpy(pre.data() + (size_t)words * 8, pk->fixed_commitments.data(), pk->fixed_commitments.size() * 8);
        if (!pk->sigma_commitments.empty()) memcpy(pre.data() + (size_t)words * 8 + pk->fixed_commitments.size() * 8, pk->sigma_commitments.data(), pk->sigma_commitments.size() * 8);
        uint8_t hsh[32];
        keccak256_host(pre.data(), pre.size(), hsh);
        uint64_t w[4];
        for (int i = 0; i < 4; i++) { w[i] = 0; for (int b = 0; b < 8; b++) w[i] |= (uint64_t)hsh[31 - (8 * i + b)] << (8 * b); }
        pk->digest = Fr::from_words(w);
    }
    // l_0, l_last, l_active_row
    GL355_TRY(D(3 * n * 32, &pk->l_polys));
    hipLaunchKernelGGL(plk_indicator_kernel, dim3(blocks(n)), dim3(256), 0, ctx->stream, pk->l_polys, pk->l_polys + 4 * n, pk->l_polys + 8 * n, n, pk->usable);
    GL355_HIP(ctx, hipGetLastError());
    for (int c = 0; c < 3; c++) GL355_TRY(lagrange_to_coeff(k_, pk->l_polys + 4ull * c * n, pk->l_polys + 4ull * c * n, work.as<uint64_t>()));
    GL355_HIP(ctx, ctx->wait());
    pk->n_fix_cos = pk->n_fixed + pk->n_perm + 3;
    {
        const uint32_t n_cosets = pk->n_pieces;              // evaluate_h visits as many cosets as the quotient has pieces (see gl355_plonk_prove)
        const size_t bytes = (size_t)n_cosets * pk->n_fix_cos * n * 32;
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        static const bool off = getenv("GL355_PLONK_NO_FIXED_COSETS") != nullptr;            // A/B and small-memory runs
        if (!off && bytes < free_b / 3) {
            GL355_TRY(D(bytes, &pk->fixed_cos));
            Scratch pre(ctx);
            GL355_TRY(pre.get(n * 32));
            GL355_TRY(plonk_fixed_cosets(k_, 0, n_cosets, pk->fixed_cos, pre.as<uint64_t>(), work.as<uint64_t>()));
            GL355_HIP(ctx, ctx->wait());
        }
    }
    work.reset();
    *out = pk.release();                 // (trim_guard frees the temporaries this call created as the function returns)
    return GL355_OK;
}

int32_t gl355_plonk_pk_info(const gl355_plonk_pk* pk, uint64_t info[8]) {
    if (!pk || !info) return GL355_E_INVALID_ARG;
    const uint64_t n_adv_q = pk->queries[0].size(), n_fix_q = pk->queries[1].size();
    uint64_t n_perm_evals = pk->n_sets ? 3ull * pk->n_sets - 1 : 0;
    const uint64_t points = pk->n_advice + 3ull * pk->n_lookups + pk->n_sets + 1 + pk->n_pieces + 2;
    const uint64_t scalars = n_adv_q + n_fix_q + 1 + pk->n_perm + n_perm_evals + 5ull * pk->n_lookups;
    info[0] = pk->k; info[1] = pk->ext_k; info[2] = pk->n_sets; info[3] = pk->n_pieces; info[4] = pk->usable;
    info[5] = 64 * points + 32 * scalars;          // proof bytes
    info[6] = pk->n_fixed; info[7] = pk->n_perm;
    return GL355_OK;
}

int32_t gl355_plonk_pk_commitments(const gl355_plonk_pk* pk, uint64_t* fixed_c, uint64_t* sigma_c) {
    if (!pk) return GL355_E_INVALID_ARG;
    if (fixed_c && !pk->fixed_commitments.empty()) memcpy(fixed_c, pk->fixed_commitments.data(), pk->fixed_commitments.size() * 8);
    if (sigma_c && !pk->sigma_commitments.empty()) memcpy(sigma_c, pk->sigma_commitments.data(), pk->sigma_commitments.size() * 8);
    return GL355_OK;
}

int32_t gl355_plonk_pk_digest(const gl355_plonk_pk* pk, uint64_t digest[4]) {
    if (!pk || !digest) return GL355_E_INVALID_ARG;
    pk->digest.to_words(digest);
    return GL355_OK;
}

int32_t gl355_plonk_pk_export_quotient(gl355_plonk_pk* pk, uint64_t* host_out) {
    if (!pk) return GL355_E_INVALID_ARG;
    pk->export_quotient = host_out;
    return GL355_OK;
}

int32_t gl355_plonk_pk_set_digest(gl355_plonk_pk* pk, const uint64_t digest[4]) {
    if (!pk || !digest) return GL355_E_INVALID_ARG;
    pk->digest = Fr::from_words(digest);
    return GL355_OK;
}

int32_t gl355_plonk_prove(gl355_ctx* h, gl355_plonk_pk* pk, const uint64_t* advice, const uint64_t* instances, const uint32_t* instance_lens, const uint8_t seed[32],
                          uint8_t* proof, uint64_t capacity, uint64_t* proof_len, uint64_t* trace, double* stage_ms) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!pk || pk->ctx != ctx || !seed || !proof || !proof_len || (pk->n_advice && !advice) || (pk->n_instance && !instance_lens))
        return ctx->fail(GL355_E_INVALID_ARG, "plonk_prove: null argument or a key of another context");
    // the inspection hook (gl355_plonk_pk_export_quotient) arms exactly ONE call: taken and cleared here, so a proof that fails before the quotient
    // stage disarms it too and no later proof writes through a pointer its caller may have freed
    uint64_t* const export_quotient = pk->export_quotient;
    pk->export_quotient = nullptr;
    const uint64_t n = pk->n, u = pk->usable;
    const uint32_t n_cosets = pk->n_pieces;
    const int32_t last_rot = -(int32_t)(pk->bf + 1);
    const BlindKey key = blind_key_from_bytes(seed);
    double ms[GL355_PLONK_STAGES];
    for (int i = 0; i < GL355_PLONK_STAGES; i++) ms[i] = 0;
    const bool timed = stage_ms != nullptr;
    auto slot = [&](int i) { return timed ? &ms[i] : nullptr; };
    KeccakTranscript tr;
    std::vector<void*> mine;                                  // this proof's device buffers
    struct Freer { Ctx* c; std::vector<void*>* v; ~Freer() { (void)c->wait(); for (void* p : *v) c->release(p); } } freer{ctx, &mine};
    auto D = [&](size_t bytes, uint64_t** ptr) -> int32_t { void* p = nullptr; GL355_TRY(ctx->alloc(std::max<size_t>(bytes, 32), &p)); mine.push_back(p); *ptr = (uint64_t*)p; return GL355_OK; };
    // a proof buffer that no later stage reads goes back to the context's allocator at once (everything runs on the context's stream, so a later
    // owner of the block is ordered behind its last reader); what is still held at the end is released by `freer`
    auto Free = [&](uint64_t*& ptr) {
        if (!ptr) return;
        for (auto it = mine.begin(); it != mine.end(); ++it) if (*it == (void*)ptr) { mine.erase(it); break; }
        ctx->release(ptr);
        ptr = nullptr;
    };
    uint64_t* work = nullptr;
    GL355_TRY(D(n * 32, &work));                              // FFT scratch of the natural-order transforms
    const Fr omega = Fr::root_of_unity(pk->k), omega_inv = omega.inv();
    auto rotate = [&](const Fr& x, int32_t r) { return x * (r >= 0 ? omega : omega_inv).pow_u64((uint64_t)(r >= 0 ? r : -r)); };

    // ---- vk, instances ------------------------------------------------------------------------------------------------------------
    tr.common_scalar(pk->digest);
    uint64_t *inst_vals = nullptr, *inst_polys = nullptr;
    GL355_TRY(D(std::max<uint32_t>(1, pk->n_instance) * n * 32, &inst_vals));
    GL355_TRY(D(std::max<uint32_t>(1, pk->n_instance) * n * 32, &inst_polys));
    {
        GL355_HIP(ctx, hipMemsetAsync(inst_vals, 0, std::max<uint32_t>(1, pk->n_instance) * n * 32, ctx->stream));
        uint64_t off = 0;
        for (uint32_t c = 0; c < pk->n_instance; c++) {
            const uint32_t len = instance_lens[c];
            if (len > u) return ctx->fail(GL355_E_INVALID_ARG, "plonk_prove: more instance values than usable rows");
            if (len && !instances) return ctx->fail(GL355_E_INVALID_ARG, "plonk_prove: instance values missing");
            if (len) {
                std::vector<uint64_t> hv(4ull * len);
                GL355_HIP(ctx, hipMemcpy(hv.data(), instances + 4 * off, 32ull * len, ptr_is_device(instances) ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
                std::vector<Fr> fv(len);
                for (uint32_t i = 0; i < len; i++) { fv[i] = Fr::from_words(hv.data() + 4 * i); tr.common_scalar(fv[i]); }
                GL355_TRY(fr_to_device(ctx, fv, inst_vals + 4ull * c * n));
            }
            off += len;
        }
        for (uint32_t c = 0; c < pk->n_instance; c++) GL355_TRY(lagrange_to_coeff(pk, inst_vals + 4ull * c * n, inst_polys + 4ull * c * n, work));
    }

    // ---- advice ---------------------------------------------------------------------------------------------------------------------
    uint64_t *adv_vals = nullptr, *adv_polys = nullptr;
    GL355_TRY(D(std::max<uint32_t>(1, pk->n_advice) * n * 32, &adv_vals));
    GL355_TRY(D(std::max<uint32_t>(1, pk->n_advice) * n * 32, &adv_polys));
    {
        Timer t(ctx, slot(GL355_PLONK_STAGE_ADVICE), "advice");
        if (pk->n_advice) {
            Staged sa(ctx);
            GL355_TRY(sa.open(advice, (size_t)pk->n_advice * n * 32, 1));
            hipLaunchKernelGGL(plk_to_mont_kernel, dim3(blocks((uint64_t)pk->n_advice * n)), dim3(256), 0, ctx->stream, sa.as<uint64_t>(), adv_vals, (uint64_t)pk->n_advice * n);
            GL355_HIP(ctx, hipGetLastError());
            GL355_HIP(ctx, ctx->wait());
        }
        for (uint32_t c = 0; c < pk->n_advice; c++) GL355_TRY(random_rows(ctx, key, PLK_STREAM_ADVICE, c, u, n - u, adv_vals + 4 * ((uint64_t)c * n + u)));
        std::vector<uint64_t> pts(8ull * pk->n_advice);
        GL355_TRY(commit_columns(pk, pk->g_lagrange, adv_vals, pk->n_advice, pts.data(), u));
        for (uint32_t c = 0; c < pk->n_advice; c++) tr.write_point(pts.data() + 8 * c);
        for (uint32_t c = 0; c < pk->n_advice; c++) GL355_TRY(lagrange_to_coeff(pk, adv_vals + 4ull * c * n, adv_polys + 4ull * c * n, work));
    }
    // column pointer tables (values / coefficient forms) on the device
    auto col_vals = [&](uint32_t kind, uint32_t idx) -> const uint64_t* { return (kind == 0 ? adv_vals : (kind == 1 ? pk->fixed_vals : inst_vals)) + 4ull * idx * n; };
    const uint32_t kind_cols[3] = {pk->n_advice, pk->n_fixed, pk->n_instance};
    const uint32_t n_all_cols = pk->n_advice + pk->n_fixed + pk->n_instance;
    uint64_t* d_ptrs = nullptr;          // pointer tables: [values: adv | fix | inst][coset: adv | fix | inst][perm vals][perm sigma vals][coset perm cols][coset sigma][coset z]
    GL355_TRY(D((size_t)(2 * n_all_cols + 5 * pk->n_perm + pk->n_sets + 16) * 8, &d_ptrs));
    const uint64_t** dp = (const uint64_t**)d_ptrs;
    const uint64_t* const* d_val_cols[3];
    {
        std::vector<const uint64_t*> v;
        for (uint32_t kd = 0; kd < 3; kd++) for (uint32_t c = 0; c < kind_cols[kd]; c++) v.push_back(col_vals(kd, c));
        GL355_TRY(ptrs_to_device(ctx, v, dp));
        d_val_cols[0] = dp; d_val_cols[1] = dp + pk->n_advice; d_val_cols[2] = dp + pk->n_advice + pk->n_fixed;
    }

    // ---- lookups: compress, permute, commit ----------------------------------------------------------------------------------------
    const Fr theta = tr.squeeze_challenge();
    const uint32_t L = pk->n_lookups;
    uint64_t *lkA = nullptr, *lkS = nullptr, *lkAp = nullptr, *lkZ = nullptr, *lk_polys = nullptr /* [L][3]: A', S', z */;
    // compressed input / table columns only for the lookups whose expressions are not a single column query (none of the reference's nine)
    uint32_t n_lk_a = 0, n_lk_s = 0;
    for (uint32_t l = 0; l < L; l++) {
        if (single_query(pk, pk->lookups[l].in_code).first >= 3) n_lk_a++;
        if (single_query(pk, pk->lookups[l].tab_code).first >= 3) n_lk_s++;
    }
    GL355_TRY(D((size_t)std::max(1u, n_lk_a) * n * 32, &lkA));
    GL355_TRY(D((size_t)std::max(1u, n_lk_s) * n * 32, &lkS));
    GL355_TRY(D((size_t)std::max(1u, 2 * L) * n * 32, &lkAp));          // A'_0 S'_0 A'_1 S'_1 ... : one batched commitment
    GL355_TRY(D((size_t)std::max(1u, L) * n * 32, &lkZ));
    GL355_TRY(D((size_t)std::max(1u, 3 * L) * n * 32, &lk_polys));
    std::vector<const uint64_t*> lk_a_ptr(L), lk_s_ptr(L);         // the compressed input / table columns (values)
    {
        uint32_t ia = 0, is = 0;
        for (uint32_t l = 0; l < L; l++) {
            lk_a_ptr[l] = single_query(pk, pk->lookups[l].in_code).first >= 3 ? lkA + 4ull * (ia++) * n : nullptr;
            lk_s_ptr[l] = single_query(pk, pk->lookups[l].tab_code).first >= 3 ? lkS + 4ull * (is++) * n : nullptr;
        }
    }
    {
        Timer t(ctx, slot(GL355_PLONK_STAGE_LOOKUP_PERMUTE), "lookup_permute");
        for (uint32_t l = 0; l < L; l++) {
            const auto qa = single_query(pk, pk->lookups[l].in_code), qs = single_query(pk, pk->lookups[l].tab_code);
            if (qa.first < 3) lk_a_ptr[l] = col_vals(qa.first, qa.second);
            else GL355_TRY(run_program(pk, pk->d_lk_code[2 * l], (uint32_t)(pk->lookups[l].in_code.size() / 4), d_val_cols, theta, nullptr, const_cast<uint64_t*>(lk_a_ptr[l]), false));
            if (qs.first < 3) lk_s_ptr[l] = col_vals(qs.first, qs.second);
            else GL355_TRY(run_program(pk, pk->d_lk_code[2 * l + 1], (uint32_t)(pk->lookups[l].tab_code.size() / 4), d_val_cols, theta, nullptr, const_cast<uint64_t*>(lk_s_ptr[l]), false));
            uint64_t* Ap = lkAp + 8ull * l * n;
            uint64_t* Sp = Ap + 4 * n;
            GL355_TRY(permute_pair(ctx, lk_a_ptr[l], lk_s_ptr[l], u, Ap, Sp));
            GL355_TRY(random_rows(ctx, key, PLK_STREAM_LOOKUP_PERMUTED, 2 * l, u, n - u, Ap + 4 * u));
            GL355_TRY(random_rows(ctx, key, PLK_STREAM_LOOKUP_PERMUTED, 2 * l + 1, u, n - u, Sp + 4 * u));
        }
        std::vector<uint64_t> pts(16ull * L);
        GL355_TRY(commit_columns(pk, pk->g_lagrange, lkAp, 2 * L, pts.data(), u));
        for (uint32_t l = 0; l < 2 * L; l++) tr.write_point(pts.data() + 8 * l);
        for (uint32_t l = 0; l < L; l++) {
            GL355_TRY(lagrange_to_coeff(pk, lkAp + 8ull * l * n, lk_polys + 12ull * l * n, work));
            GL355_TRY(lagrange_to_coeff(pk, lkAp + 8ull * l * n + 4 * n, lk_polys + 12ull * l * n + 4 * n, work));
        }
    }

    // ---- permutation grand products ----------------------------------------------------------------------------------------------------
    const Fr beta = tr.squeeze_challenge();
    const Fr gamma = tr.squeeze_challenge();
    uint64_t *perm_z = nullptr, *perm_polys = nullptr, *num = nullptr, *den = nullptr, *ratio = nullptr, *d_small = nullptr;
    GL355_TRY(D((size_t)std::max(1u, pk->n_sets) * n * 32, &perm_z));
    GL355_TRY(D((size_t)std::max(1u, pk->n_sets) * n * 32, &perm_polys));
    GL355_TRY(D(n * 32, &num));
    GL355_TRY(D(n * 32, &den));
    GL355_TRY(D(n * 32, &ratio));
    GL355_TRY(D(256, &d_small));                     // [0..3] the constant one, [4] a flag word
    uint32_t* d_bad = reinterpret_cast<uint32_t*>(d_small + 4);
    {
        std::vector<Fr> one(1, Fr::one());
        GL355_TRY(fr_to_device(ctx, one, d_small));
        GL355_HIP(ctx, hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    }
    const uint64_t** d_perm_vals = dp + 2 * n_all_cols;
    const uint64_t** d_perm_sig = d_perm_vals + pk->n_perm;
    {
        Timer t(ctx, slot(GL355_PLONK_STAGE_PERMUTATION), "permutation");
        std::vector<const uint64_t*> pv, psg;
        for (uint32_t j = 0; j < pk->n_perm; j++) { pv.push_back(col_vals(pk->perm_cols[j].first, pk->perm_cols[j].second)); psg.push_back(pk->sigma_vals + 4ull * j * n); }
        GL355_TRY(ptrs_to_device(ctx, pv, d_perm_vals));
        GL355_TRY(ptrs_to_device(ctx, psg, d_perm_sig));
        for (uint32_t s = 0; s < pk->n_sets; s++) {
            PlkPermRowArgs a;
            a.n = n; a.j0 = s * pk->chunk_len; a.j1 = std::min(pk->n_perm, (s + 1) * pk->chunk_len);
            a.vals = d_perm_vals; a.sigma = d_perm_sig; a.omega_pows = pk->omega_pows; a.delta_pows = pk->delta_pows;
            a.beta = to_dev(beta); a.gamma = to_dev(gamma); a.num = num; a.den = den;
            hipLaunchKernelGGL(plk_perm_rows_kernel, dim3(blocks(n)), dim3(256), 0, ctx->stream, a);
            hipLaunchKernelGGL(plk_batch_div_kernel, dim3(blocks((n + PLK_INV_CHUNK - 1) / PLK_INV_CHUNK, 64)), dim3(64), 0, ctx->stream, (const uint64_t*)num, den, ratio, n, d_bad);
            GL355_HIP(ctx, hipGetLastError());
            uint64_t* z = perm_z + 4ull * s * n;
            // z_s[0] = z_{s-1}[usable]: read before the blinding of set s overwrites nothing of set s - 1 (its rows > usable only)
            GL355_TRY(running_product(ctx, ratio, n, s ? perm_z + 4 * ((uint64_t)(s - 1) * n + u) : d_small, z));
            GL355_TRY(random_rows(ctx, key, PLK_STREAM_PERM_Z, s, u + 1, n - u - 1, z + 4 * (u + 1)));
        }
        std::vector<uint64_t> pts(8ull * pk->n_sets);
        GL355_TRY(commit_columns(pk, pk->g_lagrange, perm_z, pk->n_sets, pts.data()));
        for (uint32_t s = 0; s < pk->n_sets; s++) tr.write_point(pts.data() + 8 * s);
        for (uint32_t s = 0; s < pk->n_sets; s++) GL355_TRY(lagrange_to_coeff(pk, perm_z + 4ull * s * n, perm_polys + 4ull * s * n, work));
    }
    // ---- lookup grand products --------------------------------------------------------------------------------------------------------
    {
        Timer t(ctx, slot(GL355_PLONK_STAGE_LOOKUP_PRODUCT), "lookup_product");
        for (uint32_t l = 0; l < L; l++) {
            const uint64_t* Ap = lkAp + 8ull * l * n;
            hipLaunchKernelGGL(plk_lookup_rows_kernel, dim3(blocks(n)), dim3(256), 0, ctx->stream, lk_a_ptr[l], lk_s_ptr[l], Ap, Ap + 4 * n, n,
                               to_dev(beta), to_dev(gamma), num, den);
            hipLaunchKernelGGL(plk_batch_div_kernel, dim3(blocks((n + PLK_INV_CHUNK - 1) / PLK_INV_CHUNK, 64)), dim3(64), 0, ctx->stream, (const uint64_t*)num, den, ratio, n, d_bad);
            GL355_HIP(ctx, hipGetLastError());
            uint64_t* z = lkZ + 4ull * l * n;
            GL355_TRY(running_product(ctx, ratio, n, d_small, z));
            GL355_TRY(random_rows(ctx, key, PLK_STREAM_LOOKUP_Z, l, u + 1, n - u - 1, z + 4 * (u + 1)));
        }
        std::vector<uint64_t> pts(8ull * L);
        GL355_TRY(commit_columns(pk, pk->g_lagrange, lkZ, L, pts.data()));
        for (uint32_t l = 0; l < L; l++) tr.write_point(pts.data() + 8 * l);
        for (uint32_t l = 0; l < L; l++) GL355_TRY(lagrange_to_coeff(pk, lkZ + 4ull * l * n, lk_polys + 12ull * l * n + 8 * n, work));
        uint32_t bad = 0;
        GL355_HIP(ctx, ctx->d2h(&bad, d_bad, 4));
        GL355_HIP(ctx, ctx->wait());
        if (bad) return ctx->fail(GL355_E_INVALID_ARG, "plonk_prove: a grand-product denominator is zero under these challenges (retry with another transcript input)");
    }
    // the value forms have done their work (commitments, grand products): from here on only coefficient forms are read
    Free(adv_vals); Free(lkA); Free(lkS); Free(lkAp); Free(lkZ); Free(perm_z); Free(num); Free(den); Free(ratio);
    // ---- vanishing argument: the random polynomial -----------------------------------------------------------------------------------------
    uint64_t* random_poly = nullptr;
    GL355_TRY(D(n * 32, &random_poly));
    {
        Timer t(ctx, slot(GL355_PLONK_STAGE_VANISHING_RANDOM), "vanishing_random");
        GL355_TRY(random_rows(ctx, key, PLK_STREAM_RANDOM_POLY, 0, 0, n, random_poly));
        uint64_t pt[8];
        GL355_TRY(commit_columns(pk, pk->g, random_poly, 1, pt));
        tr.write_point(pt);
    }

    // ---- evaluate_h -----------------------------------------------------------------------------------------------------------------------
    const Fr y = tr.squeeze_challenge();
    uint64_t *h_ext = nullptr, *acc = nullptr, *cos = nullptr, *a_in = nullptr, *s_in = nullptr, *pre = nullptr, *fix_tmp = nullptr;
    const uint32_t n_dyn = pk->n_advice + pk->n_instance + pk->n_sets + 3 * L;           // per-proof polynomials: advice | instance | perm z | lookups (A' S' z)
    GL355_TRY(D((size_t)pk->n_pieces * n * 32, &h_ext));              // [n_pieces] the quotient restricted to each coset
    GL355_TRY(D(n * 32, &acc));
    GL355_TRY(D((size_t)std::max(1u, n_dyn) * n * 32, &cos));
    GL355_TRY(D(n * 32, &a_in));
    GL355_TRY(D(n * 32, &s_in));
    GL355_TRY(D(n * 32, &pre));
    if (!pk->fixed_cos) GL355_TRY(D((size_t)pk->n_fix_cos * n * 32, &fix_tmp));
    {
        Timer t(ctx, slot(GL355_PLONK_STAGE_EVALUATE_H), "evaluate_h");
        std::vector<const uint64_t*> src;
        for (uint32_t c = 0; c < pk->n_advice; c++) src.push_back(adv_polys + 4ull * c * n);
        for (uint32_t c = 0; c < pk->n_instance; c++) src.push_back(inst_polys + 4ull * c * n);
        for (uint32_t s = 0; s < pk->n_sets; s++) src.push_back(perm_polys + 4ull * s * n);
        for (uint32_t l = 0; l < 3 * L; l++) src.push_back(lk_polys + 4ull * l * n);
        auto dyn = [&](uint32_t i) -> const uint64_t* { return cos + 4ull * i * n; };
        const uint32_t o_inst = pk->n_advice, o_z = o_inst + pk->n_instance, o_lk = o_z + pk->n_sets;
        const uint64_t** d_cos_cols = dp + n_all_cols;
        const uint64_t* const* d_cos_kind[3] = {d_cos_cols, d_cos_cols + pk->n_advice, d_cos_cols + pk->n_advice + pk->n_fixed};
        const uint64_t** d_cperm = d_perm_sig + pk->n_perm;
        const uint64_t** d_csig = d_cperm + pk->n_perm;
        const uint64_t** d_cz = d_csig + pk->n_perm;
        const Fr ext_omega = Fr::root_of_unity(pk->ext_k);

        Fr base = plonk_zeta();                                    // zeta * ext_omega^c
        for (uint32_t c = 0; c < n_cosets; c++) {
            // the key's polynomials on this coset: precomputed, or recomputed into fix_tmp
            const uint64_t* fix = pk->fixed_cos ? pk->fixed_cos + 4ull * (uint64_t)c * pk->n_fix_cos * n : fix_tmp;
            if (!pk->fixed_cos) GL355_TRY(plonk_fixed_cosets(pk, c, c + 1, fix_tmp, pre, work));
            auto fixc = [&](uint32_t i) -> const uint64_t* { return fix + 4ull * i * n; };
            auto col_cos = [&](uint32_t kind, uint32_t idx) -> const uint64_t* { return kind == 0 ? dyn(idx) : (kind == 1 ? fixc(idx) : dyn(o_inst + idx)); };
            {
                std::vector<const uint64_t*> v;
                for (uint32_t kd = 0; kd < 3; kd++) for (uint32_t i = 0; i < kind_cols[kd]; i++) v.push_back(col_cos(kd, i));
                GL355_TRY(ptrs_to_device(ctx, v, d_cos_cols));
                v.clear();
                for (uint32_t j = 0; j < pk->n_perm; j++) v.push_back(col_cos(pk->perm_cols[j].first, pk->perm_cols[j].second));
                GL355_TRY(ptrs_to_device(ctx, v, d_cperm));
                v.clear();
                for (uint32_t j = 0; j < pk->n_perm; j++) v.push_back(fixc(pk->n_fixed + j));
                GL355_TRY(ptrs_to_device(ctx, v, d_csig));
                v.clear();
                for (uint32_t s = 0; s < pk->n_sets; s++) v.push_back(dyn(o_z + s));
                GL355_TRY(ptrs_to_device(ctx, v, d_cz));
            }
            const uint64_t *c_l0 = fixc(pk->n_fixed + pk->n_perm), *c_ll = c_l0 + 4 * n, *c_la = c_l0 + 8 * n;
            uint64_t bw[4];
            base.to_words(bw);
            // (round 6: the coset transform in block form -- no power table, no product per coefficient, broadcast twiddles in the high stages)
            for (uint32_t i = 0; i < n_dyn; i++) GL355_TRY(bn254_fr_ntt_mont_coset_dif(ctx, src[i], n, cos + 4ull * i * n, pk->k, pk->tw_fwd, bw, pre, i == 0));
            // custom gates
            GL355_TRY(run_program(pk, pk->d_gate_code, (uint32_t)(pk->gate_code.size() / 4), d_cos_kind, y, nullptr, acc, true));
            if (pk->n_sets) {
                PlkPermHArgs a;
                a.n = n; a.log_n = pk->k; a.n_sets = pk->n_sets; a.chunk_len = pk->chunk_len; a.n_perm = pk->n_perm; a.last_rot = last_rot; a.acc = acc;
                a.l0 = c_l0; a.l_last = c_ll; a.l_active = c_la;
                a.z = d_cz; a.sigma = d_csig; a.col = d_cperm; a.omega_pows = pk->omega_pows; a.delta_pows = pk->delta_pows;
                a.y = to_dev(y); a.beta = to_dev(beta); a.gamma = to_dev(gamma); a.coset_base = to_dev(base);
                hipLaunchKernelGGL(plk_perm_h_kernel, dim3(blocks(n)), dim3(256), 0, ctx->stream, a);
                GL355_HIP(ctx, hipGetLastError());
            }
            for (uint32_t l = 0; l < L; l++) {
                const auto qa = single_query(pk, pk->lookups[l].in_code), qs = single_query(pk, pk->lookups[l].tab_code);
                if (qa.first == 3) GL355_TRY(run_program(pk, pk->d_lk_code[2 * l], (uint32_t)(pk->lookups[l].in_code.size() / 4), d_cos_kind, theta, nullptr, a_in, true));
                if (qs.first == 3) GL355_TRY(run_program(pk, pk->d_lk_code[2 * l + 1], (uint32_t)(pk->lookups[l].tab_code.size() / 4), d_cos_kind, theta, nullptr, s_in, true));
                PlkLookupHArgs a;
                a.n = n; a.log_n = pk->k; a.acc = acc; a.l0 = c_l0; a.l_last = c_ll; a.l_active = c_la;
                a.ap = dyn(o_lk + 3 * l); a.sp = dyn(o_lk + 3 * l + 1); a.z = dyn(o_lk + 3 * l + 2);
                a.a_in = qa.first < 3 ? col_cos(qa.first, qa.second) : a_in; a.s_in = qs.first < 3 ? col_cos(qs.first, qs.second) : s_in;
                a.y = to_dev(y); a.beta = to_dev(beta); a.gamma = to_dev(gamma);
                hipLaunchKernelGGL(plk_lookup_h_kernel, dim3(blocks(n)), dim3(256), 0, ctx->stream, a);
                GL355_HIP(ctx, hipGetLastError());
            }
            const Fr t_inv = (base.pow_u64(n) - Fr::one()).inv();                 // 1 / ((zeta omega_ext^c)^n - 1)
            hipLaunchKernelGGL(plk_finish_h_kernel, dim3(blocks(n)), dim3(256), 0, ctx->stream, (const uint64_t*)acc, n, (uint64_t)c, to_dev(t_inv), h_ext);
            GL355_HIP(ctx, hipGetLastError());
            base = base * ext_omega;
        }
    }
    // ---- vanishing argument: h's coefficients, pieces, commitments ---------------------------------------------------------------------------
    // halo2 evaluates h on the whole extended domain (2^(extended_k - k) cosets of the 2^k domain) and inverts one transform of that size.  The
    // same polynomial follows from `degree - 1` cosets: h = sum_p X^(p n) h_p with deg h_p < n, and on the coset g_c H, X^n is the constant
    // t_c = g_c^n, so h restricted to the coset IS the polynomial R_c = sum_p t_c^p h_p (degree < n): one inverse coset transform of size n
    // per coset gives R_c, and the pieces are h_p = sum_c (V^-1)[p][c] R_c with the Vandermonde matrix V[c][p] = t_c^p -- a host-side inversion
    // of a (degree - 1)^2 matrix and one linear combination per piece.  5 of 8 cosets at the reference's degree 6: 3/8 of evaluate_h's
    // transforms and kernels gone, no transform over the extended domain at all, same bytes (the quotient is unique).
    Free(cos); Free(acc); Free(a_in); Free(s_in); Free(fix_tmp);
    uint64_t* h_coeffs = nullptr;                                      // [n_pieces] the quotient's pieces
    GL355_TRY(D((size_t)pk->n_pieces * n * 32, &h_coeffs));
    {
        Timer t(ctx, slot(GL355_PLONK_STAGE_QUOTIENT_COMMIT), "quotient_commit");
        const uint32_t P = pk->n_pieces;
        const Fr ext_omega = Fr::root_of_unity(pk->ext_k), n_inv = Fr::from_u64(n).inv();
        std::vector<Fr> tc(P);
        Fr base = plonk_zeta();
        for (uint32_t c = 0; c < P; c++) {
            // R_c: values on g_c H in bit-reversed order -> coefficients: inverse transform (no gather), coefficient i times g_c^-i / n
            uint64_t bw[4], fw[4];
            base.inv().to_words(bw);
            n_inv.to_words(fw);
            GL355_TRY(bn254_fr_power_table(ctx, bw, fw, n, pre));
            GL355_TRY(bn254_fr_ntt_mont_from_bitrev(ctx, h_ext + 4ull * c * n, h_ext + 4ull * c * n, n, pk->k, pk->tw_inv, pre, nullptr));
            tc[c] = base.pow_u64(n);
            base = base * ext_omega;
        }
        // V^-1 by Gauss-Jordan over Fr (P <= 9)
        std::vector<std::vector<Fr>> M(P, std::vector<Fr>(2 * P, Fr::zero()));
        for (uint32_t c = 0; c < P; c++) {
            Fr pw = Fr::one();
            for (uint32_t q = 0; q < P; q++) { M[c][q] = pw; pw = pw * tc[c]; }
            M[c][P + c] = Fr::one();
        }
        for (uint32_t col = 0; col < P; col++) {
            uint32_t piv = col;
            while (piv < P && M[piv][col].is_zero()) piv++;
            if (piv == P) return ctx->fail(GL355_E_HIP, "plonk_prove: singular coset matrix (internal)");
            std::swap(M[piv], M[col]);
            const Fr inv = M[col][col].inv();
            for (uint32_t j = 0; j < 2 * P; j++) M[col][j] = M[col][j] * inv;
            for (uint32_t r = 0; r < P; r++) {
                if (r == col || M[r][col].is_zero()) continue;
                const Fr f = M[r][col];
                for (uint32_t j = 0; j < 2 * P; j++) M[r][j] = M[r][j] - f * M[col][j];
            }
        }
        std::vector<const uint64_t*> rs;
        for (uint32_t c = 0; c < P; c++) rs.push_back(h_ext + 4ull * c * n);
        for (uint32_t q = 0; q < P; q++) {
            std::vector<Fr> cs(P);
            for (uint32_t c = 0; c < P; c++) cs[c] = M[q][P + c];          // (V^-1)[q][c]
            GL355_TRY(lincomb(pk, rs, cs, {}, nullptr, h_coeffs + 4ull * q * n));
        }
        if (export_quotient) {             // gl355_plonk_pk_export_quotient: the pieces as canonical integers, for a comparison with the extended-domain quotient
            uint64_t* plain = nullptr;
            GL355_TRY(D((size_t)P * n * 32, &plain));
            hipLaunchKernelGGL(plk_from_mont_kernel, dim3(blocks((uint64_t)P * n)), dim3(256), 0, ctx->stream, (const uint64_t*)h_coeffs, plain, (uint64_t)P * n);
            GL355_HIP(ctx, hipGetLastError());
            GL355_HIP(ctx, ctx->d2h(export_quotient, plain, (size_t)P * n * 32));
            GL355_HIP(ctx, ctx->wait());
        }
        std::vector<uint64_t> pts(8ull * pk->n_pieces);
        GL355_TRY(commit_columns(pk, pk->g, h_coeffs, pk->n_pieces, pts.data()));
        for (uint32_t i = 0; i < pk->n_pieces; i++) tr.write_point(pts.data() + 8 * i);
    }

    // ---- evaluations at x ----------------------------------------------------------------------------------------------------------------------
    const Fr x = tr.squeeze_challenge();
    const Fr xn = x.pow_u64(n);
    const Fr x_next = rotate(x, 1), x_last = rotate(x, last_rot), x_inv = rotate(x, -1);
    // (label, polynomial, point): every evaluation the proof carries, in transcript order, then h and the random polynomial
    struct Q { uint32_t poly_id; const uint64_t* poly; Fr point; };
    std::vector<Q> qs;
    // polynomial ids: one per committed polynomial (the SHPLONK grouping is by polynomial)
    auto id_adv = [&](uint32_t c) { return c; };
    auto id_fix = [&](uint32_t c) { return 1000 + c; };
    auto id_sig = [&](uint32_t j) { return 2000 + j; };
    auto id_pz = [&](uint32_t s) { return 3000 + s; };
    auto id_lk = [&](uint32_t l, uint32_t w) { return 4000 + 3 * l + w; };          // w: 0 A', 1 S', 2 z
    const uint32_t ID_H = 9000, ID_RANDOM = 9001;
    uint64_t* h_poly = nullptr;
    GL355_TRY(D(n * 32, &h_poly));
    std::vector<Fr> evals;
    {
        Timer t(ctx, slot(GL355_PLONK_STAGE_EVALUATIONS), "evaluations");
        // h(X) = sum_i x^(n i) h_i(X)
        {
            std::vector<const uint64_t*> ps;
            std::vector<Fr> cs;
            Fr p = Fr::one();
            for (uint32_t i = 0; i < pk->n_pieces; i++) { ps.push_back(h_coeffs + 4ull * i * n); cs.push_back(p); p = p * xn; }
            GL355_TRY(lincomb(pk, ps, cs, {}, nullptr, h_poly));
        }
        for (auto& q : pk->queries[0]) qs.push_back({id_adv(q.first), adv_polys + 4ull * q.first * n, rotate(x, q.second)});
        for (auto& q : pk->queries[1]) qs.push_back({id_fix(q.first), pk->fixed_polys + 4ull * q.first * n, rotate(x, q.second)});
        qs.push_back({ID_RANDOM, random_poly, x});
        for (uint32_t j = 0; j < pk->n_perm; j++) qs.push_back({id_sig(j), pk->sigma_polys + 4ull * j * n, x});
        for (uint32_t s = 0; s < pk->n_sets; s++) {
            qs.push_back({id_pz(s), perm_polys + 4ull * s * n, x});
            qs.push_back({id_pz(s), perm_polys + 4ull * s * n, x_next});
            if (s + 1 < pk->n_sets) qs.push_back({id_pz(s), perm_polys + 4ull * s * n, x_last});
        }
        for (uint32_t l = 0; l < L; l++) {
            const uint64_t *Ap = lk_polys + 12ull * l * n, *Sp = Ap + 4 * n, *Z = Ap + 8 * n;
            qs.push_back({id_lk(l, 2), Z, x});
            qs.push_back({id_lk(l, 2), Z, x_next});
            qs.push_back({id_lk(l, 0), Ap, x});
            qs.push_back({id_lk(l, 0), Ap, x_inv});
            qs.push_back({id_lk(l, 1), Sp, x});
        }
        const size_t n_written = qs.size();
        qs.push_back({ID_H, h_poly, x});
        std::vector<const uint64_t*> ps;
        std::vector<Fr> pts;
        for (auto& q : qs) { ps.push_back(q.poly); pts.push_back(q.point); }
        GL355_TRY(eval_polys(pk, ps, pts, evals));
        for (size_t i = 0; i < n_written; i++) tr.write_scalar(evals[i]);
    }
    // evaluation of (poly id, point) from the table above
    bool eval_missing = false;              // a (polynomial, point) pair SHPLONK asks for that the evaluation stage never computed: an internal error, not a zero
    auto eval_of = [&](uint32_t id, const Fr& pt) -> Fr {
        for (size_t i = 0; i < qs.size(); i++) if (qs[i].poly_id == id && qs[i].point == pt) return evals[i];
        eval_missing = true;
        return Fr::zero();
    };

    // ---- SHPLONK -------------------------------------------------------------------------------------------------------------------------------
    {
        Timer t(ctx, slot(GL355_PLONK_STAGE_SHPLONK), "shplonk");
        // the opening queries in create_proof's order
        struct OQ { uint32_t id; const uint64_t* poly; Fr point; };
        std::vector<OQ> oq;
        for (auto& q : pk->queries[0]) oq.push_back({id_adv(q.first), adv_polys + 4ull * q.first * n, rotate(x, q.second)});
        for (uint32_t s = 0; s < pk->n_sets; s++) { oq.push_back({id_pz(s), perm_polys + 4ull * s * n, x}); oq.push_back({id_pz(s), perm_polys + 4ull * s * n, x_next}); }
        for (uint32_t s = pk->n_sets; s-- > 0;) if (s + 1 < pk->n_sets) oq.push_back({id_pz(s), perm_polys + 4ull * s * n, x_last});
        for (uint32_t l = 0; l < L; l++) {
            const uint64_t *Ap = lk_polys + 12ull * l * n, *Sp = Ap + 4 * n, *Z = Ap + 8 * n;
            oq.push_back({id_lk(l, 2), Z, x}); oq.push_back({id_lk(l, 0), Ap, x}); oq.push_back({id_lk(l, 1), Sp, x});
            oq.push_back({id_lk(l, 0), Ap, x_inv}); oq.push_back({id_lk(l, 2), Z, x_next});
        }
        for (auto& q : pk->queries[1]) oq.push_back({id_fix(q.first), pk->fixed_polys + 4ull * q.first * n, rotate(x, q.second)});
        for (uint32_t j = 0; j < pk->n_perm; j++) oq.push_back({id_sig(j), pk->sigma_polys + 4ull * j * n, x});
        oq.push_back({ID_H, h_poly, x});
        oq.push_back({ID_RANDOM, random_poly, x});
        const Fr sy = tr.squeeze_challenge();
        const Fr sv = tr.squeeze_challenge();
        // construct_intermediate_sets: polynomials in first-appearance order with their sorted point sets; rotation sets = the distinct
        // point sets in first-appearance order
        auto less = [](const Fr& a, const Fr& b) { return a.less_than(b); };
        struct PolyPts { uint32_t id; const uint64_t* poly; std::vector<Fr> pts; };
        std::vector<PolyPts> polys;
        for (auto& q : oq) {
            size_t i = 0;
            for (; i < polys.size(); i++) if (polys[i].id == q.id) break;
            if (i == polys.size()) polys.push_back({q.id, q.poly, {}});
            bool have = false;
            for (auto& p : polys[i].pts) have = have || p == q.point;
            if (!have) polys[i].pts.push_back(q.point);
        }
        struct RSet { std::vector<Fr> pts; std::vector<size_t> members; };
        std::vector<RSet> sets;
        std::vector<Fr> super;
        for (size_t i = 0; i < polys.size(); i++) {
            std::sort(polys[i].pts.begin(), polys[i].pts.end(), less);
            size_t s = 0;
            for (; s < sets.size(); s++) if (sets[s].pts == polys[i].pts) break;
            if (s == sets.size()) sets.push_back({polys[i].pts, {}});
            sets[s].members.push_back(i);
            for (auto& p : polys[i].pts) { bool have = false; for (auto& q : super) have = have || q == p; if (!have) super.push_back(p); }
        }
        std::sort(super.begin(), super.end(), less);
        // h(X) = sum_i v^i (sum_j y^j (p_ij(X) - r_ij(X))) / prod_{t in S_i} (X - t)
        uint64_t *sh_h = nullptr, *bufa = nullptr, *bufb = nullptr;
        GL355_TRY(D(n * 32, &sh_h));
        GL355_TRY(D(n * 32, &bufa));
        GL355_TRY(D(n * 32, &bufb));
        GL355_HIP(ctx, hipMemsetAsync(sh_h, 0, n * 32, ctx->stream));
        std::vector<std::vector<std::vector<Fr>>> r_coeffs(sets.size());      // [set][member] low-degree remainders
        Fr vi = Fr::one();
        for (size_t s = 0; s < sets.size(); s++) {
            std::vector<const uint64_t*> ps;
            std::vector<Fr> cs, low(sets[s].pts.size(), Fr::zero());
            Fr yj = Fr::one();
            for (size_t m : sets[s].members) {
                std::vector<Fr> ev;
                for (auto& p : sets[s].pts) ev.push_back(eval_of(polys[m].id, p));
                if (eval_missing) return ctx->fail(GL355_E_UNSUPPORTED, "plonk_prove: an opening asks for an evaluation that was not computed");
                r_coeffs[s].push_back(interpolate(sets[s].pts, ev));
                for (size_t t = 0; t < low.size(); t++) low[t] = low[t] + yj * r_coeffs[s].back()[t];
                ps.push_back(polys[m].poly);
                cs.push_back(yj);
                yj = yj * sy;
            }
            GL355_TRY(lincomb(pk, ps, cs, low, nullptr, bufa));
            uint64_t *src_ = bufa, *dst_ = bufb;
            for (auto& p : sets[s].pts) {
                uint64_t pw[4];
                p.to_words(pw);
                GL355_TRY(kzg_divide(ctx, src_, n, h_from_words(pw), 1, dst_, 0, nullptr));
                std::swap(src_, dst_);
            }
            GL355_TRY(lincomb(pk, {src_}, {vi}, {}, sh_h, sh_h));
            vi = vi * sv;
        }
        uint64_t pt[8];
        GL355_TRY(commit_columns(pk, pk->g, sh_h, 1, pt));
        tr.write_point(pt);
        const Fr su = tr.squeeze_challenge();
        // L(X) = (sum_i v^i Z_{T \ S_i}(u) sum_j y^j (p_ij(X) - r_ij(u)) - Z_T(u) h(X)) / Z_{T \ S_0}(u)
        Fr zt = Fr::one();
        for (auto& p : super) zt = zt * (su - p);
        std::vector<Fr> zdiff(sets.size(), Fr::one());
        for (size_t s = 0; s < sets.size(); s++)
            for (auto& p : super) { bool in = false; for (auto& q : sets[s].pts) in = in || q == p; if (!in) zdiff[s] = zdiff[s] * (su - p); }
        const Fr z0inv = zdiff[0].inv();
        std::vector<const uint64_t*> ps;
        std::vector<Fr> cs;
        Fr cterm = Fr::zero();
        vi = Fr::one();
        for (size_t s = 0; s < sets.size(); s++) {
            const Fr scale = vi * zdiff[s] * z0inv;
            Fr yj = Fr::one();
            for (size_t mi = 0; mi < sets[s].members.size(); mi++) {
                ps.push_back(polys[sets[s].members[mi]].poly);
                cs.push_back(scale * yj);
                cterm = cterm + scale * yj * horner(r_coeffs[s][mi], su);
                yj = yj * sy;
            }
            vi = vi * sv;
        }
        ps.push_back(sh_h);
        cs.push_back((zt * z0inv).neg());
        GL355_TRY(lincomb(pk, ps, cs, {cterm}, nullptr, bufa));
        uint64_t uw[4];
        su.to_words(uw);
        GL355_TRY(kzg_divide(ctx, bufa, n, h_from_words(uw), 1, bufb, 0, nullptr));
        GL355_TRY(commit_columns(pk, pk->g, bufb, 1, pt));
        tr.write_point(pt);
        if (trace) { sy.to_words(trace + 20); sv.to_words(trace + 24); su.to_words(trace + 28); }
    }
    if (trace) { theta.to_words(trace); beta.to_words(trace + 4); gamma.to_words(trace + 8); y.to_words(trace + 12); x.to_words(trace + 16); }
    GL355_HIP(ctx, ctx->wait());
    *proof_len = tr.proof.size();
    if (tr.proof.size() > capacity) return ctx->fail(GL355_E_INVALID_ARG, "plonk_prove: proof buffer too small (gl355_plonk_pk_info gives the size)");
    memcpy(proof, tr.proof.data(), tr.proof.size());
    if (stage_ms) for (int i = 0; i < GL355_PLONK_STAGES; i++) stage_ms[i] = ms[i];
    return GL355_OK;
}

}  // extern "C"
