// Witness tape: a straight-line program that recomputes every wire of a circuit's non-trivial rows from
// a flat input vector.  It plays the role of plonky2's witness generators (`generate_partial_witness`,
// run inside `data.prove(pw)` at src/plonky2_semaphore/recursion.rs:167-168 and wrapper.rs:55): the
// recursive-verifier circuit's ~6000 gate rows (Poseidon, Reducing, Arithmetic, RandomAccess, BaseSum,
// ...) are a fixed function of the inner proofs' words.  The tape is recorded once by the host-side
// builder (stark-verifier_amd/gadgets.py) and replayed here per proof: sequential host work
// (about 5k Poseidon permutations with their S-box-input wires + ~0.5M field operations), single thread,
// no device involvement -- callers overlap it with the GPU prove of the previous proof.
//
// Wire layouts of the gate rows follow the reference's chip/plonk/gates/*.rs (arithmetic.rs,
// arithmetic_extension.rs, poseidon.rs:329-380, poseidon_mds.rs, base_sum.rs, random_access.rs,
// reducing.rs, reducing_extension.rs).
#include "gl355_internal.h"

#include <atomic>
#include <thread>
#include <vector>

using namespace gl355;

namespace {
constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
}

// entries [begin, end) of the tape on the (already initialised) rows
static int32_t run_entries(const uint64_t* tape, uint64_t begin, uint64_t end, const uint64_t* inputs, uint64_t n_inputs, uint64_t* rows,
                           uint64_t n_words, uint32_t num_wires, uint64_t* failed_op) {
    uint64_t* W = rows;
    const uint64_t n_ops = end;
#define CHK(i, span)                                                        \
    if ((uint64_t)(i) >= n_words || (uint64_t)(span) > n_words - (uint64_t)(i)) { /* no wrap for offsets near 2^64 */ \
        if (failed_op) *failed_op = t;                                        \
        return GL355_E_INVALID_ARG;                                           \
    }
#define ROWCHK(i)                                                           \
    CHK(i, num_wires)                                                        \
    if ((i) % num_wires) {                                                    \
        if (failed_op) *failed_op = t;                                        \
        return GL355_E_INVALID_ARG;                                           \
    }
    for (uint64_t t = begin; t < n_ops; t++) {
        const uint64_t* e = tape + 5 * t;
        const uint64_t a = e[1], b = e[2], c = e[3], d = e[4];
        switch (e[0]) {
        case GL355_TAPE_CONST:
            CHK(a, 1) W[a] = gl_canon(b);
            break;
        case GL355_TAPE_INPUT:
            CHK(a, 1)
            if (b >= n_inputs) { if (failed_op) *failed_op = t; return GL355_E_INVALID_ARG; }
            W[a] = gl_canon(inputs[b]);
            break;
        case GL355_TAPE_COPY:
            CHK(a, 1) CHK(b, 1) W[a] = W[b];
            break;
        case GL355_TAPE_ASSERT_EQ:
            CHK(a, 1) CHK(b, 1)
            if (W[a] != W[b]) { if (failed_op) *failed_op = t; return GL355_E_WITNESS; }
            break;
        case GL355_TAPE_ARITH:
            CHK(a, 4)
            W[a + 3] = gl_canon(gl_add(gl_mul(gl_mul(W[a], W[a + 1]), b), gl_mul(W[a + 2], c)));
            break;
        case GL355_TAPE_ARITH_EXT: {
            CHK(a, 8)
            gl2 pr = gl2_mul(gl2_make(W[a], W[a + 1]), gl2_make(W[a + 2], W[a + 3]));
            gl2 r = gl2_add(gl2_mul_base(pr, b), gl2_mul_base(gl2_make(W[a + 4], W[a + 5]), c));
            r = gl2_canon(r);
            W[a + 6] = r.c0;
            W[a + 7] = r.c1;
            break;
        }
        case GL355_TAPE_POSEIDON: {
            ROWCHK(a)
            uint64_t in[12];
            memcpy(in, W + a, sizeof in);
            const uint64_t swap = W[a + 24];
            if (swap > 1) { if (failed_op) *failed_op = t; return GL355_E_WITNESS; }
            gl355_poseidon_gate_witness(in, swap, W + a);
            break;
        }
        case GL355_TAPE_MDS_EXT: {
            ROWCHK(a)
            for (int r = 0; r < 12; r++)
                for (int k = 0; k < 2; k++) {
                    unsigned __int128 acc = 0;
                    for (int i = 0; i < 12; i++) acc += (unsigned __int128)W[a + 2 * ((i + r) % 12) + k] * CIRC[i];
                    if (r == 0) acc += (unsigned __int128)W[a + k] * 8;
                    W[a + 2 * (12 + r) + k] = gl_canon(gl_reduce128((uint64_t)acc, (uint64_t)(acc >> 64)));
                }
            break;
        }
        case GL355_TAPE_BASE_SUM: {
            ROWCHK(a)
            if (b > 63 || 1 + b > num_wires || (W[a] >> b)) { if (failed_op) *failed_op = t; return b > 63 ? GL355_E_INVALID_ARG : GL355_E_WITNESS; }
            for (uint64_t i = 0; i < b; i++) W[a + 1 + i] = (W[a] >> i) & 1;
            break;
        }
        case GL355_TAPE_RANDOM_ACCESS: {   // RandomAccessGate{bits 4, copies 4, extra constants 2}
            ROWCHK(a)
            if (b >= 4) { if (failed_op) *failed_op = t; return GL355_E_INVALID_ARG; }
            const uint64_t idx = W[a + 18 * b];
            if (idx >= 16) { if (failed_op) *failed_op = t; return GL355_E_WITNESS; }
            W[a + 18 * b + 1] = W[a + 18 * b + 2 + idx];
            for (int k = 0; k < 4; k++) W[a + 74 + 4 * b + k] = (idx >> k) & 1;
            break;
        }
        case GL355_TAPE_REDUCING: {   // b = number of coefficients, c = 1 for ReducingExtensionGate
            ROWCHK(a)
            const uint64_t n = b, ext = c;
            const uint64_t start_accs = 6 + (ext ? 2 * n : n);
            if (n == 0 || start_accs + 2 * (n - 1) > num_wires) { if (failed_op) *failed_op = t; return GL355_E_INVALID_ARG; }
            const gl2 alpha = gl2_make(W[a + 2], W[a + 3]);
            gl2 acc = gl2_make(W[a + 4], W[a + 5]);
            for (uint64_t i = 0; i < n; i++) {
                const gl2 cf = ext ? gl2_make(W[a + 6 + 2 * i], W[a + 7 + 2 * i]) : gl2_make(W[a + 6 + i], 0);
                acc = gl2_canon(gl2_add(gl2_mul(acc, alpha), cf));
                const uint64_t o = i == n - 1 ? 0 : start_accs + 2 * i;
                W[a + o] = acc.c0;
                W[a + o + 1] = acc.c1;
            }
            break;
        }
        case GL355_TAPE_LO32:
            CHK(a, 1) CHK(b, 1) W[a] = W[b] & 0xFFFFFFFFull;
            break;
        case GL355_TAPE_HI32:
            CHK(a, 1) CHK(b, 1) W[a] = W[b] >> 32;
            break;
        case GL355_TAPE_EXT_INV: {
            CHK(a, 1) CHK(b, 1) CHK(c, 1) CHK(d, 1)
            if (W[c] == 0 && W[d] == 0) { if (failed_op) *failed_op = t; return GL355_E_WITNESS; }
            const gl2 r = gl2_canon(gl2_inv(gl2_make(W[c], W[d])));
            W[a] = r.c0;
            W[b] = r.c1;
            break;
        }
        default:
            if (failed_op) *failed_op = t;
            return GL355_E_INVALID_ARG;
        }
    }
#undef CHK
#undef ROWCHK
    return GL355_OK;
}

// Static check of every offset a tape entry touches (what run_entries checks while it executes): done once when an artifact is
// loaded, so that the device interpreter (witness_tape_dev.hip) can run without bounds checks.  Returns the first bad entry or ~0.
namespace gl355 {
uint64_t tape_validate(const uint64_t* tape, uint64_t n_ops, uint64_t n_inputs, uint64_t n_words, uint32_t num_wires) {
    auto ok = [&](uint64_t i, uint64_t span) { return i < n_words && span <= n_words - i; };
    auto row = [&](uint64_t i) { return ok(i, num_wires) && i % num_wires == 0; };
    if (num_wires < 135) return 0;
    for (uint64_t t = 0; t < n_ops; t++) {
        const uint64_t* e = tape + 5 * t;
        const uint64_t a = e[1], b = e[2], c = e[3], d = e[4];
        bool good = false;
        switch (e[0]) {
        case GL355_TAPE_CONST: good = ok(a, 1); break;
        case GL355_TAPE_INPUT: good = ok(a, 1) && b < n_inputs; break;
        case GL355_TAPE_COPY: case GL355_TAPE_ASSERT_EQ: case GL355_TAPE_LO32: case GL355_TAPE_HI32: good = ok(a, 1) && ok(b, 1); break;
        case GL355_TAPE_ARITH: good = ok(a, 4); break;
        case GL355_TAPE_ARITH_EXT: good = ok(a, 8); break;
        case GL355_TAPE_POSEIDON: case GL355_TAPE_MDS_EXT: good = row(a); break;
        case GL355_TAPE_BASE_SUM: good = row(a) && b <= 63 && 1 + b <= num_wires; break;
        case GL355_TAPE_RANDOM_ACCESS: good = row(a) && b < 4; break;
        case GL355_TAPE_REDUCING: good = row(a) && b != 0 && b <= num_wires && 6 + (c ? 2 * b : b) + 2 * (b - 1) <= num_wires; break;
        case GL355_TAPE_EXT_INV: good = ok(a, 1) && ok(b, 1) && ok(c, 1) && ok(d, 1); break;
        default: good = false;
        }
        if (!good) return t;
    }
    return ~0ull;
}
}  // namespace gl355

extern "C" int32_t gl355_witness_replay(const uint64_t* tape, uint64_t n_ops, const uint64_t* inputs, uint64_t n_inputs,
                                        uint64_t* rows, uint64_t n_words, uint32_t num_wires, uint64_t* failed_op) {
    if (!tape || !rows || (!inputs && n_inputs) || num_wires < 135) return GL355_E_INVALID_ARG;
    if (failed_op) *failed_op = ~0ull;
    memset(rows, 0, n_words * 8);
    return run_entries(tape, 0, n_ops, inputs, n_inputs, rows, n_words, num_wires, failed_op);
}

// The same with the independent segments of a segmented tape (gadgets.py begin_segment / end_segment, e.g. the FRI query
// rounds of a verifier circuit) spread over `threads` host threads: entries [0, n_seq) run first on the calling thread, then
// segment k = the next seg_lens[k] entries.  The builder has checked that a segment reads only the sequential part and
// itself, so the rows are the same as those of the sequential replay; the reported failing entry is the smallest one.
extern "C" int32_t gl355_witness_replay_segmented(const uint64_t* tape, uint64_t n_ops, uint64_t n_seq, const uint64_t* seg_lens, uint32_t n_segs,
                                                  uint32_t threads, const uint64_t* inputs, uint64_t n_inputs, uint64_t* rows, uint64_t n_words,
                                                  uint32_t num_wires, uint64_t* failed_op) {
    if (!tape || !rows || (!inputs && n_inputs) || num_wires < 135 || (!seg_lens && n_segs) || n_seq > n_ops) return GL355_E_INVALID_ARG;
    uint64_t total = n_seq;
    for (uint32_t k = 0; k < n_segs; k++) total += seg_lens[k];
    if (total != n_ops) return GL355_E_INVALID_ARG;
    if (failed_op) *failed_op = ~0ull;
    memset(rows, 0, n_words * 8);
    int32_t rc = run_entries(tape, 0, n_seq, inputs, n_inputs, rows, n_words, num_wires, failed_op);
    if (rc != GL355_OK || n_segs == 0) return rc;
    if (threads <= 1) return run_entries(tape, n_seq, n_ops, inputs, n_inputs, rows, n_words, num_wires, failed_op);
    std::vector<uint64_t> start(n_segs + 1, n_seq);
    for (uint32_t k = 0; k < n_segs; k++) start[k + 1] = start[k] + seg_lens[k];
    const uint32_t nt = threads < n_segs ? threads : n_segs;
    std::vector<int32_t> rcs(nt, GL355_OK);
    std::vector<uint64_t> fails(nt, ~0ull);
    std::atomic<uint32_t> next{0};
    auto worker = [&](uint32_t t) {
        for (;;) {
            const uint32_t k = next.fetch_add(1);
            if (k >= n_segs) return;
            uint64_t f = ~0ull;
            const int32_t r = run_entries(tape, start[k], start[k + 1], inputs, n_inputs, rows, n_words, num_wires, &f);
            if (r != GL355_OK && f < fails[t]) { rcs[t] = r; fails[t] = f; }
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < nt; t++) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    uint64_t best = ~0ull;
    for (uint32_t t = 0; t < nt; t++)
        if (rcs[t] != GL355_OK && fails[t] < best) { best = fails[t]; rc = rcs[t]; }
    if (failed_op && best != ~0ull) *failed_op = best;
    return rc;
}
