// Witness generation of a recorded circuit ON THE DEVICE: the tape of witness_tape.cpp -- plonky2's witness
// generators run inside `data.prove(pw)` (src/plonky2_semaphore/recursion.rs:72-86,167-168, wrapper.rs:49-55) -- interpreted by
// the GPU, so a unit's ~6.4 MB of witness rows are produced where the prover consumes them: the host uploads the inner proof's
// flat words (175 KB) instead of the rows, and spends no core on the 4 174 Poseidon gate rows + ~155 k field operations per unit.
//
// The tape is a straight-line program with a sequential part (the in-circuit transcript, openings, vanishing identity) and 28
// independent segments (the FRI query rounds; the builder has checked that a segment reads only the sequential part and
// itself).  One 64-lane wave interprets one (unit, segment): lane 0 executes the scalar entries in order; a POSEIDON entry --
// a whole PoseidonGate row, wire layout chip/plonk/gates/poseidon.rs:329-380 -- runs the lane-parallel permutation (one state
// element per lane, MDS row through a 24-slot LDS ring) and every lane stores the S-box-input wires it owns.  This is latency-
// bound work on a handful of waves (8 units x 29 waves): tens of milliseconds per batch, but it runs on the context's side stream
// under the proving of the previous batch and takes ~0.1 % of the chip's issue slots.
//
// Offsets are validated once when the artifact is loaded (tape_validate, witness_tape.cpp), so the interpreter does not bounds-
// check; data-dependent failures (ASSERT_EQ, range conditions = an invalid inner proof) are reported as the smallest failing
// entry per unit, exactly like the host replay.
#include "gl355_internal.h"
#include "poseidon.cuh"

namespace gl355 {

__device__ __constant__ const uint32_t TAPE_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};

// stores of one lane -> loads of another lane of the same wave, through global memory.  The lanes share one CU and therefore
// one vector L1 (write-through): WORKGROUP scope is enough -- the stores only have to complete (s_waitcnt).  An agent-scope fence
// here also writes back and invalidates the XCD's L2, ~67 000 times per batch, and costs every other kernel on the chip its L2
// hits (measured: -5 % job throughput).
#define TAPE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
#define TAPE_WAVE_LDS_SYNC()                                    \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)

struct TapeDevArgs {
    const uint64_t* tape;          // [n_ops][5]
    const uint64_t* seg_start;     // [n_segs + 1] entry indices (seg_start[0] = n_seq)
    uint64_t n_seq;
    const uint64_t* inputs; uint64_t n_inputs;     // [units][n_inputs]
    uint64_t* rows; uint64_t n_words;              // [units][n_words]
    unsigned long long* status;                    // [units]: smallest failing entry, ~0 = none
};

// one PoseidonGate row: lanes 0..11 hold the state; wires as gl355_poseidon_gate_witness (host_transcript.cpp)
GL_DEV void tape_poseidon_row(uint64_t* row, int li, uint64_t* ring) {
    const bool active = li < 12;
    const int me = active ? li : 0;
    const uint64_t in = active ? row[li] : 0;
    const uint64_t swap = row[24];
    uint64_t s = in;
    if (li < 8) {
        const uint64_t partner = row[li ^ 4];
        if (li < 4) {
            const uint64_t delta = swap ? gl_canon(gl_sub(partner, in)) : 0;       // swap * (rhs - lhs)
            row[25 + li] = delta;
            s = gl_add(in, delta);
        } else {
            const uint64_t delta = swap ? gl_canon(gl_sub(in, partner)) : 0;
            s = gl_sub(in, delta);
        }
    }
    s = gl_add_canonical(s, PSD_ALL_RC[me]);
#pragma unroll 1
    for (int r = 0; r < 30; r++) {
        const uint64_t rc_next = PSD_ALL_RC[12 * (r + 1) + me];
        const bool full = r < 4 || r >= 26;
        // the S-box input (state + round constants) is a wire of the gate except in the very first round
        if (active) {
            if (r >= 1 && r < 4) row[29 + 12 * (r - 1) + li] = gl_canon(s);
            else if (r >= 26) row[87 + 12 * (r - 26) + li] = gl_canon(s);
            else if (r >= 4 && r < 26 && li == 0) row[65 + (r - 4)] = gl_canon(s);
        }
        if (full || li == 0) s = psd_sbox(s);
        if (active) { ring[me] = s; ring[me + 12] = s; }
        TAPE_WAVE_LDS_SYNC();
        uint64_t al = (uint32_t)rc_next, ah = rc_next >> 32;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const uint64_t x = ring[me + j];
            const uint32_t c = TAPE_CIRC[j] + ((me == 0 && j == 0) ? 8u : 0u);
            al += (uint64_t)(uint32_t)x * c;
            ah += (uint64_t)(uint32_t)(x >> 32) * c;
        }
        TAPE_WAVE_LDS_SYNC();
        s = psd_recombine(al, ah);
    }
    if (active) row[12 + li] = gl_canon(s);
}

__global__ void __launch_bounds__(64) tape_replay_kernel(TapeDevArgs a, uint32_t phase) {
    __shared__ uint64_t ring[24];
    const int lane = threadIdx.x;
    const uint32_t u = blockIdx.y;
    const uint64_t begin = phase == 0 ? 0 : a.seg_start[blockIdx.x];
    const uint64_t end = phase == 0 ? a.n_seq : a.seg_start[blockIdx.x + 1];
    uint64_t* W = a.rows + (uint64_t)u * a.n_words;
    const uint64_t* in = a.inputs + (uint64_t)u * a.n_inputs;
    for (uint64_t t = begin; t < end; t++) {
        const uint64_t* e = a.tape + 5 * t;
        const uint64_t op = e[0], x = e[1], b = e[2], c = e[3], d = e[4];
        int fail = 0;
        if (op == GL355_TAPE_POSEIDON) {
            TAPE_FENCE();
            if (W[x + 24] > 1) fail = 1;
            else tape_poseidon_row(W + x, lane & 15, ring);      // lanes 16..63 mirror lanes 0..15 of their group: same values, same stores
            TAPE_FENCE();
        } else if (lane == 0) {
            switch (op) {
            case GL355_TAPE_CONST: W[x] = gl_canon(b); break;
            case GL355_TAPE_INPUT: W[x] = gl_canon(in[b]); break;
            case GL355_TAPE_COPY: W[x] = W[b]; break;
            case GL355_TAPE_ASSERT_EQ: if (W[x] != W[b]) fail = 1; break;
            case GL355_TAPE_ARITH:
                W[x + 3] = gl_canon(gl_add(gl_mul(gl_mul(W[x], W[x + 1]), b), gl_mul(W[x + 2], c)));
                break;
            case GL355_TAPE_ARITH_EXT: {
                const gl2 pr = gl2_mul(gl2_make(W[x], W[x + 1]), gl2_make(W[x + 2], W[x + 3]));
                const gl2 r = gl2_canon(gl2_add(gl2_mul_base(pr, b), gl2_mul_base(gl2_make(W[x + 4], W[x + 5]), c)));
                W[x + 6] = r.c0; W[x + 7] = r.c1;
                break;
            }
            case GL355_TAPE_MDS_EXT:
                for (int r = 0; r < 12; r++)
                    for (int k = 0; k < 2; k++) {
                        uint64_t acc = 0;
                        for (int i = 0; i < 12; i++) acc = gl_add(acc, gl_mul_small(W[x + 2 * ((i + r) % 12) + k], TAPE_CIRC[i]));
                        if (r == 0) acc = gl_add(acc, gl_mul_small(W[x + k], 8));
                        W[x + 2 * (12 + r) + k] = gl_canon(acc);
                    }
                break;
            case GL355_TAPE_BASE_SUM: {
                const uint64_t v = W[x];
                if (v >> b) { fail = 1; break; }
                for (uint64_t i = 0; i < b; i++) W[x + 1 + i] = (v >> i) & 1;
                break;
            }
            case GL355_TAPE_RANDOM_ACCESS: {
                const uint64_t idx = W[x + 18 * b];
                if (idx >= 16) { fail = 1; break; }
                W[x + 18 * b + 1] = W[x + 18 * b + 2 + idx];
                for (int k = 0; k < 4; k++) W[x + 74 + 4 * b + k] = (idx >> k) & 1;
                break;
            }
            case GL355_TAPE_REDUCING: {
                const uint64_t n = b, ext = c;
                const uint64_t start_accs = 6 + (ext ? 2 * n : n);
                const gl2 alpha = gl2_make(W[x + 2], W[x + 3]);
                gl2 acc = gl2_make(W[x + 4], W[x + 5]);
                for (uint64_t i = 0; i < n; i++) {
                    const gl2 cf = ext ? gl2_make(W[x + 6 + 2 * i], W[x + 7 + 2 * i]) : gl2_make(W[x + 6 + i], 0);
                    acc = gl2_canon(gl2_add(gl2_mul(acc, alpha), cf));
                    const uint64_t o = i == n - 1 ? 0 : start_accs + 2 * i;
                    W[x + o] = acc.c0; W[x + o + 1] = acc.c1;
                }
                break;
            }
            case GL355_TAPE_LO32: W[x] = W[b] & 0xFFFFFFFFull; break;
            case GL355_TAPE_HI32: W[x] = W[b] >> 32; break;
            case GL355_TAPE_EXT_INV: {
                if (W[c] == 0 && W[d] == 0) { fail = 1; break; }
                const gl2 r = gl2_canon(gl2_inv(gl2_make(W[c], W[d])));
                W[x] = r.c0; W[b] = r.c1;
                break;
            }
            default: fail = 1; break;
            }
        }
        fail = __shfl(fail, 0);      // lane 0 decides (a POSEIDON failure is seen by every lane alike)
        if (fail) {
            if (lane == 0) atomicMin(a.status + u, (unsigned long long)t);
            break;
        }
    }
}

// public inputs of every unit: rows[pi_pos[i]]
__global__ void tape_gather_kernel(const uint64_t* rows, uint64_t n_words, const uint64_t* pi_pos, uint32_t n_pi, uint64_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, u = blockIdx.y;
    if (i < n_pi) out[(uint64_t)u * n_pi + i] = rows[(uint64_t)u * n_words + pi_pos[i]];
}

// enqueue the replay of n_units units on `stream`: d_inputs [units][n_inputs] -> d_rows [units][n_words] (zeroed first),
// d_status [units] (smallest failing entry or ~0), d_pis [units][n_pi]
int32_t tape_replay_dev(hipStream_t stream, const uint64_t* d_tape, const uint64_t* d_seg_start, uint64_t n_seq, uint32_t n_segs, uint32_t n_units,
                        const uint64_t* d_inputs, uint64_t n_inputs, uint64_t* d_rows, uint64_t n_words, const uint64_t* d_pi_pos, uint32_t n_pi,
                        uint64_t* d_status, uint64_t* d_pis) {
    if (hipMemsetAsync(d_rows, 0, (size_t)n_units * n_words * 8, stream) != hipSuccess) return GL355_E_HIP;
    if (hipMemsetAsync(d_status, 0xFF, (size_t)n_units * 8, stream) != hipSuccess) return GL355_E_HIP;
    TapeDevArgs a;
    a.tape = d_tape; a.seg_start = d_seg_start; a.n_seq = n_seq; a.inputs = d_inputs; a.n_inputs = n_inputs;
    a.rows = d_rows; a.n_words = n_words; a.status = reinterpret_cast<unsigned long long*>(d_status);
    if (n_seq) hipLaunchKernelGGL(tape_replay_kernel, dim3(1, n_units), dim3(64), 0, stream, a, 0u);
    if (n_segs) hipLaunchKernelGGL(tape_replay_kernel, dim3(n_segs, n_units), dim3(64), 0, stream, a, 1u);
    if (n_pi) hipLaunchKernelGGL(tape_gather_kernel, dim3((n_pi + 63) / 64, n_units), dim3(64), 0, stream, d_rows, n_words, d_pi_pos, n_pi, d_pis);
    return hipGetLastError() == hipSuccess ? GL355_OK : GL355_E_HIP;
}

}  // namespace gl355
