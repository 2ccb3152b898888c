// Host-side (CPU, sequential) pieces of the prover that SURVEY.md 8(a) keeps on the host: the
// Fiat-Shamir Challenger (a15) and the PoseidonGate witness row.  They are tiny and strictly
// sequential (about 50 permutations per proof), so they run on the calling thread; the data-parallel
// Poseidon work (leaf hashing, Merkle levels, PoW grinding) stays in merkle.hip.
//
// Replaces plonky2::iop::challenger::Challenger (transcript order pinned by
// src/plonky2_verifier/chip/plonk/plonk_verifier_chip.rs:55-154, sponge mechanics by
// chip/hasher_chip.rs:48-89) and plonky2::gates::poseidon::PoseidonGate's generator (wire layout
// chip/plonk/gates/poseidon.rs:329-380, round structure :634-686).
#include "gl355_internal.h"

#define PSD_TABLE_QUAL static const
#include "poseidon_tables.h"

namespace gl355 {

static inline uint64_t h_sbox(uint64_t x) {
    uint64_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
}
// MDS layer on the 32-bit halves: 12 x 6-bit constants keep both partial sums below 2^42, so the inner loops are plain
// 64-bit multiply-adds over a doubled array (no modulo indexing, no 128-bit arithmetic) and vectorise
static void h_mds(uint64_t s[12]) {
    static const uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint32_t lo[24], hi[24];
    for (int i = 0; i < 12; i++) { lo[i] = lo[i + 12] = (uint32_t)s[i]; hi[i] = hi[i + 12] = (uint32_t)(s[i] >> 32); }
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
        for (int i = 0; i < 12; i++) { al += (uint64_t)lo[i + r] * CIRC[i]; ah += (uint64_t)hi[i + r] * CIRC[i]; }
        if (r == 0) { al += (uint64_t)lo[0] * 8; ah += (uint64_t)hi[0] * 8; }
        // value = al + ah * 2^32 (< 2^75): ah * 2^32 = (ah >> 32) * 2^64 + (ah << 32)
        const uint64_t mid = ah << 32;
        const unsigned __int128 w0 = (unsigned __int128)al + mid;
        const uint64_t r0 = (uint64_t)w0;
        const uint64_t top = (ah >> 32) + (uint64_t)(w0 >> 64);
        const uint64_t t = top * GL_EPS;                  // top < 2^11
        uint64_t r1 = r0 + t;
        if (__builtin_unpredictable(r1 < t)) r1 += GL_EPS;
        s[r] = r1;
    }
}
// fast-partial form; sbox_in (optional) receives the 22 partial-round S-box inputs, full_in the
// S-box inputs of the 8 full rounds (12 each) -- exactly the values PoseidonGate stores as wires
static void h_permute(uint64_t s[12], uint64_t* full_in /*[8][12]*/, uint64_t* part_in /*[22]*/) {
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) {
            s[i] = gl_add(s[i], PSD_FULL_RC[12 * r + i]);
            if (full_in) full_in[12 * r + i] = gl_canon(s[i]);
            s[i] = h_sbox(s[i]);
        }
        h_mds(s);
    }
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], PSD_PART_FIRST[i]);
    {
        uint64_t t[12];
        t[0] = s[0];
        for (int c = 1; c < 12; c++) {
            uint64_t acc = 0;
            for (int r = 1; r < 12; r++) acc = gl_add(acc, gl_mul(s[r], PSD_PART_INIT[(r - 1) * 11 + (c - 1)]));
            t[c] = acc;
        }
        memcpy(s, t, sizeof t);
    }
    for (int r = 0; r < 22; r++) {
        if (part_in) part_in[r] = gl_canon(s[0]);
        uint64_t s0 = gl_add(h_sbox(s[0]), PSD_PART_RC[r]);
        uint64_t d = gl_mul_small(s0, 25);
        for (int i = 1; i < 12; i++) {
            d = gl_add(d, gl_mul(s[i], PSD_PART_WHAT[r * 11 + (i - 1)]));
            s[i] = gl_add(s[i], gl_mul(s0, PSD_PART_VS[r * 11 + (i - 1)]));
        }
        s[0] = d;
    }
    for (int r = 4; r < 8; r++) {
        for (int i = 0; i < 12; i++) {
            s[i] = gl_add(s[i], PSD_FULL_RC[12 * r + i]);
            if (full_in) full_in[12 * r + i] = gl_canon(s[i]);
            s[i] = h_sbox(s[i]);
        }
        h_mds(s);
    }
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}

}  // namespace gl355

using namespace gl355;

extern "C" {

int32_t gl355_host_poseidon_permute(uint64_t state[12]) {
    if (!state) return GL355_E_INVALID_ARG;
    h_permute(state, nullptr, nullptr);
    return GL355_OK;
}

static void permute_h(int32_t hasher, uint64_t st[12]) {
    if (hasher == GL355_HASH_BN254_POSEIDON) host_bn254_permute(st);
    else h_permute(st, nullptr, nullptr);
}
int32_t gl355_host_permute_h(int32_t hasher, uint64_t state[12]) {
    if (!state || (hasher != GL355_HASH_POSEIDON && hasher != GL355_HASH_BN254_POSEIDON)) return GL355_E_INVALID_ARG;
    for (int i = 0; i < 12; i++) state[i] = gl_canon(state[i]);
    permute_h(hasher, state);
    return GL355_OK;
}
int32_t gl355_host_hash_no_pad_h(int32_t hasher, const uint64_t* in, uint64_t len, uint64_t out[4]) {
    if ((!in && len) || !out || (hasher != GL355_HASH_POSEIDON && hasher != GL355_HASH_BN254_POSEIDON)) return GL355_E_INVALID_ARG;
    uint64_t st[12] = {0};
    for (uint64_t off = 0; off < len; off += 8) {
        const uint64_t m = len - off < 8 ? len - off : 8;
        for (uint64_t i = 0; i < m; i++) st[i] = gl_canon(in[off + i]);
        permute_h(hasher, st);
    }
    memcpy(out, st, 32);
    return GL355_OK;
}
int32_t gl355_host_hash_no_pad(const uint64_t* in, uint64_t len, uint64_t out[4]) {
    return gl355_host_hash_no_pad_h(GL355_HASH_POSEIDON, in, len, out);
}

int32_t gl355_challenger_init(gl355_challenger* c) {
    if (!c) return GL355_E_INVALID_ARG;
    memset(c, 0, sizeof *c);
    return GL355_OK;
}
int32_t gl355_challenger_init_h(gl355_challenger* c, int32_t hasher) {
    if (!c || (hasher != GL355_HASH_POSEIDON && hasher != GL355_HASH_BN254_POSEIDON)) return GL355_E_INVALID_ARG;
    memset(c, 0, sizeof *c);
    c->hasher = hasher;
    return GL355_OK;
}
static void duplex(gl355_challenger* c) {
    for (uint32_t i = 0; i < c->in_len; i++) c->state[i] = c->in_buf[i];
    c->in_len = 0;
    permute_h(c->hasher, c->state);
    memcpy(c->out_buf, c->state, 64);
    c->out_len = 8;
}
int32_t gl355_challenger_observe(gl355_challenger* c, const uint64_t* elems, uint64_t n) {
    if (!c || (!elems && n)) return GL355_E_INVALID_ARG;
    for (uint64_t i = 0; i < n; i++) {
        c->out_len = 0;  // any new input invalidates buffered outputs
        c->in_buf[c->in_len++] = gl_canon(elems[i]);
        if (c->in_len == 8) duplex(c);
    }
    return GL355_OK;
}
int32_t gl355_challenger_squeeze(gl355_challenger* c, uint64_t* out, uint64_t n) {
    if (!c || !out) return GL355_E_INVALID_ARG;
    for (uint64_t i = 0; i < n; i++) {
        if (c->in_len > 0 || c->out_len == 0) duplex(c);
        out[i] = c->out_buf[--c->out_len];  // pops from the END of the rate part
    }
    return GL355_OK;
}
// state and witness position for the PoW search: the pending inputs are written over the sponge state,
// the candidate goes to slot in_len, then one permutation; the response is state[7]
int32_t gl355_challenger_pow_state(const gl355_challenger* c, uint64_t state[12], uint32_t* pos) {
    if (!c || !state || !pos) return GL355_E_INVALID_ARG;
    if (c->in_len >= 8) return GL355_E_INVALID_ARG;
    memcpy(state, c->state, 96);
    for (uint32_t i = 0; i < c->in_len; i++) state[i] = c->in_buf[i];
    *pos = c->in_len;
    return GL355_OK;
}

// All 135 wires of one PoseidonGate row from its 12 inputs and the swap bit.
int32_t gl355_poseidon_gate_witness(const uint64_t inputs[12], uint64_t swap, uint64_t wires[135]) {
    if (!inputs || !wires || swap > 1) return GL355_E_INVALID_ARG;
    for (int i = 0; i < 135; i++) wires[i] = 0;
    for (int i = 0; i < 12; i++) wires[i] = gl_canon(inputs[i]);
    wires[24] = swap;
    uint64_t s[12];
    for (int i = 0; i < 12; i++) s[i] = wires[i];
    for (int i = 0; i < 4; i++) {
        const uint64_t delta = swap ? gl_canon(gl_sub(wires[4 + i], wires[i])) : 0;  // swap * (rhs - lhs)
        wires[25 + i] = delta;
        s[i] = gl_add(wires[i], delta);
        s[4 + i] = gl_sub(wires[4 + i], delta);
    }
    uint64_t full_in[96], part_in[22];
    h_permute(s, full_in, part_in);
    // first full round's S-box inputs are not wires; rounds 1..3 -> wires 29..64
    for (int r = 1; r < 4; r++)
        for (int i = 0; i < 12; i++) wires[29 + 12 * (r - 1) + i] = full_in[12 * r + i];
    for (int r = 0; r < 22; r++) wires[65 + r] = part_in[r];
    for (int r = 0; r < 4; r++)
        for (int i = 0; i < 12; i++) wires[87 + 12 * r + i] = full_in[12 * (4 + r) + i];
    for (int i = 0; i < 12; i++) wires[12 + i] = s[i];
    return GL355_OK;
}

}  // extern "C"

// Witness rows of the Semaphore circuit (src/plonky2_semaphore/circuit.rs:67-99 fill_semaphore_targets +
// the PoseidonGate / BaseSumGate generators), in the row order of stark-verifier_amd/semaphore.py:
//   0 PublicInputGate | 1,2 public-input hash permutations | 3 BaseSum{h} (index bits) | 4 leaf hash |
//   5..5+h-1 Merkle levels | 5+h nullifier hash | 6+h ConstantGate (zeros)
// rows: (h + 7) x 135 values; public_inputs: root | nullifier | topic.
extern "C" int32_t gl355_semaphore_witness(const uint64_t private_key[4], const uint64_t topic[4], uint64_t index,
                                           const uint64_t* siblings, uint32_t height, uint64_t* rows, uint64_t public_inputs[12]) {
    if (!private_key || !topic || (!siblings && height) || !rows || !public_inputs || height > 63) return GL355_E_INVALID_ARG;
    const uint32_t n_rows = height + 7;
    memset(rows, 0, (size_t)n_rows * 135 * 8);
    uint64_t* r_pi = rows, *r_h1 = rows + 135, *r_h2 = rows + 2 * 135, *r_bits = rows + 3 * 135, *r_leaf = rows + 4 * 135;
    uint64_t* r_null = rows + (size_t)(5 + height) * 135;
    uint64_t in[12];
    for (int i = 0; i < 4; i++) { in[i] = private_key[i]; in[4 + i] = 0; in[8 + i] = 0; }
    gl355_poseidon_gate_witness(in, 0, r_leaf);
    uint64_t state[4];
    memcpy(state, r_leaf + 12, 32);
    r_bits[0] = index;
    for (uint32_t l = 0; l < height; l++) {
        const uint64_t bit = (index >> l) & 1;
        r_bits[1 + l] = bit;
        for (int i = 0; i < 4; i++) { in[i] = state[i]; in[4 + i] = siblings[4 * l + i]; in[8 + i] = 0; }
        uint64_t* row = rows + (size_t)(5 + l) * 135;
        gl355_poseidon_gate_witness(in, bit, row);
        memcpy(state, row + 12, 32);
    }
    for (int i = 0; i < 4; i++) { in[i] = private_key[i]; in[4 + i] = topic[i]; in[8 + i] = 0; }
    gl355_poseidon_gate_witness(in, 0, r_null);
    for (int i = 0; i < 4; i++) {
        public_inputs[i] = state[i];
        public_inputs[4 + i] = r_null[12 + i];
        public_inputs[8 + i] = gl_canon(topic[i]);
    }
    for (int i = 0; i < 8; i++) in[i] = public_inputs[i];
    for (int i = 8; i < 12; i++) in[i] = 0;
    gl355_poseidon_gate_witness(in, 0, r_h1);
    for (int i = 0; i < 4; i++) in[i] = public_inputs[8 + i];
    for (int i = 4; i < 12; i++) in[i] = r_h1[12 + i];
    gl355_poseidon_gate_witness(in, 0, r_h2);
    for (int i = 0; i < 4; i++) r_pi[i] = r_h2[12 + i];
    return GL355_OK;
}
