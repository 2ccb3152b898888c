// Product-side verifier: `CircuitData::verify(proof)` (src/plonky2_semaphore/access_set.rs:170-175, signal verification) for the flat
// proofs this library makes -- host code, no device involved (a verification is ~300 permutations and a few thousand field products).
// The checks are the ones the reference's own verifier performs in-circuit (src/plonky2_verifier/chip):
//   get_challenges                 chip/plonk/plonk_verifier_chip.rs:55-154
//   verify_proof_with_challenges   chip/plonk/plonk_verifier_chip.rs:156-242 (vanishing identity at zeta)
//   eval_vanishing_poly            chip/plonk/vanishing_poly.rs:18-218, gate filters chip/plonk/gates/mod.rs:87-132, gates/*.rs
//   FRI                            chip/fri_chip.rs:58-376, Merkle paths chip/merkle_proof_chip.rs:39-87
// Everything is evaluated in the quadratic extension (openings are extension elements); the gates over the extension ALGEBRA
// (chip/goldilocks_extension_algebra_chip.rs:112-146) work on pairs of extension elements.  The Poseidon gate is evaluated in the
// dense form (state + constants, S-box, full MDS): the same 123 polynomials in the wires as the reference's fast-partial form.
// tests/plonk_verifier.py is the independent Python restatement the test-suite compares this against (accept / reject agree).
#include "gl355_internal.h"

#define PSD_TABLE_QUAL static const
#include "poseidon_tables.h"

#include <string>
#include <vector>

using namespace gl355;

namespace {

thread_local std::string g_verify_error;
const uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};

typedef gl2 E;
inline E e_base(uint64_t v) { return gl2_make(gl_canon(v), 0); }
inline E e_c(E a) { return gl2_canon(a); }
inline bool e_eq(E a, E b) { a = e_c(a); b = e_c(b); return a.c0 == b.c0 && a.c1 == b.c1; }
inline E e_small(E a, uint32_t k) { return gl2_make(gl_mul_small(a.c0, k), gl_mul_small(a.c1, k)); }
E e_pow(E a, uint64_t e) { return gl2_pow(a, e); }
// sum_i a^i t_i
E reduce_with_powers(const E* t, size_t n, E a) {
    E acc = gl2_make(0, 0);
    for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, a), t[i]);
    return acc;
}
E e_sbox(E x) {
    const E x2 = gl2_mul(x, x), x4 = gl2_mul(x2, x2);
    return gl2_mul(gl2_mul(x, x2), x4);
}
// full MDS layer + the next round's constants (row 30 of PSD_ALL_RC is zero)
void e_mds(E s[12], const uint64_t* rc) {
    E o[12];
    for (int r = 0; r < 12; r++) {
        E acc = e_base(rc[r]);
        for (int i = 0; i < 12; i++) acc = gl2_add(acc, e_small(s[(i + r) % 12], CIRC[i]));
        if (r == 0) acc = gl2_add(acc, e_small(s[0], 8));
        o[r] = acc;
    }
    for (int r = 0; r < 12; r++) s[r] = o[r];
}

struct Alg { E a0, a1; };      // element of the extension algebra: a0 + a1 X, X^2 = 7, components in the extension field
inline Alg alg_mul(Alg a, Alg b) {
    return Alg{gl2_add(gl2_mul(a.a0, b.a0), e_small(gl2_mul(a.a1, b.a1), 7)), gl2_add(gl2_mul(a.a0, b.a1), gl2_mul(a.a1, b.a0))};
}
inline Alg alg_add(Alg a, Alg b) { return Alg{gl2_add(a.a0, b.a0), gl2_add(a.a1, b.a1)}; }
inline Alg alg_sub(Alg a, Alg b) { return Alg{gl2_sub(a.a0, b.a0), gl2_sub(a.a1, b.a1)}; }
inline Alg alg_scal(E c, Alg a) { return Alg{gl2_mul(c, a.a0), gl2_mul(c, a.a1)}; }

// constraints of one gate at the opened point: consts = the gate constants (after the selectors), w = the wires
// (num_constants = gate constants present in the opening; every read of consts[] / w[] is bounded against the verifier data here,
// which may come from an untrusted artifact)
bool eval_gate(const gl355_gate& gt, const E* consts, const E* w, const uint64_t pi_hash[4], uint32_t num_wires, uint32_t num_constants,
               std::vector<E>& c) {
    c.clear();
    auto alg = [&](uint32_t j) { return Alg{w[j], w[j + 1]}; };
    auto push_alg = [&](Alg v) { c.push_back(v.a0); c.push_back(v.a1); };
    const uint32_t p = gt.param;
    // gt.param is untrusted: every bound below is computed in 64 bits (4 * p, 8 * p, 1 + p and 6 + 2 * p wrap in 32), and no gate whose
    // parameter is a count can have more of them than there are wires (RANDOM_ACCESS packs three byte-sized fields instead)
    const uint64_t P = p, NW = num_wires;
    if (gt.type != GL355_GATE_RANDOM_ACCESS && P > NW) return false;
    switch (gt.type) {
    case GL355_GATE_NOOP: return true;
    case GL355_GATE_CONSTANT:
        if (p > num_constants || p > num_wires) return false;
        for (uint32_t i = 0; i < p; i++) c.push_back(gl2_sub(consts[i], w[i]));
        return true;
    case GL355_GATE_PUBLIC_INPUT:
        if (num_wires < 4) return false;
        for (int i = 0; i < 4; i++) c.push_back(gl2_sub(w[i], e_base(pi_hash[i])));
        return true;
    case GL355_GATE_BASE_SUM: {
        if (1 + P > NW) return false;
        c.push_back(gl2_sub(reduce_with_powers(w + 1, p, e_base(2)), w[0]));
        for (uint32_t i = 0; i < p; i++) c.push_back(gl2_sub(gl2_mul(w[1 + i], w[1 + i]), w[1 + i]));
        return true;
    }
    case GL355_GATE_ARITHMETIC:
        if (4 * P > NW || num_constants < 2) return false;
        for (uint32_t i = 0; i < p; i++)
            c.push_back(gl2_sub(w[4 * i + 3], gl2_add(gl2_mul(gl2_mul(w[4 * i], w[4 * i + 1]), consts[0]), gl2_mul(w[4 * i + 2], consts[1]))));
        return true;
    case GL355_GATE_POSEIDON: {
        if (num_wires < 135) return false;
        const E swap = w[24];
        c.push_back(gl2_sub(gl2_mul(swap, swap), swap));
        E s[12];
        for (int i = 0; i < 4; i++) {
            const E lhs = w[i], rhs = w[i + 4], delta = w[25 + i];
            c.push_back(gl2_sub(gl2_mul(swap, gl2_sub(rhs, lhs)), delta));
            s[i] = gl2_add(lhs, delta);
            s[i + 4] = gl2_sub(rhs, delta);
        }
        for (int i = 8; i < 12; i++) s[i] = w[i];
        for (int i = 0; i < 12; i++) s[i] = gl2_add(s[i], e_base(PSD_ALL_RC[i]));
        for (int r = 0; r < 4; r++) {
            for (int i = 0; i < 12; i++) {
                if (r != 0) { const E sin = w[29 + 12 * (r - 1) + i]; c.push_back(gl2_sub(s[i], sin)); s[i] = sin; }
                s[i] = e_sbox(s[i]);
            }
            e_mds(s, &PSD_ALL_RC[12 * (r + 1)]);
        }
        for (int r = 0; r < 22; r++) {
            const E sin = w[65 + r];
            c.push_back(gl2_sub(s[0], sin));
            s[0] = e_sbox(sin);
            e_mds(s, &PSD_ALL_RC[12 * (5 + r)]);
        }
        for (int r = 0; r < 4; r++) {
            for (int i = 0; i < 12; i++) { const E sin = w[87 + 12 * r + i]; c.push_back(gl2_sub(s[i], sin)); s[i] = e_sbox(sin); }
            e_mds(s, &PSD_ALL_RC[12 * (27 + r)]);
        }
        for (int i = 0; i < 12; i++) c.push_back(gl2_sub(s[i], w[12 + i]));
        return true;
    }
    case GL355_GATE_ARITHMETIC_EXT:
        if (8 * P > NW || num_constants < 2) return false;
        for (uint32_t i = 0; i < p; i++)
            push_alg(alg_sub(alg(8 * i + 6), alg_add(alg_scal(consts[0], alg_mul(alg(8 * i), alg(8 * i + 2))), alg_scal(consts[1], alg(8 * i + 4)))));
        return true;
    case GL355_GATE_MUL_EXT:
        if (6 * P > NW || num_constants < 1) return false;
        for (uint32_t i = 0; i < p; i++) push_alg(alg_sub(alg(6 * i + 4), alg_scal(consts[0], alg_mul(alg(6 * i), alg(6 * i + 2)))));
        return true;
    case GL355_GATE_POSEIDON_MDS:
        if (num_wires < 48) return false;
        for (uint32_t r = 0; r < 12; r++) {
            Alg acc{gl2_make(0, 0), gl2_make(0, 0)};
            for (uint32_t i = 0; i < 12; i++) acc = alg_add(acc, alg_scal(e_base(CIRC[i]), alg(2 * ((i + r) % 12))));
            if (r == 0) acc = alg_add(acc, alg_scal(e_base(8), alg(0)));
            push_alg(alg_sub(alg(2 * (12 + r)), acc));
        }
        return true;
    case GL355_GATE_RANDOM_ACCESS: {
        const uint32_t bits = p & 0xFF, copies = (p >> 8) & 0xFF, extra = (p >> 16) & 0xFF;
        const uint32_t vec = 1u << bits, routed = (2 + vec) * copies + extra;
        if (bits > 8 || (uint64_t)routed + (uint64_t)copies * bits > NW || extra > num_constants) return false;
        for (uint32_t cp = 0; cp < copies; cp++) {
            const uint32_t b0 = (2 + vec) * cp;
            const E* bl = w + routed + cp * bits;
            for (uint32_t i = 0; i < bits; i++) c.push_back(gl2_sub(gl2_mul(bl[i], bl[i]), bl[i]));
            c.push_back(gl2_sub(reduce_with_powers(bl, bits, e_base(2)), w[b0]));
            std::vector<E> items(w + b0 + 2, w + b0 + 2 + vec);
            for (uint32_t l = 0; l < bits; l++) {
                for (size_t k = 0; k < items.size() / 2; k++) items[k] = gl2_add(gl2_mul(bl[l], gl2_sub(items[2 * k + 1], items[2 * k])), items[2 * k]);
                items.resize(items.size() / 2);
            }
            c.push_back(gl2_sub(items[0], w[b0 + 1]));
        }
        for (uint32_t i = 0; i < extra; i++) c.push_back(gl2_sub(consts[i], w[(2 + vec) * copies + i]));
        return true;
    }
    case GL355_GATE_REDUCING:
    case GL355_GATE_REDUCING_EXT: {
        const bool isext = gt.type == GL355_GATE_REDUCING_EXT;
        if (p == 0 || 6 + (isext ? 2 * P : P) + 2 * (P - 1) > NW) return false;
        const uint32_t start_accs = 6 + (isext ? 2 * p : p);
        const Alg alpha = alg(2);
        Alg acc = alg(4);
        for (uint32_t i = 0; i < p; i++) {
            const Alg coeff = isext ? alg(6 + 2 * i) : Alg{w[6 + i], gl2_make(0, 0)};
            const Alg acc_i = (i == p - 1) ? alg(0) : alg(start_accs + 2 * i);
            push_alg(alg_sub(alg_add(alg_mul(acc, alpha), coeff), acc_i));
            acc = acc_i;
        }
        return true;
    }
    default: return false;
    }
}

struct Hash4 { uint64_t v[4]; };
Hash4 two_to_one(int32_t hasher, const uint64_t* l, const uint64_t* r) {
    uint64_t st[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    gl355_host_permute_h(hasher, st);
    Hash4 h;
    memcpy(h.v, st, 32);
    return h;
}
// MerkleTree::verify with a cap (merkle_proof_chip.rs:39-87): bit k of the index = 1 -> the current node is the right child
bool merkle_verify(int32_t hasher, const uint64_t* leaf, uint32_t leaf_len, uint64_t index, const uint64_t* sibs, uint32_t layers,
                   const uint64_t* cap, uint32_t cap_height) {
    Hash4 cur;
    if (leaf_len <= 4) { for (uint32_t i = 0; i < 4; i++) cur.v[i] = i < leaf_len ? gl_canon(leaf[i]) : 0; }
    else gl355_host_hash_no_pad_h(hasher, leaf, leaf_len, cur.v);
    for (uint32_t k = 0; k < layers; k++) {
        cur = ((index >> k) & 1) ? two_to_one(hasher, sibs + 4 * k, cur.v) : two_to_one(hasher, cur.v, sibs + 4 * k);
    }
    const uint64_t ci = index >> layers;
    if (ci >= (1ull << cap_height)) return false;
    for (int i = 0; i < 4; i++) if (gl_canon(cap[4 * ci + i]) != cur.v[i]) return false;
    return true;
}

int32_t fail(const char* why) { g_verify_error = why; return GL355_E_VERIFY; }

}  // namespace

extern "C" {

const char* gl355_verify_last_error(void) { return g_verify_error.c_str(); }

static int32_t verify_impl(const gl355_verifier_data* vd, const uint64_t* proof, uint64_t proof_words, const uint64_t* public_inputs,
                           uint32_t n_public_inputs);
int32_t gl355_verify(const gl355_verifier_data* vd, const uint64_t* proof, uint64_t proof_words, const uint64_t* public_inputs,
                     uint32_t n_public_inputs) {
    try {
        return verify_impl(vd, proof, proof_words, public_inputs, n_public_inputs);
    } catch (...) {                      // allocation failure: nothing is thrown across the C boundary
        g_verify_error = "verify: out of host memory";
        return GL355_E_OOM;
    }
}
static int32_t verify_impl(const gl355_verifier_data* vd, const uint64_t* proof, uint64_t proof_words, const uint64_t* public_inputs,
                           uint32_t n_public_inputs) {
    g_verify_error.clear();
    if (!vd || !vd->circuit || !vd->constants_sigmas_cap || !vd->k_is || !proof || (!public_inputs && n_public_inputs)) { g_verify_error = "verify: null argument"; return GL355_E_INVALID_ARG; }
    const gl355_circuit& c = *vd->circuit;
    const uint32_t nch = c.num_challenges, qdf = c.max_degree, npp = c.num_partial_products, routed = c.num_routed_wires;
    const uint32_t cap_h = vd->cap_height, L = vd->n_fri_layers, nq = vd->num_queries, lde_bits = c.degree_bits + c.rate_bits;
    const int32_t hasher = vd->hasher;
    if (nch == 0 || nch > 4 || L > 32 || lde_bits > 40 || c.num_gates > GL355_MAX_GATES || qdf == 0 || c.num_selectors == 0 ||
        L > lde_bits || cap_h > lde_bits - L ||        // every FRI layer tree (2^(lde_bits - l) leaves... the last has lde_bits - L path bits) must reach its cap
        (hasher != GL355_HASH_POSEIDON && hasher != GL355_HASH_BN254_POSEIDON)) { g_verify_error = "verify: unsupported verifier data"; return GL355_E_INVALID_ARG; }
    const bool zk = vd->zero_knowledge != 0;
    const uint64_t n_cap = 1ull << cap_h;
    const uint32_t widths[4] = {c.num_selectors + c.num_constants + routed, c.num_wires, nch * (1 + npp), nch * qdf};
    // ---- shape of the flat proof (include/gl355.h) ----------------------------------------------------------------------------
    gl355_prover_data pd;
    memset(&pd, 0, sizeof pd);
    pd.circuit = vd->circuit; pd.cap_height = cap_h; pd.num_queries = nq; pd.n_fri_layers = L; pd.zero_knowledge = vd->zero_knowledge;
    const uint64_t need = gl355_proof_words(&pd);
    if (proof_words != need || proof[0] != need) return fail("proof length does not match the circuit");
    if (proof[1] != c.degree_bits || proof[2] != L || proof[3] != nq || proof[4] != n_public_inputs || proof[5] != (uint64_t)zk || proof[6] != cap_h || proof[7] != nch)
        return fail("proof header does not match the verifier data");
    const uint64_t* p = proof + 8;
    const uint64_t* wires_cap = p; p += n_cap * 4;
    const uint64_t* zs_cap = p; p += n_cap * 4;
    const uint64_t* q_cap = p; p += n_cap * 4;
    uint32_t n_open = 0;
    for (int o = 0; o < 4; o++) n_open += widths[o];
    const uint64_t* open_words = p; p += 2ull * (n_open + nch);
    const uint64_t* fri_caps = p; p += (uint64_t)L * n_cap * 4;
    const uint64_t final_len = (1ull << c.degree_bits) >> L;
    const uint64_t* final_poly = p; p += 2 * final_len;
    const uint64_t pow_witness = *p++;
    const uint64_t* queries = p;
    // public inputs are field elements too: v and v + p hash alike (gl355_host_hash_no_pad canonicalises), so a consumer comparing
    // or de-duplicating them as raw u64 (nullifiers, topics) could be aliased -- only the canonical encoding is accepted
    for (uint32_t i = 0; i < n_public_inputs; i++)
        if (public_inputs[i] >= GL_P) return fail("non-canonical public input");
    // every field element of the proof must be canonical (plonky2 deserialisation rejects anything else)
    for (const uint64_t* q = proof + 8; q < proof + need; q++)
        if (*q >= GL_P) return fail("non-canonical field element in the proof");
    std::vector<E> open(n_open + nch);
    for (uint32_t i = 0; i < n_open + nch; i++) open[i] = gl2_make(open_words[2 * i], open_words[2 * i + 1]);
    const E* o_consts = open.data();                                  // selectors | gate constants
    const E* o_sigmas = o_consts + c.num_selectors + c.num_constants;
    const E* o_wires = o_sigmas + routed;
    const E* o_zs = o_wires + c.num_wires;
    const E* o_pp = o_zs + nch;
    const E* o_quot = o_pp + nch * npp;
    const E* o_zs_next = open.data() + n_open;
    // ---- get_challenges ----------------------------------------------------------------------------------------------------------
    uint64_t pi_hash[4];
    gl355_host_hash_no_pad(public_inputs, n_public_inputs, pi_hash);
    gl355_challenger ch;
    gl355_challenger_init_h(&ch, hasher);
    gl355_challenger_observe(&ch, vd->circuit_digest, 4);
    gl355_challenger_observe(&ch, pi_hash, 4);
    gl355_challenger_observe(&ch, wires_cap, n_cap * 4);
    uint64_t betas[4], gammas[4], alphas[4], z2[2];
    gl355_challenger_squeeze(&ch, betas, nch);
    gl355_challenger_squeeze(&ch, gammas, nch);
    gl355_challenger_observe(&ch, zs_cap, n_cap * 4);
    gl355_challenger_squeeze(&ch, alphas, nch);
    gl355_challenger_observe(&ch, q_cap, n_cap * 4);
    gl355_challenger_squeeze(&ch, z2, 2);
    const E zeta = gl2_make(gl_canon(z2[0]), gl_canon(z2[1]));
    gl355_challenger_observe(&ch, open_words, 2ull * (n_open + nch));
    gl355_challenger_squeeze(&ch, z2, 2);
    const E fri_alpha = gl2_make(gl_canon(z2[0]), gl_canon(z2[1]));
    std::vector<E> fri_betas(L);
    for (uint32_t l = 0; l < L; l++) {
        gl355_challenger_observe(&ch, fri_caps + (uint64_t)l * n_cap * 4, n_cap * 4);
        gl355_challenger_squeeze(&ch, z2, 2);
        fri_betas[l] = gl2_make(gl_canon(z2[0]), gl_canon(z2[1]));
    }
    gl355_challenger_observe(&ch, final_poly, 2 * final_len);
    gl355_challenger_observe(&ch, &pow_witness, 1);
    uint64_t pow_response;
    gl355_challenger_squeeze(&ch, &pow_response, 1);
    std::vector<uint64_t> q_idx(nq);
    gl355_challenger_squeeze(&ch, q_idx.data(), nq);
    // ---- vanishing identity at zeta (plonk_verifier_chip.rs:174-210) ------------------------------------------------------------------
    const uint64_t n = 1ull << c.degree_bits;
    const E zeta_pow_n = e_pow(zeta, n);
    const E one = gl2_make(1, 0);
    std::vector<E> terms;
    {
        // gate constraints with filters
        std::vector<E> allc, gc;
        for (uint32_t gi = 0; gi < c.num_gates; gi++) {
            const gl355_gate& gt = c.gates[gi];
            if (gt.type > GL355_GATE_TYPE_MAX || gt.selector_index >= c.num_selectors || gt.group_end > c.num_gates || gt.group_start > gt.group_end)
                { g_verify_error = "verify: bad gate table"; return GL355_E_INVALID_ARG; }
            if (!eval_gate(gt, o_consts + c.num_selectors, o_wires, pi_hash, c.num_wires, c.num_constants, gc)) { g_verify_error = "verify: gate does not fit the wire / constant count"; return GL355_E_INVALID_ARG; }
            const E sel = o_consts[gt.selector_index];
            E filt = one;
            for (uint32_t k = gt.group_start; k < gt.group_end; k++)
                if (k != gi) filt = gl2_mul(filt, gl2_sub(e_base(k), sel));
            if (c.num_selectors > 1) filt = gl2_mul(filt, gl2_sub(e_base(0xFFFFFFFFull), sel));
            if (gc.size() > allc.size()) allc.resize(gc.size(), gl2_make(0, 0));
            for (size_t k = 0; k < gc.size(); k++) allc[k] = gl2_add(allc[k], gl2_mul(filt, gc[k]));
        }
        // L0(x) = (x^n - 1) / (n (x - 1))
        const E nx = gl2_sub(gl2_mul_base(zeta, gl_canon(n % GL_P)), e_base(n % GL_P));
        if (e_eq(nx, gl2_make(0, 0))) return fail("zeta is the first root of unity");
        const E l0 = gl2_mul(gl2_sub(zeta_pow_n, one), gl2_inv(nx));
        for (uint32_t i = 0; i < nch; i++) terms.push_back(gl2_sub(gl2_mul(l0, o_zs[i]), l0));
        const uint32_t n_chunks = (routed + qdf - 1) / qdf;
        if (npp + 1 != n_chunks) { g_verify_error = "verify: inconsistent partial-product count"; return GL355_E_INVALID_ARG; }
        for (uint32_t i = 0; i < nch; i++) {
            const E beta = e_base(betas[i]), gamma = e_base(gammas[i]);
            for (uint32_t chk = 0; chk < n_chunks; chk++) {
                E np = one, dp = one;
                for (uint32_t j = chk * qdf; j < (chk + 1) * qdf && j < routed; j++) {
                    const E wg = gl2_add(o_wires[j], gamma);
                    np = gl2_mul(np, gl2_add(gl2_mul(beta, gl2_mul_base(zeta, gl_canon(vd->k_is[j]))), wg));
                    dp = gl2_mul(dp, gl2_add(gl2_mul(beta, o_sigmas[j]), wg));
                }
                const E prev = chk == 0 ? o_zs[i] : o_pp[i * npp + chk - 1];
                const E next = chk + 1 < n_chunks ? o_pp[i * npp + chk] : o_zs_next[i];
                terms.push_back(gl2_sub(gl2_mul(prev, np), gl2_mul(next, dp)));
            }
        }
        terms.insert(terms.end(), allc.begin(), allc.end());
    }
    const E z_h = gl2_sub(zeta_pow_n, one);
    for (uint32_t i = 0; i < nch; i++) {
        const E van = reduce_with_powers(terms.data(), terms.size(), e_base(alphas[i]));
        const E rhs = gl2_mul(z_h, reduce_with_powers(o_quot + i * qdf, qdf, zeta_pow_n));
        if (!e_eq(van, rhs)) return fail("quotient identity fails at zeta");
    }
    // ---- FRI (fri_chip.rs:58-376) -------------------------------------------------------------------------------------------------------
    if (vd->pow_bits && vd->pow_bits < 64 && (pow_response >> (64 - vd->pow_bits)) != 0) return fail("proof of work");
    const uint64_t g = gl_root_of_unity(c.degree_bits);
    const E zeta_next = gl2_make(gl_canon(gl_mul(zeta.c0, g)), gl_canon(gl_mul(zeta.c1, g)));
    const E red_zeta = reduce_with_powers(open.data(), n_open, fri_alpha);
    const E red_next = reduce_with_powers(o_zs_next, nch, fri_alpha);
    const E alpha_pow_next = e_pow(fri_alpha, nch);
    const uint64_t N = 1ull << lde_bits;
    const uint64_t omega = gl_root_of_unity(lde_bits);
    const uint32_t depth0 = lde_bits - cap_h;
    const uint64_t* caps[4] = {vd->constants_sigmas_cap, wires_cap, zs_cap, q_cap};
    std::vector<E> evals;
    const uint64_t* q = queries;
    std::vector<E> fpoly(final_len);
    for (uint64_t i = 0; i < final_len; i++) fpoly[i] = gl2_make(final_poly[2 * i], final_poly[2 * i + 1]);
    for (uint32_t qi = 0; qi < nq; qi++) {
        const uint64_t x_index = q_idx[qi] & (N - 1);
        if (*q++ != x_index) return fail("query index does not follow from the transcript");
        const uint64_t* leaves[4];
        for (int o = 0; o < 4; o++) {
            const uint32_t ll = widths[o] + ((zk && o > 0) ? GL355_SALT_SIZE : 0);
            leaves[o] = q;
            for (uint32_t i = 0; i < ll; i++) if (q[i] >= GL_P) return fail("non-canonical field element in an opened leaf");
            q += ll;
            if (!merkle_verify(hasher, leaves[o], ll, x_index, q, depth0, caps[o], cap_h)) return fail("initial tree opening does not verify");
            q += (uint64_t)depth0 * 4;
        }
        // x = 7 * omega^bitrev(x_index)
        uint64_t rev = 0;
        for (uint32_t b = 0; b < lde_bits; b++) rev |= ((x_index >> b) & 1) << (lde_bits - 1 - b);
        uint64_t xx = gl_canon(gl_mul(7, gl_pow(omega, rev)));
        // batch_initial_polynomials (fri_chip.rs:112-149): ((sum alpha^i p_i(x)) - reduced opening) / (x - point), batches combined with alpha^|batch|
        evals.clear();
        for (int o = 0; o < 4; o++)
            for (uint32_t i = 0; i < widths[o]; i++) evals.push_back(e_base(leaves[o][i]));
        E total;
        {
            const E num = gl2_sub(reduce_with_powers(evals.data(), evals.size(), fri_alpha), red_zeta);
            const E den = gl2_sub(e_base(xx), zeta);
            if (e_eq(den, gl2_make(0, 0))) return fail("query point equals zeta");
            total = gl2_mul(num, gl2_inv(den));
        }
        {
            evals.clear();
            for (uint32_t i = 0; i < nch; i++) evals.push_back(e_base(leaves[2][i]));
            const E num = gl2_sub(reduce_with_powers(evals.data(), evals.size(), fri_alpha), red_next);
            const E den = gl2_sub(e_base(xx), zeta_next);
            if (e_eq(den, gl2_make(0, 0))) return fail("query point equals g * zeta");
            total = gl2_add(gl2_mul(total, alpha_pow_next), gl2_mul(num, gl2_inv(den)));
        }
        E prev = total;
        uint64_t idx = x_index;
        for (uint32_t l = 0; l < L; l++) {
            for (int i = 0; i < 4; i++) if (q[i] >= GL_P) return fail("non-canonical field element in a FRI layer opening");
            const E ev[2] = {gl2_make(q[0], q[1]), gl2_make(q[2], q[3])};
            const uint64_t* ev_words = q;
            q += 4;
            const uint64_t within = idx & 1, coset_index = idx >> 1;
            if (!e_eq(ev[within], prev)) return fail("FRI fold consistency");
            // next_eval (fri_chip.rs:168-226), arity 2: the two points are (x0, -x0)
            const uint64_t start = within == 0 ? xx : gl_canon(gl_neg(xx));
            const E a0 = e_base(start), b0 = e_base(gl_canon(gl_neg(start)));
            const E numer = gl2_mul(gl2_sub(fri_betas[l], a0), gl2_sub(ev[1], ev[0]));
            prev = gl2_add(ev[0], gl2_mul(numer, gl2_inv(gl2_sub(b0, a0))));
            const uint32_t layers = lde_bits - 1 - l - cap_h;
            if (!merkle_verify(hasher, ev_words, 4, coset_index, q, layers, fri_caps + (uint64_t)l * n_cap * 4, cap_h)) return fail("FRI layer opening does not verify");
            q += (uint64_t)layers * 4;
            xx = gl_canon(gl_mul(xx, xx));
            idx = coset_index;
        }
        if (!e_eq(reduce_with_powers(fpoly.data(), fpoly.size(), e_base(xx)), prev)) return fail("final polynomial");
    }
    if ((uint64_t)(q - proof) != need) return fail("proof length");
    return GL355_OK;
}

}  // extern "C"
