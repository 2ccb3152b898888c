// Kernels of the Halo2 / KZG prover's data-parallel stages (host sequencing in plonk_bn254.hip).  Everything here works on
// bn256::Fr arrays resident in HBM in Montgomery form (8 x 32-bit limbs, bn254_field.cuh; values may be lazily reduced, < 2r), one lane per
// row / coefficient, consecutive lanes on consecutive 32-byte elements.  All of it is integer VALU work on 256-bit operands (a field product
// is 128 v_mad_u64_u32); no kernel here is HBM-bound except the vector ones at the end (linear combinations, conversions).
#pragma once
#include "bn254_field.cuh"
#include "blinding.cuh"

namespace gl355 {

constexpr uint32_t PLK_MAX_REGS = 12;        // halo2.py MAX_REGS
enum { PLK_OP_ADD = 0, PLK_OP_SUB = 1, PLK_OP_MUL = 2, PLK_OP_EMIT = 3, PLK_OP_NEG = 4, PLK_OP_MOV = 5 };
enum { PLK_K_REG = 0, PLK_K_CONST = 1, PLK_K_ADVICE = 2, PLK_K_FIXED = 3, PLK_K_INSTANCE = 4 };
// random-scalar streams (include/gl355.h, oracle/halo2_model.py)
enum { PLK_STREAM_ADVICE = 0x11, PLK_STREAM_LOOKUP_PERMUTED = 0x12, PLK_STREAM_PERM_Z = 0x13, PLK_STREAM_LOOKUP_Z = 0x14, PLK_STREAM_RANDOM_POLY = 0x15 };

GL_DEV u256 fr_one() { return u_const(BN254C_FR_ONE); }
GL_DEV u256 fr_neg(const u256& a) { return m_sub<F_R>(u_zero(), a); }

// ---- conversions / fills ---------------------------------------------------------------------------------------------------------
__global__ void plk_to_mont_kernel(const uint64_t* in, uint64_t* out, uint64_t n) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    store256(out + 4 * i, m_from_int<F_R>(load256(in + 4 * i)));
}
__global__ void plk_from_mont_kernel(const uint64_t* in, uint64_t* out, uint64_t n) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    store256(out + 4 * i, m_to_int<F_R>(load256(in + 4 * i)));
}
// out[i] = scalar `first + i` of stream (stream, a) under the key: 64 key-stream bytes as a 512-bit little-endian integer mod r (the map of
// Fr::from_uniform_bytes): lo + hi 2^256, in Montgomery form lo R + (hi R) R^2 R^-1
__global__ void plk_random_kernel(BlindKey key, uint32_t stream, uint32_t a, uint64_t first, uint64_t count, uint64_t* out) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint64_t idx = first + t;
    uint32_t o[16];
    chacha20_block(key, (uint32_t)idx, stream, a, (uint32_t)(idx >> 32), o);
    u256 lo, hi;
#pragma unroll
    for (int j = 0; j < 8; j++) { lo.l[j] = o[j]; hi.l[j] = o[8 + j]; }
    const u256 v = m_add<F_R>(m_from_int<F_R>(lo), m_mul<F_R>(m_from_int<F_R>(hi), u_const(f_r2<F_R>())));
    store256(out + 4 * t, v);
}
__global__ void plk_fill_kernel(uint64_t* out, uint64_t n, u256 v) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) store256(out + 4 * i, v);
}
// Lagrange-basis indicator columns as values: l_0, l_last (row `usable`), l_active (rows below `usable`)
__global__ void plk_indicator_kernel(uint64_t* l0, uint64_t* llast, uint64_t* lactive, uint64_t n, uint64_t usable) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u256 one = fr_one(), zero = u_zero();
    store256(l0 + 4 * i, i == 0 ? one : zero);
    store256(llast + 4 * i, i == usable ? one : zero);
    store256(lactive + 4 * i, i < usable ? one : zero);
}
// sigma_j[i] = delta^(column) omega^(row) of the cell (j, i) maps to (permutation::keygen::Assembly::build_pk)
__global__ void plk_sigma_kernel(const uint32_t* mapping /* [n_perm][n][2] */, uint64_t n, uint32_t n_perm, const uint64_t* delta_pows, const uint64_t* omega_pows,
                                 uint64_t* out /* [n_perm][n] */, uint32_t* out_of_range) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n * n_perm) return;
    const uint32_t cj = mapping[2 * t], ci = mapping[2 * t + 1];
    if (cj >= n_perm || ci >= n) { atomicOr(out_of_range, 1u); return; }
    store256(out + 4 * t, m_mul<F_R>(load256(delta_pows + 4 * cj), load256(omega_pows + 4 * ci)));
}

// ---- the expression evaluator ------------------------------------------------------------------------------------------------------
// One lane per point of a 2^k-point domain (the value domain or one coset of the extended domain: in both a rotation by r is an index shift
// by r).  The program is straight-line code over PLK_MAX_REGS 256-bit registers; operands name a register, a constant of the pool or a
// column query.  EMIT
// folds a value into the running sum: sum = sum * fold + value -- `fold` is y for the constraint polynomials of evaluate_h and theta for the
// expressions a lookup compresses.
struct PlkEvalArgs {
    const uint32_t* code;            // [n_instr][4]: op, dst, a, b
    uint32_t n_instr;
    const uint64_t* consts;          // Montgomery
    const uint64_t* const* cols[3];  // per kind (advice, fixed, instance): column base pointers
    const int32_t* q_col[3];         // per kind: query -> column
    const int32_t* q_rot[3];         // per kind: query -> rotation
    uint64_t n;
    uint32_t log_n, bitrev;          // bitrev: the columns hold point bitrev(i) at position i (cosets as bn254_fr_ntt_mont_dif leaves them)
    u256 fold;
    const uint64_t* acc_in;          // running sums to continue from, or null (zero)
    uint64_t* acc_out;
};
// position of the point `rot` steps after the point at position i
GL_DEV uint64_t plk_rotated(uint64_t i, int32_t rot, uint64_t n, uint32_t log_n, uint32_t bitrev) {
    if (rot == 0) return i;
    if (!bitrev) return (uint64_t)((int64_t)i + (int64_t)n + rot) & (n - 1);
    const uint64_t j = __brevll(i) >> (64 - log_n);
    return __brevll((uint64_t)((int64_t)j + (int64_t)n + rot) & (n - 1)) >> (64 - log_n);
}
// (Measured alternative: the file in LDS, [register][limb][lane] with one wave per workgroup -- no scratch, but 24 KB per wave leave six waves
// per CU, and evaluate_h at k = 23 took 627 ms against 598 ms with the scratch-backed file below; the scratch lines of a wave stay in L2.)
#define PLK_REG_CASES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
// the register file as twelve named values: the compiler still places it in scratch memory (400 bytes per lane), as it does an indexed array
struct PlkRegs {
#define PLK_DECL(K) u256 r##K;
    PLK_REG_CASES(PLK_DECL)
#undef PLK_DECL
};
GL_DEV u256 plk_reg_read(const PlkRegs& f, uint32_t i) {
    switch (i) {
#define PLK_RD(K) case K: return f.r##K;
        PLK_REG_CASES(PLK_RD)
#undef PLK_RD
    default: return f.r0;
    }
}
GL_DEV void plk_reg_write(PlkRegs& f, uint32_t i, const u256& v) {
    switch (i) {
#define PLK_WR(K) case K: f.r##K = v; break;
        PLK_REG_CASES(PLK_WR)
#undef PLK_WR
    default: break;
    }
}
GL_DEV u256 plk_operand(const PlkEvalArgs& a, const PlkRegs& f, uint32_t operand, uint64_t i) {
    const uint32_t kind = operand >> 24, idx = operand & 0xFFFFFFu;
    if (kind == PLK_K_REG) return plk_reg_read(f, idx);
    if (kind == PLK_K_CONST) return load256(a.consts + 4 * idx);
    const uint32_t kd = kind - PLK_K_ADVICE;
    const int32_t col = a.q_col[kd][idx], rot = a.q_rot[kd][idx];
    return load256(a.cols[kd][col] + 4 * plk_rotated(i, rot, a.n, a.log_n, a.bitrev));
}
__global__ void __launch_bounds__(256) plk_eval_kernel(PlkEvalArgs a) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    PlkRegs f;
#define PLK_ZERO(K) f.r##K = u_zero();
    PLK_REG_CASES(PLK_ZERO)
#undef PLK_ZERO
    u256 acc = a.acc_in ? load256(a.acc_in + 4 * i) : u_zero();
#pragma unroll 1
    for (uint32_t pc = 0; pc < a.n_instr; pc++) {
        const uint32_t op = a.code[4 * pc], dst = a.code[4 * pc + 1], oa = a.code[4 * pc + 2], ob = a.code[4 * pc + 3];
        const u256 x = plk_operand(a, f, oa, i);
        if (op == PLK_OP_EMIT) { acc = m_add<F_R>(m_mul<F_R>(acc, a.fold), x); continue; }
        u256 v;
        if (op == PLK_OP_NEG) v = fr_neg(x);
        else if (op == PLK_OP_MOV) v = x;
        else {
            const u256 y = plk_operand(a, f, ob, i);
            v = op == PLK_OP_ADD ? m_add<F_R>(x, y) : (op == PLK_OP_SUB ? m_sub<F_R>(x, y) : m_mul<F_R>(x, y));
        }
        plk_reg_write(f, dst, v);
    }
    store256(a.acc_out + 4 * i, acc);
}

// ---- evaluate_h: the permutation and lookup constraints on one coset ------------------------------------------------------------------
// (plonk/evaluation.rs: the verifier's `expressions` order -- l_0 (1 - z_0), l_last (z_l^2 - z_l), l_0 (z_s - z_{s-1}(omega^last X)) for s >= 1,
// then per set (z_s(omega X) prod (v + beta sigma + gamma) - z_s(X) prod (v + delta^j beta X + gamma)) l_active)
struct PlkPermHArgs {
    uint64_t n;
    uint32_t log_n;                  // the coset arrays are in bit-reversed order
    uint32_t n_sets, chunk_len, n_perm;
    int32_t last_rot;
    uint64_t* acc;
    const uint64_t *l0, *l_last, *l_active;
    const uint64_t* const* z;        // [n_sets]   coset evaluations of the product polynomials
    const uint64_t* const* sigma;    // [n_perm]   ... of the sigma polynomials
    const uint64_t* const* col;      // [n_perm]   ... of the permutation's columns
    const uint64_t* omega_pows;      // omega^i, i < n
    const uint64_t* delta_pows;      // delta^j, j < n_perm
    u256 y, beta, gamma, coset_base;
};
__global__ void __launch_bounds__(256) plk_perm_h_kernel(PlkPermHArgs a) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint64_t nxt = plk_rotated(i, 1, a.n, a.log_n, 1), lst = plk_rotated(i, a.last_rot, a.n, a.log_n, 1), nat = __brevll(i) >> (64 - a.log_n);
    u256 acc = load256(a.acc + 4 * i);
    const u256 l0 = load256(a.l0 + 4 * i), ll = load256(a.l_last + 4 * i), la = load256(a.l_active + 4 * i), one = fr_one();
    acc = m_add<F_R>(m_mul<F_R>(acc, a.y), m_mul<F_R>(l0, m_sub<F_R>(one, load256(a.z[0] + 4 * i))));
    {
        const u256 zl = load256(a.z[a.n_sets - 1] + 4 * i);
        acc = m_add<F_R>(m_mul<F_R>(acc, a.y), m_mul<F_R>(ll, m_sub<F_R>(m_mul<F_R>(zl, zl), zl)));
    }
#pragma unroll 1
    for (uint32_t s = 1; s < a.n_sets; s++)
        acc = m_add<F_R>(m_mul<F_R>(acc, a.y), m_mul<F_R>(l0, m_sub<F_R>(load256(a.z[s] + 4 * i), load256(a.z[s - 1] + 4 * lst))));
    const u256 bx = m_mul<F_R>(a.beta, m_mul<F_R>(a.coset_base, load256(a.omega_pows + 4 * nat)));      // beta x
#pragma unroll 1
    for (uint32_t s = 0; s < a.n_sets; s++) {
        u256 left = load256(a.z[s] + 4 * nxt), right = load256(a.z[s] + 4 * i);
        const uint32_t j1 = min(a.n_perm, (s + 1) * a.chunk_len);
#pragma unroll 1
        for (uint32_t j = s * a.chunk_len; j < j1; j++) {
            const u256 v = load256(a.col[j] + 4 * i);
            left = m_mul<F_R>(left, m_add<F_R>(m_add<F_R>(v, m_mul<F_R>(a.beta, load256(a.sigma[j] + 4 * i))), a.gamma));
            right = m_mul<F_R>(right, m_add<F_R>(m_add<F_R>(v, m_mul<F_R>(load256(a.delta_pows + 4 * j), bx)), a.gamma));
        }
        acc = m_add<F_R>(m_mul<F_R>(acc, a.y), m_mul<F_R>(m_sub<F_R>(left, right), la));
    }
    store256(a.acc + 4 * i, acc);
}
// one lookup: l_0 (1 - z), l_last (z^2 - z), (z(omega X)(a' + beta)(s' + gamma) - z(X)(A + beta)(S + gamma)) l_active, l_0 (a' - s'),
// (a' - s')(a' - a'(omega^-1 X)) l_active
struct PlkLookupHArgs {
    uint64_t n;
    uint32_t log_n;
    uint64_t* acc;
    const uint64_t *l0, *l_last, *l_active, *z, *ap, *sp, *a_in, *s_in;
    u256 y, beta, gamma;
};
__global__ void __launch_bounds__(256) plk_lookup_h_kernel(PlkLookupHArgs a) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const uint64_t nxt = plk_rotated(i, 1, a.n, a.log_n, 1), prv = plk_rotated(i, -1, a.n, a.log_n, 1);
    u256 acc = load256(a.acc + 4 * i);
    const u256 l0 = load256(a.l0 + 4 * i), ll = load256(a.l_last + 4 * i), la = load256(a.l_active + 4 * i), one = fr_one();
    const u256 z = load256(a.z + 4 * i), ap = load256(a.ap + 4 * i), sp = load256(a.sp + 4 * i);
    acc = m_add<F_R>(m_mul<F_R>(acc, a.y), m_mul<F_R>(l0, m_sub<F_R>(one, z)));
    acc = m_add<F_R>(m_mul<F_R>(acc, a.y), m_mul<F_R>(ll, m_sub<F_R>(m_mul<F_R>(z, z), z)));
    const u256 left = m_mul<F_R>(m_mul<F_R>(load256(a.z + 4 * nxt), m_add<F_R>(ap, a.beta)), m_add<F_R>(sp, a.gamma));
    const u256 right = m_mul<F_R>(m_mul<F_R>(z, m_add<F_R>(load256(a.a_in + 4 * i), a.beta)), m_add<F_R>(load256(a.s_in + 4 * i), a.gamma));
    acc = m_add<F_R>(m_mul<F_R>(acc, a.y), m_mul<F_R>(m_sub<F_R>(left, right), la));
    const u256 d = m_sub<F_R>(ap, sp);
    acc = m_add<F_R>(m_mul<F_R>(acc, a.y), m_mul<F_R>(l0, d));
    acc = m_add<F_R>(m_mul<F_R>(acc, a.y), m_mul<F_R>(m_mul<F_R>(d, m_sub<F_R>(ap, load256(a.ap + 4 * prv))), la));
    store256(a.acc + 4 * i, acc);
}
// h on the extended domain in BIT-REVERSED order (what the gather-free inverse transform reads): point j of coset c is extended index
// j 2^e + c, whose reversal is bitrev_e(c) 2^k + bitrev_k(j) -- the coset's own (bit-reversed) array, whole, at block bitrev_e(c).  Divided by
// X^n - 1, which is constant on a coset.
__global__ void plk_finish_h_kernel(const uint64_t* acc, uint64_t n, uint64_t block, u256 t_inv, uint64_t* h_ext) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    store256(h_ext + 4 * (block * n + i), m_mul<F_R>(load256(acc + 4 * i), t_inv));
}

// ---- grand products ----------------------------------------------------------------------------------------------------------------
// permutation::prover: per row the two products over the columns of one set
struct PlkPermRowArgs {
    uint64_t n;
    uint32_t j0, j1;                 // the set's columns [j0, j1) of the permutation
    const uint64_t* const* vals;     // [n_perm] column values
    const uint64_t* const* sigma;    // [n_perm] sigma values
    const uint64_t* omega_pows;
    const uint64_t* delta_pows;
    u256 beta, gamma;
    uint64_t *num, *den;
};
__global__ void __launch_bounds__(256) plk_perm_rows_kernel(PlkPermRowArgs a) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    u256 num = fr_one(), den = num;
    const u256 bw = m_mul<F_R>(a.beta, load256(a.omega_pows + 4 * i));
#pragma unroll 1
    for (uint32_t j = a.j0; j < a.j1; j++) {
        const u256 v = load256(a.vals[j] + 4 * i);
        den = m_mul<F_R>(den, m_add<F_R>(m_add<F_R>(v, m_mul<F_R>(a.beta, load256(a.sigma[j] + 4 * i))), a.gamma));
        num = m_mul<F_R>(num, m_add<F_R>(m_add<F_R>(v, m_mul<F_R>(load256(a.delta_pows + 4 * j), bw)), a.gamma));
    }
    store256(a.num + 4 * i, num);
    store256(a.den + 4 * i, den);
}
// lookup::prover: num = (A + beta)(S + gamma), den = (A' + beta)(S' + gamma)
__global__ void __launch_bounds__(256) plk_lookup_rows_kernel(const uint64_t* A, const uint64_t* S, const uint64_t* Ap, const uint64_t* Sp, uint64_t n, u256 beta, u256 gamma,
                                                              uint64_t* num, uint64_t* den) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    store256(num + 4 * i, m_mul<F_R>(m_add<F_R>(load256(A + 4 * i), beta), m_add<F_R>(load256(S + 4 * i), gamma)));
    store256(den + 4 * i, m_mul<F_R>(m_add<F_R>(load256(Ap + 4 * i), beta), m_add<F_R>(load256(Sp + 4 * i), gamma)));
}
// out[i] = num[i] / den[i]: a lane takes PLK_INV_CHUNK consecutive rows, one inversion per lane (Montgomery's trick); den is used as scratch.
// A zero denominator (probability ~ n / r over the challenges) is reported through *bad.
constexpr int PLK_INV_CHUNK = 64;        // an inversion is ~380 products: 6 per row at 64 rows per lane (24 at 16)
__global__ void __launch_bounds__(64) plk_batch_div_kernel(const uint64_t* num, uint64_t* den, uint64_t* out, uint64_t n, uint32_t* bad) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t i0 = t * PLK_INV_CHUNK;
    if (i0 >= n) return;
    const uint64_t i1 = min(n, i0 + PLK_INV_CHUNK);
    u256 acc = fr_one();
#pragma unroll 1
    for (uint64_t i = i0; i < i1; i++) {                 // out[i] = product of the denominators before i
        const u256 d = load256(den + 4 * i);
        if (m_is_zero<F_R>(d)) atomicOr(bad, 1u);
        store256(out + 4 * i, acc);
        acc = m_mul<F_R>(acc, d);
    }
    u256 inv = m_inv<F_R>(acc);
#pragma unroll 1
    for (uint64_t i = i1; i-- > i0;) {
        const u256 d = load256(den + 4 * i);
        const u256 dinv = m_mul<F_R>(inv, load256(out + 4 * i));
        inv = m_mul<F_R>(inv, d);
        store256(out + 4 * i, m_mul<F_R>(load256(num + 4 * i), dinv));
    }
}
// Running product: z[0] = start, z[i] = start * prod_{j < i} r[j] (i < n).  Levels like the division of bn254_curve.hip: (1) a lane's chunk
// product, (2) the same problem on the chunk products, (3) a lane walks its chunk from its start value.
constexpr uint32_t PLK_SCAN_CHUNK = 64;
__global__ void plk_scan_chunk_kernel(const uint64_t* r, uint64_t n, uint64_t* P) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t s0 = t * PLK_SCAN_CHUNK;
    if (s0 >= n) return;
    const uint64_t e0 = min(n, s0 + PLK_SCAN_CHUNK);
    u256 p = fr_one();
#pragma unroll 1
    for (uint64_t j = s0; j < e0; j++) p = m_mul<F_R>(p, load256(r + 4 * j));
    store256(P + 4 * t, p);
}
// starts == nullptr: one chunk, lane 0 starts from *start
__global__ void plk_scan_walk_kernel(const uint64_t* r, uint64_t n, const uint64_t* starts, const uint64_t* start, uint64_t* z) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t s0 = t * PLK_SCAN_CHUNK;
    if (s0 >= n) return;
    const uint64_t e0 = min(n, s0 + PLK_SCAN_CHUNK);
    u256 cur = starts ? load256(starts + 4 * t) : load256(start);
#pragma unroll 1
    for (uint64_t j = s0; j < e0; j++) {
        store256(z + 4 * j, cur);
        cur = m_mul<F_R>(cur, load256(r + 4 * j));
    }
}

// ---- the lookup argument's permuted columns (lookup::prover::permute_expression_pair) -----------------------------------------------
GL_DEV bool p_eq(const uint64_t* a, const uint64_t* b) { return ((a[0] ^ b[0]) | (a[1] ^ b[1]) | (a[2] ^ b[2]) | (a[3] ^ b[3])) == 0; }
GL_DEV bool p_less(const uint64_t* a, const uint64_t* b) {
#pragma unroll
    for (int l = 3; l >= 0; l--) { if (a[l] != b[l]) return a[l] < b[l]; }
    return false;
}
__global__ void plk_limb_or_kernel(const uint64_t* v, uint64_t n, unsigned long long* out4) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t l1 = 0, l2 = 0, l3 = 0;
    if (i < n) { l1 = v[4 * i + 1]; l2 = v[4 * i + 2]; l3 = v[4 * i + 3]; }
#pragma unroll
    for (int o = 32; o; o >>= 1) { l1 |= __shfl_xor(l1, o); l2 |= __shfl_xor(l2, o); l3 |= __shfl_xor(l3, o); }
    if (__lane_id() == 0) { if (l1) atomicOr(out4 + 1, l1); if (l2) atomicOr(out4 + 2, l2); if (l3) atomicOr(out4 + 3, l3); }
}
// Montgomery -> plain copies of `cols` columns of n scalars with the rows >= tail zeroed (blinding rows are full-size random scalars: they
// are committed by a second, tiny MSM so that the body keeps its short bit length)
__global__ void plk_from_mont_body_kernel(const uint64_t* in, uint64_t* out, uint64_t n, uint64_t tail, uint64_t total) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint64_t row = t % n;
    store256(out + 4 * t, row < tail ? m_to_int<F_R>(load256(in + 4 * t)) : u_zero());
}
// the tails: out[c][r] = plain(in[c][tail + r]), r < n - tail
__global__ void plk_from_mont_tail_kernel(const uint64_t* in, uint64_t* out, uint64_t n, uint64_t tail, uint32_t cols) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, nt = n - tail;
    if (t >= nt * cols) return;
    const uint64_t c = t / nt, r = t % nt;
    store256(out + 4 * t, m_to_int<F_R>(load256(in + 4 * (c * n + tail + r))));
}
// per column (blockIdx.y) the OR of every scalar's four limbs: the column's bit length.  A few hundred blocks per column stride over it and
// issue one atomic each (one atomic per wave on the same four words serialised 131 072 of them: 14 ms per call at k = 23)
__global__ void __launch_bounds__(256) plk_column_or_kernel(const uint64_t* v, uint64_t n, unsigned long long* out /* [columns][4] */) {
    __shared__ unsigned long long sh[4][4];
    const uint64_t* col = v + 4 * (uint64_t)blockIdx.y * n;
    uint64_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        l0 |= col[4 * i]; l1 |= col[4 * i + 1]; l2 |= col[4 * i + 2]; l3 |= col[4 * i + 3];
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) { l0 |= __shfl_xor(l0, o); l1 |= __shfl_xor(l1, o); l2 |= __shfl_xor(l2, o); l3 |= __shfl_xor(l3, o); }
    const uint32_t wave = threadIdx.x >> 6;
    if (__lane_id() == 0) { sh[wave][0] = l0; sh[wave][1] = l1; sh[wave][2] = l2; sh[wave][3] = l3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long o = sh[0][threadIdx.x] | sh[1][threadIdx.x] | sh[2][threadIdx.x] | sh[3][threadIdx.x];
        if (o) atomicOr(out + 4 * blockIdx.y + threadIdx.x, o);
    }
}
__global__ void plk_iota_kernel(uint32_t* idx, uint64_t n) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)i;
}
__global__ void plk_gather_limb_kernel(const uint64_t* v, const uint32_t* idx, uint32_t limb, uint64_t* keys, uint64_t n) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) keys[i] = v[4 * (uint64_t)idx[i] + limb];
}
__global__ void plk_gather_rows_kernel(const uint64_t* v, const uint32_t* idx, uint64_t* out, uint64_t n) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t* s = v + 4 * (uint64_t)idx[i];
    out[4 * i] = s[0]; out[4 * i + 1] = s[1]; out[4 * i + 2] = s[2]; out[4 * i + 3] = s[3];
}
// a: the input sorted, t: the table sorted (plain integers, u rows each).  rep[i] = 1 where a[i] repeats a[i - 1]; left[i] = 1 where t[i] is
// NOT the first copy of a value that occurs in a (those first copies face the first occurrences of their value in the permuted table)
__global__ void plk_lookup_flags_kernel(const uint64_t* a, const uint64_t* t, uint64_t u, uint32_t* rep, uint32_t* left) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= u) return;
    rep[i] = (i > 0 && p_eq(a + 4 * i, a + 4 * (i - 1))) ? 1u : 0u;
    bool consumed = false;
    if (i == 0 || !p_eq(t + 4 * i, t + 4 * (i - 1))) {
        uint64_t lo = 0, hi = u;                                  // lower bound of t[i] in a
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (p_less(a + 4 * mid, t + 4 * i)) lo = mid + 1; else hi = mid;
        }
        consumed = lo < u && p_eq(a + 4 * lo, t + 4 * i);
    }
    left[i] = consumed ? 0u : 1u;
}
// leftovers (ascending) to a dense list
__global__ void plk_lookup_compact_kernel(const uint64_t* t, const uint32_t* left, const uint32_t* left_pos, uint64_t u, uint64_t* list) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= u || !left[i]) return;
    uint64_t* d = list + 4 * (uint64_t)left_pos[i];
    d[0] = t[4 * i]; d[1] = t[4 * i + 1]; d[2] = t[4 * i + 2]; d[3] = t[4 * i + 3];
}
// the permuted table: a first occurrence faces its own value; the r-th repeated row (ascending) takes leftover n_rep - 1 - r
__global__ void plk_lookup_table_kernel(const uint64_t* a, const uint32_t* rep, const uint32_t* rep_pos, const uint64_t* list, uint32_t n_rep, uint64_t u, uint64_t* out) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= u) return;
    const uint64_t* s = rep[i] ? list + 4 * (uint64_t)(n_rep - 1 - rep_pos[i]) : a + 4 * i;
    out[4 * i] = s[0]; out[4 * i + 1] = s[1]; out[4 * i + 2] = s[2]; out[4 * i + 3] = s[3];
}

// ---- evaluations and vector operations -----------------------------------------------------------------------------------------------
// value of polynomial q (coefficients, n of them) at point q: blocks of 256 lanes take segments of 256 x PLK_EVAL_PER coefficients, lane t
// the coefficients seg + t + 256 j by Horner in x^256 (coalesced), times x^(seg + t); LDS tree; one partial per block
constexpr uint32_t PLK_EVAL_PER = 32;
struct PlkEvalPolyArgs {
    const uint64_t* const* polys;    // [n_q]
    const uint64_t* points;          // [n_q] Montgomery
    const uint64_t* points256;       // [n_q] x^256
    uint64_t n;
    uint32_t blocks_per_q;
    uint64_t* partial;               // [n_q][blocks_per_q]
};
__global__ void __launch_bounds__(256) plk_eval_poly_kernel(PlkEvalPolyArgs a) {
    __shared__ uint32_t sh[8][256];
    const uint32_t q = blockIdx.y, tid = threadIdx.x;
    const uint64_t seg = (uint64_t)blockIdx.x * 256 * PLK_EVAL_PER;
    const uint64_t* c = a.polys[q];
    const u256 x = load256(a.points + 4 * q), x256 = load256(a.points256 + 4 * q);
    u256 acc = u_zero();
#pragma unroll 1
    for (int j = PLK_EVAL_PER - 1; j >= 0; j--) {
        const uint64_t i = seg + tid + 256ull * j;
        acc = m_mul<F_R>(acc, x256);
        if (i < a.n) acc = m_add<F_R>(acc, load256(c + 4 * i));
    }
    acc = m_mul<F_R>(acc, m_pow_u64<F_R>(x, seg + tid));
#pragma unroll
    for (int l = 0; l < 8; l++) sh[l][tid] = acc.l[l];
    __syncthreads();
    for (uint32_t st = 128; st; st >>= 1) {
        if (tid < st) {
            u256 p, r;
#pragma unroll
            for (int l = 0; l < 8; l++) { p.l[l] = sh[l][tid]; r.l[l] = sh[l][tid + st]; }
            p = m_add<F_R>(p, r);
#pragma unroll
            for (int l = 0; l < 8; l++) sh[l][tid] = p.l[l];
        }
        __syncthreads();
    }
    if (tid == 0) {
        u256 p;
#pragma unroll
        for (int l = 0; l < 8; l++) p.l[l] = sh[l][0];
        store256(a.partial + 4 * ((uint64_t)q * a.blocks_per_q + blockIdx.x), p);
    }
}
__global__ void plk_eval_sum_kernel(const uint64_t* partial, uint32_t n_q, uint32_t blocks_per_q, uint64_t* out /* Montgomery, canonical */) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_q) return;
    u256 s = u_zero();
    for (uint32_t b = 0; b < blocks_per_q; b++) s = m_add<F_R>(s, load256(partial + 4 * ((uint64_t)q * blocks_per_q + b)));
    store256(out + 4 * q, m_canon<F_R>(s));
}
// out[i] = (acc_in ? acc_in[i] : 0) + sum_j coeff[j] polys[j][i]  -  low[i] (i < n_low)
struct PlkLincombArgs {
    const uint64_t* const* polys;
    const uint64_t* coeffs;          // Montgomery
    uint32_t count, n_low;
    uint64_t n;
    const uint64_t* acc_in;
    uint64_t* out;
    u256 low[4];
};
__global__ void __launch_bounds__(256) plk_lincomb_kernel(PlkLincombArgs a) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    u256 s = a.acc_in ? load256(a.acc_in + 4 * i) : u_zero();
#pragma unroll 1
    for (uint32_t j = 0; j < a.count; j++) s = m_add<F_R>(s, m_mul<F_R>(load256(a.coeffs + 4 * j), load256(a.polys[j] + 4 * i)));
    if (i < a.n_low) s = m_sub<F_R>(s, a.low[i]);
    store256(a.out + 4 * i, s);
}

}  // namespace gl355
