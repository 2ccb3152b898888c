// Constraint / quotient batch evaluation for gfx950 (a10 of SURVEY.md 8).
//
// Replaces plonky2::plonk::prover::compute_quotient_polys ->
// vanishing_poly::eval_vanishing_poly_base_batch -> Gate::eval_unfiltered_base_batch (+ ZeroPolyOnCoset),
// reached from the reference through CircuitData::prove (src/plonky2_semaphore/access_set.rs:94,
// recursion.rs:168, wrapper.rs:55).  The formula is the one the reference's verifier re-evaluates at
// zeta: src/plonky2_verifier/chip/plonk/vanishing_poly.rs:18-153 (term order :110-123), gate filter
// chip/plonk/gates/mod.rs:87-132, quotient identity chip/plonk/plonk_verifier_chip.rs:174-210, and
// the gate evaluators chip/plonk/gates/*.rs (all 12 gate kinds of the dispatch table gates/mod.rs:141-196).
//
// One lane = one point x = 7*omega^i of the quotient coset (size n * 2^qdb).  The three committed
// oracles are column-major in bit-reversed row order, and lane t works on storage row t, so every
// one of the ~240 column reads per point is a perfectly coalesced 512-byte wave access; nothing is
// gathered except the 2 "next row" Z values.  All arithmetic is base field (the challenges alpha,
// beta, gamma are base-field elements, plonk_verifier_chip.rs:64-96).  The term list
//   [L0(x)(Z_c(x)-1)]_c || [prev*prod(num) - next*prod(den)]_{c,chunk} || [sum_g filter_g * constraint_{g,k}]_k
// is reduced with powers of each alpha on the fly, so no per-lane constraint array exists.
// Roofline: integer VALU (a PoseidonGate row costs about one permutation), not HBM.
#include "gl355_internal.h"
#include "poseidon.cuh"

namespace gl355 {

struct QuotArgs {
    gl355_circuit c;
    const uint64_t* cs;       // constants_sigmas LDE  [num_selectors + num_constants + routed][N]
    const uint64_t* wires;    // wires LDE             [num_wires (+salt)][N]
    const uint64_t* zs;       // Z + partial products  [num_challenges * (1 + num_pp) (+salt)][N]
    uint64_t lde_stride;      // N
    uint32_t qbits;           // log2 of the quotient domain
    const uint64_t* k_is;     // [routed]
    const uint64_t* xw_lo;    // omega_{Nq}^e two-level table
    const uint64_t* xw_hi;
    uint64_t zh_inv[16];      // 1 / (x^n - 1) for the 2^qdb classes of i mod 2^qdb
    uint64_t zh[16];          // x^n - 1
    uint64_t n_inv;           // 1/n
    uint64_t* out;            // [unit][num_challenges][Nq], storage (bit-reversed) order
    const uint64_t* alpha_pw; // alpha_{u,c}^k at [(u * num_challenges + c) * pw_stride + k], k < pw_stride (alpha_table_kernel)
    uint32_t pw_stride;
    // the lock-step units of one prover context (blockIdx.y): constants_sigmas is shared, wires / zs / out have unit strides,
    // every unit has its own challenges and public-input hash
    uint64_t wires_us, zs_us;
    uint64_t betas[GL355_MAX_UNITS * 4], gammas[GL355_MAX_UNITS * 4], pi_hash[GL355_MAX_UNITS * 4];
};

// alpha-power accumulation of the constraint stream: term k of the stream is weighted alpha_c^k.  The powers are the same for
// every lane, so they come from a table built once per launch (alpha_table_kernel) and read through the scalar cache
// (constant address space, wave-uniform index) instead of being multiplied along per lane -- one product per push and
// challenge instead of two.  NCH (the number of challenges) is a template parameter so that acc stays in registers (a
// run-time bound puts the array in scratch memory and every push() becomes scratch loads + stores).
typedef const __attribute__((address_space(4))) uint64_t* PwTable;
template <int NCH>
struct AlphaAcc {
    uint64_t acc[NCH];
    PwTable tab;
    uint32_t stride, idx;
    GL_DEV void init(const QuotArgs& a, uint32_t unit, uint32_t first = 0) {
#pragma unroll
        for (int c = 0; c < NCH; c++) acc[c] = 0;
        tab = (PwTable)(a.alpha_pw + (uint64_t)unit * NCH * a.pw_stride); stride = a.pw_stride; idx = first;
    }
    GL_DEV void push(uint64_t term) {
#pragma unroll
        for (int c = 0; c < NCH; c++) acc[c] = gl_add(acc[c], gl_mul(term, tab[c * stride + idx]));
        idx++;
    }
    GL_DEV void push_at(uint64_t term, uint32_t k) {       // term k of the stream, out of order
#pragma unroll
        for (int c = 0; c < NCH; c++) acc[c] = gl_add(acc[c], gl_mul(term, tab[c * stride + k]));
    }
};
// per-gate accumulation: sum_k alpha^(base+k) * c_k, later multiplied by the gate's filter.
//
// LAZILY REDUCED for two challenges (round 3): about 500 terms are pushed per point of the recursive circuit and a push was a modular
// product and a modular sum per challenge -- 40 of the kernel's ~76 VALU instructions per term.  Here a gate's sum is kept as three 64-bit
// columns (weights 1, 2^32, 2^64) with a 32-bit overflow count each; a push is four multiply-adds per challenge whose carry-outs
// are added into the counts -- 16 instructions for both challenges -- and the field value is formed once per gate (value()).
// The carry-outs live in scalar register pairs; gfx950 wants two wait states between a VALU write of a scalar register and the VALU read
// of it and the hazard recogniser does not look inside inline asm, so the two challenges' instructions are interleaved (every carry
// is consumed four instructions after it was produced).
#ifndef GL355_QUOT_LAZY
#define GL355_QUOT_LAZY 1
#endif
template <int NCH>
struct LazyAcc : AlphaAcc<NCH> {
    GL_DEV uint64_t value(int c) const { return this->acc[c]; }
};
#if GL355_QUOT_LAZY && defined(__HIP_DEVICE_COMPILE__)
template <>
struct LazyAcc<2> {
    uint64_t A0[2], A1[2], A2[2];
    uint32_t k0[2], k1[2], k2[2];
    PwTable tab;
    uint32_t stride, idx;
    GL_DEV void init(const QuotArgs& a, uint32_t unit, uint32_t first = 0) {
#pragma unroll
        for (int c = 0; c < 2; c++) { A0[c] = A1[c] = A2[c] = 0; k0[c] = k1[c] = k2[c] = 0; }
        tab = (PwTable)(a.alpha_pw + (uint64_t)unit * 2 * a.pw_stride); stride = a.pw_stride; idx = first;
    }
    GL_DEV void push(uint64_t term) {
        const uint64_t ax = tab[idx], ay = tab[stride + idx];
        const uint32_t t0 = (uint32_t)term, t1 = (uint32_t)(term >> 32);
        uint64_t c0, c1, c2, c3;
        asm("v_mad_u64_u32 %0, %12, %16, %18, %0\n\t"
            "v_mad_u64_u32 %1, %13, %16, %20, %1\n\t"
            "v_mad_u64_u32 %2, %14, %16, %19, %2\n\t"
            "v_mad_u64_u32 %3, %15, %16, %21, %3\n\t"
            "v_addc_co_u32_e64 %6, vcc, 0, %6, %12\n\t"
            "v_addc_co_u32_e64 %7, vcc, 0, %7, %13\n\t"
            "v_addc_co_u32_e64 %8, vcc, 0, %8, %14\n\t"
            "v_addc_co_u32_e64 %9, vcc, 0, %9, %15\n\t"
            "v_mad_u64_u32 %2, %12, %17, %18, %2\n\t"
            "v_mad_u64_u32 %3, %13, %17, %20, %3\n\t"
            "v_mad_u64_u32 %4, %14, %17, %19, %4\n\t"
            "v_mad_u64_u32 %5, %15, %17, %21, %5\n\t"
            "v_addc_co_u32_e64 %8, vcc, 0, %8, %12\n\t"
            "v_addc_co_u32_e64 %9, vcc, 0, %9, %13\n\t"
            "v_addc_co_u32_e64 %10, vcc, 0, %10, %14\n\t"
            "v_addc_co_u32_e64 %11, vcc, 0, %11, %15"
            : "+v"(A0[0]), "+v"(A0[1]), "+v"(A1[0]), "+v"(A1[1]), "+v"(A2[0]), "+v"(A2[1]),          // 0..5
              "+v"(k0[0]), "+v"(k0[1]), "+v"(k1[0]), "+v"(k1[1]), "+v"(k2[0]), "+v"(k2[1]),          // 6..11
              "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)                                                // 12..15
            : "v"(t0), "v"(t1),                                                                          // 16, 17
              "s"((uint32_t)ax), "s"((uint32_t)(ax >> 32)), "s"((uint32_t)ay), "s"((uint32_t)(ay >> 32)) // 18..21
            : "vcc");
        idx++;
    }
    // A0 + 2^32 A1 + 2^64 (A2 + k0) + 2^96 k1 + 2^128 k2  with  2^96 = -1, 2^128 = -2^32 (mod p)
    GL_DEV uint64_t value(int c) const {
        uint64_t r = gl_reduce128(A0[c], A2[c]);
        r = gl_add(r, gl_mul_2exp<32>(A1[c]));
        r = gl_add(r, gl_reduce128(0, k0[c]));
        r = gl_sub(r, k1[c]);
        return gl_sub(r, (uint64_t)k2[c] << 32);
    }
};
#endif
template <int NCH>
using GateAccT = LazyAcc<NCH>;

// alpha_c^k for k < stride, one thread per entry (square-and-multiply over the bits of k)
struct AlphaTabArgs { uint64_t alphas[GL355_MAX_UNITS * 4]; uint32_t nch, stride; uint64_t* out; };
__global__ void alpha_table_kernel(AlphaTabArgs a) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, u = blockIdx.y;
    if (k >= a.stride) return;
    for (uint32_t c = 0; c < a.nch; c++) a.out[((uint64_t)u * a.nch + c) * a.stride + k] = gl_canon(gl_pow(a.alphas[u * 4 + c], k));
}

// a point's wire row: column j of the wires LDE at this lane's row (global memory), or of the LDS copy of the row (STAGE variant)
struct WireSrc { const uint64_t* p; uint64_t stride, idx; };
#define WIRE(j) (w.p[(uint64_t)(j) * w.stride + w.idx])
#define CONST(j) (a.cs[(uint64_t)(j) * a.lde_stride + t])

// PoseidonGate: 123 constraints (gates/poseidon.rs:592-698); wire layout :329-380
template <class GateAcc>
GL_DEV void gate_poseidon(const QuotArgs& a, const WireSrc& w, uint64_t t, GateAcc& g) {
    const uint64_t swap = WIRE(24);
    g.push(gl_sub(gl_mul(swap, swap), swap));
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t lhs = WIRE(i), rhs = WIRE(i + 4), delta = WIRE(25 + i);
        g.push(gl_sub(gl_mul(swap, gl_sub(rhs, lhs)), delta));
        s[i] = gl_add(lhs, delta);
        s[i + 4] = gl_sub(rhs, delta);
    }
#pragma unroll
    for (int i = 8; i < 12; i++) s[i] = WIRE(i);
    // every MDS layer also adds the NEXT round's constants (psd_mds; row 30 of the table is zero), so s is "state + constants",
    // the quantity the S-box-input wires are constrained to, whenever a round starts
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_canonical(s[i], PSD_ALL_RC[i]);
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        if (r != 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const uint64_t sin = WIRE(29 + 12 * (r - 1) + i);
                g.push(gl_sub(s[i], sin));
                s[i] = sin;
            }
        }
#pragma unroll
        for (int i = 0; i < 12; i += 4) psd_sbox4(s[i], s[i + 1], s[i + 2], s[i + 3]);       // four S-boxes in lock-step (poseidon.cuh)
        psd_mds(s, &PSD_ALL_RC[12 * (r + 1)]);
    }
    // partial rounds in the dense form: the S-box input of round r is s[0] + RC[4+r][0] in either form, so the
    // 22 constraints (and the state handed to the closing full rounds) are the same polynomials in the wires
    // as in the reference's sparse formulation (gates/poseidon.rs:652-673), at fewer VALU cycles
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
        const uint64_t sin = WIRE(65 + r);
        g.push(gl_sub(s[0], sin));
        s[0] = psd_sbox(sin);
        psd_mds(s, &PSD_ALL_RC[12 * (5 + r)]);
    }
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const uint64_t sin = WIRE(87 + 12 * r + i);
            g.push(gl_sub(s[i], sin));
            s[i] = sin;
        }
#pragma unroll
        for (int i = 0; i < 12; i += 4) psd_sbox4(s[i], s[i + 1], s[i + 2], s[i + 3]);
        psd_mds(s, &PSD_ALL_RC[12 * (27 + r)]);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) g.push(gl_sub(s[i], WIRE(12 + i)));
}

// BaseSumGate<2>{num_limbs}: sum_i limb_i 2^i - sum ; limb (limb - 1)   (gates/base_sum.rs:37-60)
template <class GateAcc>
GL_DEV void gate_base_sum(const QuotArgs& a, const WireSrc& w, uint64_t t, GateAcc& g, uint32_t num_limbs) {
    uint64_t acc = 0;
    for (uint32_t i = num_limbs; i-- > 0;) acc = gl_add(gl_add(acc, acc), WIRE(1 + i));
    g.push(gl_sub(acc, WIRE(0)));
    for (uint32_t i = 0; i < num_limbs; i++) {
        const uint64_t l = WIRE(1 + i);
        g.push(gl_sub(gl_mul(l, l), l));
    }
}
// ConstantGate{n}: const_i - wire_i   (gates/constant.rs:31-36); gate constants follow the selectors
template <class GateAcc>
GL_DEV void gate_constant(const QuotArgs& a, const WireSrc& w, uint64_t t, GateAcc& g, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) g.push(gl_sub(CONST(a.c.num_selectors + i), WIRE(i)));
}
// PublicInputGate: wire_i - pi_hash_i   (gates/public_input.rs:32-39)
template <class GateAcc>
GL_DEV void gate_public_input(const QuotArgs& a, const WireSrc& w, uint64_t t, GateAcc& g, uint32_t unit) {
    for (uint32_t i = 0; i < 4; i++) g.push(gl_sub(WIRE(i), a.pi_hash[unit * 4 + i]));
}
// ArithmeticGate{num_ops}: out - (c0 m0 m1 + c1 addend)   (gates/arithmetic.rs:47-68)
template <class GateAcc>
GL_DEV void gate_arithmetic(const QuotArgs& a, const WireSrc& w, uint64_t t, GateAcc& g, uint32_t num_ops) {
    const uint64_t c0 = CONST(a.c.num_selectors), c1 = CONST(a.c.num_selectors + 1);
    for (uint32_t i = 0; i < num_ops; i++) {
        const uint64_t m0 = WIRE(4 * i), m1 = WIRE(4 * i + 1), ad = WIRE(4 * i + 2), out = WIRE(4 * i + 3);
        g.push(gl_sub(out, gl_add(gl_mul(gl_mul(m0, m1), c0), gl_mul(ad, c1))));
    }
}

// ---- gates over the extension algebra: an F_p^2 element occupies two consecutive wires; on the coset
// every wire value is a base-field number, so the algebra product is the plain F_p^2 product
// (chip/goldilocks_extension_algebra_chip.rs:112-146).  Each algebra constraint yields 2 terms.
#define WIRE2(j) gl2_make(WIRE(j), WIRE((j) + 1))
template <class GateAcc>
GL_DEV void push2(GateAcc& g, gl2 v) { g.push(v.c0); g.push(v.c1); }

// ArithmeticExtensionGate{num_ops}: out - (c0 m0 m1 + c1 addend)   (gates/arithmetic_extension.rs:22-80)
template <class GateAcc>
GL_DEV void gate_arithmetic_ext(const QuotArgs& a, const WireSrc& w, uint64_t t, GateAcc& g, uint32_t num_ops) {
    const uint64_t c0 = CONST(a.c.num_selectors), c1 = CONST(a.c.num_selectors + 1);
    for (uint32_t i = 0; i < num_ops; i++) {
        const gl2 m0 = WIRE2(8 * i), m1 = WIRE2(8 * i + 2), ad = WIRE2(8 * i + 4), out = WIRE2(8 * i + 6);
        const gl2 comp = gl2_add(gl2_mul_base(gl2_mul(m0, m1), c0), gl2_mul_base(ad, c1));
        push2(g, gl2_sub(out, comp));
    }
}
// MulExtensionGate{num_ops}: out - c0 m0 m1   (gates/multiplication_extension.rs:22-68)
template <class GateAcc>
GL_DEV void gate_mul_ext(const QuotArgs& a, const WireSrc& w, uint64_t t, GateAcc& g, uint32_t num_ops) {
    const uint64_t c0 = CONST(a.c.num_selectors);
    for (uint32_t i = 0; i < num_ops; i++) {
        const gl2 m0 = WIRE2(6 * i), m1 = WIRE2(6 * i + 2), out = WIRE2(6 * i + 4);
        push2(g, gl2_sub(out, gl2_mul_base(gl2_mul(m0, m1), c0)));
    }
}
// PoseidonMdsGate: out_r - sum_i CIRC[i] in[(i+r)%12] - DIAG[r] in[r]   (gates/poseidon_mds.rs:26-126)
template <class GateAcc>
GL_DEV void gate_poseidon_mds(const QuotArgs& a, const WireSrc& w, uint64_t t, GateAcc& g) {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    for (uint32_t r = 0; r < 12; r++) {
        gl2 acc = gl2_make(0, 0);
        for (uint32_t i = 0; i < 12; i++) {
            const gl2 in = WIRE2(2 * ((i + r) % 12));
            acc = gl2_add(acc, gl2_make(gl_mul_small(in.c0, CIRC[i]), gl_mul_small(in.c1, CIRC[i])));
        }
        if (r == 0) {
            const gl2 in = WIRE2(0);
            acc = gl2_add(acc, gl2_make(gl_mul_small(in.c0, 8), gl_mul_small(in.c1, 8)));
        }
        push2(g, gl2_sub(WIRE2(2 * (12 + r)), acc));
    }
}
// RandomAccessGate{bits, copies, extra}   (gates/random_access.rs:27-147)
template <class GateAcc>
GL_DEV void gate_random_access(const QuotArgs& a, const WireSrc& w, uint64_t t, GateAcc& g, uint32_t param) {
    const uint32_t bits = param & 0xFF, copies = (param >> 8) & 0xFF, extra = (param >> 16) & 0xFF;
    const uint32_t vec = 1u << bits, routed = (2 + vec) * copies + extra;
    for (uint32_t c = 0; c < copies; c++) {
        const uint32_t base = (2 + vec) * c;
        uint64_t recon = 0;
        for (uint32_t i = 0; i < bits; i++) {
            const uint64_t b = WIRE(routed + c * bits + i);
            g.push(gl_sub(gl_mul(b, b), b));
        }
        for (uint32_t i = bits; i-- > 0;) recon = gl_add(gl_add(recon, recon), WIRE(routed + c * bits + i));
        g.push(gl_sub(recon, WIRE(base)));
        // fold the list: item = x + b (y - x) per pair, one bit per level (bits <= 4 => <= 16 items)
        uint64_t items[16];
        for (uint32_t i = 0; i < 16; i++) items[i] = i < vec ? WIRE(base + 2 + i) : 0;
        uint32_t len = vec;
        for (uint32_t lvl = 0; lvl < bits; lvl++) {
            const uint64_t b = WIRE(routed + c * bits + lvl);
            for (uint32_t k = 0; k < 8; k++) {
                if (k < len / 2) items[k] = gl_add(gl_mul(b, gl_sub(items[2 * k + 1], items[2 * k])), items[2 * k]);
            }
            len >>= 1;
        }
        g.push(gl_sub(items[0], WIRE(base + 1)));
    }
    for (uint32_t i = 0; i < extra; i++) g.push(gl_sub(CONST(a.c.num_selectors + i), WIRE((2 + vec) * copies + i)));
}
// ReducingGate{n} / ReducingExtensionGate{n}: acc*alpha + coeff - acc_i   (gates/reducing.rs:20-85,
// gates/reducing_extension.rs:20-87); the last accumulator is the output (wires 0..1)
template <bool EXT, class GateAcc>
GL_DEV void gate_reducing(const QuotArgs& a, const WireSrc& w, uint64_t t, GateAcc& g, uint32_t n) {
    const gl2 alpha = WIRE2(2);
    gl2 acc = WIRE2(4);
    const uint32_t start_accs = 6 + (EXT ? 2 * n : n);
    for (uint32_t i = 0; i < n; i++) {
        const gl2 coeff = EXT ? WIRE2(6 + 2 * i) : gl2_make(WIRE(6 + i), 0);
        const gl2 acc_i = (i == n - 1) ? WIRE2(0) : WIRE2(start_accs + 2 * i);
        push2(g, gl2_sub(gl2_add(gl2_mul(acc, alpha), coeff), acc_i));
        acc = acc_i;
    }
}

// STAGE (experiment of round 2, not instantiated any more: profiles/r02_quotient_ab.txt): the wave first copies its 64 points' wire rows (num_wires x 64 x 8 B = 69 KB) into LDS and
// every evaluator reads them from there -- each wire column crosses the fabric once instead of ~4 times, at 2 waves per CU.
template <int NCH, bool STAGE = false>
__global__ void __launch_bounds__(STAGE ? 64 : 128) __attribute__((amdgpu_waves_per_eu(STAGE ? 1 : 3, 3))) quotient_kernel(QuotArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t wire_lds[];
    const uint64_t nq = 1ull << a.qbits;
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= nq) return;
    const uint32_t unit = blockIdx.y;
    const uint64_t* __restrict__ wglob = a.wires + (uint64_t)unit * a.wires_us;
    WireSrc w{wglob, a.lde_stride, t};
    if constexpr (STAGE) {
        for (uint32_t j = 0; j < a.c.num_wires; j++) wire_lds[j * 64 + threadIdx.x] = wglob[(uint64_t)j * a.lde_stride + t];
        w = WireSrc{wire_lds, 64, threadIdx.x};        // one wave per block: no barrier needed, each lane reads only its own column entries
    }
    const uint64_t* __restrict__ zs = a.zs + (uint64_t)unit * a.zs_us;
    const uint32_t qdb = a.qbits - a.c.degree_bits;
    const uint64_t iq = __brevll(t) >> (64 - a.qbits);   // natural index of storage row t
    const uint64_t x = gl_mul_small(gl_mul(a.xw_lo[iq & 4095], a.xw_hi[iq >> 12]), 7);
    // next row: natural index iq + 2^qdb (g*x), its storage row is bitrev of that
    const uint64_t iq_next = (iq + (1ull << qdb)) & (nq - 1);
    const uint64_t t_next = __brevll(iq_next) >> (64 - a.qbits);
    constexpr uint32_t nch = NCH;
    const uint32_t npp = a.c.num_partial_products;
    const uint32_t routed = a.c.num_routed_wires, chunk = a.c.max_degree;
    const uint32_t n_sel = a.c.num_selectors, n_cst = a.c.num_constants;

    AlphaAcc<NCH> total;
    total.init(a, unit);
    // ---- L0(x) (Z_c(x) - 1);  L0(x) = (x^n - 1) / (n (x - 1))  (vanishing_poly.rs:155-178) ---------
    const uint64_t zh_inv = a.zh_inv[iq & ((1u << qdb) - 1)];
    const uint64_t zh = a.zh[iq & ((1u << qdb) - 1)];  // x^n - 1 (never zero on the coset)
    const uint64_t l0 = gl_mul(gl_mul(zh, a.n_inv), gl_inv(gl_sub(x, 1)));
    for (uint32_t c = 0; c < nch; c++) {
        const uint64_t z = zs[(uint64_t)c * a.lde_stride + t];
        total.push(gl_sub(gl_mul(l0, z), l0));
    }
    // ---- partial products (vanishing_poly.rs:54-108, 183-218) -------------------------------------
    // every routed wire and its sigma value are loaded ONCE and serve all challenges (the term of (challenge c, chunk ch) keeps
    // its place c * n_chunks + ch in the stream: push_at)
    {
        const uint32_t n_chunks = (routed + chunk - 1) / chunk;
        uint64_t prev[NCH], bx[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            prev[c] = zs[(uint64_t)c * a.lde_stride + t];
            bx[c] = gl_mul(a.betas[unit * 4 + c], x);
        }
        const uint32_t first = total.idx;
        for (uint32_t ch = 0; ch < n_chunks; ch++) {
            uint64_t num[NCH], den[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) { num[c] = 1; den[c] = 1; }
            for (uint32_t j = ch * chunk; j < (ch + 1) * chunk && j < routed; j++) {
                const uint64_t wj = WIRE(j), sj = CONST(n_sel + n_cst + j), kj = a.k_is[j];
                if constexpr (NCH == 2) {
                    // the eight products of a wire in two lock-step groups of four (gl_mul_multi: partners fill the carry wait states)
                    const uint64_t a1[4] = {bx[0], bx[1], a.betas[unit * 4], a.betas[unit * 4 + 1]}, b1[4] = {kj, kj, sj, sj};
                    uint64_t p1[4], p2[4];
                    gl_mul_multi<4>(a1, b1, p1);
                    const uint64_t wg0 = gl_add(wj, a.gammas[unit * 4]), wg1 = gl_add(wj, a.gammas[unit * 4 + 1]);
                    const uint64_t a2[4] = {num[0], num[1], den[0], den[1]};
                    const uint64_t b2[4] = {gl_add(wg0, p1[0]), gl_add(wg1, p1[1]), gl_add(wg0, p1[2]), gl_add(wg1, p1[3])};
                    gl_mul_multi<4>(a2, b2, p2);
                    num[0] = p2[0]; num[1] = p2[1]; den[0] = p2[2]; den[1] = p2[3];
                } else {
#pragma unroll
                    for (int c = 0; c < NCH; c++) {
                        const uint64_t wg = gl_add(wj, a.gammas[unit * 4 + c]);
                        num[c] = gl_mul(num[c], gl_add(wg, gl_mul(bx[c], kj)));
                        den[c] = gl_mul(den[c], gl_add(wg, gl_mul(a.betas[unit * 4 + c], sj)));
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const uint64_t next = (ch + 1 < n_chunks) ? zs[(uint64_t)(nch + c * npp + ch) * a.lde_stride + t]
                                                          : zs[(uint64_t)c * a.lde_stride + t_next];
                total.push_at(gl_sub(gl_mul(prev[c], num[c]), gl_mul(next, den[c])), first + c * n_chunks + ch);
                prev[c] = next;
            }
        }
        total.idx = first + nch * n_chunks;
    }
    // ---- gate constraints, each gate's stream multiplied by its filter (gates/mod.rs:87-132) ---------
    uint64_t gate_sum[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) gate_sum[c] = 0;
    for (uint32_t gi = 0; gi < a.c.num_gates; gi++) {
        const gl355_gate gt = a.c.gates[gi];
        if (gt.type == GL355_GATE_NOOP) continue;
        GateAccT<NCH> g;
        g.init(a, unit, total.idx);
        switch (gt.type) {
            case GL355_GATE_POSEIDON: gate_poseidon(a, w, t, g); break;
            case GL355_GATE_BASE_SUM: gate_base_sum(a, w, t, g, gt.param); break;
            case GL355_GATE_CONSTANT: gate_constant(a, w, t, g, gt.param); break;
            case GL355_GATE_PUBLIC_INPUT: gate_public_input(a, w, t, g, unit); break;
            case GL355_GATE_ARITHMETIC: gate_arithmetic(a, w, t, g, gt.param); break;
            case GL355_GATE_ARITHMETIC_EXT: gate_arithmetic_ext(a, w, t, g, gt.param); break;
            case GL355_GATE_MUL_EXT: gate_mul_ext(a, w, t, g, gt.param); break;
            case GL355_GATE_POSEIDON_MDS: gate_poseidon_mds(a, w, t, g); break;
            case GL355_GATE_RANDOM_ACCESS: gate_random_access(a, w, t, g, gt.param); break;
            case GL355_GATE_REDUCING: gate_reducing<false>(a, w, t, g, gt.param); break;
            case GL355_GATE_REDUCING_EXT: gate_reducing<true>(a, w, t, g, gt.param); break;
            default: break;
        }
        const uint64_t sel = CONST(gt.selector_index);
        uint64_t filter = 1;
        for (uint32_t k = gt.group_start; k < gt.group_end; k++)
            if (k != gi) filter = gl_mul(filter, gl_sub(k, sel));
        if (n_sel > 1) filter = gl_mul(filter, gl_sub(0xFFFFFFFFull, sel));  // UNUSED_SELECTOR = u32::MAX
#pragma unroll
        for (uint32_t c = 0; c < nch; c++) gate_sum[c] = gl_add(gate_sum[c], gl_mul(filter, g.value(c)));
    }
    for (uint32_t c = 0; c < nch; c++) {
        const uint64_t v = gl_mul(gl_add(total.acc[c], gate_sum[c]), zh_inv);
        a.out[((uint64_t)unit * nch + c) * nq + t] = gl_canon(v);
    }
}

int32_t quotient_dev(Ctx* ctx, const gl355_circuit* c, const uint64_t* cs_lde, const uint64_t* wires_lde,
                     const uint64_t* zs_lde, uint64_t lde_stride, const uint64_t* k_is_dev, const uint64_t* betas,
                     const uint64_t* gammas, const uint64_t* alphas, const uint64_t pi_hash[4], uint64_t* out_values) {
    uint64_t b[4] = {0, 0, 0, 0}, g[4] = {0, 0, 0, 0}, al[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < c->num_challenges && i < 4; i++) { b[i] = betas[i]; g[i] = gammas[i]; al[i] = alphas[i]; }
    return quotient_units_dev(ctx, c, 1, cs_lde, wires_lde, 0, zs_lde, 0, lde_stride, k_is_dev, b, g, al, pi_hash, out_values);
}

// B lock-step units (blockIdx.y): betas / gammas / alphas / pi_hashes are [B][4]
int32_t quotient_units_dev(Ctx* ctx, const gl355_circuit* c, uint32_t B, const uint64_t* cs_lde, const uint64_t* wires_lde, uint64_t wires_us,
                           const uint64_t* zs_lde, uint64_t zs_us, uint64_t lde_stride, const uint64_t* k_is_dev, const uint64_t* betas,
                           const uint64_t* gammas, const uint64_t* alphas, const uint64_t* pi_hashes, uint64_t* out_values) {
    uint32_t qdb = 0;
    while ((1u << qdb) < c->max_degree) qdb++;
    if ((1u << qdb) != c->max_degree || qdb > c->rate_bits || qdb > 4) return ctx->fail(GL355_E_UNSUPPORTED, "quotient: degree factor must be a power of two <= 2^rate_bits");
    if (c->num_challenges == 0 || c->num_challenges > 4) return ctx->fail(GL355_E_UNSUPPORTED, "quotient: 1..4 challenges");
    if (c->num_gates > GL355_MAX_GATES) return ctx->fail(GL355_E_INVALID_ARG, "quotient: too many gates");
    if (B == 0 || B > GL355_MAX_UNITS) return ctx->fail(GL355_E_INVALID_ARG, "quotient: 1..GL355_MAX_UNITS units");
    QuotArgs a;
    memset(&a, 0, sizeof a);
    a.c = *c;
    a.cs = cs_lde; a.wires = wires_lde; a.zs = zs_lde; a.lde_stride = lde_stride; a.wires_us = wires_us; a.zs_us = zs_us;
    a.qbits = c->degree_bits + qdb;
    a.k_is = k_is_dev;
    GL355_TRY(ctx->pow_tables(gl_root_of_unity(a.qbits), &a.xw_lo, &a.xw_hi));
    // x^n = 7^n * omega_{2^qdb}^(i mod 2^qdb)
    const uint64_t sn = gl_pow(7, 1ull << c->degree_bits);
    const uint64_t wq = gl_root_of_unity(qdb);
    uint64_t w = 1;
    for (uint32_t k = 0; k < (1u << qdb); k++) {
        a.zh[k] = gl_canon(gl_sub(gl_mul(sn, w), 1));
        a.zh_inv[k] = gl_canon(gl_inv(a.zh[k]));
        w = gl_mul(w, wq);
    }
    AlphaTabArgs ta;
    memset(&ta, 0, sizeof ta);
    for (uint32_t u = 0; u < B; u++) {
        for (uint32_t i = 0; i < c->num_challenges; i++) {
            a.betas[u * 4 + i] = gl_canon(betas[u * 4 + i]); a.gammas[u * 4 + i] = gl_canon(gammas[u * 4 + i]);
            ta.alphas[u * 4 + i] = gl_canon(alphas[u * 4 + i]);
        }
        for (int i = 0; i < 4; i++) a.pi_hash[u * 4 + i] = gl_canon(pi_hashes[u * 4 + i]);
    }
    a.n_inv = gl_canon(gl_inv((1ull << c->degree_bits) % GL_P));
    a.out = out_values;
    // length of the constraint stream = the permutation-argument prefix + the longest gate (constraint counts as in
    // gates/*.rs num_constraints; the gate evaluators below push exactly that many terms)
    uint32_t longest = 0;
    for (uint32_t g = 0; g < c->num_gates; g++) {
        const uint32_t p = c->gates[g].param;
        uint32_t k = 0;
        switch (c->gates[g].type) {
            case GL355_GATE_CONSTANT: k = p; break;
            case GL355_GATE_PUBLIC_INPUT: k = 4; break;
            case GL355_GATE_BASE_SUM: k = 1 + p; break;
            case GL355_GATE_POSEIDON: k = 123; break;
            case GL355_GATE_ARITHMETIC: k = p; break;
            case GL355_GATE_ARITHMETIC_EXT: case GL355_GATE_MUL_EXT: case GL355_GATE_REDUCING: case GL355_GATE_REDUCING_EXT: k = 2 * p; break;
            case GL355_GATE_POSEIDON_MDS: k = 24; break;
            case GL355_GATE_RANDOM_ACCESS: k = ((p >> 8) & 0xFF) * ((p & 0xFF) + 2) + ((p >> 16) & 0xFF); break;
            default: break;
        }
        longest = k > longest ? k : longest;
    }
    const uint32_t n_chunks = (c->num_routed_wires + c->max_degree - 1) / c->max_degree;
    a.pw_stride = ((c->num_challenges * (1 + n_chunks) + longest + 1) + 63) & ~63u;
    Scratch pw(ctx);
    GL355_TRY(pw.get((size_t)B * c->num_challenges * a.pw_stride * 8));
    a.alpha_pw = pw.as<uint64_t>();
    ta.nch = c->num_challenges; ta.stride = a.pw_stride; ta.out = pw.as<uint64_t>();
    hipLaunchKernelGGL(alpha_table_kernel, dim3((a.pw_stride + 255) / 256, B), dim3(256), 0, ctx->stream, ta);
    GL355_HIP(ctx, hipGetLastError());
    const uint64_t nq = 1ull << a.qbits;
    // every column of the three oracles once per point of the quotient coset + the result
    ProfScope ps(ctx, "quotient_kernel", (uint64_t)B * nq * 8 * ((uint64_t)c->num_selectors + c->num_constants + c->num_routed_wires + c->num_wires +
                                                   (uint64_t)c->num_challenges * (2 + c->num_partial_products)));
    // (the LDS-staged variant quotient_kernel<NCH, true> measured slower -- profiles/r02_quotient_ab.txt -- and is no longer instantiated)
    const dim3 grid((uint32_t)((nq + 127) / 128), B);
    switch (c->num_challenges) {
        case 1: hipLaunchKernelGGL(quotient_kernel<1>, grid, dim3(128), 0, ctx->stream, a); break;
        case 2: hipLaunchKernelGGL(quotient_kernel<2>, grid, dim3(128), 0, ctx->stream, a); break;
        case 3: hipLaunchKernelGGL(quotient_kernel<3>, grid, dim3(128), 0, ctx->stream, a); break;
        default: hipLaunchKernelGGL(quotient_kernel<4>, grid, dim3(128), 0, ctx->stream, a); break;
    }
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

}  // namespace gl355
