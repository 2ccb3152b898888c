// The LDE passes of the commit path on CARRY-FREE 24-BIT LIMBS (round 3; a2 / a3 of SURVEY.md 8; conventions as in ntt.hip:
// omega_N = 7^((p-1)/N) chip/fri_chip.rs:162-163, bit-reversed evaluation order chip/fri_chip.rs:245-264).
//
// Why.  gfx950's SIMD-32 issues plain 32-bit add / sub / and / shift-right at twice the rate of everything that carries, multiplies
// or is 64 bits wide (tools/ubench/ubench_alu2.hip, profiles/r03_ubench_alu2.txt), and in Goldilocks 2^96 = -1.  A value held as four
// signed 32-bit limbs  v = l0 + l1 X + l2 X^2 + l3 X^3,  X = 2^24  (so X^4 = -1, arithmetic mod 2^96 + 1 = (2^32 + 1) p), with ~7 bits of
// headroom per limb, makes
//   * a radix-2 butterfly 4 + 4 full-rate instructions and NO carries (9 half-rate instructions on the 3 x 32-bit lazily reduced
//     form of round 2: measured 2.07x per butterfly),
//   * every twiddle inside a radix-8 network (omega_8 = -2^24, omega_4 = 2^48) a RENAMING of limbs (with signs folded into the
//     subtraction that produces them): free,
//   * the twiddles between the two radix-8 rounds of a 64-point transform (omega_64 = 2^39: shifts by multiples of 3 bits) a limb
//     rotation plus a bit shift that doubles as the carry normalisation: ~13 cheap instructions, no multiplication,
// so a 4096-point row is two radix-64 "super-rounds" with ONE general twiddle product per element between them (three in round 2),
// and the 32-point column pass has none inside.  Leaving the limb form (before a general product, and at the end) is
//   value = sum_i (l_i + beta_i) 2^(24 i) mod p  (beta = a multiple of p with all limbs ~1.5 * 2^28, so the operands are non-negative):
// multiply-adds into two 64-bit accumulators that cannot overflow and one 128-bit reduction.  (The same sum with tabulated words
// W_i = w X^i mod p is the product a w itself, l24_mul4 -- 8 multiply-adds, no conversion first; measured and not used: its 32 bytes
// of table per element made the passes bound by the CU's vector-memory path, profiles/r03_ubench_ntt_l24.txt.)
// Magnitudes: split limbs < 2^24; a radix-8 network multiplies the bound by 8 (2^27); the shift step renormalises to < 2^25; the second
// network gives < 2^28; l + beta < 2^29.4, four products < 2^61.4 each.  (static_asserts and the GPU parity tests hold this up.)
#pragma once
#include "gl355_internal.h"
#include "ntt_kernels.cuh"

// experiments (tools/ubench/ubench_ntt_l24.hip; results are wrong with a KO bit set): 1 = no global loads, 2 = no global stores,
// 4 = mid twiddles from constants instead of the table
#ifndef GL355_L24_KO
#define GL355_L24_KO 0
#endif
// 1: a persistent row block keeps its eight mid twiddles in registers for all its rows (128 VGPRs + scratch); 0: re-read per row
#ifndef GL355_L24_TW_REGS
#define GL355_L24_TW_REGS 0
#endif

namespace gl355 {

struct L24 { int32_t l[4]; };
constexpr uint32_t L24_MASK = 0xFFFFFFu;
// beta: sum_i beta_i 2^(24 i) = 0 mod p, every limb within [2^28, 2^29) (LLL + nearest plane on the lattice of multiples of p in limb
// form, tools/l24_bias.py; checked there and by tests/test_l24_model.py)
constexpr uint32_t L24_BETA[4] = {402653208u, 402653160u, 402653160u, 402653160u};

GL_DEV L24 l24_split(uint64_t x) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    L24 r;
    r.l[0] = (int32_t)(lo & L24_MASK);
    r.l[1] = (int32_t)(__builtin_amdgcn_alignbit(hi, lo, 24) & L24_MASK);
    r.l[2] = (int32_t)(hi >> 16);
    r.l[3] = 0;
    return r;
}

// (a, b) <- (a + b, +-(a - b) * X^RHO): the limb rotation and both signs are folded into which operand each subtraction takes from
template <int RHO, bool NEG>
GL_DEV void l24_bfly(L24& a, L24& b) {
    L24 s, d;
#pragma unroll
    for (int i = 0; i < 4; i++) s.l[i] = a.l[i] + b.l[i];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int j = (i - RHO) & 3;
        const bool wrap = i < RHO;                      // X^4 = -1
        d.l[i] = (NEG != wrap) ? b.l[j] - a.l[j] : a.l[j] - b.l[j];
    }
    a = s; b = d;
}
// twiddle omega_16^(+-E), E even, as (limb rotation, sign): forward -2^24, 2^48, -2^72; inverse 2^72, -2^48, 2^24 (ntt_kernels.cuh l96_bfly)
template <bool INV, int E>
GL_DEV void l24_bfly_w(L24& a, L24& b) {
    constexpr int FWD_R[4] = {0, 1, 2, 3};
    constexpr bool FWD_NEG[4] = {false, true, false, true};
    constexpr int INV_R[4] = {0, 3, 2, 1};
    constexpr bool INV_NEG[4] = {false, false, true, false};
    l24_bfly<INV ? INV_R[E / 2] : FWD_R[E / 2], INV ? INV_NEG[E / 2] : FWD_NEG[E / 2]>(a, b);
}
// radix-8 DIF network, y[pos] <- Y[bitrev(pos)] (the data flow of dif8_lazy)
template <bool INV>
GL_DEV void dif8_l24(L24 (&y)[8]) {
    l24_bfly_w<INV, 0>(y[0], y[4]); l24_bfly_w<INV, 2>(y[1], y[5]); l24_bfly_w<INV, 4>(y[2], y[6]); l24_bfly_w<INV, 6>(y[3], y[7]);
    l24_bfly_w<INV, 0>(y[0], y[2]); l24_bfly_w<INV, 4>(y[1], y[3]); l24_bfly_w<INV, 0>(y[4], y[6]); l24_bfly_w<INV, 4>(y[5], y[7]);
    l24_bfly_w<INV, 0>(y[0], y[1]); l24_bfly_w<INV, 0>(y[2], y[3]); l24_bfly_w<INV, 0>(y[4], y[5]); l24_bfly_w<INV, 0>(y[6], y[7]);
}
// radix-4 DIF network on y[0..3]
template <bool INV>
GL_DEV void dif4_l24(L24 (&y)[4]) {
    l24_bfly_w<INV, 0>(y[0], y[2]); l24_bfly_w<INV, 4>(y[1], y[3]);
    l24_bfly_w<INV, 0>(y[0], y[1]); l24_bfly_w<INV, 0>(y[2], y[3]);
}

// e * 2^S for a compile-time 0 <= S < 192, renormalised: S = 96 s + 24 a + b.  Limb i splits at bit 24 - b into g (kept, shifted up by
// b) and h (carried into limb i + 1; out of limb 3 it wraps to limb 0 negated), then the limbs rotate by a; |out| < 2^24 + |e| 2^(b - 24).
// With S = 0 it is the plain carry normalisation.
template <int S>
GL_DEV L24 l24_shift(const L24& e) {
    static_assert(S >= 0 && S < 192, "shift out of range");
    constexpr bool SG = S >= 96;
    constexpr int A = (S % 96) / 24, B = (S % 96) % 24;
    int32_t g[4], h[4], y[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { g[i] = e.l[i] & (int32_t)((1u << (24 - B)) - 1); h[i] = e.l[i] >> (24 - B); }
    y[0] = (g[0] << B) - h[3];
#pragma unroll
    for (int i = 1; i < 4; i++) y[i] = (g[i] << B) + h[i - 1];
    L24 r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const bool neg = SG != (i + A >= 4);
        r.l[(i + A) & 3] = neg ? -y[i] : y[i];
    }
    return r;
}
// the twiddles after the FIRST radix-8 round of a 2^M-point transform (M = 6: omega_64 = 2^39; M = 5: omega_32 = 2^78): slot q holds
// output k0 = bitrev3(q) of butterfly R, multiplied by omega^(R k0); every slot is renormalised (slot 0 / R = 0: shift 0)
template <int M, int R, bool INV>
GL_DEV void l24_twiddles(L24 (&y)[8]) {
    constexpr int MULT = M == 6 ? 39 : 78;
#define GL355_L24_TW(Q, K0) { constexpr int S0 = (MULT * R * K0) % 192; y[Q] = l24_shift<INV ? (192 - S0) % 192 : S0>(y[Q]); }
    // (scheduling fences every two elements: with all eight in flight the g / h / y temporaries alone are ~100 registers)
    GL355_L24_TW(0, 0) GL355_L24_TW(1, 4) __builtin_amdgcn_sched_barrier(0);
    GL355_L24_TW(2, 2) GL355_L24_TW(3, 6) __builtin_amdgcn_sched_barrier(0);
    GL355_L24_TW(4, 1) GL355_L24_TW(5, 5) __builtin_amdgcn_sched_barrier(0);
    GL355_L24_TW(6, 3) GL355_L24_TW(7, 7)
#undef GL355_L24_TW
}
// r is wave-uniform: one scalar branch, then straight-line code with compile-time shifts and limb renamings
template <int M, bool INV>
GL_DEV void l24_twiddles_r(L24 (&y)[8], uint32_t r) {
    switch (r) {
        case 0: l24_twiddles<M, 0, INV>(y); break;
        case 1: l24_twiddles<M, 1, INV>(y); break;
        case 2: l24_twiddles<M, 2, INV>(y); break;
        case 3: l24_twiddles<M, 3, INV>(y); break;
        case 4: if constexpr (M == 6) l24_twiddles<M, 4, INV>(y); break;
        case 5: if constexpr (M == 6) l24_twiddles<M, 5, INV>(y); break;
        case 6: if constexpr (M == 6) l24_twiddles<M, 6, INV>(y); break;
        default: if constexpr (M == 6) l24_twiddles<M, 7, INV>(y); break;
    }
}

// a * w mod p (any u64 representative) for limbs |l_i| < 2^28 and w given as the four words W_i = w X^i mod p
GL_DEV uint64_t l24_mul4(const L24& a, uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3) {
    const uint32_t b0 = (uint32_t)a.l[0] + L24_BETA[0], b1 = (uint32_t)a.l[1] + L24_BETA[1], b2 = (uint32_t)a.l[2] + L24_BETA[2],
                   b3 = (uint32_t)a.l[3] + L24_BETA[3];
    uint64_t slo = (uint64_t)b0 * (uint32_t)w0;
    slo += (uint64_t)b1 * (uint32_t)w1; slo += (uint64_t)b2 * (uint32_t)w2; slo += (uint64_t)b3 * (uint32_t)w3;
    uint64_t shi = (uint64_t)b0 * (uint32_t)(w0 >> 32);
    shi += (uint64_t)b1 * (uint32_t)(w1 >> 32); shi += (uint64_t)b2 * (uint32_t)(w2 >> 32); shi += (uint64_t)b3 * (uint32_t)(w3 >> 32);
    // slo + shi 2^32 as (lo, hi): only the middle word adds
    uint32_t mid;
    const bool c = __builtin_uadd_overflow((uint32_t)(slo >> 32), (uint32_t)shi, &mid);
    const uint64_t lo = ((uint64_t)mid << 32) | (uint32_t)slo, hi = (shi >> 32) + (c ? 1u : 0u);
    return gl_reduce128(lo, hi);
}
GL_DEV uint64_t l24_mul4(const L24& a, const uint64_t* __restrict__ w) {
    const ulonglong2 p0 = *reinterpret_cast<const ulonglong2*>(w), p1 = *reinterpret_cast<const ulonglong2*>(w + 2);
    return l24_mul4(a, p0.x, p0.y, p1.x, p1.y);
}
// the limbs as a u64 (any representative): the same sum with W_i = 2^(24 i) mod p (2^72 = 2^40 - 2^8); the compiler folds the zero words
GL_DEV uint64_t l24_value(const L24& a) { return l24_mul4(a, 1ull, 1ull << 24, 1ull << 48, (1ull << 40) - (1ull << 8)); }

// ------------------------------------------------------------------------------------------------------------------------------
// Row pass: 4096-point rows, 512-thread blocks = two radix-64 super-rounds on a 64 x 64 view (index = 64 u + v): A over u (stride 64),
// the general twiddle omega_4096^(v kA) (a.mid: one word per cell), B over v.  Wave w is butterfly r = w of every first round, so the
// shift twiddles are compile-time per branch.  The tile lives in LDS as 16-byte limb quads at index + (index >> 6) (row stride 65
// quads: the B rounds walk a lane stride of 65 x 16 bytes, conflict-free per 16-lane group); the two exchanges that carry reduced
// 8-byte values (between the super-rounds, and the transposition to store order) use the low half of a thread's OWN cells, so no
// barrier is needed before writing them.  Blocks are PERSISTENT (grid = 2 per CU): a block's eight mid twiddles per thread stay in
// registers for all its rows -- fetched per row they were 128 KB of L2 reads per 64 KB of data and, through the CU's 64-B/clk vector
// memory path, a fifth of the pass (profiles/r03_ubench_ntt_l24.txt) -- and the next row's elements are fetched while this one is
// transformed.  Output order: plain bit reversal (identical to ntt_rows_r8_kernel<12>), canonical.
// ------------------------------------------------------------------------------------------------------------------------------
GL_DEV uint32_t l24_phys(uint32_t idx) { return idx + (idx >> 6); }
// A 4096-element row through buffer instructions: resource descriptor and the 4096 q byte offsets in scalar registers, ONE vector register
// (8 tid) of address per thread -- the flat-address form keeps eight 64-bit pointers per direction alive (32 VGPRs the persistent loop does not have)
typedef uint32_t l24_u32x2 __attribute__((ext_vector_type(2)));
GL_DEV __amdgpu_buffer_rsrc_t l24_row_rsrc(const uint64_t* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 4096 * 8, 0x00020000); }
GL_DEV uint64_t l24_row_load(__amdgpu_buffer_rsrc_t r, uint32_t tid8, int q) {
    const l24_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)tid8, q * 4096, 0);
    return ((uint64_t)v.y << 32) | v.x;
}
GL_DEV void l24_row_store(__amdgpu_buffer_rsrc_t r, uint32_t tid8, int q, uint64_t x) {
    l24_u32x2 v; v.x = (uint32_t)x; v.y = (uint32_t)(x >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)tid8, q * 4096, 0);
}
constexpr size_t L24_ROWS_LDS_BYTES = (4096 + 64) * 16;

template <int WPE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE))) ntt_rows_l24_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_raw[];
    int4* lq = reinterpret_cast<int4*>(lds_raw);
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const uint64_t total_rows = ((uint64_t)a.batch) << a.log_rows;
    // cell addresses written out as base + compile-time offset (the padding idx + (idx >> 6) is linear inside each access pattern), so the
    // compiler issues ds instructions with immediate offsets from five base registers instead of keeping ~40 precomputed addresses alive
    auto put = [&](uint32_t cell, const L24& v) { lq[cell] = make_int4(v.l[0], v.l[1], v.l[2], v.l[3]); };
    auto get = [&](uint32_t cell) { const int4 q = lq[cell]; L24 v; v.l[0] = q.x; v.l[1] = q.y; v.l[2] = q.z; v.l[3] = q.w; return v; };
    auto put8 = [&](uint32_t cell, uint64_t v) { lds_raw[2 * cell] = v; };      // the low 8 bytes of a quad cell
    auto get8 = [&](uint32_t cell) { return lds_raw[2 * cell]; };
    const uint32_t cA1 = 65 * w + lane;          // idx = 64 (8 q + w) + lane      -> cell cA1 + 520 q
    const uint32_t cA2 = 520 * w + lane;         // idx = 64 (8 w + r) + lane      -> cell cA2 + 65 r
    const uint32_t cB1 = 65 * lane + w;          // idx = 64 lane + 8 q + w        -> cell cB1 + 8 q
    const uint32_t cB2 = 65 * lane + 8 * w;      // idx = 64 lane + 8 w + r        -> cell cB2 + r
    const uint32_t cST = tid + w;                // idx = tid + 512 q              -> cell cST + 520 q
    auto row_ptr = [&](uint64_t row, const uint64_t* base, uint64_t stride) {
        const uint64_t col = row >> a.log_rows, rin = row & ((1ull << a.log_rows) - 1);
        return base + col * stride + (rin << 12);
    };
#if GL355_L24_TW_REGS
    uint64_t tw[8];                                         // cell (8 w + s, lane) of the mid table, s < 8: the same for every row
#pragma unroll
    for (int s = 0; s < 8; s++) tw[s] = (GL355_L24_KO & 4) ? 3 + s + w : a.mid[64 * (8 * w + s) + lane];
#endif
    uint64_t row = blockIdx.x;
    const uint32_t tid8 = tid * 8;
    uint64_t x[8];
    if (row < total_rows) {
        const __amdgpu_buffer_rsrc_t rin = l24_row_rsrc(row_ptr(row, a.in, a.in_col_stride));
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = (GL355_L24_KO & 1) ? (uint64_t)tid * 0x9E3779B97F4A7C15ull + q : l24_row_load(rin, tid8, q);
    }
    while (row < total_rows) {
        L24 y[8];
        // A1: the thread that loaded elements tid + 512 q holds u = 8 q + w, v = lane: its own first-round butterfly (r = w)
#pragma unroll
        for (int q = 0; q < 8; q++) y[q] = l24_split(x[q]);
        const uint64_t next = row + gridDim.x;
        if (next < total_rows) {
            const __amdgpu_buffer_rsrc_t rin = l24_row_rsrc(row_ptr(next, a.in, a.in_col_stride));
#pragma unroll
            for (int q = 0; q < 8; q++) x[q] = (GL355_L24_KO & 1) ? x[q] + next : l24_row_load(rin, tid8, q);
        }
        dif8_l24<false>(y);
        l24_twiddles_r<6, false>(y, w);
#pragma unroll
        for (int q = 0; q < 8; q++) put(cA1 + 520 * q, y[q]);
#if !GL355_L24_TW_REGS
        uint64_t tw[8];                                     // fetched per row (32 KB per tile, L2-resident), issued before the barrier: held
#pragma unroll                                              // across rows they cost 16 VGPRs and pushed the kernel into scratch
        for (int s = 0; s < 8; s++) tw[s] = (GL355_L24_KO & 4) ? 3 + s + w : a.mid[64 * (8 * w + s) + lane];
#endif
        __syncthreads();
        // A2: u = 8 w + r over r, then the general twiddle of the 64 x 64 split; 8-byte products into the thread's own cells
#pragma unroll
        for (int r = 0; r < 8; r++) y[r] = get(cA2 + 65 * r);
        dif8_l24<false>(y);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            put8(cA2 + 65 * s, gl_mul(l24_value(y[s]), tw[s]));
            if (s & 1) __builtin_amdgcn_sched_barrier(0);   // two elements in flight, not eight: their temporaries would not fit 128 VGPRs
        }
        __syncthreads();
        // B1: u-slot = lane, v = 8 q + w; reads and writes the same eight cells
#pragma unroll
        for (int q = 0; q < 8; q++) y[q] = l24_split(get8(cB1 + 8 * q));
        dif8_l24<false>(y);
        l24_twiddles_r<6, false>(y, w);
#pragma unroll
        for (int q = 0; q < 8; q++) put(cB1 + 8 * q, y[q]);
        __syncthreads();
        // B2: v = 8 w + r over r; results leave the limb form (own cells again), then the transposition to store order
#pragma unroll
        for (int r = 0; r < 8; r++) y[r] = get(cB2 + r);
        dif8_l24<false>(y);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            put8(cB2 + s, gl_canon(l24_value(y[s])));
            if (s & 1) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        const __amdgpu_buffer_rsrc_t rout = l24_row_rsrc(row_ptr(row, a.out, a.out_col_stride));
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint64_t v = get8(cST + 520 * q);
            if (!(GL355_L24_KO & 2) || v == 0x123456789ull) l24_row_store(rout, tid8, q, v);
        }
        __syncthreads();                                    // the tile is free for the next row's quads
        row = next;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same row pass with the limb quads exchanged in TWO SUB-STEPS (limbs 0, 1 of every element, then limbs 2, 3) through 8-byte
// cells: the tile is 33 KB of LDS instead of 65, so a CU holds 3-4 tiles = 6-8 waves per SIMD instead of 4 -- the limb kernels were
// never short of issue slots, they were short of waves to cover their global loads and LDS round trips.  Same cells,
// same padding, same output as ntt_rows_l24_kernel; 8 barriers per row instead of 5.  A hi-pair sub-step is read only by the thread
// that later overwrites those cells, so the 8-byte exchanges still start without a barrier.  grid = rows: one row per block; a
// smaller grid walks rows blockIdx, blockIdx + grid, ... (PREFETCH: the next row's elements are fetched under this row's transform).
// ------------------------------------------------------------------------------------------------------------------------------
constexpr size_t L24S_ROWS_LDS_BYTES = (4096 + 64) * 8;
template <int WPE, bool PREFETCH>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE))) ntt_rows_l24s_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_raw[];
    int2* lp = reinterpret_cast<int2*>(lds_raw);
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const uint64_t total_rows = ((uint64_t)a.batch) << a.log_rows;
    const uint32_t cA1 = 65 * w + lane;          // idx = 64 (8 q + w) + lane      -> cell cA1 + 520 q
    const uint32_t cA2 = 520 * w + lane;         // idx = 64 (8 w + r) + lane      -> cell cA2 + 65 r
    const uint32_t cB1 = 65 * lane + w;          // idx = 64 lane + 8 q + w        -> cell cB1 + 8 q
    const uint32_t cB2 = 65 * lane + 8 * w;      // idx = 64 lane + 8 w + r        -> cell cB2 + r
    const uint32_t cST = tid + w;                // idx = tid + 512 q              -> cell cST + 520 q
    auto row_ptr = [&](uint64_t row, const uint64_t* base, uint64_t stride) {
        const uint64_t col = row >> a.log_rows, rin = row & ((1ull << a.log_rows) - 1);
        return base + col * stride + (rin << 12);
    };
    uint64_t row = blockIdx.x;
    const uint32_t tid8 = tid * 8;
    uint64_t x[8];
    if (PREFETCH && row < total_rows) {
        const __amdgpu_buffer_rsrc_t rin = l24_row_rsrc(row_ptr(row, a.in, a.in_col_stride));
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = l24_row_load(rin, tid8, q);
    }
    while (row < total_rows) {
        L24 y[8];
        const uint64_t next = row + gridDim.x;
        if (!PREFETCH) {
            const __amdgpu_buffer_rsrc_t rin = l24_row_rsrc(row_ptr(row, a.in, a.in_col_stride));
#pragma unroll
            for (int q = 0; q < 8; q++) x[q] = (GL355_L24_KO & 1) ? (uint64_t)tid * 0x9E3779B97F4A7C15ull + q : l24_row_load(rin, tid8, q);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) y[q] = l24_split(x[q]);
        if (PREFETCH && next < total_rows) {
            const __amdgpu_buffer_rsrc_t rin = l24_row_rsrc(row_ptr(next, a.in, a.in_col_stride));
#pragma unroll
            for (int q = 0; q < 8; q++) x[q] = l24_row_load(rin, tid8, q);
        }
        // A1 (own butterfly r = w of the loaded elements), exchange to A2 in two sub-steps
        dif8_l24<false>(y);
        l24_twiddles_r<6, false>(y, w);
#pragma unroll
        for (int q = 0; q < 8; q++) lp[cA1 + 520 * q] = make_int2(y[q].l[0], y[q].l[1]);
        uint64_t tw[8];
#pragma unroll
        for (int s = 0; s < 8; s++) tw[s] = (GL355_L24_KO & 4) ? 3 + s + w : a.mid[64 * (8 * w + s) + lane];
        __syncthreads();
        L24 z[8];
#pragma unroll
        for (int r = 0; r < 8; r++) { const int2 t = lp[cA2 + 65 * r]; z[r].l[0] = t.x; z[r].l[1] = t.y; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; q++) lp[cA1 + 520 * q] = make_int2(y[q].l[2], y[q].l[3]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; r++) { const int2 t = lp[cA2 + 65 * r]; z[r].l[2] = t.x; z[r].l[3] = t.y; }
        // A2, the general twiddle; 8-byte products into the cells whose hi pairs only this thread has read
        dif8_l24<false>(z);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            lds_raw[cA2 + 65 * s] = gl_mul(l24_value(z[s]), tw[s]);
            if (s & 1) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // B1 reads and writes the same eight cells
#pragma unroll
        for (int q = 0; q < 8; q++) y[q] = l24_split(lds_raw[cB1 + 8 * q]);
        dif8_l24<false>(y);
        l24_twiddles_r<6, false>(y, w);
#pragma unroll
        for (int q = 0; q < 8; q++) lp[cB1 + 8 * q] = make_int2(y[q].l[0], y[q].l[1]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; r++) { const int2 t = lp[cB2 + r]; z[r].l[0] = t.x; z[r].l[1] = t.y; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; q++) lp[cB1 + 8 * q] = make_int2(y[q].l[2], y[q].l[3]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; r++) { const int2 t = lp[cB2 + r]; z[r].l[2] = t.x; z[r].l[3] = t.y; }
        // B2: results leave the limb form (own cells), then the transposition to store order
        dif8_l24<false>(z);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            lds_raw[cB2 + s] = gl_canon(l24_value(z[s]));
            if (s & 1) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        const __amdgpu_buffer_rsrc_t rout = l24_row_rsrc(row_ptr(row, a.out, a.out_col_stride));
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint64_t v = lds_raw[cST + 520 * q];
            if (!(GL355_L24_KO & 2) || v == 0x123456789ull) l24_row_store(rout, tid8, q, v);
        }
        row = next;
        if (row < total_rows) __syncthreads();               // the tile is free for the next row
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 6 experiment: the limb-QUAD row pass (16-byte cells, 65-KB tile, 5 barriers per row) as ONE ROW PER BLOCK -- the quad kernel above is
// persistent with a register-staged prefetch (128 VGPRs); this is its arithmetic and exchange pattern with the launch shape of the shipped
// split-exchange kernel (fresh blocks, no prefetch): two blocks per CU either way (LDS here, registers there), 3 barriers fewer per row.
// ------------------------------------------------------------------------------------------------------------------------------
template <int WPE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE))) ntt_rows_l24q_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_raw[];
    int4* lq = reinterpret_cast<int4*>(lds_raw);
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    auto put = [&](uint32_t cell, const L24& v) { lq[cell] = make_int4(v.l[0], v.l[1], v.l[2], v.l[3]); };
    auto get = [&](uint32_t cell) { const int4 q = lq[cell]; L24 v; v.l[0] = q.x; v.l[1] = q.y; v.l[2] = q.z; v.l[3] = q.w; return v; };
    auto put8 = [&](uint32_t cell, uint64_t v) { lds_raw[2 * cell] = v; };
    auto get8 = [&](uint32_t cell) { return lds_raw[2 * cell]; };
    const uint32_t cA1 = 65 * w + lane, cA2 = 520 * w + lane, cB1 = 65 * lane + w, cB2 = 65 * lane + 8 * w, cST = tid + w;
    const uint64_t row = blockIdx.x;
    const uint64_t col = row >> a.log_rows, rin = row & ((1ull << a.log_rows) - 1);
    const uint32_t tid8 = tid * 8;
    const __amdgpu_buffer_rsrc_t rs_in = l24_row_rsrc(a.in + col * a.in_col_stride + (rin << 12));
    L24 y[8];
#pragma unroll
    for (int q = 0; q < 8; q++) y[q] = l24_split(l24_row_load(rs_in, tid8, q));
    dif8_l24<false>(y);
    l24_twiddles_r<6, false>(y, w);
#pragma unroll
    for (int q = 0; q < 8; q++) put(cA1 + 520 * q, y[q]);
    uint64_t tw[8];
#pragma unroll
    for (int s = 0; s < 8; s++) tw[s] = a.mid[64 * (8 * w + s) + lane];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) y[r] = get(cA2 + 65 * r);
    dif8_l24<false>(y);
#pragma unroll
    for (int s = 0; s < 8; s++) {
        put8(cA2 + 65 * s, gl_mul(l24_value(y[s]), tw[s]));
        if (s & 1) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; q++) y[q] = l24_split(get8(cB1 + 8 * q));
    dif8_l24<false>(y);
    l24_twiddles_r<6, false>(y, w);
#pragma unroll
    for (int q = 0; q < 8; q++) put(cB1 + 8 * q, y[q]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; r++) y[r] = get(cB2 + r);
    dif8_l24<false>(y);
#pragma unroll
    for (int s = 0; s < 8; s++) {
        put8(cB2 + s, gl_canon(l24_value(y[s])));
        if (s & 1) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs_out = l24_row_rsrc(a.out + col * a.out_col_stride + (rin << 12));
#pragma unroll
    for (int q = 0; q < 8; q++) l24_row_store(rs_out, tid8, q, get8(cST + 520 * q));
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 6 experiment: the split-exchange row pass with the NEXT row fetched by LDS-DMA (global_load_lds_dwordx4: global -> LDS without passing
// through registers) while this row is transformed.  The register-staged prefetch of round 3 / 5 cost 16 VGPRs and with them the kernel's occupancy
// (profiles/r05_ubench_ntt_l24s.txt: 0.66-0.70 ms against 0.56); a DMA costs none, only 32 KB of LDS for the raw row next to the 33-KB tile
// (2 blocks per CU either way).  Persistent blocks walk rows blockIdx, blockIdx + grid, ...  Every barrier is a raw s_barrier behind lgkmcnt(0):
// __syncthreads() would drain the DMA in flight (its fence waits vmcnt(0)).  One vector-memory counter orders everything: at the top of an
// iteration the row's four DMA pieces per wave are older than the previous row's eight stores, so vmcnt(8) retires exactly them; the barrier that
// follows makes the other waves' pieces visible.  The DMA of row r + 1 is issued behind the first barrier of row r, when every thread has read
// its elements of row r out of the raw buffer.  Same cells, same arithmetic, same output as ntt_rows_l24s_kernel.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr size_t L24D_ROWS_LDS_BYTES = 0;      // static LDS: (4096 + 64) * 8 + 4096 * 8 = 66 048 B
#define GL355_L24D_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int WPE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE))) ntt_rows_l24d_kernel(PassArgs a) {
    // two DISTINCT LDS objects: the compiler orders a ds access behind an LDS-DMA in flight unless it can prove they do not alias, and inside one
    // dynamic array it cannot (it then waits vmcnt(0) in front of every tile access, i.e. right behind the DMA's issue)
    __shared__ __attribute__((aligned(16))) uint64_t lds_raw[4096 + 64];
    __shared__ __attribute__((aligned(16))) uint64_t raw[4096];
    int2* lp = reinterpret_cast<int2*>(lds_raw);
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const uint64_t total_rows = ((uint64_t)a.batch) << a.log_rows;
    const uint32_t cA1 = 65 * w + lane, cA2 = 520 * w + lane, cB1 = 65 * lane + w, cB2 = 65 * lane + 8 * w, cST = tid + w;
    auto row_ptr = [&](uint64_t row, const uint64_t* base, uint64_t stride) {
        const uint64_t col = row >> a.log_rows, rin = row & ((1ull << a.log_rows) - 1);
        return base + col * stride + (rin << 12);
    };
    typedef __attribute__((address_space(3))) void* lds_vptr;
    typedef const __attribute__((address_space(1))) void* glb_vptr;
    // wave w fetches elements [512 w, 512 w + 512) of the row as four 1-KB pieces: lane l of piece j brings elements 512 w + 128 j + 2 l, + 1
    auto dma_row = [&](const uint64_t* rowp) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t e0 = 512 * w + 128 * j;
            __builtin_amdgcn_global_load_lds((glb_vptr)(rowp + e0 + 2 * lane), (lds_vptr)(raw + e0), 16, 0, 0);
        }
    };
    uint64_t row = blockIdx.x;
    const uint32_t tid8 = tid * 8;
    const uint32_t raw_addr = (uint32_t)(uintptr_t)(lds_vptr)raw + tid8;      // LDS byte address of this thread's first element
    // the eight mid twiddles of a thread are the same for every row: held in registers by the persistent block (the LDS limit of two blocks per CU
    // leaves 128 VGPRs per lane; a load inside the loop would be waited for with vmcnt(0) and drain the DMA in flight)
    uint64_t tw[8];
#pragma unroll
    for (int s = 0; s < 8; s++) tw[s] = a.mid[64 * (8 * w + s) + lane];
    if (row < total_rows) dma_row(row_ptr(row, a.in, a.in_col_stride));
    bool first = true;
    while (row < total_rows) {
        const uint64_t next = row + gridDim.x;
        // this row's DMA pieces have landed (they are older than the previous row's 8 stores), then everybody's
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        first = false;
        GL355_L24D_BARRIER();
        L24 y[8], z[8];
        {
            // the raw row through ds_read_b64 written out: as C loads the compiler orders them behind "the DMA that may still be in flight" with a
            // vmcnt(0) of its own, which would also wait for the previous row's stores
            uint64_t x[8];
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:4096\n\tds_read_b64 %2, %8 offset:8192\n\tds_read_b64 %3, %8 offset:12288\n\t"
                         "ds_read_b64 %4, %8 offset:16384\n\tds_read_b64 %5, %8 offset:20480\n\tds_read_b64 %6, %8 offset:24576\n\tds_read_b64 %7, %8 offset:28672\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]) : "v"(raw_addr) : "memory");
#pragma unroll
            for (int q = 0; q < 8; q++) y[q] = l24_split(x[q]);
        }
        dif8_l24<false>(y);
        l24_twiddles_r<6, false>(y, w);
#pragma unroll
        for (int q = 0; q < 8; q++) lp[cA1 + 520 * q] = make_int2(y[q].l[0], y[q].l[1]);
        GL355_L24D_BARRIER();                                  // the raw row has been consumed by every thread
        if (next < total_rows) dma_row(row_ptr(next, a.in, a.in_col_stride));
#pragma unroll
        for (int r = 0; r < 8; r++) { const int2 t = lp[cA2 + 65 * r]; z[r].l[0] = t.x; z[r].l[1] = t.y; }
        GL355_L24D_BARRIER();
#pragma unroll
        for (int q = 0; q < 8; q++) lp[cA1 + 520 * q] = make_int2(y[q].l[2], y[q].l[3]);
        GL355_L24D_BARRIER();
#pragma unroll
        for (int r = 0; r < 8; r++) { const int2 t = lp[cA2 + 65 * r]; z[r].l[2] = t.x; z[r].l[3] = t.y; }
        dif8_l24<false>(z);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            lds_raw[cA2 + 65 * s] = gl_mul(l24_value(z[s]), tw[s]);
            if (s & 1) __builtin_amdgcn_sched_barrier(0);
        }
        GL355_L24D_BARRIER();
#pragma unroll
        for (int q = 0; q < 8; q++) y[q] = l24_split(lds_raw[cB1 + 8 * q]);
        dif8_l24<false>(y);
        l24_twiddles_r<6, false>(y, w);
#pragma unroll
        for (int q = 0; q < 8; q++) lp[cB1 + 8 * q] = make_int2(y[q].l[0], y[q].l[1]);
        GL355_L24D_BARRIER();
#pragma unroll
        for (int r = 0; r < 8; r++) { const int2 t = lp[cB2 + r]; z[r].l[0] = t.x; z[r].l[1] = t.y; }
        GL355_L24D_BARRIER();
#pragma unroll
        for (int q = 0; q < 8; q++) lp[cB1 + 8 * q] = make_int2(y[q].l[2], y[q].l[3]);
        GL355_L24D_BARRIER();
#pragma unroll
        for (int r = 0; r < 8; r++) { const int2 t = lp[cB2 + r]; z[r].l[2] = t.x; z[r].l[3] = t.y; }
        dif8_l24<false>(z);
#pragma unroll
        for (int s = 0; s < 8; s++) {
            lds_raw[cB2 + s] = gl_canon(l24_value(z[s]));
            if (s & 1) __builtin_amdgcn_sched_barrier(0);
        }
        GL355_L24D_BARRIER();
        const __amdgpu_buffer_rsrc_t rout = l24_row_rsrc(row_ptr(row, a.out, a.out_col_stride));
#pragma unroll
        for (int q = 0; q < 8; q++) l24_row_store(rout, tid8, q, lds_raw[cST + 520 * q]);
        row = next;
        // (the next iteration's first barrier also frees the tile: every thread has read its store-order cells before it gets there)
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Column pass of the LDE over all cosets (the shape of ntt_cols_r8_cosets_kernel<5>): 32 rows x 128 columns per tile, 32 = 8 x 4 with
// omega_32 = 2^78 shift twiddles between the radix-8 and the radix-4 round, the 4-step twiddle (a.step_full) at the store.  Threads
// tid >> 7 = r are wave-uniform.  A thread stores to the same eight places with the same step twiddles for every coset: they are loaded
// once per tile.  blockIdx is mapped so that an XCD
// (blockIdx % 8) only ever touches 4 of the 32 column tiles: its L2 holds those slices of the step / pre / ratio tables.
// ------------------------------------------------------------------------------------------------------------------------------
constexpr size_t L24_COLS_LDS_BYTES = 4096 * 16;
template <int WPE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE))) ntt_cols_l24_cosets_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_raw[];
    int4* lq = reinterpret_cast<int4*>(lds_raw);
    constexpr uint32_t LOG_TC = 7, TC = 128;
    const uint32_t tid = threadIdx.x, r = tid >> 7, cc = tid & (TC - 1);
    const uint32_t log_n2 = a.log_rows;                     // 12
    const uint32_t tiles_per_col = (1u << log_n2) >> LOG_TC; // 32
    uint32_t tile, colu;
    if (tiles_per_col == 32) { tile = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) & 3); colu = blockIdx.x >> 5; }
    else { tile = blockIdx.x % tiles_per_col; colu = blockIdx.x / tiles_per_col; }
    const uint64_t col = colu, c0 = (uint64_t)tile << LOG_TC;
    const uint64_t* in = a.in + col * a.in_col_stride;
    uint64_t v[8], step[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint64_t gi = ((uint64_t)(r + 4 * q) << log_n2) + c0 + cc;
        v[q] = gl_mul((GL355_L24_KO & 1) ? gi : in[gi], a.pre_full[gi]);
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; t2++)
#pragma unroll
        for (int k = 0; k < 4; k++) step[4 * t2 + k] = a.step_full[((uint64_t)(4 * (r + 4 * t2) + k) << log_n2) + c0 + cc];
    const CosetSlots slots = coset_slots_of(a);
    for (uint32_t c = 0; c < a.n_cosets; c++) {
        if (c) {        // the ratio table is re-read per coset (L2-resident, coalesced) rather than held: 16 VGPRs less, no spills
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = gl_mul(v[q], a.ratio_full[((uint64_t)(r + 4 * q) << log_n2) + c0 + cc]);
        }
        uint64_t* out = a.out + (uint64_t)coset_slot_at(slots, c) * a.coset_out_stride + col * a.out_col_stride;
        L24 y[8];
#pragma unroll
        for (int q = 0; q < 8; q++) y[q] = l24_split(v[q]);
        dif8_l24<false>(y);
        l24_twiddles_r<5, false>(y, r);
#pragma unroll
        for (int q = 0; q < 8; q++) lq[TC * (4 * q + r) + cc] = make_int4(y[q].l[0], y[q].l[1], y[q].l[2], y[q].l[3]);
        __syncthreads();
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) {                    // two radix-4 tasks: rows 4 q' + {0..3}, q' = r and r + 4
            const uint32_t qp = r + 4 * t2;
            L24 z[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const int4 qd = lq[TC * (4 * qp + k) + cc]; z[k].l[0] = qd.x; z[k].l[1] = qd.y; z[k].l[2] = qd.z; z[k].l[3] = qd.w; }
            dif4_l24<false>(z);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint64_t go = ((uint64_t)(4 * qp + k) << log_n2) + c0 + cc;
                const uint64_t val = gl_mul(l24_value(z[k]), step[4 * t2 + k]);
                if (!(GL355_L24_KO & 2) || val == 0x123456789ull) out[go] = val;
            }
        }
        __syncthreads();
    }
}

// The column pass with the limb quads exchanged in two sub-steps (8-byte cells; see ntt_rows_l24s_kernel): a tile of 32 rows x 2^LOG_TC
// columns on 4 * 2^LOG_TC threads is 32 KB (LOG_TC = 7) or 16 KB (6) of LDS, so the CU holds as many tiles as the registers allow.
GL_DEV __amdgpu_buffer_rsrc_t l24_buf_rsrc(const uint64_t* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000); }
GL_DEV uint64_t l24_buf_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const l24_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
    return ((uint64_t)v.y << 32) | v.x;
}
GL_DEV void l24_buf_store(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, uint64_t x) {
    l24_u32x2 v; v.x = (uint32_t)x; v.y = (uint32_t)(x >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, (int)soff, 0);
}
// (all global accesses through buffer instructions: two vector registers of byte offsets -- one for the load pattern, one for the store
// pattern -- and scalar row offsets, instead of ~30 registers of flat addresses)
// MODE 0: the ratio table re-read per coset; 1: held in registers (no load follows a store inside the loop: on this ISA one counter tracks
// loads and stores in order, so waiting for a load issued after stores waits for their acknowledgements too); 2: one coset per block
// (blockIdx.y), the input and that coset's pre table re-read through L2, no loop at all
template <int LOG_TC, int WPE, int MODE = 0>
__global__ void __launch_bounds__(4 << LOG_TC) __attribute__((amdgpu_waves_per_eu(WPE))) ntt_cols_l24s_cosets_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds_raw[];
    int2* lp = reinterpret_cast<int2*>(lds_raw);
    constexpr uint32_t TC = 1u << LOG_TC;
    const uint32_t tid = threadIdx.x, r = tid >> LOG_TC, cc = tid & (TC - 1);
    const uint32_t log_n2 = a.log_rows;                     // 12
    const uint32_t tiles_per_col = (1u << log_n2) >> LOG_TC; // 32 or 64
    uint32_t tile, colu;
    if (tiles_per_col == 32) { tile = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) & 3); colu = blockIdx.x >> 5; }
    else if (tiles_per_col == 64) { tile = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) & 7); colu = blockIdx.x >> 6; }
    else { tile = blockIdx.x % tiles_per_col; colu = blockIdx.x / tiles_per_col; }
    const uint64_t col = colu;
    const uint32_t c0 = tile << LOG_TC;
    const uint32_t n_bytes = 8u << (log_n2 + 5);            // one coset block / one table: 32 rows of 2^log_n2 words
    const uint32_t row_bytes = 8u << log_n2;
    const uint32_t vld = ((r << log_n2) + c0 + cc) * 8;      // row r + 4 q: + 4 q row_bytes
    const uint32_t vst = (((4 * r) << log_n2) + c0 + cc) * 8; // row 4 (r + 4 t2) + k: + (16 t2 + k) row_bytes
    const __amdgpu_buffer_rsrc_t rs_in = l24_buf_rsrc(a.in + col * a.in_col_stride, n_bytes), rs_pre = l24_buf_rsrc(a.pre_full, n_bytes),
                                 rs_ratio = l24_buf_rsrc(a.ratio_full, n_bytes), rs_step = l24_buf_rsrc(a.step_full, n_bytes);
    uint64_t v[8], step[8], ratio[MODE == 1 ? 8 : 1];
    const __amdgpu_buffer_rsrc_t rs_pre_c = MODE == 2 ? l24_buf_rsrc(a.pre_full + (uint64_t)blockIdx.y * a.pre_full_stride, n_bytes) : rs_pre;
#pragma unroll
    for (int q = 0; q < 8; q++)
        v[q] = gl_mul((GL355_L24_KO & 1) ? vld + q : l24_buf_load(rs_in, vld, 4 * q * row_bytes), l24_buf_load(rs_pre_c, vld, 4 * q * row_bytes));
    if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 8; q++) ratio[q] = l24_buf_load(rs_ratio, vld, 4 * q * row_bytes);
    }
#pragma unroll
    for (int t2 = 0; t2 < 2; t2++)
#pragma unroll
        for (int k = 0; k < 4; k++) step[4 * t2 + k] = l24_buf_load(rs_step, vst, (16 * t2 + k) * row_bytes);
    const uint32_t c_begin = MODE == 2 ? blockIdx.y : 0, c_end = MODE == 2 ? blockIdx.y + 1 : a.n_cosets;
    const CosetSlots slots = coset_slots_of(a);
    for (uint32_t c = c_begin; c < c_end; c++) {
        if (c != c_begin) {
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = gl_mul(v[q], MODE == 1 ? ratio[q] : l24_buf_load(rs_ratio, vld, 4 * q * row_bytes));
            __syncthreads();                                // the previous coset's hi pairs have been read
        }
        const __amdgpu_buffer_rsrc_t rs_out = l24_buf_rsrc(a.out + (uint64_t)coset_slot_at(slots, c) * a.coset_out_stride + col * a.out_col_stride, n_bytes);
        L24 y[8], z[8];
#pragma unroll
        for (int q = 0; q < 8; q++) y[q] = l24_split(v[q]);
        dif8_l24<false>(y);
        l24_twiddles_r<5, false>(y, r);
#pragma unroll
        for (int q = 0; q < 8; q++) lp[TC * (4 * q + r) + cc] = make_int2(y[q].l[0], y[q].l[1]);
        __syncthreads();
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++)
#pragma unroll
            for (int k = 0; k < 4; k++) { const int2 t = lp[TC * (4 * (r + 4 * t2) + k) + cc]; z[4 * t2 + k].l[0] = t.x; z[4 * t2 + k].l[1] = t.y; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; q++) lp[TC * (4 * q + r) + cc] = make_int2(y[q].l[2], y[q].l[3]);
        __syncthreads();
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++)
#pragma unroll
            for (int k = 0; k < 4; k++) { const int2 t = lp[TC * (4 * (r + 4 * t2) + k) + cc]; z[4 * t2 + k].l[2] = t.x; z[4 * t2 + k].l[3] = t.y; }
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) {                    // two radix-4 tasks: rows 4 q' + {0..3}, q' = r and r + 4
            L24 t4[4] = {z[4 * t2], z[4 * t2 + 1], z[4 * t2 + 2], z[4 * t2 + 3]};
            dif4_l24<false>(t4);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint64_t val = gl_mul(l24_value(t4[k]), step[4 * t2 + k]);
                if (!(GL355_L24_KO & 2) || val == 0x123456789ull) l24_buf_store(rs_out, vst, (16 * t2 + k) * row_bytes, val);
                if (k & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Column pass of a two-pass LDE whose column dimension is 2, 4 or 8 (n = 2^13 / 2^14, the in-proof sizes, and 2^15, the aggregation
// nodes, with 4096-point rows): nothing to exchange, so no LDS and no tile -- a thread owns one index i2 < 4096 of one polynomial, loads
// its 2 / 4 / 8 coefficients and table words once and, for every coset, scales (coset ratio), runs the radix-2 / 4 / 8 network in
// registers (shift twiddles), applies the 4-step twiddle and stores.  Same tables, output order and values (mod p) as
// ntt_cols_r8_cosets_kernel<1 / 2 / 3>.  grid = (4096 / 256, columns).
// ------------------------------------------------------------------------------------------------------------------------------
template <int LOG_T>
__global__ void __launch_bounds__(256) ntt_cols_small_cosets_kernel(PassArgs a) {
    static_assert(LOG_T >= 1 && LOG_T <= 3, "2, 4 or 8 rows");
    constexpr int T = 1 << LOG_T;
    const uint32_t log_n2 = a.log_rows;
    const uint64_t i2 = blockIdx.x * 256u + threadIdx.x, col = blockIdx.y;
    const uint64_t* in = a.in + col * a.in_col_stride;
    uint64_t v[T], ratio[T], step[T];
#pragma unroll
    for (int k = 0; k < T; k++) {
        const uint64_t gi = ((uint64_t)k << log_n2) + i2;
        v[k] = gl_mul(in[gi], a.pre_full[gi]);
        ratio[k] = a.ratio_full[gi];
        step[k] = a.step_full[gi];
    }
    const CosetSlots slots = coset_slots_of(a);
    for (uint32_t c = 0; c < a.n_cosets; c++) {
        if (c) {
#pragma unroll
            for (int k = 0; k < T; k++) v[k] = gl_mul(v[k], ratio[k]);
        }
        uint64_t* out = a.out + (uint64_t)coset_slot_at(slots, c) * a.coset_out_stride + col * a.out_col_stride + i2;
        uint64_t y[T];
        if constexpr (LOG_T == 1) {
            y[0] = gl_add(v[0], v[1]); y[1] = gl_sub(v[0], v[1]);
        } else if constexpr (LOG_T == 2) {
            const uint64_t s0 = gl_add(v[0], v[2]), d0 = gl_sub(v[0], v[2]), s1 = gl_add(v[1], v[3]), d1 = gl_mul_2exp<48>(gl_sub(v[1], v[3]));
            y[0] = gl_add(s0, s1); y[1] = gl_sub(s0, s1); y[2] = gl_add(d0, d1); y[3] = gl_sub(d0, d1);
        } else {
            // radix-8 DIF network in registers (the data flow of dif8_l24): omega_8 = -2^24, omega_4 = 2^48, omega_8^3 = -2^72, the signs
            // folded into the order of the subtraction
            uint64_t t[8];
            t[0] = gl_add(v[0], v[4]); t[4] = gl_sub(v[0], v[4]);
            t[1] = gl_add(v[1], v[5]); t[5] = gl_mul_2exp<24>(gl_sub(v[5], v[1]));
            t[2] = gl_add(v[2], v[6]); t[6] = gl_mul_2exp<48>(gl_sub(v[2], v[6]));
            t[3] = gl_add(v[3], v[7]); t[7] = gl_mul_2exp<72>(gl_sub(v[7], v[3]));
            uint64_t u[8];
            u[0] = gl_add(t[0], t[2]); u[2] = gl_sub(t[0], t[2]);
            u[1] = gl_add(t[1], t[3]); u[3] = gl_mul_2exp<48>(gl_sub(t[1], t[3]));
            u[4] = gl_add(t[4], t[6]); u[6] = gl_sub(t[4], t[6]);
            u[5] = gl_add(t[5], t[7]); u[7] = gl_mul_2exp<48>(gl_sub(t[5], t[7]));
#pragma unroll
            for (int k = 0; k < 8; k += 2) { y[k] = gl_add(u[k], u[k + 1]); y[k + 1] = gl_sub(u[k], u[k + 1]); }
        }
#pragma unroll
        for (int k = 0; k < T; k++) out[(uint64_t)k << log_n2] = gl_mul(y[k], step[k]);
    }
}
hipError_t launch_cols_small_cosets(const PassArgs& a, uint32_t log_t, hipStream_t s);

hipError_t launch_rows_l24(const PassArgs& a, hipStream_t s);
hipError_t launch_cols_l24_cosets(const PassArgs& a, hipStream_t s);

}  // namespace gl355
