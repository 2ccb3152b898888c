// Device side of the Goldilocks NTT passes (a2, a3 of SURVEY.md 8): radix-16 / radix-8 butterfly networks on registers, the LDS
// rounds of a tile and the two pass kernels.  Included by ntt.hip (the product) and by tools/ubench/ubench_ntt_rows.hip (timing
// experiments on exactly this code).  Conventions and the design notes are at the top of ntt.hip.
#pragma once
#include "gl355_internal.h"
// GL355_NTT_KO (tools/ubench only; results are wrong with any bit set): 1 = twiddles from a register instead of the table,
// 2 = no global loads of the tile, 4 = no twiddle products, 8 = no butterfly network, 16 = pre / step multipliers from registers
// instead of their tables, 32 = no pre / step products, 64 = column pass writes rows at a padded stride
#ifndef GL355_NTT_KO
#define GL355_NTT_KO 0
#endif
// 1 (default): the radix-8 rounds run the lazily reduced butterfly network (dif8_lazy); 0: the reduce-every-time network (A/B in the ubench)
#ifndef GL355_NTT_R8_LAZY
#define GL355_NTT_R8_LAZY 1
#endif

namespace gl355 {

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
GL_DEV uint32_t brev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// LDS index padding: one extra element every 16 keeps the stride-16 / stride-256 register rounds
// conflict-free for ds_read_b64 (NTT section)
GL_DEV uint32_t lds_phys(uint32_t idx) { return idx + (idx >> 4); }
// PAD = 1 adds one element every 256 as well: the transposing read of the natural -> natural flow (lanes over bit-reversed rows:
// strides of 128 / 256 elements) is then conflict-free too; LDS elements: 2^LT + 2^(LT-4) + 2^(LT-8)
template <int PAD> GL_DEV uint32_t lds_phys_t(uint32_t idx) { return PAD ? idx + (idx >> 4) + (idx >> 8) : idx + (idx >> 4); }

// g^e from a two-level table: lo[j] = g^j (j < 4096), hi[j] = g^(4096 j)
GL_DEV uint64_t pow2lvl(const uint64_t* __restrict__ lo, const uint64_t* __restrict__ hi, uint64_t e) {
    uint64_t v = lo[e & 4095];
    if (hi) v = gl_mul(v, hi[e >> 12]);
    return v;
}

// omega_16^j for the in-register radix-16 butterflies (forward / inverse), filled at ctx creation
__constant__ uint64_t c_w16[2][8];

// (a - b) * omega_16^(+-E) with shifts only: omega_16 = 2^156 = -2^60, omega_16^-1 = 2^36, so
//   forward  E=1..7: -2^60, -2^24, +2^84, +2^48, +2^12, -2^72, -2^36
//   inverse  E=1..7: +2^36, +2^72, -2^12, -2^48, -2^84, +2^24, +2^60
// (a negative sign is absorbed by computing b - a instead of a - b).
template <bool INV, int E>
GL_DEV uint64_t sub_mul_w16(uint64_t a, uint64_t b) {
    constexpr int FWD_S[8] = {0, 60, 24, 84, 48, 12, 72, 36};
    constexpr bool FWD_NEG[8] = {false, true, true, false, false, false, true, true};
    constexpr int INV_S[8] = {0, 36, 72, 12, 48, 84, 24, 60};
    constexpr bool INV_NEG[8] = {false, false, false, true, true, true, false, false};
    constexpr int S = INV ? INV_S[E] : FWD_S[E];
    constexpr bool NEG = INV ? INV_NEG[E] : FWD_NEG[E];
    const uint64_t d = NEG ? gl_sub(b, a) : gl_sub(a, b);
    return gl_mul_2exp<S>(d);
}

// In-register DIF butterfly network on 2^RHO values: x[pos] <- X[bitrev(pos)].
template <int RHO, bool INV>
GL_DEV void dif_regs_generic(uint64_t (&x)[16]) {
#pragma unroll
    for (int s = 0; s < RHO; s++) {
        const int half = 1 << (RHO - 1 - s);
#pragma unroll
        for (int blk = 0; blk < (1 << s); blk++) {
#pragma unroll
            for (int j = 0; j < half; j++) {
                const int i0 = blk * 2 * half + j, i1 = i0 + half;
                uint64_t a = x[i0], b = x[i1];
                x[i0] = gl_add(a, b);
                uint64_t d = gl_sub(a, b);
                // twiddle omega_{2*half}^j = omega_16^(j * 8 / half)
                const int e = j * (8 / half);
                x[i1] = (e == 0) ? d : gl_mul(d, c_w16[INV ? 1 : 0][e]);
            }
        }
    }
}

template <bool INV, int HALF, int J>
GL_DEV void dif_pair(uint64_t (&x)[16], int i0) {
    const uint64_t a = x[i0], b = x[i0 + HALF];
    x[i0] = gl_add(a, b);
    x[i0 + HALF] = sub_mul_w16<INV, J * (8 / HALF)>(a, b);
}
template <bool INV, int HALF, int BLK, int J>
GL_DEV void dif_stage_unrolled(uint64_t (&x)[16]) {
    if constexpr (J < HALF) {
        dif_pair<INV, HALF, J>(x, BLK * 2 * HALF + J);
        dif_stage_unrolled<INV, HALF, BLK, J + 1>(x);
    }
}
template <bool INV, int HALF, int NBLK, int BLK>
GL_DEV void dif_stage_blocks(uint64_t (&x)[16]) {
    if constexpr (BLK < NBLK) {
        dif_stage_unrolled<INV, HALF, BLK, 0>(x);
        dif_stage_blocks<INV, HALF, NBLK, BLK + 1>(x);
    }
}
// shift-twiddle version of the same network (all internal twiddles are powers of two)
template <int RHO, bool INV>
GL_DEV void dif_regs(uint64_t (&x)[16]) {
    if constexpr (RHO >= 4) dif_stage_blocks<INV, 8, 1, 0>(x);
    if constexpr (RHO >= 3) dif_stage_blocks<INV, 4, 1 << (RHO - 3), 0>(x);
    if constexpr (RHO >= 2) dif_stage_blocks<INV, 2, 1 << (RHO - 2), 0>(x);
    dif_stage_blocks<INV, 1, 1 << (RHO - 1), 0>(x);
}

// ---- radix-8 network on lazily reduced operands (the radix-8 commit-path kernels) ---------------------------------------------
// Between the three butterfly layers a value is kept as v + k * 2^64 with a small non-negative k (a third 32-bit limb) instead of
// being reduced after every addition: a sum is one 3-limb carry chain, a difference is a + (m p - b) with m p >= b (m = 2, 4, 8 by
// layer, so k <= 14 at the end), the shift-twiddles (only 2^24, 2^48, 2^72 occur in radix 8) take the third limb in through one
// funnel shift, and one multiply-add per output folds k back (k * 2^64 = k * EPS).  189 VALU instructions per 8 points instead of
// 245 for the reduce-every-time form (compare-and-select corrections), bit for bit the same values mod p.
struct L96 { uint64_t v; uint32_t k; };
typedef unsigned __int128 gl_u128;
GL_DEV L96 l96(uint64_t v) { L96 r; r.v = v; r.k = 0; return r; }
GL_DEV gl_u128 l96_int(L96 a) { return ((gl_u128)a.k << 64) | a.v; }
GL_DEV L96 l96_of(gl_u128 s) { L96 r; r.v = (uint64_t)s; r.k = (uint32_t)(s >> 64); return r; }
GL_DEV L96 l96_add(L96 a, L96 b) { return l96_of(l96_int(a) + l96_int(b)); }
template <int M> GL_DEV L96 l96_sub(L96 a, L96 b) {        // a + (M p - b); the caller guarantees b <= M p.  M p = (M-1) 2^64 + (2^32 - M) 2^32 + M
    const gl_u128 mp = ((gl_u128)(M - 1) << 64) | ((uint64_t)(0x100000000ull - M) << 32) | (uint64_t)M;
    return l96_of(l96_int(a) + (mp - l96_int(b)));
}
#if !defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the declarations to parse (gl_field.cuh defines these for the device pass)
GL_DEV uint64_t gl_dev_add_mul_eps(uint64_t x, uint32_t k);
GL_DEV uint64_t gl_dev_sub32(uint32_t lo0, uint32_t lo1, uint32_t sub32);
#endif
GL_DEV uint64_t l96_norm(L96 a) { return gl_dev_add_mul_eps(a.v, a.k); }
// (v + k 2^64) * 2^S mod p as any u64, for the shifts of the radix-8 network and k < 2^8.  With T = 2^32: T^2 = EPS, T^3 = -1.
template <int S> GL_DEV uint64_t l96_shift(L96 a) {
    static_assert(S == 24 || S == 48 || S == 72, "radix-8 shift-twiddles only");
    const uint32_t x0 = (uint32_t)a.v, x1 = (uint32_t)(a.v >> 32), x2 = a.k;
    if constexpr (S == 24) {
        return gl_dev_add_mul_eps(a.v << 24, __builtin_amdgcn_alignbit(x2, x1, 8));           // low 64 bits + (bits 64..95) * EPS
    } else {
        constexpr int sh = S == 48 ? 16 : 8;                                                    // y = x << sh = (y0, y1, y2), y2 < 2^32
        const uint32_t y0 = x0 << sh, y1 = __builtin_amdgcn_alignbit(x1, x0, 32 - sh), y2 = __builtin_amdgcn_alignbit(x2, x1, 32 - sh);
        if constexpr (S == 48) return gl_dev_add_mul_eps(gl_dev_sub32(0, y0, y2), y1);          // y T = y0 T + y1 EPS - y2
        else return gl_sub((uint64_t)y0 * 0xFFFFFFFFull, ((uint64_t)y2 << 32) | y1);           // y T^2 = y0 EPS - (y1 + y2 T)
    }
}
template <bool INV, int E, int M>
GL_DEV void l96_bfly(L96& a, L96& b) {                      // (a, b) <- (a + b, (a - b) * omega_16^(+-E)), E even
    constexpr int FWD_S[4] = {0, 24, 48, 72};
    constexpr bool FWD_NEG[4] = {false, true, false, true};
    constexpr int INV_S[4] = {0, 72, 48, 24};
    constexpr bool INV_NEG[4] = {false, false, true, false};
    constexpr int S = INV ? INV_S[E / 2] : FWD_S[E / 2];
    constexpr bool NEG = INV ? INV_NEG[E / 2] : FWD_NEG[E / 2];
    const L96 s = l96_add(a, b);
    const L96 d = NEG ? l96_sub<M>(b, a) : l96_sub<M>(a, b);
    a = s;
    if constexpr (S == 0) b = d;
    else b = l96(l96_shift<S>(d));
}
template <bool INV>
GL_DEV void dif8_lazy(uint64_t (&x)[16]) {
    L96 y[8];
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] = l96(x[i]);
    // bounds: inputs < 2^64; layer 1 out < 3 * 2^64; layer 2 out < 7 * 2^64 (subtrahends < 4 p); layer 3 out < 15 * 2^64 (subtrahends < 8 p)
    l96_bfly<INV, 0, 2>(y[0], y[4]); l96_bfly<INV, 2, 2>(y[1], y[5]); l96_bfly<INV, 4, 2>(y[2], y[6]); l96_bfly<INV, 6, 2>(y[3], y[7]);
    l96_bfly<INV, 0, 4>(y[0], y[2]); l96_bfly<INV, 4, 4>(y[1], y[3]); l96_bfly<INV, 0, 4>(y[4], y[6]); l96_bfly<INV, 4, 4>(y[5], y[7]);
    l96_bfly<INV, 0, 8>(y[0], y[1]); l96_bfly<INV, 0, 8>(y[2], y[3]); l96_bfly<INV, 0, 8>(y[4], y[5]); l96_bfly<INV, 0, 8>(y[6], y[7]);
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = l96_norm(y[i]);
}

// One DIF round of radix 2^RHO on an LDS tile.  The transform currently consists of independent
// blocks of 2^m (transform units); transform bit 0 sits at tile-index bit LO.  `tw` is the ROUND-MAJOR twiddle table of this
// radix and direction (Ctx::twr): tw[2^m + (k0 << (m - RHO)) + r] = omega_{2^m}^(+-r * k0), so that the 64 lanes of a wave
// (consecutive r) read 64 consecutive words per register q.  With the plain omega_{2^14}^e table the same loads were gathers
// at a stride of k0 * 2^(14-m) words -- up to 64 cache lines per wave instruction, and a third of the row pass's time
// (tools/ubench/ubench_ntt_rows.hip, knock-out 1).
template <int LT, int RHO, bool INV, bool LAZY = false, int PAD = 0>
GL_DEV void dif_round(uint64_t* lds, const uint64_t* __restrict__ tw, int m, int LO, int tid, int nthreads) {
    constexpr int R = 1 << RHO;
    const int tasks = (1 << LT) >> RHO;
    const int fbit = LO + m - RHO;  // lowest tile-index bit of the radix field
    for (int t = tid; t < tasks; t += nthreads) {
        const uint32_t low = t & ((1u << fbit) - 1), high = t >> fbit;
        const uint32_t idx0 = (high << (fbit + RHO)) | low;
        uint64_t x[16];
#pragma unroll
        for (int q = 0; q < R; q++) x[q] = lds[lds_phys_t<PAD>(idx0 + ((uint32_t)q << fbit))];
        if constexpr (!(GL355_NTT_KO & 8)) {
            if constexpr (LAZY && RHO == 3) dif8_lazy<INV>(x);
            else dif_regs<RHO, INV>(x);
        }
        if ((GL355_NTT_KO & 4) == 0 && m > RHO) {
            // output k0 = bitrev(q) of this butterfly is multiplied by omega_{2^m}^(r*k0)
            const uint32_t r = (idx0 >> LO) & ((1u << (m - RHO)) - 1);
            const uint64_t* __restrict__ twm = tw + (1u << m) + r;   // round-major table: lanes with consecutive r read consecutive words
#pragma unroll
            for (int q = 1; q < R; q++) {
                const uint32_t k0 = brev(q, RHO);
                if constexpr (GL355_NTT_KO & 1) x[q] = gl_mul(x[q], x[0] + r * k0);
                else x[q] = gl_mul(x[q], twm[k0 << (m - RHO)]);
            }
        }
#pragma unroll
        for (int q = 0; q < R; q++) lds[lds_phys_t<PAD>(idx0 + ((uint32_t)q << fbit))] = x[q];
    }
}

// all rounds for LOG_T transform bits starting at tile bit LO
template <int LT, int LOG_T, bool INV>
GL_DEV void dif_tile(uint64_t* lds, const uint64_t* __restrict__ tw, int LO, int tid, int nthreads) {
    int m = LOG_T;
#pragma unroll
    for (int round = 0; round < LOG_T / 4; round++) {
        dif_round<LT, 4, INV>(lds, tw, m, LO, tid, nthreads);
        m -= 4;
        __syncthreads();
    }
    constexpr int REM = LOG_T % 4;
    if constexpr (REM == 3) dif_round<LT, 3, INV>(lds, tw, 3, LO, tid, nthreads);
    if constexpr (REM == 2) dif_round<LT, 2, INV>(lds, tw, 2, LO, tid, nthreads);
    if constexpr (REM == 1) dif_round<LT, 1, INV>(lds, tw, 1, LO, tid, nthreads);
    if constexpr (REM != 0) __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// pass kernels
// ------------------------------------------------------------------------------------------------
struct PassArgs {
    const uint64_t* in;
    uint64_t* out;
    uint64_t in_col_stride;    // elements between polynomial columns (input)
    uint64_t out_col_stride;   // elements between polynomial columns (output)
    uint32_t batch;            // polynomial columns
    uint32_t n_cosets;         // independent (table, output-offset) variants per column (LDE); >= 1
    uint64_t coset_out_stride; // output offset of coset c = coset_slot[c] * coset_out_stride
    uint8_t coset_slot[16];
    uint32_t log_n;            // log2 of the whole transform (per column, per coset)
    uint32_t log_rows;         // row kernel: rows per column = 2^log_rows; col kernel: log2(N2)
    const uint64_t* tw;        // round-major twiddles of the kernel's radix and direction (Ctx::twr)
    const uint64_t* tw_r8;     // the same for the radix-8 kernels
    const uint64_t* pre_lo;    // optional multiplier g^i on natural-order INPUT index i
    const uint64_t* pre_hi;    //   tables of coset c at pre_lo + c*4096, pre_hi + c*4096
    const uint64_t* post_lo;   // optional multiplier on natural-order OUTPUT index
    const uint64_t* post_hi;
    const uint64_t* step_lo;   // 4-step twiddle omega_N^(+-e): lo/hi tables (col kernel only)
    const uint64_t* step_hi;
    const uint64_t* pre_full;  // optional full table g_c^i (coset c at + c*pre_full_stride): 1 load + 1 mul
    uint64_t pre_full_stride;
    const uint64_t* step_full; // optional full 4-step twiddle table in STORE order (col kernel, first pass)
    const uint64_t* ratio_full; // LDE column pass over all cosets in one block: (base[c+1] / base[c])^i, constant over c (Ctx::full_pow_table)
    const uint64_t* mid;       // 24-bit-limb row kernel (ntt_l24.cuh): the twiddles between the two radix-64 super-rounds of a 4096-point row, LDS-cell order
    uint64_t scale;            // constant multiplier at store (1 = none)
    uint32_t in_bitrev;        // input transform index is bit-reversed in memory
    uint32_t out_natural;      // write natural order (else DIF-native bit-reversed order)
    uint32_t canon;            // canonicalise at store (last pass)
};

// coset_slot[c] inside a coset loop WITHOUT a memory access (round 6).  Indexed by the loop counter the byte array is read from the kernel-argument
// segment by a vector load, and the s_waitcnt vmcnt(0) in front of its use also waits for every store of the previous coset: the all-cosets column
// kernels drained their store queue eight times per tile (the "missing overlap" of DESIGN 4.1).  The sixteen bytes are taken once, as two scalars.
struct CosetSlots { uint64_t lo, hi; };
GL_DEV CosetSlots coset_slots_of(const PassArgs& a) {
    CosetSlots s;
    __builtin_memcpy(&s.lo, a.coset_slot, 8);
    __builtin_memcpy(&s.hi, a.coset_slot + 8, 8);
    return s;
}
GL_DEV uint32_t coset_slot_at(const CosetSlots& s, uint32_t c) { return (uint32_t)(((c < 8 ? s.lo : s.hi) >> (8 * (c & 7))) & 0xffu); }

// radix-8 commit-path kernels, compiled in ntt_r8.hip
hipError_t launch_rows_r8(const PassArgs& a, uint32_t log_t, bool inv, hipStream_t s);
hipError_t launch_cols_r8(const PassArgs& a, uint32_t log_t, bool inv, hipStream_t s);
hipError_t launch_cols_r8_big(const PassArgs& a, uint32_t log_t, uint32_t lt, hipStream_t s);
hipError_t launch_cols_r8_cosets(const PassArgs& a, uint32_t log_t, hipStream_t s);

// Row pass: each row = 2^LOG_T contiguous elements; a tile packs 2^(LT-LOG_T) rows.
// FAST = the commit-path shape with every optional multiplier compiled out: rows pass = no pre / post multiplier, no scaling,
// natural tile order in and out, canonical store; cols pass = full pre and step tables, nothing else.  The general kernels test
// those options per element at run time (~110 branches per phase); the LDE of a commitment never uses them.
template <int LT, int LOG_T, bool INV, bool FAST = false, int WPE = (LT == 12 ? 3 : 1)>
__global__ void __launch_bounds__(1 << (LT - 4)) __attribute__((amdgpu_waves_per_eu(WPE))) ntt_rows_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    constexpr int NT = 1 << (LT - 4);
    constexpr int RPT = 1 << (LT - LOG_T);  // rows per tile
    const int tid = threadIdx.x;
    const uint32_t coset = blockIdx.x % a.n_cosets;
    const uint64_t tile = blockIdx.x / a.n_cosets;
    const uint64_t row0 = tile * RPT;  // global row id over (column, row-in-column)
    const uint64_t rows_per_col = 1ull << a.log_rows;
    const uint64_t total_rows = rows_per_col * a.batch;
    const uint64_t* pre_lo = a.pre_lo ? a.pre_lo + (uint64_t)coset * 4096 : nullptr;
    const uint64_t* pre_hi = (a.pre_lo && a.pre_hi) ? a.pre_hi + (uint64_t)coset * 4096 : nullptr;

#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t g = tid + i * NT;
        const uint32_t lr = g >> LOG_T, e = g & ((1u << LOG_T) - 1);
        const uint64_t row = row0 + lr;
        uint64_t v = 0;
        if (row < total_rows) {
            const uint64_t col = row >> a.log_rows, rin = row & (rows_per_col - 1);
            if constexpr (GL355_NTT_KO & 2) v = col * a.in_col_stride + (rin << LOG_T) + e;
            else v = a.in[col * a.in_col_stride + (rin << LOG_T) + e];
            if constexpr (!FAST) {
                if (a.pre_full) v = gl_mul(v, a.pre_full[(uint64_t)coset * a.pre_full_stride + (rin << LOG_T) + e]);
                else if (pre_lo) v = gl_mul(v, pow2lvl(pre_lo, pre_hi, (rin << LOG_T) + e));
            }
        }
        const uint32_t le = (!FAST && a.in_bitrev) ? brev(e, LOG_T) : e;
        lds[lds_phys((lr << LOG_T) | le)] = v;
    }
    __syncthreads();
    dif_tile<LT, LOG_T, INV>(lds, a.tw, 0, tid, NT);
    const uint64_t out_base = (uint64_t)a.coset_slot[coset] * a.coset_out_stride;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t g = tid + i * NT;
        const uint32_t lr = g >> LOG_T, e = g & ((1u << LOG_T) - 1);
        const uint64_t row = row0 + lr;
        if (row < total_rows) {
            const uint64_t col = row >> a.log_rows, rin = row & (rows_per_col - 1);
            const uint32_t le = (!FAST && a.out_natural) ? brev(e, LOG_T) : e;
            uint64_t v = lds[lds_phys((lr << LOG_T) | le)];
            if constexpr (!FAST) {
                if (a.post_lo) v = gl_mul(v, pow2lvl(a.post_lo, a.post_hi, (rin << LOG_T) + e));
                if (a.scale != 1) v = gl_mul(v, a.scale);
                if (a.canon) v = gl_canon(v);
            } else v = gl_canon(v);
            a.out[out_base + col * a.out_col_stride + (rin << LOG_T) + e] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Radix-8 kernels for the commit-path shapes (forward, natural order in, bit-reversed out, at most a full pre table): 8 elements
// per thread and task.  One more LDS exchange than radix 16 (12 layers = 4 rounds instead of 3) and 7/8 instead of 15/16 twiddle
// products per element and round -- but 64-72 VGPRs instead of 168, i.e. 6-8 waves per SIMD instead of 3 to cover tile load,
// tile store and the barriers, fewer shift-twiddles per layer (5 per 8 elements and round instead of 17 per 16), and with that
// many waves the 15-instruction inline-asm product (GL_MUL_VARIANT 1, ntt_r8.hip) pays: measured in tools/ubench/ubench_ntt_rows.hip
// 1.25 -> 0.83 ms for the row pass and 0.75 -> 0.58 ms for the column pass of the 2^17 -> 2^20 x 135 LDE.
// ------------------------------------------------------------------------------------------------
template <int LT, int LOG_T, bool INV, int PAD = 0>
GL_DEV void dif_tile_r8(uint64_t* lds, const uint64_t* __restrict__ tw, int LO, int tid, int nthreads) {
    int m = LOG_T;
#pragma unroll
    for (int round = 0; round < LOG_T / 3; round++) {
        dif_round<LT, 3, INV, GL355_NTT_R8_LAZY != 0, PAD>(lds, tw, m, LO, tid, nthreads);
        m -= 3;
        __syncthreads();
    }
    constexpr int REM = LOG_T % 3;
    if constexpr (REM == 2) dif_round<LT, 2, INV, false, PAD>(lds, tw, 2, LO, tid, nthreads);
    if constexpr (REM == 1) dif_round<LT, 1, INV, false, PAD>(lds, tw, 1, LO, tid, nthreads);
    if constexpr (REM != 0) __syncthreads();
}
// The first radix-8 round straight from the registers that loaded the tile, and the last round straight to its consumer: a thread that
// loaded tile elements tid + q * NT (q < 8, NT = 2^(LT-3) threads) holds exactly the operands of its own first-round butterfly (the first
// DIF round pairs elements 2^(LT-3) apart), so the load -> LDS -> barrier -> LDS -> registers detour is skipped; and the last round's
// results (no twiddles follow) can go to `sink(tile index, value)` instead of back to LDS.  GL355_NTT_R8_DIRECT=0 keeps the staged form (A/B).
#ifndef GL355_NTT_R8_DIRECT
#define GL355_NTT_R8_DIRECT 1
#endif
template <int LT, int LOG_T, bool INV, int PAD = 0>
GL_DEV void dif_first_round_regs(uint64_t (&x)[16], uint64_t* lds, const uint64_t* __restrict__ tw, int LO, int tid) {
    static_assert(LOG_T >= 3, "needs a radix-8 first round");
    constexpr int fbit = LT - 3;
    if constexpr (!(GL355_NTT_KO & 8)) {
        if constexpr (GL355_NTT_R8_LAZY != 0) dif8_lazy<INV>(x);
        else dif_regs<3, INV>(x);
    }
    if constexpr (LOG_T > 3 && (GL355_NTT_KO & 4) == 0) {
        const uint32_t r = ((uint32_t)tid >> LO) & ((1u << (LOG_T - 3)) - 1);
        const uint64_t* __restrict__ twm = tw + (1u << LOG_T) + r;
#pragma unroll
        for (int q = 1; q < 8; q++) {
            if constexpr (GL355_NTT_KO & 1) x[q] = gl_mul(x[q], x[0] + r * brev(q, 3));
            else x[q] = gl_mul(x[q], twm[brev(q, 3) << (LOG_T - 3)]);
        }
    }
#pragma unroll
    for (int q = 0; q < 8; q++) lds[lds_phys_t<PAD>((uint32_t)tid + ((uint32_t)q << fbit))] = x[q];
}
// last round of radix 2^RHO (m == RHO: no twiddles): LDS -> network -> sink(tile index, value)
template <int LT, int RHO, bool INV, class Sink, int PAD = 0>
GL_DEV void dif_last_round_sink(const uint64_t* lds, int LO, int tid, int nthreads, Sink sink) {
    constexpr int R = 1 << RHO;
    const int tasks = (1 << LT) >> RHO;
    const int fbit = LO;
    // unrolled (the trip count is a compile-time constant at every call site: a radix-2 / radix-4 last round is 4 / 2 trips): rolled, every trip's
    // table loads in the sink were issued and waited for behind the previous trip's stores -- one exposed load latency per trip (round 6)
#pragma unroll
    for (int t = tid; t < tasks; t += nthreads) {
        const uint32_t low = t & ((1u << fbit) - 1), high = t >> fbit;
        const uint32_t idx0 = (high << (fbit + RHO)) | low;
        uint64_t x[16];
#pragma unroll
        for (int q = 0; q < R; q++) x[q] = lds[lds_phys_t<PAD>(idx0 + ((uint32_t)q << fbit))];
        if constexpr (!(GL355_NTT_KO & 8)) {
            if constexpr (GL355_NTT_R8_LAZY != 0 && RHO == 3) dif8_lazy<INV>(x);
            else dif_regs<RHO, INV>(x);
        }
#pragma unroll
        for (int q = 0; q < R; q++) sink(idx0 + ((uint32_t)q << fbit), x[q]);
    }
}
// The same with a TABLE WORD per output (the 4-step twiddle applied at the store): fetch(tile index) for all of a thread's outputs first, then the
// network and the stores.  Written as `out[go] = v * table[go]` inside the sink, every table load sat behind the previous output's store (the
// compiler cannot prove that `out` and the table do not alias) and was waited for with vmcnt(0), which also drains that store: eight exposed load
// latencies per thread and tile (round 6; the all-cosets column kernels always held their eight step words in registers).
template <int LT, int RHO, bool INV, int PAD, class Fetch, class Sink>
GL_DEV void dif_last_round_sink_tab(const uint64_t* lds, int LO, int tid, int nthreads, Fetch fetch, Sink sink) {
    constexpr int R = 1 << RHO;
    const int tasks = (1 << LT) >> RHO;
    const int fbit = LO;
    constexpr int MAXW = 16;                                 // table words a thread holds at once
    uint64_t w[MAXW];
    const int trips = (tasks + nthreads - 1) / nthreads;     // compile-time at every call site
    const bool all_first = trips * R <= MAXW;
    if (all_first) {
#pragma unroll
        for (int k = 0; k < MAXW / R; k++) {
            const int t = tid + k * nthreads;
            if (k < trips && t < tasks) {
                const uint32_t idx0 = ((uint32_t)(t >> fbit) << (fbit + RHO)) | (t & ((1u << fbit) - 1));
#pragma unroll
                for (int q = 0; q < R; q++) w[k * R + q] = fetch(idx0 + ((uint32_t)q << fbit));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < (all_first ? MAXW / R : 64); k++) {
        const int t = tid + k * nthreads;
        if (k >= trips || t >= tasks) break;
        const uint32_t low = t & ((1u << fbit) - 1), high = t >> fbit;
        const uint32_t idx0 = (high << (fbit + RHO)) | low;
        uint64_t x[16], wk[R];
#pragma unroll
        for (int q = 0; q < R; q++) wk[q] = all_first ? w[(k * R + q) % MAXW] : fetch(idx0 + ((uint32_t)q << fbit));
#pragma unroll
        for (int q = 0; q < R; q++) x[q] = lds[lds_phys_t<PAD>(idx0 + ((uint32_t)q << fbit))];
        if constexpr (GL355_NTT_R8_LAZY != 0 && RHO == 3) dif8_lazy<INV>(x);
        else dif_regs<RHO, INV>(x);
#pragma unroll
        for (int q = 0; q < R; q++) sink(idx0 + ((uint32_t)q << fbit), x[q], wk[q]);
    }
}
// rounds between a register-fed first round and (KEEP_LAST) a sunk last round; x holds the thread's 8 loaded elements
template <int LT, int LOG_T, bool INV, bool KEEP_LAST, int PAD = 0>
GL_DEV void dif_tile_r8_regs(uint64_t (&x)[16], uint64_t* lds, const uint64_t* __restrict__ tw, int LO, int tid, int nthreads) {
    constexpr int FULL = LOG_T / 3, REM = LOG_T % 3;
    static_assert(!KEEP_LAST || LOG_T >= 4, "the last round must not be the first");
    dif_first_round_regs<LT, LOG_T, INV, PAD>(x, lds, tw, LO, tid);
    __syncthreads();
    int m = LOG_T - 3;
    constexpr int MID = FULL - 1 - ((KEEP_LAST && REM == 0) ? 1 : 0);       // radix-8 rounds done here after the first
#pragma unroll
    for (int round = 0; round < MID; round++) {
        dif_round<LT, 3, INV, GL355_NTT_R8_LAZY != 0, PAD>(lds, tw, m, LO, tid, nthreads);
        m -= 3;
        __syncthreads();
    }
    if constexpr (!KEEP_LAST) {
        if constexpr (REM == 2) dif_round<LT, 2, INV, false, PAD>(lds, tw, 2, LO, tid, nthreads);
        if constexpr (REM == 1) dif_round<LT, 1, INV, false, PAD>(lds, tw, 1, LO, tid, nthreads);
        if constexpr (REM != 0) __syncthreads();
    }
}
template <int LOG_T> struct R8Last { static constexpr int RHO = (LOG_T % 3) ? (LOG_T % 3) : 3; };

// Row pass / single pass: one 2^LT-point row per tile; PRE = multiply by the full table a.pre_full (coset powers) at the load.
// blockIdx = tile * n_cosets + coset like the radix-16 kernel (each XCD keeps one coset's table in its L2).
// INV: the inverse transform (omega^-1 twiddles and shifts); it also honours a.out_natural (natural order within the row: the single-pass
// ifft) and a.scale (the 1/n), both wave-uniform tests at the store.
template <int LT, bool PRE, int WPE, bool INV = false>
__global__ void __launch_bounds__(LT >= 13 ? 1024 : 512) __attribute__((amdgpu_waves_per_eu(WPE))) ntt_rows_r8_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    constexpr int NT = LT >= 13 ? 1024 : 512, EPT = (1 << LT) / NT;
    const int tid = threadIdx.x;
    const uint32_t coset = blockIdx.x % a.n_cosets;
    const uint64_t row = blockIdx.x / a.n_cosets;           // global row id over (column, row in column)
    const uint64_t col = row >> a.log_rows, rin = row & ((1ull << a.log_rows) - 1);
    const uint64_t* in = a.in + col * a.in_col_stride + (rin << LT);
    const uint64_t* pre = PRE ? a.pre_full + (uint64_t)coset * a.pre_full_stride + (rin << LT) : nullptr;
    uint64_t* out = a.out + (uint64_t)a.coset_slot[coset] * a.coset_out_stride + col * a.out_col_stride + (rin << LT);
    // measured: 8192-point tiles gain 5 % from the register-fed first round, 4096-point tiles at 64 VGPRs lose 4 % (more spilled registers)
    if constexpr (EPT == 8 && LT == 13 && GL355_NTT_R8_DIRECT != 0) {
        uint64_t x[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t g = tid + i * NT;
            x[i] = (GL355_NTT_KO & 2) ? (uint64_t)g + row : in[g];
            if constexpr (PRE) x[i] = gl_mul(x[i], pre[g]);
        }
        dif_tile_r8_regs<LT, LT, INV, false>(x, lds, a.tw_r8, 0, tid, NT);
    } else {
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            const uint32_t g = tid + i * NT;
            uint64_t v = (GL355_NTT_KO & 2) ? (uint64_t)g + row : in[g];
            if constexpr (PRE) v = gl_mul(v, pre[g]);
            lds[lds_phys(g)] = v;
        }
        __syncthreads();
        dif_tile_r8<LT, LT, INV>(lds, a.tw_r8, 0, tid, NT);
    }
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const uint32_t g = tid + i * NT;
        if constexpr (INV) {
            uint64_t v = lds[lds_phys(a.out_natural ? brev(g, LT) : g)];
            if (a.scale != 1) v = gl_mul(v, a.scale);
            out[g] = gl_canon(v);
        } else out[g] = gl_canon(lds[lds_phys(g)]);
    }
}

// Column pass: transform over the row index of an [2^LOG_T][N2] matrix (N2 = 2^log_rows... here
// a.log_rows holds log2(N2)); a tile is all 2^LOG_T rows x TC = 2^(12-LOG_T) adjacent columns.
template <int LOG_T, bool INV, bool FAST = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) ntt_cols_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    constexpr int LT = 12, NT = 256;
    constexpr int LOG_TC = LT - LOG_T, TC = 1 << LOG_TC;
    const int tid = threadIdx.x;
    const uint32_t log_n2 = a.log_rows;
    const uint64_t n2 = 1ull << log_n2;
    const uint32_t coset = blockIdx.x % a.n_cosets;
    const uint64_t tile = blockIdx.x / a.n_cosets;
    const uint64_t tiles_per_col = n2 >> LOG_TC;
    const uint64_t col = tile / tiles_per_col;
    const uint64_t c0 = (tile % tiles_per_col) << LOG_TC;  // first matrix column of the tile
    const uint64_t* in = a.in + col * a.in_col_stride;
    uint64_t* out = a.out + (uint64_t)a.coset_slot[coset] * a.coset_out_stride + col * a.out_col_stride;
    const uint64_t* pre_lo = a.pre_lo ? a.pre_lo + (uint64_t)coset * 4096 : nullptr;
    const uint64_t* pre_hi = (a.pre_lo && a.pre_hi) ? a.pre_hi + (uint64_t)coset * 4096 : nullptr;
    const bool step_at_load = a.in_bitrev != 0;  // second pass of the bitrev -> natural flow

#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t g = tid + i * NT;
        const uint32_t r = g >> LOG_TC, cc = g & (TC - 1);
        const uint64_t gi = ((uint64_t)r << log_n2) + c0 + cc;
        uint64_t v = in[gi];
        uint32_t lr = r;  // logical transform index
        if constexpr (FAST) {
            v = gl_mul(v, a.pre_full[(uint64_t)coset * a.pre_full_stride + gi]);
        } else {
            lr = a.in_bitrev ? brev(r, LOG_T) : r;
            if (a.pre_full) v = gl_mul(v, a.pre_full[(uint64_t)coset * a.pre_full_stride + gi]);
            else if (pre_lo) v = gl_mul(v, pow2lvl(pre_lo, pre_hi, gi));
            if (step_at_load && a.step_lo) v = gl_mul(v, pow2lvl(a.step_lo, a.step_hi, (uint64_t)lr * (c0 + cc)));
        }
        lds[lds_phys((lr << LOG_TC) | cc)] = v;
    }
    __syncthreads();
    dif_tile<LT, LOG_T, INV>(lds, a.tw, LOG_TC, tid, NT);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t g = tid + i * NT;
        const uint32_t r = g >> LOG_TC, cc = g & (TC - 1);
        const uint32_t lr = (!FAST && a.out_natural) ? brev(r, LOG_T) : r;  // LDS row holding output row r
        uint64_t v = lds[lds_phys((lr << LOG_TC) | cc)];
        const uint64_t go = ((uint64_t)r << log_n2) + c0 + cc;
        if constexpr (FAST) {
            v = gl_mul(v, a.step_full[go]);
        } else {
            if (!step_at_load && a.step_full) v = gl_mul(v, a.step_full[go]);
            else if (!step_at_load && a.step_lo) {
                const uint32_t k1 = a.out_natural ? r : brev(r, LOG_T);  // transform output index
                v = gl_mul(v, pow2lvl(a.step_lo, a.step_hi, (uint64_t)k1 * (c0 + cc)));
            }
            if (a.post_lo) v = gl_mul(v, pow2lvl(a.post_lo, a.post_hi, go));
            if (a.scale != 1) v = gl_mul(v, a.scale);
            if (a.canon) v = gl_canon(v);
        }
        out[go] = v;
    }
}

// Column pass of the commit path (full step table, PRE = full pre table, nothing else) with radix-8 rounds: 8 elements per thread, 512 threads per tile.
// LT: log2 of the tile (12: 4096 elements on 512 threads; 13 / 14: 8192 / 16384 elements on 1024 threads, round 5 -- column passes over 2^9 .. 2^11
// points with 16- / 8-column tiles, so that a 2^21 .. 2^23-point transform is TWO passes with 4096-point rows instead of three).  A tile of 8
// columns moves 64-byte halves of 128-byte lines: with PAIR the tile holding the other halves runs on the same XCD right after it (blocks are
// dealt to the XCDs round-robin: block 8 q + x takes tile 2 (8 (q >> 1) + x) + (q & 1), as in ntt_cols_r8_nat_kernel).
template <int LOG_T, bool PRE, int WPE, bool INV = false, int LT = 12>
__global__ void __launch_bounds__(LT == 12 ? 512 : 1024) __attribute__((amdgpu_waves_per_eu(WPE))) ntt_cols_r8_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    constexpr int NT = LT == 12 ? 512 : 1024, EPT = (1 << LT) / NT;
    constexpr int LOG_TC = LT - LOG_T, TC = 1 << LOG_TC;
    static_assert(LOG_TC >= 0, "tile smaller than a column");
    const int tid = threadIdx.x;
    const uint32_t log_n2 = a.log_rows;
    const uint64_t n2 = 1ull << log_n2;
    uint64_t bt = blockIdx.x;
    if constexpr (LT > 12 && TC * 8 < 128) {
        if ((gridDim.x & 15) == 0 && a.n_cosets == 1) { const uint64_t q = bt >> 3, xcd = bt & 7; bt = ((((q >> 1) << 3) + xcd) << 1) + (q & 1); }
    }
    const uint32_t coset = bt % a.n_cosets;
    const uint64_t tile = bt / a.n_cosets;
    const uint64_t tiles_per_col = n2 >> LOG_TC;
    const uint64_t col = tile / tiles_per_col;
    const uint64_t c0 = (tile % tiles_per_col) << LOG_TC;
    const uint64_t* in = a.in + col * a.in_col_stride;
    const uint64_t* pre = PRE ? a.pre_full + (uint64_t)coset * a.pre_full_stride : nullptr;
    uint64_t* out = a.out + (uint64_t)a.coset_slot[coset] * a.coset_out_stride + col * a.out_col_stride;
    auto store = [&](uint32_t g, uint64_t v) {
        const uint32_t r = g >> LOG_TC, cc = g & (TC - 1);
        const uint64_t go = ((uint64_t)r << log_n2) + c0 + cc;
        if constexpr (!(GL355_NTT_KO & 32)) v = gl_mul(v, (GL355_NTT_KO & 16) ? go + 5 : a.step_full[go]);
        if constexpr (GL355_NTT_KO & 64) out[go + r * 16] = v;      // padded row stride (timing experiment: channel camping?)
        else out[go] = v;
    };
    if constexpr (LOG_T >= 4 && GL355_NTT_R8_DIRECT != 0 && EPT == 8) {
        uint64_t x[16];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t g = tid + i * NT;
            const uint64_t gi = ((uint64_t)(g >> LOG_TC) << log_n2) + c0 + (g & (TC - 1));
            x[i] = (GL355_NTT_KO & 2) ? gi : in[gi];
            if constexpr (PRE && !(GL355_NTT_KO & 32)) x[i] = gl_mul(x[i], (GL355_NTT_KO & 16) ? gi + 3 : pre[gi]);
        }
        dif_tile_r8_regs<LT, LOG_T, INV, true>(x, lds, a.tw_r8, LOG_TC, tid, NT);
        if constexpr (GL355_NTT_KO & (16 | 32 | 64)) dif_last_round_sink<LT, R8Last<LOG_T>::RHO, INV>(lds, LOG_TC, tid, NT, store);
        else {
            auto addr = [&](uint32_t g) { return ((uint64_t)(g >> LOG_TC) << log_n2) + c0 + (g & (TC - 1)); };
            dif_last_round_sink_tab<LT, R8Last<LOG_T>::RHO, INV, 0>(lds, LOG_TC, tid, NT, [&](uint32_t g) { return a.step_full[addr(g)]; },
                                                                   [&](uint32_t g, uint64_t v, uint64_t w) { out[addr(g)] = gl_mul(v, w); });
        }
    } else {
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            const uint32_t g = tid + i * NT;
            const uint32_t r = g >> LOG_TC, cc = g & (TC - 1);
            const uint64_t gi = ((uint64_t)r << log_n2) + c0 + cc;
            uint64_t v = (GL355_NTT_KO & 2) ? gi : in[gi];
            if constexpr (PRE && !(GL355_NTT_KO & 32)) v = gl_mul(v, (GL355_NTT_KO & 16) ? gi + 3 : pre[gi]);
            lds[lds_phys(g)] = v;
        }
        __syncthreads();
        dif_tile_r8<LT, LOG_T, INV>(lds, a.tw_r8, LOG_TC, tid, NT);
#pragma unroll
        for (int i = 0; i < EPT; i++) store(tid + i * NT, lds[lds_phys(tid + i * NT)]);
    }
}

// ------------------------------------------------------------------------------------------------
// Natural order in, natural order out in TWO passes (round 3): N = N1 * N2, input index i = i1 * N2 + i2, output k = k1 + N1 * k2.
//   pass A  column pass over i1 (tile = all N1 rows x TC adjacent i2), the 4-step twiddle omega_N^(k1 i2), and the tile leaves
//           TRANSPOSED: element (k1, i2) goes to intermediate position i2 * N1 + k1 -- runs of N1 contiguous words, read from LDS with the
//           row index bit-reversed, so the bit reversal of this dimension costs nothing in HBM;
//   pass B  column pass over i2 on that [N2][N1] matrix (tile = all N2 rows x TC adjacent k1), rows written at bitrev: position
//           k2 * N1 + k1 = the natural index.
// The DIF-DIF flow above leaves bit-reversed order (what the commitment wants) and needed a third pass (bitrev_tiled_kernel, 0.59 of
// 1.99 ms at 2^20 x 135) for natural order.  Both tiles are 2^LT elements with TC >= 8 (64-byte segments): N1, N2 <= 2^(LT - 3).
// step_t: the 4-step twiddles in pass A's STORE order, step_t[i2 * N1 + k1] = omega^(k1 i2) (Ctx::nat_step_table).
// ------------------------------------------------------------------------------------------------
template <int LT, int LOG_T, int PASS, int WPE>
__global__ void __launch_bounds__(1 << (LT - 3)) __attribute__((amdgpu_waves_per_eu(WPE))) ntt_cols_r8_nat_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    constexpr int NT = 1 << (LT - 3);
    constexpr int LOG_TC = LT - LOG_T, TC = 1 << LOG_TC;
    static_assert(LOG_TC >= 3, "tiles of at least 8 adjacent columns");
    const int tid = threadIdx.x;
    const uint32_t log_n2 = a.log_rows;                     // log2 of the row stride of the matrix this pass walks
    const uint64_t tiles_per_col = (1ull << log_n2) >> LOG_TC;
    // A tile of 8 columns reads / writes 64-byte halves of 128-byte lines; the tile with the other halves must run on the SAME XCD (its
    // L2), soon after: blocks are dealt to the XCDs round-robin, so block 8 q + x takes tile 2 (8 (q >> 1) + x) + (q & 1)
    uint64_t bt = blockIdx.x;
    if constexpr (TC * 8 < 128) {
        if ((gridDim.x & 15) == 0) { const uint64_t q = bt >> 3, xcd = bt & 7; bt = ((((q >> 1) << 3) + xcd) << 1) + (q & 1); }
    }
    const uint64_t col = bt / tiles_per_col;
    const uint64_t c0 = (bt % tiles_per_col) << LOG_TC;
    const uint64_t* in = a.in + col * a.in_col_stride;
    uint64_t* out = a.out + col * a.out_col_stride;
    uint64_t x[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t g = tid + i * NT;
        x[i] = in[((uint64_t)(g >> LOG_TC) << log_n2) + c0 + (g & (TC - 1))];
    }
    if constexpr (PASS == 0) {
        dif_tile_r8_regs<LT, LOG_T, false, false, 1>(x, lds, a.tw_r8, LOG_TC, tid, NT);
        // out[(c0 + cc) * N1 + k1] = tile[bitrev(k1)][cc] * omega^(k1 (c0 + cc)): lanes run over k1
        uint64_t sw[8];                      // the eight step words first: a load behind a store waits for that store (see dif_last_round_sink_tab)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t g = tid + i * NT;
            sw[i] = a.step_full[((c0 + (g >> LOG_T)) << LOG_T) + (g & ((1u << LOG_T) - 1))];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t g = tid + i * NT;
            const uint32_t k1 = g & ((1u << LOG_T) - 1), cc = g >> LOG_T;
            const uint64_t go = ((c0 + cc) << LOG_T) + k1;
            out[go] = gl_mul(lds[lds_phys_t<1>((brev(k1, LOG_T) << LOG_TC) | cc)], sw[i]);
        }
    } else {
        auto store = [&](uint32_t g, uint64_t v) {
            const uint32_t r = g >> LOG_TC, cc = g & (TC - 1);
            out[((uint64_t)brev(r, LOG_T) << log_n2) + c0 + cc] = gl_canon(v);
        };
        dif_tile_r8_regs<LT, LOG_T, false, true, 1>(x, lds, a.tw_r8, LOG_TC, tid, NT);
        dif_last_round_sink<LT, R8Last<LOG_T>::RHO, false, decltype(store), 1>(lds, LOG_TC, tid, NT, store);
    }
}
hipError_t launch_cols_r8_nat(const PassArgs& a, uint32_t log_t, int pass, hipStream_t s);

// Column pass of an LDE whose coset bases form a geometric sequence (base[c] = shift * w^c: PolynomialCoeffs::lde): ONE block takes a
// tile through ALL cosets.  The per-coset kernel above reads the coefficients once per coset (8 x 141 MB through the fabric at
// 2^17 x 135) and a 1-MiB pre table per coset; here the coefficients and two tables (base[0]^i and w^i) are read once, the value of
// coset c + 1 is the value of coset c times w^i (the same one product per element and coset), and only the outputs stream.
template <int LOG_T, int WPE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE))) ntt_cols_r8_cosets_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    constexpr int LT = 12, NT = 512;
    constexpr int LOG_TC = LT - LOG_T, TC = 1 << LOG_TC;
    const int tid = threadIdx.x;
    const uint32_t log_n2 = a.log_rows;
    const uint64_t n2 = 1ull << log_n2;
    const uint64_t tiles_per_col = n2 >> LOG_TC;
    const uint64_t col = blockIdx.x / tiles_per_col;
    const uint64_t c0 = (blockIdx.x % tiles_per_col) << LOG_TC;
    const uint64_t* in = a.in + col * a.in_col_stride;
    uint64_t v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t g = tid + i * NT;
        const uint64_t gi = ((uint64_t)(g >> LOG_TC) << log_n2) + c0 + (g & (TC - 1));
        v[i] = gl_mul(in[gi], a.pre_full[gi]);
    }
    const CosetSlots slots = coset_slots_of(a);
    for (uint32_t c = 0; c < a.n_cosets; c++) {
        if (c) {        // the ratio table is re-read per coset (L2-resident, coalesced) rather than held: 16 VGPRs less, no spills
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t g = tid + i * NT;
                v[i] = gl_mul(v[i], a.ratio_full[((uint64_t)(g >> LOG_TC) << log_n2) + c0 + (g & (TC - 1))]);
            }
        }
        uint64_t* out = a.out + (uint64_t)coset_slot_at(slots, c) * a.coset_out_stride + col * a.out_col_stride;
        auto store = [&](uint32_t g, uint64_t val) {
            const uint64_t go = ((uint64_t)(g >> LOG_TC) << log_n2) + c0 + (g & (TC - 1));
            out[go] = gl_mul(val, a.step_full[go]);
        };
        if constexpr (LOG_T >= 4 && GL355_NTT_R8_DIRECT != 0) {
            uint64_t x[16];
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = v[i];
            dif_tile_r8_regs<LT, LOG_T, false, true>(x, lds, a.tw_r8, LOG_TC, tid, NT);
            dif_last_round_sink<LT, R8Last<LOG_T>::RHO, false>(lds, LOG_TC, tid, NT, store);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) lds[lds_phys(tid + i * NT)] = v[i];
            __syncthreads();
            dif_tile_r8<LT, LOG_T, false>(lds, a.tw_r8, LOG_TC, tid, NT);
#pragma unroll
            for (int i = 0; i < 8; i++) store(tid + i * NT, lds[lds_phys(tid + i * NT)]);
        }
        __syncthreads();
    }
}

}  // namespace gl355
