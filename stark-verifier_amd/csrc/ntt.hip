// Batched radix-2 Goldilocks NTT / inverse NTT / coset NTT / LDE for gfx950 (a2, a3, a5 of SURVEY.md 8).
//
// Replaces plonky2_field::fft::{fft_with_options, ifft_with_options}, PolynomialCoeffs::{lde,
// coset_fft_with_options}, PolynomialValues::{ifft, coset_ifft} and plonky2_util::{transpose,
// reverse_index_bits_in_place}, reached from the reference inside CircuitBuilder::build /
// CircuitData::prove (src/plonky2_semaphore/access_set.rs:91,94; recursion.rs:167-168).
// Conventions pinned by the reference's verifier: omega_N = 7^((p-1)/N) (chip/fri_chip.rs:162-163),
// coset generator 7 (chip/plonk/plonk_verifier_chip.rs:225-227), bit-reversed evaluation order
// (chip/fri_chip.rs:245-264).
//
// Design (MI355X-first, not plonky2's layer-by-layer loop):
//   * one workgroup owns a TILE of 2^LT elements (LT = 12..14) held in LDS (160 KiB/CU);
//     each thread keeps 16 elements in VGPRs and runs radix-16 decimation-in-frequency rounds
//     (4 butterfly layers per LDS round-trip), so a 2^12 tile needs 3 LDS exchanges, not 12;
//   * sizes up to 2^14 are ONE pass over HBM; larger sizes use the 4-step split N = N1*N2:
//     a "column" pass (tiles of N1 rows x TC adjacent columns, every global access a full
//     TC*8-byte segment) and a "row" pass (contiguous rows), each a single read + single write;
//   * DIF produces bit-reversed order for free, which is exactly the Merkle-leaf order the FRI
//     commit wants, so the commit path never runs a separate permutation or transpose;
//   * coset scaling, the 1/n of the inverse transform and the 4-step twiddles are fused into the
//     load / store of the passes (two-level power tables: g^e = lo[e & 4095] * hi[e >> 12]);
//   * the LDE runs the 2^rate_bits cosets as independent size-n transforms (the first rate_bits
//     layers of the zero-padded size-N transform are trivial); coset c's output is the contiguous
//     block bitrev(c) of the bit-reversed result, and workgroup id % n_cosets selects the coset so
//     that with 8 cosets each XCD's L2 keeps exactly one coset's power table.
// Field products: the compiler-scheduled formulation (gl_field.cuh, GL_MUL_VARIANT 2).  These kernels run 2-4 waves per SIMD
// between LDS exchanges, so what matters is that the 16 independent butterflies of a thread interleave freely; the
// inline-asm formulation has fewer instructions but serialises each product (measured: 4-7 % slower single-pass tiles).
#ifndef GL_MUL_VARIANT
#define GL_MUL_VARIANT 2
#endif
#include "gl355_internal.h"
#include "ntt_kernels.cuh"

namespace gl355 {

// the 24-bit-limb passes of the LDE (ntt_l24.hip)
hipError_t launch_rows_l24(const PassArgs& a, hipStream_t s);
hipError_t launch_cols_l24_cosets(const PassArgs& a, hipStream_t s);
hipError_t launch_cols_small_cosets(const PassArgs& a, uint32_t log_t, hipStream_t s);
// out[c][i] = in[c][bitrev(i)] over 2^log_n entries of `width` u64 each (a5:
// reverse_index_bits_in_place; out-of-place, or in place via swap when in == out).
__global__ void bitrev_permute_kernel(const uint64_t* in, uint64_t* out, uint32_t log_n, uint32_t width,
                                      uint64_t in_col_stride, uint64_t out_col_stride, uint32_t batch) {
    const uint64_t n = 1ull << log_n;
    const uint64_t total = n * batch;
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < total;
         g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t col = g >> log_n, i = g & (n - 1);
        const uint64_t j = brev((uint32_t)i, log_n);
        if (in == out) {
            if (i < j) {
                for (uint32_t w = 0; w < width; w++) {
                    uint64_t* pa = out + col * out_col_stride + i * width + w;
                    uint64_t* pb = out + col * out_col_stride + j * width + w;
                    uint64_t ta = *pa, tb = *pb;
                    *pa = tb; *pb = ta;
                }
            }
        } else {
            for (uint32_t w = 0; w < width; w++)
                out[col * out_col_stride + i * width + w] = in[col * in_col_stride + j * width + w];
        }
    }
}

// The same permutation for width == 1 and log_n >= 10 at HBM speed: index i = (a | m | b) with 5-bit a, b maps to
// (rev b | rev m | rev a), so for a fixed middle part m the 32 x 32 block over (a, b) is read as 32 contiguous 256-byte rows,
// transposed (with both coordinates bit-reversed) through LDS, and written as 32 contiguous rows of block rev(m).  In place,
// block m and block rev(m) are exchanged by one workgroup (the one with m <= rev m).  The element-wise kernel above scatters
// 8-byte accesses and takes longer than the transform it follows (3.7 ms vs 2.8 ms at 2^20 x 135).
// (64 x 64 tiles -- 512-byte rows on both sides, 66 KB of LDS per block -- measured twice as slow: 1.33 vs 0.61 ms at 2^20 x 135.)
__global__ void __launch_bounds__(256) bitrev_tiled_kernel(const uint64_t* in, uint64_t* out, uint32_t log_n, uint64_t in_col_stride,
                                                         uint64_t out_col_stride) {
    __shared__ uint64_t t0[32][33], t1[32][33];
    const uint32_t mid_bits = log_n - 10;
    // Block order (round 5): a tile's 32 rows are 2^(log_n - 5) words apart -- 1 MB at 2^22 points -- so a block touches 32 pages on the read side
    // and 32 on the write side, and with m = blockIdx the write side of neighbouring blocks (rev m) is scattered over the whole column: 1.8 TB/s at
    // 2^22, 1.7 at 2^23 against 2.5 at 2^21.  256 consecutive blocks now take every combination of the LOW four and the HIGH four bits of m: their
    // reads (low bits vary) and their writes (rev of the high bits varies) both fall into 32 runs of 4 KB.
    uint32_t m = blockIdx.x;
    if (mid_bits >= 8) m = ((m >> 4 & 15u) << (mid_bits - 4)) | ((m >> 8) << 4) | (m & 15u);
    const uint32_t rm = brev(m, mid_bits);
    const bool in_place = in == out;
    if (in_place && m > rm) return;
    const uint64_t col = blockIdx.y;
    const uint64_t* src = in + col * in_col_stride;
    uint64_t* dst = out + col * out_col_stride;
    const uint32_t x = threadIdx.x & 31, y0 = threadIdx.x >> 5;
    const uint32_t hi_shift = log_n - 5;
    const bool both = in_place && m != rm;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t a = y0 + 8 * k;
        t0[a][x] = src[((uint64_t)a << hi_shift) | ((uint64_t)m << 5) | x];
        if (both) t1[a][x] = src[((uint64_t)a << hi_shift) | ((uint64_t)rm << 5) | x];
    }
    __syncthreads();
    const uint32_t rx = brev(x, 5);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t bp = y0 + 8 * k;                 // output row = rev(b), output column x = rev(a)
        dst[((uint64_t)bp << hi_shift) | ((uint64_t)rm << 5) | x] = t0[rx][brev(bp, 5)];
        if (both) dst[((uint64_t)bp << hi_shift) | ((uint64_t)m << 5) | x] = t1[rx][brev(bp, 5)];
    }
}

// out[r][c] = in[c][perm(r)] : column-major [cols][rows] -> row-major [rows][cols], optionally
// reading row bitrev(r) (a5: transpose; used to export plonky2-layout leaves).  32x32 LDS tiles.
__global__ void transpose_kernel(const uint64_t* in, uint64_t* out, uint64_t rows, uint32_t cols,
                                 uint64_t in_col_stride, uint32_t out_row_stride, uint32_t log_rows_brev) {
    __shared__ uint64_t tile[32][33];
    const uint64_t r0 = (uint64_t)blockIdx.x * 32;
    const uint32_t c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const uint32_t c = c0 + k;
        const uint64_t r = r0 + tx;
        if (c < cols && r < rows) {
            const uint64_t rr = log_rows_brev ? brev((uint32_t)r, log_rows_brev) : r;
            tile[k][tx] = in[(uint64_t)c * in_col_stride + rr];
        }
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const uint64_t r = r0 + k;
        const uint32_t c = c0 + tx;
        if (c < cols && r < rows) out[r * out_row_stride + c] = tile[tx][k];
    }
}

// full-size multiplier tables (built once per (bases, size), cached on the context): they trade one
// modmul + one gather per element for a coalesced 8-byte read that stays L2-resident per XCD
__global__ void build_pow_table_kernel(const uint64_t* lo, const uint64_t* hi, uint32_t n_cosets, uint64_t n, uint64_t* out) {
    const uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (g >= n * n_cosets) return;
    const uint64_t c = g / n, i = g % n;
    out[g] = gl_canon(gl_mul(lo[c * 4096 + (i & 4095)], hi[c * 4096 + (i >> 12)]));
}
// step_full[r * N2 + i2] = omega^(bitrev(r, l1) * i2): the 4-step twiddle in the column pass's store order
__global__ void build_step_table_kernel(const uint64_t* lo, const uint64_t* hi, uint32_t l1, uint32_t l2, uint64_t* out) {
    const uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (g >= (1ull << (l1 + l2))) return;
    const uint64_t r = g >> l2, i2 = g & ((1ull << l2) - 1);
    const uint64_t e = (uint64_t)brev((uint32_t)r, l1) * i2;
    out[g] = gl_canon(gl_mul(lo[e & 4095], hi[e >> 12]));
}

// step_t[i2 * N1 + k1] = omega^(k1 * i2): the 4-step twiddles in the store order of pass A of the natural -> natural flow
__global__ void build_nat_step_table_kernel(const uint64_t* lo, const uint64_t* hi, uint32_t l1, uint32_t l2, uint64_t* out) {
    const uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (g >= (1ull << (l1 + l2))) return;
    const uint64_t i2 = g >> l1, k1 = g & ((1ull << l1) - 1);
    const uint64_t e = k1 * i2;
    out[g] = gl_canon(gl_mul(lo[e & 4095], hi[e >> 12]));
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <int LT, int LOG_T>
static hipError_t launch_rows_lt(const PassArgs& a, bool inv, uint64_t blocks, hipStream_t s) {
    const size_t shmem = ((1u << LT) + (1u << (LT - 4))) * sizeof(uint64_t);
    constexpr int NT = 1 << (LT - 4);
    // (the FAST instantiation and 4 waves per SIMD measured no better for rows: HISTORY rounds 2-3)
    if (inv) {
        auto k = ntt_rows_kernel<LT, LOG_T, true>;
        if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(NT), shmem, s, a);
    } else {
        auto k = ntt_rows_kernel<LT, LOG_T, false>;
        if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(NT), shmem, s, a);
    }
    return hipGetLastError();
}

static hipError_t launch_rows(const PassArgs& a, uint32_t log_t, bool inv, hipStream_t s) {
    // 4096-point rows of the two-pass LDE / forward transform on 24-bit limbs (a.mid is set by ntt_run for exactly that shape)
    if (a.mid && !inv && log_t == 12 && a.n_cosets == 1 && !a.pre_full && !a.pre_lo && !a.post_lo && a.scale == 1 && a.canon && !a.in_bitrev &&
        !a.out_natural)
        return launch_rows_l24(a, s);
    // the commit-path shape (forward, whole rows of 2^12..2^14 points in natural order, at most a full pre table) runs the
    // radix-8 kernel
    if (!inv && log_t >= 12 && log_t <= 14 && !a.post_lo && a.scale == 1 && a.canon && !a.in_bitrev && !a.out_natural &&
        (a.pre_full || !a.pre_lo || !a.pre_hi)) {
        PassArgs b = a;
        if (!b.pre_full && b.pre_lo) { b.pre_full = b.pre_lo; b.pre_full_stride = 4096; }   // n <= 2^12: the one-level table is the full one
        return launch_rows_r8(b, log_t, false, s);
    }
    // ... and so do the inverse forms without multiplier tables (values -> coefficients: natural order within the row and the 1/n at the store)
    if (inv && log_t >= 12 && log_t <= 14 && !a.post_lo && !a.pre_lo && !a.pre_full && a.canon && !a.in_bitrev)
        return launch_rows_r8(a, log_t, true, s);
    const uint64_t total_rows = ((uint64_t)a.batch) << a.log_rows;
    const int lt = log_t <= 12 ? 12 : (int)log_t;
    const uint64_t rpt = 1ull << (lt - log_t);
    const uint64_t blocks = ((total_rows + rpt - 1) / rpt) * a.n_cosets;
    switch (log_t) {
#define GL355_ROW_CASE(L) case L: return launch_rows_lt<12, L>(a, inv, blocks, s);
        GL355_ROW_CASE(1) GL355_ROW_CASE(2) GL355_ROW_CASE(3) GL355_ROW_CASE(4) GL355_ROW_CASE(5)
        GL355_ROW_CASE(6) GL355_ROW_CASE(7) GL355_ROW_CASE(8) GL355_ROW_CASE(9) GL355_ROW_CASE(10)
        GL355_ROW_CASE(11) GL355_ROW_CASE(12)
#undef GL355_ROW_CASE
        case 13: return launch_rows_lt<13, 13>(a, inv, blocks, s);
        case 14: return launch_rows_lt<14, 14>(a, inv, blocks, s);
        default: return hipErrorInvalidValue;
    }
}

template <int LOG_T>
static hipError_t launch_cols_t(const PassArgs& a, bool inv, uint64_t blocks, hipStream_t s) {
    const size_t shmem = (4096 + 256) * sizeof(uint64_t);
    const bool fast = !inv && a.pre_full && a.step_full && !a.in_bitrev && !a.out_natural && !a.post_lo && a.scale == 1 && !a.canon;
    const bool r8 = a.step_full && (inv ? !a.pre_lo && !a.pre_full : (a.pre_full || !a.pre_lo)) && !a.in_bitrev && !a.out_natural && !a.post_lo &&
                    a.scale == 1 && !a.canon;
    if (r8 && !inv && a.ratio_full && a.pre_full && LOG_T == 5 && a.log_rows == 12) return launch_cols_l24_cosets(a, s);
    // column dimension 2, 4 or 8 (n = 2^13 .. 2^15 with 4096-point rows): the streaming kernel, no tile (profiles/r03b_single_pass_ab.txt)
    if (r8 && !inv && a.ratio_full && a.pre_full && LOG_T <= 3 && a.log_rows == 12 && a.batch <= 65535)
        return launch_cols_small_cosets(a, LOG_T, s);
    if (r8 && !inv && a.ratio_full && a.pre_full) return launch_cols_r8_cosets(a, LOG_T, s);
    if (r8) return launch_cols_r8(a, LOG_T, inv, s);
    if (fast) hipLaunchKernelGGL((ntt_cols_kernel<LOG_T, false, true>), dim3((uint32_t)blocks), dim3(256), shmem, s, a);
    else if (inv) hipLaunchKernelGGL((ntt_cols_kernel<LOG_T, true>), dim3((uint32_t)blocks), dim3(256), shmem, s, a);
    else hipLaunchKernelGGL((ntt_cols_kernel<LOG_T, false>), dim3((uint32_t)blocks), dim3(256), shmem, s, a);
    return hipGetLastError();
}

static hipError_t launch_cols(const PassArgs& a, uint32_t log_t, bool inv, hipStream_t s) {
    const uint64_t n2 = 1ull << a.log_rows;
    const uint64_t tc = 1ull << (12 - log_t);
    const uint64_t blocks = (n2 / tc) * a.batch * a.n_cosets;
    switch (log_t) {
#define GL355_COL_CASE(L) case L: return launch_cols_t<L>(a, inv, blocks, s);
        GL355_COL_CASE(1) GL355_COL_CASE(2) GL355_COL_CASE(3) GL355_COL_CASE(4) GL355_COL_CASE(5)
        GL355_COL_CASE(6) GL355_COL_CASE(7) GL355_COL_CASE(8) GL355_COL_CASE(9) GL355_COL_CASE(10)
        GL355_COL_CASE(11) GL355_COL_CASE(12)
#undef GL355_COL_CASE
        default: return hipErrorInvalidValue;
    }
}

// Split of a two-pass transform: N = N1 * N2, first-pass dimension N1 (columns kernel in the
// natural->bitrev flow, rows kernel in the bitrev->natural flow).
static void split_two_pass(uint32_t log_n, bool natural_in, uint32_t* log_n1, uint32_t* log_n2) {
    // the column pass works on tiles of 2^(12 - LOG_T) adjacent columns: with 4096-point rows a 2^22 / 2^23-point transform leaves it
    // 4 / 2 columns (32- / 16-byte accesses: 570 / 450 GB/s against 860 at 2^21), so from 2^22 on the row dimension is 2^14
    const uint32_t row = log_n >= 22 ? 14 : 12;
    if (natural_in) {  // cols over N1 (<= 2^12, prefers >= 16 cols per tile), rows over N2
        *log_n1 = log_n - row; *log_n2 = row;
    } else {           // rows over N1 (contiguous), then cols over N2 (<= 2^12)
        *log_n1 = row; *log_n2 = log_n - row;
    }
}

int32_t ntt_run(Ctx* ctx, const NttPlan& p) {
    if (p.log_n == 0) {
        if (p.in != p.out) GL355_HIP(ctx, hipMemcpyAsync(p.out, p.in, sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
        return GL355_OK;
    }
    if (p.log_n > 24) return ctx->fail(GL355_E_UNSUPPORTED, "ntt: log_n > 24 unsupported");
    if (p.n_cosets > 16 || p.n_cosets == 0) return ctx->fail(GL355_E_INVALID_ARG, "ntt: bad coset count");
    const bool inv = p.inverse;
    PassArgs a;
    memset(&a, 0, sizeof a);
    a.batch = p.batch;
    a.n_cosets = p.n_cosets;
    a.coset_out_stride = p.coset_out_stride;
    for (uint32_t c = 0; c < 16; c++) a.coset_slot[c] = c < p.n_cosets ? p.coset_slot[c] : 0;
    a.log_n = p.log_n;
    a.tw = ctx->twr(4, inv);
    a.tw_r8 = ctx->twr(3, inv);
    a.scale = 1;

    // n = 2^13, 2^14 in one pass need a 64 / 128-KB LDS tile and every VGPR of the CU: nothing else can share the CU with such
    // a block.  GL355_OPT_NTT_SINGLE_PASS_MAX_LOG < 14 sends the commit-path shape (natural in, bit-reversed out, no
    // post-multiplier, contiguous coset blocks) of those sizes through the two-pass path with 4096-point tiles instead.
    const bool two_pass_ok = !p.in_bitrev && p.out_bitrev && !p.post_lo && (p.n_cosets == 1 || p.coset_out_stride == (1ull << p.log_n));
    if (p.log_n <= 12 || (p.log_n <= 14 && !(two_pass_ok && p.log_n > ctx->ntt_single_pass_max_log))) {
        // single pass over HBM
        a.in = p.in; a.out = p.out;
        a.in_col_stride = p.in_col_stride; a.out_col_stride = p.out_col_stride;
        a.log_rows = 0;
        a.pre_lo = p.pre_lo; a.pre_hi = p.log_n > 12 ? p.pre_hi : nullptr;
        if (p.pre_lo && p.log_n > 12) {  // two-level lookups would cost an extra modmul per element
            GL355_TRY(ctx->full_pow_table(p.pre_lo, p.pre_hi, p.n_cosets, p.log_n, &a.pre_full));
            a.pre_full_stride = 1ull << p.log_n;
        }
        a.post_lo = p.post_lo; a.post_hi = p.log_n > 12 ? p.post_hi : nullptr;
        a.scale = p.scale;
        a.in_bitrev = p.in_bitrev; a.out_natural = p.out_bitrev ? 0 : 1;
        a.canon = 1;
        ProfScope ps(ctx, "ntt_rows_single_pass", ((uint64_t)p.batch << p.log_n) * 8 * (1 + p.n_cosets));
        GL355_HIP(ctx, launch_rows(a, p.log_n, inv, ctx->stream));
        return GL355_OK;
    }

    // two passes; 4-step twiddle omega_N^(+-e) tables
    const uint64_t* step_lo; const uint64_t* step_hi;
    int32_t rc = ctx->pow_tables(inv ? gl_inv(gl_root_of_unity(p.log_n)) : gl_root_of_unity(p.log_n), &step_lo, &step_hi);
    if (rc) return rc;
    uint32_t l1, l2;
    // natural order in AND out, plain forward transform of 2^12 < n <= 2^20 points: two column-type passes with the transposition in LDS
    // (ntt_cols_r8_nat_kernel) instead of two passes + the bit-reversal pass (A/B: profiles/r03_ntt_natural_two_pass.txt)
    if (!inv && !p.in_bitrev && !p.out_bitrev && p.n_cosets == 1 && !p.pre_lo && !p.post_lo && p.scale == 1 &&
        p.log_n <= 20 && p.coset_slot[0] == 0) {
        l2 = p.log_n / 2; l1 = p.log_n - l2;               // l1 >= l2; both <= 10
        Scratch mid(ctx);
        GL355_TRY(mid.get(((uint64_t)p.batch << p.log_n) * 8));
        const uint64_t n = 1ull << p.log_n;
        a.in = p.in; a.in_col_stride = p.in_col_stride;
        a.out = mid.as<uint64_t>(); a.out_col_stride = n;
        a.log_rows = l2;                                   // pass A walks the [N1][N2] matrix: row stride N2
        GL355_TRY(ctx->nat_step_table(step_lo, step_hi, l1, l2, &a.step_full));
        { ProfScope ps(ctx, "ntt_cols_pass1", ((uint64_t)p.batch << p.log_n) * 8); GL355_HIP(ctx, launch_cols_r8_nat(a, l1, 0, ctx->stream)); }
        PassArgs b = a;
        b.in = mid.as<uint64_t>(); b.in_col_stride = n;
        b.out = p.out; b.out_col_stride = p.out_col_stride;
        b.log_rows = l1;                                   // pass B walks the [N2][N1] matrix: row stride N1
        b.step_full = nullptr;
        { ProfScope ps(ctx, "ntt_cols_pass2", ((uint64_t)p.batch << p.log_n) * 8); GL355_HIP(ctx, launch_cols_r8_nat(b, l2, 1, ctx->stream)); }
        return GL355_OK;
    }
    if (!p.in_bitrev) {
        // natural in -> (cols over N1, twiddle at store) -> (rows over N2) -> bit-reversed out
        split_two_pass(p.log_n, true, &l1, &l2);
        // From 2^22 points on, THREE passes over 4096-element tiles beat two with 16384-point rows (one block per CU, 4 waves per SIMD):
        // after the column pass over N1 = N / 2^17 the rows are independent 2^17-point transforms, each the usual column + row pair.
        // Forward, single coset, contiguous output columns, no pre-multiplier (plain forward NTT: cfg-2 (C)); 2^22 x 16: 1.44 -> see DESIGN 4.1
        const bool three = !inv && p.log_n >= 22 && p.n_cosets == 1 && !p.pre_lo && !p.post_lo && p.scale == 1 &&
                           p.out_col_stride == (1ull << p.log_n) && p.coset_slot[0] == 0;
        // Round 5: from 2^21 points on the same shape runs in TWO passes with 4096-point limb rows: the column pass over 2^(log_n - 12) = 2^9 .. 2^11
        // points on 8192- / 16384-element tiles (16 columns per tile at 2^21 / 2^22, 8 at 2^23: launch_cols_r8_big) -- 32 bytes of HBM traffic per
        // element instead of the 48 of three passes.  2^23 (2^11-point columns, 16384-element tiles with the CU to themselves) measured slower than
        // three passes and stays there (every split measured: profiles/r05_ntt_big_ab.txt).
        const bool big_shape = !inv && p.log_n >= 21 && p.log_n <= 22 && p.n_cosets == 1 && !p.pre_lo && !p.post_lo && p.scale == 1 &&
                               p.out_col_stride == (1ull << p.log_n) && p.coset_slot[0] == 0;
        const uint32_t big_lt = big_shape ? 13 : 0;        // 8192-element column tiles: 16 columns per tile at 2^21, 8 at 2^22
        const uint32_t big_row = 12;
        const bool three_now = three && !big_lt;
        if (big_lt) { l2 = big_row; l1 = p.log_n - big_row; }
        else if (three) { l2 = 17; l1 = p.log_n - l2; }
        a.in = p.in; a.out = p.out;
        a.in_col_stride = p.in_col_stride; a.out_col_stride = p.out_col_stride;
        a.log_rows = l2;  // log2(N2): row stride of the N1 x N2 matrix
        a.pre_lo = p.pre_lo; a.pre_hi = p.pre_hi;
        a.step_lo = step_lo; a.step_hi = step_hi;
        if (p.log_n <= 20 || three_now || big_lt) {
            // full-size tables (<= 8 MiB each; the three-pass form: up to 128 MiB of 288 GiB): 1 load + 1 modmul per element instead
            // of 2 + 2 (beyond 2^20 measured a wash for the radix-16 kernels: 2^21 slower, 2^22 / 2^23 +2-3 %)
            if (p.pre_lo && ((uint64_t)p.n_cosets << p.log_n) <= (1ull << 21)) {
                GL355_TRY(ctx->full_pow_table(p.pre_lo, p.pre_hi, p.n_cosets, p.log_n, &a.pre_full));
                a.pre_full_stride = 1ull << p.log_n;
            }
            GL355_TRY(ctx->full_step_table(step_lo, step_hi, l1, l2, inv, &a.step_full));
            if (a.pre_full && p.n_cosets > 1 && p.coset_ratio) {
                const uint64_t *rlo, *rhi;
                GL355_TRY(ctx->pow_tables(p.coset_ratio, &rlo, &rhi));
                GL355_TRY(ctx->full_pow_table(rlo, rhi, 1, p.log_n, &a.ratio_full));
            }
        }
        a.in_bitrev = 0; a.out_natural = 0; a.canon = 0;
        {
            ProfScope ps(ctx, "ntt_cols_pass1", ((uint64_t)p.batch << p.log_n) * 8);
            if (big_lt) GL355_HIP(ctx, launch_cols_r8_big(a, l1, big_lt, ctx->stream));
            else GL355_HIP(ctx, launch_cols(a, l1, inv, ctx->stream));
        }
        if (three_now) {
            NttPlan sub;
            sub.in = p.out; sub.out = p.out;
            sub.in_col_stride = sub.out_col_stride = 1ull << l2;
            sub.log_n = l2; sub.batch = p.batch << l1;
            sub.inverse = false; sub.in_bitrev = false; sub.out_bitrev = true;
            GL355_TRY(ntt_run(ctx, sub));
            if (!p.out_bitrev) GL355_TRY(bitrev_permute(ctx, p.out, p.out, p.log_n, 1, p.out_col_stride, p.out_col_stride, p.batch));
            return GL355_OK;
        }
        PassArgs b = a;
        b.in = p.out; b.in_col_stride = p.out_col_stride;
        // each coset's intermediate lives in its own output block: rows pass runs per coset slot
        b.pre_lo = b.pre_hi = nullptr; b.step_lo = b.step_hi = nullptr; b.pre_full = nullptr; b.step_full = nullptr; b.ratio_full = nullptr;
        if (!inv && l2 == 12 && p.scale == 1) GL355_TRY(ctx->l24_mid_table(&b.mid));
        b.log_rows = l1;  // rows per column = N1
        b.scale = p.scale; b.canon = 1;
        if (p.n_cosets == 1) {
            b.in = p.out + (uint64_t)a.coset_slot[0] * a.coset_out_stride;
            ProfScope ps(ctx, "ntt_rows_pass2", ((uint64_t)p.batch << p.log_n) * 8);
            GL355_HIP(ctx, launch_rows(b, l2, inv, ctx->stream));
        } else {
            // the coset blocks are contiguous sub-ranges of every output column: treat (column,
            // coset) pairs as 2^(l1) * n_cosets rows per column -- requires the blocks to tile the
            // column, which is how the LDE lays them out (coset_out_stride == n).
            if (p.coset_out_stride != (1ull << p.log_n))
                return ctx->fail(GL355_E_INVALID_ARG, "ntt: multi-coset two-pass needs contiguous coset blocks");
            uint32_t lc = 0;
            while ((1u << lc) < p.n_cosets) lc++;
            b.n_cosets = 1; b.coset_slot[0] = 0; b.coset_out_stride = 0;
            b.log_rows = l1 + lc;
            ProfScope ps(ctx, "ntt_rows_pass2", ((uint64_t)p.batch * p.n_cosets << p.log_n) * 8);
            GL355_HIP(ctx, launch_rows(b, l2, inv, ctx->stream));
        }
        if (!p.out_bitrev) {
            // natural order requested: one extra permutation pass (not used on the commit path)
            if (p.n_cosets != 1) return ctx->fail(GL355_E_UNSUPPORTED, "ntt: natural-order multi-coset output is done by the caller");
            GL355_TRY(bitrev_permute(ctx, p.out, p.out, p.log_n, 1, p.out_col_stride, p.out_col_stride, p.batch));
        }
        if (p.post_lo) return ctx->fail(GL355_E_UNSUPPORTED, "ntt: post-multiplier needs natural-order output from the bitrev-input flow");
        return GL355_OK;
    }
    // bit-reversed in -> (rows over N1, natural within row) -> (cols over N2, twiddle at load) -> natural out
    if (p.n_cosets != 1) return ctx->fail(GL355_E_UNSUPPORTED, "ntt: bitrev-input flow is single-coset");
    if (p.out_bitrev) return ctx->fail(GL355_E_UNSUPPORTED, "ntt: bitrev -> bitrev not provided");
    split_two_pass(p.log_n, false, &l1, &l2);
    a.in = p.in; a.out = p.out;
    a.in_col_stride = p.in_col_stride; a.out_col_stride = p.out_col_stride;
    a.log_rows = l2;  // rows per column = N2 (row index = bitrev(i2))
    a.in_bitrev = 1; a.out_natural = 1; a.canon = 0;
    { ProfScope ps(ctx, "ntt_bitrev_in_rows_pass1", ((uint64_t)p.batch << p.log_n) * 8); GL355_HIP(ctx, launch_rows(a, l1, inv, ctx->stream)); }
    PassArgs b = a;
    b.in = p.out; b.in_col_stride = p.out_col_stride;
    b.log_rows = l1;  // matrix is [N2 rows][N1 columns]
    b.step_lo = step_lo; b.step_hi = step_hi;
    b.post_lo = p.post_lo; b.post_hi = p.post_hi;
    b.scale = p.scale; b.canon = 1;
    b.in_bitrev = 1; b.out_natural = 1;
    ProfScope ps(ctx, "ntt_bitrev_in_cols_pass2", ((uint64_t)p.batch << p.log_n) * 8);
    GL355_HIP(ctx, launch_cols(b, l2, inv, ctx->stream));
    return GL355_OK;
}

int32_t bitrev_permute(Ctx* ctx, const uint64_t* in, uint64_t* out, uint32_t log_n, uint32_t width,
                       uint64_t in_col_stride, uint64_t out_col_stride, uint32_t batch) {
    const uint64_t total = ((uint64_t)batch) << log_n;
    if (total == 0) return GL355_OK;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256 * 16);
    ProfScope ps(ctx, "bitrev_permute", total * 16);
    if (width == 1 && log_n >= 10 && batch <= 65535) {
        hipLaunchKernelGGL(bitrev_tiled_kernel, dim3(1u << (log_n - 10), batch), dim3(256), 0, ctx->stream, in, out, log_n, in_col_stride, out_col_stride);
        GL355_HIP(ctx, hipGetLastError());
        return GL355_OK;
    }
    hipLaunchKernelGGL(bitrev_permute_kernel, dim3(blocks), dim3(256), 0, ctx->stream, in, out, log_n, width,
                       in_col_stride, out_col_stride, batch);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

int32_t transpose_cols_to_rows(Ctx* ctx, const uint64_t* in, uint64_t* out, uint64_t rows, uint32_t cols,
                               uint64_t in_col_stride, uint32_t out_row_stride, uint32_t log_rows_brev) {
    if (rows == 0 || cols == 0) return GL355_OK;
    dim3 grid((uint32_t)((rows + 31) / 32), (cols + 31) / 32);
    ProfScope ps(ctx, "transpose", rows * cols * 16);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, ctx->stream, in, out, rows, cols, in_col_stride,
                       out_row_stride, log_rows_brev);
    GL355_HIP(ctx, hipGetLastError());
    return GL355_OK;
}

int32_t Ctx::full_pow_table(const uint64_t* lo, const uint64_t* hi, uint32_t n_cosets, uint32_t log_n, const uint64_t** out) {
    const std::vector<uint64_t> key{1, (uint64_t)(uintptr_t)lo, n_cosets, log_n};
    auto it = full_cache.find(key);
    if (it != full_cache.end()) { *out = it->second; return GL355_OK; }
    const uint64_t total = (uint64_t)n_cosets << log_n;
    uint64_t* d = nullptr;
    GL355_HIP(this, hipMalloc((void**)&d, total * 8));
    hipLaunchKernelGGL(build_pow_table_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, lo, hi, n_cosets, 1ull << log_n, d);
    GL355_HIP(this, hipGetLastError());
    full_cache[key] = d;
    *out = d;
    return GL355_OK;
}
int32_t Ctx::full_step_table(const uint64_t* lo, const uint64_t* hi, uint32_t l1, uint32_t l2, bool inv, const uint64_t** out) {
    const std::vector<uint64_t> key{2, l1, l2, inv ? 1ull : 0ull};
    auto it = full_cache.find(key);
    if (it != full_cache.end()) { *out = it->second; return GL355_OK; }
    const uint64_t total = 1ull << (l1 + l2);
    uint64_t* d = nullptr;
    GL355_HIP(this, hipMalloc((void**)&d, total * 8));
    hipLaunchKernelGGL(build_step_table_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, lo, hi, l1, l2, d);
    GL355_HIP(this, hipGetLastError());
    full_cache[key] = d;
    *out = d;
    return GL355_OK;
}

int32_t Ctx::nat_step_table(const uint64_t* lo, const uint64_t* hi, uint32_t l1, uint32_t l2, const uint64_t** out) {
    const std::vector<uint64_t> key{5, l1, l2};
    auto it = full_cache.find(key);
    if (it != full_cache.end()) { *out = it->second; return GL355_OK; }
    const uint64_t total = 1ull << (l1 + l2);
    uint64_t* d = nullptr;
    GL355_HIP(this, hipMalloc((void**)&d, total * 8));
    hipLaunchKernelGGL(build_nat_step_table_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, lo, hi, l1, l2, d);
    GL355_HIP(this, hipGetLastError());
    full_cache[key] = d;
    *out = d;
    return GL355_OK;
}

int32_t ntt_init_constants(Ctx* ctx) {
    uint64_t w16[2][8];
    const uint64_t w = gl_root_of_unity(4), wi = gl_inv(w);
    uint64_t x = 1, y = 1;
    for (int j = 0; j < 8; j++) { w16[0][j] = gl_canon(x); w16[1][j] = gl_canon(y); x = gl_mul(x, w); y = gl_mul(y, wi); }
    GL355_HIP(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(c_w16), w16, sizeof w16, 0, hipMemcpyHostToDevice, ctx->stream));
    GL355_HIP(ctx, ctx->wait());
    return GL355_OK;
}

}  // namespace gl355
