// The radix-8 pass kernels of the commit path (csrc/ntt_kernels.cuh), compiled with the 15-instruction inline-asm field product:
// at 6-8 waves per SIMD its hazard wait states are hidden and the shorter instruction stream wins (ntt.hip keeps the
// compiler-scheduled product for the radix-16 kernels, which run at 3 waves per SIMD).  a2 / a3 of SURVEY.md 8; reference call
// sites as in ntt.hip (plonky2 fft_with_options / coset_fft_with_options inside CircuitData::prove, access_set.rs:94).
#define GL_MUL_VARIANT 1
#include "gl355_internal.h"
#include "ntt_kernels.cuh"
#include <cstdlib>

namespace gl355 {

template <int LT>
static hipError_t launch_rows8_inv(const PassArgs& a, uint64_t blocks, hipStream_t s) {       // inverse: no pre table
    constexpr int NT = LT >= 13 ? 1024 : 512;
    constexpr int WPE = LT <= 13 ? 8 : 4;
    const size_t shmem = ((1u << LT) + (1u << (LT - 4))) * sizeof(uint64_t);
    auto k = ntt_rows_r8_kernel<LT, false, WPE, true>;
    if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(NT), shmem, s, a);
    return hipGetLastError();
}
template <int LT>
static hipError_t launch_rows8_lt(const PassArgs& a, uint64_t blocks, hipStream_t s) {
    constexpr int NT = LT >= 13 ? 1024 : 512;
    // 8192-point tiles: two 1024-thread blocks per CU need <= 64 VGPRs (0.65 -> 0.53 ms
    // for 8 units' 2^13 -> 2^16 LDEs, a couple of spilled dwords included); a 16384-point tile has the CU to itself
    constexpr int WPE = LT <= 13 ? 8 : 4;                     // 4096-point tiles: four 512-thread blocks per CU (0.79 -> 0.76 ms against three at 68 VGPRs)
    const size_t shmem = ((1u << LT) + (1u << (LT - 4))) * sizeof(uint64_t);
    if (a.pre_full) {
        auto k = ntt_rows_r8_kernel<LT, true, WPE>;
        if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(NT), shmem, s, a);
    } else {
        auto k = ntt_rows_r8_kernel<LT, false, WPE>;
        if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(NT), shmem, s, a);
    }
    return hipGetLastError();
}

// rows of 2^log_t points (log_t = 12, 13, 14), a.batch << a.log_rows of them, times a.n_cosets
hipError_t launch_rows_r8(const PassArgs& a, uint32_t log_t, bool inv, hipStream_t s) {
    const uint64_t blocks = (((uint64_t)a.batch) << a.log_rows) * a.n_cosets;
    if (inv) {
        switch (log_t) {
            case 12: return launch_rows8_inv<12>(a, blocks, s);
            case 13: return launch_rows8_inv<13>(a, blocks, s);
            case 14: return launch_rows8_inv<14>(a, blocks, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (log_t) {
        case 12: return launch_rows8_lt<12>(a, blocks, s);
        case 13: return launch_rows8_lt<13>(a, blocks, s);
        case 14: {
            // a 16384-point tile at 4 waves per SIMD needs an EMPTY CU (all its VGPRs and 136 KB of LDS): alone it is the fastest form,
            // under a many-context load the blocks wait for CUs to drain (rocprof: 6 ms resident for 0.4 ms of work).
            // (8 waves per SIMD, trading registers for the chance to share a CU, measured no better: HISTORY round 3)
            return launch_rows8_lt<14>(a, blocks, s);
        }
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_cols_r8(const PassArgs& a, uint32_t log_t, bool inv, hipStream_t s) {
    const uint64_t n2 = 1ull << a.log_rows;
    const uint64_t tc = 1ull << (12 - log_t);
    const uint64_t blocks = (n2 / tc) * a.batch * a.n_cosets;
    const size_t shmem = (4096 + 256) * sizeof(uint64_t);
    switch (log_t) {
#define GL355_COL8_CASE(L) case L:                                                                                            \
        if (inv) hipLaunchKernelGGL((ntt_cols_r8_kernel<L, false, 4, true>), dim3((uint32_t)blocks), dim3(512), shmem, s, a);  \
        else if (a.pre_full) hipLaunchKernelGGL((ntt_cols_r8_kernel<L, true, 4>), dim3((uint32_t)blocks), dim3(512), shmem, s, a);  \
        else hipLaunchKernelGGL((ntt_cols_r8_kernel<L, false, 4>), dim3((uint32_t)blocks), dim3(512), shmem, s, a);            \
        break;
        GL355_COL8_CASE(1) GL355_COL8_CASE(2) GL355_COL8_CASE(3) GL355_COL8_CASE(4) GL355_COL8_CASE(5) GL355_COL8_CASE(6)
        GL355_COL8_CASE(7) GL355_COL8_CASE(8) GL355_COL8_CASE(9) GL355_COL8_CASE(10) GL355_COL8_CASE(11) GL355_COL8_CASE(12)
#undef GL355_COL8_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// forward column passes over 2^9 .. 2^11 points on 8192- / 16384-element tiles (lt = 13, 14): the first pass of a 2^21 .. 2^23-point transform whose
// second pass is 4096-point rows.  lt = 13: two 1024-thread blocks per CU at <= 64 VGPRs; lt = 14: the 136-KB tile has the CU to itself.
template <int LOG_T, int LT>
static hipError_t launch_cols8_big_t(const PassArgs& a, hipStream_t s) {
    constexpr int WPE = LT == 13 ? 8 : 4;
    const size_t shmem = ((1u << LT) + (1u << (LT - 4))) * sizeof(uint64_t);
    const uint64_t blocks = ((1ull << a.log_rows) >> (LT - LOG_T)) * a.batch * a.n_cosets;
    if (a.pre_full) {
        auto k = ntt_cols_r8_kernel<LOG_T, true, WPE, false, LT>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(1024), shmem, s, a);
    } else {
        auto k = ntt_cols_r8_kernel<LOG_T, false, WPE, false, LT>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(1024), shmem, s, a);
    }
    return hipGetLastError();
}
hipError_t launch_cols_r8_big(const PassArgs& a, uint32_t log_t, uint32_t lt, hipStream_t s) {
    if (lt == 13) {
        switch (log_t) {
            case 9: return launch_cols8_big_t<9, 13>(a, s);
            case 10: return launch_cols8_big_t<10, 13>(a, s);
            default: return hipErrorInvalidValue;
        }
    }
    if (lt == 14) {
        switch (log_t) {
            case 10: return launch_cols8_big_t<10, 14>(a, s);
            case 11: return launch_cols8_big_t<11, 14>(a, s);
            default: return hipErrorInvalidValue;
        }
    }
    return hipErrorInvalidValue;
}

// natural -> natural in two passes (ntt_kernels.cuh): pass 0 over a.log_n - a.log_rows... the caller sets a.log_rows = log2 of the row stride;
// tiles of 4096 elements while 8 adjacent columns fit (log_t <= 9), 8192 elements for log_t = 10
template <int LT, int LOG_T, int PASS>
static hipError_t launch_nat_t(const PassArgs& a, hipStream_t s) {
    constexpr int NT = 1 << (LT - 3);
    constexpr int WPE = LT == 12 ? 4 : 8;                  // 8192-element tiles: two 1024-thread blocks per CU need <= 64 VGPRs
    const size_t shmem = ((1u << LT) + (1u << (LT - 4)) + (1u << (LT - 8))) * sizeof(uint64_t);
    const uint64_t blocks = ((1ull << a.log_rows) >> (LT - LOG_T)) * a.batch;
    auto k = ntt_cols_r8_nat_kernel<LT, LOG_T, PASS, WPE>;
    if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k, dim3((uint32_t)blocks), dim3(NT), shmem, s, a);
    return hipGetLastError();
}
hipError_t launch_cols_r8_nat(const PassArgs& a, uint32_t log_t, int pass, hipStream_t s) {
    switch (log_t * 2 + (pass ? 1 : 0)) {
        case 6 * 2: return launch_nat_t<12, 6, 0>(a, s);
        case 6 * 2 + 1: return launch_nat_t<12, 6, 1>(a, s);
        case 7 * 2: return launch_nat_t<12, 7, 0>(a, s);
        case 7 * 2 + 1: return launch_nat_t<12, 7, 1>(a, s);
        case 8 * 2: return launch_nat_t<12, 8, 0>(a, s);
        case 8 * 2 + 1: return launch_nat_t<12, 8, 1>(a, s);
        case 9 * 2: return launch_nat_t<12, 9, 0>(a, s);
        case 9 * 2 + 1: return launch_nat_t<12, 9, 1>(a, s);
        case 10 * 2: return launch_nat_t<13, 10, 0>(a, s);
        case 10 * 2 + 1: return launch_nat_t<13, 10, 1>(a, s);
        default: return hipErrorInvalidValue;
    }
}

// all cosets of a tile in one block (a.ratio_full set): blocks over (column, tile)
hipError_t launch_cols_r8_cosets(const PassArgs& a, uint32_t log_t, hipStream_t s) {
    const uint64_t n2 = 1ull << a.log_rows;
    const uint64_t tc = 1ull << (12 - log_t);
    const uint64_t blocks = (n2 / tc) * a.batch;
    const size_t shmem = (4096 + 256) * sizeof(uint64_t);
    switch (log_t) {
#define GL355_COL8C_CASE(L) case L: hipLaunchKernelGGL((ntt_cols_r8_cosets_kernel<L, 4>), dim3((uint32_t)blocks), dim3(512), shmem, s, a); break;
        GL355_COL8C_CASE(1) GL355_COL8C_CASE(2) GL355_COL8C_CASE(3) GL355_COL8C_CASE(4) GL355_COL8C_CASE(5) GL355_COL8C_CASE(6)
        GL355_COL8C_CASE(7) GL355_COL8C_CASE(8) GL355_COL8C_CASE(9) GL355_COL8C_CASE(10) GL355_COL8C_CASE(11) GL355_COL8C_CASE(12)
#undef GL355_COL8C_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gl355
