// The 24-bit-limb LDE passes (csrc/ntt_l24.cuh): launchers and the two table builders.  Compiled with the 15-instruction inline-asm field
// product like ntt_r8.hip (the coset-ratio products of the column pass and the 128-bit reductions of the limb products use it).
// a2 / a3 of SURVEY.md 8; reference call sites as in ntt.hip (plonky2 coset_fft_with_options inside CircuitData::prove, access_set.rs:94).
#define GL_MUL_VARIANT 1
#include "gl355_internal.h"
#include "ntt_l24.cuh"
#include <algorithm>

namespace gl355 {

// mid[64 u + v] = omega_4096^(bitrev6(u) v): cell (u, v) of the row tile holds output kA = bitrev6(u) of the first radix-64
// super-round (ntt_rows_l24_kernel)
__global__ void build_mid_kernel(uint64_t root4096, uint64_t* out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 4096) return;
    const uint32_t u = g >> 6, v = g & 63;
    out[g] = gl_canon(gl_pow(root4096, (uint64_t)(__brev(u) >> 26) * v));
}

int32_t Ctx::l24_mid_table(const uint64_t** out) {
    const std::vector<uint64_t> key{3, 12};
    auto it = full_cache.find(key);
    if (it != full_cache.end()) { *out = it->second; return GL355_OK; }
    uint64_t* d = nullptr;
    GL355_HIP(this, hipMalloc((void**)&d, 4096 * 8));
    hipLaunchKernelGGL(build_mid_kernel, dim3(16), dim3(256), 0, stream, gl_root_of_unity(12), d);
    GL355_HIP(this, hipGetLastError());
    full_cache[key] = d;
    *out = d;
    return GL355_OK;
}

// rows of 4096 points: a.batch << a.log_rows of them (forward, natural order in, bit-reversed canonical out, no multiplier tables).
// The split-exchange kernel, one row per block, 33 KB of LDS.  (The persistent limb-quad kernel of the first version, ntt_rows_l24_kernel, and the
// prefetching variants stay in ntt_l24.cuh for tools/ubench/ubench_ntt_l24.hip: profiles/r03_ubench_ntt_l24s.txt, r05_ubench_ntt_l24s.txt.)
hipError_t launch_rows_l24(const PassArgs& a, hipStream_t s) {
    const uint64_t total = ((uint64_t)a.batch) << a.log_rows;
    auto k = ntt_rows_l24s_kernel<5, false>;
    hipLaunchKernelGGL(k, dim3((uint32_t)total), dim3(512), L24S_ROWS_LDS_BYTES, s, a);
    return hipGetLastError();
}
// 32-point column pass over all cosets: blocks over (column, 64-column tile)
hipError_t launch_cols_l24_cosets(const PassArgs& a, hipStream_t s) {
    const uint64_t blocks = ((1ull << a.log_rows) >> 6) * a.batch;
    // ratio table held in registers (116 VGPRs, 4 tiles per CU): no load follows a store inside the coset loop (MODE 1 of the kernel; the variants
    // that re-read it per coset or take one coset per block measured slower: profiles/r03_ubench_ntt_l24s.txt (4))
    hipLaunchKernelGGL((ntt_cols_l24s_cosets_kernel<6, 4, 1>), dim3((uint32_t)blocks), dim3(256), 32 * 64 * 8, s, a);
    return hipGetLastError();
}

// column dimension 2 or 4 (log_rows == 12): the streaming kernel
hipError_t launch_cols_small_cosets(const PassArgs& a, uint32_t log_t, hipStream_t s) {
    const dim3 grid((1u << a.log_rows) / 256u, a.batch);
    if (log_t == 1) hipLaunchKernelGGL(ntt_cols_small_cosets_kernel<1>, grid, dim3(256), 0, s, a);
    else if (log_t == 2) hipLaunchKernelGGL(ntt_cols_small_cosets_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(ntt_cols_small_cosets_kernel<3>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace gl355
