// Timing experiments on the NTT pass kernels as shipped (csrc/ntt_kernels.cuh), outside the library:
//   * the two passes of the 2^17 -> 2^20 x 135 LDE (column pass over 32-point transforms, row pass over 4096-point rows, in place),
//   * the in-proof single-pass LDEs (8 units x 135 columns of 2^13 / 2^14 coefficients -> 8 cosets),
// radix-16 against radix-8 kernels.  Random operands (the clocks follow the data's toggle rate); the twiddle tables hold random words
// too -- only the access pattern matters for the time.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I stark-verifier_amd/csrc [-DGL_MUL_VARIANT=1|2] [-DGL355_NTT_KO=mask] tools/ubench/ubench_ntt_rows.hip
// GL355_NTT_KO knocks parts of the kernels out (ntt_kernels.cuh; results are wrong then, only the time means something): this is how
// the scattered twiddle gathers of the first version were found to cost a third of the row pass.
#ifndef GL_MUL_VARIANT
#define GL_MUL_VARIANT 2
#endif
#include "ntt_kernels.cuh"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gl355;
namespace gl355 {   // declared by the header for the library build; not used here
hipError_t launch_rows_r8(const PassArgs&, uint32_t, bool, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_cols_r8(const PassArgs&, uint32_t, bool, hipStream_t) { return hipErrorNotSupported; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(uint64_t* p, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = z % GL_P;
    }
}
template <typename F> static float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b));
    CK(hipGetLastError());
    return t / reps;
}
static void report(const char* what, float ms, double gb) {
    printf("MV=%d KO=%-2d %-28s %.3f ms  %.0f GB/s\n", GL_MUL_VARIANT, GL355_NTT_KO, what, ms, gb / ms * 1e3);
}

template <int LT>
static void single_pass(uint64_t* tw, uint64_t* pre, uint64_t* cin, uint64_t* buf) {
    const uint32_t cols = 8 * 135;
    const uint64_t n = 1ull << LT;
    PassArgs a; memset(&a, 0, sizeof a);
    a.in = cin; a.out = buf; a.in_col_stride = n; a.out_col_stride = 8 * n; a.batch = cols; a.n_cosets = 8; a.coset_out_stride = n;
    for (int i = 0; i < 8; i++) a.coset_slot[i] = i;
    a.log_n = LT; a.log_rows = 0; a.tw = tw; a.tw_r8 = tw; a.pre_full = pre; a.pre_full_stride = n; a.scale = 1; a.canon = 1;
    const uint32_t blocks = cols * 8;
    const size_t sh = ((1u << LT) + (1u << (LT - 4))) * 8;
    const double gb = 9.0 * cols * n * 8 / 1e9;   // algorithmic: coefficients in, 8 cosets out
    char name[64];
    { auto k = ntt_rows_kernel<LT, LT, false, false, 1>; CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
      snprintf(name, sizeof name, "single 2^%d r16", LT);
      report(name, timeit([&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(1 << (LT - 4)), sh, 0, a); }, 10), gb); }
    { auto k = ntt_rows_r8_kernel<LT, true, 4>; CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
      snprintf(name, sizeof name, "single 2^%d r8 wpe4", LT);
      report(name, timeit([&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), sh, 0, a); }, 10), gb); }
    if (LT == 13) { auto k = ntt_rows_r8_kernel<LT, true, 8>; CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
      snprintf(name, sizeof name, "single 2^%d r8 wpe8", LT);
      report(name, timeit([&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), sh, 0, a); }, 10), gb); }
}

int main(int argc, char** argv) {
    const uint32_t batch = argc > 1 ? atoi(argv[1]) : 135;
    const uint64_t N = 1ull << 20, n = 1ull << 17;
    const uint64_t big = std::max<uint64_t>(batch * N, 8ull * 135 * 8 * (1ull << 14));
    uint64_t *buf, *tw, *cin, *pre, *step;
    CK(hipMalloc(&buf, (big + (1 << 20)) * 8));
    CK(hipMalloc(&tw, (1 << 15) * 8));   // stands in for the round-major twiddle table
    CK(hipMalloc(&cin, std::max<uint64_t>(batch * n, 8ull * 135 * (1ull << 14)) * 8));
    CK(hipMalloc(&pre, 8 * n * 8));
    CK(hipMalloc(&step, n * 8));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf, big);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, tw, 1ull << 15);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, cin, std::max<uint64_t>(batch * n, 8ull * 135 * (1ull << 14)));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, pre, 8 * n);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, step, n);
    CK(hipDeviceSynchronize());

    // row pass of the LDE (pass 2): 135 x 256 rows of 4096 points, in place
    PassArgs a; memset(&a, 0, sizeof a);
    a.in = buf; a.out = buf; a.in_col_stride = N; a.out_col_stride = N; a.batch = batch; a.n_cosets = 1; a.log_n = 17; a.log_rows = 8;
    a.tw = tw; a.tw_r8 = tw; a.scale = 1; a.canon = 1;
    const uint32_t blocks = batch * 256;
    const size_t sh = (4096 + 256) * 8;
    const double gb = 2.0 * batch * N * 8 / 1e9;   // bytes moved
    report("rows 2^12 r16 wpe3", timeit([&] { hipLaunchKernelGGL((ntt_rows_kernel<12, 12, false, false, 3>), dim3(blocks), dim3(256), sh, 0, a); }, 10), gb);
    report("rows 2^12 r8 wpe6", timeit([&] { hipLaunchKernelGGL((ntt_rows_r8_kernel<12, false, 6>), dim3(blocks), dim3(512), sh, 0, a); }, 10), gb);
    report("rows 2^12 r8 wpe8", timeit([&] { hipLaunchKernelGGL((ntt_rows_r8_kernel<12, false, 8>), dim3(blocks), dim3(512), sh, 0, a); }, 10), gb);

    // column pass of the LDE (pass 1): 2^17 coefficients per column -> 8 cosets, 32-point transforms over 128-column tiles
    PassArgs c; memset(&c, 0, sizeof c);
    c.in = cin; c.out = buf; c.in_col_stride = n; c.out_col_stride = N; c.batch = batch; c.n_cosets = 8; c.coset_out_stride = n;
    for (int i = 0; i < 8; i++) c.coset_slot[i] = i;
    c.log_n = 17; c.log_rows = 12; c.tw = tw; c.tw_r8 = tw; c.pre_full = pre; c.pre_full_stride = n; c.step_full = step; c.scale = 1;
    const uint32_t cblocks = 32 * batch * 8;
    const double cgb = (8.0 * batch * n + batch * N) * 8 / 1e9;
    report("cols 2^5 r16 fast", timeit([&] { hipLaunchKernelGGL((ntt_cols_kernel<5, false, true>), dim3(cblocks), dim3(256), sh, 0, c); }, 10), cgb);
    report("cols 2^5 r8 wpe4", timeit([&] { hipLaunchKernelGGL((ntt_cols_r8_kernel<5, true, 4>), dim3(cblocks), dim3(512), sh, 0, c); }, 10), cgb);
    report("cols 2^5 r8 wpe6", timeit([&] { hipLaunchKernelGGL((ntt_cols_r8_kernel<5, true, 6>), dim3(cblocks), dim3(512), sh, 0, c); }, 10), cgb);

    single_pass<13>(tw, pre, cin, buf);
    single_pass<14>(tw, pre, cin, buf);
    CK(hipDeviceSynchronize());
    return 0;
}
