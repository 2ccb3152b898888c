// Timing experiments on the 24-bit-limb LDE passes as shipped (csrc/ntt_l24.cuh), outside the library: the row pass (135 x 256 rows of
// 4096 points, in place) and the all-cosets column pass of the 2^17 -> 2^20 x 135 LDE.  Random operands and table words (only the access
// pattern matters for the time; parity is the library's tests').  Build variants with -DGL355_L24_KO=mask:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I stark-verifier_amd/csrc -DGL_MUL_VARIANT=1 tools/ubench/ubench_ntt_l24.hip -o ...
#ifndef GL_MUL_VARIANT
#define GL_MUL_VARIANT 1
#endif
#include "ntt_l24.cuh"
#include <cstdio>
#include <cstdlib>
using namespace gl355;
namespace gl355 {   // declared by the headers for the library build; not used here
hipError_t launch_rows_r8(const PassArgs&, uint32_t, bool, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_cols_r8(const PassArgs&, uint32_t, bool, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_cols_r8_cosets(const PassArgs&, uint32_t, hipStream_t) { return hipErrorNotSupported; }
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
    printf("KO=%-2d %-24s %.3f ms  %.0f GB/s moved\n", GL355_L24_KO, what, ms, gb / ms * 1e3);
}
template <int WPE> static void rows(const PassArgs& a, uint32_t blocks, double gb) {
    auto k = ntt_rows_l24_kernel<WPE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L24_ROWS_LDS_BYTES));
    char name[64]; snprintf(name, sizeof name, "rows 2^12 l24 wpe%d", WPE);
    report(name, timeit([&] { hipLaunchKernelGGL(k, dim3(512), dim3(512), L24_ROWS_LDS_BYTES, 0, a); }, 10), gb);   // persistent: 2 blocks per CU
}
template <int WPE, bool PF> static void rows_s(const PassArgs& a, uint32_t grid, double gb) {
    auto k = ntt_rows_l24s_kernel<WPE, PF>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L24S_ROWS_LDS_BYTES));
    char name[64]; snprintf(name, sizeof name, "rows 2^12 l24s wpe%d pf%d g%u", WPE, (int)PF, grid);
    report(name, timeit([&] { hipLaunchKernelGGL(k, dim3(grid), dim3(512), L24S_ROWS_LDS_BYTES, 0, a); }, 10), gb);
}
template <int WPE> static void cols(const PassArgs& c, uint32_t blocks, double gb) {
    auto k = ntt_cols_l24_cosets_kernel<WPE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L24_COLS_LDS_BYTES));
    char name[64]; snprintf(name, sizeof name, "cols 2^5 l24 cosets wpe%d", WPE);
    report(name, timeit([&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(512), L24_COLS_LDS_BYTES, 0, c); }, 10), gb);
}

template <int LOG_TC, int WPE, int MODE = 0> static void cols_s(const PassArgs& c, uint32_t batch, double gb) {
    auto k = ntt_cols_l24s_cosets_kernel<LOG_TC, WPE, MODE>;
    const size_t lds = (size_t)32 * 8 << LOG_TC;
    const uint32_t blocks = (4096u >> LOG_TC) * batch;
    char name[64]; snprintf(name, sizeof name, "cols 2^5 l24s tc%d wpe%d mode%d", 1 << LOG_TC, WPE, MODE);
    report(name, timeit([&] { hipLaunchKernelGGL(k, dim3(blocks, MODE == 2 ? c.n_cosets : 1), dim3(4 << LOG_TC), lds, 0, c); }, 30), gb);
}
// shader clock under load: a one-wave kernel on a second stream spins for ~ticks of the 100-MHz real-time counter and reports the
// shader-clock cycles (s_memtime) that passed meanwhile
__global__ void clock_probe_kernel(uint64_t ticks, uint64_t* out) {
    const uint64_t w0 = wall_clock64(), c0 = clock64();
    uint64_t w1 = w0;
    while (w1 - w0 < ticks) { __builtin_amdgcn_s_sleep(8); w1 = wall_clock64(); }
    const uint64_t c1 = clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
template <typename F> static void clocked(const char* what, F f, int reps) {
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    uint64_t* d; CK(hipMalloc(&d, 16)); uint64_t h[2];
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) { f(); if (i == reps / 2) hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s2, 40000, d); }
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float t; CK(hipEventElapsedTime(&t, a, b));
    CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("%-40s %4d reps  %.3f ms each   shader clock %.0f MHz (probe window %.2f ms)\n", what, reps, t / reps, h[0] / (h[1] / 100.0), h[1] / 1e5);
    CK(hipFree(d)); CK(hipStreamDestroy(s2));
}
int main(int argc, char** argv) {
    const uint32_t batch = argc > 1 ? atoi(argv[1]) : 135;
    const uint64_t N = 1ull << 20, n = 1ull << 17;
    uint64_t *buf, *cin, *pre, *ratio, *step, *mid;
    CK(hipMalloc(&buf, (batch * N + (1 << 20)) * 8));
    CK(hipMalloc(&cin, batch * n * 8));
    CK(hipMalloc(&pre, 8 * n * 8)); CK(hipMalloc(&ratio, n * 8));
    CK(hipMalloc(&step, n * 8)); CK(hipMalloc(&mid, 4096 * 8));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, buf, batch * N);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, cin, batch * n);
    hipLaunchKernelGGL(fill_kernel, dim3(512), dim3(256), 0, 0, pre, 8 * n);
    hipLaunchKernelGGL(fill_kernel, dim3(512), dim3(256), 0, 0, ratio, n);
    hipLaunchKernelGGL(fill_kernel, dim3(512), dim3(256), 0, 0, step, n);
    hipLaunchKernelGGL(fill_kernel, dim3(16), dim3(256), 0, 0, mid, 4096);
    CK(hipDeviceSynchronize());
    PassArgs a; memset(&a, 0, sizeof a);
    a.in = buf; a.out = buf; a.in_col_stride = N; a.out_col_stride = N; a.batch = batch; a.n_cosets = 1; a.log_n = 17; a.log_rows = 8;
    a.scale = 1; a.canon = 1; a.mid = mid;
    const double gb = 2.0 * batch * N * 8 / 1e9;
    rows<2>(a, batch * 256, gb); rows<3>(a, batch * 256, gb); rows<4>(a, batch * 256, gb);
    rows_s<4, false>(a, batch * 256, gb); rows_s<5, false>(a, batch * 256, gb); rows_s<6, false>(a, batch * 256, gb); rows_s<8, false>(a, batch * 256, gb);
    rows_s<6, false>(a, 768, gb); rows_s<8, false>(a, 1024, gb);
    rows_s<4, true>(a, 512, gb); rows_s<5, true>(a, 512, gb); rows_s<6, true>(a, 768, gb); rows_s<8, true>(a, 1024, gb);
    PassArgs c; memset(&c, 0, sizeof c);
    c.in = cin; c.out = buf; c.in_col_stride = n; c.out_col_stride = N; c.batch = batch; c.n_cosets = 8; c.coset_out_stride = n;
    for (int i = 0; i < 8; i++) c.coset_slot[i] = i;
    c.log_n = 17; c.log_rows = 12; c.pre_full = pre; c.pre_full_stride = n; c.ratio_full = ratio; c.step_full = step; c.scale = 1;
    const double cgb = (batch * n + batch * N) * 8.0 / 1e9;
    cols<2>(c, 32 * batch, cgb); cols<3>(c, 32 * batch, cgb); cols<4>(c, 32 * batch, cgb);
    cols_s<7, 4>(c, batch, cgb); cols_s<7, 5>(c, batch, cgb); cols_s<7, 6>(c, batch, cgb);
    cols_s<6, 4>(c, batch, cgb); cols_s<6, 5>(c, batch, cgb); cols_s<6, 6>(c, batch, cgb); cols_s<6, 8>(c, batch, cgb);
    cols_s<6, 4, 1>(c, batch, cgb); cols_s<6, 5, 1>(c, batch, cgb); cols_s<7, 4, 1>(c, batch, cgb);
    cols_s<6, 4, 2>(c, batch, cgb); cols_s<6, 6, 2>(c, batch, cgb); cols_s<6, 8, 2>(c, batch, cgb); cols_s<7, 6, 2>(c, batch, cgb); cols_s<7, 8, 2>(c, batch, cgb);
    CK(hipDeviceSynchronize());
    {   // the shipped pair alone and alternating, short and long: what the device's clock does under each
        auto kr = ntt_rows_l24s_kernel<5, false>; auto kc = ntt_cols_l24s_cosets_kernel<6, 5>;
        auto fr = [&] { hipLaunchKernelGGL(kr, dim3(batch * 256), dim3(512), L24S_ROWS_LDS_BYTES, 0, a); };
        auto fc = [&] { hipLaunchKernelGGL(kc, dim3(64 * batch), dim3(256), 32 * 64 * 8, 0, c); };
        for (int reps : {10, 200}) {
            clocked("rows l24s alone", fr, reps);
            clocked("cols l24s alone", fc, reps);
            clocked("cols + rows alternating (per pair)", [&] { fc(); fr(); }, reps);
        }
        // per-kernel times inside the alternating sequence
        hipEvent_t e[3]; for (auto& x : e) CK(hipEventCreate(&x));
        float tc = 0, tr = 0;
        for (int i = 0; i < 50; i++) {
            CK(hipEventRecord(e[0])); fc(); CK(hipEventRecord(e[1])); fr(); CK(hipEventRecord(e[2])); CK(hipEventSynchronize(e[2]));
            float x, y; CK(hipEventElapsedTime(&x, e[0], e[1])); CK(hipEventElapsedTime(&y, e[1], e[2]));
            if (i >= 10) { tc += x; tr += y; }
        }
        printf("alternating, per kernel: cols %.3f ms  rows %.3f ms\n", tc / 40, tr / 40);
    }
    return 0;
}
