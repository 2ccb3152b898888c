// Round 6 A/B: the row pass of the 2^17 -> 2^20 x 135 LDE as shipped (ntt_rows_l24s_kernel<5, false>, one row per block) against the LDS-DMA variant
// (ntt_rows_l24d_kernel: persistent blocks, the next row fetched global -> LDS by global_load_lds_dwordx4 while this one is transformed, mid twiddles in
// registers).  Checks the two outputs word for word on random input first, then times both: alone (200 launches) and alternating with the column pass.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I stark-verifier_amd/csrc tools/ubench/ubench_ntt_l24d.hip -o tools/ubench/bin/ubench_ntt_l24d
#ifndef GL_MUL_VARIANT
#define GL_MUL_VARIANT 1
#endif
#include "ntt_l24.cuh"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gl355;
namespace gl355 {
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
__global__ void diff_kernel(const uint64_t* a, const uint64_t* b, uint64_t n, unsigned long long* bad) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (a[i] != b[i]) atomicAdd(bad, 1ull);
}
template <typename F> static float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 12; i++) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b));
    CK(hipGetLastError());
    return t / reps;
}
int main(int argc, char** argv) {
    const uint32_t batch = argc > 1 ? atoi(argv[1]) : 135;
    const uint64_t N = 1ull << 20, n = 1ull << 17;
    uint64_t *in, *o1, *o2, *cin, *pre, *ratio, *step, *mid;
    unsigned long long* bad;
    CK(hipMalloc(&in, batch * N * 8)); CK(hipMalloc(&o1, batch * N * 8)); CK(hipMalloc(&o2, batch * N * 8));
    CK(hipMalloc(&cin, batch * n * 8)); CK(hipMalloc(&pre, 8 * n * 8)); CK(hipMalloc(&ratio, n * 8)); CK(hipMalloc(&step, n * 8)); CK(hipMalloc(&mid, 4096 * 8));
    CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, in, batch * N);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, cin, batch * n);
    hipLaunchKernelGGL(fill_kernel, dim3(512), dim3(256), 0, 0, pre, 8 * n);
    hipLaunchKernelGGL(fill_kernel, dim3(512), dim3(256), 0, 0, ratio, n);
    hipLaunchKernelGGL(fill_kernel, dim3(512), dim3(256), 0, 0, step, n);
    hipLaunchKernelGGL(fill_kernel, dim3(16), dim3(256), 0, 0, mid, 4096);
    CK(hipDeviceSynchronize());
    PassArgs a; memset(&a, 0, sizeof a);
    a.in = in; a.out = o1; a.in_col_stride = N; a.out_col_stride = N; a.batch = batch; a.n_cosets = 1; a.log_n = 17; a.log_rows = 8;
    a.scale = 1; a.canon = 1; a.mid = mid;
    PassArgs d = a; d.out = o2;
    auto ks = ntt_rows_l24s_kernel<5, false>;
    CK(hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L24S_ROWS_LDS_BYTES));
    auto kd = ntt_rows_l24d_kernel<4>;
    int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kd, 512, 0));
    int dev = 0, n_cu = 256; CK(hipGetDevice(&dev)); CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    printf("l24d: %d block(s) per CU resident, %d CUs\n", per_cu, n_cu);
    const uint32_t rows = batch * 256;
    auto fs = [&](const PassArgs& x) { hipLaunchKernelGGL(ks, dim3(rows), dim3(512), L24S_ROWS_LDS_BYTES, 0, x); };
    auto fd = [&](const PassArgs& x, uint32_t grid) { hipLaunchKernelGGL(kd, dim3(grid), dim3(512), 0, 0, x); };
    fs(a); fd(d, 2 * n_cu);
    hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, 0, o1, o2, batch * N, bad);
    unsigned long long hb = 0; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    printf("l24d vs l24s on %llu words: %llu differ%s\n", (unsigned long long)(batch * N), hb, hb ? "  <-- MISMATCH" : "  (bit-exact)");
    // a second grid shape (uneven tail) must agree as well
    CK(hipMemset(o2, 0, batch * N * 8)); CK(hipMemset(bad, 0, 8));
    fd(d, 2 * n_cu - 37);
    hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, 0, o1, o2, batch * N, bad);
    CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    printf("l24d (grid %d) vs l24s: %llu differ\n", 2 * n_cu - 37, hb);
    const double gb = 2.0 * batch * N * 8 / 1e9;
    for (int round = 0; round < 2; round++) {
        float ts = timeit([&] { fs(a); }, 200);
        printf("rows l24s (shipped)        %.3f ms  %.0f GB/s moved\n", ts, gb / ts * 1e3);
        for (uint32_t g : {(uint32_t)n_cu, (uint32_t)(2 * n_cu), (uint32_t)(3 * n_cu), (uint32_t)(4 * n_cu)}) {
            float td = timeit([&] { fd(d, g); }, 200);
            printf("rows l24d (LDS-DMA) g=%-5u %.3f ms  %.0f GB/s moved\n", g, td, gb / td * 1e3);
        }
    }
    {   // the limb-quad exchange, one row per block (65-KB tile, 5 barriers)
        PassArgs qa = a; qa.out = o2;
        CK(hipMemset(o2, 0, batch * N * 8)); CK(hipMemset(bad, 0, 8));
        auto run_q = [&](auto kq, const char* name) {
            CK(hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L24_ROWS_LDS_BYTES));
            hipLaunchKernelGGL(kq, dim3(rows), dim3(512), L24_ROWS_LDS_BYTES, 0, qa);
            CK(hipMemset(bad, 0, 8));
            hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, 0, o1, o2, batch * N, bad);
            CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
            float tq = timeit([&] { hipLaunchKernelGGL(kq, dim3(rows), dim3(512), L24_ROWS_LDS_BYTES, 0, qa); }, 200);
            printf("rows %s (quad cells, one row per block)  %.3f ms  %.0f GB/s moved   vs l24s: %llu words differ\n", name, tq, gb / tq * 1e3, hb);
        };
        run_q(ntt_rows_l24q_kernel<4>, "l24q wpe4");
        run_q(ntt_rows_l24q_kernel<5>, "l24q wpe5");
        run_q(ntt_rows_l24q_kernel<3>, "l24q wpe3");
    }
    // in the LDE: column pass then row pass, per pair
    PassArgs c; memset(&c, 0, sizeof c);
    c.in = cin; c.out = in; c.in_col_stride = n; c.out_col_stride = N; c.batch = batch; c.n_cosets = 8; c.coset_out_stride = n;
    for (int i = 0; i < 8; i++) c.coset_slot[i] = i;
    c.log_n = 17; c.log_rows = 12; c.pre_full = pre; c.pre_full_stride = n; c.ratio_full = ratio; c.step_full = step; c.scale = 1;
    auto kc = ntt_cols_l24s_cosets_kernel<6, 4, 1>;
    auto fc = [&] { hipLaunchKernelGGL(kc, dim3(64 * batch), dim3(256), 32 * 64 * 8, 0, c); };
    PassArgs ai = a; ai.out = in;          // in place, as the library runs it
    PassArgs di = d; di.out = in;
    float p1 = timeit([&] { fc(); fs(ai); }, 200), p2 = timeit([&] { fc(); fd(di, 2 * n_cu); }, 200);
    printf("cols + rows l24s per LDE   %.3f ms\ncols + rows l24d per LDE   %.3f ms\n", p1, p2);
    return 0;
}
