// Does the 256-MiB Infinity Cache keep a freshly WRITTEN buffer for the next kernel's reads?  (The two-pass LDE writes a 1.13-GB
// intermediate in pass 1 and reads it in pass 2: processed in column groups the intermediate of a group could stay on the die.)
// write kernel -> read kernel -> in-place read-modify-write kernel over S bytes, S = 32 MB ... 2 GB; GB/s per kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/ubench_mall.hip -o tools/ubench/bin/ubench_mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
__global__ void __launch_bounds__(256) wr_kernel(ulonglong2* p, size_t n16, u64 v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_ulonglong2(v + i, v ^ i);
}
__global__ void __launch_bounds__(256) rd_kernel(const ulonglong2* p, size_t n16, u64* out) {
    u64 acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const ulonglong2 x = p[i]; acc += x.x ^ x.y; }
    if (acc == 0x123456789ull) out[0] = acc;
}
__global__ void __launch_bounds__(256) rmw_kernel(ulonglong2* p, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { ulonglong2 x = p[i]; x.x += 3; x.y ^= 5; p[i] = x; }
}
// the store pattern of the LDE's column pass: block = (column, 64-column tile), thread (r, cc) writes rows 4r+{0..3}, 4(r+4)+{0..3} of every
// coset: 512-byte runs at a 32-KB stride, cosets 1 MB apart, columns 8 MB apart
__global__ void __launch_bounds__(256) wr_pattern_kernel(u64* out, int tc_log, int order) {
    const unsigned tc = 1u << tc_log, tid = threadIdx.x, r = tid >> tc_log, cc = tid & (tc - 1);
    const unsigned tiles = 4096u >> tc_log;
    unsigned tile, col;
    if (order == 0) { tile = blockIdx.x % tiles; col = blockIdx.x / tiles; }
    else { tile = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) % (tiles / 8)); col = blockIdx.x / tiles; }
    u64* base = out + (size_t)col * (1u << 20) + tile * tc + cc;
    for (int c = 0; c < 8; c++)
        for (int t2 = 0; t2 < 2; t2++)
            for (int k = 0; k < 4; k++) base[(size_t)c * (1u << 17) + (size_t)(4 * (r + 4 * t2) + k) * 4096] = (u64)c * k + tid;
}
int main() {
    const size_t maxb = 2ull << 30;
    void* buf; u64* out; CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&out, 8));
    hipEvent_t e[4]; for (auto& x : e) CK(hipEventCreate(&x));
    for (size_t mb : {32, 64, 96, 128, 160, 192, 224, 256, 384, 512, 1024, 2048}) {
        const size_t bytes = mb << 20, n16 = bytes / 16;
        float tw = 0, tr = 0, tm = 0; const int reps = 20;
        for (int i = 0; i < reps + 5; i++) {
            CK(hipEventRecord(e[0])); hipLaunchKernelGGL(wr_kernel, dim3(2048), dim3(256), 0, 0, (ulonglong2*)buf, n16, (u64)i);
            CK(hipEventRecord(e[1])); hipLaunchKernelGGL(rd_kernel, dim3(2048), dim3(256), 0, 0, (const ulonglong2*)buf, n16, out);
            CK(hipEventRecord(e[2])); hipLaunchKernelGGL(rmw_kernel, dim3(2048), dim3(256), 0, 0, (ulonglong2*)buf, n16);
            CK(hipEventRecord(e[3])); CK(hipEventSynchronize(e[3]));
            float a, b, c; CK(hipEventElapsedTime(&a, e[0], e[1])); CK(hipEventElapsedTime(&b, e[1], e[2])); CK(hipEventElapsedTime(&c, e[2], e[3]));
            if (i >= 5) { tw += a; tr += b; tm += c; }
        }
        printf("%5zu MB   write %7.0f GB/s   read-after-write %7.0f GB/s   in-place rmw (bytes in + out) %7.0f GB/s\n", mb,
               bytes / (tw / reps) / 1e6, bytes / (tr / reps) / 1e6, 2.0 * bytes / (tm / reps) / 1e6);
    }
    {   // 135 columns x 2^20 words = 1.13 GB
        const size_t bytes = 135ull << 23;
        for (int tc_log : {6, 7}) for (int order : {0, 1}) {
            const unsigned blocks = 135u * (4096u >> tc_log);
            float t = 0; const int reps = 20;
            for (int i = 0; i < reps + 5; i++) {
                CK(hipEventRecord(e[0])); hipLaunchKernelGGL(wr_pattern_kernel, dim3(blocks), dim3(4u << tc_log), 0, 0, (u64*)buf, tc_log, order);
                CK(hipEventRecord(e[1])); CK(hipEventSynchronize(e[1]));
                float a; CK(hipEventElapsedTime(&a, e[0], e[1])); if (i >= 5) t += a;
            }
            printf("column-pass store pattern, %3d-column tiles, block order %d: %.3f ms  %7.0f GB/s\n", 1 << tc_log, order, t / reps, bytes / (t / reps) / 1e6);
        }
    }
    return 0;
}
