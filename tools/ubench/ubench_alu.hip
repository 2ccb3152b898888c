// Micro-benchmarks for the integer ALU ops the Goldilocks kernels lean on (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 ubench_alu.hip -o ubench_alu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
    uint64_t d;
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
    return d;
}

__global__ void k_mad_u64_u32(uint64_t* out, uint32_t seed) {
    uint64_t acc[ILP];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) acc[j] = mad64((uint32_t)acc[j], b, acc[j]);
    }
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mul_lo(uint64_t* out, uint32_t seed) {
    uint32_t acc[ILP];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(acc[j]) : "v"(acc[j]), "v"(b));
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mul_hi(uint64_t* out, uint32_t seed) {
    uint32_t acc[ILP];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(acc[j]) : "v"(acc[j]), "v"(b));
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mad_u32_u24(uint64_t* out, uint32_t seed) {
    uint32_t acc[ILP];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(acc[j]) : "v"(acc[j]), "v"(b), "v"(acc[j]));
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_add_u32(uint64_t* out, uint32_t seed) {
    uint32_t acc[ILP];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) asm volatile("v_add_u32 %0, %1, %2" : "=v"(acc[j]) : "v"(acc[j]), "v"(b));
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_addc(uint64_t* out, uint32_t seed) {   // add_co + addc pair = one 64-bit add
    uint64_t acc[ILP];
    uint64_t a = threadIdx.x * 2654435761ull + seed, b = blockIdx.x * 40503ull + 0x123456789ull;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) {
            uint32_t lo = (uint32_t)acc[j], hi = acc[j] >> 32;
            asm volatile("v_add_co_u32 %0, vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc"
                         : "=&v"(lo), "=v"(hi) : "v"(lo), "v"(hi), "v"((uint32_t)b), "v"((uint32_t)(b >> 32)) : "vcc");
            acc[j] = ((uint64_t)hi << 32) | lo;
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_lshl_add_u64(uint64_t* out, uint32_t seed) {
    uint64_t acc[ILP];
    uint64_t a = threadIdx.x * 2654435761ull + seed, b = blockIdx.x * 40503ull + 0x123456789ull;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(acc[j]) : "v"(acc[j]), "v"(b));
    }
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fma_f64(uint64_t* out, uint32_t seed) {
    double acc[ILP];
    double a = 1.0 + 1e-9 * (threadIdx.x + seed), b = 1.0 - 1e-9 * blockIdx.x;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(acc[j]) : "v"(acc[j]), "v"(b), "v"(acc[j]));
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s += acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}

__global__ void k_dot2_u16(uint64_t* out, uint32_t seed) {
    uint32_t acc[ILP];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) asm volatile("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(acc[j]) : "v"(acc[j]), "v"(b), "v"(acc[j]));
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_dot4_u8(uint64_t* out, uint32_t seed) {
    uint32_t acc[ILP];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) asm volatile("v_dot4_u32_u8 %0, %1, %2, %3" : "=v"(acc[j]) : "v"(acc[j]), "v"(b), "v"(acc[j]));
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_perm(uint64_t* out, uint32_t seed) {
    uint32_t acc[ILP];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(acc[j]) : "v"(acc[j]), "v"(b), "v"(0x05040100u));
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_cndmask(uint64_t* out, uint32_t seed) {
    uint32_t acc[ILP];
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(acc[j]) : "v"(acc[j]), "v"(b));
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_cmp64(uint64_t* out, uint32_t seed) {
    uint64_t acc[ILP];
    uint64_t a = threadIdx.x * 2654435761ull + seed, b = blockIdx.x * 40503ull + 0x123456789ull;
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(acc[j]), "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + cnt;
}

// full Goldilocks modmul, compiler-scheduled
__device__ __forceinline__ uint64_t gl_mul(uint64_t a, uint64_t b) {
    uint32_t a0 = (uint32_t)a, a1 = a >> 32, b0 = (uint32_t)b, b1 = b >> 32;
    uint64_t t = (uint64_t)a0 * b0;
    uint64_t u = (uint64_t)a0 * b1 + (t >> 32);
    uint64_t v = (uint64_t)a1 * b0 + (uint32_t)u;
    uint64_t w = (uint64_t)a1 * b1 + (u >> 32) + (v >> 32);
    uint64_t lo = (v << 32) | (uint32_t)t;
    uint32_t w0 = (uint32_t)w, w1 = w >> 32;
    uint64_t t0 = lo - w1;
    if (lo < w1) t0 -= 0xFFFFFFFFull;
    uint64_t t1 = ((uint64_t)w0 << 32) - w0;
    uint64_t r = t0 + t1;
    if (r < t1) r += 0xFFFFFFFFull;
    return r;
}
__global__ void k_gl_mul(uint64_t* out, uint32_t seed) {
    uint64_t acc[ILP];
    uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ull + seed, b = blockIdx.x * 0xD1B54A32D192ED03ull + 0x123456789ull;
#pragma unroll
    for (int j = 0; j < ILP; j++) acc[j] = a + j;
    for (int i = 0; i < ITERS / 4; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) acc[j] = gl_mul(acc[j], b);
    }
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
int run(const char* name, K kern, double ops_per_thread, uint64_t* d_out, int blocks, int threads) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, (uint32_t)rep);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double total = ops_per_thread * blocks * threads;
    double rate = total / (best * 1e-3);
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
    double cyc = 1024.0 * 2.4e9 * 64.0 / rate;
    printf("%-16s %8.3f ms  %8.2f Tops/s (lane-ops)  ~%.2f cyc/wave-instr/SIMD @2.4GHz\n", name, best, rate / 1e12, cyc);
    return 0;
}

int main() {
    int blocks = 256 * 8, threads = 256;
    uint64_t* d_out;
    CHECK(hipMalloc(&d_out, sizeof(uint64_t) * blocks * threads));
    double n = (double)ITERS * ILP;
    run("v_add_u32", k_add_u32, n, d_out, blocks, threads);
    run("add_co+addc", k_addc, n, d_out, blocks, threads);
    run("v_lshl_add_u64", k_lshl_add_u64, n, d_out, blocks, threads);
    run("v_mul_lo_u32", k_mul_lo, n, d_out, blocks, threads);
    run("v_mul_hi_u32", k_mul_hi, n, d_out, blocks, threads);
    run("v_mad_u32_u24", k_mad_u32_u24, n, d_out, blocks, threads);
    run("v_mad_u64_u32", k_mad_u64_u32, n, d_out, blocks, threads);
    run("v_fma_f64", k_fma_f64, n, d_out, blocks, threads);
    run("v_dot2_u32_u16", k_dot2_u16, n, d_out, blocks, threads);
    run("v_dot4_u32_u8", k_dot4_u8, n, d_out, blocks, threads);
    run("v_perm_b32", k_perm, n, d_out, blocks, threads);
    run("v_cndmask_b32", k_cndmask, n, d_out, blocks, threads);
    run("v_cmp_lt_u64", k_cmp64, n, d_out, blocks, threads);
    run("gl_mul (full)", k_gl_mul, n / 4, d_out, blocks, threads);
    return 0;
}
