// select / carry idioms: how expensive is v_cndmask really, and what are the alternatives?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
constexpr int ITERS = 4096, ILP = 8;
#define LOOP(BODY) \
    uint32_t acc[ILP]; uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u; \
    _Pragma("unroll") for (int j = 0; j < ILP; j++) acc[j] = a + j; \
    for (int i = 0; i < ITERS; i++) { _Pragma("unroll") for (int j = 0; j < ILP; j++) { BODY } } \
    uint32_t s = 0; _Pragma("unroll") for (int j = 0; j < ILP; j++) s ^= acc[j]; out[blockIdx.x * blockDim.x + threadIdx.x] = s;

__global__ void k_cnd_vcc(uint64_t* out, uint32_t seed) { LOOP(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(acc[j]) : "v"(acc[j]), "v"(b));) }
__global__ void k_cnd_e64(uint64_t* out, uint32_t seed) { LOOP(asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(acc[j]) : "v"(acc[j]), "v"(b) : "s20", "s21");) }
__global__ void k_cmp_cnd(uint64_t* out, uint32_t seed) { LOOP(asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(acc[j]) : "v"(acc[j]), "v"(b) : "vcc");) }
__global__ void k_cmp_cnd_nonop(uint64_t* out, uint32_t seed) { LOOP(asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(acc[j]) : "v"(acc[j]), "v"(b) : "vcc");) }
__global__ void k_addco_addc(uint64_t* out, uint32_t seed) { LOOP(asm volatile("v_add_co_u32 %0, vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "=v"(acc[j]) : "v"(acc[j]), "v"(b) : "vcc");) }
__global__ void k_min_u32(uint64_t* out, uint32_t seed) { LOOP(asm volatile("v_min_u32 %0, %1, %2" : "=v"(acc[j]) : "v"(acc[j]), "v"(b));) }
__global__ void k_and_or(uint64_t* out, uint32_t seed) { LOOP(asm volatile("v_and_or_b32 %0, %1, %2, %1" : "=v"(acc[j]) : "v"(acc[j]), "v"(b));) }
__global__ void k_mov(uint64_t* out, uint32_t seed) { LOOP(asm volatile("v_mov_b32 %0, %1" : "=v"(acc[j]) : "v"(acc[j] ^ b));) }
__global__ void k_add3(uint64_t* out, uint32_t seed) { LOOP(asm volatile("v_add3_u32 %0, %1, %2, %1" : "=v"(acc[j]) : "v"(acc[j]), "v"(b));) }

template <typename K> void run(const char* name, K kern, double per_iter_instrs, uint64_t* d_out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 2048, threads = 256;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 1u); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) { hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, (uint32_t)r); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    double groups = (double)ITERS * ILP * blocks * threads;  // BODY executions (lane level)
    double cyc = 1024.0 * 2.4e9 * 64.0 / (groups / (best * 1e-3));
    printf("%-22s %8.3f ms  ~%.2f cyc per BODY (%g instr) per wave/SIMD @2.4GHz\n", name, best, cyc, per_iter_instrs);
}
int main() {
    uint64_t* d; hipMalloc(&d, 8 * 2048 * 256);
    run("v_mov_b32 (+xor)", k_mov, 2, d);
    run("v_min_u32", k_min_u32, 1, d);
    run("v_add3_u32", k_add3, 1, d);
    run("v_and_or_b32", k_and_or, 1, d);
    run("cndmask vcc", k_cnd_vcc, 1, d);
    run("cndmask e64 sgpr", k_cnd_e64, 1, d);
    run("cmp+nop+cndmask", k_cmp_cnd, 3, d);
    run("cmp+cndmask (no nop)", k_cmp_cnd_nonop, 2, d);
    run("add_co+addc", k_addco_addc, 2, d);
    return 0;
}
