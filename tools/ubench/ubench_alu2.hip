// Which 32-bit integer VALU instructions issue at the full SIMD-32 rate (2 clk per wave64) on gfx950, and which at half rate?
// (round 3: the NTT butterflies in carry-free 24-bit limbs only pay if add / sub / shift / mask are full-rate.)
// Build: hipcc --offload-arch=gfx950 -O3 ubench_alu2.hip -o ubench_alu2 ; run on the GPU box.  Every kernel runs ILP independent
// dependency chains per lane; 2048 blocks x 256 threads = 8 waves per SIMD.  The clock is measured with s_memtime-free arithmetic:
// cycles are reported relative to the measured rate of v_fma_f32 (documented 2 clk per wave64) in the same run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int ILP = 8;

// acc <- op(acc, b [, acc]) as a 3-register asm template
#define K3(NAME, ASM)                                                                                        \
    __global__ void NAME(uint64_t* out, uint32_t seed) {                                                   \
        uint32_t acc[ILP];                                                                                 \
        uint32_t a = threadIdx.x * 2654435761u + seed, b = (blockIdx.x * 40503u + 12345u) & 15u;          \
        _Pragma("unroll") for (int j = 0; j < ILP; j++) acc[j] = a + j;                                   \
        for (int i = 0; i < ITERS; i++) {                                                                  \
            _Pragma("unroll") for (int j = 0; j < ILP; j++) asm volatile(ASM : "=v"(acc[j]) : "v"(acc[j]), "v"(b), "0"(acc[j])); \
        }                                                                                                  \
        uint32_t s = 0;                                                                                    \
        _Pragma("unroll") for (int j = 0; j < ILP; j++) s ^= acc[j];                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                    \
    }

K3(k_fma_f32, "v_fma_f32 %0, %1, %2, %3")
K3(k_add_u32, "v_add_u32 %0, %1, %2")
K3(k_sub_u32, "v_sub_u32 %0, %1, %2")
K3(k_and_b32, "v_and_b32 %0, %1, %2")
K3(k_or_b32, "v_or_b32 %0, %1, %2")
K3(k_xor_b32, "v_xor_b32 %0, %1, %2")
K3(k_lshlrev, "v_lshlrev_b32 %0, %2, %1")
K3(k_lshrrev, "v_lshrrev_b32 %0, %2, %1")
K3(k_ashrrev, "v_ashrrev_i32 %0, %2, %1")
K3(k_bfe_u32, "v_bfe_u32 %0, %1, %2, 11")
K3(k_bfe_i32, "v_bfe_i32 %0, %1, %2, 11")
K3(k_alignbit, "v_alignbit_b32 %0, %1, %3, %2")
K3(k_add3, "v_add3_u32 %0, %1, %2, %3")
K3(k_lshl_add, "v_lshl_add_u32 %0, %1, 3, %2")
K3(k_add_lshl, "v_add_lshl_u32 %0, %1, %2, 3")
K3(k_and_or, "v_and_or_b32 %0, %1, %2, %3")
K3(k_lshl_or, "v_lshl_or_b32 %0, %1, 3, %2")
K3(k_xad, "v_xad_u32 %0, %1, %2, %3")
K3(k_mad_i32_i24, "v_mad_i32_i24 %0, %1, %2, %3")
K3(k_mul_u32_u24, "v_mul_u32_u24 %0, %1, %2")
K3(k_mul_lo, "v_mul_lo_u32 %0, %1, %2")
K3(k_min_u32, "v_min_u32 %0, %1, %2")
K3(k_max_i32, "v_max_i32 %0, %1, %2")
K3(k_sad, "v_sad_u32 %0, %1, %2, %3")
K3(k_perm, "v_perm_b32 %0, %1, %2, %3")
K3(k_mov, "v_mov_b32 %0, %1")
K3(k_add_co, "v_add_co_u32 %0, vcc, %1, %2")
K3(k_sub_co, "v_sub_co_u32 %0, vcc, %1, %2")
K3(k_addc, "v_addc_co_u32 %0, vcc, %1, %2, vcc")
K3(k_cndmask, "v_cndmask_b32 %0, %1, %2, vcc")
K3(k_pk_add_u16, "v_pk_add_u16 %0, %1, %2")
K3(k_pk_sub_i16, "v_pk_sub_i16 %0, %1, %2")
K3(k_pk_mad_u16, "v_pk_mad_u16 %0, %1, %2, %3")
K3(k_pk_lshl, "v_pk_lshlrev_b16 %0, %2, %1")
K3(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %1")
K3(k_mul_f32, "v_mul_f32 %0, %1, %2")
K3(k_add_f32, "v_add_f32 %0, %1, %2")
K3(k_dot4_i32_i8, "v_dot4_i32_i8 %0, %1, %2, %3")

// 64-bit forms
#define K64(NAME, ASM)                                                                                       \
    __global__ void NAME(uint64_t* out, uint32_t seed) {                                                   \
        uint64_t acc[ILP];                                                                                 \
        uint64_t a = threadIdx.x * 2654435761ull + seed, b = blockIdx.x * 40503ull + 0x123456789ull;       \
        uint32_t b32 = (uint32_t)b;                                                                        \
        _Pragma("unroll") for (int j = 0; j < ILP; j++) acc[j] = a + j;                                   \
        for (int i = 0; i < ITERS; i++) {                                                                  \
            _Pragma("unroll") for (int j = 0; j < ILP; j++) asm volatile(ASM : "=v"(acc[j]) : "v"(acc[j]), "v"(b), "v"(b32), "0"(acc[j])); \
        }                                                                                                  \
        uint64_t s = 0;                                                                                    \
        _Pragma("unroll") for (int j = 0; j < ILP; j++) s ^= acc[j];                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                    \
    }
K64(k_lshl_add_u64, "v_lshl_add_u64 %0, %1, 0, %2")
K64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %3, %3, %1")
K64(k_mad_i64_i32, "v_mad_i64_i32 %0, vcc, %3, %3, %1")
K64(k_lshlrev_b64, "v_lshlrev_b64 %0, 3, %1")
K64(k_lshrrev_b64, "v_lshrrev_b64 %0, 3, %1")
K64(k_pk_add_f32, "v_pk_add_f32 %0, %1, %2")
K64(k_pk_fma_f32, "v_pk_fma_f32 %0, %1, %2, %1")
K64(k_add_f64, "v_add_f64 %0, %1, %2")
K64(k_pk_mov, "v_pk_mov_b32 %0, %1, %2")

// composite: one radix-2 butterfly (s, d) = (a + b, a - b) in the candidate representations; a, b live in registers, results replace them
// (1) 3 x 32-bit limbs, carry chains (the lazily reduced form shipped in round 2): add = 3, sub (a + (Mp - b)) = 6
__global__ void k_bfly_l96(uint64_t* out, uint32_t seed) {
    uint32_t a0[ILP / 2], a1[ILP / 2], a2[ILP / 2], b0[ILP / 2], b1[ILP / 2], b2[ILP / 2];
    uint32_t x = threadIdx.x * 2654435761u + seed;
#pragma unroll
    for (int j = 0; j < ILP / 2; j++) { a0[j] = x + j; a1[j] = x * 3 + j; a2[j] = j; b0[j] = x * 5 + j; b1[j] = x * 7 + j; b2[j] = j + 1; }
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP / 2; j++) {
            uint32_t s0, s1, s2, n0, n1, n2;
            asm volatile("v_add_co_u32 %0, vcc, %6, %9\n\tv_addc_co_u32 %1, vcc, %7, %10, vcc\n\tv_addc_co_u32 %2, vcc, %8, %11, vcc\n\t"
                         "v_sub_co_u32 %3, vcc, 2, %9\n\tv_subb_co_u32 %4, vcc, -3, %10, vcc\n\tv_subb_co_u32 %5, vcc, 1, %11, vcc\n\t"
                         "v_add_co_u32 %3, vcc, %6, %3\n\tv_addc_co_u32 %4, vcc, %7, %4, vcc\n\tv_addc_co_u32 %5, vcc, %8, %5, vcc"
                         : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(n0), "=&v"(n1), "=&v"(n2)
                         : "v"(a0[j]), "v"(a1[j]), "v"(a2[j]), "v"(b0[j]), "v"(b1[j]), "v"(b2[j]) : "vcc");
            a0[j] = s0; a1[j] = s1; a2[j] = s2 & 15; b0[j] = n0; b1[j] = n1; b2[j] = n2 & 15;
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP / 2; j++) s ^= a0[j] ^ a1[j] ^ a2[j] ^ b0[j] ^ b1[j] ^ b2[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// (2) 4 x 24-bit limbs in 32-bit registers, no carries: add = 4 v_add_u32, sub = 4 v_sub_u32
__global__ void k_bfly_l24(uint64_t* out, uint32_t seed) {
    uint32_t a[ILP / 2][4], b[ILP / 2][4];
    uint32_t x = threadIdx.x * 2654435761u + seed;
#pragma unroll
    for (int j = 0; j < ILP / 2; j++)
#pragma unroll
        for (int k = 0; k < 4; k++) { a[j][k] = x * (2 * k + 1) + j; b[j][k] = x * (2 * k + 9) + j; }
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP / 2; j++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t s, d;
                asm volatile("v_add_u32 %0, %2, %3\n\tv_sub_u32 %1, %2, %3" : "=&v"(s), "=&v"(d) : "v"(a[j][k]), "v"(b[j][k]));
                a[j][k] = s; b[j][k] = d;
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP / 2; j++)
#pragma unroll
        for (int k = 0; k < 4; k++) s ^= a[j][k] ^ b[j][k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// (3) the same with packed 16-bit limbs: 6 x 16-bit limbs... two limbs per register need headroom the format does not have; timing only:
// v_pk_add_u16 / v_pk_sub_i16 on 3 registers per value
__global__ void k_bfly_pk16(uint64_t* out, uint32_t seed) {
    uint32_t a[ILP / 2][3], b[ILP / 2][3];
    uint32_t x = threadIdx.x * 2654435761u + seed;
#pragma unroll
    for (int j = 0; j < ILP / 2; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) { a[j][k] = x * (2 * k + 1) + j; b[j][k] = x * (2 * k + 9) + j; }
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int j = 0; j < ILP / 2; j++) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                uint32_t s, d;
                asm volatile("v_pk_add_u16 %0, %2, %3\n\tv_pk_sub_i16 %1, %2, %3" : "=&v"(s), "=&v"(d) : "v"(a[j][k]), "v"(b[j][k]));
                a[j][k] = s; b[j][k] = d;
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ILP / 2; j++)
#pragma unroll
        for (int k = 0; k < 3; k++) s ^= a[j][k] ^ b[j][k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double g_fma_ms = 0;
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
    const double per_op = best / ops_per_thread;       // relative unit
    if (g_fma_ms == 0) g_fma_ms = per_op;
    double total = ops_per_thread * blocks * threads;
    double rate = total / (best * 1e-3);
    printf("%-18s %8.3f ms  %8.2f Tops/s (lane-ops)  %5.2f x v_fma_f32 = ~%.2f clk per wave instruction\n", name, best, rate / 1e12, per_op / g_fma_ms,
           2.0 * per_op / g_fma_ms);
    return 0;
}

int main() {
    int blocks = 256 * 8, threads = 256;
    uint64_t* d_out;
    CHECK(hipMalloc(&d_out, sizeof(uint64_t) * blocks * threads));
    double n = (double)ITERS * ILP;
#define R(K) run(#K, K, n, d_out, blocks, threads)
    R(k_fma_f32); R(k_add_u32); R(k_sub_u32); R(k_and_b32); R(k_or_b32); R(k_xor_b32); R(k_lshlrev); R(k_lshrrev); R(k_ashrrev);
    R(k_bfe_u32); R(k_bfe_i32); R(k_alignbit); R(k_add3); R(k_lshl_add); R(k_add_lshl); R(k_and_or); R(k_lshl_or); R(k_xad);
    R(k_mad_i32_i24); R(k_mul_u32_u24); R(k_mul_lo); R(k_min_u32); R(k_max_i32); R(k_sad); R(k_perm); R(k_mov);
    R(k_add_co); R(k_sub_co); R(k_addc); R(k_cndmask); R(k_pk_add_u16); R(k_pk_sub_i16); R(k_pk_mad_u16); R(k_pk_lshl);
    R(k_cvt_f32_u32); R(k_mul_f32); R(k_add_f32); R(k_dot4_i32_i8);
    R(k_lshl_add_u64); R(k_mad_u64_u32); R(k_mad_i64_i32); R(k_lshlrev_b64); R(k_lshrrev_b64); R(k_pk_add_f32); R(k_pk_fma_f32); R(k_add_f64); R(k_pk_mov);
    // composites: per iteration ILP/2 butterflies; report per butterfly
    run("bfly 3x32 carry", k_bfly_l96, (double)ITERS * (ILP / 2), d_out, blocks, threads);
    run("bfly 4x24 nocarry", k_bfly_l24, (double)ITERS * (ILP / 2), d_out, blocks, threads);
    run("bfly pk16 x3", k_bfly_pk16, (double)ITERS * (ILP / 2), d_out, blocks, threads);
    return 0;
}
