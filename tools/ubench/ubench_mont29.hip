// A/B of two BN254 Montgomery products on gfx950:
//   m_mul   8 x 32-bit limbs, R = 2^256, the hand-scheduled asm of bn254_mmul_asm.inc (128 mads + 128 carry adds)
//   mont29  9 x 29-bit limbs, R' = 2^261: every column accumulates in a 64-bit register without carries (162 mads + 9 x (mul_lo, and,
//           64-bit shift, 64-bit add) + a 9-step normalisation); plain C, the compiler's v_mad_u64_u32 accumulate-in-place
//   hipcc -O3 --offload-arch=gfx950 -I stark-verifier_amd/csrc -I include tools/ubench/ubench_mont29.hip -o tools/ubench/bin/ubench_mont29
// Prints ns per wave-product at full occupancy for chains of dependent products (x <- x * y), and checks both against each other through
// the domain change (a * b * 2^-256 = mont29(a, b * 2^5)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "bn254_field.cuh"
using namespace gl355;

struct f29 { uint32_t l[9]; };
__device__ __constant__ const uint32_t FR29_M[9] = {0x10000001, 0x1f0fac9f, 0x0e5c2450, 0x07d090f3, 0x1585d283, 0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e};
#define FR29_N0 0x0fffffffu   /* -m^-1 mod 2^29 */
#define MASK29 0x1fffffffu
__device__ __forceinline__ f29 to29(const u256& a) {         // same integer, 29-bit slices
    f29 r;
    uint64_t w[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) w[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int bit = 29 * j, k = bit >> 6, o = bit & 63;
        uint64_t v = w[k] >> o;
        if (o > 35 && k + 1 < 5) v |= w[k + 1] << (64 - o);
        r.l[j] = (uint32_t)v & MASK29;
    }
    return r;
}
__device__ __forceinline__ u256 from29(const f29& a) {       // limbs normalised, value < 2^256
    uint64_t w[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int bit = 29 * j, k = bit >> 6, o = bit & 63;
        w[k] |= (uint64_t)a.l[j] << o;
        if (o > 35 && k + 1 < 5) w[k + 1] |= (uint64_t)a.l[j] >> (64 - o);
    }
    u256 r;
    for (int i = 0; i < 4; i++) { r.l[2 * i] = (uint32_t)w[i]; r.l[2 * i + 1] = (uint32_t)(w[i] >> 32); }
    return r;
}
// a * b * 2^-261 mod m (some representative, limbs < 2^29): a's limbs may be up to 2^30, b's < 2^29
__device__ __forceinline__ f29 mont29(f29 a, f29 b) {
    uint64_t t[10];
#pragma unroll
    for (int j = 0; j < 10; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)a.l[j] * b.l[i];
        const uint32_t q = ((uint32_t)t[0] * FR29_N0) & MASK29;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)q * FR29_M[j];
        const uint64_t c = t[0] >> 29;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] = t[j + 1];
        t[9] = 0;
        t[0] += c;
    }
    f29 r;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        r.l[j] = (uint32_t)t[j] & MASK29;
        if (j < 8) t[j + 1] += t[j] >> 29;
        else r.l[8] = (uint32_t)t[8];
    }
    return r;
}
__global__ void __launch_bounds__(256) k_mmul(uint64_t* data, int iters) {
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    u256 x = load256(data + 8 * i), y = load256(data + 8 * i + 4);
    for (int k = 0; k < iters; k++) x = m_mul<F_R>(x, y);
    store256(data + 8 * i, x);
}
__global__ void __launch_bounds__(256) k_mont29(uint64_t* data, int iters) {
    const uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    const u256 x8 = load256(data + 8 * i), y8 = load256(data + 8 * i + 4);
    f29 x = to29(x8);
    // b' = y * 2^5 as an integer (< 2^260): the product then equals x y 2^-256 (mod m), the value m_mul gives
    u256 ys;
    for (int j = 7; j >= 0; j--) ys.l[j] = (y8.l[j] << 5) | (j ? y8.l[j - 1] >> 27 : 0);
    f29 y = to29(ys);
    y.l[8] |= (y8.l[7] >> 27) << 24;        // the five bits shifted out of the top word: bit 256 + .. = limb 8 bit 24 + ..
    for (int k = 0; k < iters; k++) x = mont29(x, y);
    // canonical: subtract m while >= m (value < 2^256 here)
    u256 r = from29(x);
    for (int k = 0; k < 8; k++) r = u_cond_sub(r, f_mod<F_R>());
    store256(data + 8 * i, r);
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 4096, iters = 512;      // 256 blocks = one wave per SIMD: the latency-bound regime of the small Merkle levels
    const size_t n = (size_t)blocks * 256;
    std::vector<uint64_t> h(n * 8);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s; }
    for (size_t i = 0; i < n; i++) { h[8 * i + 3] &= 0x1fffffffffffffffull; h[8 * i + 7] &= 0x1fffffffffffffffull; }   // < 2^253 < m
    uint64_t *d1, *d2;
    hipMalloc(&d1, n * 64); hipMalloc(&d2, n * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        uint64_t* d = which ? d2 : d1;
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            hipMemcpy(d, h.data(), n * 64, hipMemcpyHostToDevice);
            hipEventRecord(e0);
            if (which) hipLaunchKernelGGL(k_mont29, dim3(blocks), dim3(256), 0, 0, d, iters);
            else hipLaunchKernelGGL(k_mmul, dim3(blocks), dim3(256), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        printf("%s: %.3f ms for %d x %zu products = %.1f G products/s\n", which ? "mont29 (9 x 29, C)" : "m_mul (8 x 32, asm)", best, iters, n, (double)iters * n / best / 1e6);
    }
    std::vector<uint64_t> r1(n * 8), r2(n * 8);
    hipMemcpy(r1.data(), d1, n * 64, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, n * 64, hipMemcpyDeviceToHost);
    size_t bad = 0;
    // m_mul leaves a value < 2m: canonicalise on the host by comparing both (r1 or r1 - m) to r2
    const uint64_t M[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    for (size_t i = 0; i < n; i++) {
        uint64_t a[4] = {r1[8 * i], r1[8 * i + 1], r1[8 * i + 2], r1[8 * i + 3]};
        bool ge = true;
        for (int k = 3; k >= 0; k--) { if (a[k] > M[k]) break; if (a[k] < M[k]) { ge = false; break; } }
        if (ge) { unsigned __int128 br = 0; for (int k = 0; k < 4; k++) { unsigned __int128 dd = (unsigned __int128)a[k] - M[k] - (uint64_t)br; a[k] = (uint64_t)dd; br = (dd >> 64) & 1; } }
        for (int k = 0; k < 4; k++) if (a[k] != r2[8 * i + k]) { bad++; break; }
    }
    printf("mismatches: %zu of %zu\n", bad, n);
    return bad != 0;
}
