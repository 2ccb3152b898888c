// Is the Poseidon MDS layer a job for the matrix cores?  (north star: "MFMA tried only for the Poseidon MDS mat-vec where it
// really is a dense 12x12 contraction".)  Measured here, on the state layout the rest of the permutation needs (one state per
// lane, 12 x u64 in VGPRs):
//   valu : the MDS layer as shipped (poseidon.cuh psd_mds: 24 v_mad_u64_u32 + 5 per row, 348 instructions per layer)
//   mfma : ONLY the matrix instructions an int8 formulation needs per layer and wave -- 64 x v_mfma_i32_16x16x32_i8 (12 of the
//          32 K-slots and 12 of the 16 rows carry data; 8 byte planes x 4 groups of 16 states x 2 halves of the 12 inputs, the
//          halves because a lane can only feed its own state's bytes into its own K-block) -- with NONE of the byte gathering,
//          sign fix-up, cross-lane return of the 16x16 result tiles and 96-bit recombination that would surround them
//   mixed: even waves run the mfma loop, odd waves an S-box-like VALU loop: do the two pipes overlap?
// Build: hipcc --offload-arch=gfx950 -O3 -I ../../stark-verifier_amd/csrc ubench_mfma_mds.hip -o ubench_mfma_mds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "poseidon.cuh"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace gl355;
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int LAYERS = 30 * 16;     // 16 permutations' worth of layers per wave

__device__ __forceinline__ void mfma_layers(uint64_t* out, uint64_t seed) {
    long a = (long)(seed * 0x9E3779B97F4A7C15ull + threadIdx.x), b = (long)(seed ^ (threadIdx.x * 0x100000001B3ull));
    v4i acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int l = 0; l < LAYERS; l++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, acc[j], 0, 0, 0);
        }
        a += l;                                                                    // operands change between layers
    }
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) s += (uint64_t)(uint32_t)acc[j][0] + (uint32_t)acc[j][1] + (uint32_t)acc[j][2] + (uint32_t)acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__device__ __forceinline__ void valu_layers(uint64_t* out, uint64_t seed) {
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = seed * (i + 1) + threadIdx.x * 0x9E3779B97F4A7C15ull;
#pragma unroll 1
    for (int l = 0; l < LAYERS; l++) psd_mds(s, &PSD_ALL_RC[12 * (l % 30)]);
    uint64_t x = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) x ^= s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__device__ __forceinline__ void sbox_layers(uint64_t* out, uint64_t seed) {     // 12 S-boxes per "layer": the VALU work of a full round
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = seed * (i + 1) + threadIdx.x * 0x9E3779B97F4A7C15ull;
#pragma unroll 1
    for (int l = 0; l < LAYERS; l++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = psd_sbox(s[i]);
    }
    uint64_t x = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) x ^= s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void __launch_bounds__(256) k_mfma(uint64_t* out, uint64_t seed) { mfma_layers(out, seed); }
__global__ void __launch_bounds__(256) k_valu(uint64_t* out, uint64_t seed) { valu_layers(out, seed); }
__global__ void __launch_bounds__(256) k_sbox(uint64_t* out, uint64_t seed) { sbox_layers(out, seed); }
__global__ void __launch_bounds__(256) k_mixed(uint64_t* out, uint64_t seed) {
    if ((threadIdx.x >> 6) & 1) sbox_layers(out, seed); else mfma_layers(out, seed);        // waves 1,3 VALU; waves 0,2 MFMA (every SIMD gets both)
}

int main() {
    const int blocks = 256 * 8, threads = 256;      // 8 waves per SIMD
    uint64_t* out;
    CHECK(hipMalloc(&out, (size_t)blocks * threads * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct { const char* name; void (*k)(uint64_t*, uint64_t); double waves_frac; } runs[] = {
        {"valu  (psd_mds as shipped)", k_valu, 1.0}, {"mfma  (64 x 16x16x32 i8 per layer, nothing else)", k_mfma, 1.0},
        {"sbox  (12 S-boxes per layer)", k_sbox, 1.0}, {"mixed (half the waves mfma, half sbox)", k_mixed, 0.5}};
    for (auto& r : runs) {
        hipLaunchKernelGGL(r.k, dim3(blocks), dim3(threads), 0, 0, out, 1ull);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(r.k, dim3(blocks), dim3(threads), 0, 0, out, 2ull);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double waves_per_simd = (double)blocks * threads / 64 / 1024;
        printf("%-52s %8.3f ms   %7.1f ns per layer per wave-slot (%.0f waves/SIMD, %d layers each)\n", r.name, ms,
               ms * 1e6 / (waves_per_simd * r.waves_frac * LAYERS), waves_per_simd, LAYERS);
    }
    return 0;
}
