// Does the issue cost of v_mad_u64_u32 on gfx950 depend on WHICH registers its operands live in (VGPR banks) and on the operand kinds?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/ubench_mad_banks.hip -o tools/ubench/bin/ubench_mad_banks
// Each variant: 8 independent accumulators, 64 multiply-adds per loop trip, 8 waves per SIMD; prints SIMD clocks per wave instruction
// (shader clock read with s_memtime around the loop of one wave is not used: wall time x the clock measured by a sleeping probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X X X X X X X X
// accumulators v[40:41] .. v[54:55]; multiplicands by variant
#define MADS(A, B)                                       \
    "v_mad_u64_u32 v[40:41], s[10:11], " A ", " B ", v[40:41]\n\t" \
    "v_mad_u64_u32 v[42:43], s[10:11], " A ", " B ", v[42:43]\n\t" \
    "v_mad_u64_u32 v[44:45], s[10:11], " A ", " B ", v[44:45]\n\t" \
    "v_mad_u64_u32 v[46:47], s[10:11], " A ", " B ", v[46:47]\n\t" \
    "v_mad_u64_u32 v[48:49], s[10:11], " A ", " B ", v[48:49]\n\t" \
    "v_mad_u64_u32 v[50:51], s[10:11], " A ", " B ", v[50:51]\n\t" \
    "v_mad_u64_u32 v[52:53], s[10:11], " A ", " B ", v[52:53]\n\t" \
    "v_mad_u64_u32 v[54:55], s[10:11], " A ", " B ", v[54:55]\n\t"
#define CLOB "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v60","v61","v62","v63","s10","s11","s12"
template <int V>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t x = seed + threadIdx.x, r = 0;
    asm volatile("v_mov_b32 v60, %0\n\tv_mov_b32 v61, %0\n\tv_mov_b32 v62, %0\n\tv_mov_b32 v63, %0\n\ts_mov_b32 s12, 0x12345\n\t"
                 "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\tv_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"
                 "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\tv_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0"
                 :: "v"(x) : CLOB);
    for (int i = 0; i < iters; i++) {
        if (V == 0) asm volatile(REP8(MADS("v62", "v63")) ::: CLOB);      // VGPR x VGPR: banks 2, 3 against accumulators in banks 0, 1 / 2, 3
        if (V == 1) asm volatile(REP8(MADS("v60", "v60")) ::: CLOB);      // VGPR x the same VGPR
        if (V == 2) asm volatile(REP8(MADS("v62", "s12")) ::: CLOB);      // VGPR x SGPR
        if (V == 3) asm volatile(REP8(MADS("v62", "17")) ::: CLOB);       // VGPR x inline constant
        if (V == 4) asm volatile(REP8(MADS("v60", "v61")) ::: CLOB);      // VGPR x VGPR: banks 0, 1
    }
    asm volatile("v_xor_b32 %0, v40, v42\n\tv_xor_b32 %0, %0, v44\n\tv_xor_b32 %0, %0, v46\n\tv_xor_b32 %0, %0, v48\n\tv_xor_b32 %0, %0, v50\n\tv_xor_b32 %0, %0, v52\n\tv_xor_b32 %0, %0, v54"
                 : "=v"(r) :: CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int V> void run(const char* what, uint32_t* d) {
    const int blocks = 2048, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters, 7u + rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double insts = (double)blocks * 4 * iters * 64;
    printf("%-44s %.3f ms  %.1f G wave-inst/s  = %.2f SIMD clk per instruction at 2.3 GHz\n", what, best, insts / best / 1e6, 1024 * 2.3e9 / (insts / (best * 1e-3)));
}
int main() {
    uint32_t* d; hipMalloc(&d, 2048 * 256 * 4);
    run<0>("VGPR x VGPR (banks 2,3) + pair", d);
    run<4>("VGPR x VGPR (banks 0,1) + pair", d);
    run<1>("VGPR x same VGPR + pair", d);
    run<2>("VGPR x SGPR + pair", d);
    run<3>("VGPR x inline constant + pair", d);
    return 0;
}
