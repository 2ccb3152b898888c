// The VALU roofline's peak, measured on the device a benchmark runs on, in the run it reports (Poseidon, Merkle and the
// constraint kernel are bound by the integer VALU issue rate, not by HBM or MFMA).
//
// gl355_valu_probe: for each instruction class one kernel that does nothing but issue that instruction -- 8 independent dependency
// chains per lane, 8 waves per SIMD on all 1024 SIMDs -- timed with HIP events on the context's stream: the rate in wave-level
// instructions per second is the chip's issue ceiling for that class at whatever clock the chip holds under that load, and the clock
// itself is read inside the kernel (s_memtime cycles over the 100-MHz s_memrealtime counter), so cost = clock * SIMDs / rate in shader
// cycles per wave instruction per SIMD needs no assumed frequency.  Classes (tools/ubench/ubench_alu2.hip surveyed ~50 opcodes: they
// fall into these rate classes):
//   FULL32  v_add_u32              plain 32-bit add / sub / logic / right shift / move
//   HALF32  v_add_co_u32           carry-producing adds, left shifts, v_mul_lo, v_add3, v_perm, v_cndmask ... (everything else 32-bit)
//   MAD64   v_mad_u64_u32          the multiply-add every field product is made of (and the 64-bit shifts)
// A kernel's peak for its own mix is the harmonic combination: 1 / sum_c f_c / rate_c.
//
// gl355_clock_probe: one wave that sleeps for `micros` of the real-time counter and reports the shader cycles that passed: the shader
// clock under whatever else runs on the device meanwhile (bench.py samples it during the timed region from its own context).
#include "gl355_internal.h"
#include "poseidon.cuh"

namespace gl355 {

constexpr int VP_ILP = 8;           // (GL355_VP_CO8 lists eight carry-out pairs)
constexpr int VP_ITERS = 512;        // x VP_UNROLL x VP_ILP instructions per lane
constexpr int VP_UNROLL = 8;         // 64 probe instructions per loop iteration: the loop's scalar compare + branch is < 2 % of the issue slots
constexpr int VP_BLOCKS = 2048;      // x 256 lanes = 8 waves per SIMD on 1024 SIMDs

struct VpClock { unsigned long long cyc, ticks; };

#define GL355_VP_CLOCK_BEGIN const unsigned long long w0 = wall_clock64(), c0 = clock64();
#define GL355_VP_CLOCK_END                                                                   \
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk->cyc = clock64() - c0; clk->ticks = wall_clock64() - w0; }

__global__ void __launch_bounds__(256) vp_full32_kernel(uint32_t* out, uint32_t seed, VpClock* clk) {
    GL355_VP_CLOCK_BEGIN
    uint32_t acc[VP_ILP];
    const uint32_t a = threadIdx.x * 2654435761u + seed, b = (blockIdx.x * 40503u + 12345u) | 1u;
#pragma unroll
    for (int j = 0; j < VP_ILP; j++) acc[j] = a + j;
#pragma unroll 1
    for (int i = 0; i < VP_ITERS; i++) {
#pragma unroll
        for (int r = 0; r < VP_UNROLL; r++)
#pragma unroll
        for (int j = 0; j < VP_ILP; j++) asm volatile("v_add_u32 %0, %1, %2" : "=v"(acc[j]) : "v"(acc[j]), "v"(b));
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < VP_ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    GL355_VP_CLOCK_END
}
// one carry-out scalar pair per chain, named in the text: a shared pair (or VCC) makes the assembler pad every instruction with a hazard nop
#define GL355_VP_CO8(M) M(0, "s[40:41]", "s40", "s41") M(1, "s[42:43]", "s42", "s43") M(2, "s[44:45]", "s44", "s45") M(3, "s[46:47]", "s46", "s47") \
                        M(4, "s[48:49]", "s48", "s49") M(5, "s[50:51]", "s50", "s51") M(6, "s[52:53]", "s52", "s53") M(7, "s[54:55]", "s54", "s55")
__global__ void __launch_bounds__(256) vp_half32_kernel(uint32_t* out, uint32_t seed, VpClock* clk) {
    GL355_VP_CLOCK_BEGIN
    uint32_t acc[VP_ILP];
    const uint32_t a = threadIdx.x * 2654435761u + seed, b = (blockIdx.x * 40503u + 12345u) | 1u;
#pragma unroll
    for (int j = 0; j < VP_ILP; j++) acc[j] = a + j;
#pragma unroll 1
    for (int i = 0; i < VP_ITERS; i++) {
#pragma unroll
        for (int r = 0; r < VP_UNROLL; r++) {
#define GL355_VP_HALF(J, PAIR, LO, HI) asm volatile("v_add_co_u32 %0, " PAIR ", %1, %2" : "=v"(acc[J]) : "v"(acc[J]), "v"(b) : LO, HI);
            GL355_VP_CO8(GL355_VP_HALF)
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < VP_ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    GL355_VP_CLOCK_END
}
__global__ void __launch_bounds__(256) vp_mad64_kernel(uint32_t* out, uint32_t seed, VpClock* clk) {
    GL355_VP_CLOCK_BEGIN
    uint64_t acc[VP_ILP];
    const uint64_t a = threadIdx.x * 2654435761ull + seed;
    const uint32_t b = (blockIdx.x * 40503u + 12345u) | 1u, c = seed | 3u;
#pragma unroll
    for (int j = 0; j < VP_ILP; j++) acc[j] = a + j;
#pragma unroll 1
    for (int i = 0; i < VP_ITERS; i++) {
#pragma unroll
        for (int r = 0; r < VP_UNROLL; r++) {
#define GL355_VP_MAD(J, PAIR, LO, HI) asm volatile("v_mad_u64_u32 %0, " PAIR ", %1, %2, %0" : "+v"(acc[J]) : "v"(b), "v"(c) : LO, HI);
            GL355_VP_CO8(GL355_VP_MAD)
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < VP_ILP; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
    GL355_VP_CLOCK_END
}
__global__ void vp_clock_kernel(unsigned long long ticks, VpClock* clk) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long w1 = w0;
    while (w1 - w0 < ticks) { __builtin_amdgcn_s_sleep(8); w1 = wall_clock64(); }
    if (threadIdx.x == 0) { clk->cyc = clock64() - c0; clk->ticks = w1 - w0; }
}


// ---- per-opcode-form probes (round 6: VERDICT r5 #1).  The three classes above are too coarse to be a ceiling: the job issued faster than the
// harmonic combination of the three single-class loops.  Below, ONE kernel per opcode FORM the library actually ships (tools/isa_mix.py lists the
// forms per kernel), each as ILP = 1 (one dependent chain per lane), 4 and 8 independent chains per lane, always 8 waves per SIMD on every SIMD.
// A form's issue cost is the best of the three; a kernel's ceiling is 1024 SIMDs x clock / sum_forms f_form x cost_form.
// Chains that carry through a scalar pair (addc / subb) keep the gfx950 two-wait-state distance between the VALU write of the pair and its VALU
// reader: at ILP 1 by an s_nop 1 in front of each instruction (the nop is a scalar-side instruction of that wave; the other seven waves of the
// SIMD issue meanwhile), at ILP >= 4 by the partner chains.
constexpr int VP_OP_BODY = 64;       // probe instructions per loop iteration, whatever the ILP
// The 64 instructions of an iteration are ONE asm statement (between separate statements the compiler's hazard recogniser pads inline asm it cannot
// see into with s_nop).  I(A, P): one instruction on the chain whose accumulator is operand A and whose scalar pair is P; %8 = b widened to the chain type, %9 = c, %11 = b (VGPRs),
// %10 = a wave-uniform 32-bit value (SGPR).  64-bit chains use the same operand numbers (register pairs).
#define VP_X2(x) x x
#define VP_X4(x) VP_X2(x) VP_X2(x)
#define VP_X8(x) VP_X4(x) VP_X4(x)
#define VP_X16(x) VP_X8(x) VP_X8(x)
#define VP_X64(x) VP_X16(x) VP_X16(x) VP_X16(x) VP_X16(x)
#define VP_BODY1(I) VP_X64(I("%0", "s[40:41]"))
#define VP_BODY4(I) VP_X16(I("%0", "s[40:41]") I("%1", "s[42:43]") I("%2", "s[44:45]") I("%3", "s[46:47]"))
#define VP_BODY8(I) VP_X8(I("%0", "s[40:41]") I("%1", "s[42:43]") I("%2", "s[44:45]") I("%3", "s[46:47]") I("%4", "s[48:49]") I("%5", "s[50:51]") I("%6", "s[52:53]") I("%7", "s[54:55]"))
#define VP_SCLOB "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"
#define GL355_VP_OP_KERNEL(NAME, TYPE, I, I1)                                                                                   \
    template <int ILP>                                                                                                          \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed, VpClock* clk) {                                   \
        GL355_VP_CLOCK_BEGIN                                                                                                    \
        TYPE acc[8];                                                                                                            \
        const uint32_t b = (blockIdx.x * 40503u + 12345u) | 1u, c = (seed & 15u) | 3u;                                         \
        const TYPE bw = (TYPE)(((uint64_t)b << 7) | c);                                                                        \
        _Pragma("unroll") for (int j = 0; j < 8; j++) acc[j] = (TYPE)(threadIdx.x * 2654435761u + seed + j);                   \
        asm volatile("s_mov_b64 s[40:41], 0x55\n\ts_mov_b64 s[42:43], 0x33\n\ts_mov_b64 s[44:45], 0x0f\n\ts_mov_b64 s[46:47], 0x17\n\t" \
                     "s_mov_b64 s[48:49], 0x71\n\ts_mov_b64 s[50:51], 0x2b\n\ts_mov_b64 s[52:53], 0x4d\n\ts_mov_b64 s[54:55], 0x63" ::: VP_SCLOB); \
        _Pragma("unroll 1") for (int i = 0; i < VP_ITERS; i++) {                                                                \
            if constexpr (ILP == 1)                                                                                             \
                asm volatile(VP_BODY1(I1) : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) \
                             : "v"(bw), "v"(c), "s"(seed), "v"(b) : VP_SCLOB);                                                          \
            else if constexpr (ILP == 4)                                                                                        \
                asm volatile(VP_BODY4(I) : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) \
                             : "v"(bw), "v"(c), "s"(seed), "v"(b) : VP_SCLOB);                                                          \
            else                                                                                                                \
                asm volatile(VP_BODY8(I) : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) \
                             : "v"(bw), "v"(c), "s"(seed), "v"(b) : VP_SCLOB);                                                          \
        }                                                                                                                       \
        TYPE s = 0;                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 8; j++) s ^= acc[j];                                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ ((uint64_t)s >> 32));                                       \
        GL355_VP_CLOCK_END                                                                                                      \
    }
#define VPI_ADD(A, P) "v_add_u32 " A ", " A ", %8\n\t"
#define VPI_SUB(A, P) "v_sub_u32 " A ", " A ", %8\n\t"
#define VPI_AND(A, P) "v_and_b32 " A ", " A ", %8\n\t"
#define VPI_LSHR(A, P) "v_lshrrev_b32 " A ", 1, " A "\n\t"
#define VPI_ASHR(A, P) "v_ashrrev_i32 " A ", 1, " A "\n\t"
#define VPI_MOV(A, P) "v_mov_b32 " A ", %8\n\t"
#define VPI_LSHL(A, P) "v_lshlrev_b32 " A ", 1, " A "\n\t"
#define VPI_ALIGN(A, P) "v_alignbit_b32 " A ", " A ", %8, 22\n\t"
#define VPI_ADD3(A, P) "v_add3_u32 " A ", " A ", %8, %9\n\t"
#define VPI_MULLO(A, P) "v_mul_lo_u32 " A ", " A ", %8\n\t"
#define VPI_ADDCO(A, P) "v_add_co_u32_e64 " A ", " P ", " A ", %8\n\t"
#define VPI_SUBCO(A, P) "v_sub_co_u32_e64 " A ", " P ", " A ", %8\n\t"
#define VPI_ADDC(A, P) "v_addc_co_u32_e64 " A ", " P ", " A ", %8, " P "\n\t"
#define VPI_SUBB(A, P) "v_subb_co_u32_e64 " A ", " P ", " A ", 0, " P "\n\t"
#define VPI_ADDC1(A, P) "s_nop 1\n\tv_addc_co_u32_e64 " A ", " P ", " A ", %8, " P "\n\t"
#define VPI_SUBB1(A, P) "s_nop 1\n\tv_subb_co_u32_e64 " A ", " P ", " A ", 0, " P "\n\t"
#define VPI_CNDMASK(A, P) "v_cndmask_b32_e64 " A ", " A ", %8, " P "\n\t"
#define VPI_CNDMASKC(A, P) "v_cndmask_b32_e64 " A ", 0, -1, " P "\n\t"
#define VPI_MAD(A, P) "v_mad_u64_u32 " A ", " P ", %11, %9, " A "\n\t"
#define VPI_MADS(A, P) "v_mad_u64_u32 " A ", " P ", %10, %9, " A "\n\t"
#define VPI_MADC(A, P) "v_mad_u64_u32 " A ", " P ", %9, 41, " A "\n\t"
#define VPI_MADM1(A, P) "v_mad_u64_u32 " A ", " P ", %9, -1, " A "\n\t"
#define VPI_LSHLADD64(A, P) "v_lshl_add_u64 " A ", " A ", 1, %8\n\t"
#define VPI_LSHL64(A, P) "v_lshlrev_b64 " A ", 1, " A "\n\t"
#define VPI_LSHR64(A, P) "v_lshrrev_b64 " A ", 1, " A "\n\t"
#define VPI_MOV64(A, P) "v_mov_b64 " A ", %8\n\t"
#define VPI_CMP64(A, P) "v_cmp_lt_u64_e64 " P ", " A ", %8\n\t"
GL355_VP_OP_KERNEL(vpo_add_u32, uint32_t, VPI_ADD, VPI_ADD)
GL355_VP_OP_KERNEL(vpo_sub_u32, uint32_t, VPI_SUB, VPI_SUB)
GL355_VP_OP_KERNEL(vpo_and_b32, uint32_t, VPI_AND, VPI_AND)
GL355_VP_OP_KERNEL(vpo_lshrrev_b32, uint32_t, VPI_LSHR, VPI_LSHR)
GL355_VP_OP_KERNEL(vpo_ashrrev_i32, uint32_t, VPI_ASHR, VPI_ASHR)
GL355_VP_OP_KERNEL(vpo_mov_b32, uint32_t, VPI_MOV, VPI_MOV)
GL355_VP_OP_KERNEL(vpo_lshlrev_b32, uint32_t, VPI_LSHL, VPI_LSHL)
GL355_VP_OP_KERNEL(vpo_alignbit_b32, uint32_t, VPI_ALIGN, VPI_ALIGN)
GL355_VP_OP_KERNEL(vpo_add3_u32, uint32_t, VPI_ADD3, VPI_ADD3)
GL355_VP_OP_KERNEL(vpo_mul_lo_u32, uint32_t, VPI_MULLO, VPI_MULLO)
GL355_VP_OP_KERNEL(vpo_add_co, uint32_t, VPI_ADDCO, VPI_ADDCO)
GL355_VP_OP_KERNEL(vpo_sub_co, uint32_t, VPI_SUBCO, VPI_SUBCO)
GL355_VP_OP_KERNEL(vpo_addc, uint32_t, VPI_ADDC, VPI_ADDC1)
GL355_VP_OP_KERNEL(vpo_subb, uint32_t, VPI_SUBB, VPI_SUBB1)
GL355_VP_OP_KERNEL(vpo_cndmask, uint32_t, VPI_CNDMASK, VPI_CNDMASK)
GL355_VP_OP_KERNEL(vpo_cndmask_const, uint32_t, VPI_CNDMASKC, VPI_CNDMASKC)
GL355_VP_OP_KERNEL(vpo_mad_vvv, uint64_t, VPI_MAD, VPI_MAD)
GL355_VP_OP_KERNEL(vpo_mad_svv, uint64_t, VPI_MADS, VPI_MADS)
GL355_VP_OP_KERNEL(vpo_mad_vcv, uint64_t, VPI_MADC, VPI_MADC)
GL355_VP_OP_KERNEL(vpo_mad_vm1v, uint64_t, VPI_MADM1, VPI_MADM1)
GL355_VP_OP_KERNEL(vpo_lshl_add_u64, uint64_t, VPI_LSHLADD64, VPI_LSHLADD64)
GL355_VP_OP_KERNEL(vpo_lshlrev_b64, uint64_t, VPI_LSHL64, VPI_LSHL64)
GL355_VP_OP_KERNEL(vpo_lshrrev_b64, uint64_t, VPI_LSHR64, VPI_LSHR64)
GL355_VP_OP_KERNEL(vpo_mov_b64, uint64_t, VPI_MOV64, VPI_MOV64)
GL355_VP_OP_KERNEL(vpo_cmp_lt_u64, uint64_t, VPI_CMP64, VPI_CMP64)

typedef void (*VpKernel)(uint32_t*, uint32_t, VpClock*);
struct VpOp { const char* name; VpKernel k[3]; };      // ILP 1, 4, 8
#define GL355_VP_ENTRY(STR, K, K1) {STR, {K<1>, K<4>, K<8>}}
static const VpOp VP_OPS[] = {
    GL355_VP_ENTRY("v_add_u32", vpo_add_u32, vpo_add_u32), GL355_VP_ENTRY("v_sub_u32", vpo_sub_u32, vpo_sub_u32),
    GL355_VP_ENTRY("v_and_b32", vpo_and_b32, vpo_and_b32), GL355_VP_ENTRY("v_lshrrev_b32", vpo_lshrrev_b32, vpo_lshrrev_b32),
    GL355_VP_ENTRY("v_ashrrev_i32", vpo_ashrrev_i32, vpo_ashrrev_i32), GL355_VP_ENTRY("v_mov_b32", vpo_mov_b32, vpo_mov_b32),
    GL355_VP_ENTRY("v_lshlrev_b32", vpo_lshlrev_b32, vpo_lshlrev_b32), GL355_VP_ENTRY("v_alignbit_b32", vpo_alignbit_b32, vpo_alignbit_b32),
    GL355_VP_ENTRY("v_add3_u32", vpo_add3_u32, vpo_add3_u32), GL355_VP_ENTRY("v_mul_lo_u32", vpo_mul_lo_u32, vpo_mul_lo_u32),
    GL355_VP_ENTRY("v_add_co_u32 sgpr", vpo_add_co, vpo_add_co), GL355_VP_ENTRY("v_sub_co_u32 sgpr", vpo_sub_co, vpo_sub_co),
    GL355_VP_ENTRY("v_addc_co_u32 sgpr", vpo_addc, vpo_addc), GL355_VP_ENTRY("v_subb_co_u32 sgpr", vpo_subb, vpo_subb),
    GL355_VP_ENTRY("v_cndmask_b32 sgpr", vpo_cndmask, vpo_cndmask), GL355_VP_ENTRY("v_cndmask_b32 0,-1,sgpr", vpo_cndmask_const, vpo_cndmask_const),
    GL355_VP_ENTRY("v_mad_u64_u32 vvv", vpo_mad_vvv, vpo_mad_vvv), GL355_VP_ENTRY("v_mad_u64_u32 svv", vpo_mad_svv, vpo_mad_svv),
    GL355_VP_ENTRY("v_mad_u64_u32 vcv", vpo_mad_vcv, vpo_mad_vcv), GL355_VP_ENTRY("v_mad_u64_u32 v,-1,v", vpo_mad_vm1v, vpo_mad_vm1v),
    GL355_VP_ENTRY("v_lshl_add_u64", vpo_lshl_add_u64, vpo_lshl_add_u64), GL355_VP_ENTRY("v_lshlrev_b64", vpo_lshlrev_b64, vpo_lshlrev_b64),
    GL355_VP_ENTRY("v_lshrrev_b64", vpo_lshrrev_b64, vpo_lshrrev_b64), GL355_VP_ENTRY("v_mov_b64", vpo_mov_b64, vpo_mov_b64),
    GL355_VP_ENTRY("v_cmp_lt_u64 sgpr", vpo_cmp_lt_u64, vpo_cmp_lt_u64),
};
constexpr uint32_t VP_N_OPS = sizeof(VP_OPS) / sizeof(VP_OPS[0]);
static_assert(VP_N_OPS == GL355_VALU_PROBE_OPS, "include/gl355.h: GL355_VALU_PROBE_OPS");

// ---- composite probes: the shipped code itself, operands in registers, no memory --------------------------------------------------------------
// 0: the field product as the hash / quotient kernels issue it (gl_mul_multi<4>, four products in lock-step, dependent from iteration to iteration)
// 1: the Poseidon permutation (psd_permute: the body of every hash kernel), one state per lane, at the occupancy its registers allow
// 2: v_mad_u64_u32 and v_add_u32 alternating in one wave (does a full-rate instruction hide behind a multiply-add?)
// 3: the same two instructions on DIFFERENT waves of a SIMD (even waves multiply-add, odd waves add)
constexpr int VP_PROD_ITERS = 2048, VP_PERM_ITERS = 48;
__global__ void __launch_bounds__(256) vpc_product_kernel(uint32_t* out, uint32_t seed, VpClock* clk) {
    GL355_VP_CLOCK_BEGIN
    uint64_t a[4], b[4], r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { a[j] = (threadIdx.x + 1) * 0x9E3779B97F4A7C15ull + seed + j; b[j] = (blockIdx.x + 3) * 0xC2B2AE3D27D4EB4Full + j; }
#pragma unroll 1
    for (int i = 0; i < VP_PROD_ITERS; i++) {
        gl_mul_multi<4>(a, b, r);
        __builtin_amdgcn_sched_barrier(0);
        gl_mul_multi<4>(r, b, a);
        __builtin_amdgcn_sched_barrier(0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a[0] ^ a[1] ^ a[2] ^ a[3]) ^ (uint32_t)((a[0] ^ a[1] ^ a[2] ^ a[3]) >> 32);
    GL355_VP_CLOCK_END
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) vpc_permute_kernel(uint32_t* out, uint32_t seed, VpClock* clk) {
    GL355_VP_CLOCK_BEGIN
    uint64_t s[12];
#pragma unroll
    for (int j = 0; j < 12; j++) s[j] = (threadIdx.x + 1 + 256ull * blockIdx.x) * 0x9E3779B97F4A7C15ull + seed + j;
#pragma unroll 1
    for (int i = 0; i < VP_PERM_ITERS; i++) psd_permute(s);
    uint64_t x = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) x ^= s[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);
    GL355_VP_CLOCK_END
}
__global__ void __launch_bounds__(256) vpc_alt_kernel(uint32_t* out, uint32_t seed, VpClock* clk) {
    GL355_VP_CLOCK_BEGIN
    uint64_t acc[8];
    uint32_t acc32[8];
    const uint32_t b = (blockIdx.x * 40503u + 12345u) | 1u, c = (seed & 15u) | 3u;
#pragma unroll
    for (int j = 0; j < 8; j++) { acc[j] = threadIdx.x * 2654435761ull + seed + j; acc32[j] = threadIdx.x + j; }
#pragma unroll 1
    for (int i = 0; i < VP_ITERS; i++) {
#define VPI_ALT(A, P) "v_mad_u64_u32 " A ", " P ", %9, %10, " A "\n\tv_add_u32 %8, %8, %9\n\t"
        asm volatile(VP_X4(VPI_ALT("%0", "s[40:41]") VPI_ALT("%1", "s[42:43]") VPI_ALT("%2", "s[44:45]") VPI_ALT("%3", "s[46:47]") VPI_ALT("%4", "s[48:49]")
                           VPI_ALT("%5", "s[50:51]") VPI_ALT("%6", "s[52:53]") VPI_ALT("%7", "s[54:55]"))
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc32[0])
                     : "v"(b), "v"(c) : VP_SCLOB);
    }
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s ^= acc[j] + acc32[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
    GL355_VP_CLOCK_END
}
__global__ void __launch_bounds__(256) vpc_split_kernel(uint32_t* out, uint32_t seed, VpClock* clk) {
    GL355_VP_CLOCK_BEGIN
    uint64_t acc[8];
    uint32_t a32[8];
    const uint32_t b = (blockIdx.x * 40503u + 12345u) | 1u, c = (seed & 15u) | 3u;
#pragma unroll
    for (int j = 0; j < 8; j++) { acc[j] = threadIdx.x * 2654435761ull + seed + j; a32[j] = threadIdx.x + j; }
    if ((threadIdx.x >> 6) & 1) {       // odd waves: 64 plain adds per iteration
#pragma unroll 1
        for (int i = 0; i < VP_ITERS; i++) {
            asm volatile(VP_BODY8(VPI_ADD) : "+v"(a32[0]), "+v"(a32[1]), "+v"(a32[2]), "+v"(a32[3]), "+v"(a32[4]), "+v"(a32[5]), "+v"(a32[6]), "+v"(a32[7])
                         : "v"(b), "v"(c), "s"(seed), "v"(b) : VP_SCLOB);
        }
    } else {                             // even waves: 64 multiply-adds per iteration
#pragma unroll 1
        for (int i = 0; i < VP_ITERS; i++) {
            asm volatile(VP_BODY8(VPI_MAD) : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                         : "v"(b), "v"(c), "s"(seed), "v"(b) : VP_SCLOB);
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s ^= acc[j] + a32[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));
    GL355_VP_CLOCK_END
}


// ---- pattern probes: runs of two opcode forms alternating inside one wave (8 waves per SIMD).  What they decide: whether an instruction that issues at
// the full rate in a loop of its own keeps that rate next to multiply-adds (it does when it comes in runs of >= 2, a lone one costs a whole multiply-add
// slot), and whether v_cndmask on constants behaves like a full-rate instruction in such runs (the lock-step product issues 10 % faster than the sum of
// its instructions' stand-alone costs).  Operands: %0..%7 64-bit chains, %8..%15 32-bit chains, %16 = b, %17 = c; scalar pairs s[40:55].
#define VPP_MAD(J, P) "v_mad_u64_u32 %" #J ", " P ", %16, %17, %" #J "\n\t"
#define VPP_ADD(J, P) "v_add_u32 %" #J ", %" #J ", %16\n\t"
#define VPP_MOV(J, P) "v_mov_b32 %" #J ", %16\n\t"
#define VPP_CND(J, P) "v_cndmask_b32_e64 %" #J ", 0, -1, " P "\n\t"
#define VPP_SUBCO(J, P) "v_sub_co_u32_e64 %" #J ", " P ", %" #J ", %16\n\t"
#define VPP_RUN4A(A) A(8, "s[40:41]") A(9, "s[42:43]") A(10, "s[44:45]") A(11, "s[46:47]")
#define VPP_RUN4B(A) A(12, "s[48:49]") A(13, "s[50:51]") A(14, "s[52:53]") A(15, "s[54:55]")
#define VPP_MAD4A VPP_MAD(0, "s[40:41]") VPP_MAD(1, "s[42:43]") VPP_MAD(2, "s[44:45]") VPP_MAD(3, "s[46:47]")
#define VPP_MAD4B VPP_MAD(4, "s[48:49]") VPP_MAD(5, "s[50:51]") VPP_MAD(6, "s[52:53]") VPP_MAD(7, "s[54:55]")
// 64 instructions per iteration in every pattern
#define VPP_44(A) VP_X4(VPP_RUN4A(A) VPP_MAD4A VPP_RUN4B(A) VPP_MAD4B)
#define VPP_22(A) VP_X4(A(8, "s[40:41]") A(9, "s[42:43]") VPP_MAD(0, "s[40:41]") VPP_MAD(1, "s[42:43]") A(10, "s[44:45]") A(11, "s[46:47]") VPP_MAD(2, "s[44:45]") VPP_MAD(3, "s[46:47]") \
                        A(12, "s[48:49]") A(13, "s[50:51]") VPP_MAD(4, "s[48:49]") VPP_MAD(5, "s[50:51]") A(14, "s[52:53]") A(15, "s[54:55]") VPP_MAD(6, "s[52:53]") VPP_MAD(7, "s[54:55]"))
#define VPP_11(A) VP_X4(A(8, "s[40:41]") VPP_MAD(0, "s[40:41]") A(9, "s[42:43]") VPP_MAD(1, "s[42:43]") A(10, "s[44:45]") VPP_MAD(2, "s[44:45]") A(11, "s[46:47]") VPP_MAD(3, "s[46:47]") \
                        A(12, "s[48:49]") VPP_MAD(4, "s[48:49]") A(13, "s[50:51]") VPP_MAD(5, "s[50:51]") A(14, "s[52:53]") VPP_MAD(6, "s[52:53]") A(15, "s[54:55]") VPP_MAD(7, "s[54:55]"))
#define VPP_CND_SUB VP_X4(VPP_RUN4A(VPP_CND) VPP_SUBCO(12, "s[48:49]") VPP_SUBCO(13, "s[50:51]") VPP_SUBCO(14, "s[52:53]") VPP_SUBCO(15, "s[54:55]") \
                          VPP_RUN4A(VPP_CND) VPP_SUBCO(12, "s[48:49]") VPP_SUBCO(13, "s[50:51]") VPP_SUBCO(14, "s[52:53]") VPP_SUBCO(15, "s[54:55]"))
#define GL355_VP_PATTERN_KERNEL(NAME, BODY)                                                                                     \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed, VpClock* clk) {                                   \
        GL355_VP_CLOCK_BEGIN                                                                                                    \
        uint64_t acc[8];                                                                                                        \
        uint32_t a32[8];                                                                                                        \
        const uint32_t b = (blockIdx.x * 40503u + 12345u) | 1u, c = (seed & 15u) | 3u;                                         \
        _Pragma("unroll") for (int j = 0; j < 8; j++) { acc[j] = threadIdx.x * 2654435761ull + seed + j; a32[j] = threadIdx.x + j; } \
        asm volatile("s_mov_b64 s[40:41], 0x55\n\ts_mov_b64 s[42:43], 0x33\n\ts_mov_b64 s[44:45], 0x0f\n\ts_mov_b64 s[46:47], 0x17\n\t" \
                     "s_mov_b64 s[48:49], 0x71\n\ts_mov_b64 s[50:51], 0x2b\n\ts_mov_b64 s[52:53], 0x4d\n\ts_mov_b64 s[54:55], 0x63" ::: VP_SCLOB); \
        _Pragma("unroll 1") for (int i = 0; i < VP_ITERS; i++)                                                                  \
            asm volatile(BODY : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),         \
                                "+v"(a32[0]), "+v"(a32[1]), "+v"(a32[2]), "+v"(a32[3]), "+v"(a32[4]), "+v"(a32[5]), "+v"(a32[6]), "+v"(a32[7])          \
                         : "v"(b), "v"(c) : VP_SCLOB);                                                                          \
        uint64_t s = 0;                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; j++) s ^= acc[j] + a32[j];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));                                                 \
        GL355_VP_CLOCK_END                                                                                                      \
    }
GL355_VP_PATTERN_KERNEL(vpp_add4_mad4, VPP_44(VPP_ADD))
GL355_VP_PATTERN_KERNEL(vpp_add2_mad2, VPP_22(VPP_ADD))
GL355_VP_PATTERN_KERNEL(vpp_add1_mad1, VPP_11(VPP_ADD))
GL355_VP_PATTERN_KERNEL(vpp_mov4_mad4, VPP_44(VPP_MOV))
GL355_VP_PATTERN_KERNEL(vpp_mov1_mad1, VPP_11(VPP_MOV))
GL355_VP_PATTERN_KERNEL(vpp_cnd4_mad4, VPP_44(VPP_CND))
GL355_VP_PATTERN_KERNEL(vpp_cnd1_mad1, VPP_11(VPP_CND))
GL355_VP_PATTERN_KERNEL(vpp_subco4_mad4, VPP_44(VPP_SUBCO))
GL355_VP_PATTERN_KERNEL(vpp_cnd4_subco4, VPP_CND_SUB)

struct VpComposite { const char* name; VpKernel k; double items_per_lane; };
static const VpComposite VP_COMPOSITES[] = {
    {"product_x4_lockstep", vpc_product_kernel, 8.0 * VP_PROD_ITERS},
    {"poseidon_permutation", vpc_permute_kernel, (double)VP_PERM_ITERS},
    {"mad64_add32_alternating_one_wave", vpc_alt_kernel, 32.0 * VP_ITERS},
    {"mad64_add32_on_different_waves", vpc_split_kernel, 64.0 * VP_ITERS},
    {"pattern 4 add_u32 + 4 mad", vpp_add4_mad4, 64.0 * VP_ITERS},
    {"pattern 2 add_u32 + 2 mad", vpp_add2_mad2, 64.0 * VP_ITERS},
    {"pattern 1 add_u32 + 1 mad", vpp_add1_mad1, 64.0 * VP_ITERS},
    {"pattern 4 mov_b32 + 4 mad", vpp_mov4_mad4, 64.0 * VP_ITERS},
    {"pattern 1 mov_b32 + 1 mad", vpp_mov1_mad1, 64.0 * VP_ITERS},
    {"pattern 4 cndmask(0,-1,sgpr) + 4 mad", vpp_cnd4_mad4, 64.0 * VP_ITERS},
    {"pattern 1 cndmask(0,-1,sgpr) + 1 mad", vpp_cnd1_mad1, 64.0 * VP_ITERS},
    {"pattern 4 sub_co + 4 mad", vpp_subco4_mad4, 64.0 * VP_ITERS},
    {"pattern 4 cndmask(0,-1,sgpr) + 4 sub_co", vpp_cnd4_subco4, 64.0 * VP_ITERS},
};
constexpr uint32_t VP_N_COMPOSITES = sizeof(VP_COMPOSITES) / sizeof(VP_COMPOSITES[0]);
static_assert(VP_N_COMPOSITES == GL355_VALU_PROBE_COMPOSITES, "include/gl355.h: GL355_VALU_PROBE_COMPOSITES");


// ---- pair probes: every unordered pair of the opcode forms that carry the job's instruction count (generated: tools/gen_valu_pairs.py) ------------
#define GL355_VP_PAIR_KERNEL(NAME, BODY)                                                                                        \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed, VpClock* clk) {                                   \
        GL355_VP_CLOCK_BEGIN                                                                                                    \
        uint64_t acc[8];                                                                                                        \
        uint32_t a32[8];                                                                                                        \
        const uint32_t b = (blockIdx.x * 40503u + 12345u) | 1u, c = (seed & 15u) | 3u;                                         \
        const uint64_t bw = ((uint64_t)b << 7) | c;                                                                             \
        _Pragma("unroll") for (int j = 0; j < 8; j++) { acc[j] = threadIdx.x * 2654435761ull + seed + j; a32[j] = threadIdx.x + j; } \
        asm volatile("s_mov_b64 s[40:41], 0x55\n\ts_mov_b64 s[42:43], 0x33\n\ts_mov_b64 s[44:45], 0x0f\n\ts_mov_b64 s[46:47], 0x17\n\t" \
                     "s_mov_b64 s[48:49], 0x71\n\ts_mov_b64 s[50:51], 0x2b\n\ts_mov_b64 s[52:53], 0x4d\n\ts_mov_b64 s[54:55], 0x63" ::: VP_SCLOB); \
        _Pragma("unroll 1") for (int i = 0; i < VP_ITERS; i++)                                                                  \
            asm volatile(BODY : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),         \
                                "+v"(a32[0]), "+v"(a32[1]), "+v"(a32[2]), "+v"(a32[3]), "+v"(a32[4]), "+v"(a32[5]), "+v"(a32[6]), "+v"(a32[7])          \
                         : "v"(b), "v"(c), "v"(bw) : VP_SCLOB);                                                                 \
        uint64_t s = 0;                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; j++) s ^= acc[j] + a32[j];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));                                                 \
        GL355_VP_CLOCK_END                                                                                                      \
    }
struct VpPair { const char* a; const char* b; VpKernel k; };
#include "valu_probe_pairs.inc"
constexpr uint32_t VP_N_PAIRS = sizeof(VP_PAIRS) / sizeof(VP_PAIRS[0]);
static_assert(VP_N_PAIRS == GL355_VALU_PROBE_PAIRS, "include/gl355.h: GL355_VALU_PROBE_PAIRS");

}  // namespace gl355

using namespace gl355;

extern "C" {

int32_t gl355_valu_probe(gl355_ctx* h, double rates_ginst_per_s[GL355_VALU_CLASSES], double shader_mhz[GL355_VALU_CLASSES]) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!rates_ginst_per_s || !shader_mhz) return ctx->fail(GL355_E_INVALID_ARG, "valu_probe: null argument");
    Scratch sc(ctx);
    GL355_TRY(sc.get((size_t)VP_BLOCKS * 256 * 4 + 64));
    uint32_t* d_out = sc.as<uint32_t>();
    VpClock* d_clk = reinterpret_cast<VpClock*>(d_out + (size_t)VP_BLOCKS * 256);
    hipEvent_t e0 = ctx->prof_event(), e1 = ctx->prof_event();
    if (!e0 || !e1) return ctx->fail(GL355_E_HIP, "valu_probe: no events");
    const double insts = (double)VP_BLOCKS * 4 /* waves */ * VP_ITERS * VP_UNROLL * VP_ILP;
    for (int c = 0; c < GL355_VALU_CLASSES; c++) {
        double best = 0, best_mhz = 0;
        for (int rep = 0; rep < 4; rep++) {                 // rep 0 warms the clocks up
            GL355_HIP(ctx, hipEventRecord(e0, ctx->stream));
            if (c == GL355_VALU_FULL32) hipLaunchKernelGGL(vp_full32_kernel, dim3(VP_BLOCKS), dim3(256), 0, ctx->stream, d_out, 17u + rep, d_clk);
            else if (c == GL355_VALU_HALF32) hipLaunchKernelGGL(vp_half32_kernel, dim3(VP_BLOCKS), dim3(256), 0, ctx->stream, d_out, 17u + rep, d_clk);
            else hipLaunchKernelGGL(vp_mad64_kernel, dim3(VP_BLOCKS), dim3(256), 0, ctx->stream, d_out, 17u + rep, d_clk);
            GL355_HIP(ctx, hipEventRecord(e1, ctx->stream));
            GL355_HIP(ctx, hipGetLastError());
            VpClock hc;
            GL355_HIP(ctx, ctx->d2h(&hc, d_clk, sizeof hc));
            GL355_HIP(ctx, ctx->wait());
            float ms = 0;
            GL355_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
            const double rate = ms > 0 ? insts / (ms * 1e-3) / 1e9 : 0;
            if (rep > 0 && rate > best) { best = rate; best_mhz = hc.ticks ? (double)hc.cyc / ((double)hc.ticks / 100.0) : 0; }
        }
        rates_ginst_per_s[c] = best;
        shader_mhz[c] = best_mhz;
    }
    ctx->ev_pool.push_back(e0);
    ctx->ev_pool.push_back(e1);
    return GL355_OK;
}

// one launch of `k` over `blocks` x 256 lanes, timed with HIP events on the context's stream; best of three after one warm-up launch
static int32_t vp_time(Ctx* ctx, VpKernel k, uint32_t blocks, uint32_t* d_out, VpClock* d_clk, double* best_ms, double* best_mhz) {
    hipEvent_t e0 = ctx->prof_event(), e1 = ctx->prof_event();
    if (!e0 || !e1) return ctx->fail(GL355_E_HIP, "valu_probe: no events");
    *best_ms = 0; *best_mhz = 0;
    for (int rep = 0; rep < 4; rep++) {
        GL355_HIP(ctx, hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, ctx->stream, d_out, 17u + rep, d_clk);
        GL355_HIP(ctx, hipEventRecord(e1, ctx->stream));
        GL355_HIP(ctx, hipGetLastError());
        VpClock hc;
        GL355_HIP(ctx, ctx->d2h(&hc, d_clk, sizeof hc));
        GL355_HIP(ctx, ctx->wait());
        float ms = 0;
        GL355_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms > 0 && (*best_ms == 0 || ms < *best_ms)) { *best_ms = ms; *best_mhz = hc.ticks ? (double)hc.cyc / ((double)hc.ticks / 100.0) : 0; }
    }
    ctx->ev_pool.push_back(e0);
    ctx->ev_pool.push_back(e1);
    return GL355_OK;
}

const char* gl355_valu_probe_op_name(uint32_t i) { return i < VP_N_OPS ? VP_OPS[i].name : nullptr; }
const char* gl355_valu_probe_composite_name(uint32_t i) { return i < VP_N_COMPOSITES ? VP_COMPOSITES[i].name : nullptr; }

int32_t gl355_valu_probe_ops(gl355_ctx* h, uint32_t ilp, double rates_ginst_per_s[GL355_VALU_PROBE_OPS], double shader_mhz[GL355_VALU_PROBE_OPS]) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!rates_ginst_per_s || !shader_mhz || (ilp != 1 && ilp != 4 && ilp != 8)) return ctx->fail(GL355_E_INVALID_ARG, "valu_probe_ops: ilp is 1, 4 or 8");
    Scratch sc(ctx);
    GL355_TRY(sc.get((size_t)VP_BLOCKS * 256 * 4 + 64));
    uint32_t* d_out = sc.as<uint32_t>();
    VpClock* d_clk = reinterpret_cast<VpClock*>(d_out + (size_t)VP_BLOCKS * 256);
    const double insts = (double)VP_BLOCKS * 4 /* waves */ * VP_ITERS * VP_OP_BODY;
    const int slot = ilp == 1 ? 0 : (ilp == 4 ? 1 : 2);
    for (uint32_t i = 0; i < VP_N_OPS; i++) {
        double ms, mhz;
        GL355_TRY(vp_time(ctx, VP_OPS[i].k[slot], VP_BLOCKS, d_out, d_clk, &ms, &mhz));
        rates_ginst_per_s[i] = ms > 0 ? insts / (ms * 1e-3) / 1e9 : 0;
        shader_mhz[i] = mhz;
    }
    return GL355_OK;
}

int32_t gl355_valu_probe_pair_names(uint32_t i, const char** form_a, const char** form_b) {
    if (i >= VP_N_PAIRS || !form_a || !form_b) return GL355_E_INVALID_ARG;
    *form_a = VP_PAIRS[i].a; *form_b = VP_PAIRS[i].b;
    return GL355_OK;
}
int32_t gl355_valu_probe_pairs(gl355_ctx* h, double rates_ginst_per_s[GL355_VALU_PROBE_PAIRS], double shader_mhz[GL355_VALU_PROBE_PAIRS]) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!rates_ginst_per_s || !shader_mhz) return ctx->fail(GL355_E_INVALID_ARG, "valu_probe_pairs: null argument");
    Scratch sc(ctx);
    GL355_TRY(sc.get((size_t)VP_BLOCKS * 256 * 4 + 64));
    uint32_t* d_out = sc.as<uint32_t>();
    VpClock* d_clk = reinterpret_cast<VpClock*>(d_out + (size_t)VP_BLOCKS * 256);
    const double insts = (double)VP_BLOCKS * 4 /* waves */ * VP_ITERS * 64;
    for (uint32_t i = 0; i < VP_N_PAIRS; i++) {
        double ms, mhz;
        GL355_TRY(vp_time(ctx, VP_PAIRS[i].k, VP_BLOCKS, d_out, d_clk, &ms, &mhz));
        rates_ginst_per_s[i] = ms > 0 ? insts / (ms * 1e-3) / 1e9 : 0;
        shader_mhz[i] = mhz;
    }
    return GL355_OK;
}

int32_t gl355_valu_probe_composite(gl355_ctx* h, uint32_t which, double* items_g_per_s, double* shader_mhz, uint32_t* waves_per_simd) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!items_g_per_s || !shader_mhz || which >= GL355_VALU_PROBE_COMPOSITES) return ctx->fail(GL355_E_INVALID_ARG, "valu_probe_composite: bad argument");
    VpKernel k = VP_COMPOSITES[which].k;
    int per_cu = 0, n_cu = 0;
    GL355_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 256, 0));
    GL355_HIP(ctx, hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device));
    if (per_cu < 1 || n_cu < 1) return ctx->fail(GL355_E_HIP, "valu_probe_composite: no occupancy");
    if (per_cu > 8) per_cu = 8;
    const uint32_t blocks = (uint32_t)per_cu * (uint32_t)n_cu;      // exactly one resident set: no second wave of blocks, no tail
    Scratch sc(ctx);
    GL355_TRY(sc.get((size_t)blocks * 256 * 4 + 64));
    uint32_t* d_out = sc.as<uint32_t>();
    VpClock* d_clk = reinterpret_cast<VpClock*>(d_out + (size_t)blocks * 256);
    double ms, mhz;
    GL355_TRY(vp_time(ctx, k, blocks, d_out, d_clk, &ms, &mhz));
    const double per_lane = VP_COMPOSITES[which].items_per_lane;      // lane-level products / permutations / instruction pairs / instructions
    *items_g_per_s = ms > 0 ? per_lane * blocks * 256.0 / (ms * 1e-3) / 1e9 : 0;
    *shader_mhz = mhz;
    if (waves_per_simd) *waves_per_simd = (uint32_t)per_cu;        // 4 waves per block, 4 SIMDs per CU
    return GL355_OK;
}

int32_t gl355_clock_probe(gl355_ctx* h, uint32_t micros, double* shader_mhz) {
    Ctx* ctx = ctx_of(h);
    if (!ctx) return GL355_E_INVALID_ARG;
    if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(GL355_E_HIP, "hipSetDevice failed");
    if (!shader_mhz || micros == 0 || micros > 1000000) return ctx->fail(GL355_E_INVALID_ARG, "clock_probe: needs 1 .. 1e6 microseconds and an output");
    Scratch sc(ctx);
    GL355_TRY(sc.get(64));
    VpClock* d_clk = sc.as<VpClock>();
    hipLaunchKernelGGL(vp_clock_kernel, dim3(1), dim3(64), 0, ctx->stream, (unsigned long long)micros * 100ull, d_clk);
    GL355_HIP(ctx, hipGetLastError());
    VpClock hc;
    GL355_HIP(ctx, ctx->d2h(&hc, d_clk, sizeof hc));
    GL355_HIP(ctx, ctx->wait());
    *shader_mhz = hc.ticks ? (double)hc.cyc / ((double)hc.ticks / 100.0) : 0;
    return GL355_OK;
}

}  // extern "C"
