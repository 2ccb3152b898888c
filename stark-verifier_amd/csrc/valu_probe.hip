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
