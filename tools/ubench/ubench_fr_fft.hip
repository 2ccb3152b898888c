// Round 6 diagnostic: where a pass of the bn256::Fr transform (fr_fft_pass_kernel, decimation in frequency on resident Montgomery data, 2^23 points:
// passes of 6, 7 and 10 stages) loses the 25 - 30 % between its time and the VALU ceiling of its instruction mix.  The pass is restated here with
// switches that REMOVE one thing at a time (results are then wrong; timing only; V & 16: the coset form's block constants instead of per-position twiddles): V & 1 the twiddle loads (one register value instead), V & 2 the barriers
// between stages, V & 4 the LDS traffic (operands stay in registers), V & 8 the product (a sum instead).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I stark-verifier_amd/csrc -I include tools/ubench/ubench_fr_fft.hip -o tools/ubench/bin/ubench_fr_fft
#include "bn254_field.cuh"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gl355;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Pass { const uint64_t* in; uint64_t* out; const uint64_t* tw; uint32_t log_n, s0, ns; };

template <int V>
__global__ void __launch_bounds__(256) pass_kernel(Pass a) {
    __shared__ uint32_t lds[8][1024];
    const uint32_t C = 1024u >> a.ns, log_c = 31 - __clz(C);
    const uint32_t tid = threadIdx.x;
    const uint32_t cblks = (1u << a.s0) >> log_c;
    const uint64_t hi = blockIdx.x / cblks, c0 = (uint64_t)(blockIdx.x % cblks) << log_c;
    const uint64_t base = (hi << (a.s0 + a.ns)) + c0;
    for (uint32_t e = tid; e < 1024; e += 256) {
        const uint32_t r = e >> log_c, c = e & (C - 1);
        const u256 x = load256(a.in + 4 * (base + ((uint64_t)r << a.s0) + c));
#pragma unroll
        for (int l = 0; l < 8; l++) lds[l][e] = x.l[l];
    }
    __syncthreads();
    u256 wfix = load256(a.tw + 4 * (tid + 1));
    u256 keep_u = load256(a.tw + 4 * (tid + 300)), keep_v = load256(a.tw + 4 * (tid + 700));
    // V & 32 (with V & 16): the two block constants of the NEXT stage are requested before this stage's butterflies
    auto btw_of = [&](uint32_t st, uint32_t b) {
        const uint32_t q = b >> log_c, lh = st - 1;
        return load256(a.tw + 4 * ((1ull << (a.log_n - a.s0 - st)) + (hi << (a.ns - st)) + (q >> lh)));
    };
    u256 wn[2];
    if (V & 32) { wn[0] = btw_of(a.ns, tid); wn[1] = btw_of(a.ns, tid + 256); }
    for (uint32_t it = 1; it <= a.ns; it++) {
        const uint32_t st = a.ns + 1 - it;
        const uint32_t s = a.s0 + st, lh = st - 1, half = 1u << lh;
        u256 wc[2];
        if (V & 32) {
            wc[0] = wn[0]; wc[1] = wn[1];
            if (st > 1) { wn[0] = btw_of(st - 1, tid); wn[1] = btw_of(st - 1, tid + 256); }
        }
#pragma unroll
        for (uint32_t b = tid; b < 512; b += 256) {
            const uint32_t q = b >> log_c, c = b & (C - 1);
            const uint32_t pos = q & (half - 1);
            const uint32_t r_lo = ((q >> lh) << (lh + 1)) | pos;
            const uint32_t e0 = (r_lo << log_c) | c, e1 = e0 + (half << log_c);
            const uint64_t j = ((uint64_t)pos << a.s0) + c0 + c;
            u256 w;
            if (V & 1) w = wfix;
            else if (V & 32) w = wc[b >= 256];
            else if (V & 16) w = load256(a.tw + 4 * ((1ull << (a.log_n - s)) + (hi << (a.ns - st)) + (q >> lh)));      // block constants (FrPass::btw)
            else w = load256(a.tw + 4 * (j << (a.log_n - s)));
            u256 u, v;
            if (V & 4) { u = keep_u; v = keep_v; }
            else {
#pragma unroll
                for (int l = 0; l < 8; l++) { u.l[l] = lds[l][e0]; v.l[l] = lds[l][e1]; }
            }
            u256 p, m;
            if (V & 16) { v = m_mul<F_R>(v, w); p = m_add<F_R>(u, v); m = m_sub<F_R>(u, v); }
            else {
                p = m_add<F_R>(u, v); m = m_sub<F_R>(u, v);
                if (V & 8) m = m_add<F_R>(m, w); else m = m_mul<F_R>(m, w);
            }
            if (V & 4) { keep_u = p; keep_v = m; }
            else {
#pragma unroll
                for (int l = 0; l < 8; l++) { lds[l][e0] = p.l[l]; lds[l][e1] = m.l[l]; }
            }
        }
        if (!(V & 2)) __syncthreads();
    }
    if (V & 4) {
#pragma unroll
        for (int l = 0; l < 8; l++) { lds[l][tid] = keep_u.l[l]; lds[l][tid + 256] = keep_v.l[l]; }
    }
    __syncthreads();
    for (uint32_t e = tid; e < 1024; e += 256) {
        const uint32_t r = e >> log_c, c = e & (C - 1);
        u256 x;
#pragma unroll
        for (int l = 0; l < 8; l++) x.l[l] = lds[l][e];
        store256(a.out + 4 * (base + ((uint64_t)r << a.s0) + c), x);
    }
}
__global__ void fill_kernel(uint64_t* p, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = (i & 3) == 3 ? (z >> 4) : z;            // 4 words per element, top word < 2^60: below the modulus
    }
}
template <typename F> static float timeit(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; i++) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b));
    CK(hipGetLastError());
    return t / reps;
}
template <int V> static void run(const char* what, Pass p) {
    const uint32_t tiles = (1u << p.log_n) / 1024;
    const float t = timeit([&] { hipLaunchKernelGGL(pass_kernel<V>, dim3(tiles), dim3(256), 0, 0, p); }, 30);
    printf("  %-44s %.3f ms\n", what, t);
}
int main() {
    const uint32_t log_n = 23;
    const uint64_t n = 1ull << log_n;
    uint64_t *x, *tw;
    CK(hipMalloc(&x, n * 32)); CK(hipMalloc(&tw, (n + 1024) * 32));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, x, n * 4);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, tw, (n + 1024) * 4);
    CK(hipDeviceSynchronize());
    const uint32_t s0s[3] = {17, 10, 0}, nss[3] = {6, 7, 10};
    for (int k = 0; k < 3; k++) {
        Pass p{x, x, tw, log_n, s0s[k], nss[k]};
        printf("pass s0 = %u, %u stages (VALU ceiling of the mix: %.3f ms at 2.3 GHz, 3.33 clk per instruction)\n", s0s[k], nss[k], nss[k] * 0.0355);
        run<0>("as shipped", p);
        run<1>("no twiddle loads", p);
        run<2>("no barriers between stages", p);
        run<3>("no twiddle loads, no barriers", p);
        run<4>("no LDS traffic", p);
        run<7>("no twiddles, barriers, LDS", p);
        run<8>("no product (a sum instead)", p);
        run<15>("loads and stores only", p);
        run<16>("block constants (coset form, as shipped)", p);
        run<48>("block constants, next stage's prefetched", p);
    }
    return 0;
}
