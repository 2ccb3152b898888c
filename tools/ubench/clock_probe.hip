// Shader clock under whatever else runs on the device: a one-wave kernel spins for ~2 ms of the 100-MHz real-time counter (s_memrealtime)
// and reports the shader-clock cycles (s_memtime) that passed meanwhile; repeated every 100 ms for argv[1] seconds.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/clock_probe.hip -o tools/ubench/bin/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
__global__ void probe(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long w1 = w0;
    while (w1 - w0 < ticks) { __builtin_amdgcn_s_sleep(8); w1 = wall_clock64(); }
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 10;
    unsigned long long *d, h[2];
    if (hipMalloc(&d, 16) != hipSuccess) return 1;
    double lo = 1e9, hi = 0, sum = 0; int n = 0;
    for (int i = 0; i < (int)(secs * 10); i++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, 200000ull, d);
        if (hipMemcpy(h, d, 16, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        const double mhz = h[0] / (h[1] / 100.0);
        if (mhz < lo) lo = mhz; if (mhz > hi) hi = mhz; sum += mhz; n++;
        if (i % 10 == 0) { printf("t=%4.1fs shader clock %.0f MHz\n", i / 10.0, mhz); fflush(stdout); }
        usleep(100000);
    }
    printf("shader clock over %d samples: mean %.0f MHz, min %.0f, max %.0f\n", n, sum / n, lo, hi);
    return 0;
}
