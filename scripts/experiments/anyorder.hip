// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (answer recorded in DESIGN.md)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long cycles, int *out) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (out) out[blockIdx.x] = 1;
}
int main() {
    int *d; hipMalloc(&d, 4096);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long long cyc = 100 * 300; // wall_clock64 ticks at 100 MHz -> 300 us
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, d);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, d);
            else hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d (%s): two 300 us kernels on one stream took %.3f ms\n", mode, mode ? "any-order" : "ordered", ms);
        }
    }
    return 0;
}
