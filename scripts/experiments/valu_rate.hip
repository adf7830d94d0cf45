// How fast does one SIMD of gfx950 issue the lane bodies' instructions?  W waves a SIMD (W workgroups of 256 threads a CU),
// every wave a loop of independent wave64 VALU instructions of one kind; prints nanoseconds and (at the measured
// engine clock) clocks per wave instruction and SIMD.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate scripts/experiments/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters, unsigned long long *clk)
{
    float x[8], y = threadIdx.x * 0.5f, z = blockIdx.x * 0.25f + 1.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = threadIdx.x + k;
    const unsigned long long m = __ballot(threadIdx.x & 1);
    if (KIND == 11) asm volatile("s_mov_b64 vcc, %0" :: "s"(m) : "vcc");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(y));
                if (KIND == 1) asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(x[k]) : "v"(x[k]), "v"(y), "v"(z));
                if (KIND == 2) asm volatile("v_add_f32_e64 %0, |%1|, %2" : "=v"(x[k]) : "v"(x[k]), "v"(y));
                if (KIND == 3) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[k]) : "v"(x[k]), "v"(y));
                if (KIND == 4) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(y));
                if (KIND == 5) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf" : "=v"(x[k]) : "v"(x[k]));
                if (KIND == 6) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(x[k]) : "v"(x[k]), "v"(y), "s"(m));
                if (KIND == 7) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[k]) : "v"(x[k]), "v"(y) : "vcc");
                if (KIND == 8) asm volatile("v_bfi_b32 %0, %3, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(y), "v"(z));
                if (KIND == 9) asm volatile("v_min_f32 %0, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(y));
                if (KIND == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(x[k]) : "v"(y));
                if (KIND == 11) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[k]) : "v"(x[k]), "v"(y) : );
                if (KIND == 12) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(x[k]) : "v"(x[(k + 1) & 7]), "v"(y), "s"(m));
                if (KIND == 13) asm volatile("v_and_b32 %0, %1, %2" : "=v"(x[k]) : "v"(x[k]), "v"(y));
                if (KIND == 14) { if (k == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x[0]), "v"(y) : "vcc"); else asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[k]) : "v"(x[k]), "v"(y)); }
                if (KIND == 15) { if (k == 0) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" :: "v"(x[0]), "v"(y) : "s20", "s21"); else asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(x[k]) : "v"(x[k]), "v"(y)); }
                if (KIND == 16) asm volatile("v_cndmask_b32_e64 %0, %1, %2, vcc" : "=v"(x[k]) : "v"(x[k]), "v"(y));
                if (KIND == 17) asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(x[k]) : "v"(x[k]), "v"(y) : "vcc");
                if (KIND == 18) asm volatile("v_addc_co_u32_e64 %0, s[22:23], %1, %2, %3" : "=v"(x[k]) : "v"(x[k]), "v"(y), "s"(m) : "s22", "s23");
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += x[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int KIND>
static void run(const char *name, int cus)
{
    const int iters = 20000;
    float *d; unsigned long long *dc, hc = 0;
    hipMalloc(&d, (size_t)cus * 8 * 256 * 4); hipMalloc(&dc, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int W : {1, 4}) {
        hipLaunchKernelGGL(k_rate<KIND>, dim3(cus * W), dim3(256), 0, 0, d, 100, dc); // warm
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(cus * W), dim3(256), 0, 0, d, iters, dc);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
        const double n_inst = (double)iters * 32;            // wave instructions a wave
        const double ns_simd = ms * 1e6 / (n_inst * W);       // per wave instruction and SIMD (W waves share it)
        printf("%-22s W=%d waves/SIMD: %8.3f ms  %.3f ns per wave-instruction per SIMD  (s_memtime ticks of wave 0: %llu = %.2f per instr)\n", name, W, ms, ns_simd, hc,
               (double)hc / n_inst);
    }
    hipFree(d); hipFree(dc);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    run<0>("v_add_f32", p.multiProcessorCount);
    run<1>("v_min3_f32", p.multiProcessorCount);
    run<2>("v_add_f32 |a|", p.multiProcessorCount);
    run<3>("v_cndmask_b32", p.multiProcessorCount);
    run<4>("v_sub_f32", p.multiProcessorCount);
    run<5>("v_mov_b32 dpp quad_perm", p.multiProcessorCount);
    run<6>("v_cndmask e64 sgpr", p.multiProcessorCount);
    run<7>("v_cmp + v_cndmask vcc", p.multiProcessorCount);
    run<8>("v_bfi_b32", p.multiProcessorCount);
    run<9>("v_min_f32", p.multiProcessorCount);
    run<10>("v_mov_b32", p.multiProcessorCount);
    run<11>("v_cndmask vcc (set)", p.multiProcessorCount);
    run<12>("v_cndmask e64 indep", p.multiProcessorCount);
    run<13>("v_and_b32", p.multiProcessorCount);
    run<14>("1 v_cmp vcc + 7 cndmask vcc", p.multiProcessorCount);
    run<15>("1 v_cmp s + 7 cndmask e64 s", p.multiProcessorCount);
    run<16>("v_cndmask e64 with vcc", p.multiProcessorCount);
    run<17>("v_addc vcc in/out", p.multiProcessorCount);
    run<18>("v_addc e64 sgpr in/out", p.multiProcessorCount);
    return 0;
}
