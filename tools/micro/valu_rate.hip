// Issue cost of transcendental and packed VALU instructions relative to v_fma_f32 on gfx950, per SIMD with 4 resident waves.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/micro/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) {          // v_fma_f32 x8 independent
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                             "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 1) {   // v_rcp_f32
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 2) {   // v_rsq_f32
                asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                             "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 3) {   // v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                             "v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
            } else if (OP == 4) {   // 1 rcp among 7 fma: does the transcendental overlap with plain ops of the same wave?
                asm volatile("v_rcp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                             "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 5) {   // v_fma_mix_f32 (f16 operands)
                asm volatile("v_fma_mix_f32 %0, %0, %0, %0\n v_fma_mix_f32 %1, %1, %1, %1\n v_fma_mix_f32 %2, %2, %2, %2\n v_fma_mix_f32 %3, %3, %3, %3\n"
                             "v_fma_mix_f32 %4, %4, %4, %4\n v_fma_mix_f32 %5, %5, %5, %5\n v_fma_mix_f32 %6, %6, %6, %6\n v_fma_mix_f32 %7, %7, %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 6) {   // dependent v_pk_fma_f32 chain (one register): the stall the compiler fills with s_nop
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0\n"
                             "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %0, %0, %0, %0"
                             : "+v"(p0));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
}

template <int OP>
static double run(const char* name, float* out, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // blocks / 256 CUs resident blocks per CU, 4 waves per block -> one wave per SIMD per block
    double instr_per_simd = (double)iters * 64.0 * (blocks / 256.0);
    double ns_per_instr = ms * 1e6 / instr_per_simd;
    printf("%-28s %8.3f ms  %.3f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms, ns_per_instr, ns_per_instr * 2.4);
    return ns_per_instr;
}

int main()
{
    float* out;
    int blocks = 256 * 4;                 // 4 blocks of 4 waves per CU: 4 waves per SIMD
    hipMalloc(&out, blocks * 256 * sizeof(float));
    int iters = 20000;
    double f = run<0>("v_fma_f32", out, blocks, iters);
    double r = run<1>("v_rcp_f32", out, blocks, iters);
    double q = run<2>("v_rsq_f32", out, blocks, iters);
    double p = run<3>("v_pk_fma_f32", out, blocks, iters);
    double m = run<4>("1 v_rcp + 7 v_fma", out, blocks, iters);
    double x = run<5>("v_fma_mix_f32", out, blocks, iters);
    double d = run<6>("dependent v_pk_fma_f32", out, blocks, iters);
    printf("relative to v_fma_f32: rcp %.2f  rsq %.2f  pk_fma %.2f  (1 rcp + 7 fma)/8 %.2f  fma_mix %.2f  dependent pk_fma %.2f\n", r / f, q / f, p / f, m / f, x / f, d / f);
    return 0;
}
