// Issue cost of VALU instruction classes on gfx950, per SIMD with 4 resident waves (eight independent registers per wave).
// Measured (ns per wave64 instruction per SIMD): v_mul/v_add/v_mov 1.00, v_fma_f32 1.29, most other VOP1/2/3 ops, every packed
// op and v_fma_mix 1.75-1.80, v_rcp/v_rsq 3.43.  Curiosity: a VOP2 v_cndmask_b32 that reads a VCC no VALU op has written since
// the previous read costs 9.8 ns (2.2 ns extra per stale reader after one v_cmp); the VOP3 form, or a v_cmp directly in
// front, is 1.78 — the loss kernel's selects all follow their own compare, and moving them to SGPR-pair masks changed nothing.
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


// one-register-in, one-register-out ops (and a few fixed forms), eight independent registers
#define RATE8(NAME, ASM)                                                                                                  \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters, float seed)                                       \
    {                                                                                                                     \
        float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,        \
              a6 = a0 + 6.f, a7 = a0 + 7.f;                                                                               \
        for (int i = 0; i < iters; i++) {                                                                                 \
            _Pragma("unroll") for (int u = 0; u < 8; u++)                                                                 \
                asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                      \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                             :                                                                                            \
                             : "vcc", "s20", "s21");                                                                                    \
        }                                                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                      \
    }
#define A_CVT_UB(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n"
#define A_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %" #k ", vcc\n"
#define A_CND64(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %" #k ", s[20:21]\n"
#define A_CNDX(k) "v_cndmask_b32 %" #k ", 1.0, %" #k ", vcc\n"
#define A_ADD(k) "v_add_f32 %" #k ", %" #k ", %" #k "\n"
#define A_FMAC(k) "v_fmac_f32 %" #k ", %" #k ", %" #k "\n"
#define A_CND64V(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %" #k ", vcc\n"
#define A_CMPCND(k) "v_cmp_gt_f32 vcc, 0.5, %" #k "\n v_cndmask_b32 %" #k ", %" #k ", %" #k ", vcc\n"
#define A_CMPCND64(k) "v_cmp_gt_f32_e64 s[20:21], 0.5, %" #k "\n v_cndmask_b32_e64 %" #k ", %" #k ", %" #k ", s[20:21]\n"
#define A_CMP3CND(k) "v_cmp_gt_f32 vcc, 0.5, %" #k "\n v_cndmask_b32 %" #k ", %" #k ", %" #k ", vcc\n v_cndmask_b32 %" #k ", %" #k ", %" #k ", vcc\n v_cndmask_b32 %" #k ", %" #k ", %" #k ", vcc\n"
#define A_CMP3CND64(k) "v_cmp_gt_f32 vcc, 0.5, %" #k "\n v_cndmask_b32_e64 %" #k ", %" #k ", %" #k ", vcc\n v_cndmask_b32_e64 %" #k ", %" #k ", %" #k ", vcc\n v_cndmask_b32_e64 %" #k ", %" #k ", %" #k ", vcc\n"
#define A_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %" #k ", %" #k "\n"
#define A_MED3(k) "v_med3_f32 %" #k ", %" #k ", %" #k ", %" #k "\n"
#define A_BFI(k) "v_bfi_b32 %" #k ", %" #k ", %" #k ", %" #k "\n"
#define A_CVT_I32(k) "v_cvt_i32_f32 %" #k ", %" #k "\n"
#define A_FRACT(k) "v_fract_f32 %" #k ", %" #k "\n"
#define A_CMP(k) "v_cmp_gt_f32 vcc, %" #k ", %" #k "\n"
#define A_MUL24(k) "v_mul_u32_u24 %" #k ", %" #k ", %" #k "\n"
#define A_ADDLSHL(k) "v_add_lshl_u32 %" #k ", %" #k ", %" #k ", 2\n"
#define A_MIN(k) "v_min_f32 %" #k ", %" #k ", %" #k "\n"
#define A_MUL(k) "v_mul_f32 %" #k ", %" #k ", %" #k "\n"
#define A_MOV(k) "v_mov_b32 %" #k ", %" #k "\n"
#define A_PKADDH(k) "v_pk_add_f16 %" #k ", %" #k ", %" #k "\n"
#define A_PKMUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %" #k "\n"
RATE8(k_cvt_ub, A_CVT_UB)
RATE8(k_cndmask, A_CNDMASK)
RATE8(k_cnd64, A_CND64)
RATE8(k_cndx, A_CNDX)
RATE8(k_add, A_ADD)
RATE8(k_fmac, A_FMAC)
RATE8(k_cnd64v, A_CND64V)
RATE8(k_cmpcnd, A_CMPCND)
RATE8(k_cmpcnd64, A_CMPCND64)
RATE8(k_cmp3cnd, A_CMP3CND)
RATE8(k_cmp3cnd64, A_CMP3CND64)
RATE8(k_max3, A_MAX3)
RATE8(k_med3, A_MED3)
RATE8(k_bfi, A_BFI)
RATE8(k_cvt_i32, A_CVT_I32)
RATE8(k_fract, A_FRACT)
RATE8(k_cmp, A_CMP)
RATE8(k_mul24, A_MUL24)
RATE8(k_addlshl, A_ADDLSHL)
RATE8(k_min, A_MIN)
RATE8(k_mul, A_MUL)
RATE8(k_mov, A_MOV)
RATE8(k_pkaddh, A_PKADDH)

template <typename K>
static double run8(const char* name, K kern, float* out, int blocks, int iters, double fma)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double ns = ms * 1e6 / ((double)iters * 64.0 * (blocks / 256.0));
    printf("%-28s %8.3f ms  %.3f ns per wave-instruction per SIMD  (%.2f x v_fma_f32)\n", name, ms, ns, ns / fma);
    return ns;
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
    run8("v_cvt_f32_ubyte1", k_cvt_ub, out, blocks, iters, f);
    run8("v_cndmask_b32 (vcc)", k_cndmask, out, blocks, iters, f);
    run8("v_cndmask_b32_e64 s[20:21]", k_cnd64, out, blocks, iters, f);
    run8("v_cndmask_b32 1.0, v, vcc", k_cndx, out, blocks, iters, f);
    run8("v_add_f32", k_add, out, blocks, iters, f);
    run8("v_fmac_f32", k_fmac, out, blocks, iters, f);
    run8("v_cndmask_b32_e64 .., vcc", k_cnd64v, out, blocks, iters, f);
    run8("[v_cmp->vcc + v_cndmask vcc] /2", k_cmpcnd, out, blocks, iters / 2, f);
    run8("[v_cmp->s[20:21] + cndmask e64] /2", k_cmpcnd64, out, blocks, iters / 2, f);
    run8("[v_cmp->vcc + 3 cndmask e32 vcc] /4", k_cmp3cnd, out, blocks, iters / 4, f);
    run8("[v_cmp->vcc + 3 cndmask e64 vcc] /4", k_cmp3cnd64, out, blocks, iters / 4, f);
    run8("v_max3_f32", k_max3, out, blocks, iters, f);
    run8("v_med3_f32", k_med3, out, blocks, iters, f);
    run8("v_bfi_b32", k_bfi, out, blocks, iters, f);
    run8("v_cvt_i32_f32", k_cvt_i32, out, blocks, iters, f);
    run8("v_fract_f32", k_fract, out, blocks, iters, f);
    run8("v_cmp_gt_f32 -> vcc", k_cmp, out, blocks, iters, f);
    run8("v_mul_u32_u24", k_mul24, out, blocks, iters, f);
    run8("v_add_lshl_u32", k_addlshl, out, blocks, iters, f);
    run8("v_min_f32", k_min, out, blocks, iters, f);
    run8("v_mul_f32", k_mul, out, blocks, iters, f);
    run8("v_mov_b32", k_mov, out, blocks, iters, f);
    run8("v_pk_add_f16", k_pkaddh, out, blocks, iters, f);
    printf("relative to v_fma_f32: rcp %.2f  rsq %.2f  pk_fma %.2f  (1 rcp + 7 fma)/8 %.2f  fma_mix %.2f  dependent pk_fma %.2f\n", r / f, q / f, p / f, m / f, x / f, d / f);
    return 0;
}
