// pcl_loss.hip — fused projection + bilinear sampling + sampling loss (+ hand-derived gradient) for gfx950.
//
// What it replaces in the reference (all un-fused ATen ops over (B,N,3) temporaries):
//   omniloc.py:190-200 / :332-353  p = R (x - t) -> cloud2idx -> sample_from_img -> mask -> ||c - rgb|| -> mean
//   omniloc.py:47,254              loss.backward()  (autograd through the above, incl. grid_sampler_2d_backward)
//
// Where the time goes on MI355X (measured, profiles/): the cloud (24 B/point) and the panorama sit in L2 / Infinity
// Cache, and ~170 fp32 operations per point-pose make the kernel VALU-ISSUE bound: an unpacked fp32 VALU
// instruction occupies a SIMD for 4 cycles per wave64.  So the design minimises issue slots:
//   - every lane carries TWO points and evaluates them against the SAME pose with packed fp32 math
//     (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two results per issue slot); the pose's R and t stay in SGPRs and
//     are broadcast by op_sel, so no register copies are spent on them;
//   - the rotation gradient is carried as the torque sum_i p_i x g_i (g = dL/dp): for R = RZ RY RX,
//     dR/dyaw = [e_z]x R, dR/dpitch = [RZ e_y]x R, dR/droll = [RZ RY e_x]x R, hence dL/dangle = axis . sum(p x g):
//     3 accumulators instead of the 9 of sum g q^T;  8 accumulators per pose in all: sum ||d||, count, sum g, sum p x g;
//   - the mask count is a scalar popcount of the compare result (SALU), not a per-lane add;
//   - atan2 is an octant reduction + degree-8 minimax polynomial (pcl_device.h), evaluated packed;
//   - RGBA8 panoramas: a 2x2 footprint is two 8-byte loads, the levels are interpolated as exact integers in fp32.
// Work decomposition: a block walks ONE contiguous chunk of the (Morton-ordered) cloud, so its lanes gather
// neighbouring texels, and evaluates G poses per loaded point pair; one wave + LDS reduction per block at the end and a
// plain store of the partials (no atomics: the second-stage reduce in pcl_gd.hip is deterministic).
#include <stdlib.h>

#include "pcl_device.h"

typedef float f2 __attribute__((ext_vector_type(2)));
#define F2(s) ((f2){(s), (s)})

struct PclLossArgs {
    const float* cloud;      // 6 planes of `stride` floats: x, y, z, -r, -g, -b
    int64_t n, stride;
    const void* pano;        // (H+2, W+2) texels, zero border: float4 (PCL_PANO_F32) or RGBA8 (PCL_PANO_U8)
    PclDims dims;
    const PclPoseRec* poses; // [B]
    int B;
    const uint8_t* visible;  // nullable [B][n]
    float* partials;         // [nchunks][B][8]
    int nchunks;             // multiple of 8
    int ngroups;             // B / G
    int flip;                    // 1: every XCD walks its chunks from the last to the first (see pcl_launch_loss)
    int seg_len;                 // chunks per contiguous run of one XCD (see the mapping at the top of pcl_loss_kernel)
    int steps_base, steps_rem;   // the cloud's ceil(n / PCL_STEP) steps are dealt out evenly: chunk c has steps_base + (c < steps_rem)
};

#define PCL_STEP (2 * PCL_BLOCK)   // points per block iteration: two per lane

// Experiments only (tools/block_trace.py builds a second library with -DPCL_BLOCK_TRACE): every block records when it
// FINISHED (100 MHz s_memrealtime) and where it ran (HW_ID, XCC_ID).  Only the end: a timestamp taken at the start has to
// live somewhere for the whole block, and this kernel sits exactly at its 4-waves-per-SIMD register budget — with start
// stamps the allocator falls back to 160 VGPRs / 3 waves and the timeline is no longer the product's.
#ifdef PCL_BLOCK_TRACE
__device__ unsigned long long* pcl_trace_buf = nullptr;
extern "C" int pcl_debug_set_block_trace(unsigned long long* buf)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(pcl_trace_buf), &buf, sizeof(buf));
}
#define PCL_TRACE_NOW() __builtin_amdgcn_s_memrealtime()
__global__ void pcl_debug_stamp_kernel(unsigned long long* slot) { *slot = PCL_TRACE_NOW(); }
extern "C" int pcl_debug_stamp(unsigned long long* slot, void* stream)
{
    hipLaunchKernelGGL(pcl_debug_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot);
    return (int)hipGetLastError();
}
#endif

__device__ __forceinline__ f2 pcl_fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// packed atan2 pieces (see pcl_atan2 in pcl_device.h for the scalar form and the error bound)
__device__ __forceinline__ f2 pcl_atan_poly2(f2 t)
{
    f2 s = t * t;
    f2 p = F2(2.4566929979e-03f);
    p = pcl_fma2(p, s, F2(-1.4401224869e-02f));
    p = pcl_fma2(p, s, F2(3.9780993102e-02f));
    p = pcl_fma2(p, s, F2(-7.2348362183e-02f));
    p = pcl_fma2(p, s, F2(1.0498935044e-01f));
    p = pcl_fma2(p, s, F2(-1.4161225936e-01f));
    p = pcl_fma2(p, s, F2(1.9985906258e-01f));
    p = pcl_fma2(p, s, F2(-3.3332596993e-01f));
    p = pcl_fma2(p, s, F2(9.9999988638e-01f));
    return p * t;
}

// first-octant angle atan(min/max) of two magnitudes (packed) and the "second is larger" flags
// (Measured and rejected, round 2: asin(min * rs) with rs = v_rsq(u^2 + v^2), the same rs squared serving the gradient's
// 1/(u^2 + v^2) — four transcendentals per point-pose instead of six (they issue at a fraction of the plain rate), same
// polynomial length: 3 393 -> 3 493 candidate-poses/s at cfg 2 (+3 %).  But asin amplifies the rounding of its argument by up
// to sqrt 2 where atan damps it by up to 2, and the argument carries the rounding of the sum of squares as well: the sample
// positions get about twice the noise, and G3's grad_t moved from 6.9e-7 to 1.7e-6 of the reference's fp64 autograd (the
// reference's own fp32 run: 3.5e-6), with or without a Newton step on rs^2.  Parity before 3 %.)
// (the min as one VOP3 with |.| modifiers, in asm: for operands that come out of the rotation's asm block the compiler
// cannot prove them canonical and would put a v_max x,x in front of every fminf)
__device__ __forceinline__ float pcl_min_abs(float a, float b)
{
    float r;
    asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 pcl_atan_ratio2(float u0, float v0, float u1, float v1)
{
    f2 mn = {pcl_min_abs(u0, v0), pcl_min_abs(u1, v1)};
    f2 rc = {__builtin_amdgcn_rcpf(fmaxf(fmaxf(fabsf(u0), fabsf(v0)), 1e-37f)), __builtin_amdgcn_rcpf(fmaxf(fmaxf(fabsf(u1), fabsf(v1)), 1e-37f))};
    return pcl_atan_poly2(mn * rc);
}

// phi = atan2(y, x) in (-pi, pi]: octant swap by select, the x < 0 reflection and the sign of y by sign transfers:
//   x < 0 ? pi - r : r  ==  pi/2 - copysign(pi/2 - r, x)        (r in [0, pi/2])
__device__ __forceinline__ f2 pcl_atan2_2(f2 y, f2 x)
{
    const float half_pi = 1.57079632679489661923f;
    float ax0 = fabsf(x.x), ax1 = fabsf(x.y), ay0 = fabsf(y.x), ay1 = fabsf(y.y);
    f2 r = pcl_atan_ratio2(x.x, y.x, x.y, y.y);
    f2 alt = F2(half_pi) - r;
    r = (f2){ay0 > ax0 ? alt.x : r.x, ay1 > ax1 ? alt.y : r.y};
    f2 w = F2(half_pi) - r;
    w = (f2){copysignf(w.x, x.x), copysignf(w.y, x.y)};
    r = F2(half_pi) - w;
    return (f2){copysignf(r.x, y.x), copysignf(r.y, y.y)};
}

// elevation e = atan2(z, rho) in [-pi/2, pi/2] for rho >= 0: no x < 0 case at all; theta = pi/2 - e
__device__ __forceinline__ f2 pcl_elevation2(f2 z, f2 rho)
{
    const float half_pi = 1.57079632679489661923f;
    float az0 = fabsf(z.x), az1 = fabsf(z.y);
    f2 r = pcl_atan_ratio2(rho.x, z.x, rho.y, z.y);
    f2 alt = F2(half_pi) - r;
    r = (f2){az0 > rho.x ? alt.x : r.x, az1 > rho.y ? alt.y : r.y};
    return (f2){copysignf(r.x, z.x), copysignf(r.y, z.y)};
}

// Raw 2x2 footprint of one point as it arrives from memory: RGBA8 -> two 8-byte texel pairs; float4 -> four taps.
template <int FMT> struct PclTaps;
template <> struct PclTaps<PCL_PANO_U8> { pcl_i2 top, bot; };
template <> struct PclTaps<PCL_PANO_F32> { int voff, row; };  // float4 texels are fetched where they are consumed
template <> struct PclTaps<PCL_PANO_F16> { pcl_i4 top, bot; };  // two half4 texels per row: (RG, B0) (RG, B0)

__device__ __forceinline__ void pcl_issue_taps(__amdgpu_buffer_rsrc_t tex, int x0, int y0, int Wp, PclTaps<PCL_PANO_U8>& o)
{
    int voff = (int)(__umul24((unsigned)y0, (unsigned)Wp) + (unsigned)x0) * 4;     // v_mad_u32_u24: full rate (v_mul_lo_u32 is quarter rate)
    o.top = pcl_texel_pair_u8(tex, voff, 0);
    o.bot = pcl_texel_pair_u8(tex, voff, Wp * 4);
}
__device__ __forceinline__ void pcl_issue_taps(__amdgpu_buffer_rsrc_t tex, int x0, int y0, int Wp, PclTaps<PCL_PANO_F16>& o)
{
    int voff = (int)(__umul24((unsigned)y0, (unsigned)Wp) + (unsigned)x0) * 8;
    o.top = __builtin_amdgcn_raw_buffer_load_b128(tex, voff, 0, 0);
    o.bot = __builtin_amdgcn_raw_buffer_load_b128(tex, voff, Wp * 8, 0);
}
__device__ __forceinline__ void pcl_issue_taps(__amdgpu_buffer_rsrc_t tex, int x0, int y0, int Wp, PclTaps<PCL_PANO_F32>& o)
{
    o.voff = (int)(__umul24((unsigned)y0, (unsigned)Wp) + (unsigned)x0) * 16;
    o.row = Wp * 16;
}
// the 12 tap components as floats (RGBA8: levels 0..255, one v_cvt_f32_ubyteN each)
__device__ __forceinline__ void pcl_unpack_taps(__amdgpu_buffer_rsrc_t, const PclTaps<PCL_PANO_U8>& r, float t[12])
{
    t[0] = pcl_ub0(r.top.x); t[1] = pcl_ub1(r.top.x); t[2] = pcl_ub2(r.top.x);
    t[3] = pcl_ub0(r.top.y); t[4] = pcl_ub1(r.top.y); t[5] = pcl_ub2(r.top.y);
    t[6] = pcl_ub0(r.bot.x); t[7] = pcl_ub1(r.bot.x); t[8] = pcl_ub2(r.bot.x);
    t[9] = pcl_ub0(r.bot.y); t[10] = pcl_ub1(r.bot.y); t[11] = pcl_ub2(r.bot.y);
}
__device__ __forceinline__ void pcl_unpack_taps(__amdgpu_buffer_rsrc_t tex, const PclTaps<PCL_PANO_F32>& r, float t[12])
{
    pcl_f4 t00 = pcl_texel(tex, r.voff, 0), t01 = pcl_texel(tex, r.voff + 16, 0);
    pcl_f4 t10 = pcl_texel(tex, r.voff, r.row), t11 = pcl_texel(tex, r.voff + 16, r.row);
    t[0] = t00.x; t[1] = t00.y; t[2] = t00.z; t[3] = t01.x; t[4] = t01.y; t[5] = t01.z;
    t[6] = t10.x; t[7] = t10.y; t[8] = t10.z; t[9] = t11.x; t[10] = t11.y; t[11] = t11.z;
}

// fp16 texels: the tap differences are exact in fp16 (integers up to 510) and computed two channels per instruction
// (v_pk_add_f16); the lerps read their fp16 operands directly (v_fma_mix_f32: fp32 fma with fp16 sources), so no tap is
// ever converted.  Every operand is the same real number as in the RGBA8 path and every fma is the same fp32 fma:
// results are bit-identical to it.
typedef _Float16 pcl_h2 __attribute__((ext_vector_type(2)));
// fma(a, lo/hi half of b, lo/hi half of c) in fp32 with fp16 sources b, c.  (Written as asm: from the C expression the
// vectoriser pairs the two points' fmas into v_pk_fma_f32 and pays a v_cvt_f32_f16 per operand for it.)
__device__ __forceinline__ float pcl_mix_lo(float a, pcl_h2 b, pcl_h2 c)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float pcl_mix_hi(float a, pcl_h2 b, pcl_h2 c)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
struct PclBilerp1 { float top[3], bot[3], dh[3]; };
template <bool GRAD>
__device__ __forceinline__ void pcl_bilerp_f16(const PclTaps<PCL_PANO_F16>& r, float fx, float fy, PclBilerp1& o)
{
    // (whole-vector bit cast + shuffles: extracting the dwords one by one and casting each to half2 makes this compiler
    // narrow the 16-byte load to ONE dword and alias all four pairs — ROCm 7.2 clang 22, wrong results)
    typedef _Float16 pcl_h8 __attribute__((ext_vector_type(8)));
    pcl_h8 tv = __builtin_bit_cast(pcl_h8, r.top), bv = __builtin_bit_cast(pcl_h8, r.bot);
    pcl_h2 t00a = __builtin_shufflevector(tv, tv, 0, 1), t00b = __builtin_shufflevector(tv, tv, 2, 3);
    pcl_h2 t01a = __builtin_shufflevector(tv, tv, 4, 5), t01b = __builtin_shufflevector(tv, tv, 6, 7);
    pcl_h2 t10a = __builtin_shufflevector(bv, bv, 0, 1), t10b = __builtin_shufflevector(bv, bv, 2, 3);
    pcl_h2 t11a = __builtin_shufflevector(bv, bv, 4, 5), t11b = __builtin_shufflevector(bv, bv, 6, 7);
    pcl_h2 dta = t01a - t00a, dtb = t01b - t00b, dba = t11a - t10a, dbb = t11b - t10b;
    o.top[0] = pcl_mix_lo(fx, dta, t00a); o.bot[0] = pcl_mix_lo(fx, dba, t10a);
    o.top[1] = pcl_mix_hi(fx, dta, t00a); o.bot[1] = pcl_mix_hi(fx, dba, t10a);
    o.top[2] = pcl_mix_lo(fx, dtb, t00b); o.bot[2] = pcl_mix_lo(fx, dbb, t10b);
    if (GRAD) {
        pcl_h2 dda = dba - dta, ddb = dbb - dtb;
        o.dh[0] = pcl_mix_lo(fy, dda, dta);
        o.dh[1] = pcl_mix_hi(fy, dda, dta);
        o.dh[2] = pcl_mix_lo(fy, ddb, dtb);
    }
}

// What the PROJECTION phase of one pose (two points, packed in .x/.y) hands to its SAMPLING phase.
template <int FMT>
struct PclProj {
    f2 px, py, pz;        // camera-frame point
    f2 rho2, rinv;        // px^2 + py^2 and 1/rho
    f2 fx, fy;            // bilinear fractions
    bool in_phi0, in_phi1, in_th0, in_th1;   // the +-0.99 clip is inactive (clamp backward passes the gradient only there)
    PclTaps<FMT> ta, tb;  // gathers in flight (point .x, point .y)
};

// Phase A: q = x - t, p = R q, cloud2idx, clip, pixel + fractions, issue the gathers.
template <int FMT>
__device__ __forceinline__ void pcl_project2(f2 x, f2 y, f2 z, const PclPoseRec* __restrict__ pose,
                                             __amdgpu_buffer_rsrc_t tex, const PclDims& dm, PclProj<FMT>& o)
{
    // q = x - t ; p = R q                                                   (omniloc.py:190-191, :332-338)
    // The pose is six 64-bit SGPR pairs (R0,R1)(R2,R3)(R4,R5)(R6,R7)(R8,t0)(t1,t2) and every scalar is broadcast to both
    // points by op_sel — written as ONE asm block: from the C expression F2(R[k]) the compiler copies each scalar into a
    // pair of its own (s_mov x2), 24 SGPRs per pose that it then spills to VGPR lanes and reads back with v_readlane
    // inside the loop.  Dependent packed-fp32 ops need one instruction between them (the compiler puts s_nop there):
    // the three rows are interleaved, so every result is used three slots later; same operations in the same order.
    // (The pairs rely on PclPoseRec's layout — t directly behind R[9], static_assert in pcl_device.h — and on the pose
    // pointer being wave-uniform: the "s" constraints below.)
    {
        const f2* __restrict__ P = reinterpret_cast<const f2*>(pose->R);
        const f2 p0 = P[0], p1 = P[1], p2 = P[2], p3 = P[3], p4 = P[4], p5 = P[5];
        f2 qx, qy, qz, px, py, pz;
        asm("v_pk_add_f32 %3, %6, %13 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"     // qx = x - t0   (hi of p4)
            "v_pk_add_f32 %4, %7, %14 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"                  // qy = y - t1   (lo of p5)
            "v_pk_add_f32 %5, %8, %14 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"     // qz = z - t2   (hi of p5)
            "v_pk_mul_f32 %0, %3, %9 op_sel_hi:[1,0]\n\t"                                            // px = qx R0
            "v_pk_mul_f32 %1, %3, %10 op_sel:[0,1] op_sel_hi:[1,1]\n\t"                              // py = qx R3
            "v_pk_mul_f32 %2, %3, %12 op_sel_hi:[1,0]\n\t"                                           // pz = qx R6
            "v_pk_fma_f32 %0, %9, %4, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                       // px += R1 qy
            "v_pk_fma_f32 %1, %11, %4, %1 op_sel_hi:[0,1,1]\n\t"                                     // py += R4 qy
            "v_pk_fma_f32 %2, %12, %4, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                      // pz += R7 qy
            "v_pk_fma_f32 %0, %10, %5, %0 op_sel_hi:[0,1,1]\n\t"                                     // px += R2 qz
            "v_pk_fma_f32 %1, %11, %5, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                      // py += R5 qz
            "v_pk_fma_f32 %2, %13, %5, %2 op_sel_hi:[0,1,1]\n\t"                                     // pz += R8 qz
            "s_nop 0"
            : "=&v"(px), "=&v"(py), "=&v"(pz), "=&v"(qx), "=&v"(qy), "=&v"(qz)
            : "v"(x), "v"(y), "v"(z), "s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(p5));
        o.px = px; o.py = py; o.pz = pz;
    }
    // cloud2idx (utils.py:44-59): gx = 1 - (atan2(py, a) + pi)/pi = -phi/pi ; gy = 2 theta/pi - 1
    f2 a = o.px + F2(1e-6f), b = o.pz + F2(1e-6f);
    o.rho2 = pcl_fma2(o.px, o.px, o.py * o.py);
    // 1/rho; rho = 0 gives rho2 * rinv = 0 and a zero gradient through rho (norm backward is 0 at 0)
    // (+1e-37: exact no-op for any normal rho2, keeps rsq finite at 0; one packed add instead of two v_max)
    f2 rg = o.rho2 + F2(1e-37f);
    o.rinv = (f2){__builtin_amdgcn_rsqf(rg.x), __builtin_amdgcn_rsqf(rg.y)};
    f2 rho = o.rho2 * o.rinv;
    f2 phi = pcl_atan2_2(o.py, a);
    f2 elev = pcl_elevation2(b, rho);                       // theta = atan2(rho, b) = pi/2 - elev
    // sample_from_img (utils.py:97-98): g = (-phi/pi, -2 elev/pi) clipped to +-0.99, unnormalised (align_corners=False),
    // +1 for the zero border.  The clip is applied to the angles (|phi| <= 0.99 pi, |elev| <= 0.495 pi: the same set up to
    // the last ulp of the threshold) so the pixel coordinate is one fma from the angle.
    const float lim_phi = 0.99f * 3.14159265358979323846f, lim_el = 0.495f * 3.14159265358979323846f;
    f2 phic = {__builtin_amdgcn_fmed3f(phi.x, -lim_phi, lim_phi), __builtin_amdgcn_fmed3f(phi.y, -lim_phi, lim_phi)};
    f2 elc = {__builtin_amdgcn_fmed3f(elev.x, -lim_el, lim_el), __builtin_amdgcn_fmed3f(elev.y, -lim_el, lim_el)};
    f2 ix = pcl_fma2(phic, F2(dm.k_ix), F2(dm.off_x));
    f2 iy = pcl_fma2(elc, F2(dm.k_iy), F2(dm.off_y));
    // ix, iy > 0 inside the border, so truncation == floor and fract == ix - floor(ix)
    pcl_issue_taps(tex, (int)ix.x, (int)iy.x, dm.Wp, o.ta);
    pcl_issue_taps(tex, (int)ix.y, (int)iy.y, dm.Wp, o.tb);
    o.fx = (f2){__builtin_amdgcn_fractf(ix.x), __builtin_amdgcn_fractf(ix.y)};
    o.fy = (f2){__builtin_amdgcn_fractf(iy.x), __builtin_amdgcn_fractf(iy.y)};
    // clamp backward passes the gradient on [-0.99, 0.99]: dL/dphi = -(W/2pi) <u, dc/dix>, dL/dtheta = (H/pi) <u, dc/diy>
    // (the constants carry the 1/255 of RGBA8 levels)
    // (a wave-uniform "no lane is clipped" fast path was tried: the extra basic block costs more in scheduling and
    // registers than the four selects it saves — 155 vs 141 us at cfg 2)
    o.in_phi0 = phi.x == phic.x; o.in_phi1 = phi.y == phic.y;
    o.in_th0 = elev.x == elc.x; o.in_th1 = elev.y == elc.y;
}

// Phase B: bilinear colour, mask, residual, gradient, accumulate.
// acc: 0 sum||d||, 1 unused here (count goes to `count`, wave-uniform), 2-4 sum g, 5-7 sum p x g — each an f2 whose
// halves are added at the end.
template <bool GRAD, int FMT>
__device__ __forceinline__ void pcl_sample2(const PclProj<FMT>& o, f2 ncr, f2 ncg, f2 ncb, bool valid0, bool valid1,
                                            unsigned long long vmask0, unsigned long long vmask1,
                                            __amdgpu_buffer_rsrc_t tex, const PclDims& dm, f2* acc, int& count)
{
    const f2 px = o.px, py = o.py, pz = o.pz, fx = o.fx, fy = o.fy;
    f2 c[3], dv[3], dtop[3], dbot[3], dhp[3];
    if constexpr (FMT == PCL_PANO_F16) {
        PclBilerp1 ba, bb;
        pcl_bilerp_f16<GRAD>(o.ta, fx.x, fy.x, ba);
        pcl_bilerp_f16<GRAD>(o.tb, fx.y, fy.y, bb);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            f2 top = {ba.top[k], bb.top[k]};
            dv[k] = (f2){ba.bot[k], bb.bot[k]} - top;
            c[k] = pcl_fma2(fy, dv[k], top);
            if (GRAD) dhp[k] = (f2){ba.dh[k], bb.dh[k]};
        }
    } else {
        float ta[12], tb[12];
        pcl_unpack_taps(tex, o.ta, ta);
        pcl_unpack_taps(tex, o.tb, tb);
        // bilinear: top/bottom rows, then vertical; both partial derivatives fall out of the same differences
#pragma unroll
        for (int k = 0; k < 3; k++) {
            f2 t00 = {ta[k], tb[k]}, t01 = {ta[3 + k], tb[3 + k]}, t10 = {ta[6 + k], tb[6 + k]}, t11 = {ta[9 + k], tb[9 + k]};
            dtop[k] = t01 - t00;
            dbot[k] = t11 - t10;
            f2 top = pcl_fma2(fx, dtop[k], t00), bot = pcl_fma2(fx, dbot[k], t10);
            dv[k] = bot - top;                                                    // dc/diy (in texel levels)
            c[k] = pcl_fma2(fy, dv[k], top);
        }
    }
    // mask: sampled colour not exactly (0,0,0)                               (omniloc.py:198, :347)
    float m0 = fmaxf(fmaxf(fabsf(c[0].x), fabsf(c[1].x)), fabsf(c[2].x));
    float m1 = fmaxf(fmaxf(fabsf(c[0].y), fabsf(c[1].y)), fabsf(c[2].y));
    bool keep0 = valid0 && m0 > 0.f, keep1 = valid1 && m1 > 0.f;
    // the count is wave-uniform bookkeeping: popcount of the compare's lane mask on the scalar unit (FCMP_OGT = 2)
    count += __builtin_popcountll(__builtin_amdgcn_fcmpf(m0, 0.f, 2) & vmask0) +
             __builtin_popcountll(__builtin_amdgcn_fcmpf(m1, 0.f, 2) & vmask1);
    // d = c - rgb; the packed cloud stores -rgb (pcl_cloud_pack), so this is one fma / add without a negation
    f2 d0, d1, d2;
    if (FMT != PCL_PANO_F32) {
        d0 = pcl_fma2(c[0], F2(dm.c_scale), ncr); d1 = pcl_fma2(c[1], F2(dm.c_scale), ncg); d2 = pcl_fma2(c[2], F2(dm.c_scale), ncb);
    } else {
        d0 = c[0] + ncr; d1 = c[1] + ncg; d2 = c[2] + ncb;
    }
    f2 n2 = pcl_fma2(d0, d0, pcl_fma2(d1, d1, d2 * d2));
    // 1/||d|| for kept points, 0 otherwise (also 0 * huge = 0 at ||d|| = 0: norm backward is 0 there)
    f2 ng = n2 + F2(1e-37f);
    f2 rn = {keep0 ? __builtin_amdgcn_rsqf(ng.x) : 0.f, keep1 ? __builtin_amdgcn_rsqf(ng.y) : 0.f};
    acc[0] = pcl_fma2(n2, rn, acc[0]);                                            // ||d|| = n2 * rsqrt(n2)
    if (GRAD) {
        // d||d||/dc = d / ||d||: the 1/||d|| is folded into the two angle factors instead of scaling d three times
        f2 dh0, dh1, dh2;                                                         // dc/dix
        if constexpr (FMT == PCL_PANO_F16) { dh0 = dhp[0]; dh1 = dhp[1]; dh2 = dhp[2]; }
        else {
            dh0 = pcl_fma2(fy, dbot[0] - dtop[0], dtop[0]);
            dh1 = pcl_fma2(fy, dbot[1] - dtop[1], dtop[1]);
            dh2 = pcl_fma2(fy, dbot[2] - dtop[2], dtop[2]);
        }
        f2 sx = pcl_fma2(d0, dh0, pcl_fma2(d1, dh1, d2 * dh2));
        f2 sy = pcl_fma2(d0, dv[0], pcl_fma2(d1, dv[1], d2 * dv[2]));
        // dL/dphi, dL/dtheta: the clip flag selects 1/||d|| or 0 (a select between two registers: selecting the CONSTANT
        // k_phi / k_theta under an SGPR lane mask needs a v_mov of the constant first, one scalar operand per VALU op)
        f2 rphi = {o.in_phi0 ? rn.x : 0.f, o.in_phi1 ? rn.y : 0.f}, rth = {o.in_th0 ? rn.x : 0.f, o.in_th1 ? rn.y : 0.f};
        f2 dphi = (sx * F2(dm.k_phi)) * rphi, dth = (sy * F2(dm.k_theta)) * rth;
        // phi = atan2(py, a): dphi/dpx = -py/s1, dphi/dpy = a/s1 ; theta = atan2(rho, b): dth/drho = b/s2, dth/dpz = -rho/s2
        f2 a = px + F2(1e-6f), b = pz + F2(1e-6f), rho = o.rho2 * o.rinv;
        f2 s1 = pcl_fma2(a, a, py * py), s2 = pcl_fma2(b, b, o.rho2);
        f2 ai = dphi * (f2){__builtin_amdgcn_rcpf(s1.x), __builtin_amdgcn_rcpf(s1.y)};
        f2 bi = dth * (f2){__builtin_amdgcn_rcpf(s2.x), __builtin_amdgcn_rcpf(s2.y)};
        f2 k = b * bi * o.rinv;                                                   // (dL/drho) / rho
        f2 g0 = pcl_fma2(k, px, -(py * ai));
        f2 g1 = pcl_fma2(k, py, a * ai);
        f2 g2 = -(rho * bi);
        acc[2] += g0; acc[3] += g1; acc[4] += g2;
        acc[5] = pcl_fma2(py, g2, pcl_fma2(-pz, g1, acc[5]));
        acc[6] = pcl_fma2(pz, g0, pcl_fma2(-px, g2, acc[6]));
        acc[7] = pcl_fma2(px, g1, pcl_fma2(-py, g0, acc[7]));
    }
}

// G poses per block, GRAD: with gradient, VIS: byte visibility mask, FMT: texel format.
// (Forcing more resident blocks per CU through __launch_bounds__ was tried: the register allocator spills, 2-4x slower.)
template <int G, bool GRAD, bool VIS, int FMT>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_loss_kernel(PclLossArgs a)
{
    // XCD-aware mapping: blocks b and b+8 share an XCD (round-robin dispatch) and each XCD has its own 4 MiB L2.  Within an
    // XCD the pose group varies fastest, so the blocks that are resident together read the same cloud chunk (it stays in
    // that XCD's L2 across the pose groups) and, the candidates being near each other, neighbouring texels.
    // Which chunks an XCD gets: chunk c belongs to XCD c mod 8 (seg_len = 1).  Round 1 gave every XCD ONE contiguous range of
    // chunks, i.e. its own eighth of the room — and the eighths differ in cost (how many points leave the panorama's valid
    // rows, how well their texels cache): tools/block_trace.py showed one to three XCDs done 10-30 % before the others in
    // every launch and idle through the tail, since the dispatcher deals blocks to XCDs round-robin whatever their speed.
    // Measured at cfg 2, one image per launch chain (bench.py single_image, same box): 1 / 2 / 4 / 8 / 32 runs per XCD
    // 2802 / 2832 / 2889 / 2933 / 2988 candidate-poses/s, and 3047 with every chunk its own run (frac 0.86 -> 0.93);
    // 8 images per launch +0.7 %, cfg 5 +2.4 %; the 1800-pose forward launch is unchanged.
    const int lq = (int)(blockIdx.x >> 3) / a.ngroups, group = (int)(blockIdx.x >> 3) - lq * a.ngroups;
    const int lc = a.flip ? (a.nchunks >> 3) - 1 - lq : lq;                                             // chunk within the XCD
    const int run = lc / a.seg_len;
    const int chunk = (run * 8 + (int)(blockIdx.x & 7)) * a.seg_len + (lc - run * a.seg_len);
    const int pose0 = group * G;

    __amdgpu_buffer_rsrc_t tex = pcl_tex_rsrc(a.pano, a.dims.H, a.dims.W, pcl_texel_bytes(FMT));
    // the cloud through a buffer resource too: 32-bit lane offsets + scalar plane offsets, no 64-bit address math
    __amdgpu_buffer_rsrc_t cld = __builtin_amdgcn_make_buffer_rsrc((void*)a.cloud, 0, (int)(a.stride * 6 * 4), 0x00020000);
    const int plane = (int)a.stride * 4;

    f2 acc[G][PCL_NACC];
    int count[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        count[g] = 0;
#pragma unroll
        for (int k = 0; k < PCL_NACC; k++) acc[g][k] = F2(0.f);
    }

    // balanced at step granularity: uniform chunks of ceil(n / nchunks) points rounded up to whole steps leave the last
    // chunks empty (1M points in 256 chunks: 4096-point chunks, the last 12 empty — one XCD finished 30 % early while the
    // other seven carried 5 % more than their share, tools/block_trace.py)
    int first = chunk * a.steps_base + min(chunk, a.steps_rem);
    int nsteps = a.steps_base + (chunk < a.steps_rem ? 1 : 0);
    // (Measured and rejected for the 1800-pose launch of trim_input_loss: walking the pose groups in super-groups of 19..152
    // groups, so that an XCD stays in one part of the panorama — 4.73 -> 4.69 ms with RGBA8 texels, 5.10 -> 5.04 ms with
    // fp16-level texels, which stay slower there despite 10 % fewer instructions.)
    // (Measured and rejected: tapered lengths — the chunks an XCD reaches first 2..8x longer than the ones it reaches last, so
    // that the tail of the launch consists of short blocks: 124.0 -> 122.5 us at cfg 2, within the run-to-run spread.)
    // (Measured and rejected: giving neighbouring work items different lengths — chunk pairs with their boundary moved by
    // 1/8..3/8 of a chunk — so that blocks resident together do not run their prologues / epilogues in phase: monotonically
    // slower, 124 -> 131 -> 137 us at cfg 2 for shifts of 1 / 2 steps: the longest block sets the tail.)
    const int begin = first * PCL_STEP;
    int end = begin + nsteps * PCL_STEP;
    if (end > (int)a.n) end = (int)a.n;
    const int last = (int)a.n - 1;

    // two points per lane: i0 = base + tid, i1 = base + 256 + tid.  The loop is unrolled by two with ping-pong register
    // sets: the loads of step k+1 are issued before step k is evaluated (their L2 latency hides under ~450 VALU
    // instructions) and no register copies are needed to rotate the buffers.
    auto load_step = [&](int base, float (&dst)[2][6]) {
        int j0 = min(base + (int)threadIdx.x, last), j1 = min(base + PCL_BLOCK + (int)threadIdx.x, last);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            dst[0][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, j0 * 4, k * plane, 0));
            dst[1][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, j1 * 4, k * plane, 0));
        }
    };
    auto eval_step = [&](int base, const float (&src)[2][6]) {
        const int i0 = base + threadIdx.x, i1 = i0 + PCL_BLOCK;
        const bool valid0 = i0 < end, valid1 = i1 < end;
        const unsigned long long vmask0 = __builtin_amdgcn_ballot_w64(valid0), vmask1 = __builtin_amdgcn_ballot_w64(valid1);
        f2 x = {src[0][0], src[1][0]}, y = {src[0][1], src[1][1]}, z = {src[0][2], src[1][2]};
        f2 ncr = {src[0][3], src[1][3]}, ncg = {src[0][4], src[1][4]}, ncb = {src[0][5], src[1][5]};
        // Projection and sampling phase per pose.  (Measured: projecting all poses of the group before sampling the
        // first — gathers of pose g in flight under the projection of pose g+1 — changes nothing at 4 waves/SIMD and
        // costs 30 VGPRs; the kernel is VALU-issue bound, not latency bound.)
#pragma unroll
        for (int g = 0; g < G; g++) {
            const PclPoseRec* __restrict__ pr = a.poses + (pose0 + g);
            bool ok0 = valid0, ok1 = valid1;
            unsigned long long m0 = vmask0, m1 = vmask1;
            if (VIS) {
                const uint8_t* vis = a.visible + (int64_t)(pose0 + g) * a.n;
                ok0 = ok0 && vis[min(i0, last)] != 0;
                ok1 = ok1 && vis[min(i1, last)] != 0;
                m0 = __builtin_amdgcn_ballot_w64(ok0); m1 = __builtin_amdgcn_ballot_w64(ok1);
            }
            // poses of several query images may share a launch: each pose record can name its own panorama
            // (same size and texel format); scalar work only
            __amdgpu_buffer_rsrc_t tg = tex;
            if (pr->pano_lo | pr->pano_hi) {
                const void* pp = (const void*)(((unsigned long long)pr->pano_hi << 32) | (unsigned long long)pr->pano_lo);
                tg = pcl_tex_rsrc(pp, a.dims.H, a.dims.W, pcl_texel_bytes(FMT));
            }
            PclProj<FMT> pj;
            pcl_project2<FMT>(x, y, z, pr, tg, a.dims, pj);
            pcl_sample2<GRAD, FMT>(pj, ncr, ncg, ncb, ok0, ok1, m0, m1, tg, a.dims, acc[g], count[g]);
        }
    };
    float bufA[2][6], bufB[2][6];
    load_step(begin, bufA);
    for (int base = begin; base < end; base += 2 * PCL_STEP) {
        load_step(base + PCL_STEP, bufB);              // clamped to the last point if past the end (evaluated as invalid)
        eval_step(base, bufA);
        if (base + PCL_STEP < end) {
            load_step(base + 2 * PCL_STEP, bufA);
            eval_step(base + PCL_STEP, bufB);
        }
    }

    // block reduction: fold the two packed halves, wave shuffle, 4 waves through LDS, plain store of the partials
    __shared__ float red[PCL_BLOCK / PCL_WAVE][G * PCL_NACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < PCL_NACC; k++) {
            if (k == 1) {
                if (lane == 0) red[wave][g * PCL_NACC + 1] = (float)count[g];   // wave-uniform popcount total
                continue;
            }
            if (!GRAD && k >= 2) continue;
            float s = pcl_wave_sum(acc[g][k].x + acc[g][k].y);
            if (lane == 0) red[wave][g * PCL_NACC + k] = s;
        }
    __syncthreads();
    if (threadIdx.x < G * PCL_NACC) {
        int g = threadIdx.x / PCL_NACC, k = threadIdx.x - g * PCL_NACC;
        float s = 0.f;
        if (GRAD || k < 2) s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        a.partials[((int64_t)chunk * a.B + pose0 + g) * PCL_NACC + k] = s;
    }
#ifdef PCL_BLOCK_TRACE
    if (pcl_trace_buf && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* t = pcl_trace_buf + (size_t)blockIdx.x * 4;
        t[0] = 0; t[1] = 0; t[2] = PCL_TRACE_NOW(); t[3] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------
// launch planning (shared with the GD loop)

// tuning knobs for tools/kbench.py (read once per process): PCL_G = poses per block (1/2/4), PCL_BLOCKS = target grid size
static int pcl_env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

struct PclPlan {
    int G, ngroups, nchunks;
    int seg_len;
    int steps_base, steps_rem;
};

static PclPlan pcl_plan(int64_t n, int B)
{
    static const int g_env = pcl_env_int("PCL_G", 0), blocks_env = pcl_env_int("PCL_BLOCKS", 4096);
    PclPlan p;
    // poses per block: 2 measured best at cfg2 (4: 146 VGPRs -> 3 waves/SIMD; 1: point loads not amortised)
    p.G = (B % 2 == 0) ? 2 : 1;
    if (g_env > 0 && B % g_env == 0) p.G = g_env;
    p.ngroups = B / p.G;
    // aim at ~4096 blocks (256 CUs x a few resident blocks x several rounds), at least one step per chunk, and at least 64
    // chunks however many poses there are: with 8 long chunks the 1800-pose launch of trim_input_loss has every block
    // sweep an eighth of the room on its own, nothing it gathers is reused by a neighbour (5.08 -> 4.83 ms at 64 chunks)
    int64_t want = blocks_env / p.ngroups;
    if (want < 64) want = 64;
    int64_t steps = (n + PCL_STEP - 1) / PCL_STEP;
    // small clouds under many poses (cfg 1 with 64 images per launch: 196 steps, 32 groups): blocks of one or two steps are
    // mostly prologue and epilogue — go for three steps per block as long as one round of resident blocks remains
    // (18.8k -> 20.2k candidate-poses/s at cfg 1 batched; 167k points x 32 candidates: 4.0 -> 3.6 ms per image)
    if (steps < 3 * want) {
        int64_t alt = steps / 3;
        if (alt < 1024 / p.ngroups) alt = 1024 / p.ngroups;
        if (alt < 32) alt = 32;
        if (alt < want) want = alt;
    }
    if (want > steps) want = steps;
    want = ((want + 7) / 8) * 8;                   // (a cloud of fewer steps than chunks leaves the surplus chunks empty)
    p.nchunks = (int)want;
    p.steps_base = (int)(steps / want);
    p.steps_rem = (int)(steps % want);
    // contiguous chunk runs per XCD (PCL_XCD_RUNS, experiments).  Default: every chunk its own run (chunk c on XCD c mod 8)
    // when the launch takes several rounds of resident blocks — the XCDs then finish together; ONE run per XCD when all blocks
    // are resident at once (the shipped 167k-point / 6-candidate shape: 984 one-step blocks): nothing to balance there, and
    // with its chunks side by side an XCD touches an eighth of the panorama instead of all of it (loss kernel 9.5 vs 10.8 us).
    // Otherwise the largest divisor of the XCD's chunk count not above the target.
    static const int runs_env = pcl_env_int("PCL_XCD_RUNS", 0);
    int cpx = p.nchunks / 8, runs = runs_env < 1 || runs_env > cpx ? cpx : runs_env;
    if (runs_env < 1 && (int64_t)p.nchunks * p.ngroups <= 1024) runs = 1;       // 256 CUs x 4 resident 256-thread blocks
    while (cpx % runs) runs--;
    p.seg_len = cpx / runs;
    return p;
}

size_t pcl_partials_bytes(int64_t n, int B)
{
    PclPlan p = pcl_plan(n, B);
    return (size_t)p.nchunks * (size_t)B * PCL_NACC * sizeof(float);
}

int pcl_plan_nchunks(int64_t n, int B) { return pcl_plan(n, B).nchunks; }

template <int G, int FMT>
static void pcl_launch_g(const PclLossArgs& a, int nblk, bool grad, bool vis, hipStream_t s)
{
    if (grad) {
        if (vis) hipLaunchKernelGGL((pcl_loss_kernel<G, true, true, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((pcl_loss_kernel<G, true, false, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
    } else {
        if (vis) hipLaunchKernelGGL((pcl_loss_kernel<G, false, true, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((pcl_loss_kernel<G, false, false, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
    }
}

template <int FMT>
static void pcl_launch_f(const PclLossArgs& a, int G, int nblk, bool grad, bool vis, hipStream_t s)
{
    if (G == 4) pcl_launch_g<4, FMT>(a, nblk, grad, vis, s);
    else if (G == 2) pcl_launch_g<2, FMT>(a, nblk, grad, vis, s);
    else pcl_launch_g<1, FMT>(a, nblk, grad, vis, s);
}

// Enqueue one fused loss(+grad) pass over the cloud for B poses; partials must hold pcl_partials_bytes(n, B).
int pcl_launch_loss(const float* cloud, int64_t n, const void* pano, int pano_format, int H, int W, const PclPoseRec* poses,
                    int B, bool grad, const uint8_t* visible, float* partials, hipStream_t s, int flip)
{
    if (pano_format != PCL_PANO_F32 && pano_format != PCL_PANO_U8 && pano_format != PCL_PANO_F16) return PCL_EINVAL;
    // 32-bit buffer addressing: 6 planes x 4 B x n must stay below 4 GiB, the padded panorama below 2 GiB
    if (n > PCL_MAX_POINTS || (int64_t)(H + 2) * (W + 2) * pcl_texel_bytes(pano_format) >= ((int64_t)1 << 31)) return PCL_EINVAL;
    PclPlan p = pcl_plan(n, B);
    PclLossArgs a;
    a.cloud = cloud; a.n = n; a.stride = pcl_cloud_stride(n);
    a.pano = pano; a.dims = pcl_make_dims(H, W, pano_format);
    a.poses = poses; a.B = B; a.visible = visible; a.partials = partials;
    a.nchunks = p.nchunks; a.ngroups = p.ngroups; a.seg_len = p.seg_len; a.flip = flip; a.steps_base = p.steps_base; a.steps_rem = p.steps_rem;
    int nblk = p.nchunks * p.ngroups;
    bool vis = visible != nullptr;
    if (pano_format == PCL_PANO_U8) pcl_launch_f<PCL_PANO_U8>(a, p.G, nblk, grad, vis, s);
    else if (pano_format == PCL_PANO_F16) pcl_launch_f<PCL_PANO_F16>(a, p.G, nblk, grad, vis, s);
    else pcl_launch_f<PCL_PANO_F32>(a, p.G, nblk, grad, vis, s);
    PCL_LAUNCH_CHECK();
    return 0;
}
