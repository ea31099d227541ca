// pcl_sample_device.h — device code shared by the kernels that project points through a pose and sample the panorama: the fused
// loss(+gradient) kernel (pcl_loss.hip) and the yaw-shared forward-only kernel of trim_input_loss (pcl_trim.hip).
// Two points per lane, packed fp32 (see the header of pcl_loss.hip for why).
#pragma once
#include "pcl_device.h"

typedef float f2 __attribute__((ext_vector_type(2)));
#define F2(s) ((f2){(s), (s)})

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
    f2 rs2;               // 1 / sqrt(rho^2 + (pz + eps)^2): the elevation's half-angle form and the gradient's 1 / s2 share it
    f2 fx, fy;            // bilinear fractions
    bool in_phi0, in_phi1, in_th0, in_th1;   // the +-0.99 clip is inactive (clamp backward passes the gradient only there)
    PclTaps<FMT> ta, tb;  // gathers in flight (point .x, point .y)
    unsigned zq0, zq1;    // DEPTH only: the z-buffer cells of the two points (squared-depth bit patterns), in flight
};

// cell indices of two points from their angles, make_pano's formula (utils.py:158-165) on the Hd x Wd grid:
//   col = trunc((gx + 1) / 2 (Wd - 1)),  (gx + 1) / 2 = 1/2 - phi / (2 pi)            (gx = -phi / pi)
//   row = trunc((gy + 1) / 2 (Hd - 1)),  (gy + 1) / 2 = 1/2 - 4 hel / (2 pi)          (gy = 2 theta / pi - 1, theta = pi/2 - 2 hel)
// 1/2, 4 and 1/(2 pi) are inline constants of the ISA: the only scalar registers this needs are Wd - 1, Hd - 1 and Wd.  No clamp:
// phi in [-pi, pi] and hel in [-pi/4, pi/4] keep col in [0, Wd - 1] and row in [0, Hd - 1] (rounding at the ends truncates
// inwards, a NaN angle converts to 0), and the z-buffer is read through a bounds-checked buffer resource.
__device__ __forceinline__ void pcl_depth_cells2(f2 phi, f2 hel, const PclDepthGrid& g, int& row0, int& col0, int& row1, int& col1)
{
    const float inv_2pi = 0.15915494309189532f;
    f2 colf = pcl_fma2(-phi, F2(inv_2pi), F2(0.5f)) * F2(g.wm1);
    f2 rowf = pcl_fma2(-(hel * F2(4.0f)), F2(inv_2pi), F2(0.5f)) * F2(g.hm1);
    row0 = (int)rowf.x; col0 = (int)colf.x; row1 = (int)rowf.y; col1 = (int)colf.y;
}

// q = x - t ; p = R q for two points (packed)                               (omniloc.py:190-191, :332-338)
// The pose is six 64-bit SGPR pairs (R0,R1)(R2,R3)(R4,R5)(R6,R7)(R8,t0)(t1,t2) and every scalar is broadcast to both
// points by op_sel — written as ONE asm block: from the C expression F2(R[k]) the compiler copies each scalar into a
// pair of its own (s_mov x2), 24 SGPRs per pose that it then spills to VGPR lanes and reads back with v_readlane
// inside the loop.  Dependent packed-fp32 ops need one instruction between them (the compiler puts s_nop there):
// the three rows are interleaved, so every result is used three slots later; same operations in the same order.
// (The pairs rely on PclPoseRec's layout — t directly behind R[9], static_assert in pcl_device.h — and on the pose
// pointer being wave-uniform: the "s" constraints below.)
struct PclPose6 { f2 p0, p1, p2, p3, p4, p5; };       // the six SGPR pairs of a pose: (R0,R1)(R2,R3)(R4,R5)(R6,R7)(R8,t0)(t1,t2)
__device__ __forceinline__ PclPose6 pcl_pose6(const PclPoseRec* __restrict__ pose)
{
    const f2* __restrict__ P = reinterpret_cast<const f2*>(pose->R);
    return PclPose6{P[0], P[1], P[2], P[3], P[4], P[5]};
}
__device__ __forceinline__ void pcl_rotate2(f2 x, f2 y, f2 z, const PclPose6& P, f2& opx, f2& opy, f2& opz)
{
    const f2 p0 = P.p0, p1 = P.p1, p2 = P.p2, p3 = P.p3, p4 = P.p4, p5 = P.p5;
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
    opx = px; opy = py; opz = pz;
}
#ifndef PCL_NO_ROTATE_PAIR
// Round 4: the rotations of BOTH poses of a block in one asm block, pose B's op in the slot behind pose A's dependent op — no
// hazard slot left in the 24 packed ops (same operations in the same order per pose: results unchanged bit for bit).  Costs the
// second pose's p (6 VGPRs) for the whole first pose's projection + sampling: 116 -> 123 VGPRs, still four waves per SIMD; the
// loop body's s_nop count falls from 80 to 70 (the other 70 sit between dependent packed ops of the two Horner chains and the
// bilinear lerps, where the compiler's scheduler finds nothing to move).  Measured A/B on one box, alternating twice: cfg 2
// 3 497 -> 3 510 candidate-poses/s (160 poses per launch: 455.1 -> 453.1 us), one image per chain 3 163 -> 3 182: +0.5 %.
// With four waves per SIMD another wave issues while one sits in a hazard slot, so the slots were mostly hidden already
// (VALU busy 0.91): the bounded attempt VERDICT r03 asked for — kept because it is free, the kernel is left alone after it.
// (-DPCL_NO_ROTATE_PAIR restores the per-pose form.)
__device__ __forceinline__ void pcl_rotate2x2(f2 x, f2 y, f2 z, const PclPose6& A, const PclPose6& B, f2& apx, f2& apy, f2& apz, f2& bpx,
                                              f2& bpy, f2& bpz)
{
    f2 aqx, aqy, aqz, bqx, bqy, bqz, ax, ay, az, bx, by, bz;
    asm("v_pk_add_f32 %6, %12, %19 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %9, %12, %25 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %7, %13, %20 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %10, %13, %26 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %8, %14, %20 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %11, %14, %26 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %0, %6, %15 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %3, %9, %21 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %1, %6, %16 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %4, %9, %22 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %2, %6, %18 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %5, %9, %24 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %15, %7, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %3, %21, %10, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %1, %17, %7, %1 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %4, %23, %10, %4 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %2, %18, %7, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %5, %24, %10, %5 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %0, %16, %8, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %3, %22, %11, %3 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %17, %8, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %4, %23, %11, %4 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %2, %19, %8, %2 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %5, %25, %11, %5 op_sel_hi:[0,1,1]\n\t"
        "s_nop 0"
        : "=&v"(ax), "=&v"(ay), "=&v"(az), "=&v"(bx), "=&v"(by), "=&v"(bz), "=&v"(aqx), "=&v"(aqy), "=&v"(aqz), "=&v"(bqx), "=&v"(bqy), "=&v"(bqz)
        : "v"(x), "v"(y), "v"(z), "s"(A.p0), "s"(A.p1), "s"(A.p2), "s"(A.p3), "s"(A.p4), "s"(A.p5), "s"(B.p0), "s"(B.p1), "s"(B.p2), "s"(B.p3),
          "s"(B.p4), "s"(B.p5));
    apx = ax; apy = ay; apz = az; bpx = bx; bpy = by; bpz = bz;
}
#endif

__device__ __forceinline__ void pcl_rotate2(f2 x, f2 y, f2 z, const PclPoseRec* __restrict__ pose, f2& opx, f2& opy, f2& opz)
{
    pcl_rotate2(x, y, z, pcl_pose6(pose), opx, opy, opz);
}

// Phase A: q = x - t, p = R q, cloud2idx, clip, pixel + fractions, issue the gathers.
template <int FMT, bool DEPTH = false>
__device__ __forceinline__ void pcl_project2_rotated(__amdgpu_buffer_rsrc_t tex, const PclDims& dm, PclProj<FMT>& o,
                                                     __amdgpu_buffer_rsrc_t zb = __amdgpu_buffer_rsrc_t(), int zoff = 0,
                                                     const PclDepthGrid* dg = nullptr);

template <int FMT, bool DEPTH = false>
__device__ __forceinline__ void pcl_project2(f2 x, f2 y, f2 z, const PclPose6& pose,
                                             __amdgpu_buffer_rsrc_t tex, const PclDims& dm, PclProj<FMT>& o,
                                             __amdgpu_buffer_rsrc_t zb = __amdgpu_buffer_rsrc_t(), int zoff = 0, const PclDepthGrid* dg = nullptr)
{
    pcl_rotate2(x, y, z, pose, o.px, o.py, o.pz);
    pcl_project2_rotated<FMT, DEPTH>(tex, dm, o, zb, zoff, dg);
}

// cloud2idx's two angles for two camera-frame points (utils.py:44-59): phi = atan2(py, px + eps) and HALF the elevation
// e = atan2(pz + eps, rho), plus the by-products the gradient reuses.  Shared by the loss / trim kernels and the depth passes
// (csrc/pcl_depth.hip), which must land a point in the same z-buffer cell as the loss kernel's lookup.
__device__ __forceinline__ void pcl_angles2(f2 px, f2 py, f2 pz, f2& rho2, f2& rinv, f2& rs2, f2& phi, f2& half_el)
{
    // cloud2idx (utils.py:44-59): gx = 1 - (atan2(py, a) + pi)/pi = -phi/pi ; gy = 2 theta/pi - 1
    f2 a = px + F2(1e-6f), b = pz + F2(1e-6f);
    rho2 = pcl_fma2(px, px, py * py);
    // 1/rho; rho = 0 gives rho2 * rinv = 0 and a zero gradient through rho (norm backward is 0 at 0)
    // (+1e-37: exact no-op for any normal rho2, keeps rsq finite at 0; one packed add instead of two v_max)
    f2 rg = rho2 + F2(1e-37f);
    rinv = (f2){__builtin_amdgcn_rsqf(rg.x), __builtin_amdgcn_rsqf(rg.y)};
    f2 rho = rho2 * rinv;
    phi = pcl_atan2_2(py, a);
    // Elevation e = atan2(b, rho) (theta = pi/2 - e) by the half-angle form  e = 2 atan(b / (r + rho)),  r = sqrt(rho^2 + b^2):
    // rho >= 0, so |b| <= r + rho and the argument is in [-1, 1] for every point — no octant swap, no reflection, no sign
    // transfer (the octant form spent a min, a max3, a compare, a select and a v_bfi per point here), and its rsq is the one
    // the gradient needs anyway (1 / s2 = rs2^2 instead of a v_rcp): 4.5 VALU instructions and one transcendental less per
    // point-pose.  The polynomial's 8.7e-8 error is doubled: 1.7e-7 rad, below the fp32 ulp of the row coordinate.
    // (+1e-37: keeps rsq and the quotient finite for a point AT the camera centre shifted by -eps; exact no-op otherwise.)
    f2 s2 = pcl_fma2(b, b, rho2) + F2(1e-37f);
    rs2 = (f2){__builtin_amdgcn_rsqf(s2.x), __builtin_amdgcn_rsqf(s2.y)};
    f2 den = pcl_fma2(s2, rs2, rho);
    half_el = pcl_atan_poly2(b * (f2){__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)});
}

// (o.px, o.py, o.pz already hold p = R (x - t))
// DEPTH: also issue the look-ups of the two points' z-buffer cells (zb: this pose's z-buffer through a buffer resource)
template <int FMT, bool DEPTH>
__device__ __forceinline__ void pcl_project2_rotated(__amdgpu_buffer_rsrc_t tex, const PclDims& dm, PclProj<FMT>& o, __amdgpu_buffer_rsrc_t zb,
                                                     int zoff, const PclDepthGrid* dg)
{
    f2 phi, half_el;
    pcl_angles2(o.px, o.py, o.pz, o.rho2, o.rinv, o.rs2, phi, half_el);
    if constexpr (DEPTH) {
        int r0, c0, r1, c1;
        pcl_depth_cells2(phi, half_el, *dg, r0, c0, r1, c1);
        o.zq0 = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(zb, (int)(__umul24((unsigned)r0, (unsigned)dg->Wd) + (unsigned)c0) * 4, zoff, 0);
        o.zq1 = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(zb, (int)(__umul24((unsigned)r1, (unsigned)dg->Wd) + (unsigned)c1) * 4, zoff, 0);
    }
    // sample_from_img (utils.py:97-98): g = (-phi/pi, -2 elev/pi) clipped to +-0.99, unnormalised (align_corners=False),
    // +1 for the zero border.  The clip is applied to the angles (|phi| <= 0.99 pi, |elev| <= 0.495 pi: the same set up to
    // the last ulp of the threshold) so the pixel coordinate is one fma from the angle.
    const float lim_phi = 0.99f * 3.14159265358979323846f, lim_hel = 0.2475f * 3.14159265358979323846f;
    f2 phic = {__builtin_amdgcn_fmed3f(phi.x, -lim_phi, lim_phi), __builtin_amdgcn_fmed3f(phi.y, -lim_phi, lim_phi)};
    f2 helc = {__builtin_amdgcn_fmed3f(half_el.x, -lim_hel, lim_hel), __builtin_amdgcn_fmed3f(half_el.y, -lim_hel, lim_hel)};
    f2 ix = pcl_fma2(phic, F2(dm.k_ix), F2(dm.off_x));
    f2 iy = pcl_fma2(helc, F2(2.f * dm.k_iy), F2(dm.off_y));
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
    o.in_th0 = half_el.x == helc.x; o.in_th1 = half_el.y == helc.y;
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
        f2 s1 = pcl_fma2(a, a, py * py);
        f2 ai = dphi * (f2){__builtin_amdgcn_rcpf(s1.x), __builtin_amdgcn_rcpf(s1.y)};
        f2 bi = dth * (o.rs2 * o.rs2);                                            // dth / s2
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

