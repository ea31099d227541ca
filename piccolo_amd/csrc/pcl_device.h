// pcl_device.h — device-side building blocks shared by the gfx950 kernels of the PICCOLO sampling-loss path.
// Wavefront = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/piccolo_hip.h"

// Experiment knobs.  The SHIPPED library reads no environment variable: every A/B switch of the measured-and-rejected log
// (profiles/EXPERIMENTS.md) is PCL_KNOB(NAME, default), which is the default — a compile-time constant — unless the library is built
// with -DPCL_EXPERIMENTS (piccolo_amd/build.py build_experiments(): lib/libpiccolo_hip_exp.so, what tools/ load through PCL_SO); only
// that build reads PCL_<NAME> from the environment.  `strings libpiccolo_hip.so | grep -c '^PCL_'` is 0 (tests/test_abi.py).
#ifdef PCL_EXPERIMENTS
#include <stdlib.h>
static inline int pcl_knob_read(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
#define PCL_KNOB(NAME, dflt) pcl_knob_read("PCL_" #NAME, (dflt))
#else
#define PCL_KNOB(NAME, dflt) (dflt)
#endif

#define PCL_WAVE 64
#define PCL_BLOCK 256
#define PCL_NACC 8  // per-pose accumulators: sum ||d||, count, sum g (3), sum p x g (3)

typedef float pcl_f4 __attribute__((ext_vector_type(4)));
typedef int pcl_i4 __attribute__((ext_vector_type(4)));
typedef int pcl_i2 __attribute__((ext_vector_type(2)));

// One candidate pose as the loss kernel consumes it (scalar loads): p = R (x - t).
struct PclPoseRec {
    float R[9];
    float t[3];
    uint32_t pano_lo, pano_hi;   // this pose's panorama (device address) when poses of several query images share a
                                 // launch; 0 = the launch's default panorama
    float pad[2];
};
static_assert(sizeof(PclPoseRec) == 64, "pose record is one 64-byte scalar-load line");
// pcl_project2 (pcl_loss.hip) reads R and t as six consecutive 64-bit SGPR pairs (R0,R1) ... (R8,t0) (t1,t2)
static_assert(offsetof(PclPoseRec, R) == 0 && offsetof(PclPoseRec, t) == 36, "pose record: t directly after R[9]");

// Per-candidate optimiser state of the on-device GD loop (pcl_gd.hip).
struct PclGdPose {
    double lr;    // this candidate's Adam learning rate (python float in the reference)
    double best;  // ReduceLROnPlateau.best
    float leaf[6];  // Adam's parameters: t(3), yaw, pitch, roll
    float fwd[6];   // parameters the next forward sees (== leaf in sequential mode; pre-clamp copy in batch mode)
    float m[6];     // exp_avg
    float v[6];     // exp_avg_sq
    float last_loss;
    int32_t num_bad;
    int32_t step;
    int32_t pad;
    double beta1_pow, beta2_pow;   // beta^step, kept as running products (Adam bias corrections)
    float sc[4];                   // sin/cos of the forward pose's yaw and pitch (chain rule of the NEXT epilogue)
};
static_assert(sizeof(PclGdPose) == 160, "GD state record");

struct PclDims {
    int H, W;    // panorama size
    int Wp;      // padded row length in texels (W + 2)
    float half_w, half_h;    // W/2, H/2            (grid_sample unnormalise)
    float off_x, off_y;      // (W-1)/2 + 1, (H-1)/2 + 1 : pixel coordinate in the zero-bordered texture
    float k_phi, k_theta;    // -W/(2 pi), H/pi     (d ix / d phi, d iy / d theta); times 1/255 for RGBA8 texels
    float c_scale;           // texel level -> colour: 1 (float texels) or 1/255 (RGBA8)
    float k_ix, k_iy;        // d ix / d phi = -W/(2 pi), d iy / d elevation = -H/pi   (pixel from the angles directly)
};

__host__ __device__ inline PclDims pcl_make_dims(int H, int W, int pano_format = PCL_PANO_F32)
{
    PclDims d;
    d.H = H; d.W = W; d.Wp = W + 2;
    d.half_w = 0.5f * (float)W; d.half_h = 0.5f * (float)H;
    d.off_x = 0.5f * (float)(W - 1) + 1.0f; d.off_y = 0.5f * (float)(H - 1) + 1.0f;
    d.k_phi = (float)(-(double)W / (2.0 * 3.14159265358979323846));
    d.k_theta = (float)((double)H / 3.14159265358979323846);
    d.c_scale = 1.0f;
    d.k_ix = (float)(-(double)W / (2.0 * 3.14159265358979323846));
    d.k_iy = (float)(-(double)H / 3.14159265358979323846);
    if (pano_format == PCL_PANO_U8 || pano_format == PCL_PANO_F16) {
        d.k_phi = (float)(-(double)W / (2.0 * 3.14159265358979323846) / 255.0);
        d.k_theta = (float)((double)H / 3.14159265358979323846 / 255.0);
        d.c_scale = (float)(1.0 / 255.0);
    }
    return d;
}

// The scatter-min depth mask's z-buffer as the projecting kernels see it (build-defined, csrc/pcl_depth.hip): an Hd x Wd grid of
// make_pano's pixels (utils.py:158-165) per pose, cell = bit pattern of the smallest squared depth that fell into it.
//   col = trunc((gx + 1) / 2 (Wd - 1)),   row = trunc((gy + 1) / 2 (Hd - 1))              (pcl_depth_cells2, pcl_sample_device.h)
// straight from the UNCLIPPED angles of pcl_angles2 — the z pass and the lookup in the loss kernel run the same instructions on
// the same inputs, so a point looks its own cell up.
struct PclDepthGrid {
    int Hd, Wd, last;           // last = Hd * Wd - 1
    float wm1, hm1;             // Wd - 1, Hd - 1
    float tol2;                 // (1 + tau)^2.  The z pass stores min(d2 * tol2) = tol2 * min d2 (rounding is monotone): visible iff d2 <= cell
};
__host__ __device__ inline PclDepthGrid pcl_make_depth_grid(int Hd, int Wd, float tau)
{
    PclDepthGrid g;
    g.Hd = Hd; g.Wd = Wd; g.last = Hd * Wd - 1;
    g.wm1 = (float)(Wd - 1); g.hm1 = (float)(Hd - 1);
    g.tol2 = (1.0f + tau) * (1.0f + tau);
    return g;
}
// what a loss launch needs to look the mask up (pcl_launch_loss)
struct PclDepthLook {
    const uint32_t* zbuf;       // [B][Hd * Wd]
    PclDepthGrid grid;
    uint32_t* zclear;           // nullable: another set of z-buffers (zclear_vec4 16-byte words) that the loss launch resets to +inf
    int64_t zclear_vec4;
};
// R = RZ(yaw) RY(pitch) RX(roll) (reference utils.py:425-453). sin/cos are evaluated in double and rounded once,
// which lands within an ulp of the fp32 values ATen computes on the host.
__device__ inline void pcl_rot_from_ypr(float yaw, float pitch, float roll, float R[9])
{
    double sy, cy, sp, cp, sr, cr;
    sincos((double)yaw, &sy, &cy);
    sincos((double)pitch, &sp, &cp);
    sincos((double)roll, &sr, &cr);
    sy = (double)(float)sy; cy = (double)(float)cy; sp = (double)(float)sp;
    cp = (double)(float)cp; sr = (double)(float)sr; cr = (double)(float)cr;
    // RZ RY = [[cy cp, -sy, cy sp], [sy cp, cy, sy sp], [-sp, 0, cp]]
    R[0] = (float)(cy * cp); R[1] = (float)(cy * sp * sr - sy * cr); R[2] = (float)(cy * sp * cr + sy * sr);
    R[3] = (float)(sy * cp); R[4] = (float)(sy * sp * sr + cy * cr); R[5] = (float)(sy * sp * cr - cy * sr);
    R[6] = (float)(-sp);     R[7] = (float)(cp * sr);                R[8] = (float)(cp * cr);
}

__device__ inline void pcl_write_pose_rec(PclPoseRec* rec, const float p[6])
{
    float R[9];
    pcl_rot_from_ypr(p[3], p[4], p[5], R);
    for (int k = 0; k < 9; k++) rec->R[k] = R[k];
    rec->t[0] = p[0]; rec->t[1] = p[1]; rec->t[2] = p[2];
    rec->pano_lo = 0u; rec->pano_hi = 0u;
    rec->pad[0] = rec->pad[1] = 0.f;
}

// Same R for the GD epilogue, which runs once per iteration on ONE lane per candidate: fp32 sincosf (<= 2 ulp, the
// precision the reference's own fp32 torch.cos/sin + mm deliver) instead of three double sincos; also returns the
// sin/cos of yaw and pitch that the next chain rule needs.
__device__ inline void pcl_write_pose_rec_fast(PclPoseRec* rec, const float p[6], float sc[4])
{
    float sy, cy, sp, cp, sr, cr;
    sincosf(p[3], &sy, &cy);
    sincosf(p[4], &sp, &cp);
    sincosf(p[5], &sr, &cr);
    double dsy = sy, dcy = cy, dsp = sp, dcp = cp, dsr = sr, dcr = cr;
    rec->R[0] = (float)(dcy * dcp); rec->R[1] = (float)(dcy * dsp * dsr - dsy * dcr); rec->R[2] = (float)(dcy * dsp * dcr + dsy * dsr);
    rec->R[3] = (float)(dsy * dcp); rec->R[4] = (float)(dsy * dsp * dsr + dcy * dcr); rec->R[5] = (float)(dsy * dsp * dcr - dcy * dsr);
    rec->R[6] = (float)(-dsp);      rec->R[7] = (float)(dcp * dsr);                   rec->R[8] = (float)(dcp * dcr);
    rec->t[0] = p[0]; rec->t[1] = p[1]; rec->t[2] = p[2];      // (pano_lo / pano_hi are left as they are)
    sc[0] = sy; sc[1] = cy; sc[2] = sp; sc[3] = cp;
}

// Equirectangular projection of a camera-frame point (reference utils.py:44-59):
//   theta = atan2(|p_xy|, p_z + 1e-6), phi = atan2(p_y, p_x + 1e-6) + pi, g = (1 - phi/pi, 2 theta/pi - 1)
// written with the same operation order as the reference so the stand-alone op matches it to an ulp or two.
// (fp contraction off: every call site — the stand-alone op, make_pano, the histogram stage's splat, bin and fix-up kernels — must
//  produce the same bits for the same point; left to the optimiser, px * px + py * py became an fma at some sites and not at others,
//  and one point in 64M landed in a neighbouring pixel: found when round 5's fix-up path was compared with the splat path)
__device__ inline void pcl_cloud2idx_point(float px, float py, float pz, float& gx, float& gy)
{
#pragma clang fp contract(off)
    const float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f;
    float rho = sqrtf(px * px + py * py);
    float theta = atan2f(rho, pz + 1e-6f);
    float phi = atan2f(py, px + 1e-6f) + pi;
    float cx = 1.0f - phi / two_pi;
    float cy = theta / pi;
    gx = 2.0f * cx - 1.0f;
    gy = 2.0f * cy - 1.0f;
}

// atan2 for the fused kernel: octant reduction to t = min/max in [0,1] (v_rcp_f32, 1 ulp), degree-8 minimax polynomial
// in t^2 (max abs error 8.7e-8 in fp32 arithmetic, i.e. below the 2.4e-7 ulp of an angle near pi), then the usual
// reflections.  ~20 VALU + 1 transcendental instead of the ~45 of the library routine.  atan2(0, 0) = 0.
__device__ __forceinline__ float pcl_atan_poly(float t)
{
    float s = t * t;
    float p = 2.4566929979e-03f;
    p = fmaf(p, s, -1.4401224869e-02f);
    p = fmaf(p, s, 3.9780993102e-02f);
    p = fmaf(p, s, -7.2348362183e-02f);
    p = fmaf(p, s, 1.0498935044e-01f);
    p = fmaf(p, s, -1.4161225936e-01f);
    p = fmaf(p, s, 1.9985906258e-01f);
    p = fmaf(p, s, -3.3332596993e-01f);
    p = fmaf(p, s, 9.9999988638e-01f);
    return p * t;
}

__device__ __forceinline__ float pcl_atan2(float y, float x)
{
    const float pi = 3.14159265358979323846f, half_pi = 1.57079632679489661923f;
    float ax = fabsf(x), ay = fabsf(y);
    float mx = fmaxf(fmaxf(ax, ay), 1e-37f), mn = fminf(ax, ay);
    float r = pcl_atan_poly(mn * __builtin_amdgcn_rcpf(mx));
    r = ay > ax ? half_pi - r : r;
    r = x < 0.f ? pi - r : r;
    return copysignf(r, y);
}

// same for y >= 0 (theta = atan2(rho, b)): result in [0, pi], no sign transfer
__device__ __forceinline__ float pcl_atan2_ypos(float y, float x)
{
    const float pi = 3.14159265358979323846f, half_pi = 1.57079632679489661923f;
    float ax = fabsf(x);
    float mx = fmaxf(fmaxf(ax, y), 1e-37f), mn = fminf(ax, y);
    float r = pcl_atan_poly(mn * __builtin_amdgcn_rcpf(mx));
    r = y > ax ? half_pi - r : r;
    return x < 0.f ? pi - r : r;
}

__device__ inline __amdgpu_buffer_rsrc_t pcl_tex_rsrc(const void* pano, int H, int W, int texel_bytes = 16)
{
    // raw buffer, 32-bit offsets, bounds-checked by hardware against the padded texture size
    return __builtin_amdgcn_make_buffer_rsrc((void*)pano, 0, (int)((size_t)(H + 2) * (size_t)(W + 2) * (size_t)texel_bytes),
                                             0x00020000);
}

// bytes per texel of a packed panorama format
__host__ __device__ constexpr int pcl_texel_bytes(int fmt) { return fmt == PCL_PANO_U8 ? 4 : fmt == PCL_PANO_F16 ? 8 : 16; }

// two horizontally adjacent RGBA8 texels in one 8-byte load
__device__ inline pcl_i2 pcl_texel_pair_u8(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff)
{
    return __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
}
// byte k of a dword -> float, one instruction each (left to itself the compiler turns the (float)(a) - (float)(b)
// pattern of the bilinear differences into a much longer integer SDWA sequence)
__device__ __forceinline__ float pcl_ub0(int v) { float f; asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(v)); return f; }
__device__ __forceinline__ float pcl_ub1(int v) { float f; asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(v)); return f; }
__device__ __forceinline__ float pcl_ub2(int v) { float f; asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(v)); return f; }

__device__ inline pcl_f4 pcl_texel(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff)
{
    pcl_i4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return __builtin_bit_cast(pcl_f4, v);
}

// Sum over the 64 lanes of a wave, returned in every lane.  Data-parallel-primitive moves inside the rows of 16 lanes
// (v_add_f32_dpp: no LDS, no waitcnt), then the four row sums through v_readlane.  (__shfl_xor compiles to ds_bpermute_b32 +
// s_waitcnt lgkmcnt(0) per step: 6 dependent LDS round trips per value — measured 4.9 us per block for the 14 sums of
// the loss kernel's epilogue, tools/block_trace.py; this form is ~0.3 us.)
template <int CTRL>
__device__ __forceinline__ float pcl_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float pcl_wave_sum(float v)
{
    v += pcl_dpp<0xB1>(v);       // quad_perm [1,0,3,2]: neighbour
    v += pcl_dpp<0x4E>(v);       // quad_perm [2,3,0,1]: other pair of the quad
    v += pcl_dpp<0x141>(v);      // row_half_mirror: the other quad of the 8
    v += pcl_dpp<0x140>(v);      // row_mirror: the other half of the row -> every lane holds its row's sum
    float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}

// the same for a double (the GD epilogue's second-stage sums): the two dwords move separately, the add is v_add_f64
template <int CTRL>
__device__ __forceinline__ double pcl_dpp_d(double v)
{
    pcl_i2 w = __builtin_bit_cast(pcl_i2, v);
    w.x = __builtin_amdgcn_update_dpp(0, w.x, CTRL, 0xf, 0xf, true);
    w.y = __builtin_amdgcn_update_dpp(0, w.y, CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, w);
}
__device__ __forceinline__ double pcl_readlane_d(double v, int lane)
{
    pcl_i2 w = __builtin_bit_cast(pcl_i2, v);
    w.x = __builtin_amdgcn_readlane(w.x, lane);
    w.y = __builtin_amdgcn_readlane(w.y, lane);
    return __builtin_bit_cast(double, w);
}
__device__ __forceinline__ double pcl_wave_sum_d(double v)
{
    v += pcl_dpp_d<0xB1>(v);
    v += pcl_dpp_d<0x4E>(v);
    v += pcl_dpp_d<0x141>(v);
    v += pcl_dpp_d<0x140>(v);
    return (pcl_readlane_d(v, 0) + pcl_readlane_d(v, 16)) + (pcl_readlane_d(v, 32) + pcl_readlane_d(v, 48));
}

// hipError_t passthrough for launch wrappers
#define PCL_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e_ = hipGetLastError();         \
        if (e_ != hipSuccess) return (int)e_;      \
    } while (0)
