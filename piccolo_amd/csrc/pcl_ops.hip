// pcl_ops.hip — the stand-alone ops of the path: cloud2idx, sample_from_img, rot_from_ypr, quantile (radix
// select), the scatter-min z-buffer and make_pano.  All element-wise / scatter kernels, HBM- or atomic-bound.
#include "pcl_device.h"

// ------------------------------------------------------------------------------------------------ cloud2idx
// utils.py:16-61.  AoS (n,3) in, (n,2) out: the stand-alone op keeps the reference's tensor layouts.
__global__ void __launch_bounds__(PCL_BLOCK) pcl_cloud2idx_kernel(const float* __restrict__ xyz, int64_t n, float* __restrict__ coord)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= n) return;
    float gx, gy;
    pcl_cloud2idx_point(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], gx, gy);
    reinterpret_cast<float2*>(coord)[i] = make_float2(gx, gy);
}

extern "C" int pcl_cloud2idx(const float* xyz, int64_t n, float* coord, void* stream)
{
    if (!xyz || !coord || n <= 0) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_cloud2idx_kernel, dim3((unsigned)((n + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0,
                       (hipStream_t)stream, xyz, n, coord);
    PCL_LAUNCH_CHECK();
    return 0;
}

// Backward of cloud2idx: grad_xyz = J^T grad_coord per point — what autograd derives for utils.py:44-59 when a caller
// composes the stand-alone op (the fused loss kernel carries its own, packed form of the same chain rule).  IEEE
// divisions: this op is HBM-bound (32 B/point), the arithmetic is free.
__global__ void __launch_bounds__(PCL_BLOCK) pcl_cloud2idx_bwd_kernel(const float* __restrict__ xyz, const float* __restrict__ gcoord,
                                                                   int64_t n, float* __restrict__ gxyz)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float pi = 3.14159265358979323846f;
    float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    float2 G = reinterpret_cast<const float2*>(gcoord)[i];
    float a = x + 1e-6f, b = z + 1e-6f, rho = sqrtf(x * x + y * y);
    float s1 = a * a + y * y, s2 = rho * rho + b * b;
    float dphi = -G.x / pi, dth = 2.f * G.y / pi;              // gx = 1 - phi/pi, gy = 2 theta/pi - 1
    float drho = dth * b / s2;
    float rx = rho > 0.f ? x / rho : 0.f, ry = rho > 0.f ? y / rho : 0.f;      // norm backward: 0 at rho = 0
    gxyz[3 * i] = dphi * (-y / s1) + drho * rx;
    gxyz[3 * i + 1] = dphi * (a / s1) + drho * ry;
    gxyz[3 * i + 2] = dth * (-rho / s2);
}

extern "C" int pcl_cloud2idx_backward(const float* xyz, const float* grad_coord, int64_t n, float* grad_xyz, void* stream)
{
    if (!xyz || !grad_coord || !grad_xyz || n <= 0) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_cloud2idx_bwd_kernel, dim3((unsigned)((n + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0,
                       (hipStream_t)stream, xyz, grad_coord, n, grad_xyz);
    PCL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------- sample_from_img
// utils.py:64-103: clip to +-0.99, grid_sample(bilinear, zeros, align_corners=False).  Same tap arithmetic as
// ATen's grid_sampler_2d (weights (1-fx)(1-fy) ... times the four taps) so the stand-alone op matches to rounding.
// the four taps of a footprint as the fp32 colours the reference samples (level formats: k / 255 by IEEE division)
template <int FMT>
__device__ __forceinline__ void pcl_fetch_taps(__amdgpu_buffer_rsrc_t tex, int x0, int y0, int Wp, pcl_f4& t00, pcl_f4& t01,
                                               pcl_f4& t10, pcl_f4& t11)
{
    if (FMT == PCL_PANO_U8) {
        // levels back to the fp32 values the reference samples: k / 255 (IEEE division, as uint8 -> .float() / 255.)
        int voff = (y0 * Wp + x0) * 4;
        pcl_i2 top = pcl_texel_pair_u8(tex, voff, 0), bot = pcl_texel_pair_u8(tex, voff, Wp * 4);
        t00 = {__fdiv_rn(pcl_ub0(top.x), 255.f), __fdiv_rn(pcl_ub1(top.x), 255.f), __fdiv_rn(pcl_ub2(top.x), 255.f), 0.f};
        t01 = {__fdiv_rn(pcl_ub0(top.y), 255.f), __fdiv_rn(pcl_ub1(top.y), 255.f), __fdiv_rn(pcl_ub2(top.y), 255.f), 0.f};
        t10 = {__fdiv_rn(pcl_ub0(bot.x), 255.f), __fdiv_rn(pcl_ub1(bot.x), 255.f), __fdiv_rn(pcl_ub2(bot.x), 255.f), 0.f};
        t11 = {__fdiv_rn(pcl_ub0(bot.y), 255.f), __fdiv_rn(pcl_ub1(bot.y), 255.f), __fdiv_rn(pcl_ub2(bot.y), 255.f), 0.f};
    } else if (FMT == PCL_PANO_F16) {
        // fp16 levels: two half4 texels per 16-byte load; (whole-vector cast + shuffles, see pcl_bilerp_f16)
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        int voff = (y0 * Wp + x0) * 8;
        h8 top = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(tex, voff, 0, 0));
        h8 bot = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(tex, voff, Wp * 8, 0));
        t00 = {__fdiv_rn((float)top[0], 255.f), __fdiv_rn((float)top[1], 255.f), __fdiv_rn((float)top[2], 255.f), 0.f};
        t01 = {__fdiv_rn((float)top[4], 255.f), __fdiv_rn((float)top[5], 255.f), __fdiv_rn((float)top[6], 255.f), 0.f};
        t10 = {__fdiv_rn((float)bot[0], 255.f), __fdiv_rn((float)bot[1], 255.f), __fdiv_rn((float)bot[2], 255.f), 0.f};
        t11 = {__fdiv_rn((float)bot[4], 255.f), __fdiv_rn((float)bot[5], 255.f), __fdiv_rn((float)bot[6], 255.f), 0.f};
    } else {
        int voff = (y0 * Wp + x0) * 16, row = Wp * 16;
        t00 = pcl_texel(tex, voff, 0); t01 = pcl_texel(tex, voff + 16, 0);
        t10 = pcl_texel(tex, voff, row); t11 = pcl_texel(tex, voff + 16, row);
    }
}

template <int FMT>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_sample_kernel(const void* __restrict__ pano, int H, int W,
                                                               const float* __restrict__ coord, int64_t n,
                                                               float* __restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= n) return;
    __amdgpu_buffer_rsrc_t tex = pcl_tex_rsrc(pano, H, W, pcl_texel_bytes(FMT));
    float2 g = reinterpret_cast<const float2*>(coord)[i];
    float gx = __builtin_amdgcn_fmed3f(g.x, -0.99f, 0.99f), gy = __builtin_amdgcn_fmed3f(g.y, -0.99f, 0.99f);
    float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    float fx0 = floorf(ix), fy0 = floorf(iy);
    int x0 = (int)fx0 + 1, y0 = (int)fy0 + 1;                 // +1: zero border
    // keep the gather inside the bordered texture for any input (|g| <= 0.99 already guarantees it for H,W >= 1)
    x0 = min(max(x0, 0), W); y0 = min(max(y0, 0), H);
    float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
    int Wp = W + 2;
    pcl_f4 t00, t01, t10, t11;
    pcl_fetch_taps<FMT>(tex, x0, y0, Wp, t00, t01, t10, t11);
    float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
    out[3 * i] = t00.x * nw + t01.x * ne + t10.x * sw + t11.x * se;
    out[3 * i + 1] = t00.y * nw + t01.y * ne + t10.y * sw + t11.y * se;
    out[3 * i + 2] = t00.z * nw + t01.z * ne + t10.z * sw + t11.z * se;
}

extern "C" int pcl_sample_from_img(const void* pano, int pano_format, int H, int W, const float* coord, int64_t n, float* rgb_out,
                                   void* stream)
{
    if (!pano || !coord || !rgb_out || n <= 0 || H <= 0 || W <= 0) return PCL_EINVAL;
    if (pano_format != PCL_PANO_F32 && pano_format != PCL_PANO_U8 && pano_format != PCL_PANO_F16) return PCL_EINVAL;
    dim3 grid((unsigned)((n + PCL_BLOCK - 1) / PCL_BLOCK));
    if (pano_format == PCL_PANO_U8)
        hipLaunchKernelGGL(pcl_sample_kernel<PCL_PANO_U8>, grid, dim3(PCL_BLOCK), 0, (hipStream_t)stream, pano, H, W, coord, n, rgb_out);
    else if (pano_format == PCL_PANO_F16)
        hipLaunchKernelGGL(pcl_sample_kernel<PCL_PANO_F16>, grid, dim3(PCL_BLOCK), 0, (hipStream_t)stream, pano, H, W, coord, n, rgb_out);
    else
        hipLaunchKernelGGL(pcl_sample_kernel<PCL_PANO_F32>, grid, dim3(PCL_BLOCK), 0, (hipStream_t)stream, pano, H, W, coord, n, rgb_out);
    PCL_LAUNCH_CHECK();
    return 0;
}

// Backward of sample_from_img (torch.clip -> grid_sampler_2d, utils.py:96-98): per point
//   grad_coord = [in range] (size / 2) <grad_out, d out / d (ix, iy)>      (clamp passes the gradient on [-0.99, 0.99])
//   grad_img[tap] += weight(tap) * grad_out                                  (float atomics, like ATen's own GPU backward;
//                                                                             taps in the zero border are dropped)
// Either output may be null.
template <int FMT>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_sample_bwd_kernel(const void* __restrict__ pano, int H, int W,
                                                                   const float* __restrict__ coord, const float* __restrict__ gout,
                                                                   int64_t n, float* __restrict__ gcoord, float* __restrict__ gimg)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= n) return;
    __amdgpu_buffer_rsrc_t tex = pcl_tex_rsrc(pano, H, W, pcl_texel_bytes(FMT));
    float2 g = reinterpret_cast<const float2*>(coord)[i];
    const bool in_x = g.x >= -0.99f && g.x <= 0.99f, in_y = g.y >= -0.99f && g.y <= 0.99f;
    float gx = __builtin_amdgcn_fmed3f(g.x, -0.99f, 0.99f), gy = __builtin_amdgcn_fmed3f(g.y, -0.99f, 0.99f);
    float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    float fx0 = floorf(ix), fy0 = floorf(iy);
    int x0 = min(max((int)fx0 + 1, 0), W), y0 = min(max((int)fy0 + 1, 0), H);      // +1: zero border
    float wx1 = ix - fx0, wx0 = (fx0 + 1.f) - ix, wy1 = iy - fy0, wy0 = (fy0 + 1.f) - iy;
    float go0 = gout[3 * i], go1 = gout[3 * i + 1], go2 = gout[3 * i + 2];
    if (gcoord) {
        pcl_f4 t00, t01, t10, t11;
        pcl_fetch_taps<FMT>(tex, x0, y0, W + 2, t00, t01, t10, t11);
        float dx0 = (t01.x - t00.x) * wy0 + (t11.x - t10.x) * wy1, dy0 = (t10.x - t00.x) * wx0 + (t11.x - t01.x) * wx1;
        float dx1 = (t01.y - t00.y) * wy0 + (t11.y - t10.y) * wy1, dy1 = (t10.y - t00.y) * wx0 + (t11.y - t01.y) * wx1;
        float dx2 = (t01.z - t00.z) * wy0 + (t11.z - t10.z) * wy1, dy2 = (t10.z - t00.z) * wx0 + (t11.z - t01.z) * wx1;
        float sx = go0 * dx0 + go1 * dx1 + go2 * dx2, sy = go0 * dy0 + go1 * dy1 + go2 * dy2;
        reinterpret_cast<float2*>(gcoord)[i] = make_float2(in_x ? sx * (0.5f * (float)W) : 0.f, in_y ? sy * (0.5f * (float)H) : 0.f);
    }
    if (gimg) {
        const float w[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            int xx = x0 - 1 + (t & 1), yy = y0 - 1 + (t >> 1);                      // back to image coordinates
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            float* p = gimg + ((int64_t)yy * W + xx) * 3;
            atomicAdd(p, w[t] * go0); atomicAdd(p + 1, w[t] * go1); atomicAdd(p + 2, w[t] * go2);
        }
    }
}

extern "C" int pcl_sample_from_img_backward(const void* pano, int pano_format, int H, int W, const float* coord, const float* grad_rgb,
                                            int64_t n, float* grad_coord, float* grad_img, void* stream)
{
    if (!pano || !coord || !grad_rgb || (!grad_coord && !grad_img) || n <= 0 || H <= 0 || W <= 0) return PCL_EINVAL;
    if (pano_format != PCL_PANO_F32 && pano_format != PCL_PANO_U8 && pano_format != PCL_PANO_F16) return PCL_EINVAL;
    dim3 grid((unsigned)((n + PCL_BLOCK - 1) / PCL_BLOCK));
    hipStream_t s = (hipStream_t)stream;
    if (pano_format == PCL_PANO_U8)
        hipLaunchKernelGGL(pcl_sample_bwd_kernel<PCL_PANO_U8>, grid, dim3(PCL_BLOCK), 0, s, pano, H, W, coord, grad_rgb, n, grad_coord, grad_img);
    else if (pano_format == PCL_PANO_F16)
        hipLaunchKernelGGL(pcl_sample_bwd_kernel<PCL_PANO_F16>, grid, dim3(PCL_BLOCK), 0, s, pano, H, W, coord, grad_rgb, n, grad_coord, grad_img);
    else
        hipLaunchKernelGGL(pcl_sample_bwd_kernel<PCL_PANO_F32>, grid, dim3(PCL_BLOCK), 0, s, pano, H, W, coord, grad_rgb, n, grad_coord, grad_img);
    PCL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------- rot_from_ypr
__global__ void pcl_rot_kernel(const float* __restrict__ rot, int B, float* __restrict__ R)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float r[9];
    pcl_rot_from_ypr(rot[3 * b], rot[3 * b + 1], rot[3 * b + 2], r);
    for (int k = 0; k < 9; k++) R[9 * b + k] = r[k];
}

extern "C" int pcl_rot_from_ypr(const float* rot, int B, float* R, void* stream)
{
    if (!rot || !R || B <= 0) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_rot_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, rot, B, R);
    PCL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------ transform_cloud
__global__ void __launch_bounds__(PCL_BLOCK) pcl_transform_kernel(const float* __restrict__ xyz, int64_t n,
                                                                  const float* __restrict__ trans,
                                                                  const float* __restrict__ rot, float* __restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= n) return;
    float R[9];
    pcl_rot_from_ypr(rot[0], rot[1], rot[2], R);   // wave-uniform; cheap next to the 24 B/point of traffic
    float qx = xyz[3 * i] - trans[0], qy = xyz[3 * i + 1] - trans[1], qz = xyz[3 * i + 2] - trans[2];
    out[3 * i] = fmaf(R[2], qz, fmaf(R[1], qy, R[0] * qx));
    out[3 * i + 1] = fmaf(R[5], qz, fmaf(R[4], qy, R[3] * qx));
    out[3 * i + 2] = fmaf(R[8], qz, fmaf(R[7], qy, R[6] * qx));
}

extern "C" int pcl_transform_cloud(const float* xyz, int64_t n, const float* trans, const float* rot, float* out, void* stream)
{
    if (!xyz || !trans || !rot || !out || n <= 0) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_transform_kernel, dim3((unsigned)((n + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0,
                       (hipStream_t)stream, xyz, n, trans, rot, out);
    PCL_LAUNCH_CHECK();
    return 0;
}

// -------------------------------------------------------------------------------------------------- quantile
// utils.py:208-229: x[argsort(x)[int(n q)]] and x[argsort(x)[int(n (1-q))]] for the three xyz columns
// (omniloc.py:53-55, :245-247).  The reference does six full argsorts (per GD iteration in the sequential path);
// here: exact order statistics by a 4-pass MSB radix select over order-preserving uint32 keys, six targets
// (3 columns x 2 ranks) selected together.  Integer atomics only -> deterministic.
struct PclSelState {
    unsigned long long rank[6];
    uint32_t prefix[6];
    uint32_t pad[2];
    unsigned long long hist[4][6][256];
};

extern "C" size_t pcl_quantile_workspace_bytes(void) { return sizeof(PclSelState); }

__device__ inline uint32_t pcl_f2key(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float pcl_key2f(uint32_t k)
{
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

__global__ void pcl_sel_init_kernel(PclSelState* st, unsigned long long r_lo, unsigned long long r_hi)
{
    int t = threadIdx.x;
    if (t < 6) { st->rank[t] = (t & 1) ? r_hi : r_lo; st->prefix[t] = 0u; }
    for (int i = t; i < 4 * 6 * 256; i += blockDim.x) (&st->hist[0][0][0])[i] = 0ull;
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_sel_hist_kernel(const float* __restrict__ xyz, int64_t n, PclSelState* st, int pass)
{
    __shared__ uint32_t h[6][256];
    for (int i = threadIdx.x; i < 6 * 256; i += PCL_BLOCK) (&h[0][0])[i] = 0u;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const uint32_t known = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    uint32_t prefix[6];
#pragma unroll
    for (int t = 0; t < 6; t++) prefix[t] = st->prefix[t];
    for (int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PCL_BLOCK) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            uint32_t k = pcl_f2key(xyz[3 * i + c]);
            uint32_t bin = (k >> shift) & 255u;
            if ((k & known) == prefix[2 * c]) atomicAdd(&h[2 * c][bin], 1u);
            if ((k & known) == prefix[2 * c + 1]) atomicAdd(&h[2 * c + 1][bin], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 6 * 256; i += PCL_BLOCK) {
        uint32_t v = (&h[0][0])[i];
        if (v) atomicAdd(&st->hist[pass][0][0] + i, (unsigned long long)v);
    }
}

__global__ void pcl_sel_scan_kernel(PclSelState* st, int pass, float* box)
{
    int t = threadIdx.x;
    if (t >= 6) return;
    const int shift = 24 - 8 * pass;
    unsigned long long r = st->rank[t], cum = 0;
    int bin = 255;
    for (int b = 0; b < 256; b++) {
        unsigned long long c = st->hist[pass][t][b];
        if (r < cum + c) { bin = b; break; }
        cum += c;
    }
    st->rank[t] = r - cum;
    uint32_t p = st->prefix[t] | ((uint32_t)bin << shift);
    st->prefix[t] = p;
    if (pass == 3) box[t] = pcl_key2f(p);
}

extern "C" int pcl_quantile_box(const float* xyz, int64_t n, double q, float* box, void* workspace, void* stream)
{
    if (!xyz || !box || !workspace || n <= 0 || !(q >= 0.0 && q <= 1.0)) return PCL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    PclSelState* st = (PclSelState*)workspace;
    long long r_lo = (long long)((double)n * q), r_hi = (long long)((double)n * (1.0 - q));  // int(len(x) * q), utils.py:223-224
    if (r_lo < 0 || r_lo >= n || r_hi < 0 || r_hi >= n) return PCL_EINVAL;                   // the reference raises IndexError
    hipLaunchKernelGGL(pcl_sel_init_kernel, dim3(1), dim3(256), 0, s, st, (unsigned long long)r_lo, (unsigned long long)r_hi);
    int64_t want = (n + PCL_BLOCK - 1) / PCL_BLOCK;
    unsigned nblk = (unsigned)(want < 2048 ? want : 2048);
    for (int pass = 0; pass < 4; pass++) {
        hipLaunchKernelGGL(pcl_sel_hist_kernel, dim3(nblk), dim3(PCL_BLOCK), 0, s, xyz, n, st, pass);
        hipLaunchKernelGGL(pcl_sel_scan_kernel, dim3(1), dim3(64), 0, s, st, pass, box);
    }
    PCL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------- scatter-min z-buffer, make_pano
// make_pano's pixel of a camera-frame point (utils.py:158-165): trunc(((g + 1) / 2) * (res - 1)), same fp32 op order.
__device__ inline void pcl_pano_pixel(float px, float py, float pz, int H, int W, int& row, int& col)
{
    float gx, gy;
    pcl_cloud2idx_point(px, py, pz, gx, gy);
    float cx = (gx + 1.0f) / 2.0f * (float)(W - 1);
    float cy = (gy + 1.0f) / 2.0f * (float)(H - 1);
    col = min(max((int)cx, 0), W - 1);
    row = min(max((int)cy, 0), H - 1);
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_fill_u64_kernel(uint64_t* p, int64_t n, uint64_t v)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i < n) p[i] = v;
}

// zbuf[pixel] = min (depth_bits << 32 | index): 64-bit atomicMin resolves "nearest point per pixel" in one pass
// (depth >= 0, so its bit pattern orders like the float; ties go to the smallest index).
__global__ void __launch_bounds__(PCL_BLOCK) pcl_scatter_min_kernel(const float* __restrict__ xyz, int64_t n, int H, int W,
                                                                    unsigned long long* __restrict__ zbuf)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= n) return;
    float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    int row, col;
    pcl_pano_pixel(x, y, z, H, W, row, col);
    float d = sqrtf(x * x + y * y + z * z);                       // torch.norm(xyz, dim=-1), utils.py:152
    unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(uint32_t)i;
    atomicMin(&zbuf[(int64_t)row * W + col], key);
}

extern "C" int pcl_scatter_min_depth(const float* xyz_cam, int64_t n, int H, int W, uint64_t* zbuf, void* stream)
{
    if (!xyz_cam || !zbuf || n <= 0 || n > 0xffffffffll || H <= 0 || W <= 0) return PCL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int64_t hw = (int64_t)H * W;
    hipLaunchKernelGGL(pcl_fill_u64_kernel, dim3((unsigned)((hw + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0, s, zbuf,
                       hw, ~0ull);
    hipLaunchKernelGGL(pcl_scatter_min_kernel, dim3((unsigned)((n + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0, s,
                       xyz_cam, n, H, W, (unsigned long long*)zbuf);
    PCL_LAUNCH_CHECK();
    return 0;
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_scatter_unpack_kernel(const uint64_t* __restrict__ zbuf, int64_t n, int64_t hw,
                                                                       float* __restrict__ zmin, int64_t* __restrict__ arg)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= hw) return;
    uint64_t k = zbuf[i];
    bool empty = k == ~0ull;
    zmin[i] = empty ? 0.f : __uint_as_float((uint32_t)(k >> 32));   // torch_scatter: empty bins keep the 0 fill
    arg[i] = empty ? n : (int64_t)(k & 0xffffffffull);              //                and arg = dim size
}

extern "C" int pcl_scatter_min_unpack(const uint64_t* zbuf, int64_t n, int H, int W, float* zmin, int64_t* argmin, void* stream)
{
    if (!zbuf || !zmin || !argmin || H <= 0 || W <= 0) return PCL_EINVAL;
    int64_t hw = (int64_t)H * W;
    hipLaunchKernelGGL(pcl_scatter_unpack_kernel, dim3((unsigned)((hw + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0,
                       (hipStream_t)stream, zbuf, n, hw, zmin, argmin);
    PCL_LAUNCH_CHECK();
    return 0;
}

// make_pano (utils.py:134-205).  The reference sorts far->near and issues nine index_put_ passes (idx8..idx1, then the
// centre, :190-198): a later pass overwrites an earlier one, and inside a pass the nearest point is the intended
// winner.  One 64-bit atomicMin per (point, splat offset) encodes exactly that priority:
//   key = (8 - pass) << 60 | depth_bits << 29 | (2^29 - 1 - index)      (min = latest pass, then nearest, then largest index)
__global__ void __launch_bounds__(PCL_BLOCK) pcl_splat_kernel(const float* __restrict__ xyz, int64_t n, int H, int W,
                                                              unsigned long long* __restrict__ zbuf)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= n) return;
    float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    int row, col;
    pcl_pano_pixel(x, y, z, H, W, row, col);
    float d = sqrtf(x * x + y * y + z * z);
    unsigned long long base = ((unsigned long long)__float_as_uint(d) << 29) | (unsigned long long)(0x1fffffffu - (uint32_t)i);
    const int drow[9] = {0, 0, -1, -1, -1, 1, 1, 1, 0};   // pass order idx8,7,6,5,4,3,2,1,centre (utils.py:173-198)
    const int dcol[9] = {-1, 1, -1, 0, 1, -1, 0, 1, 0};
#pragma unroll
    for (int p = 0; p < 9; p++) {
        int r = min(max(row + drow[p], 0), H - 1), c = min(max(col + dcol[p], 0), W - 1);
        atomicMin(&zbuf[(int64_t)r * W + c], ((unsigned long long)(8 - p) << 60) | base);
    }
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_pano_resolve_kernel(const uint64_t* __restrict__ zbuf, int64_t hw,
                                                                     const float* __restrict__ rgb, float* __restrict__ image)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= hw) return;
    uint64_t k = zbuf[i];
    float r = 0.f, g = 0.f, b = 0.f;
    if (k != ~0ull) {
        int64_t j = (int64_t)(0x1fffffffu - (uint32_t)(k & 0x1fffffffull));
        r = rgb[3 * j] * 255.f; g = rgb[3 * j + 1] * 255.f; b = rgb[3 * j + 2] * 255.f;   // image * 255, utils.py:200
    }
    image[3 * i] = r; image[3 * i + 1] = g; image[3 * i + 2] = b;
}

extern "C" int pcl_make_pano(const float* xyz_cam, const float* rgb, int64_t n, int H, int W, float* image, uint64_t* workspace,
                             void* stream)
{
    if (!xyz_cam || !rgb || !image || !workspace || n <= 0 || n > 0x1fffffffll || H <= 0 || W <= 0) return PCL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int64_t hw = (int64_t)H * W;
    unsigned gp = (unsigned)((hw + PCL_BLOCK - 1) / PCL_BLOCK);
    hipLaunchKernelGGL(pcl_fill_u64_kernel, dim3(gp), dim3(PCL_BLOCK), 0, s, workspace, hw, ~0ull);
    hipLaunchKernelGGL(pcl_splat_kernel, dim3((unsigned)((n + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0, s, xyz_cam, n, H,
                       W, (unsigned long long*)workspace);
    hipLaunchKernelGGL(pcl_pano_resolve_kernel, dim3(gp), dim3(PCL_BLOCK), 0, s, workspace, hw, rgb, image);
    PCL_LAUNCH_CHECK();
    return 0;
}
