// pcl_loss.hip — fused projection + bilinear sampling + sampling loss (+ hand-derived gradient) for gfx950.
//
// What it replaces in the reference (all un-fused ATen ops over (B,N,3) temporaries):
//   omniloc.py:190-200 / :332-353  p = R (x - t) -> cloud2idx -> sample_from_img -> mask -> ||c - rgb|| -> mean
//   omniloc.py:47,254              loss.backward()  (autograd through the above, incl. grid_sampler_2d_backward)
//
// Work decomposition (HBM-bound gather/scatter shape, no MFMA):
//   - the cloud is 6 SoA planes; a block walks ONE contiguous chunk of it (space-filling-curve order makes the
//     chunk a compact surface patch, so the texels its lanes gather share cache lines and stay in this XCD's L2);
//   - a block evaluates G candidate poses per point it loads (pose records come in through scalar loads), keeping
//     8 accumulators per pose in registers for the whole chunk:  sum ||d||, count, sum g, sum p x g   with
//     g = dL/dp.  The rotation gradient is carried as the torque sum_i p_i x g_i: for R = RZ RY RX,
//     dR/dyaw = [e_z]x R, dR/dpitch = [RZ e_y]x R, dR/droll = [RZ RY e_x]x R, so dL/dangle = axis . sum(p x g)
//     — 3 accumulators instead of the 9 of sum g q^T;
//   - one wave reduction + one LDS reduction per block at the very end, plain stores of the block's partials
//     (no atomics: the second-stage reduce in pcl_finish / the GD epilogue is deterministic).
//
// Per point-pose the kernel reads 24 B of cloud (amortised over G poses) and gathers 4 x 16 B texels.
#include "pcl_device.h"

struct PclLossArgs {
    const float* cloud;      // 6 planes of `stride` floats
    int64_t n, stride;
    const float* pano;       // (H+2, W+2) float4 texels, zero border
    PclDims dims;
    const PclPoseRec* poses; // [B]
    int B;
    const uint8_t* visible;  // nullable [B][n]
    float* partials;         // [nchunks][B][8]
    int nchunks;             // multiple of 8
    int64_t chunk_len;       // multiple of PCL_BLOCK
};

// One point against one pose. acc: 0 sum||d||, 1 count, 2-4 sum g, 5-7 sum p x g.
template <bool GRAD>
__device__ __forceinline__ void pcl_point_pose(float x, float y, float z, float cr, float cg, float cb, bool valid,
                                               const float* __restrict__ R, const float* __restrict__ t,
                                               __amdgpu_buffer_rsrc_t tex, const PclDims& dm, float* acc)
{
    const float inv_pi = 0.31830988618379067154f;
    // q = x - t ; p = R q                                                   (omniloc.py:190-191, :332-338)
    float qx = x - t[0], qy = y - t[1], qz = z - t[2];
    float px = fmaf(R[2], qz, fmaf(R[1], qy, R[0] * qx));
    float py = fmaf(R[5], qz, fmaf(R[4], qy, R[3] * qx));
    float pz = fmaf(R[8], qz, fmaf(R[7], qy, R[6] * qx));
    // cloud2idx (utils.py:44-59): gx = 1 - (atan2(py, a) + pi)/pi = -phi/pi ; gy = 2 theta/pi - 1
    float a = px + 1e-6f, b = pz + 1e-6f;
    float rho2 = fmaf(px, px, py * py);
    float rinv = rho2 > 0.f ? __builtin_amdgcn_rsqf(rho2) : 0.f;
    float rho = rho2 * rinv;
    float phi = atan2f(py, a);
    float theta = atan2f(rho, b);
    float gx = -phi * inv_pi;
    float gy = fmaf(theta, 2.0f * inv_pi, -1.0f);
    // sample_from_img (utils.py:97-98): clip to +-0.99, unnormalise (align_corners=False), +1 for the zero border
    float gxc = __builtin_amdgcn_fmed3f(gx, -0.99f, 0.99f);
    float gyc = __builtin_amdgcn_fmed3f(gy, -0.99f, 0.99f);
    float ix = fmaf(gxc, dm.half_w, dm.off_x);
    float iy = fmaf(gyc, dm.half_h, dm.off_y);
    int x0 = (int)ix, y0 = (int)iy;            // ix, iy > 0 inside the border, so truncation == floor
    float fx = ix - (float)x0, fy = iy - (float)y0;
    int voff = (y0 * dm.Wp + x0) * 16;
    int row = dm.Wp * 16;
    pcl_f4 t00 = pcl_texel(tex, voff, 0);
    pcl_f4 t01 = pcl_texel(tex, voff + 16, 0);
    pcl_f4 t10 = pcl_texel(tex, voff, row);
    pcl_f4 t11 = pcl_texel(tex, voff + 16, row);
    // bilinear: top/bot rows, then vertical; the two partial derivatives fall out of the same differences
    float dt0 = t01.x - t00.x, dt1 = t01.y - t00.y, dt2 = t01.z - t00.z;
    float db0 = t11.x - t10.x, db1 = t11.y - t10.y, db2 = t11.z - t10.z;
    float top0 = fmaf(fx, dt0, t00.x), top1 = fmaf(fx, dt1, t00.y), top2 = fmaf(fx, dt2, t00.z);
    float bot0 = fmaf(fx, db0, t10.x), bot1 = fmaf(fx, db1, t10.y), bot2 = fmaf(fx, db2, t10.z);
    float dv0 = bot0 - top0, dv1 = bot1 - top1, dv2 = bot2 - top2;              // dc/diy
    float c0 = fmaf(fy, dv0, top0), c1 = fmaf(fy, dv1, top1), c2 = fmaf(fy, dv2, top2);
    // mask: sampled colour not exactly (0,0,0)                               (omniloc.py:198, :347)
    bool keep = valid && !(c0 == 0.f && c1 == 0.f && c2 == 0.f);
    float d0 = c0 - cr, d1 = c1 - cg, d2 = c2 - cb;
    float n2 = fmaf(d0, d0, fmaf(d1, d1, d2 * d2));
    float rn = (keep && n2 > 0.f) ? __builtin_amdgcn_rsqf(n2) : 0.f;            // 0 also kills the gradient at n = 0
    acc[0] = fmaf(n2, rn, acc[0]);                                              // ||d|| = n2 * rsqrt(n2)
    acc[1] += keep ? 1.f : 0.f;
    if (GRAD) {
        float dh0 = fmaf(fy, db0 - dt0, dt0), dh1 = fmaf(fy, db1 - dt1, dt1), dh2 = fmaf(fy, db2 - dt2, dt2);  // dc/dix
        float u0 = d0 * rn, u1 = d1 * rn, u2 = d2 * rn;                         // d||d||/dc
        float sx = fmaf(u0, dh0, fmaf(u1, dh1, u2 * dh2));
        float sy = fmaf(u0, dv0, fmaf(u1, dv1, u2 * dv2));
        // through unnormalise + clip (clamp passes the gradient on [-0.99, 0.99]) to the angles
        float dphi = (gx == gxc) ? dm.k_phi * sx : 0.f;                          // dL/dphi   = -(W/2pi) sx
        float dth = (gy == gyc) ? dm.k_theta * sy : 0.f;                         // dL/dtheta =  (H/pi)  sy
        // phi = atan2(py, a): dphi/dpx = -py/s1, dphi/dpy = a/s1 ; theta = atan2(rho, b): dth/drho = b/s2, dth/dpz = -rho/s2
        float s1 = fmaf(a, a, py * py), s2 = fmaf(b, b, rho2);
        float ai = dphi * __builtin_amdgcn_rcpf(s1);
        float bi = dth * __builtin_amdgcn_rcpf(s2);
        float k = b * bi * rinv;                                                 // (dL/drho) / rho
        float g0 = fmaf(k, px, -py * ai);
        float g1 = fmaf(k, py, a * ai);
        float g2 = -rho * bi;
        acc[2] += g0; acc[3] += g1; acc[4] += g2;
        acc[5] = fmaf(py, g2, fmaf(-pz, g1, acc[5]));
        acc[6] = fmaf(pz, g0, fmaf(-px, g2, acc[6]));
        acc[7] = fmaf(px, g1, fmaf(-py, g0, acc[7]));
    }
}

template <int G, bool GRAD, bool VIS>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_loss_kernel(PclLossArgs a)
{
    // XCD-aware mapping: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a contiguous
    // range of (pose group, chunk) work items: neighbouring chunks share texels along their common boundary.
    const int nblk = gridDim.x;                       // = nchunks * ngroups, multiple of 8
    const int per_xcd = nblk >> 3;
    const int v = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int group = v / a.nchunks, chunk = v - group * a.nchunks;
    const int pose0 = group * G;

    __amdgpu_buffer_rsrc_t tex = pcl_tex_rsrc(a.pano, a.dims.H, a.dims.W);
    const float* __restrict__ X = a.cloud;
    const int64_t S = a.stride;

    float acc[G][PCL_NACC];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < PCL_NACC; k++) acc[g][k] = 0.f;

    const int64_t begin = (int64_t)chunk * a.chunk_len;
    int64_t end = begin + a.chunk_len;
    if (end > a.n) end = a.n;
    for (int64_t base = begin; base < end; base += PCL_BLOCK) {
        int64_t i = base + threadIdx.x;
        bool valid = i < end;
        int64_t j = valid ? i : (a.n - 1);
        float x = X[j], y = X[S + j], z = X[2 * S + j];
        float cr = X[3 * S + j], cg = X[4 * S + j], cb = X[5 * S + j];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const PclPoseRec* __restrict__ pr = a.poses + (pose0 + g);
            bool ok = valid;
            if (VIS) ok = ok && a.visible[(int64_t)(pose0 + g) * a.n + j] != 0;
            pcl_point_pose<GRAD>(x, y, z, cr, cg, cb, ok, pr->R, pr->t, tex, a.dims, acc[g]);
        }
    }

    // block reduction: wave shuffle, then 4 waves through LDS, plain store of the partials
    __shared__ float red[PCL_BLOCK / PCL_WAVE][G * PCL_NACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < PCL_NACC; k++) {
            if (!GRAD && k >= 2) continue;
            float s = pcl_wave_sum(acc[g][k]);
            if (lane == 0) red[wave][g * PCL_NACC + k] = s;
        }
    __syncthreads();
    if (threadIdx.x < G * PCL_NACC) {
        int g = threadIdx.x / PCL_NACC, k = threadIdx.x - g * PCL_NACC;
        float s = 0.f;
        if (GRAD || k < 2) s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        a.partials[((int64_t)chunk * a.B + pose0 + g) * PCL_NACC + k] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------
// launch planning (shared with the GD loop)

struct PclPlan {
    int G, ngroups, nchunks;
    int64_t chunk_len;
};

static PclPlan pcl_plan(int64_t n, int B)
{
    PclPlan p;
    p.G = (B % 4 == 0) ? 4 : ((B % 2 == 0) ? 2 : 1);
    p.ngroups = B / p.G;
    // aim at ~4096 blocks (256 CUs x 8 resident blocks x 2 rounds), at least one 256-point step per chunk
    int64_t want = 4096 / p.ngroups;
    if (want < 8) want = 8;
    int64_t max_chunks = (n + PCL_BLOCK - 1) / PCL_BLOCK;
    if (want > max_chunks) want = max_chunks;
    want = ((want + 7) / 8) * 8;
    int64_t len = (n + want - 1) / want;
    len = ((len + PCL_BLOCK - 1) / PCL_BLOCK) * PCL_BLOCK;
    p.nchunks = (int)want;
    p.chunk_len = len;
    return p;
}

size_t pcl_partials_bytes(int64_t n, int B)
{
    PclPlan p = pcl_plan(n, B);
    return (size_t)p.nchunks * (size_t)B * PCL_NACC * sizeof(float);
}

int pcl_plan_nchunks(int64_t n, int B) { return pcl_plan(n, B).nchunks; }

template <int G>
static void pcl_launch_g(const PclLossArgs& a, int nblk, bool grad, bool vis, hipStream_t s)
{
    if (grad) {
        if (vis) hipLaunchKernelGGL((pcl_loss_kernel<G, true, true>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((pcl_loss_kernel<G, true, false>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
    } else {
        if (vis) hipLaunchKernelGGL((pcl_loss_kernel<G, false, true>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((pcl_loss_kernel<G, false, false>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
    }
}

// Enqueue one fused loss(+grad) pass over the cloud for B poses; partials must hold pcl_partials_bytes(n, B).
int pcl_launch_loss(const float* cloud, int64_t n, const float* pano, int H, int W, const PclPoseRec* poses, int B,
                    bool grad, const uint8_t* visible, float* partials, hipStream_t s)
{
    PclPlan p = pcl_plan(n, B);
    PclLossArgs a;
    a.cloud = cloud; a.n = n; a.stride = pcl_cloud_stride(n);
    a.pano = pano; a.dims = pcl_make_dims(H, W);
    a.poses = poses; a.B = B; a.visible = visible; a.partials = partials;
    a.nchunks = p.nchunks; a.chunk_len = p.chunk_len;
    int nblk = p.nchunks * p.ngroups;
    bool vis = visible != nullptr;
    if (p.G == 4) pcl_launch_g<4>(a, nblk, grad, vis, s);
    else if (p.G == 2) pcl_launch_g<2>(a, nblk, grad, vis, s);
    else pcl_launch_g<1>(a, nblk, grad, vis, s);
    PCL_LAUNCH_CHECK();
    return 0;
}
