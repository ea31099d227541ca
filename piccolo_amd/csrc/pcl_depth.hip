// pcl_depth.hip — the scatter-min depth mask of the north star, per candidate pose, on the PACKED cloud.
//
// Build-defined (the reference imports torch_scatter.scatter_min at utils.py:6 and never calls it; its loss has no
// occlusion test): for every pose b, pixel = make_pano's pixel of p = R_b (x - t_b) at the panorama's resolution
// (utils.py:158-165), zmin[b][pixel] = min over the points landing there of ||p||^2 (a scatter-min), and point i is
// visible for pose b iff ||p_i||^2 <= zmin * (1 + tau)^2.  The byte mask feeds pcl_loss_kernel<.., VIS = true, ..>.
// Off by default: with the mask off the loss is exactly the reference's.
//
// Two passes over the cloud per call, both projection + one 4-byte access per point-pose:
//   z pass   : 32-bit atomicMin of the squared depth's bit pattern (>= 0, so it orders like the float).  Done as one
//              global atomic per point-pose it is the slow shape of the chip — 616 us at cfg 2, six times the mark pass
//              that does the same projection without atomics — so it is LDS-tiled (below): a block resolves its
//              Morton-compact pixel patch with LDS atomics and flushes it row by row, 276 us.
//   mark pass: re-project, compare with the z-buffer, write one byte per point-pose.
#include <stdlib.h>

#include "pcl_device.h"

struct PclDepthArgs {
    const float* cloud;
    int64_t n, stride;
    const PclPoseRec* poses;
    int B, H, W;
    float tol2;           // (1 + tau)^2
    uint32_t* zbuf;       // [B][H*W]
    uint8_t* visible;     // [B][n]
    const uint32_t* refresh;   // null, or: pose b is processed only when refresh[b * refresh_stride] != 0 (its mask is kept otherwise)
    int refresh_stride;        // in 32-bit words
};

// make_pano's pixel (utils.py:158-165) from the fused kernel's own atan2, plus the squared depth
__device__ __forceinline__ void pcl_depth_point(float x, float y, float z, const PclPoseRec* __restrict__ pr, int H, int W,
                                                int& pix, float& d2)
{
    const float inv_pi = 0.31830988618379067154f;
    float qx = x - pr->t[0], qy = y - pr->t[1], qz = z - pr->t[2];
    float px = fmaf(pr->R[2], qz, fmaf(pr->R[1], qy, pr->R[0] * qx));
    float py = fmaf(pr->R[5], qz, fmaf(pr->R[4], qy, pr->R[3] * qx));
    float pz = fmaf(pr->R[8], qz, fmaf(pr->R[7], qy, pr->R[6] * qx));
    float rho2 = fmaf(px, px, py * py);
    float rho = rho2 * __builtin_amdgcn_rsqf(rho2 + 1e-37f);
    float gx = -pcl_atan2(py, px + 1e-6f) * inv_pi;                       // in [-1, 1]
    float gy = fmaf(pcl_atan2_ypos(rho, pz + 1e-6f), 2.0f * inv_pi, -1.0f);
    int col = (int)((gx + 1.0f) * 0.5f * (float)(W - 1));                 // trunc(((g + 1) / 2) * (res - 1))
    int row = (int)((gy + 1.0f) * 0.5f * (float)(H - 1));
    col = min(max(col, 0), W - 1);
    row = min(max(row, 0), H - 1);
    pix = row * W + col;
    d2 = fmaf(pz, pz, rho2);
}

template <bool MARK>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_depth_kernel(PclDepthArgs a)
{
    // grid.x over points, grid.y over poses; pose record through scalar loads
    const int b = blockIdx.y;
    if (a.refresh && !a.refresh[(int64_t)b * a.refresh_stride]) return;
    const PclPoseRec* __restrict__ pr = a.poses + b;
    const int64_t hw = (int64_t)a.H * a.W;
    for (int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * PCL_BLOCK) {
        int pix; float d2;
        pcl_depth_point(a.cloud[i], a.cloud[a.stride + i], a.cloud[2 * a.stride + i], pr, a.H, a.W, pix, d2);
        uint32_t* cell = a.zbuf + (int64_t)b * hw + pix;
        if (MARK) a.visible[(int64_t)b * a.n + i] = d2 <= __uint_as_float(*cell) * a.tol2 ? 1 : 0;
        else atomicMin(cell, __float_as_uint(d2));
    }
}

// LDS-tiled z pass.  A block takes PCL_ZT_PTS Morton-contiguous points of one pose: their pixels form a compact patch.
// The patch's top-left corner is found with a block-wide min, the points inside a 64 x 128-pixel window at that corner
// are resolved with LDS atomicMin (32 KB tile), and the tile is then flushed row by row: each wave issues its global
// atomicMins on 64 CONSECUTIVE pixels (one 256-byte segment, the shape global atomics run fastest in) and only for
// cells that received a point.  Points outside the window (sparse clouds, the wrap seam) go straight to the global
// z-buffer.  Measured at cfg 2 (1M points, 32 poses, 32M point-poses): 616 us for the direct scatter, 276 us tiled
// (the mark pass, same projection without atomics: 100 us).  The number of global atomics is the same (~one per
// distinct pixel hit); what changes is their shape.
template <int PCL_ZT_H, int PCL_ZT_W, int PCL_ZT_PTS>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_zpass_tiled_kernel(PclDepthArgs a)
{
    constexpr int PCL_ZT_PER_THREAD = PCL_ZT_PTS / PCL_BLOCK;
    __shared__ uint32_t tile[PCL_ZT_H * PCL_ZT_W];
    __shared__ int org[3];
    const int b = blockIdx.y;
    if (a.refresh && !a.refresh[(int64_t)b * a.refresh_stride]) return;            // (block-uniform, before the first barrier)
    const PclPoseRec* __restrict__ pr = a.poses + b;
    uint32_t* __restrict__ zb = a.zbuf + (int64_t)b * a.H * a.W;
    const uint32_t INF = 0x7f800000u;
    for (int i = threadIdx.x; i < PCL_ZT_H * PCL_ZT_W; i += PCL_BLOCK) tile[i] = INF;
    if (threadIdx.x < 3) org[threadIdx.x] = 0;
    __syncthreads();

    int pix[PCL_ZT_PER_THREAD];
    uint32_t key[PCL_ZT_PER_THREAD];
    int rsum = 0, csum = 0, cnt = 0;
    const int64_t base = (int64_t)blockIdx.x * PCL_ZT_PTS;
#pragma unroll
    for (int k = 0; k < PCL_ZT_PER_THREAD; k++) {
        int64_t i = base + k * PCL_BLOCK + threadIdx.x;
        pix[k] = -1;
        key[k] = INF;
        if (i < a.n) {
            float d2;
            pcl_depth_point(a.cloud[i], a.cloud[a.stride + i], a.cloud[2 * a.stride + i], pr, a.H, a.W, pix[k], d2);
            key[k] = __float_as_uint(d2);
            int r = pix[k] / a.W, c = pix[k] - r * a.W;
            rsum += r; csum += c; cnt += 1;
        }
    }
    // window centred on the block's mean pixel (a min-corner anchor is dragged away by a few outliers, e.g. when the
    // chunk straddles two walls)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rsum += __shfl_xor(rsum, o, 64);
        csum += __shfl_xor(csum, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&org[0], rsum); atomicAdd(&org[1], csum); atomicAdd(&org[2], cnt); }
    __syncthreads();
    const int npts = max(org[2], 1);
    const int r0 = org[0] / npts - PCL_ZT_H / 2, c0 = org[1] / npts - PCL_ZT_W / 2;
#pragma unroll
    for (int k = 0; k < PCL_ZT_PER_THREAD; k++) {
        if (pix[k] < 0) continue;
        int r = pix[k] / a.W, c = pix[k] - r * a.W;
        unsigned tr = (unsigned)(r - r0), tc = (unsigned)(c - c0);
        if (tr < PCL_ZT_H && tc < PCL_ZT_W) atomicMin(&tile[tr * PCL_ZT_W + tc], key[k]);
        else atomicMin(&zb[pix[k]], key[k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PCL_ZT_H * PCL_ZT_W; i += PCL_BLOCK) {
        uint32_t v = tile[i];
        int r = r0 + i / PCL_ZT_W, c = c0 + (i & (PCL_ZT_W - 1));
        if (v != INF && r >= 0 && r < a.H && c >= 0 && c < a.W) atomicMin(&zb[(int64_t)r * a.W + c], v);
    }
}

// z-buffer of pose blockIdx.y <- v (skipped for poses whose mask is kept)
__global__ void __launch_bounds__(PCL_BLOCK) pcl_fill_u32_kernel(uint32_t* p, int64_t per_pose, uint32_t v, const uint32_t* __restrict__ refresh,
                                                                 int refresh_stride)
{
    if (refresh && !refresh[(int64_t)blockIdx.y * refresh_stride]) return;
    uint32_t* q = p + (int64_t)blockIdx.y * per_pose;
    for (int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x; i < per_pose; i += (int64_t)gridDim.x * PCL_BLOCK) q[i] = v;
}

size_t pcl_depth_zbuf_bytes(int B, int H, int W) { return (size_t)B * (size_t)H * (size_t)W * sizeof(uint32_t); }

// zbuf: pcl_depth_zbuf_bytes; visible: B * n bytes.  Used by pcl_depth_mask and by the GD loop (pcl_gd.hip).
int pcl_launch_depth_mask(const float* cloud, int64_t n, const PclPoseRec* poses, int B, int H, int W, float tau,
                          uint32_t* zbuf, uint8_t* visible, const uint32_t* refresh, int refresh_stride, hipStream_t s)
{
    PclDepthArgs a;
    a.cloud = cloud; a.n = n; a.stride = pcl_cloud_stride(n);
    a.poses = poses; a.B = B; a.H = H; a.W = W;
    a.tol2 = (1.0f + tau) * (1.0f + tau);
    a.zbuf = zbuf; a.visible = visible; a.refresh = refresh; a.refresh_stride = refresh_stride;
    const int fill_x = B >= 32 ? 64 : (2048 + B - 1) / B;
    hipLaunchKernelGGL(pcl_fill_u32_kernel, dim3((unsigned)fill_x, (unsigned)B), dim3(PCL_BLOCK), 0, s, zbuf, (int64_t)H * W, 0x7f800000u, refresh, refresh_stride);   // +inf
    int64_t want = (n + PCL_BLOCK - 1) / PCL_BLOCK;
    dim3 grid((unsigned)(want < 1024 ? want : 1024), (unsigned)B);
    static const bool direct = getenv("PCL_ZPASS_DIRECT") != nullptr;       // A/B knob: the untiled scatter
    if (direct) hipLaunchKernelGGL(pcl_depth_kernel<false>, grid, dim3(PCL_BLOCK), 0, s, a);
    else {
        // 64 x 128 window, 2048 points per block: best of {32..128} x {64,128} x {256..4096} at cfg 2 (276 us; 288-374 us
        // for the others, 2x slower at 256 points per block — one pixel per point here, unlike the 3x3 splats of
        // pcl_hist.hip where 256 wins; the direct scatter takes 616 us)
        constexpr int TH = 64, TW = 128, PTS = 2048;
        hipLaunchKernelGGL((pcl_zpass_tiled_kernel<TH, TW, PTS>), dim3((unsigned)((n + PTS - 1) / PTS), (unsigned)B),
                           dim3(PCL_BLOCK), 0, s, a);
    }
    hipLaunchKernelGGL(pcl_depth_kernel<true>, grid, dim3(PCL_BLOCK), 0, s, a);
    PCL_LAUNCH_CHECK();
    return 0;
}

__global__ void pcl_depth_pose_setup_kernel(const float* __restrict__ trans, const float* __restrict__ rot, int B, PclPoseRec* recs)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float p[6] = {trans[3 * b], trans[3 * b + 1], trans[3 * b + 2], rot[3 * b], rot[3 * b + 1], rot[3 * b + 2]};
    pcl_write_pose_rec(&recs[b], p);
}

extern "C" size_t pcl_depth_workspace_bytes(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * sizeof(PclPoseRec) + pcl_depth_zbuf_bytes(B, H, W);
}

extern "C" int pcl_depth_mask(const float* cloud, int64_t n, const float* trans, const float* rot, int B, int H, int W, float tau,
                              uint8_t* visible, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!cloud || !trans || !rot || !visible || !workspace || n <= 0 || B <= 0 || B > 65535 || H <= 0 || W <= 0 || !(tau >= 0.f))
        return PCL_EINVAL;
    if (workspace_bytes < pcl_depth_workspace_bytes(B, H, W)) return PCL_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    PclPoseRec* recs = (PclPoseRec*)workspace;
    uint32_t* zbuf = (uint32_t*)((char*)workspace + (size_t)B * sizeof(PclPoseRec));
    hipLaunchKernelGGL(pcl_depth_pose_setup_kernel, dim3((B + 255) / 256), dim3(256), 0, s, trans, rot, B, recs);
    return pcl_launch_depth_mask(cloud, n, recs, B, H, W, tau, zbuf, visible, nullptr, 1, s);
}
