// pcl_hist.hip — the second trimming stage of the initialisation (reference utils.py:510-588, color_utils.py:68-144)
// for a BATCH of candidate poses, fused into three kernels:
//   1. pcl_splat_poses_kernel : make_pano's z-buffered 3x3 splat (see pcl_ops.hip) for every candidate pose at once,
//                               straight from the packed world-frame cloud (p = R (x - t) computed in the kernel), with
//                               LDS-tiled atomicMin;
//   2. pcl_hist_accum_kernel  : per (candidate, block, pixel range): resolve the z-buffer to colours and histogram the
//                               pixels where both the render and the query are non-black in LDS (mode 0: the query image's
//                               own non-black pixels), merged into global counters with integer atomics;
//   3. pcl_hist_final_kernel  : normalise (query) / intersect with the query histogram (candidates).
// The reference renders one candidate at a time (argsort + nine index_put_ passes) and builds every histogram with a
// dozen tensor ops; the rendered image is never materialised here.
// Only the middle block rows h = 1 .. num_split_h - 2 are used (utils.py:556); block j <-> (h = 1 + j / nsw, w = j % nsw).
#include "pcl_device.h"

#define PCL_HBINS 512   // 8 x 8 x 8

__device__ inline void pcl_pano_pixel_ref(float px, float py, float pz, int H, int W, int& row, int& col)
{
    // make_pano's pixel (utils.py:158-165) with the reference's operation order (same as pcl_ops.hip)
    float gx, gy;
    pcl_cloud2idx_point(px, py, pz, gx, gy);
    float cx = (gx + 1.0f) / 2.0f * (float)(W - 1);
    float cy = (gy + 1.0f) / 2.0f * (float)(H - 1);
    col = min(max((int)cx, 0), W - 1);
    row = min(max((int)cy, 0), H - 1);
}

// Batched splat over the PACKED (Morton-ordered) cloud, LDS-tiled like the depth mask's z pass (pcl_depth.hip): a block
// takes PTS contiguous points of one candidate, resolves the 3x3 splats that fall inside a TH x TW-pixel window centred on
// the block's mean pixel with 64-bit LDS atomicMin, then flushes the touched cells row by row (each wave's global
// atomics on 64 consecutive pixels).  The priority key is make_pano's: latest pass, then nearest, then largest index
// (pcl_ops.hip); the index is the PACKED slot, so the colour is looked up in the packed cloud's (negated) colour planes.
template <int TH, int TW, int PTS>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_splat_poses_kernel(const float* __restrict__ cloud, int64_t n, int64_t stride,
                                                                    const PclPoseRec* __restrict__ poses, int H, int W,
                                                                    unsigned long long* __restrict__ zbuf)
{
    constexpr int PER_THREAD = PTS / PCL_BLOCK;
    __shared__ unsigned long long tile[TH * TW];
    __shared__ int org[3];
    const PclPoseRec* __restrict__ pr = poses + blockIdx.y;
    unsigned long long* __restrict__ zb = zbuf + (int64_t)blockIdx.y * H * W;
    const int drow[9] = {0, 0, -1, -1, -1, 1, 1, 1, 0};   // pass order idx8,7,6,5,4,3,2,1,centre (utils.py:173-198)
    const int dcol[9] = {-1, 1, -1, 0, 1, -1, 0, 1, 0};
    for (int i = threadIdx.x; i < TH * TW; i += PCL_BLOCK) tile[i] = ~0ull;
    if (threadIdx.x < 3) org[threadIdx.x] = 0;
    __syncthreads();

    int row[PER_THREAD], col[PER_THREAD];
    unsigned long long base[PER_THREAD];
    int rsum = 0, csum = 0, cnt = 0;
    const int64_t first = (int64_t)blockIdx.x * PTS;
#pragma unroll
    for (int k = 0; k < PER_THREAD; k++) {
        int64_t i = first + k * PCL_BLOCK + threadIdx.x;
        row[k] = -1;
        if (i < n) {
            float qx = cloud[i] - pr->t[0], qy = cloud[stride + i] - pr->t[1], qz = cloud[2 * stride + i] - pr->t[2];
            float px = fmaf(pr->R[2], qz, fmaf(pr->R[1], qy, pr->R[0] * qx));
            float py = fmaf(pr->R[5], qz, fmaf(pr->R[4], qy, pr->R[3] * qx));
            float pz = fmaf(pr->R[8], qz, fmaf(pr->R[7], qy, pr->R[6] * qx));
            pcl_pano_pixel_ref(px, py, pz, H, W, row[k], col[k]);
            float d = sqrtf(px * px + py * py + pz * pz);
            base[k] = ((unsigned long long)__float_as_uint(d) << 29) | (unsigned long long)(0x1fffffffu - (uint32_t)i);
            rsum += row[k]; csum += col[k]; cnt += 1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        rsum += __shfl_xor(rsum, o, 64);
        csum += __shfl_xor(csum, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&org[0], rsum); atomicAdd(&org[1], csum); atomicAdd(&org[2], cnt); }
    __syncthreads();
    const int npts = max(org[2], 1);
    const int r0 = org[0] / npts - TH / 2, c0 = org[1] / npts - TW / 2;
#pragma unroll
    for (int k = 0; k < PER_THREAD; k++) {
        if (row[k] < 0) continue;
#pragma unroll
        for (int p = 0; p < 9; p++) {
            int r = min(max(row[k] + drow[p], 0), H - 1), c = min(max(col[k] + dcol[p], 0), W - 1);
            unsigned long long key = ((unsigned long long)(8 - p) << 60) | base[k];
            unsigned tr = (unsigned)(r - r0), tc = (unsigned)(c - c0);
            if (tr < TH && tc < TW) atomicMin(&tile[tr * TW + tc], key);
            else atomicMin(&zb[(int64_t)r * W + c], key);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TH * TW; i += PCL_BLOCK) {
        unsigned long long v = tile[i];
        int r = r0 + i / TW, c = c0 + i % TW;
        if (v != ~0ull && r >= 0 && r < H && c >= 0 && c < W) atomicMin(&zb[(int64_t)r * W + c], v);
    }
}

__device__ inline int pcl_hist_code(float r, float g, float b)
{
    // value.long() // ceil(255 / 8) per channel, r + 8 g + 64 b (color_utils.py:86-95)
    return ((int)r >> 5) + 8 * ((int)g >> 5) + 64 * ((int)b >> 5);
}

// Histograms in two steps so that a handful of image blocks still fills the chip: every (block, candidate) is cut into
// PCL_HSUB pixel ranges, each range is histogrammed in LDS by its own workgroup and its non-empty bins are added to a
// global counter array (integer atomics: deterministic); a finalise kernel then normalises / intersects.
#define PCL_HSUB 16

// MODE 0: query image (zbuf unused, cand = 0)   MODE 1: candidate renders
template <int MODE>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_hist_accum_kernel(const unsigned long long* __restrict__ zbuf, const float* __restrict__ cloud,
                                                                   int64_t stride, const float* __restrict__ img, int H, int W, int nsh,
                                                                   int nsw, unsigned int* __restrict__ ghist)
{
    __shared__ unsigned int hist[PCL_HBINS];
    const int blk = blockIdx.x / PCL_HSUB, sub = blockIdx.x - blk * PCL_HSUB, cand = blockIdx.y, nblk = gridDim.x / PCL_HSUB;
    const int bh = H / nsh, bw = W / nsw;
    const int h = 1 + blk / nsw, w = blk - (h - 1) * nsw;
    for (int i = threadIdx.x; i < PCL_HBINS; i += PCL_BLOCK) hist[i] = 0u;
    __syncthreads();
    const unsigned long long* zb = MODE == 1 ? zbuf + (int64_t)cand * H * W : nullptr;
    const int total = bh * bw, per = (total + PCL_HSUB - 1) / PCL_HSUB;
    const int lo = sub * per, hi = min(lo + per, total);
    for (int idx = lo + threadIdx.x; idx < hi; idx += PCL_BLOCK) {
        int r = h * bh + idx / bw, c = w * bw + idx % bw;
        int64_t pix = (int64_t)r * W + c;
        float q0 = img[3 * pix], q1 = img[3 * pix + 1], q2 = img[3 * pix + 2];
        bool qm = !(q0 == 0.f && q1 == 0.f && q2 == 0.f);            // query pixel not black
        if (MODE == 0) {
            if (qm) atomicAdd(&hist[pcl_hist_code(q0 * 255.f, q1 * 255.f, q2 * 255.f)], 1u);
        } else {
            unsigned long long k = zb[pix];
            if (qm && k != ~0ull) {
                int64_t j = (int64_t)(0x1fffffffu - (uint32_t)(k & 0x1fffffffull));
                // image * 255 (utils.py:200); the packed cloud holds -rgb in planes 3..5
                float p0 = -cloud[3 * stride + j] * 255.f, p1 = -cloud[4 * stride + j] * 255.f, p2 = -cloud[5 * stride + j] * 255.f;
                if (!(p0 == 0.f && p1 == 0.f && p2 == 0.f)) atomicAdd(&hist[pcl_hist_code(p0, p1, p2)], 1u);
            }
        }
    }
    __syncthreads();
    unsigned int* g = ghist + ((int64_t)cand * nblk + blk) * PCL_HBINS;
    for (int i = threadIdx.x; i < PCL_HBINS; i += PCL_BLOCK)
        if (hist[i]) atomicAdd(&g[i], hist[i]);
}

// MODE 0: qhist[blk][512] = hist / hist.sum(), nimg[blk].   MODE 1: inter[cand][blk] = sum min(h / h.sum(), qhist), nproj.
template <int MODE>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_hist_final_kernel(const unsigned int* __restrict__ ghist, float* __restrict__ qhist,
                                                                   int* __restrict__ nimg, float* __restrict__ inter,
                                                                   int* __restrict__ nproj)
{
    __shared__ float red[PCL_BLOCK / PCL_WAVE];
    const int blk = blockIdx.x, cand = blockIdx.y, nblk = gridDim.x;
    const unsigned int* g = ghist + ((int64_t)cand * nblk + blk) * PCL_HBINS;
    unsigned int c0 = g[threadIdx.x], c1 = g[threadIdx.x + PCL_BLOCK];
    float s = pcl_wave_sum((float)(c0 + c1));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float total = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    if (MODE == 0) {
        qhist[(int64_t)blk * PCL_HBINS + threadIdx.x] = (float)c0 / total;              // hist / hist.sum()
        qhist[(int64_t)blk * PCL_HBINS + threadIdx.x + PCL_BLOCK] = (float)c1 / total;
        if (threadIdx.x == 0) nimg[blk] = (int)total;
    } else {
        const float* qh = qhist + (int64_t)blk * PCL_HBINS;
        float v = fminf((float)c0 / total, qh[threadIdx.x]) + fminf((float)c1 / total, qh[threadIdx.x + PCL_BLOCK]);
        if (!(total > 0.f)) v = 0.f;
        v = pcl_wave_sum(v);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            inter[(int64_t)cand * nblk + blk] = red[0] + red[1] + red[2] + red[3];      // torch.min(h1, h2).sum()
            nproj[(int64_t)cand * nblk + blk] = (int)total;
        }
    }
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_fill_u64b_kernel(unsigned long long* p, int64_t n, unsigned long long v)
{
    for (int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PCL_BLOCK) p[i] = v;
}

__global__ void pcl_hist_pose_setup_kernel(const float* __restrict__ trans, const float* __restrict__ rot, int B, PclPoseRec* recs)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float p[6] = {trans[3 * b], trans[3 * b + 1], trans[3 * b + 2], rot[3 * b], rot[3 * b + 1], rot[3 * b + 2]};
    pcl_write_pose_rec(&recs[b], p);
}

static size_t hist_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t pcl_hist_trim_workspace_bytes(int ncand, int H, int W, int nsh, int nsw)
{
    if (ncand <= 0 || H <= 0 || W <= 0 || nsh < 3 || nsw < 1) return 0;
    const size_t nblk = (size_t)(nsh - 2) * nsw;
    return hist_align((size_t)ncand * sizeof(PclPoseRec)) + hist_align((size_t)ncand * H * W * 8) +
           hist_align(nblk * PCL_HBINS * sizeof(float)) + hist_align((size_t)(ncand + 1) * nblk * PCL_HBINS * sizeof(unsigned int));
}

extern "C" int pcl_hist_trim_scores(const float* cloud, int64_t n, const float* img_hwc, int H, int W,
                                    const float* trans, const float* rot, int ncand, int nsh, int nsw, float* inter, int* nproj,
                                    int* nimg, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!cloud || !img_hwc || !trans || !rot || !inter || !nproj || !nimg || !workspace) return PCL_EINVAL;
    if (n <= 0 || n > 0x1fffffffll || ncand <= 0 || ncand > 65535 || H <= 0 || W <= 0 || nsh < 3 || nsw < 1) return PCL_EINVAL;
    if (H / nsh <= 0 || W / nsw <= 0) return PCL_EINVAL;
    if (workspace_bytes < pcl_hist_trim_workspace_bytes(ncand, H, W, nsh, nsw)) return PCL_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    PclPoseRec* recs = (PclPoseRec*)ws;
    ws += hist_align((size_t)ncand * sizeof(PclPoseRec));
    unsigned long long* zbuf = (unsigned long long*)ws;
    ws += hist_align((size_t)ncand * H * W * 8);
    float* qhist = (float*)ws;
    const int nblk = (nsh - 2) * nsw;
    ws += hist_align((size_t)nblk * PCL_HBINS * sizeof(float));
    unsigned int* ghist_q = (unsigned int*)ws;                       // [nblk][512], then [ncand][nblk][512]
    unsigned int* ghist_c = ghist_q + (size_t)nblk * PCL_HBINS;
    (void)hipMemsetAsync(ghist_q, 0, (size_t)(ncand + 1) * nblk * PCL_HBINS * sizeof(unsigned int), s);
    hipLaunchKernelGGL(pcl_hist_pose_setup_kernel, dim3((ncand + 255) / 256), dim3(256), 0, s, trans, rot, ncand, recs);
    hipLaunchKernelGGL(pcl_fill_u64b_kernel, dim3(2048), dim3(PCL_BLOCK), 0, s, zbuf, (int64_t)ncand * H * W, ~0ull);
    const int64_t stride = pcl_cloud_stride(n);
    // 64 x 64-pixel LDS window (32 KB of 64-bit cells) per 256 consecutive (Morton-ordered) points: a compact surface
    // patch whose splats nearly all land inside the window.  Measured at cfg-2 size, 64 candidates (whole trimming
    // stage): 7.6 ms with 2048 points per block — their patch is wider than the window at close range and the overflow
    // goes to global atomics one splat at a time —, 5.4 / 4.2 / 3.6 ms with 1024 / 512 / 256; other windows at
    // 256-512 points: 64x96 3.8, 48x64 4.5, 48x48 4.0 ms.
    constexpr int TH = 64, TW = 64, PTS = 256;
    hipLaunchKernelGGL((pcl_splat_poses_kernel<TH, TW, PTS>), dim3((unsigned)((n + PTS - 1) / PTS), (unsigned)ncand),
                       dim3(PCL_BLOCK), 0, s, cloud, n, stride, recs, H, W, zbuf);
    hipLaunchKernelGGL(pcl_hist_accum_kernel<0>, dim3(nblk * PCL_HSUB, 1), dim3(PCL_BLOCK), 0, s, (const unsigned long long*)nullptr,
                       cloud, stride, img_hwc, H, W, nsh, nsw, ghist_q);
    hipLaunchKernelGGL(pcl_hist_final_kernel<0>, dim3(nblk, 1), dim3(PCL_BLOCK), 0, s, ghist_q, qhist, nimg, inter, nproj);
    hipLaunchKernelGGL(pcl_hist_accum_kernel<1>, dim3(nblk * PCL_HSUB, ncand), dim3(PCL_BLOCK), 0, s, zbuf, cloud, stride, img_hwc, H, W,
                       nsh, nsw, ghist_c);
    hipLaunchKernelGGL(pcl_hist_final_kernel<1>, dim3(nblk, ncand), dim3(PCL_BLOCK), 0, s, ghist_c, qhist, nimg, inter, nproj);
    PCL_LAUNCH_CHECK();
    return 0;
}

// Scores with the reference's slot semantics.  The reference keeps ONE vector `hist_intersect_split` of nsh * nsw slots for
// all candidates (allocated before the candidate loop, utils.py:539): a block with nothing to histogram — no rendered or no
// query pixel — writes 0 into its slot and `break`s out of its block ROW (utils.py:568-571); the row's remaining slots
// are not touched, so they still hold what the last candidate that got that far left there (possibly itself a stale
// value), NaNs cleaned to 0 in place (utils.py:579).  score = sum of the slots / (nsh * nsw) (utils.py:580).  Pinned by
// G19 (per-candidate slot vectors read out of the running reference function).
// The carry makes the candidates a sequential chain per slot: one block, thread j owns slot j (strided if there are more
// slots than threads) and walks the candidates in order, its carried value in LDS; thread 0 sums each candidate's slots in
// slot order (deterministic).  K is a few dozen candidates, nblk a few dozen slots: microseconds.
__global__ void __launch_bounds__(256) pcl_hist_score_kernel(const float* __restrict__ inter, const int* __restrict__ nproj,
                                                             const int* __restrict__ nimg, int ncand, int nsh, int nsw,
                                                             float* __restrict__ score)
{
    extern __shared__ float slot[];                       // [nblk] current content of hist_intersect_split (middle rows)
    const int nblk = (nsh - 2) * nsw;
    for (int j = threadIdx.x; j < nblk; j += blockDim.x) slot[j] = 0.f;
    __syncthreads();
    for (int cand = 0; cand < ncand; cand++) {
        const int* np = nproj + (int64_t)cand * nblk;
        for (int j = threadIdx.x; j < nblk; j += blockDim.x) {
            const int row0 = (j / nsw) * nsw;
            int first_empty = nsw;                        // position in the row where the reference breaks
            for (int w = 0; w < nsw; w++)
                if (np[row0 + w] == 0 || nimg[row0 + w] == 0) { first_empty = w; break; }
            const int w = j - row0;
            if (w < first_empty) {
                float v = inter[(int64_t)cand * nblk + j];
                slot[j] = (v == v) ? v : 0.f;
            } else if (w == first_empty) slot[j] = 0.f;   // (w > first_empty: untouched, keeps the earlier candidate's value)
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float total = 0.f;
            for (int j = 0; j < nblk; j++) total += slot[j];
            score[cand] = total / (float)(nsh * nsw);
        }
        __syncthreads();
    }
}

extern "C" int pcl_hist_trim_reduce(const float* inter, const int32_t* nproj, const int32_t* nimg, int ncand, int nsh, int nsw, float* score,
                                    void* stream)
{
    if (!inter || !nproj || !nimg || !score || ncand <= 0 || nsh < 3 || nsw < 1) return PCL_EINVAL;
    const size_t lds = (size_t)(nsh - 2) * nsw * sizeof(float);
    if (lds > 60000) return PCL_EINVAL;                   // > 15000 blocks: not a block grid this stage is meant for
    hipLaunchKernelGGL(pcl_hist_score_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, inter, nproj, nimg, ncand, nsh, nsw, score);
    PCL_LAUNCH_CHECK();
    return 0;
}
