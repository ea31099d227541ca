// pcl_hist.hip — the second trimming stage of the initialisation (reference utils.py:510-588, color_utils.py:68-144)
// for a BATCH of candidate poses.  Two renderers feed the same histogram counters (bit-identical results):
//   tile-binned (default, round 2; one pass since round 5): pcl_bin_kernel / pcl_bin_rank_kernel bin every candidate's
//                               points by the 64 x 64-pixel tiles their 3 x 3 splats touch; pcl_tile_resolve_hist_kernel
//                               resolves one tile per workgroup in LDS and histograms the winners — no z-buffer in HBM;
//   z-buffer splat (round 1, kept for workspaces sized without n and as the cross-check):
//     pcl_splat_poses_kernel  : make_pano's z-buffered 3x3 splat (see pcl_ops.hip) for every candidate pose at once,
//                               straight from the packed world-frame cloud (p = R (x - t) computed in the kernel), with
//                               LDS-tiled atomicMin;
//     pcl_hist_accum_kernel<1>: per (candidate, block, pixel range): resolve the z-buffer to colours and histogram the
//                               pixels where both the render and the query are non-black in LDS;
//   then, for both: pcl_hist_accum_kernel<0> (the query image's own non-black pixels), pcl_hist_final_kernel (normalise the
//   query histograms / intersect the candidates' with them), pcl_hist_score_kernel (scores with the reference's slot rule).
// The reference renders one candidate at a time (argsort + nine index_put_ passes) and builds every histogram with a
// dozen tensor ops; the rendered image is never materialised here.
// Only the middle block rows h = 1 .. num_split_h - 2 are used (utils.py:556); block j <-> (h = 1 + j / nsw, w = j % nsw).
#include <stdlib.h>

#include "pcl_device.h"

#define PCL_HBINS 512   // 8 x 8 x 8

// ||p|| with ONE rounding sequence at every call site (splat, bin, fix-up): the depth decides which point owns a pixel
__device__ __forceinline__ float pcl_point_depth(float px, float py, float pz)
{
#pragma clang fp contract(off)
    return sqrtf(px * px + py * py + pz * pz);
}

__device__ inline void pcl_pano_pixel_ref(float px, float py, float pz, int H, int W, int& row, int& col)
{
#pragma clang fp contract(off)
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
            float d = pcl_point_depth(px, py, pz);
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

// ------------------------------------------------------------------------------------------------------------------------
// Tile-binned render + histogram (round 2): no z-buffer in HBM.
// The splat path above pays ~2 global 64-bit atomics per rendered pixel (the LDS windows of neighbouring blocks overlap),
// a 16 MB-per-candidate z-buffer fill and a second pass that reads it back (rocprofv3, profiles/r02/e_init_stage_*: splat 659
// us per 16 candidates with the VALUs 34 % busy, 530 MB of atomic traffic; fill 49 us; accumulate 123 us).  Here every
// candidate's points are first BINNED by the 64 x 64-pixel image tile(s) their 3 x 3 splat touches (grouped by tile inside
// the block with an LDS prefix: one global atomic per (block, tile)); then one workgroup per
// (tile, candidate) resolves its tile completely in LDS with the very same 64-bit priority keys and histograms the winners
// straight into the block histograms.  Same keys, same winners, same integer counts as the splat path (bit-identical scores,
// tests/test_hip_harness.py).  The point lists live where the z-buffer would have been.
#define PCL_TS 64                      // tile edge in pixels
#define PCL_TS_SHIFT 6

// Several query images of one room in one set of launches (round 4): candidate c is scored against image c / cpi.  The float
// images' addresses travel as kernel arguments.
#define PCL_HIST_MAX_IMAGES 32
struct PclImgList { const float* p[PCL_HIST_MAX_IMAGES]; };

struct PclBinArgs {
    const float* cloud;
    int64_t n, stride;
    const PclPoseRec* poses;
    int H, W, ntx, nt;                 // tiles per row, tiles per image
    int ty_lo, ty_hi;                  // tile rows that hold pixels of the scored block rows 1 .. nsh-2 (utils.py:556): the
                                       // tiles above and below are never binned — half the image at num_split_h = 4
    uint32_t* lists;                   // [ncand][3][cap]  per entry: pixel (row << 16 | col), depth bits, packed point slot
    int64_t cap;                       // entries per candidate (4 n: a 3 x 3 splat touches at most four tiles)
    float fast_margin_x, fast_margin_y; // a fast-formula pixel coordinate farther than this from an integer is certain (0: reference formula for all)
    // ONE-PASS binning (round 5).  Rounds 2-4 counted (one projection), scanned and scattered (a second projection; round 5 first cached
    // the count pass's pixels: 8 B per point and candidate written and read back).  Now a block's entries go to ITS OWN region of the
    // candidate's list area (block b: entries [b * 4 PCL_BIN_PTS, (b + 1) * 4 PCL_BIN_PTS) — 4 n in all, the same exact worst case),
    // grouped by tile with a block-local LDS prefix; a tile's list is then a handful of RUNS, one per block that touched it.  Measured
    // A/B in one build (tools/hist_stage_bench.py): 1.23 -> 1.06 ms per 64 candidates at 1M points, 0.42 -> 0.37 per 50 at 167k
    // (count 342 + scan 32 + scatter 268 -> bin 435 + rank 7 us; the resolve kernel pays 492 -> 531 for finding its entries run by run).
    unsigned long long* stat;          // [ncand][nt]      runs << 32 | entries of the tile (one 64-bit atomic per (block, tile))
    uint2* runs;                       // [ncand][nt][nb]  the tile's r-th run: (its first entry in the list area, entries of the tile in runs 0 .. r-1
                                       //                  — the atomic's return value IS the exclusive prefix, in arrival order)
    int4* heads;                       // [ncand][nt]      the resolve launch's order: (tile, runs, entries, 0) by decreasing entries
    int nb;                            // blocks per candidate = ceil(n / PCL_BIN_PTS)
};

// inclusive prefix sum over the lanes of a wave (six ds_bpermute steps; the callers run it once per block)
__device__ __forceinline__ int pcl_wave_scan_incl(int v)
{
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(v, d, 64);
        if (lane >= d) v += u;
    }
    return v;
}

// pixel of packed point i for pose pr, exactly as pcl_splat_poses_kernel computes it
__device__ __forceinline__ void pcl_bin_project(const PclBinArgs& a, const PclPoseRec* __restrict__ pr, int64_t i, int& row, int& col, float& d)
{
    float qx = a.cloud[i] - pr->t[0], qy = a.cloud[a.stride + i] - pr->t[1], qz = a.cloud[2 * a.stride + i] - pr->t[2];
    float px = fmaf(pr->R[2], qz, fmaf(pr->R[1], qy, pr->R[0] * qx));
    float py = fmaf(pr->R[5], qz, fmaf(pr->R[4], qy, pr->R[3] * qx));
    float pz = fmaf(pr->R[8], qz, fmaf(pr->R[7], qy, pr->R[6] * qx));
    pcl_pano_pixel_ref(px, py, pz, a.H, a.W, row, col);
    d = pcl_point_depth(px, py, pz);
}

// The same pixel from the loss kernel's fast atan2 (octant reduction + degree-8 polynomial, v_rcp / v_rsq: ~45 instructions against
// ~170 for the two library atan2f of pcl_pano_pixel_ref), with a CERTIFICATE: the pixel coordinates come out within a few fp32 ulps
// of the reference formula's (angle error <= 6e-7 rad against the library's, the affine chain's roundings: <= 1e-6 W pixels in all),
// so whenever both coordinates are farther than `margin` from an integer — and inside the image — truncation gives the reference's
// pixel.  Returns false otherwise (0.5-1.5 % of the points): those are recomputed with the reference formula (pcl_bin_kernel's fix-up
// queue).  The bit-identity of the whole stage against the z-buffer splat path, which only knows the reference formula, is what
// tests/test_hip_harness.py::test_hist_trim_tile_binned_equals_the_zbuffer_path checks on every scene it runs.
__device__ __forceinline__ bool pcl_pano_pixel_fast(float px, float py, float pz, int H, int W, float margin_x, float margin_y, int& row, int& col)
{
    const float pi = 3.14159265358979323846f, inv_two_pi = 0.15915494309189532f, inv_pi = 0.31830988618379067154f;
    const float rho2 = fmaf(px, px, py * py);
    const float rho = rho2 * __builtin_amdgcn_rsqf(rho2 + 1e-37f);
    const float theta = pcl_atan2_ypos(rho, pz + 1e-6f);
    const float phi = pcl_atan2(py, px + 1e-6f) + pi;
    const float gx = 2.0f * (1.0f - phi * inv_two_pi) - 1.0f, gy = 2.0f * (theta * inv_pi) - 1.0f;
    const float cx = (gx + 1.0f) * 0.5f * (float)(W - 1), cy = (gy + 1.0f) * 0.5f * (float)(H - 1);
    const float fx = cx - floorf(cx), fy = cy - floorf(cy);
    col = (int)cx; row = (int)cy;
    // (a NaN coordinate fails every compare: not certain)
    return fx > margin_x && fx < 1.0f - margin_x && fy > margin_y && fy < 1.0f - margin_y && cx > margin_x && cx < (float)(W - 1) - margin_x &&
           cy > margin_y && cy < (float)(H - 1) - margin_y;
}

// the (up to four) tiles a 3 x 3 splat centred on (row, col) touches, after the clamp to the image
__device__ __forceinline__ void pcl_bin_tiles(int row, int col, int H, int W, int ntx, int ty_lo, int ty_hi, int tiles[4])
{
    // the up to four tiles the 3 x 3 splat of pixel (row, col) touches, -1 for the absent ones: four fixed slots, so that the
    // callers index them with compile-time constants (a count + a runtime-indexed array put the array in scratch memory)
    int ra = max(row - 1, 0) >> PCL_TS_SHIFT, rb = min(row + 1, H - 1) >> PCL_TS_SHIFT;
    int ca = max(col - 1, 0) >> PCL_TS_SHIFT, cb = min(col + 1, W - 1) >> PCL_TS_SHIFT;
    const bool top = ra >= ty_lo && ra <= ty_hi, bot = rb != ra && rb >= ty_lo && rb <= ty_hi, two = cb != ca;
    tiles[0] = top ? ra * ntx + ca : -1;
    tiles[1] = top && two ? ra * ntx + cb : -1;
    tiles[2] = bot ? rb * ntx + ca : -1;
    tiles[3] = bot && two ? rb * ntx + cb : -1;
}

// A block takes PCL_BIN_PTS consecutive (Morton-ordered) points: the zeroing and the prefix of the nt LDS counters are per block, and
// with 256 points per block they were most of the kernel (round 2: 114 / 189 us per 16 candidates at 1M points; a wave-aggregated LDS
// add instead of one atomic per lane did not help: 130 / 213 us).
#define PCL_BIN_PTS 2048
// Pre-dedup (round 4).  Two points of one candidate with the SAME centre pixel splat the same nine cells with the same pass
// priorities, so the farther one loses every one of them: it can be dropped before it is ever listed.  Morton order puts the
// points of a pixel next to each other — a far wall seen from the other end of the room has 16 points per pixel, its tile
// 68 000 list entries (30x the mean: the resolve kernel's tail).  Consecutive lanes hold consecutive Morton points, so a lane
// compares its (pixel, depth) with the lanes up to PCL_BIN_NEIGH places to either side of it in its row of 16 (DPP row shifts: no
// LDS, no barrier) and drops out when one of them has the same pixel and a strictly smaller depth.  (Equal depths: both stay.)
// The surviving keys are untouched: every cell's winner, hence every score,
// is bit-identical to the undeduplicated lists (tests: against the z-buffer splat path, which never dedups).  A first version
// resolved the block's pixels in a 96 x 96 LDS window (exact within the block): resolve 613 -> 454 us per 64 candidates at 1M
// points, but 138 / 128 us MORE in the count / scatter kernels (window fill, two barriers, 40 KB of LDS) — a net loss.
// PCL_BIN_DEDUP=0 turns it off (A/B).
#define PCL_BIN_NEIGH 4
template <int SH>
__device__ __forceinline__ bool pcl_bin_dominated_by(uint32_t npix, uint32_t dep)
{
    // row_shr:SH = 0x110 + SH (lane i reads lane i - SH of its row), row_shl:SH = 0x100 + SH, bound_ctrl: lanes without a source read 0.
    // The pixels travel COMPLEMENTED (npix = ~pix): the 0 of a missing source is then the complement of the "no pixel" marker 0xffffffff,
    // which equals no live lane's pixel (an `old` operand per DPP read instead cost one v_mov each: 128 of the kernel's ~1300 VALU
    // instructions per thread).
    const uint32_t pa = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)npix, 0x110 + SH, 0xf, 0xf, true);
    const uint32_t da = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dep, 0x110 + SH, 0xf, 0xf, true);
    const uint32_t pb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)npix, 0x100 + SH, 0xf, 0xf, true);
    const uint32_t db = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dep, 0x100 + SH, 0xf, 0xf, true);
    return (int)((pa == npix) & (da < dep)) | (int)((pb == npix) & (db < dep));
}

template <bool DEDUP>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_bin_kernel(PclBinArgs a)
{
    constexpr int PER = PCL_BIN_PTS / PCL_BLOCK;
    extern __shared__ int lds[];                       // cnt[nt], base[nt]
    int* cnt = lds;
    int* base = lds + a.nt;
    // the fix-up queue of the fast projection — (k << 8 | thread) of every point whose pixel the fast formula could not certify, and
    // the reference formula's pixel for it
    __shared__ uint16_t fixq[PCL_BIN_PTS];
    __shared__ uint32_t fixpix[PCL_BIN_PTS];
    __shared__ int fixn;
    __shared__ int wave_total[PCL_BLOCK / PCL_WAVE];
    const int cand = blockIdx.y;
    for (int t = threadIdx.x; t < a.nt; t += PCL_BLOCK) cnt[t] = 0;
    if (threadIdx.x == 0) fixn = 0;
    __syncthreads();
    const int64_t first = (int64_t)blockIdx.x * PCL_BIN_PTS + threadIdx.x;
    uint32_t pix[PER], dep[PER];
    const PclPoseRec* __restrict__ pr = a.poses + cand;
    int pend[PER];
    // 32-bit offsets through buffer resources (n < 2^28: a plane of the cloud / of the lists is below 4 GB): the 64-bit address of every
    // load and of three stores per entry was a fifth of the kernel's VALU instructions
    __amdgpu_buffer_rsrc_t cld = __builtin_amdgcn_make_buffer_rsrc((void*)a.cloud, 0, (int)(a.stride * 3 * 4), 0x00020000);
    const int plane = (int)a.stride * 4;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int64_t i = first + (int64_t)k * PCL_BLOCK;
        pix[k] = 0xffffffffu;                          // (row 65535 does not exist: H < 65536)
        dep[k] = 0u;
        pend[k] = -1;
        if (i < a.n) {
            const int voff = (int)i * 4;
            float qx = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cld, voff, 0, 0)) - pr->t[0];
            float qy = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cld, voff, plane, 0)) - pr->t[1];
            float qz = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(cld, voff, 2 * plane, 0)) - pr->t[2];
            float px = fmaf(pr->R[2], qz, fmaf(pr->R[1], qy, pr->R[0] * qx));
            float py = fmaf(pr->R[5], qz, fmaf(pr->R[4], qy, pr->R[3] * qx));
            float pz = fmaf(pr->R[8], qz, fmaf(pr->R[7], qy, pr->R[6] * qx));
            int row, col;
            dep[k] = __float_as_uint(pcl_point_depth(px, py, pz));
            if (a.fast_margin_x > 0.f && pcl_pano_pixel_fast(px, py, pz, a.H, a.W, a.fast_margin_x, a.fast_margin_y, row, col)) {
                pix[k] = ((uint32_t)row << 16) | (uint32_t)col;
            } else {
                pend[k] = atomicAdd(&fixn, 1);
                fixq[pend[k]] = (uint16_t)((k << 8) | threadIdx.x);
            }
        }
    }
    __syncthreads();
    const int nfix = fixn;
    for (int e = threadIdx.x; e < nfix; e += PCL_BLOCK) {
        const int k = fixq[e] >> 8, owner = fixq[e] & 255;
        const int64_t i = (int64_t)blockIdx.x * PCL_BIN_PTS + owner + (int64_t)k * PCL_BLOCK;
        int row, col;
        float d;
        pcl_bin_project(a, pr, i, row, col, d);        // the reference formula (two library atan2f)
        fixpix[e] = ((uint32_t)row << 16) | (uint32_t)col;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++)
        if (pend[k] >= 0) pix[k] = fixpix[pend[k]];
    if (DEDUP) {
#pragma unroll
        for (int k = 0; k < PER; k++) {
            // (`|`, not `||`: every lane takes part in every DPP read — a short-circuit would switch source lanes off)
            const uint32_t npix = ~pix[k];             // (a lane without a pixel sends 0: it may "match" a missing source — and stays without a pixel)
            int dom = (int)pcl_bin_dominated_by<1>(npix, dep[k]) | (int)pcl_bin_dominated_by<2>(npix, dep[k]);
            if (PCL_BIN_NEIGH >= 3) dom |= (int)pcl_bin_dominated_by<3>(npix, dep[k]);
            if (PCL_BIN_NEIGH >= 4) dom |= (int)pcl_bin_dominated_by<4>(npix, dep[k]);
            if (dom) pix[k] = 0xffffffffu;             // a nearer point owns this pixel (the compares above all saw the originals:
        }                                              // k is a different point set per trip)
    }
    int tiles[PER][4];
#pragma unroll
    for (int k = 0; k < PER; k++) {
#pragma unroll
        for (int j = 0; j < 4; j++) tiles[k][j] = -1;
        if (pix[k] != 0xffffffffu) {
            pcl_bin_tiles((int)(pix[k] >> 16), (int)(pix[k] & 0xffffu), a.H, a.W, a.ntx, a.ty_lo, a.ty_hi, tiles[k]);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (tiles[k][j] >= 0) atomicAdd(&cnt[tiles[k][j]], 1);
        }
    }
    __syncthreads();
    // block-local exclusive prefix of the tile counters: thread i owns counters [i per_thread, (i + 1) per_thread)
    const int per_thread = (a.nt + PCL_BLOCK - 1) / PCL_BLOCK, t_lo = (int)threadIdx.x * per_thread, t_hi = min(t_lo + per_thread, a.nt);
    int s = 0;
    for (int t = t_lo; t < t_hi; t++) s += cnt[t];
    const int incl = pcl_wave_scan_incl(s);
    if ((threadIdx.x & 63) == 63) wave_total[threadIdx.x >> 6] = incl;
    __syncthreads();
    int run = incl - s;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) run += wave_total[w];
    for (int t = t_lo; t < t_hi; t++) {
        const int c = cnt[t];
        base[t] = run + (int)blockIdx.x * (4 * PCL_BIN_PTS);   // first entry of this (block, tile) run inside the candidate's list area
        run += c;
        cnt[t] = 0;                                    // becomes the run's cursor
    }
    __syncthreads();
    uint32_t* list = a.lists + (int64_t)cand * 3 * a.cap;
    __amdgpu_buffer_rsrc_t l_pix = __builtin_amdgcn_make_buffer_rsrc((void*)list, 0, (int)(a.cap * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t l_dep = __builtin_amdgcn_make_buffer_rsrc((void*)(list + a.cap), 0, (int)(a.cap * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t l_id = __builtin_amdgcn_make_buffer_rsrc((void*)(list + 2 * a.cap), 0, (int)(a.cap * 4), 0x00020000);
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t i = (uint32_t)(first + (int64_t)k * PCL_BLOCK);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int t = tiles[k][j];
            if (t < 0) continue;
            const int pos = (base[t] + atomicAdd(&cnt[t], 1)) * 4;
            // the projection travels with the entry: the resolve kernel reads 12 coalesced bytes per entry instead of chasing
            // slot -> x, y, z and projecting again
            __builtin_amdgcn_raw_buffer_store_b32(pix[k], l_pix, pos, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(dep[k], l_dep, pos, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(i, l_id, pos, 0, 0);
        }
    }
    // publish the runs LAST: the atomics' round trips then delay nobody (the counters hold the run lengths again once every entry
    // has been written)
    __syncthreads();
    unsigned long long* st = a.stat + (int64_t)cand * a.nt;
    uint2* runs = a.runs + (int64_t)cand * a.nt * a.nb;
    for (int t = t_lo; t < t_hi; t++) {
        const int c = cnt[t];
        if (!c) continue;
        const unsigned long long old = atomicAdd(&st[t], (1ull << 32) | (unsigned long long)c);
        runs[(int64_t)t * a.nb + (int64_t)(old >> 32)] = make_uint2((uint32_t)base[t], (uint32_t)old);   // (first entry, tile entries before it)
    }
}

// The resolve launch's order for one candidate — tiles by decreasing entry count (ties by index), each with its run count.
// A block ranks 64 tiles, each wave against a quarter of the counts (a block per candidate with every thread walking all nt counts was
// 29 us of LDS latency for 512 tiles, as much as the scan it replaced).
__global__ void __launch_bounds__(PCL_BLOCK) pcl_bin_rank_kernel(PclBinArgs a)
{
    extern __shared__ int cl[];                        // [nt rounded up to 4] counts, then [4][64] partial ranks
    const int cand = blockIdx.x, nt4 = (a.nt + 3) & ~3;
    int* part = cl + nt4;
    const uint2* __restrict__ st = (const uint2*)(a.stat + (int64_t)cand * a.nt);      // .x = entries, .y = runs
    for (int u = threadIdx.x; u < nt4; u += PCL_BLOCK) cl[u] = u < a.nt ? (int)st[u].x : -1;      // (-1: ranks behind every tile)
    __syncthreads();
    const int t = (int)blockIdx.y * 64 + (int)(threadIdx.x & 63), seg = (int)(threadIdx.x >> 6);
    const int per = ((nt4 >> 2) + 3) / 4 * 4;          // counts per wave, a multiple of four
    const int u_lo = min(seg * per, nt4), u_hi = min(u_lo + per, nt4);
    const int c = t < a.nt ? cl[t] : 0;
    int rank = 0;
#pragma unroll 4
    for (int u = u_lo; u < u_hi; u += 4) {
        const int v0 = cl[u], v1 = cl[u + 1], v2 = cl[u + 2], v3 = cl[u + 3];
        rank += (v0 > c || (v0 == c && u < t)) ? 1 : 0;
        rank += (v1 > c || (v1 == c && u + 1 < t)) ? 1 : 0;
        rank += (v2 > c || (v2 == c && u + 2 < t)) ? 1 : 0;
        rank += (v3 > c || (v3 == c && u + 3 < t)) ? 1 : 0;
    }
    part[threadIdx.x] = rank;
    __syncthreads();
    if (seg == 0 && t < a.nt) {
        rank = part[threadIdx.x] + part[64 + threadIdx.x] + part[128 + threadIdx.x] + part[192 + threadIdx.x];
        a.heads[(int64_t)cand * a.nt + rank] = make_int4(t, (int)st[t].y, c, 0);
    }
}

// One workgroup per (tile, candidate): resolve the tile in LDS (same keys as the splat path), then histogram the winners of
// the pixels where the query image is not black into the block histograms (LDS for the up to 2 x 2 histogram blocks a tile
// overlaps; tiny blocks — more than that per tile — go to the global counters directly).
// (1024 threads: the heaviest tile of a candidate — up to 20x the mean — sets the duration of the launch)
// (1024 threads: the heaviest tile of a candidate — up to 20x the mean — sets the duration of the launch.  Round 4 tried 256-thread
//  workgroups for sparse clouds, eight resident per CU instead of two: slower at every shape, PCL_RESOLVE_THREADS=256 keeps the A/B)
#ifdef PCL_BLOCK_TRACE                                 // experiments: tools/block_trace.py-style end stamps (see pcl_loss.hip)
__device__ unsigned long long* pcl_hist_trace_buf = nullptr;
extern "C" int pcl_debug_set_hist_trace(unsigned long long* buf)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(pcl_hist_trace_buf), &buf, sizeof(buf));
}
#endif
template <int PCL_RESOLVE_THREADS>
__global__ void __launch_bounds__(PCL_RESOLVE_THREADS) pcl_tile_resolve_hist_kernel(PclBinArgs a, const uint8_t* __restrict__ qmask, const uint16_t* __restrict__ codes, int cpi,
                                                                          int nsh, int nsw, unsigned int* __restrict__ ghist)
{
    const uint8_t* __restrict__ qm = qmask + (int64_t)((int)blockIdx.x / cpi) * a.H * a.W;      // (blockIdx.x = candidate)
    // the tile with a halo of two pixels: every splat pixel of every listed entry has a cell (an entry is listed when its 3 x 3
    // splat touches the tile, so its centre is at most one pixel outside), and the nine writes need no membership test
    constexpr int TW = PCL_TS + 4;
    __shared__ unsigned long long tile[TW * TW];
    __shared__ unsigned int hist[4][PCL_HBINS];
    // candidate fastest: workgroups are handed out x first, so rank 0 — the heaviest tile — of EVERY candidate starts before
    // any rank-1 tile (tile-major launches started the last candidate's heaviest tile at 94 % of the launch)
    const int cand = blockIdx.x;
    const int4 hd = a.heads[(int64_t)cand * a.nt + blockIdx.y];        // heaviest tiles first
    const int t = hd.x, nruns = hd.y, total = hd.z;
#ifdef PCL_BLOCK_TRACE
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int ty = t / a.ntx, tx = t - ty * a.ntx;
    const int drow[9] = {0, 0, -1, -1, -1, 1, 1, 1, 0};   // pass order idx8,7,6,5,4,3,2,1,centre (utils.py:173-198)
    const int dcol[9] = {-1, 1, -1, 0, 1, -1, 0, 1, 0};
    const uint32_t* list = a.lists + (int64_t)cand * 3 * a.cap;
    if (total == 0) return;                            // nothing projects here (or the tile is outside the scored rows)
    // The tile's entries are `nruns` runs (one per bin block that touched the tile) somewhere in the candidate's list
    // area.  Entry e of the tile lives at run_first[r] + (e - run_pre[r]) for the last run r with run_pre[r] <= e; the runs' (first
    // entry, entries of the tile before the run) are staged in LDS, every lane finds its run by bisection.  The first PCL_RESOLVE_THREADS
    // runs are requested here, before the tile is initialised (their latency hides behind it); a tile fed by more blocks than that
    // goes through in batches, e0 .. e1 = the entries of one batch.
    __shared__ uint32_t run_first[PCL_RESOLVE_THREADS], run_pre[PCL_RESOLVE_THREADS];
    // LDS budget: tile 36 992 + histograms 8 192 + slab 12 288 + run tables 8 192 = 65 664 bytes for 1024 threads — above the
    // 64 KiB of older parts ON PURPOSE: gfx950 has 160 KB per CU and the launch wants exactly two of these workgroups per CU (16 waves each,
    // 58 VGPRs); the run tables are live together with the slab during the walk, so nothing here can alias.  gfx950 only, like the library.
    static_assert(sizeof(unsigned long long) * TW * TW + 4 * PCL_HBINS * 4 + 5 * PCL_RESOLVE_THREADS * 4 <= 80 * 1024,
                  "two resolve workgroups per CU: at most 80 KB of LDS each (gfx950: 160 KB per CU)");
    const uint2* runs = a.runs + ((int64_t)cand * a.nt + t) * a.nb;
    uint2 run0 = make_uint2(0u, 0u);
    if ((int)threadIdx.x < nruns) run0 = runs[threadIdx.x];
    // ... and so are this thread's query-mask bytes: which of its pixels can count at all (inside the image, in a scored block row, query
    // pixel not black) does not depend on the render — one bit per pixel, loaded while the list is walked instead of in front of the
    // dependent colour gather of the histogram phase
    const int bh = a.H / nsh, bw = a.W / nsw, nblk = (nsh - 2) * nsw;
    const int h_lo = (ty * PCL_TS) / bh, w_lo = (tx * PCL_TS) / bw;       // first histogram block row / column of this tile
    const bool big_blocks = bh >= PCL_TS && bw >= PCL_TS;
    constexpr int NPIX = PCL_TS * PCL_TS / PCL_RESOLVE_THREADS;
    uint8_t qbyte[NPIX];
#pragma unroll
    for (int k = 0; k < NPIX; k++) {
        const int i = (int)threadIdx.x + k * PCL_RESOLVE_THREADS;
        const int r = ty * PCL_TS + (i >> PCL_TS_SHIFT), c = tx * PCL_TS + (i & (PCL_TS - 1));
        // (blocks at least a tile wide and high — every shipped config — span at most two block rows / columns per tile: a compare
        //  instead of two integer divisions per pixel)
        const int h = big_blocks ? h_lo + (r >= (h_lo + 1) * bh ? 1 : 0) : r / bh;
        const int w = big_blocks ? w_lo + (c >= (w_lo + 1) * bw ? 1 : 0) : c / bw;
        const bool scored = r < a.H && c < a.W && h >= 1 && h <= nsh - 2 && w < nsw;      // only the middle block rows (utils.py:556)
        qbyte[k] = scored ? qm[(int64_t)r * a.W + c] : (uint8_t)0;                        // 0: query pixel black (mask written with the query histograms)
    }
    const int r_org = ty * PCL_TS - 2, c_org = tx * PCL_TS - 2;
    for (int i = threadIdx.x; i < TW * TW; i += PCL_RESOLVE_THREADS) tile[i] = ~0ull;
    for (int i = threadIdx.x; i < 4 * PCL_HBINS; i += PCL_RESOLVE_THREADS) (&hist[0][0])[i] = 0u;
    run_first[threadIdx.x] = run0.x; run_pre[threadIdx.x] = run0.y;
    __syncthreads();
    // The list is walked in slabs of PCL_RESOLVE_THREADS entries staged through LDS: the loads are coalesced (lane = entry),
    // but the entries of a list are Morton neighbours — 64 consecutive ones land on a handful of pixels, and LDS atomics of
    // one wave on one address serialise.  Each lane therefore takes entry (17 tid) mod 1024 of the slab: a wave's lanes hold
    // entries from all over the slab (stride 17: no bank conflicts on the way out either).  The next slab's loads are in flight
    // while the current one is resolved.  Per-workgroup timeline (tools/hist_trace.py) of a 4000-16000-entry list: walk 21.4 ->
    // 17.6 us; with the atomics compiled out 13.7, with the tile reads out as well 9.9 — over half of the walk is the list
    // itself arriving from memory (12 B per entry, 2.5 TB/s over the whole launch).
    __shared__ uint32_t slab[3][PCL_RESOLVE_THREADS];
    const int nbatch = (nruns + PCL_RESOLVE_THREADS - 1) / PCL_RESOLVE_THREADS;
    const int j = ((int)threadIdx.x * 17) & (PCL_RESOLVE_THREADS - 1);
  for (int batch = 0; batch < nbatch; batch++) {
    const int r0 = batch * PCL_RESOLVE_THREADS, nr = min(PCL_RESOLVE_THREADS, nruns - r0);
    if (batch > 0) {                                   // (the previous batch's last slab trip ended with a barrier: the tables are free)
        if ((int)threadIdx.x < nr) { const uint2 r = runs[r0 + (int)threadIdx.x]; run_first[threadIdx.x] = r.x; run_pre[threadIdx.x] = r.y; }
        __syncthreads();
    }
    const int e0 = (int)run_pre[0], e1 = r0 + nr < nruns ? (int)runs[r0 + nr].y : total;
    auto entry_at = [&](int e) -> int64_t {
        int lo = 0, hi = nr;                           // the last run that starts at or before e
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if ((int)run_pre[mid] <= e) lo = mid; else hi = mid;
        }
        return (int64_t)run_first[lo] + (e - (int)run_pre[lo]);
    };
    uint32_t npix = 0xffffffffu, ndep = 0u, nid = 0u;
    {
        const int e = e0 + (int)threadIdx.x;
        if (e < e1) { const int64_t q = entry_at(e); npix = list[q]; ndep = list[a.cap + q]; nid = list[2 * a.cap + q]; }
    }
    for (int base = e0; base < e1; base += PCL_RESOLVE_THREADS) {
        slab[0][threadIdx.x] = npix; slab[1][threadIdx.x] = ndep; slab[2][threadIdx.x] = nid;
        __syncthreads();
        {
            const int e = base + PCL_RESOLVE_THREADS + (int)threadIdx.x;
            npix = 0xffffffffu;
            if (e < e1) { const int64_t q = entry_at(e); npix = list[q]; ndep = list[a.cap + q]; nid = list[2 * a.cap + q]; }
        }
        const uint32_t pix = slab[0][j], dbits = slab[1][j], i = slab[2][j];
        if (pix != 0xffffffffu) {
            const int row = (int)(pix >> 16), col = (int)(pix & 0xffffu);
            const unsigned long long bk = ((unsigned long long)dbits << 29) | (unsigned long long)(0x1fffffffu - i);
            // clamped at the image border as the reference's index arithmetic is (utils.py:173-198), then tile-local
            const int r3[3] = {(max(row - 1, 0) - r_org) * TW, (row - r_org) * TW, (min(row + 1, a.H - 1) - r_org) * TW};
            const int c3[3] = {max(col - 1, 0) - c_org, col - c_org, min(col + 1, a.W - 1) - c_org};
            // A plain read first: cells only ever decrease, so a key that does not beat what is already there can be dropped
            // without the atomic (in the dense tiles — a far wall seen from the other end of the room — almost every key
            // loses).  All nine reads are issued before the first compare: one LDS round trip per entry instead of nine
            // (a stale value only costs an atomic that loses).
            unsigned long long cur[9];
#pragma unroll
            // (relaxed atomic loads, not volatile ones: a volatile read of LDS compiles to a system-coherent FLAT load with a
            // full wait behind it — nine serial round trips through the memory pipeline per entry)
            for (int p = 0; p < 9; p++)
                cur[p] = __hip_atomic_load(&tile[r3[drow[p] + 1] + c3[dcol[p] + 1]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int p = 0; p < 9; p++) {
                const unsigned long long key = ((unsigned long long)(8 - p) << 60) | bk;
                if (key < cur[p]) atomicMin(&tile[r3[drow[p] + 1] + c3[dcol[p] + 1]], key);
            }
        }
        __syncthreads();                               // the slab is overwritten at the top of the next trip
    }
  }
    __syncthreads();
#ifdef PCL_BLOCK_TRACE
    const unsigned long long trace_t1 = __builtin_amdgcn_s_memrealtime();
#endif
    // Winners -> histograms.  A fixed cost per (tile, candidate): 4.3 us of a typical 8 us workgroup at the shipped shape (167k points:
    // 9 300 non-empty workgroups of ~500 entries, tools/hist_trace.py).  Round 4 tried, A/B on one box, and dropped: the pixels in
    // rounds of four with three stages (cells + query pixels, then colours, then atomics: four independent loads in flight per
    // stage) — 0.434 -> 0.460 ms for the stage at 167k x 50, 1.35 -> 1.41 at 1M x 64; wave-voted histogram adds (one LDS atomic
    // per distinct colour code of a wave instead of 64 on one address) — 0.433 -> 0.490 / 1.34 -> 1.40.  Neither the load latency
    // nor the same-address atomics are what the loop waits for.
    unsigned int* g = ghist + (int64_t)cand * nblk * PCL_HBINS;
#pragma unroll
    for (int kk = 0; kk < NPIX; kk++) {
        if (!qbyte[kk]) continue;
        const int i = (int)threadIdx.x + kk * PCL_RESOLVE_THREADS;
        const unsigned long long k = tile[((i >> PCL_TS_SHIFT) + 2) * TW + (i & (PCL_TS - 1)) + 2];
        if (k == ~0ull) continue;
        const int r = ty * PCL_TS + (i >> PCL_TS_SHIFT), c = tx * PCL_TS + (i & (PCL_TS - 1));
        const int h = big_blocks ? h_lo + (r >= (h_lo + 1) * bh ? 1 : 0) : r / bh;
        const int w = big_blocks ? w_lo + (c >= (w_lo + 1) * bw ? 1 : 0) : c / bw;
        const int64_t j = (int64_t)(0x1fffffffu - (uint32_t)(k & 0x1fffffffull));
        const int code = (int)codes[j];                                  // the winner's 8 x 8 x 8 colour code (pcl_hist_codes_kernel)
        if (code == 0xffff) continue;                                    // its colour is exactly black
        const int blk = (h - 1) * nsw + w;
        const int sh = h - h_lo, sw = w - w_lo;
        if (sh < 2 && sw < 2) atomicAdd(&hist[sh * 2 + sw][code], 1u);
        else atomicAdd(&g[(int64_t)blk * PCL_HBINS + code], 1u);
    }
    __syncthreads();
#ifdef PCL_BLOCK_TRACE
    const unsigned long long trace_t2 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int i = threadIdx.x; i < 4 * PCL_HBINS; i += PCL_RESOLVE_THREADS) {
        const unsigned int v = (&hist[0][0])[i];
        if (!v) continue;
        const int slot = i / PCL_HBINS, code = i - slot * PCL_HBINS;
        const int h = h_lo + (slot >> 1), w = w_lo + (slot & 1);
        if (h < 1 || h > nsh - 2 || w >= nsw) continue;                  // (cannot hold counts: guarded when accumulated)
        atomicAdd(&g[(int64_t)((h - 1) * nsw + w) * PCL_HBINS + code], v);
    }
#ifdef PCL_BLOCK_TRACE
    if (pcl_hist_trace_buf && threadIdx.x == 0) {
        unsigned long long* tb = pcl_hist_trace_buf + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 5;
        tb[0] = __builtin_amdgcn_s_memrealtime();
        tb[1] = (unsigned long long)total;
        tb[2] = trace_t0;
        tb[3] = trace_t1;
        tb[4] = trace_t2;
    }
#endif
}


// Histograms in two steps so that a handful of image blocks still fills the chip: every (block, candidate) is cut into
// PCL_HSUB pixel ranges, each range is histogrammed in LDS by its own workgroup and its non-empty bins are added to a
// global counter array (integer atomics: deterministic); a finalise kernel then normalises / intersects.
#define PCL_HSUB 16
#define PCL_HSUB_QUERY 64       // the query image's own histograms: 8 blocks x 64 ranges = 512 workgroups (16 ranges: 128 workgroups walked the
                                // 25 MB image at 1.5 TB/s, 16 us per call; round 6)

// MODE 0: query image (zbuf unused, cand = 0)   MODE 1: candidate renders
template <int MODE>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_hist_accum_kernel(const unsigned long long* __restrict__ zbuf, const float* __restrict__ cloud,
                                                                   int64_t stride, PclImgList imgs, int cpi, int H, int W, int nsh,
                                                                   int nsw, unsigned int* __restrict__ ghist, uint8_t* __restrict__ qmask)
{
    // MODE 0: blockIdx.y = query image (its own histograms);  MODE 1: blockIdx.y = candidate, scored against image cand / cpi
    const float* __restrict__ img = imgs.p[MODE == 0 ? (int)blockIdx.y : (int)blockIdx.y / cpi];
    uint8_t* __restrict__ qmask_out = (MODE == 0 && qmask) ? qmask + (int64_t)blockIdx.y * H * W : nullptr;
    __shared__ unsigned int hist[PCL_HBINS];
    constexpr int HSUB = MODE == 0 ? PCL_HSUB_QUERY : PCL_HSUB;
    const int blk = blockIdx.x / HSUB, sub = blockIdx.x - blk * HSUB, cand = blockIdx.y, nblk = gridDim.x / HSUB;
    const int bh = H / nsh, bw = W / nsw;
    const int h = 1 + blk / nsw, w = blk - (h - 1) * nsw;
    for (int i = threadIdx.x; i < PCL_HBINS; i += PCL_BLOCK) hist[i] = 0u;
    __syncthreads();
    const unsigned long long* zb = MODE == 1 ? zbuf + (int64_t)cand * H * W : nullptr;
    const int total = bh * bw, per = (total + HSUB - 1) / HSUB;
    const int lo = sub * per, hi = min(lo + per, total);
    for (int idx = lo + threadIdx.x; idx < hi; idx += PCL_BLOCK) {
        int r = h * bh + idx / bw, c = w * bw + idx % bw;
        int64_t pix = (int64_t)r * W + c;
        float q0 = img[3 * pix], q1 = img[3 * pix + 1], q2 = img[3 * pix + 2];
        bool qm = !(q0 == 0.f && q1 == 0.f && q2 == 0.f);            // query pixel not black
        if (MODE == 0) {
            if (qmask_out) qmask_out[pix] = qm ? 1 : 0;              // what the resolve kernel asks of the query image: one byte, not 12
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

// inter[cand][blk] = sum min(h / h.sum(), q / q.sum()) (torch.min(h1, h2).sum(), color_utils.py:122-144), nproj[cand][blk] = h.sum(),
// nimg[image][blk] = q.sum(); h = the candidate's block histogram, q = the block histogram of its query image (image cand / cpi),
// both as integer counts.  (Rounds 1-5 normalised the query histograms in a launch of their own; the same divisions are made here,
// per candidate — 512 more per block, one launch less per call.)
__global__ void __launch_bounds__(PCL_BLOCK) pcl_hist_final_kernel(const unsigned int* __restrict__ ghist_c, const unsigned int* __restrict__ ghist_q,
                                                                   int* __restrict__ nimg, float* __restrict__ inter, int* __restrict__ nproj, int cpi)
{
    __shared__ float red[2][PCL_BLOCK / PCL_WAVE];
    const int blk = blockIdx.x, cand = blockIdx.y, nblk = gridDim.x, image = cand / cpi;
    const unsigned int* g = ghist_c + ((int64_t)cand * nblk + blk) * PCL_HBINS;
    const unsigned int* q = ghist_q + ((int64_t)image * nblk + blk) * PCL_HBINS;
    const unsigned int c0 = g[threadIdx.x], c1 = g[threadIdx.x + PCL_BLOCK], q0 = q[threadIdx.x], q1 = q[threadIdx.x + PCL_BLOCK];
    const float s = pcl_wave_sum((float)(c0 + c1)), sq = pcl_wave_sum((float)(q0 + q1));
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = sq; }
    __syncthreads();
    const float total = red[0][0] + red[0][1] + red[0][2] + red[0][3], qtotal = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    __syncthreads();
    // hist / hist.sum() on both sides (0 / 0 = NaN for an empty query block: fminf then returns the candidate's value, as torch.min
    // of the reference's NaN-free path never sees it — such a block is skipped by the slot rule, pcl_hist_score_kernel)
    float v = fminf((float)c0 / total, (float)q0 / qtotal) + fminf((float)c1 / total, (float)q1 / qtotal);
    if (!(total > 0.f)) v = 0.f;
    v = pcl_wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        inter[(int64_t)cand * nblk + blk] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        nproj[(int64_t)cand * nblk + blk] = (int)total;
        if (cand == image * cpi) nimg[(int64_t)image * nblk + blk] = (int)qtotal;
    }
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_fill_u64b_kernel(unsigned long long* p, int64_t n, unsigned long long v)
{
    for (int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PCL_BLOCK) p[i] = v;
}

// Everything the stage needs before its first real kernel, in ONE launch (rounds 1-5: a memset, a pose-setup launch, a second memset and
// the colour-code launch — 20 us of four dependent 5 us launches per image at the shipped shape): the histogram counters and the tiles'
// statistics zeroed, the candidates' pose records, and the 8 x 8 x 8 colour code of every packed point (tile-binned path: stat / codes
// non-null).  Grid-stride sections; every word is written by exactly one thread.
__global__ void __launch_bounds__(PCL_BLOCK) pcl_hist_prepare_kernel(const float* __restrict__ trans, const float* __restrict__ rot, int ncand, PclPoseRec* recs,
                                                                     unsigned int* __restrict__ ghist, int64_t ghist_words, unsigned long long* __restrict__ stat,
                                                                     int64_t stat_words, const float* __restrict__ cloud, int64_t n, int64_t stride,
                                                                     uint16_t* __restrict__ codes)
{
    const int64_t tid = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x, nth = (int64_t)gridDim.x * PCL_BLOCK;
    if (tid < ncand) {
        const int b = (int)tid;
        float p[6] = {trans[3 * b], trans[3 * b + 1], trans[3 * b + 2], rot[3 * b], rot[3 * b + 1], rot[3 * b + 2]};
        pcl_write_pose_rec(&recs[b], p);
    }
    pcl_i4* g4 = (pcl_i4*)ghist;                                   // (the counter area is a whole number of 16-byte words: 512 bins per block)
    const pcl_i4 z4 = {0, 0, 0, 0};
    for (int64_t i = tid; i < (ghist_words >> 2); i += nth) g4[i] = z4;
    if (stat)
        for (int64_t i = tid; i < stat_words; i += nth) stat[i] = 0ull;
    if (codes)
        for (int64_t j = tid; j < n; j += nth) {
            // image * 255 (utils.py:200); the packed cloud holds -rgb in planes 3..5
            const float p0 = -cloud[3 * stride + j] * 255.f, p1 = -cloud[4 * stride + j] * 255.f, p2 = -cloud[5 * stride + j] * 255.f;
            codes[j] = (p0 == 0.f && p1 == 0.f && p2 == 0.f) ? (uint16_t)0xffffu : (uint16_t)pcl_hist_code(p0, p1, p2);
        }
}

static size_t hist_align(size_t v) { return (v + 255) & ~(size_t)255; }

// can the tile-binned render be used at all for this cloud / panorama (16-bit pixel fields, 28-bit slots, <= 4096 tiles)?
static bool hist_binned_ok(int64_t n, int H, int W)
{
    const int64_t nt = (int64_t)((W + PCL_TS - 1) / PCL_TS) * ((H + PCL_TS - 1) / PCL_TS);
    return nt <= 4096 && H < 65536 && W < 65536 && n < ((int64_t)1 << 28);
}

// bytes of the render area per candidate: the z-buffer of the splat path, or — when n is given — the larger of that and the
// tile-binned path's bookkeeping (the run tables: nt n / 256 bytes — 4 % of the lists for a 2048 x 1024 panorama) + point lists (4 n
// entries of 12 bytes: the exact worst case, nothing can overflow)
static size_t hist_render_bytes(int64_t n, int H, int W)
{
    size_t zb = (size_t)H * W * 8;
    if (n <= 0) return zb;
    const size_t nt = (size_t)((W + PCL_TS - 1) / PCL_TS) * ((H + PCL_TS - 1) / PCL_TS);
    if (!hist_binned_ok(n, H, W)) return zb;       // the launch would take the splat path anyway: no lists to hold
    const size_t nb = (size_t)((n + PCL_BIN_PTS - 1) / PCL_BIN_PTS);
    size_t binned = nt * (16 + 8 + nb * 8) + (size_t)4 * n * 12;      // launch order, tile statistics, run tables + lists
    return binned > zb ? binned : zb;
}

static size_t hist_workspace_bytes(int64_t n, int ncand, int H, int W, int nsh, int nsw, int nimages = 1)
{
    if (ncand <= 0 || nimages <= 0 || H <= 0 || W <= 0 || nsh < 3 || nsw < 1) return 0;
    const size_t nblk = (size_t)(nsh - 2) * nsw;
    return hist_align((size_t)ncand * sizeof(PclPoseRec)) + hist_align((size_t)ncand * hist_render_bytes(n, H, W)) +
           hist_align((size_t)nimages * nblk * PCL_HBINS * sizeof(float)) +
           hist_align((size_t)(ncand + nimages) * nblk * PCL_HBINS * sizeof(unsigned int)) +
           (n > 0 ? hist_align((size_t)nimages * H * W) + hist_align((size_t)n * sizeof(uint16_t)) : 0);     // query masks, colour codes
}

extern "C" size_t pcl_hist_trim_workspace_bytes_n(int64_t n, int ncand, int H, int W, int nsh, int nsw)
{
    return n > 0 ? hist_workspace_bytes(n, ncand, H, W, nsh, nsw) : 0;
}

extern "C" size_t pcl_hist_trim_workspace_bytes(int ncand, int H, int W, int nsh, int nsw)
{
    return hist_workspace_bytes(0, ncand, H, W, nsh, nsw);
}

extern "C" size_t pcl_hist_trim_images_workspace_bytes(int64_t n, int nimages, int cand_per_image, int H, int W, int nsh, int nsw)
{
    if (n <= 0 || nimages <= 0 || nimages > PCL_HIST_MAX_IMAGES || cand_per_image <= 0) return 0;
    return hist_workspace_bytes(n, nimages * cand_per_image, H, W, nsh, nsw, nimages);
}

// candidates [i * cand_per_image, (i + 1) * cand_per_image) are scored against imgs_host[i]; nimg [nimages][nblk]
extern "C" int pcl_hist_trim_scores_images(const float* cloud, int64_t n, const float* const* imgs_host, int nimages, int cand_per_image, int H,
                                           int W, const float* trans, const float* rot, int nsh, int nsw, float* inter, int* nproj,
                                           int* nimg, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!cloud || !imgs_host || !trans || !rot || !inter || !nproj || !nimg || !workspace) return PCL_EINVAL;
    if (nimages <= 0 || nimages > PCL_HIST_MAX_IMAGES || cand_per_image <= 0) return PCL_EINVAL;
    const int ncand = nimages * cand_per_image, cpi = cand_per_image;
    PclImgList imgs;
    for (int i = 0; i < PCL_HIST_MAX_IMAGES; i++) {
        imgs.p[i] = i < nimages ? imgs_host[i] : nullptr;
        if (i < nimages && !imgs_host[i]) return PCL_EINVAL;
    }
    if (n <= 0 || n > 0x1fffffffll || ncand > 65535 || H <= 0 || W <= 0 || nsh < 3 || nsw < 1) return PCL_EINVAL;
    if (H / nsh <= 0 || W / nsw <= 0) return PCL_EINVAL;
    if (workspace_bytes < hist_workspace_bytes(0, ncand, H, W, nsh, nsw, nimages)) return PCL_EWORKSPACE;
    // a workspace sized with n (pcl_hist_trim_workspace_bytes_n / pcl_hist_trim_images_workspace_bytes) selects the tile-binned
    // path, the smaller one of pcl_hist_trim_workspace_bytes(...) the z-buffer splat
    const bool roomy = workspace_bytes >= hist_workspace_bytes(n, ncand, H, W, nsh, nsw, nimages);
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    PclPoseRec* recs = (PclPoseRec*)ws;
    ws += hist_align((size_t)ncand * sizeof(PclPoseRec));
    unsigned long long* zbuf = (unsigned long long*)ws;
    ws += hist_align((size_t)ncand * hist_render_bytes(roomy ? n : 0, H, W));
    const int nblk = (nsh - 2) * nsw;
    ws += hist_align((size_t)nimages * nblk * PCL_HBINS * sizeof(float));       // (rounds 1-5: the normalised query histograms; the layout is kept)
    unsigned int* ghist_q = (unsigned int*)ws;                       // [nimages][nblk][512], then [ncand][nblk][512]
    unsigned int* ghist_c = ghist_q + (size_t)nimages * nblk * PCL_HBINS;
    ws += hist_align((size_t)(ncand + nimages) * nblk * PCL_HBINS * sizeof(unsigned int));
    uint8_t* qmask = roomy ? (uint8_t*)ws : nullptr;                 // [nimages][H * W] (tile-binned path only)
    uint16_t* codes = roomy ? (uint16_t*)(ws + hist_align((size_t)nimages * H * W)) : nullptr;
    const int64_t stride = pcl_cloud_stride(n);
    const int ntx = (W + PCL_TS - 1) / PCL_TS, nty = (H + PCL_TS - 1) / PCL_TS, nt = ntx * nty;
    const bool binned = roomy && hist_binned_ok(n, H, W);
    // layout of the tile-binned render area: [ncand] x heads[nt] (16 B), [ncand] x stat[nt] (8 B), [ncand] x runs[nt][nb] (8 B), [ncand] x lists[3][cap]
    unsigned long long* stat = binned ? (unsigned long long*)((int4*)zbuf + (int64_t)ncand * nt) : nullptr;
    {
        const int64_t ghist_words = (int64_t)(ncand + nimages) * nblk * PCL_HBINS, stat_words = binned ? (int64_t)ncand * nt : 0;
        int64_t work = binned ? n : 0;
        if (work < (ghist_words >> 2)) work = ghist_words >> 2;
        if (work < stat_words) work = stat_words;
        if (work < ncand) work = ncand;
        const int64_t blocks = (work + PCL_BLOCK - 1) / PCL_BLOCK;
        hipLaunchKernelGGL(pcl_hist_prepare_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(PCL_BLOCK), 0, s, trans, rot, ncand, recs, ghist_q,
                           ghist_words, stat, stat_words, cloud, n, stride, binned ? codes : (uint16_t*)nullptr);
    }
    hipLaunchKernelGGL(pcl_hist_accum_kernel<0>, dim3(nblk * PCL_HSUB_QUERY, nimages), dim3(PCL_BLOCK), 0, s, (const unsigned long long*)nullptr,
                       cloud, stride, imgs, cpi, H, W, nsh, nsw, ghist_q, qmask);
    // Tile-binned path when the caller sized the workspace for it (pcl_hist_trim_workspace_bytes_n); a caller that passes the smaller
    // pcl_hist_trim_workspace_bytes gets the z-buffer splat — how the tests compare the two bit for bit.
    const int64_t cap = 4 * n;
    if (binned) {
        PclBinArgs b;
        b.cloud = cloud; b.n = n; b.stride = stride; b.poses = recs; b.H = H; b.W = W; b.ntx = ntx; b.nt = nt;
        b.nb = (int)((n + PCL_BIN_PTS - 1) / PCL_BIN_PTS);
        b.heads = (int4*)zbuf;
        b.stat = stat;
        b.runs = (uint2*)(b.stat + (int64_t)ncand * nt);
        b.lists = (uint32_t*)(b.runs + (int64_t)ncand * nt * b.nb);
        b.cap = cap;
        // margins of the fast projection's certificate: 1.5e-6 x the image size (three times the error budget in the kernel's comment),
        // at least 1e-3 pixel; PCL_BIN_EXACT=1: the reference formula for every point (A/B, and the cross-check of the certificate)
        const bool exact_env = PCL_KNOB(BIN_EXACT, 0) != 0;
        b.fast_margin_x = exact_env ? 0.f : fmaxf(1e-3f, 1.5e-6f * (float)W);
        b.fast_margin_y = exact_env ? 0.f : fmaxf(1e-3f, 1.5e-6f * (float)H);
        const int bh = H / nsh, r_hi = (nsh - 1) * bh - 1;
        b.ty_lo = bh >> PCL_TS_SHIFT; b.ty_hi = (r_hi < H - 1 ? r_hi : H - 1) >> PCL_TS_SHIFT;
        // (the tiles' statistics were zeroed and the colour codes written by pcl_hist_prepare_kernel)
        dim3 pgrid((unsigned)b.nb, (unsigned)ncand);
        // pre-dedup pays where pixels hold several points: measured (round 4, two-pass form) at 1M points on 2048 x 1024 (0.5 points per
        // pixel) resolve 613 -> 498, scatter 491 -> 437, count 283 -> 355 us per 64 candidates (-7 % for the stage); at 167k points (0.08
        // per pixel) nothing is dropped and the compares cost 9 us per 50 candidates — hence the density gate.  PCL_BIN_DEDUP=0 / 1 forces.
        const int dedup_env = PCL_KNOB(BIN_DEDUP, -1);
        const bool dedup = dedup_env >= 0 ? dedup_env != 0 : 4 * n >= (int64_t)H * W;
        if (dedup) hipLaunchKernelGGL((pcl_bin_kernel<true>), pgrid, dim3(PCL_BLOCK), (size_t)2 * nt * sizeof(int), s, b);
        else hipLaunchKernelGGL((pcl_bin_kernel<false>), pgrid, dim3(PCL_BLOCK), (size_t)2 * nt * sizeof(int), s, b);
        hipLaunchKernelGGL(pcl_bin_rank_kernel, dim3(ncand, (nt + 63) / 64), dim3(PCL_BLOCK), (size_t)(((nt + 3) & ~3) + PCL_BLOCK) * sizeof(int), s, b);
        const int rt_env = PCL_KNOB(RESOLVE_THREADS, 0);
        const int rt = rt_env == 256 ? 256 : 1024;      // measured: 256 threads LOSE at both shapes (0.434 -> 0.489 ms at 167k x 50, 1.35 -> 1.53 at 1M x 64)
        if (rt == 1024) hipLaunchKernelGGL(pcl_tile_resolve_hist_kernel<1024>, dim3(ncand, nt), dim3(1024), 0, s, b, qmask, codes, cpi, nsh, nsw, ghist_c);
        else hipLaunchKernelGGL(pcl_tile_resolve_hist_kernel<256>, dim3(ncand, nt), dim3(256), 0, s, b, qmask, codes, cpi, nsh, nsw, ghist_c);
    } else {
        hipLaunchKernelGGL(pcl_fill_u64b_kernel, dim3(2048), dim3(PCL_BLOCK), 0, s, zbuf, (int64_t)ncand * H * W, ~0ull);
        // 64 x 64-pixel LDS window (32 KB of 64-bit cells) per 256 consecutive (Morton-ordered) points: a compact surface
        // patch whose splats nearly all land inside the window.  Measured at cfg-2 size, 64 candidates (whole trimming
        // stage): 7.6 ms with 2048 points per block — their patch is wider than the window at close range and the overflow
        // goes to global atomics one splat at a time —, 5.4 / 4.2 / 3.6 ms with 1024 / 512 / 256; other windows at
        // 256-512 points: 64x96 3.8, 48x64 4.5, 48x48 4.0 ms.
        constexpr int TH = 64, TW = 64, PTS = 256;
        hipLaunchKernelGGL((pcl_splat_poses_kernel<TH, TW, PTS>), dim3((unsigned)((n + PTS - 1) / PTS), (unsigned)ncand),
                           dim3(PCL_BLOCK), 0, s, cloud, n, stride, recs, H, W, zbuf);
        hipLaunchKernelGGL(pcl_hist_accum_kernel<1>, dim3(nblk * PCL_HSUB, ncand), dim3(PCL_BLOCK), 0, s, zbuf, cloud, stride, imgs, cpi, H, W,
                           nsh, nsw, ghist_c, (uint8_t*)nullptr);
    }
    hipLaunchKernelGGL(pcl_hist_final_kernel, dim3(nblk, ncand), dim3(PCL_BLOCK), 0, s, (const unsigned int*)ghist_c, (const unsigned int*)ghist_q, nimg, inter,
                       nproj, cpi);
    PCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcl_hist_trim_scores(const float* cloud, int64_t n, const float* img_hwc, int H, int W,
                                    const float* trans, const float* rot, int ncand, int nsh, int nsw, float* inter, int* nproj,
                                    int* nimg, void* workspace, size_t workspace_bytes, void* stream)
{
    return pcl_hist_trim_scores_images(cloud, n, &img_hwc, 1, ncand, H, W, trans, rot, nsh, nsw, inter, nproj, nimg, workspace, workspace_bytes,
                                       stream);
}

// Scores with the reference's slot semantics.  The reference keeps ONE vector `hist_intersect_split` of nsh * nsw slots for
// all candidates (allocated before the candidate loop, utils.py:539): a block with nothing to histogram — no rendered or no
// query pixel — writes 0 into its slot and `break`s out of its block ROW (utils.py:568-571); the row's remaining slots
// are not touched, so they still hold what the last candidate that got that far left there (possibly itself a stale
// value), NaNs cleaned to 0 in place (utils.py:579).  score = sum of the slots / (nsh * nsw) (utils.py:580).  Pinned by
// G19 (per-candidate slot vectors read out of the running reference function).
// The carry makes the candidates a sequential chain per slot: one block, thread j owns slot j (strided if there are more
// slots than threads) and walks the candidates in order with its carried value in a register, writing the slot's content
// after every candidate to LDS; then one thread per candidate sums that candidate's slots in slot order (deterministic).
// Candidates go through in chunks that fit the LDS table; no barrier inside the walk (the first version synchronised twice
// per candidate: 78 us per call for 64 candidates).
#define PCL_SCORE_LDS_FLOATS 6144
__global__ void __launch_bounds__(256) pcl_hist_score_kernel(const float* __restrict__ inter_all, const int* __restrict__ nproj_all,
                                                             const int* __restrict__ nimg_all, int ncand, int nsh, int nsw,
                                                             float* __restrict__ score_all)
{
    // one block per query image: its ncand candidates form their own chain (the reference's slot vector lives in one call of
    // trim_input_hist_secondary, i.e. one image)
    const float* __restrict__ inter = inter_all + (int64_t)blockIdx.x * ncand * ((nsh - 2) * nsw);
    const int* __restrict__ nproj = nproj_all + (int64_t)blockIdx.x * ncand * ((nsh - 2) * nsw);
    const int* __restrict__ nimg = nimg_all + (int64_t)blockIdx.x * ((nsh - 2) * nsw);
    float* __restrict__ score = score_all + (int64_t)blockIdx.x * ncand;
    __shared__ float eff[PCL_SCORE_LDS_FLOATS];           // [chunk][nblk]: slot contents after each candidate of the chunk
    __shared__ float carry_slot[1024];                    // carried slot values between chunks (nblk <= 1024)
    __shared__ int brk[PCL_SCORE_LDS_FLOATS];             // [chunk][rows]: first empty block of the row (nsw if none)
    const int nblk = (nsh - 2) * nsw;
    const int chunk = max(1, PCL_SCORE_LDS_FLOATS / nblk);
    for (int j = threadIdx.x; j < nblk; j += blockDim.x) carry_slot[j] = 0.f;
    __syncthreads();
    const int nrows = nsh - 2;
    for (int c0 = 0; c0 < ncand; c0 += chunk) {
        const int c1 = min(c0 + chunk, ncand);
        // stage the chunk: the intersections (NaN -> 0) and, per (candidate, block row), where the reference breaks
        for (int i = threadIdx.x; i < (c1 - c0) * nblk; i += blockDim.x) {
            float v = inter[(int64_t)c0 * nblk + i];
            eff[i] = (v == v) ? v : 0.f;
        }
        for (int i = threadIdx.x; i < (c1 - c0) * nrows; i += blockDim.x) {
            const int cand = c0 + i / nrows, row0 = (i % nrows) * nsw;
            const int* np = nproj + (int64_t)cand * nblk;
            int first_empty = nsw;
            for (int ww = 0; ww < nsw; ww++)
                if (np[row0 + ww] == 0 || nimg[row0 + ww] == 0) { first_empty = ww; break; }
            brk[i] = first_empty;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < nblk; j += blockDim.x) {
            const int row = j / nsw, w = j - row * nsw;
            float carry = carry_slot[j];
            for (int cand = c0; cand < c1; cand++) {
                const int first_empty = brk[(cand - c0) * nrows + row];
                if (w < first_empty) carry = eff[(cand - c0) * nblk + j];
                else if (w == first_empty) carry = 0.f;   // (w > first_empty: untouched, keeps the earlier candidate's value)
                eff[(cand - c0) * nblk + j] = carry;
            }
            carry_slot[j] = carry;
        }
        __syncthreads();
        for (int cand = c0 + threadIdx.x; cand < c1; cand += blockDim.x) {
            float total = 0.f;
            for (int j = 0; j < nblk; j++) total += eff[(cand - c0) * nblk + j];
            score[cand] = total / (float)(nsh * nsw);
        }
        __syncthreads();
    }
}

extern "C" int pcl_hist_trim_reduce_images(const float* inter, const int32_t* nproj, const int32_t* nimg, int nimages, int cand_per_image, int nsh,
                                           int nsw, float* score, void* stream)
{
    if (!inter || !nproj || !nimg || !score || nimages <= 0 || cand_per_image <= 0 || nsh < 3 || nsw < 1) return PCL_EINVAL;
    if ((nsh - 2) * nsw > 1024) return PCL_EINVAL;        // > 1024 blocks: not a block grid this stage is meant for
    hipLaunchKernelGGL(pcl_hist_score_kernel, dim3(nimages), dim3(256), 0, (hipStream_t)stream, inter, nproj, nimg, cand_per_image, nsh, nsw, score);
    PCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcl_hist_trim_reduce(const float* inter, const int32_t* nproj, const int32_t* nimg, int ncand, int nsh, int nsw, float* score,
                                    void* stream)
{
    return pcl_hist_trim_reduce_images(inter, nproj, nimg, 1, ncand, nsh, nsw, score, stream);
}
