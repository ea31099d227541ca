// pcl_depth.hip — the scatter-min depth mask of the north star, per candidate pose, on the PACKED cloud.
//
// Build-defined (the reference imports torch_scatter.scatter_min at utils.py:6 and never calls it; its loss has no occlusion
// test).  For every pose b: cell = make_pano's pixel (utils.py:158-165) of p = R_b (x - t_b) on an Hd x Wd grid — the DEPTH
// RESOLUTION, chosen by point density, not the panorama's — zmin2[b][cell] = min over the points landing there of ||p||^2 (a
// scatter-min; stored already scaled by (1 + tau)^2), and point i is visible for pose b iff ||p_i||^2 <= zmin2 (1 + tau)^2.  Off by default: with the mask off the loss
// is exactly the reference's.
//
// Why the grid is coarse (round 5; tools/depth_recall.py, profiles/r05/depth_recall.txt): a z-buffer can only hide a point when an
// occluder's point lands in the SAME cell.  At the panorama's resolution (1M points over 2048 x 1024 pixels: 0.5 points per
// pixel) an occluded wall point is usually alone in its pixel: measured against analytic ray / box occlusion the round-4 mask
// found 35 % of the occluded points (9 % at the shipped 167k points).  With >= 12 points per cell recall is 0.93-0.97; the price
// of a coarse cell is that a surface seen at a grazing angle spans more than tau in depth inside one cell and hides its own far
// side, so tau grows with the cell's angular size (pcl_depth_default: tau = 3.5 pi / Hd, clipped to [0.02, 0.15]: precision
// 0.93-0.99).  1M points, every point an occluder sample: 200 x 400 cells — make_pano's own default resolution (utils.py:134) — and
// tau = 0.055; with the default occluder stride of 2: 144 x 288 cells, tau = 0.076, the same recall / precision (pcl_depth_default).
//
// Kernels per GD iteration (all B poses; DESIGN.md section 4.5, the experiments behind every choice: profiles/EXPERIMENTS.md section 9):
//   z pass   : pcl_zpass_kernel.  Occluder samples (every zstride-th packed point) quad-interleaved over the lanes, whole 16-byte loads;
//              the loss kernel's own packed projection (pcl_rotate2 + pcl_angles2: the same instructions on the same inputs, so a
//              sample lands in the cell the loss kernel will look up); a block's 4096 Morton-consecutive samples are resolved with LDS
//              atomicMin in a 48 x 128-cell window anchored on the run's middle sample (columns wrap at the +-pi seam), the ones
//              outside it through a small queue in a second 32 x 64 window, the rest (0.6 %) by one global atomic each; flush: one
//              global atomicMin per NON-EMPTY window cell, 64 consecutive cells per wave instruction.  Global atomics execute at the
//              memory side (MI355X_MICROARCH.md: not in L2), one 64-byte request per touched segment: 15k requests per pose.
//              Grids narrower than a window: pcl_zcache_kernel (an LDS cache of coarse tiles); PCL_ZFORM=3: the untiled scatter (A/B).
//   (lookup) : there is no mark pass and no byte mask in the GD loop: pcl_loss_kernel<.., VIS = 2, ..> reads the cell of each point
//              while its texels are in flight (pcl_sample_device.h, pcl_project2_rotated<FMT, DEPTH>) and resets a slice of the OTHER
//              z-buffer set for the next iteration — so no fill launch either, except in front of a call's first iteration.
//   mark     : only for the stand-alone pcl_depth_mask (a byte mask for callers of pcl_sampling_loss's `visible`).
#include <stdlib.h>

#include "pcl_sample_device.h"

struct PclZArgs {
    const float* cloud;
    int64_t n, stride;
    const PclPoseRec* poses;
    int B;
    PclDepthGrid g;
    uint32_t* zbuf;       // [B][Hd * Wd]
    uint8_t* visible;     // [B][n]   (mark pass)
    int zstride;          // the z pass reads every zstride-th point of the packed (Morton-ordered) cloud: nz = ceil(n / zstride) occluder samples
    int64_t nz;
};

#define PCL_Z_INF 0x7f800000u

// two points of pose P -> cells (the loss kernel's own instructions; NOT clamped: in range for every finite point, see
// pcl_depth_cells2 — whoever addresses global memory with them clamps) and squared depths
__device__ __forceinline__ void pcl_depth_pair(f2 x, f2 y, f2 z, const PclPose6& P, const PclDepthGrid& g, int& row0, int& col0, int& row1,
                                               int& col1, f2& d2)
{
    f2 px, py, pz, rho2, rinv, rs2, phi, hel;
    pcl_rotate2(x, y, z, P, px, py, pz);
    pcl_angles2(px, py, pz, rho2, rinv, rs2, phi, hel);
    pcl_depth_cells2(phi, hel, g, row0, col0, row1, col1);
    d2 = pcl_fma2(pz, pz, rho2);
}
__device__ __forceinline__ int pcl_depth_clamped_cell(int row, int col, const PclDepthGrid& g)
{
    return min(max(row, 0), g.Hd - 1) * g.Wd + min(max(col, 0), g.Wd - 1);
}

// block -> (chunk, pose): blocks b and b + 8 share an XCD (round-robin dispatch); within an XCD the pose varies fastest, so the
// blocks resident together read the same cloud chunk out of that XCD's L2
__device__ __forceinline__ void pcl_z_block(int B, int& chunk, int& pose)
{
    const int xcd = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3), cj = j / B;
    pose = j - cj * B;
    chunk = cj * 8 + xcd;
}

// A block's occluder samples, QUAD-interleaved: thread t takes the samples base + 4 NT q + 4 t + e (q-th quad, e = 0..3): a quad is
// one 16-byte load per plane (stride 1) and the lanes of a wave read consecutive quads — fully coalesced — while the 64 lanes of one
// LDS-atomic instruction hold samples 4 apart in the Morton order, spread over ~20 cells.  (Lane-major, i = base + NT k + t: 64
// Morton neighbours share ~5 cells and serialise on them — ~50 cycles per LDS atomic instruction, 40 % of the first z pass.
// Thread-major, 16 consecutive samples per thread: the fewest conflicts, but every lane of a load reads its own cache line — with an
// occluder stride of 2 the z pass of HALF the points took 94 us against 80 us for all of them.)
// pcl_depth_quad: points first + 4 q .. + 3 of this thread -> cells (not clamped) and scaled squared-depth keys.
// FULL: every point of the block exists (all blocks but a cloud's last): no end-of-cloud tests.
// (i counts occluder SAMPLES: sample i is packed point i * zstride; zstride 1: one 16-byte load per plane)
template <bool FULL>
__device__ __forceinline__ void pcl_depth_quad(const PclZArgs& a, __amdgpu_buffer_rsrc_t cld, int i, const PclPose6& P, int (&row)[4], int (&col)[4],
                                               uint32_t (&key)[4])
{
    const int plane = (int)a.stride * 4, last = (int)a.nz - 1;                 // (the planes are padded to 256 floats: whole 16-byte loads)
    pcl_f4 vx, vy, vz;
    auto ld = [&](int idx, int soff) { return __builtin_bit_cast(pcl_f4, __builtin_amdgcn_raw_buffer_load_b128(cld, idx * 4, soff, 0)); };
    if (a.zstride == 1) {                                                      // (wave-uniform branches)
        vx = ld(i, 0); vy = ld(i, plane); vz = ld(i, 2 * plane);
    } else if (a.zstride == 2) {
        // strides 2 and 4: still whole 16-byte loads of CONSECUTIVE points, the samples picked out of them.  (Dword loads at the
        // samples' addresses are one cache access per LANE: measured at stride 2, 49.6M L1 accesses per launch against 7.8M, and the z
        // pass of half the points took longer than that of all of them.)
        pcl_f4 a0 = ld(2 * i, 0), a1 = ld(2 * i + 4, 0), b0 = ld(2 * i, plane), b1 = ld(2 * i + 4, plane), c0 = ld(2 * i, 2 * plane), c1 = ld(2 * i + 4, 2 * plane);
        vx = (pcl_f4){a0.x, a0.z, a1.x, a1.z}; vy = (pcl_f4){b0.x, b0.z, b1.x, b1.z}; vz = (pcl_f4){c0.x, c0.z, c1.x, c1.z};
    } else if (a.zstride == 4) {
        vx = (pcl_f4){ld(4 * i, 0).x, ld(4 * i + 4, 0).x, ld(4 * i + 8, 0).x, ld(4 * i + 12, 0).x};
        vy = (pcl_f4){ld(4 * i, plane).x, ld(4 * i + 4, plane).x, ld(4 * i + 8, plane).x, ld(4 * i + 12, plane).x};
        vz = (pcl_f4){ld(4 * i, 2 * plane).x, ld(4 * i + 4, 2 * plane).x, ld(4 * i + 8, 2 * plane).x, ld(4 * i + 12, 2 * plane).x};
    } else {
        const int o = i * a.zstride * 4, d = a.zstride * 4;                    // (a sample past the end reads padding or 0: its key is +inf)
        vx = (pcl_f4){__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, o, 0, 0)),
                      __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, o + d, 0, 0)),
                      __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, o + 2 * d, 0, 0)),
                      __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, o + 3 * d, 0, 0))};
        vy = (pcl_f4){__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, o, plane, 0)),
                      __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, o + d, plane, 0)),
                      __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, o + 2 * d, plane, 0)),
                      __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, o + 3 * d, plane, 0))};
        vz = (pcl_f4){__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, o, 2 * plane, 0)),
                      __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, o + d, 2 * plane, 0)),
                      __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, o + 2 * d, 2 * plane, 0)),
                      __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, o + 3 * d, 2 * plane, 0))};
    }
    f2 d2a, d2b;
    pcl_depth_pair((f2){vx.x, vx.y}, (f2){vy.x, vy.y}, (f2){vz.x, vz.y}, P, a.g, row[0], col[0], row[1], col[1], d2a);
    pcl_depth_pair((f2){vx.z, vx.w}, (f2){vy.z, vy.w}, (f2){vz.z, vz.w}, P, a.g, row[2], col[2], row[3], col[3], d2b);
    d2a = d2a * F2(a.g.tol2); d2b = d2b * F2(a.g.tol2);                         // cell = (zmin (1 + tau))^2: the lookup compares d2 with it
    key[0] = FULL || i <= last ? __float_as_uint(d2a.x) : PCL_Z_INF;           // (a point past the end never wins a cell)
    key[1] = FULL || i + 1 <= last ? __float_as_uint(d2a.y) : PCL_Z_INF;
    key[2] = FULL || i + 2 <= last ? __float_as_uint(d2b.x) : PCL_Z_INF;
    key[3] = FULL || i + 3 <= last ? __float_as_uint(d2b.y) : PCL_Z_INF;
}
template <int PAIRS, bool FULL>
__device__ __forceinline__ void pcl_depth_block_points(const PclZArgs& a, int64_t base, const PclPose6& P, int (&row)[2 * PAIRS], int (&col)[2 * PAIRS],
                                                       uint32_t (&key)[2 * PAIRS])
{
    static_assert(PAIRS % 2 == 0, "four points per 16-byte load");
    __amdgpu_buffer_rsrc_t cld = __builtin_amdgcn_make_buffer_rsrc((void*)a.cloud, 0, (int)(a.stride * 3 * 4), 0x00020000);
    const int first = (int)base + 4 * (int)threadIdx.x, qstep = 4 * (int)blockDim.x;
#pragma unroll
    for (int q = 0; q < PAIRS / 2; q++) {
        int r4[4], c4[4];
        uint32_t k4[4];
        pcl_depth_quad<FULL>(a, cld, first + qstep * q, P, r4, c4, k4);
#pragma unroll
        for (int j = 0; j < 4; j++) { row[4 * q + j] = r4[j]; col[4 * q + j] = c4[j]; key[4 * q + j] = k4[j]; }
    }
}

// LDS-tiled z pass (header).  TW <= Wd required (the launcher falls back to the direct kernel for smaller grids).
// The window is anchored on ONE point of the block — the middle one of its Morton run, which thread 0 holds — so that its cell is
// the window's centre.  (First version: the block's MEAN cell, columns averaged as wrapped offsets: three wave reductions, LDS
// atomics, a second barrier and two integer divisions per block, ~150 of the kernel's 700 VALU instructions per thread — the
// window has room to spare for an off-centre anchor: a 2048-point run covers ~15 x 15 cells of the default grids, the window 32 x 64.)
// SECOND: a second, smaller window for what the first one misses.  A Morton run is compact in SPACE, not in the image: it may
// straddle two walls, or lie near a pole — 4.5 % of the samples fall outside a 48 x 128 window at cfg 2, each one a memory-side atomic
// of its own, and those were most of the launch's 34k requests per pose (CPU simulation, profiles/r05/experiments/zpass_windows.txt:
// 44.6k direct + 8.5k flushed segments; with a 32 x 64 second window 6.1k + 9.3k).  The outside samples go to a small LDS queue
// (cell, key); after the barrier the first queued sample's cell anchors window B and every queued sample is resolved by ONE thread
// (one pass over at most NQ entries instead of a second sweep over all 2 PAIRS samples of every thread).
template <int TH, int TW, int PTS, int NT, bool SECOND = false>
__global__ void __launch_bounds__(NT) pcl_zpass_kernel(PclZArgs a)
{
    constexpr int PAIRS = PTS / (2 * NT);
    constexpr int TH2 = 32, TW2 = 64, NQ = 768;            // window B, queue capacity (an overflowing sample goes to global memory at once)
    static_assert(PAIRS >= 2 && PAIRS % 2 == 0 && PTS % (2 * NT) == 0, "whole point quads per lane");
    static_assert((TW & (TW - 1)) == 0 && (TH * TW) % (4 * NT) == 0, "window: a power-of-two width, whole 16-byte words per thread");
    __shared__ __attribute__((aligned(16))) uint32_t tile[TH * TW];
    __shared__ __attribute__((aligned(16))) uint32_t tile2[SECOND ? TH2 * TW2 : 4];
    __shared__ uint32_t qcell[SECOND ? NQ : 1], qkey[SECOND ? NQ : 1];
    __shared__ int org[2], qn;
    int chunk, b;
    pcl_z_block(a.B, chunk, b);
    const int64_t base = (int64_t)chunk * PTS;                                 // (in occluder samples)
    if (base >= a.nz) return;                                                  // (block-uniform, before the first barrier)
    const PclPose6 P = pcl_pose6(a.poses + b);
    uint32_t* __restrict__ zb = a.zbuf + (int64_t)b * (a.g.last + 1);
    {
        const pcl_i4 inf4 = {(int)PCL_Z_INF, (int)PCL_Z_INF, (int)PCL_Z_INF, (int)PCL_Z_INF};
        pcl_i4* t4 = reinterpret_cast<pcl_i4*>(tile);
#pragma unroll
        for (int i = 0; i < TH * TW / 4 / NT; i++) t4[i * NT + threadIdx.x] = inf4;
        if constexpr (SECOND) {
            static_assert((TH2 * TW2) % (4 * NT) == 0, "window B: whole 16-byte words per thread");
            pcl_i4* u4 = reinterpret_cast<pcl_i4*>(tile2);
#pragma unroll
            for (int i = 0; i < TH2 * TW2 / 4 / NT; i++) u4[i * NT + threadIdx.x] = inf4;
            if (threadIdx.x == 0) qn = 0;
        }
    }
    const int last = (int)a.nz - 1, Wd = a.g.Wd;
    int row[2 * PAIRS], col[2 * PAIRS];
    uint32_t key[2 * PAIRS];
    const bool full = base + PTS <= a.nz;                                      // (block-uniform)
    if (full) pcl_depth_block_points<PAIRS, true>(a, base, P, row, col, key);
    else pcl_depth_block_points<PAIRS, false>(a, base, P, row, col, key);
    // the middle sample of the run is the first of thread 0's quad PAIRS / 4 (a run cut short by the end of the cloud: the run's first)
    const bool short_run = base + PTS / 2 > (int64_t)last;
    if (threadIdx.x == 0) { org[0] = short_run ? row[0] : row[PAIRS]; org[1] = short_run ? col[0] : col[PAIRS]; }
    __syncthreads();
    const int r0 = org[0] - TH / 2;
    int c0 = org[1] - TW / 2;
    c0 = c0 < 0 ? c0 + Wd : c0;                                                // window columns c0 .. c0 + TW - 1 (mod Wd)
#pragma unroll
    for (int k = 0; k < 2 * PAIRS; k++) {
        int tc = col[k] - c0;
        tc = tc < 0 ? tc + Wd : tc;
        const unsigned tr = (unsigned)(row[k] - r0);
        if (!full && key[k] == PCL_Z_INF) continue;
        if (tr < (unsigned)TH && (unsigned)tc < (unsigned)TW) atomicMin(&tile[tr * TW + tc], key[k]);
        else {
            const unsigned cell = (unsigned)pcl_depth_clamped_cell(row[k], col[k], a.g);
            int pos = NQ;
            if constexpr (SECOND) pos = atomicAdd(&qn, 1);
            if (pos < NQ) { qcell[pos] = cell; qkey[pos] = key[k]; }
            else atomicMin(&zb[cell], key[k]);
        }
    }
    __syncthreads();
    int r2 = 0, c2 = 0;
    if constexpr (SECOND) {
        const int nq = min(qn, NQ);
        if (nq > 0) {                                                          // (block-uniform)
            const unsigned cell0 = qcell[0];
            r2 = (int)(cell0 / (unsigned)Wd) - TH2 / 2;
            c2 = (int)(cell0 % (unsigned)Wd) - TW2 / 2;
            c2 = c2 < 0 ? c2 + Wd : c2;
            for (int e = threadIdx.x; e < nq; e += NT) {
                const unsigned cell = qcell[e];
                const int r = (int)(cell / (unsigned)Wd), c = (int)(cell % (unsigned)Wd);
                int tc = c - c2;
                tc = tc < 0 ? tc + Wd : tc;
                const unsigned tr = (unsigned)(r - r2);
                if (tr < (unsigned)TH2 && (unsigned)tc < (unsigned)TW2) atomicMin(&tile2[tr * TW2 + tc], qkey[e]);
                else atomicMin(&zb[cell], qkey[e]);
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < TH * TW; i += NT) {
        const uint32_t v = tile[i];
        if (v == PCL_Z_INF) continue;
        const int r = r0 + i / TW;
        int c = c0 + (i & (TW - 1));
        c = c >= Wd ? c - Wd : c;
        atomicMin(&zb[(unsigned)pcl_depth_clamped_cell(r, c, a.g)], v);
    }
    if constexpr (SECOND) {
        if (min(qn, NQ) > 0) {
            for (int i = threadIdx.x; i < TH2 * TW2; i += NT) {
                const uint32_t v = tile2[i];
                if (v == PCL_Z_INF) continue;
                const int r = r2 + i / TW2;
                int c = c2 + (i & (TW2 - 1));
                c = c >= Wd ? c - Wd : c;
                atomicMin(&zb[(unsigned)pcl_depth_clamped_cell(r, c, a.g)], v);
            }
        }
    }
}

// z pass, coarse-tile CACHE form (round 5; taken for grids narrower than a window).  What a single window loses: a Morton run is compact in SPACE, not
// in the image — it may straddle two walls, or lie near a pole where a small patch spans every column — and every point outside
// the window is one global atomic of its own: measured 5.7 % of the points at cfg 2 (simulated on the CPU with the oracle's pixels,
// profiles/r05/experiments/zpass_windows.txt: 57k direct atomics + 10k flushed 64-byte segments per pose), and those 5.7 % were
// ~60 % of the kernel's time (109 us; 170 us with 4096-point blocks, where 10.8 % fall outside).  Here the LDS holds S coarse tiles
// of 8 x 16 cells (one 64-byte segment of the z-buffer per tile row), direct-mapped by (5 tr + tc) mod S: a point CLAIMS the slot
// of its tile with one compare-and-swap on the slot's tag (empty -> my tile), resolves into it with an LDS atomicMin when the tag
// is its tile, and goes to the global z-buffer only when another tile holds the slot (S = 32, 2048 points: 0.7 % of the points;
// S = 64, 4096 points: 0.7 %; total memory-side requests per pose 67k -> 18k / 16k).  Disjoint patches each get their own slots — no
// anchor, no second pass.  On the wide default grids it LOSES to the window form (237 vs 195 us per iteration at cfg 2): the
// compare-and-swap of 2048 samples on ~6 tags serialises in LDS.  Flush: each wave walks its share of the slots, skips the empty ones on the scalar unit, and issues the
// claimed ones' cells as 64 consecutive cells per instruction = four whole 64-byte segments.
template <int S, int PTS>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_zcache_kernel(PclZArgs a)
{
    constexpr int PAIRS = PTS / (2 * PCL_BLOCK), CH = 8, CW = 16, CELLS = CH * CW;
    static_assert(PAIRS >= 1 && PTS % (2 * PCL_BLOCK) == 0, "whole point pairs per lane");
    static_assert((S & (S - 1)) == 0 && S >= 16 && S <= 64 * (PCL_BLOCK / PCL_WAVE), "slots: a power of two, at most 64 per wave");
    __shared__ __attribute__((aligned(16))) uint32_t tile[S * CELLS];
    __shared__ uint32_t tags[S];
    int chunk, b;
    pcl_z_block(a.B, chunk, b);
    const int64_t base = (int64_t)chunk * PTS;                                 // (in occluder samples)
    if (base >= a.nz) return;                                                  // (block-uniform, before the first barrier)
    const PclPose6 P = pcl_pose6(a.poses + b);
    uint32_t* __restrict__ zb = a.zbuf + (int64_t)b * (a.g.last + 1);
    {
        const pcl_i4 inf4 = {(int)PCL_Z_INF, (int)PCL_Z_INF, (int)PCL_Z_INF, (int)PCL_Z_INF};
        pcl_i4* t4 = reinterpret_cast<pcl_i4*>(tile);
#pragma unroll
        for (int i = 0; i < S * CELLS / 4 / PCL_BLOCK; i++) t4[i * PCL_BLOCK + threadIdx.x] = inf4;
        if (threadIdx.x < S) tags[threadIdx.x] = 0xffffffffu;
    }
    int row[2 * PAIRS], col[2 * PAIRS];
    uint32_t key[2 * PAIRS];
    pcl_depth_block_points<PAIRS, false>(a, base, P, row, col, key);
    __syncthreads();                                                           // tile and tags initialised
    // claim: all compare-and-swaps of a thread in flight together; tag = (tile row << 16) | tile column
    uint32_t tag[2 * PAIRS], old[2 * PAIRS];
    int slot[2 * PAIRS];
#pragma unroll
    for (int k = 0; k < 2 * PAIRS; k++) {
        const int tr = row[k] >> 3, tc = col[k] >> 4;
        tag[k] = ((uint32_t)tr << 16) | (uint32_t)(tc & 0xffff);
        slot[k] = (tr * 5 + tc) & (S - 1);
        old[k] = key[k] == PCL_Z_INF ? 0u : atomicCAS(&tags[slot[k]], 0xffffffffu, tag[k]);
    }
#pragma unroll
    for (int k = 0; k < 2 * PAIRS; k++) {
        if (key[k] == PCL_Z_INF) continue;
        if (old[k] == 0xffffffffu || old[k] == tag[k]) atomicMin(&tile[slot[k] * CELLS + ((row[k] & (CH - 1)) << 4) + (col[k] & (CW - 1))], key[k]);
        else atomicMin(&zb[pcl_depth_clamped_cell(row[k], col[k], a.g)], key[k]);
    }
    __syncthreads();
    // flush: wave w takes the slots w, w + 4, ...; lane l holds the tag of the wave's l-th slot
    const int lane = threadIdx.x & (PCL_WAVE - 1), wave = threadIdx.x >> 6;
    constexpr int NW = PCL_BLOCK / PCL_WAVE, PER_WAVE = S / NW;
    const uint32_t mytag = lane < PER_WAVE ? tags[wave + NW * lane] : 0xffffffffu;
    const int Wd = a.g.Wd, Hd = a.g.Hd;
#pragma unroll 1
    for (int q = 0; q < PER_WAVE; q++) {
        const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)mytag, q);
        if (t == 0xffffffffu) continue;                                        // (wave-uniform: scalar compare and branch)
        const int sl = wave + NW * q, r0 = (int)(t >> 16) * CH, c0 = (int)(t & 0xffffu) * CW;
#pragma unroll
        for (int h = 0; h < CELLS / PCL_WAVE; h++) {
            const int i = h * PCL_WAVE + lane;
            const uint32_t v = tile[sl * CELLS + i];
            const int r = r0 + (i >> 4), c = c0 + (i & (CW - 1));
            if (v != PCL_Z_INF && r < Hd && c < Wd) atomicMin(&zb[r * Wd + c], v);
        }
    }
}

// MARK = false: the untiled z pass (grids smaller than a window; PCL_ZPASS_DIRECT=1: the A/B knob) — one global atomicMin per
// point-pose.  MARK = true: byte mask of the stand-alone pcl_depth_mask.  Two points per lane, 2 * PCL_BLOCK points per step.
template <bool MARK>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_depth_kernel(PclZArgs a, int pts_per_block)
{
    int chunk, b;
    pcl_z_block(a.B, chunk, b);
    // MARK: every point; z pass: the occluder samples (sample i = packed point i * zstride)
    const int64_t count = MARK ? a.n : a.nz;
    const int step = MARK ? 1 : a.zstride;
    const int64_t base = (int64_t)chunk * pts_per_block;
    if (base >= count) return;
    const PclPose6 P = pcl_pose6(a.poses + b);
    uint32_t* __restrict__ zb = a.zbuf + (int64_t)b * (a.g.last + 1);
    __amdgpu_buffer_rsrc_t cld = __builtin_amdgcn_make_buffer_rsrc((void*)a.cloud, 0, (int)(a.stride * 3 * 4), 0x00020000);
    const int plane = (int)a.stride * 4, last = (int)count - 1;
    const int end = (int)min(count, base + pts_per_block);
    for (int s0 = (int)base; s0 < end; s0 += 2 * PCL_BLOCK) {
        const int i0 = s0 + (int)threadIdx.x, i1 = i0 + PCL_BLOCK;
        const int j0 = min(i0, last) * step, j1 = min(i1, last) * step;
        f2 x = {__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, j0 * 4, 0, 0)),
                __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, j1 * 4, 0, 0))};
        f2 y = {__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, j0 * 4, plane, 0)),
                __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, j1 * 4, plane, 0))};
        f2 z = {__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, j0 * 4, 2 * plane, 0)),
                __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, j1 * 4, 2 * plane, 0))};
        int r0, c0, r1, c1;
        f2 d2;
        pcl_depth_pair(x, y, z, P, a.g, r0, c0, r1, c1, d2);
        uint32_t* cell0 = zb + pcl_depth_clamped_cell(r0, c0, a.g);
        uint32_t* cell1 = zb + pcl_depth_clamped_cell(r1, c1, a.g);
        if (MARK) {
            if (i0 < end) a.visible[(int64_t)b * a.n + i0] = d2.x <= __uint_as_float(*cell0) ? 1 : 0;
            if (i1 < end) a.visible[(int64_t)b * a.n + i1] = d2.y <= __uint_as_float(*cell1) ? 1 : 0;
        } else {
            d2 = d2 * F2(a.g.tol2);
            if (i0 < end) atomicMin(cell0, __float_as_uint(d2.x));
            if (i1 < end) atomicMin(cell1, __float_as_uint(d2.y));
        }
    }
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_fill_u32x4_kernel(pcl_i4* p, int64_t n4, int v)
{
    const pcl_i4 w = {v, v, v, v};
    for (int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x; i < n4; i += (int64_t)gridDim.x * PCL_BLOCK) p[i] = w;
}

// bytes of B z-buffers (a whole number of 16-byte words: the fill stores 16 bytes per lane)
size_t pcl_depth_zbuf_bytes(int B, int Hd, int Wd) { return (((size_t)B * (size_t)Hd * (size_t)Wd * sizeof(uint32_t)) + 15) & ~(size_t)15; }

static int pcl_depth_check(int64_t n, int B, const PclDepthGrid& g)
{
    if (n <= 0 || n > PCL_MAX_POINTS || B <= 0 || g.Hd < 2 || g.Wd < 2 || g.Hd >= (1 << 24) || g.Wd >= (1 << 24)) return PCL_EINVAL;
    if ((int64_t)B * g.Hd * g.Wd * 4 >= ((int64_t)1 << 32)) return PCL_EINVAL;         // the loss kernel's 32-bit buffer descriptor
    return 0;
}

// (fill +) z pass for the B poses of `poses`: zbuf [B][Hd * Wd] afterwards holds every cell's smallest squared depth (+inf: empty).
// fill = false: the caller guarantees the buffers hold +inf already (the GD loop: the previous iteration's loss launch reset them).
// zstride: the z-buffers are built from every zstride-th point of the packed cloud (1 = all of them).
int pcl_launch_zbuffers(const float* cloud, int64_t n, const PclPoseRec* poses, int B, const PclDepthGrid& g, int zstride, uint32_t* zbuf, bool fill,
                        hipStream_t s)
{
    int rc = pcl_depth_check(n, B, g);
    if (rc) return rc;
    if (zstride < 1 || zstride > 64) return PCL_EINVAL;
    PclZArgs a;
    a.cloud = cloud; a.n = n; a.stride = pcl_cloud_stride(n);
    a.poses = poses; a.B = B; a.g = g; a.zbuf = zbuf; a.visible = nullptr;
    a.zstride = zstride; a.nz = (n + zstride - 1) / zstride;
    const int64_t n4 = (int64_t)(pcl_depth_zbuf_bytes(B, g.Hd, g.Wd) / 16);
    const int64_t fill_blocks = (n4 + PCL_BLOCK - 1) / PCL_BLOCK;
    if (fill)
        hipLaunchKernelGGL(pcl_fill_u32x4_kernel, dim3((unsigned)(fill_blocks < 2048 ? fill_blocks : 2048)), dim3(PCL_BLOCK), 0, s, (pcl_i4*)zbuf, n4,
                           (int)PCL_Z_INF);
    // Which z pass (PCL_ZFORM = 1 cache / 2 window / 3 untiled scatter forces one, PCL_ZSECOND=0 drops the second window: A/B).
    // Default: the LDS window — 48 x 128 cells + a 32 x 64 second window, 4096 samples per block on dense grids (>= 2 samples per cell:
    // every default grid); 64 x 128 cells, 2048 samples on grids so fine that most cells hold at most one sample (a grid at the
    // panorama's resolution: round 3's tuning) — and the coarse-tile cache for grids narrower than a window.
    // Measured at cfg 2, 400 x 200 grid, every point, per GD iteration (z + loss + epilogue; profiles/EXPERIMENTS.md section 9):
    // window 48 x 128 + second window 195 us, without it 208, 64 x 128 206-209, 32 x 128 216, 64 x 64 225, 512-thread blocks 206,
    // 2048 samples per block 226; cache 32 slots / 2048 samples 237, 64 / 4096 247; the first form (mean-centred 32 x 64 window,
    // lane-major samples) 229.
    static const int form_env = PCL_KNOB(ZFORM, 0), second_env = PCL_KNOB(ZSECOND, 1);
    const bool dense = (double)a.nz >= 2.0 * (double)g.Hd * (double)g.Wd;
    int form = form_env >= 1 && form_env <= 3 ? form_env : (g.Wd >= 128 ? 2 : 1);
    if (form == 2 && g.Wd < 128) form = 1;
    if (g.Hd >= 65536 * 8 || g.Wd >= 65536 * 16) form = 3;                      // (the cache's 16-bit tile coordinates)
    const int PTS = form == 2 && dense ? 4096 : 2048;
    const int64_t chunks = (a.nz + PTS - 1) / PTS, chunks8 = (chunks + 7) / 8 * 8;
    if (chunks8 * B > 0x7fffffff) return PCL_EINVAL;
    const dim3 grid((unsigned)(chunks8 * B));
    if (form == 1) hipLaunchKernelGGL((pcl_zcache_kernel<32, 2048>), grid, dim3(PCL_BLOCK), 0, s, a);
    else if (form == 2) {
        if (PTS == 2048) hipLaunchKernelGGL((pcl_zpass_kernel<64, 128, 2048, 256>), grid, dim3(256), 0, s, a);
        else if (!second_env) hipLaunchKernelGGL((pcl_zpass_kernel<48, 128, 4096, 256>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((pcl_zpass_kernel<48, 128, 4096, 256, true>), grid, dim3(256), 0, s, a);
    } else hipLaunchKernelGGL(pcl_depth_kernel<false>, grid, dim3(PCL_BLOCK), 0, s, a, PTS);
    PCL_LAUNCH_CHECK();
    return 0;
}

// The default occluder sampling, grid and tolerance for an n-point cloud seen in an H x W panorama (header of this file;
// tools/depth_recall.py, profiles/r05/depth_recall.txt).
//   grid   : >= 12 occluder samples per cell, Wd = 2 Hd, Hd a multiple of 8 (at least 16), never finer than the panorama;
//   tau    : 3.5 pi / Hd in [0.02, 0.15] (a coarser cell needs a larger tolerance);
//   stride : what the mask can hide is set by the SAMPLES PER CELL and what it hides wrongly by the cell's angular size against tau —
//            not by reading every point: at 1M points a z-buffer built from every 2nd point on the grid ITS count calls for
//            (288 x 144, tau 0.076) scores recall 0.943 / precision 0.954 against analytic occlusion, from all points (400 x 200,
//            tau 0.055) 0.948 / 0.953 — at half the z pass.  Default: the largest of 1, 2, 4 that keeps Hd >= 128 (so that tau
//            stays below 0.09); every point is still TESTED against the current pose's z-buffer at every iteration.
//            stride_in > 0 fixes it (1 = every point builds the z-buffer).
static int pcl_depth_grid_h(int64_t m)
{
    int hd = 16;
    while ((int64_t)(hd + 8) * (hd + 8) * 24 <= m && hd + 8 <= 4096) hd += 8;
    return hd;
}
extern "C" int pcl_depth_default(int64_t n, int H, int W, int stride_in, int* depth_h_host, int* depth_w_host, float* tau_host, int* stride_host)
{
    if (n <= 0 || H <= 0 || W <= 0 || stride_in < 0 || stride_in > 64) return PCL_EINVAL;
    int stride = stride_in;
    if (stride == 0) {
        stride = 1;
        for (int c = 2; c <= 4; c *= 2)
            if (pcl_depth_grid_h((n + c - 1) / c) >= 128) stride = c;      // (the sample count the grid below is sized from)
    }
    int hd = pcl_depth_grid_h((n + stride - 1) / stride);
    int wd = 2 * hd;
    if (hd > H || wd > W) { hd = H; wd = W; }
    float tau = (float)(3.5 * 3.14159265358979323846 / (double)hd);
    tau = tau < 0.02f ? 0.02f : (tau > 0.15f ? 0.15f : tau);
    if (depth_h_host) *depth_h_host = hd;
    if (depth_w_host) *depth_w_host = wd;
    if (tau_host) *tau_host = tau;
    if (stride_host) *stride_host = stride;
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

extern "C" int pcl_depth_mask(const float* cloud, int64_t n, const float* trans, const float* rot, int B, int H, int W, float tau, int stride,
                              uint8_t* visible, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!cloud || !trans || !rot || !visible || !workspace || n <= 0 || B <= 0 || H <= 0 || W <= 0 || !(tau >= 0.f) || stride < 1 || stride > 64)
        return PCL_EINVAL;
    if (workspace_bytes < pcl_depth_workspace_bytes(B, H, W)) return PCL_EWORKSPACE;
    const PclDepthGrid g = pcl_make_depth_grid(H, W, tau);
    int rc = pcl_depth_check(n, B, g);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    PclPoseRec* recs = (PclPoseRec*)workspace;
    uint32_t* zbuf = (uint32_t*)((char*)workspace + (size_t)B * sizeof(PclPoseRec));
    hipLaunchKernelGGL(pcl_depth_pose_setup_kernel, dim3((B + 255) / 256), dim3(256), 0, s, trans, rot, B, recs);
    rc = pcl_launch_zbuffers(cloud, n, recs, B, g, stride, zbuf, true, s);
    if (rc) return rc;
    PclZArgs a;
    a.cloud = cloud; a.n = n; a.stride = pcl_cloud_stride(n);
    a.poses = recs; a.B = B; a.g = g; a.zbuf = zbuf; a.visible = visible;
    a.zstride = 1; a.nz = n;
    constexpr int PTS = 4096;
    const int64_t chunks8 = ((n + PTS - 1) / PTS + 7) / 8 * 8;
    if (chunks8 * B > 0x7fffffff) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_depth_kernel<true>, dim3((unsigned)(chunks8 * B)), dim3(PCL_BLOCK), 0, s, a, PTS);
    PCL_LAUNCH_CHECK();
    return 0;
}
