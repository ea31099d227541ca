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

#include "pcl_gd_device.h"
#include "pcl_sample_device.h"

struct PclLossArgs {
    const float* cloud;      // 6 planes of `stride` floats: x, y, z, -r, -g, -b
    int64_t n, stride;
    const void* pano;        // (H+2, W+2) texels, zero border: float4 (PCL_PANO_F32) or RGBA8 (PCL_PANO_U8)
    PclDims dims;
    const PclPoseRec* poses; // [B]
    int B;
    const uint8_t* visible;  // nullable [B][n]                                             (VIS == 1)
    const uint32_t* zbuf;    // nullable [B][Hd * Wd]: the poses' z-buffers, pcl_depth.hip    (VIS == 2)
    PclDepthGrid dgrid;
    pcl_i4* zclear;          // nullable: the z-buffers the NEXT iteration's z pass fills — this launch resets them to +inf, a slice per block
    int zclear_per_block;    //   (16-byte words per block; zclear_total in all), so that no fill launch stands between two iterations
    int64_t zclear_total;
    float* partials;         // [ngroups][nchunks][G][8] (pcl_partials_row)
    int nchunks;             // multiple of 8
    int ngroups;             // B / G
    int flip;                    // 1: every XCD walks its chunks from the last to the first (see pcl_launch_loss)
    int seg_len;                 // chunks per contiguous run of one XCD (see the mapping at the top of pcl_loss_kernel)
    int xcd_groups;              // 1: the XCDs split the pose groups (ngroups % 8 == 0), every XCD walks all chunks
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

// G poses per block, GRAD: with gradient, FMT: texel format.
// VIS: 0 = the reference's loss; 1 = a caller-supplied byte mask per point-pose multiplies into the mask; 2 = the scatter-min depth
// mask looked up where it is needed: the pose's z-buffer (built by pcl_zpass_kernel for THIS pose just before the launch) is read at
// the point's own cell while its texels are in flight, visible iff d^2 <= zmin^2 (1 + tau)^2 — no byte mask, no second projection.
// (Forcing more resident blocks per CU through __launch_bounds__ was tried: the register allocator spills, 2-4x slower.)
// FUSED: the block first finishes the PREVIOUS GD iteration for its own poses (PclFuseArgs, pcl_gd_device.h) and evaluates the
// poses that come out of it; the block of chunk 0 also stores the optimiser state.  No block waits for another one.
template <int G, bool GRAD, int VIS, int FMT, bool FUSED>
__device__ __forceinline__ void pcl_loss_body(const PclLossArgs& a, const PclFuseArgs& f)
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
    int group, chunk;
    if (a.xcd_groups) {
        // XCD <-> pose GROUPS instead of XCD <-> chunks (round 4): XCD x evaluates the groups [x * ngroups / 8, (x + 1) * ngroups / 8)
        // over ALL chunks.  For launches whose candidates read DIFFERENT panoramas — the images of one room in one chain, image i's
        // candidates a contiguous range of groups — an XCD's L2 then holds ONE image's texture (plus the whole of a small cloud)
        // instead of a slice of the cloud and every texture of the launch.
        const int gpx = a.ngroups >> 3, j = (int)(blockIdx.x >> 3), cj = j / gpx;
        group = (int)(blockIdx.x & 7) * gpx + (j - cj * gpx);
        chunk = a.flip ? a.nchunks - 1 - cj : cj;
    } else {
        const int lq = (int)(blockIdx.x >> 3) / a.ngroups;
        group = (int)(blockIdx.x >> 3) - lq * a.ngroups;
        const int lc = a.flip ? (a.nchunks >> 3) - 1 - lq : lq;                                         // chunk within the XCD
        const int run = lc / a.seg_len;
        chunk = (run * 8 + (int)(blockIdx.x & 7)) * a.seg_len + (lc - run * a.seg_len);
    }
    const int pose0 = group * G;

    // the poses this block evaluates, as SGPR pairs: straight from the pose records, or (FUSED) out of the optimiser update below
    PclPose6 P6[G];
    unsigned PANO[G][2];                          // FUSED: the poses' panorama addresses as uniform values
    __amdgpu_buffer_rsrc_t tex = pcl_tex_rsrc(a.pano, a.dims.H, a.dims.W, pcl_texel_bytes(FMT));
    // the cloud through a buffer resource too: 32-bit lane offsets + scalar plane offsets, no 64-bit address math
    __amdgpu_buffer_rsrc_t cld = __builtin_amdgcn_make_buffer_rsrc((void*)a.cloud, 0, (int)(a.stride * 6 * 4), 0x00020000);
    const int plane = (int)a.stride * 4;
    __amdgpu_buffer_rsrc_t zb = __amdgpu_buffer_rsrc_t();
    if constexpr (VIS == 2) zb = __builtin_amdgcn_make_buffer_rsrc((void*)a.zbuf, 0, (int)((unsigned)a.B * (unsigned)(a.dgrid.last + 1) * 4u), 0x00020000);

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
#ifndef PCL_NO_ROTATE_PAIR
        f2 rot_p[G][3];
#ifdef PCL_TIMING_ONLY_NO_ROTATION
        // TIMING ONLY (profiles/r05/experiments/mfma_rotation_bound.txt): what the kernel would take if q = x - t, p = R q cost the
        // VALUs nothing — the upper bound of moving the rotation to the matrix pipe.  The poses still differ (a per-pose offset from one
        // scalar, two packed adds), so nothing downstream folds away; the RESULTS ARE WRONG.
        if constexpr (G == 2) {
            const f2 oa = F2((a.poses + pose0)->t[0]), ob = F2((a.poses + pose0 + 1)->t[0]);
            rot_p[0][0] = x - oa; rot_p[0][1] = y; rot_p[0][2] = z;
            rot_p[1][0] = x - ob; rot_p[1][1] = y; rot_p[1][2] = z;
        } else
#endif
        if constexpr (G == 2) {
            if constexpr (FUSED) pcl_rotate2x2(x, y, z, P6[0], P6[1], rot_p[0][0], rot_p[0][1], rot_p[0][2], rot_p[1][0], rot_p[1][1], rot_p[1][2]);
            else pcl_rotate2x2(x, y, z, pcl_pose6(a.poses + pose0), pcl_pose6(a.poses + pose0 + 1), rot_p[0][0], rot_p[0][1], rot_p[0][2],
                               rot_p[1][0], rot_p[1][1], rot_p[1][2]);
        }
#endif
#pragma unroll
        for (int g = 0; g < G; g++) {
            const PclPoseRec* __restrict__ pr = a.poses + (pose0 + g);
            bool ok0 = valid0, ok1 = valid1;
            unsigned long long m0 = vmask0, m1 = vmask1;
            if constexpr (VIS == 1) {
                const uint8_t* vis = a.visible + (int64_t)(pose0 + g) * a.n;
                ok0 = ok0 && vis[min(i0, last)] != 0;
                ok1 = ok1 && vis[min(i1, last)] != 0;
                m0 = __builtin_amdgcn_ballot_w64(ok0); m1 = __builtin_amdgcn_ballot_w64(ok1);
            }
            // poses of several query images may share a launch: each pose record can name its own panorama
            // (same size and texel format); scalar work only
            __amdgpu_buffer_rsrc_t tg = tex;
            unsigned pano_lo = pr->pano_lo, pano_hi = pr->pano_hi;
            if constexpr (FUSED) {
                // this kernel also STORES pose records (the block of chunk 0), so the compiler reads them with vector loads
                // and would wrap every gather in a waterfall loop over a "divergent" texture descriptor (16 such loops: the
                // first fused version was 17 us slower per launch at cfg 2): the address is uniform, say so
                pano_lo = PANO[g][0]; pano_hi = PANO[g][1];
            }
            if (pano_lo | pano_hi) {
                const void* pp = (const void*)(((unsigned long long)pano_hi << 32) | (unsigned long long)pano_lo);
                tg = pcl_tex_rsrc(pp, a.dims.H, a.dims.W, pcl_texel_bytes(FMT));
            }
            PclProj<FMT> pj;
            // this pose's z-buffer (scalar offset: 32 unsigned bits, B z-buffers together stay below 4 GiB — pcl_launch_zbuffers checks)
            const int zoff = VIS == 2 ? (int)((unsigned)(pose0 + g) * (unsigned)(a.dgrid.last + 1) * 4u) : 0;
#ifndef PCL_NO_ROTATE_PAIR
            if constexpr (G == 2) {
                pj.px = rot_p[g][0]; pj.py = rot_p[g][1]; pj.pz = rot_p[g][2];
                pcl_project2_rotated<FMT, VIS == 2>(tg, a.dims, pj, zb, zoff, &a.dgrid);
            } else
#endif
            if constexpr (FUSED) pcl_project2<FMT, VIS == 2>(x, y, z, P6[g], tg, a.dims, pj, zb, zoff, &a.dgrid);
            else pcl_project2<FMT, VIS == 2>(x, y, z, pcl_pose6(pr), tg, a.dims, pj, zb, zoff, &a.dgrid);
            if constexpr (VIS == 2) {
                // visible iff d^2 <= (zmin (1 + tau))^2 = the cell (the z pass stores the scaled minimum); an empty cell holds +inf
                // (lane masks combined on the scalar unit — no short-circuit: a conditional look-up would split the loop body)
                f2 d2 = pcl_fma2(pj.pz, pj.pz, pj.rho2);
                const bool v0 = d2.x <= __uint_as_float(pj.zq0), v1 = d2.y <= __uint_as_float(pj.zq1);
                ok0 = ok0 & v0; ok1 = ok1 & v1;
                m0 = __builtin_amdgcn_ballot_w64(ok0); m1 = __builtin_amdgcn_ballot_w64(ok1);
            }
            pcl_sample2<GRAD, FMT>(pj, ncr, ncg, ncb, ok0, ok1, m0, m1, tg, a.dims, acc[g], count[g]);
        }
    };
    float bufA[2][6], bufB[2][6];
    load_step(begin, bufA);                            // (in flight while a fused block finishes the previous iteration)
    if constexpr (FUSED) {
        static_assert(PCL_BLOCK == PCL_GD_THREADS, "the fused prologue reduces with the epilogue kernel's thread layout");
        __shared__ double gd_rows[(PCL_GD_THREADS / 16) * 2 * G][4];
        __shared__ double gd_sums[G][PCL_NACC];
        __shared__ float pose_sh[G][12];
        pcl_gd_finish_group<G, true>(f.partials_in, a.nchunks, group, (int)threadIdx.x, f.st_in, f.recs_in, f.st_out, f.recs_out, chunk == 0,
                                     f.box, f.factor, f.patience, f.mode, f.loss_out, gd_rows, gd_sums, pose_sh);
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; g++) {
            float v[12];
#pragma unroll
            for (int k = 0; k < 12; k++) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pose_sh[g][k])));
            P6[g] = PclPose6{(f2){v[0], v[1]}, (f2){v[2], v[3]}, (f2){v[4], v[5]}, (f2){v[6], v[7]}, (f2){v[8], v[9]}, (f2){v[10], v[11]}};
            const PclPoseRec* pr = a.poses + (pose0 + g);
            PANO[g][0] = (unsigned)__builtin_amdgcn_readfirstlane((int)pr->pano_lo);
            PANO[g][1] = (unsigned)__builtin_amdgcn_readfirstlane((int)pr->pano_hi);
        }
    }

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
        a.partials[pcl_partials_row(a.nchunks, G, group, chunk, g) + k] = s;          // (one contiguous 32 G-byte row per block)
    }
    // the other set of z-buffers, for the next iteration's z pass: this block's slice back to +inf.  At the END of the block: in front of
    // the loop the slice's address arithmetic stayed live through it and the allocator gave up the four-wave budget (164 VGPRs).
    if constexpr (VIS == 2) {
        if (a.zclear) {
            const pcl_i4 inf4 = {0x7f800000, 0x7f800000, 0x7f800000, 0x7f800000};
            const int64_t z0 = (int64_t)blockIdx.x * a.zclear_per_block;
            for (int i = threadIdx.x; i < a.zclear_per_block; i += PCL_BLOCK)
                if (z0 + i < a.zclear_total) a.zclear[z0 + i] = inf4;
        }
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

template <int G, bool GRAD, int VIS, int FMT>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_loss_kernel(PclLossArgs a)
{
    pcl_loss_body<G, GRAD, VIS, FMT, false>(a, PclFuseArgs{});
}

template <int G, int FMT>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_loss_fused_kernel(PclLossArgs a, PclFuseArgs f)
{
    pcl_loss_body<G, true, 0, FMT, true>(a, f);
}

// ------------------------------------------------------------------------------------------------------------
// launch planning (shared with the GD loop)

// tuning knobs for tools/kbench.py (experiments build only, read once per process): PCL_G = poses per block (1/2/4), PCL_BLOCKS =
// target grid size

struct PclPlan {
    int G, ngroups, nchunks;
    int seg_len;
    int steps_base, steps_rem;
};

static PclPlan pcl_plan(int64_t n, int B)
{
    static const int g_env = PCL_KNOB(G, 0), blocks_env = PCL_KNOB(BLOCKS, 4096);
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
    // large clouds (round 5): a chunk should stay a small piece of the room whatever the grid size asks for — at 4M points on a
    // 4096 x 2048 panorama, 32 candidates all over the room, 512 chunks instead of 256: 447 -> 404 us per iteration; 10M points
    // (cfg 5): 882 -> 869; 3M points on 2048 x 1024: 280 -> 282 (tools/iter_latency.py with PCL_BLOCKS).  Up to 2M points the
    // grid-size rule alone is better (cfg 2's multi-image chains and cfg 3 lose 0.2-2 % with more chunks).
    if (n > 2000000 && want < n / 8192) want = n / 8192;
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
    static const int runs_env = PCL_KNOB(XCD_RUNS, 0);
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
int pcl_plan_nblocks(int64_t n, int B) { PclPlan p = pcl_plan(n, B); return p.nchunks * p.ngroups; }
int pcl_plan_G(int64_t n, int B) { return pcl_plan(n, B).G; }

// the same decomposition for a kernel with its own notion of a pose group (pcl_trim.hip: one block = one chunk x one
// (translation, rotation class) slot): chunks, XCD runs and balanced steps as for `ngroups` groups of two poses
void pcl_plan_for_groups(int64_t n, int ngroups, int* nchunks, int* seg_len, int* steps_base, int* steps_rem)
{
    PclPlan p = pcl_plan(n, 2 * ngroups);
    static const int chunks_env = PCL_KNOB(TRIM_CHUNKS, 0);          // experiments
    // Hundreds of slots share every chunk, so the plan above settles on its minimum of 64 chunks whatever the cloud — and a chunk of a
    // large cloud is then a large piece of the room: at 4M points on 4096 x 2048 (62k points per chunk) the 1800-pose launch took 16.3 ms,
    // with 20k points per chunk 13.3, with 8k 13.0; 10M points: 30.1 / 27.9 (20k) / 28.2 (10k) ms; at 2048 x 1024 the count hardly matters
    // (1M: 3.19 / 3.18 / 3.26 ms at 16k / 8k / 4k points per chunk; 3M: 8.28 / 8.24 / 8.59 at 47k / 20k / 6k) — round 5, tools/trim_u8p.py
    // with PCL_TRIM_CHUNKS.  Hence: at most 16k points per chunk above 2M points (the refinement's large-cloud rule in pcl_plan, 8k, is a
    // little worse there: 10M points 26.9 against 26.2 ms, 3M on 2048 x 1024 7.95 against 7.78).
    // Round 6: up to 2M points at most 8k points per chunk.  With the row-sorted work list (pcl_trim_order) a band of the texture is shared by
    // the chunks that land in it, and a chunk that spans fewer rows spills less into the neighbouring bands: 1M points, 64 -> 128 chunks:
    // memory-side 9.7 -> 6.6 GB per launch, L2 hit 0.82 -> 0.88, 3.16 -> 3.13 ms (profiles/r06: t2 against the PCL_TRIM_CHUNKS=128 run).
    int auto_chunks = 0;
    const int64_t per_chunk = n <= 2000000 ? 8192 : 16384;
    if (chunks_env < 8 && n > 64 * per_chunk) auto_chunks = (int)((n + per_chunk - 1) / per_chunk);
    if (chunks_env >= 8 || auto_chunks) {
        int64_t steps = (n + PCL_STEP - 1) / PCL_STEP, want = (((chunks_env >= 8 ? chunks_env : auto_chunks) + 7) / 8) * 8;
        if (want > steps) want = ((steps + 7) / 8) * 8;
        p.nchunks = (int)want; p.steps_base = (int)(steps / want); p.steps_rem = (int)(steps % want);
        static const int runs_env = PCL_KNOB(TRIM_RUNS, 0);
        int cpx = p.nchunks / 8, runs = runs_env < 1 || runs_env > cpx ? cpx : runs_env;
        while (cpx % runs) runs--;
        p.seg_len = cpx / runs;
    }
    *nchunks = p.nchunks; *seg_len = p.seg_len; *steps_base = p.steps_base; *steps_rem = p.steps_rem;
}

template <int G, int FMT>
static void pcl_launch_g(const PclLossArgs& a, int nblk, bool grad, int vis, hipStream_t s)
{
    if (grad) {
        if (vis == 2) hipLaunchKernelGGL((pcl_loss_kernel<G, true, 2, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
        else if (vis) hipLaunchKernelGGL((pcl_loss_kernel<G, true, 1, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((pcl_loss_kernel<G, true, 0, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
    } else {
        if (vis == 2) hipLaunchKernelGGL((pcl_loss_kernel<G, false, 2, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
        else if (vis) hipLaunchKernelGGL((pcl_loss_kernel<G, false, 1, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((pcl_loss_kernel<G, false, 0, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a);
    }
}

template <int FMT>
static void pcl_launch_f(const PclLossArgs& a, int G, int nblk, bool grad, int vis, hipStream_t s)
{
    if (G == 4) pcl_launch_g<4, FMT>(a, nblk, grad, vis, s);
    else if (G == 2) pcl_launch_g<2, FMT>(a, nblk, grad, vis, s);
    else pcl_launch_g<1, FMT>(a, nblk, grad, vis, s);
}

// Enqueue one fused loss(+grad) pass over the cloud for B poses; partials must hold pcl_partials_bytes(n, B).
template <int FMT>
static void pcl_launch_fused(const PclLossArgs& a, const PclFuseArgs& f, int G, int nblk, hipStream_t s)
{
    if (G == 4) hipLaunchKernelGGL((pcl_loss_fused_kernel<4, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a, f);
    else if (G == 2) hipLaunchKernelGGL((pcl_loss_fused_kernel<2, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a, f);
    else hipLaunchKernelGGL((pcl_loss_fused_kernel<1, FMT>), dim3(nblk), dim3(PCL_BLOCK), 0, s, a, f);
}

// `fuse` (nullable): finish the previous GD iteration in the prologue of every block (gradient pass without visibility only)
// `depth` (nullable): the poses' z-buffers and their grid — the scatter-min depth mask looked up inside the kernel (VIS == 2)
int pcl_launch_loss(const float* cloud, int64_t n, const void* pano, int pano_format, int H, int W, const PclPoseRec* poses,
                    int B, bool grad, const uint8_t* visible, float* partials, hipStream_t s, int flip, const PclFuseArgs* fuse,
                    const PclDepthLook* depth)
{
    if (pano_format != PCL_PANO_F32 && pano_format != PCL_PANO_U8 && pano_format != PCL_PANO_F16) return PCL_EINVAL;
    // 32-bit buffer addressing: 6 planes x 4 B x n must stay below 4 GiB, the padded panorama below 2 GiB
    if (n > PCL_MAX_POINTS || (int64_t)(H + 2) * (W + 2) * pcl_texel_bytes(pano_format) >= ((int64_t)1 << 31)) return PCL_EINVAL;
    PclPlan p = pcl_plan(n, B);
    PclLossArgs a;
    a.cloud = cloud; a.n = n; a.stride = pcl_cloud_stride(n);
    a.pano = pano; a.dims = pcl_make_dims(H, W, pano_format);
    a.poses = poses; a.B = B; a.visible = visible; a.partials = partials;
    a.zbuf = nullptr; a.dgrid = PclDepthGrid{};
    a.zclear = nullptr; a.zclear_per_block = 0; a.zclear_total = 0;
    if (depth) {
        if (visible || !depth->zbuf || depth->grid.Hd <= 0 || depth->grid.Wd <= 0) return PCL_EINVAL;
        if ((int64_t)B * depth->grid.Hd * depth->grid.Wd * 4 >= ((int64_t)1 << 32)) return PCL_EINVAL;     // one 32-bit buffer descriptor
        a.zbuf = depth->zbuf; a.dgrid = depth->grid;
        if (depth->zclear) {
            const int64_t nblk64 = (int64_t)p.nchunks * p.ngroups;
            a.zclear = (pcl_i4*)depth->zclear;
            a.zclear_total = depth->zclear_vec4;
            a.zclear_per_block = (int)((depth->zclear_vec4 + nblk64 - 1) / nblk64);
        }
    }
    a.nchunks = p.nchunks; a.ngroups = p.ngroups; a.seg_len = p.seg_len; a.flip = flip & 1; a.steps_base = p.steps_base; a.steps_rem = p.steps_rem;
    // bit 1 of `flip`: the poses of this launch read several panoramas (pcl_gd_hyper.images > 1) — the XCDs split the pose groups
    // instead of the chunks when they divide evenly.  Measured per iteration (tools/iter_latency.py, ITER_IMAGES=8): 167k points x 48
    // candidates of 8 images 64.0 -> 49.8 us (poses all over the room) / 51.3 -> 41.7 us (near the ground truth), 1M points x 256
    // candidates of 8 images 797 -> 761 us.  The partial sums per (group, chunk) are the same: results unchanged bit for bit.
    // PCL_XCD_GROUPS=0 / 1 forces the mapping off / on (A/B).
    static const int xg_env = PCL_KNOB(XCD_GROUPS, -1);
    a.xcd_groups = ((flip & 2) != 0 || xg_env == 1) && xg_env != 0 && p.ngroups % 8 == 0 ? 1 : 0;
    int nblk = p.nchunks * p.ngroups;
    const int vis = depth ? 2 : visible != nullptr ? 1 : 0;
    if (fuse) {
        if (!grad || vis) return PCL_EINVAL;
        if (pano_format == PCL_PANO_U8) pcl_launch_fused<PCL_PANO_U8>(a, *fuse, p.G, nblk, s);
        else if (pano_format == PCL_PANO_F16) pcl_launch_fused<PCL_PANO_F16>(a, *fuse, p.G, nblk, s);
        else pcl_launch_fused<PCL_PANO_F32>(a, *fuse, p.G, nblk, s);
        PCL_LAUNCH_CHECK();
        return 0;
    }
    if (pano_format == PCL_PANO_U8) pcl_launch_f<PCL_PANO_U8>(a, p.G, nblk, grad, vis, s);
    else if (pano_format == PCL_PANO_F16) pcl_launch_f<PCL_PANO_F16>(a, p.G, nblk, grad, vis, s);
    else pcl_launch_f<PCL_PANO_F32>(a, p.G, nblk, grad, vis, s);
    PCL_LAUNCH_CHECK();
    return 0;
}
