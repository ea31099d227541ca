// pcl_trim.hip — trim_input_loss (utils.py:462-507): the forward-only sampling loss of ALL K x R (translation, rotation)
// candidate pairs of the initialisation stage, with the projection shared between the rotations that differ only in yaw.
//
// The reference loops K * R forwards in Python (utils.py:484-499: p = R (x - t) -> cloud2idx -> sample_from_img -> mask ->
// mean ||c - rgb||).  The candidate rotations are a GRID (utils.py:321-360): yaw x pitch x roll, R = RZ(yaw) RY(pitch) RX(roll).
// Write R = RZ(delta) M (M: any rotation of R's class, see below) and q' = M (x - t):  p = RZ(delta) q'  has  p_z = q'_z  and
// p_x^2 + p_y^2 = q'_x^2 + q'_y^2, so
//     theta = atan2(rho, p_z + eps)                    does not depend on delta at all (panorama row and its fraction shared),
//     phi   = atan2(p_y, p_x + eps) = phi0 + delta - eps p_y / rho^2 + O(eps^2 / rho^2),   phi0 = atan2(q'_y, q'_x)
// (first-order carry of the reference's `x + 1e-6`, utils.py:48-51; eps / rho < 1e-2 for every point farther than 0.1 mm from the
// camera's vertical axis — waves that hold a nearer point evaluate phi exactly, see `tiny` below).  A block therefore rotates,
// normalises and takes BOTH atan2s once per point for up to PCL_TRIM_Y yaws of one class and per yaw only computes p_y, shifts
// the column, gathers and accumulates: 51.5 VALU instructions per point-pose on the reference's grids against 78.5 of the generic
// forward kernel (profiles/r03/t_trim_bench_*; DESIGN.md §4.6 — with that the kernel is bound by the texture path).
//
// Which rotations share: R_a and R_b differ by a yaw — R_b = RZ(delta) R_a — exactly when their THIRD ROWS are equal (RZ leaves
// the z row alone).  Equal (pitch, roll) is the obvious case; the reference's 3-DoF grid of quarter turns (24 distinct rotations out
// of 4 x 4 x 4, utils.py:338-360) falls into 6 such classes of 4 although only 8 of its 16 (pitch, roll) pairs repeat: e.g.
// (0, pi, pi) = RZ(pi) (0, 0, 0).  Classes are found ON THE DEVICE from the (R, 3) rotation table (third rows, computed in double,
// equal to 4e-7 — the table's quarter turns are fp32 roundings of pi/2, pi, ...: as matrices the grid's "equal" rotations differ by
// 1e-7, which is also how far the fp32 rotation matrix the reference multiplies with is from the ideal one), the class's first rotation
// supplies the matrix, every member its yaw relative to it (double).  The entry point takes the reference's arguments as they
// are; pairs come back as the reference's row-major loss_table[i, j] (utils.py:497).
#include <stdlib.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "pcl_sample_device.h"

#define PCL_TRIM_Y 4        // yaws evaluated per loaded point pair (partials row = PCL_TRIM_Y x {sum ||d||, count} = PCL_NACC floats)
static_assert(2 * PCL_TRIM_Y == PCL_NACC, "a trim partials row has the size of a loss partials row");

// One class of rotations (equal third row), up to PCL_TRIM_Y of its members — read through scalar loads.
struct PclTrimGroup {
    float ns[PCL_TRIM_Y], nc[PCL_TRIM_Y]; // -sin / -cos of the member's yaw relative to the class's first rotation
    float turn[PCL_TRIM_Y];               // 0.5 - yaw / 2 pi, that yaw reduced to [0, 2 pi): in (-0.5, 0.5]
    int rot_idx[PCL_TRIM_Y];              // row of the rotation table
    int ny;
    int leader;                           // row of the class's first rotation (its matrix is the slot's pose record)
    float row1[PCL_TRIM_Y][3];            // second row of the member's OWN fp32 rotation matrix (exact sign of p_y at the seam)
    int pad[2];
};
static_assert(sizeof(PclTrimGroup) == 128, "trim group record");

// blob handed back to the caller: header + PclTrimGroup[R]
struct PclTrimHeader {
    int ngroups, R;
    int pad[30];
};
static_assert(sizeof(PclTrimHeader) == 128, "trim groups header");

void pcl_plan_for_groups(int64_t n, int ngroups, int* nchunks, int* seg_len, int* steps_base, int* steps_rem);

// ---------------------------------------------------------------- classes of the rotation table (one block)
// R = RZ(yaw) RY(pitch) RX(roll) in double from the fp32 angles
__device__ inline void pcl_trim_rot_d(const float* ypr, double R[9])
{
    double sy, cy, sp, cp, sr, cr;
    sincos((double)ypr[0], &sy, &cy);
    sincos((double)ypr[1], &sp, &cp);
    sincos((double)ypr[2], &sr, &cr);
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

__global__ void __launch_bounds__(256) pcl_trim_groups_kernel(const float* __restrict__ rot, int R, PclTrimHeader* hdr, PclTrimGroup* groups)
{
    extern __shared__ double shd[];              // zrow[3 R] doubles, then leader[R], pos[R], base[R] ints
    double* zrow = shd;
    int* leader = reinterpret_cast<int*>(shd + 3 * R);
    int* pos = leader + R;
    int* base = pos + R;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        double M[9];
        pcl_trim_rot_d(rot + 3 * r, M);
        zrow[3 * r] = M[6]; zrow[3 * r + 1] = M[7]; zrow[3 * r + 2] = M[8];
    }
    __syncthreads();
    const double tol = 4e-7;
    auto same = [&](int i, int j) {
        return fabs(zrow[3 * i] - zrow[3 * j]) <= tol && fabs(zrow[3 * i + 1] - zrow[3 * j + 1]) <= tol && fabs(zrow[3 * i + 2] - zrow[3 * j + 2]) <= tol;
    };
    // leader = the first rotation with the same third row; it must be a leader itself (the relation is not transitive at the
    // edge of the tolerance: a rotation only joins a class through the class's first member)
    if (threadIdx.x == 0) {
        for (int r = 0; r < R; r++) {
            int l = r;
            for (int j = 0; j < r; j++)
                if (leader[j] == j && same(j, r)) { l = j; break; }
            leader[r] = l;
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        int p = 0;
        for (int j = leader[r]; j < r; j++) p += leader[j] == leader[r];
        pos[r] = p;
    }
    __syncthreads();
    // sub-groups of PCL_TRIM_Y yaws: classes in the order of their first rotation
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        int b = 0;
        if (leader[r] == r) {
            for (int l = 0; l < r; l++) {
                if (leader[l] != l) continue;
                int size = 0;
                for (int j = l; j < R; j++) size += leader[j] == l;
                b += (size + PCL_TRIM_Y - 1) / PCL_TRIM_Y;
            }
        }
        base[r] = b;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const int l = leader[r];
        int size = 0;
        for (int j = l; j < R; j++) size += leader[j] == l;
        const int g = base[l] + pos[r] / PCL_TRIM_Y, y = pos[r] % PCL_TRIM_Y;
        PclTrimGroup* gr = groups + g;
        // the member's yaw relative to the class's first rotation: RZ(delta) = R_member R_leader^T (double)
        double sd = 0.0, cd = 1.0, t = 0.0;
        if (l != r) {
            double A[9], B[9];
            pcl_trim_rot_d(rot + 3 * r, A);
            pcl_trim_rot_d(rot + 3 * l, B);
            const double a00 = A[0] * B[0] + A[1] * B[1] + A[2] * B[2], a10 = A[3] * B[0] + A[4] * B[1] + A[5] * B[2];
            const double delta = atan2(a10, a00);
            sincos(delta, &sd, &cd);
            const double two_pi = 6.283185307179586476925287;
            t = delta / two_pi;
            t -= floor(t);
        }
        // (what the kernel adds is 0.5 - turns: its column coordinate runs against phi; and it wants -sin, -cos: see there)
        gr->ns[y] = (float)-sd; gr->nc[y] = (float)-cd; gr->turn[y] = (float)(0.5 - t); gr->rot_idx[y] = r;
        {
            float Rm[9];
            pcl_rot_from_ypr(rot[3 * r], rot[3 * r + 1], rot[3 * r + 2], Rm);       // the matrix the generic kernel multiplies with
            gr->row1[y][0] = Rm[3]; gr->row1[y][1] = Rm[4]; gr->row1[y][2] = Rm[5];
        }
        if (y == 0) {
            const int left = size - (pos[r] / PCL_TRIM_Y) * PCL_TRIM_Y;
            gr->ny = left < PCL_TRIM_Y ? left : PCL_TRIM_Y;
            gr->leader = l;
            for (int k = gr->ny; k < PCL_TRIM_Y; k++) {
                gr->ns[k] = 0.f; gr->nc[k] = -1.f; gr->turn[k] = 0.5f; gr->rot_idx[k] = -1;
                gr->row1[k][0] = 0.f; gr->row1[k][1] = 1.f; gr->row1[k][2] = 0.f;
            }
        }
    }
    if (threadIdx.x == 0) {
        hdr->R = R;
        int total = 0;
        for (int l = 0; l < R; l++) {
            if (leader[l] != l) continue;
            int size = 0;
            for (int j = l; j < R; j++) size += leader[j] == l;
            total += (size + PCL_TRIM_Y - 1) / PCL_TRIM_Y;
        }
        hdr->ngroups = total;
    }
}

extern "C" size_t pcl_trim_groups_bytes(int R) { return R > 0 ? sizeof(PclTrimHeader) + (size_t)R * sizeof(PclTrimGroup) : 0; }

extern "C" int pcl_trim_groups(const float* rot, int R, void* groups, void* stream)
{
    if (!rot || !groups || R <= 0 || R > 1024) return PCL_EINVAL;      // (36 R bytes of LDS; the reference's grids have 8 .. 64 rotations)
    PclTrimHeader* hdr = (PclTrimHeader*)groups;
    hipLaunchKernelGGL(pcl_trim_groups_kernel, dim3(1), dim3(256), (size_t)R * (3 * sizeof(double) + 3 * sizeof(int)), (hipStream_t)stream, rot, R, hdr,
                       (PclTrimGroup*)(hdr + 1));
    PCL_LAUNCH_CHECK();
    return 0;
}

// pose records of the (group, translation) slots: R = the class's first rotation, t = trans[k]; slot = g * K + k
__global__ void pcl_trim_pose_setup_kernel(const float* __restrict__ trans, const float* __restrict__ rot, int K, const PclTrimHeader* __restrict__ hdr,
                                           const PclTrimGroup* __restrict__ groups, int ngroups, PclPoseRec* recs)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= ngroups * K) return;
    const int g = slot / K, k = slot - g * K;
    if (g >= hdr->ngroups) return;
    const int l = groups[g].leader;
    float p[6] = {trans[3 * k], trans[3 * k + 1], trans[3 * k + 2], rot[3 * l], rot[3 * l + 1], rot[3 * l + 2]};
    pcl_write_pose_rec(&recs[slot], p);
}

// ---------------------------------------------------------------- the forward pass
#define PCL_TRIM_MAX_IMAGES 32      // query images per launch (their packed panoramas' addresses travel as kernel arguments)

struct PclTrimArgs {
    const float* cloud;
    int64_t n, stride;
    const void* pano[PCL_TRIM_MAX_IMAGES];   // image i is evaluated by the blocks whose slot index falls into [i * nslots, (i + 1) * nslots)
    int nimages;
    int xcd_images;                  // 1: the XCDs split the images (nimages % 8 == 0), every XCD walks all chunks
    PclDims dims;
    const PclPoseRec* poses;         // [ngroups * K]
    const PclTrimHeader* hdr;
    const PclTrimGroup* groups;
    int K, nslots;                   // nslots = ngroups (host's count) * K, per image
    float* partials;                 // [nchunks][nimages * nslots][PCL_TRIM_Y][2]
    int nchunks, seg_len, steps_base, steps_rem;
    const int* order;                // nullable: pcl_trim_order's blob — header, then [nchunks * nslots] items = chunk * nslots + slot
};

#define PCL_STEP (2 * PCL_BLOCK)

// PCL_PANO_U8P (include/piccolo_hip.h): RGBA8 with the rows interleaved in pairs.  Round 4: the launch is bound by the texture unit's
// line rate (one L1 line lookup per cycle per CU, TA busy 0.80 at 167k points, profiles/r04/t_trim_167k_*), and with row-major
// texels every sample costs two accesses (its two rows).  Here the footprint of a sample that starts on an EVEN row is one 16-byte
// access, on an odd row two (issued for the odd lanes only): 1.5 accesses per sample, the same texture bytes.  (Round 3's vertical
// PAIRS — every texel stored with the one below it, one access per sample but twice the texture — lost at the sparse shape: 1.05 ->
// 1.25 ms, the 16.8 MB no longer lived in the L2s.)  The four dwords are sorted into the RGBA8 sampler's (top pair, bottom pair):
// the same texels, the same arithmetic, tables bit-identical to PCL_PANO_U8's.
template <> struct PclTaps<PCL_PANO_U8P> { pcl_i4 a, b; };

// gathers of one point for the panorama row offset `row` (texels; U8P: element-row offset (y0 >> 1) * Wp) and column x0
template <int FMT>
__device__ __forceinline__ void pcl_issue_taps_row(__amdgpu_buffer_rsrc_t tex, int row, int x0, int Wp, bool odd, PclTaps<FMT>& o);
template <>
__device__ __forceinline__ void pcl_issue_taps_row<PCL_PANO_U8P>(__amdgpu_buffer_rsrc_t tex, int row, int x0, int Wp, bool odd, PclTaps<PCL_PANO_U8P>& o)
{
    const int voff = (row + x0) * 8;
    o.a = __builtin_amdgcn_raw_buffer_load_b128(tex, voff, 0, 0);                    // elements (x0, k), (x0 + 1, k): rows 2k, 2k + 1
    if (odd) o.b = __builtin_amdgcn_raw_buffer_load_b128(tex, voff, Wp * 8, 0);      // the footprint's second row lives in element row k + 1
}
__device__ __forceinline__ PclTaps<PCL_PANO_U8> pcl_taps_u8_of(const PclTaps<PCL_PANO_U8P>& p, bool odd)
{
    PclTaps<PCL_PANO_U8> t;
    t.top = (pcl_i2){odd ? p.a.y : p.a.x, odd ? p.a.w : p.a.z};
    t.bot = (pcl_i2){odd ? p.b.x : p.a.y, odd ? p.b.z : p.a.w};
    return t;
}
// PCL_PANO_U8V: every texel stored with the one below it — the footprint is one 16-byte access on any row, for twice the texture
template <> struct PclTaps<PCL_PANO_U8V> { pcl_i4 a; };
template <>
__device__ __forceinline__ void pcl_issue_taps_row<PCL_PANO_U8V>(__amdgpu_buffer_rsrc_t tex, int row, int x0, int Wp, bool, PclTaps<PCL_PANO_U8V>& o)
{
    o.a = __builtin_amdgcn_raw_buffer_load_b128(tex, (row + x0) * 8, 0, 0);          // elements (x0, y0), (x0 + 1, y0)
}
__device__ __forceinline__ PclTaps<PCL_PANO_U8> pcl_taps_u8_of(const PclTaps<PCL_PANO_U8V>& p, bool)
{
    PclTaps<PCL_PANO_U8> t;
    t.top = (pcl_i2){p.a.x, p.a.z};
    t.bot = (pcl_i2){p.a.y, p.a.w};
    return t;
}
template <>
__device__ __forceinline__ void pcl_issue_taps_row<PCL_PANO_U8>(__amdgpu_buffer_rsrc_t tex, int row, int x0, int Wp, bool, PclTaps<PCL_PANO_U8>& o)
{
    int voff = (row + x0) * 4;
    o.top = pcl_texel_pair_u8(tex, voff, 0);
    o.bot = pcl_texel_pair_u8(tex, voff, Wp * 4);
}
template <>
__device__ __forceinline__ void pcl_issue_taps_row<PCL_PANO_F16>(__amdgpu_buffer_rsrc_t tex, int row, int x0, int Wp, bool, PclTaps<PCL_PANO_F16>& o)
{
    int voff = (row + x0) * 8;
    o.top = __builtin_amdgcn_raw_buffer_load_b128(tex, voff, 0, 0);
    o.bot = __builtin_amdgcn_raw_buffer_load_b128(tex, voff, Wp * 8, 0);
}
template <>
__device__ __forceinline__ void pcl_issue_taps_row<PCL_PANO_F32>(__amdgpu_buffer_rsrc_t, int row, int x0, int Wp, bool, PclTaps<PCL_PANO_F32>& o)
{
    o.voff = (row + x0) * 16;
    o.row = Wp * 16;
}

// (The U8P variant allocates 129 VGPRs — three waves per SIMD where the row-major one, 109, runs four.  Bounding it to four
//  (__launch_bounds__(PCL_BLOCK, 4): 127 VGPRs, three dwords spilled outside the loop) was measured A/B on one box, tables
//  identical: 0.834 -> 0.874 ms at 167k points, 3.336 -> 3.316 ms at 1M — more waves only queue at the texture unit.)
template <int FMT>
__global__ void __launch_bounds__(PCL_BLOCK) pcl_trim_kernel(PclTrimArgs a)
{
    // same XCD-aware mapping as pcl_loss_kernel: blocks b and b + 8 share an XCD; within an XCD the slot varies fastest, so the
    // blocks resident together read the same cloud chunk; consecutive slots are neighbouring translations of one class
    // (several query images in one launch: the slot index runs over image x (group, translation); the chunks are those of the
    //  single-image launch, so every (image, slot, chunk) partial sum — and with it the table — has the single-image launch's bits)
    const int nslots_all = a.nimages * a.nslots;
    int image, slot, slot_all, chunk;
    if (a.order && a.order[0] == 0x524f5450 && a.order[1] == a.nchunks && a.order[2] == a.nslots) {   // (a list of another cloud / grid: plain mapping)
        // ROW-SORTED work list (round 6, pcl_trim_order): XCD x evaluates its contiguous eighth of the list — the (chunk, slot) items whose
        // chunk lands in its bands of panorama rows, band after band, chunk-major inside a band — image after image, so that its L2 holds a
        // band of ONE texture for all the poses instead of every pose streaming its own region through it
        const int per_band = (a.nchunks * a.nslots) >> 3, idx = (int)(blockIdx.x >> 3);
        image = idx / per_band;
        const int item = a.order[64 + (int)(blockIdx.x & 7) * per_band + (idx - image * per_band)];
        chunk = item / a.nslots;
        slot = item - chunk * a.nslots;
        slot_all = image * a.nslots + slot;
    } else if (a.xcd_images) {
        // XCD <-> images (round 4; nimages a multiple of 8, small cloud): XCD x takes the images x, x + 8, ... over all chunks, so its
        // L2 holds one image's texture and the whole cloud instead of an eighth of the cloud and every texture of the launch
        const int ipx = a.nimages >> 3, j = (int)(blockIdx.x >> 3), r = j / ipx;
        image = (j - r * ipx) * 8 + (int)(blockIdx.x & 7);
        chunk = r / a.nslots;
        slot = r - chunk * a.nslots;
        slot_all = image * a.nslots + slot;
    } else {
        const int lq = (int)(blockIdx.x >> 3) / nslots_all;
        slot_all = (int)(blockIdx.x >> 3) - lq * nslots_all;
        image = slot_all / a.nslots; slot = slot_all - image * a.nslots;
        const int run = lq / a.seg_len;
        chunk = (run * 8 + (int)(blockIdx.x & 7)) * a.seg_len + (lq - run * a.seg_len);
    }
    const int g = slot / a.K;
    const PclTrimGroup* __restrict__ gr = a.groups + g;
    float* out = a.partials + ((int64_t)chunk * nslots_all + slot_all) * PCL_NACC;
    if (g >= a.hdr->ngroups) {                                   // (the host's group count is an upper bound)
        if (threadIdx.x < PCL_NACC) out[threadIdx.x] = 0.f;
        return;
    }
    const int ny = gr->ny;
    const PclPoseRec* __restrict__ pose = a.poses + slot;

    __amdgpu_buffer_rsrc_t tex = FMT == PCL_PANO_U8P
        ? __builtin_amdgcn_make_buffer_rsrc((void*)a.pano[image], 0, (int)((size_t)((a.dims.H + 3) >> 1) * (size_t)a.dims.Wp * 8), 0x00020000)
        : FMT == PCL_PANO_U8V ? pcl_tex_rsrc(a.pano[image], a.dims.H, a.dims.W, 8)
        : pcl_tex_rsrc(a.pano[image], a.dims.H, a.dims.W, pcl_texel_bytes(FMT));
    __amdgpu_buffer_rsrc_t cld = __builtin_amdgcn_make_buffer_rsrc((void*)a.cloud, 0, (int)(a.stride * 6 * 4), 0x00020000);
    const int plane = (int)a.stride * 4;

    f2 acc[PCL_TRIM_Y][PCL_NACC];          // only [y][0] is used by the forward-only sampler
    int count[PCL_TRIM_Y];
#pragma unroll
    for (int y = 0; y < PCL_TRIM_Y; y++) { count[y] = 0; acc[y][0] = F2(0.f); }

    int first = chunk * a.steps_base + min(chunk, a.steps_rem);
    int nsteps = a.steps_base + (chunk < a.steps_rem ? 1 : 0);
    const int begin = first * PCL_STEP;
    int end = begin + nsteps * PCL_STEP;
    if (end > (int)a.n) end = (int)a.n;
    const int last = (int)a.n - 1;

    auto load_step = [&](int base, float (&dst)[2][6]) {
        int j0 = min(base + (int)threadIdx.x, last), j1 = min(base + PCL_BLOCK + (int)threadIdx.x, last);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            dst[0][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, j0 * 4, k * plane, 0));
            dst[1][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cld, j1 * 4, k * plane, 0));
        }
    };
    const float inv_two_pi = 0.15915494309189533577f;
    auto eval_step = [&](int base, const float (&src)[2][6]) {
        const int i0 = base + threadIdx.x, i1 = i0 + PCL_BLOCK;
        const bool valid0 = i0 < end, valid1 = i1 < end;
        const unsigned long long vmask0 = __builtin_amdgcn_ballot_w64(valid0), vmask1 = __builtin_amdgcn_ballot_w64(valid1);
        f2 x = {src[0][0], src[1][0]}, y = {src[0][1], src[1][1]}, z = {src[0][2], src[1][2]};
        f2 ncr = {src[0][3], src[1][3]}, ncg = {src[0][4], src[1][4]}, ncb = {src[0][5], src[1][5]};
        // ---- shared by the yaws: q' = RY RX (x - t), rho, the row (theta) and phi0
        f2 qx, qy, qz;
        pcl_rotate2(x, y, z, pose, qx, qy, qz);
        f2 rho2 = pcl_fma2(qx, qx, qy * qy);
        f2 rg = rho2 + F2(1e-37f);
        f2 rinv = {__builtin_amdgcn_rsqf(rg.x), __builtin_amdgcn_rsqf(rg.y)};
        f2 rho = rho2 * rinv;
        f2 elev = pcl_elevation2(qz + F2(1e-6f), rho);
        const float lim_el = 0.495f * 3.14159265358979323846f;
        f2 elc = {__builtin_amdgcn_fmed3f(elev.x, -lim_el, lim_el), __builtin_amdgcn_fmed3f(elev.y, -lim_el, lim_el)};
        f2 iy = pcl_fma2(elc, F2(a.dims.k_iy), F2(a.dims.off_y));
        const int y0a = (int)iy.x, y0b = (int)iy.y;
        const bool odd0 = FMT == PCL_PANO_U8P && (y0a & 1), odd1 = FMT == PCL_PANO_U8P && (y0b & 1);
        const int row0 = (int)__umul24((unsigned)(FMT == PCL_PANO_U8P ? y0a >> 1 : y0a), (unsigned)a.dims.Wp);
        const int row1 = (int)__umul24((unsigned)(FMT == PCL_PANO_U8P ? y0b >> 1 : y0b), (unsigned)a.dims.Wp);
        const f2 fy = {__builtin_amdgcn_fractf(iy.x), __builtin_amdgcn_fractf(iy.y)};
        // The column in turns, counted the way the image runs: f = 0.5 - phi / 2 pi in [0, 1] <-> phi in [-pi, pi]; g = f - 0.5.
        // -phi0 / 2 pi, and the first-order carry of the reference's p_x + 1e-6:  d phi = -eps p_y / rho^2
        f2 u0 = pcl_atan2_2(qy, qx) * F2(-inv_two_pi);
        f2 nkk = (rinv * rinv) * F2(-1e-6f * inv_two_pi);
        // a point within 0.1 mm of the camera's vertical axis: the expansion in eps / rho no longer holds — the wave takes
        // the exact atan2(p_y, p_x + eps) for its lanes that need it (wave-uniform branch, practically never taken)
        const bool tiny0 = rho2.x < 1e-8f, tiny1 = rho2.y < 1e-8f;
        const bool any_tiny = __builtin_amdgcn_ballot_w64(tiny0 || tiny1) != 0ull;
        // phase A for every yaw of the group: column, fractions, gathers issued — phase B samples them.  (All the group's
        // gathers are in flight together: with four panorama regions per block the kernel waits on texels, not on VALU issue —
        // measured VALU busy 0.78 with the phases interleaved per yaw.)
        PclProj<FMT> pj[PCL_TRIM_Y];
#pragma unroll
        for (int yy = 0; yy < PCL_TRIM_Y; yy++) {
            if (yy >= ny) break;
            const float ns = gr->ns[yy], nc = gr->nc[yy], turn = gr->turn[yy];
            f2 npy = pcl_fma2(F2(ns), qx, F2(nc) * qy);                     // -p_y
            f2 u = pcl_fma2(nkk, npy, u0) + F2(turn);                       // 0.5 - phi / 2 pi, before wrapping into [0, 1)
            f2 gg = (f2){__builtin_amdgcn_fractf(u.x), __builtin_amdgcn_fractf(u.y)} - F2(0.5f);
            if (any_tiny) {
                f2 px = pcl_fma2(F2(-nc), qx, F2(ns) * qy);
                f2 ge = pcl_atan2_2(-npy, px + F2(1e-6f)) * F2(-inv_two_pi);
                gg = (f2){tiny0 ? ge.x : gg.x, tiny1 ? ge.y : gg.y};
            }
            // Which END of the panorama a point at phi = +-pi belongs to is decided by the sign of p_y, like atan2 decides it
            // (phi >= 0 <=> p_y >= +0 <=> g <= 0) — and the sum phi0 + yaw has forgotten on which side it started.  Whole planes of
            // a synthetic room sit exactly there under quarter-turn rotations (camera at floor height: q_z = 0), with a p_y of 1e-8
            // whose sign only the member's OWN fp32 matrix knows: lanes within 1e-5 turns of the seam take the sign of
            // p_y = R[1,:] (x - t) evaluated like the generic kernel evaluates it (wave-uniform branch; a handful of waves per
            // launch on scanned data).
            const bool seam0 = fabsf(gg.x) > 0.49999f, seam1 = fabsf(gg.y) > 0.49999f;
            if (__builtin_amdgcn_ballot_w64(seam0 || seam1) != 0ull) {
                const float r3 = gr->row1[yy][0], r4 = gr->row1[yy][1], r5 = gr->row1[yy][2];
                f2 wx = x - F2(pose->t[0]), wy = y - F2(pose->t[1]), wz = z - F2(pose->t[2]);
                f2 pye = pcl_fma2(F2(r5), wz, pcl_fma2(F2(r4), wy, wx * F2(r3)));
                gg = (f2){seam0 ? copysignf(gg.x, -pye.x) : gg.x, seam1 ? copysignf(gg.y, -pye.y) : gg.y};
            }
            // clip to |g_x| <= 0.99 (utils.py:97), pixel coordinate in the bordered texture: ix = W (g + 0.5) - 0.5 + 1
            f2 gc = {__builtin_amdgcn_fmed3f(gg.x, -0.495f, 0.495f), __builtin_amdgcn_fmed3f(gg.y, -0.495f, 0.495f)};
            f2 ix = pcl_fma2(gc, F2(2.f * a.dims.half_w), F2(a.dims.half_w + 0.5f));
            pj[yy].px = pj[yy].py = pj[yy].pz = F2(0.f);                    // (read only by the gradient variant of the sampler)
            pj[yy].fx = (f2){__builtin_amdgcn_fractf(ix.x), __builtin_amdgcn_fractf(ix.y)};
            pj[yy].fy = fy;
            pcl_issue_taps_row<FMT>(tex, row0, (int)ix.x, a.dims.Wp, odd0, pj[yy].ta);
            pcl_issue_taps_row<FMT>(tex, row1, (int)ix.y, a.dims.Wp, odd1, pj[yy].tb);
        }
#pragma unroll
        for (int yy = 0; yy < PCL_TRIM_Y; yy++) {
            if (yy >= ny) break;
            if constexpr (FMT == PCL_PANO_U8P || FMT == PCL_PANO_U8V) {
                PclProj<PCL_PANO_U8> q;
                q.px = q.py = q.pz = F2(0.f);
                q.fx = pj[yy].fx; q.fy = pj[yy].fy;
                q.ta = pcl_taps_u8_of(pj[yy].ta, odd0);
                q.tb = pcl_taps_u8_of(pj[yy].tb, odd1);
                pcl_sample2<false, PCL_PANO_U8>(q, ncr, ncg, ncb, valid0, valid1, vmask0, vmask1, tex, a.dims, acc[yy], count[yy]);
            } else
                pcl_sample2<false, FMT>(pj[yy], ncr, ncg, ncb, valid0, valid1, vmask0, vmask1, tex, a.dims, acc[yy], count[yy]);
        }
    };
    float bufA[2][6], bufB[2][6];
    load_step(begin, bufA);
    for (int base = begin; base < end; base += 2 * PCL_STEP) {
        load_step(base + PCL_STEP, bufB);
        eval_step(base, bufA);
        if (base + PCL_STEP < end) {
            load_step(base + 2 * PCL_STEP, bufA);
            eval_step(base + PCL_STEP, bufB);
        }
    }

    __shared__ float red[PCL_BLOCK / PCL_WAVE][PCL_NACC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int y = 0; y < PCL_TRIM_Y; y++) {
        float s = pcl_wave_sum(acc[y][0].x + acc[y][0].y);
        if (lane == 0) { red[wave][2 * y] = s; red[wave][2 * y + 1] = (float)count[y]; }
    }
    __syncthreads();
    if (threadIdx.x < PCL_NACC) out[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// loss_table[k][rot] = sum ||d|| / count over the chunks (fixed order, double): one thread per (slot, yaw)
__global__ void __launch_bounds__(256) pcl_trim_finish_kernel(const float* __restrict__ partials, int nchunks, int nslots, int nimages, int K, int R,
                                                              const PclTrimHeader* __restrict__ hdr, const PclTrimGroup* __restrict__ groups,
                                                              float* __restrict__ loss_table, float* __restrict__ count_table)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nimages * nslots * PCL_TRIM_Y) return;
    const int slot_all = id / PCL_TRIM_Y, y = id - slot_all * PCL_TRIM_Y;
    const int image = slot_all / nslots, slot = slot_all - image * nslots;
    const int g = slot / K, k = slot - g * K;
    // a `groups` blob of another rotation table, or more groups than the caller launched: nothing is written and the table
    // keeps the NaNs pcl_trim_loss filled it with (a stale or partly filled table must not rank)
    if (hdr->R != R || hdr->ngroups * K > nslots) return;
    if (g >= hdr->ngroups || y >= groups[g].ny) return;
    // (32 chunks' loads in flight at a time, added in chunk order: the sums of rounds 3-5 bit for bit.  The partial sums were written
    //  by other XCDs, every load is a trip to the memory side: a thread that waited for each of its 64+ loads in turn made this launch
    //  19-22 us — 2 % of the 1800-pose launch it finishes)
    double s0 = 0.0, s1 = 0.0;
    const int64_t cstride = (int64_t)nimages * nslots * PCL_NACC;
    const float* p0 = partials + (int64_t)slot_all * PCL_NACC + 2 * y;
    for (int c = 0; c < nchunks; c += 32) {
        float2 v[32];
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const int ck = c + k < nchunks ? c + k : nchunks - 1;                  // (a clamped address: the load is always issued)
            v[k] = *(const float2*)(p0 + (int64_t)ck * cstride);
        }
#pragma unroll
        for (int k = 0; k < 32; k++)
            if (c + k < nchunks) { s0 += (double)v[k].x; s1 += (double)v[k].y; }
    }
    const int j = groups[g].rot_idx[y];
    const int64_t o = ((int64_t)image * K + k) * R + j;
    loss_table[o] = (float)s0 / (float)s1;                            // 0 / 0 = NaN like the reference's mean of nothing
    if (count_table) count_table[o] = (float)s1;
}

static size_t trim_align(size_t v) { return (v + 255) & ~(size_t)255; }

// ---------------------------------------------------------------- the row-sorted work list (round 6)
// Why.  Every (chunk, slot) block streams its own region of the texture: with the blocks in (chunk, slot) order an XCD's L2 sees, for one
// chunk, the regions of hundreds of views all over the panorama — no line is touched twice before it is evicted, and the launch moved
// 16-18 GB through the memory side for 41 MB of unique data (1M points, 1800 poses, `U8V` texels).  The ROW a chunk
// lands in does not depend on the yaw (the four yaws of a slot share theta) and is known from the chunk's centroid: rank the items by
// that row, cut the ranking into `bands` equal parts (a multiple of 8), give every XCD a contiguous eighth of them and walk each band
// chunk by chunk.  An XCD's L2 then holds one band of the texture (2 MB of 17) for ALL the poses: 16.4 -> 6.6 GB per launch, L2 hit
// 0.70 -> 0.88 (profiles/r06: t1 / t2, tools/trim_pmc.py; the host-made prototype: profiles/EXPERIMENTS.md section 10).  The list depends on the cloud, the candidate grid and the texture's size —
// not on the query image: it is built ONCE per room (pcl_trim_order; two radix sorts of chunks x slots keys) and handed to every
// image's launch.  The partial sum of every (chunk, slot) is what it was: tables are bit-identical with and without the list.
struct PclTrimOrderHdr {
    int magic, nchunks, nslots, bands;
    int pad[60];
};
static_assert(sizeof(PclTrimOrderHdr) == 256, "trim order header");
#define PCL_TRIM_ORDER_MAGIC 0x524f5450

struct PclTrimSortArgs {
    const float* cloud;
    int64_t n, stride;
    const PclPoseRec* poses;
    int nslots, nchunks, steps_base, steps_rem, bands;
    float* cent;                    // [nchunks][4]: centroid of the chunk's points, count
};

__global__ void __launch_bounds__(PCL_BLOCK) pcl_trim_centroid_kernel(PclTrimSortArgs a)
{
    const int chunk = blockIdx.x;
    const int first = chunk * a.steps_base + min(chunk, a.steps_rem), nsteps = a.steps_base + (chunk < a.steps_rem ? 1 : 0);
    const int64_t begin = (int64_t)first * (2 * PCL_BLOCK);
    int64_t end = begin + (int64_t)nsteps * (2 * PCL_BLOCK);
    if (end > a.n) end = a.n;
    float sx = 0.f, sy = 0.f, sz = 0.f, cnt = 0.f;
    for (int64_t i = begin + threadIdx.x; i < end; i += PCL_BLOCK) {
        sx += a.cloud[i]; sy += a.cloud[a.stride + i]; sz += a.cloud[2 * a.stride + i]; cnt += 1.f;
    }
    __shared__ float red[PCL_BLOCK / PCL_WAVE][4];
    sx = pcl_wave_sum(sx); sy = pcl_wave_sum(sy); sz = pcl_wave_sum(sz); cnt = pcl_wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { float* r = red[threadIdx.x >> 6]; r[0] = sx; r[1] = sy; r[2] = sz; r[3] = cnt; }
    __syncthreads();
    if (threadIdx.x < 4) {
        float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const float c = red[0][3] + red[1][3] + red[2][3] + red[3][3];
        a.cent[4 * chunk + threadIdx.x] = threadIdx.x < 3 ? (c > 0.f ? v / c : 0.f) : c;
    }
}

// key of item = chunk * nslots + slot: the panorama row (16 bits) of the chunk's centroid seen from the slot's pose
__global__ void __launch_bounds__(PCL_BLOCK) pcl_trim_rowkey_kernel(PclTrimSortArgs a, unsigned int* __restrict__ key, unsigned int* __restrict__ val)
{
    const int item = blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (item >= a.nchunks * a.nslots) return;
    const int chunk = item / a.nslots, slot = item - chunk * a.nslots;
    const PclPoseRec* p = a.poses + slot;
    const float cx = a.cent[4 * chunk] - p->t[0], cy = a.cent[4 * chunk + 1] - p->t[1], cz = a.cent[4 * chunk + 2] - p->t[2];
    const float qx = p->R[0] * cx + p->R[1] * cy + p->R[2] * cz, qy = p->R[3] * cx + p->R[4] * cy + p->R[5] * cz;
    const float qz = p->R[6] * cx + p->R[7] * cy + p->R[8] * cz;
    const float el = atan2f(qz, sqrtf(qx * qx + qy * qy));                       // elevation: the row, monotonically
    const float f = (0.5f - el * 0.31830988618379067154f) * 65535.f;
    key[item] = f == f ? (unsigned int)fminf(fmaxf(f, 0.f), 65535.f) : 0u;        // (a slot beyond the table's groups has no pose: anywhere)
    val[item] = (unsigned int)item;
}

// position in the row ranking -> band; second key = band * nchunks + chunk (chunk-major inside the band)
__global__ void __launch_bounds__(PCL_BLOCK) pcl_trim_bandkey_kernel(PclTrimSortArgs a, const unsigned int* __restrict__ ranked, unsigned int* __restrict__ key)
{
    const int pos = blockIdx.x * PCL_BLOCK + threadIdx.x;
    const int M = a.nchunks * a.nslots;
    if (pos >= M) return;
    const int band = (int)(((int64_t)pos * a.bands) / M);
    key[pos] = (unsigned int)(band * a.nchunks + (int)(ranked[pos] / (unsigned int)a.nslots));
}

// the header is what makes a list VALID for the trim kernel: cleared before the list is touched, written after the last sort (a call that
// fails half way leaves a blob the kernel ignores, never a valid header over a half-written list)
__global__ void pcl_trim_order_hdr_kernel(PclTrimOrderHdr* h, int magic, int nchunks, int nslots, int bands)
{
    if (threadIdx.x == 0) { h->magic = magic; h->nchunks = nchunks; h->nslots = nslots; h->bands = bands; }
}

// temporary storage of the two sorts (16-bit row keys, 32-bit (band, chunk) keys): the larger of the two queries
static size_t trim_sort_temp_bytes(size_t M)
{
    size_t b16 = 0, b32 = 0;
    (void)rocprim::radix_sort_pairs<rocprim::default_config, const unsigned int*, unsigned int*, const unsigned int*, unsigned int*>(
        nullptr, b16, nullptr, nullptr, nullptr, nullptr, M, 0, 16, nullptr, false);
    (void)rocprim::radix_sort_pairs<rocprim::default_config, const unsigned int*, unsigned int*, const unsigned int*, unsigned int*>(
        nullptr, b32, nullptr, nullptr, nullptr, nullptr, M, 0, 32, nullptr, false);
    return b16 > b32 ? b16 : b32;
}

// bands: a multiple of 8, sized so that one band of the texture is about 2 MB (an XCD's L2 is 4 MB and also holds the cloud chunks)
static int trim_bands(int pano_format, int H, int W)
{
    const int64_t tex = (int64_t)(H + 3) * (W + 2) * (pano_format == PCL_PANO_U8P ? 4 : pano_format == PCL_PANO_U8V ? 8 : pcl_texel_bytes(pano_format));
    int per_xcd = (int)((tex / 8 + PCL_KNOB(TRIM_BAND_BYTES, 2500000) - 1) / PCL_KNOB(TRIM_BAND_BYTES, 2500000));
    if (per_xcd < 1) per_xcd = 1;
    if (per_xcd > 64) per_xcd = 64;
    return 8 * per_xcd;
}

static size_t trim_workspace_bytes(int64_t n, int K, int ngroups, int nimages)
{
    if (n <= 0 || K <= 0 || ngroups <= 0 || nimages <= 0) return 0;
    int nchunks, seg_len, sb, sr;
    pcl_plan_for_groups(n, ngroups * K, &nchunks, &seg_len, &sb, &sr);
    return trim_align((size_t)ngroups * K * sizeof(PclPoseRec)) + trim_align((size_t)nchunks * nimages * ngroups * K * PCL_NACC * sizeof(float));
}

static int trim_plan_chunks(int64_t n, int nslots)
{
    int nchunks, seg_len, sb, sr;
    pcl_plan_for_groups(n, nslots, &nchunks, &seg_len, &sb, &sr);
    return nchunks;
}

extern "C" size_t pcl_trim_order_bytes(int64_t n, int K, int ngroups)
{
    if (n <= 0 || K <= 0 || ngroups <= 0) return 0;
    return sizeof(PclTrimOrderHdr) + trim_align((size_t)trim_plan_chunks(n, ngroups * K) * ngroups * K * sizeof(int));
}

extern "C" size_t pcl_trim_order_workspace_bytes(int64_t n, int K, int ngroups)
{
    if (n <= 0 || K <= 0 || ngroups <= 0) return 0;
    const size_t nchunks = (size_t)trim_plan_chunks(n, ngroups * K), M = nchunks * ngroups * K;
    return trim_align((size_t)ngroups * K * sizeof(PclPoseRec)) + trim_align(nchunks * 4 * sizeof(float)) + 3 * trim_align(M * sizeof(int)) +
           trim_align(trim_sort_temp_bytes(M));
}

extern "C" int pcl_trim_order(const float* cloud, int64_t n, int pano_format, int H, int W, const float* trans, int K, const float* rot, int R,
                              const void* groups, int ngroups, void* order, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!cloud || !trans || !rot || !groups || !order || !workspace) return PCL_EINVAL;
    if (n <= 0 || n > PCL_MAX_POINTS || K <= 0 || R <= 0 || ngroups <= 0 || ngroups > R || H <= 0 || W <= 0) return PCL_EINVAL;
    if (pano_format < PCL_PANO_F32 || pano_format > PCL_PANO_U8V) return PCL_EINVAL;
    const int nslots = ngroups * K;
    if ((int64_t)nslots > (1 << 24)) return PCL_EINVAL;
    if (workspace_bytes < pcl_trim_order_workspace_bytes(n, K, ngroups)) return PCL_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const PclTrimHeader* hdr = (const PclTrimHeader*)groups;
    const PclTrimGroup* grs = (const PclTrimGroup*)(hdr + 1);
    PclTrimSortArgs so;
    int seg_len;
    pcl_plan_for_groups(n, nslots, &so.nchunks, &seg_len, &so.steps_base, &so.steps_rem);
    const int64_t M64 = (int64_t)so.nchunks * nslots;
    if (M64 > 0x3fffffffll) return PCL_EINVAL;
    const size_t M = (size_t)M64;
    so.cloud = cloud; so.n = n; so.stride = pcl_cloud_stride(n); so.nslots = nslots;
    so.bands = trim_bands(pano_format, H, W);
    char* w = (char*)workspace;
    PclPoseRec* recs = (PclPoseRec*)w; w += trim_align((size_t)nslots * sizeof(PclPoseRec));
    so.poses = recs;
    so.cent = (float*)w; w += trim_align((size_t)so.nchunks * 4 * sizeof(float));
    unsigned int* k0 = (unsigned int*)w; w += trim_align(M * sizeof(int));
    unsigned int* k1 = (unsigned int*)w; w += trim_align(M * sizeof(int));
    unsigned int* v0 = (unsigned int*)w; w += trim_align(M * sizeof(int));
    void* temp = w;
    size_t temp_bytes = trim_sort_temp_bytes(M);
    PclTrimOrderHdr* oh = (PclTrimOrderHdr*)order;
    unsigned int* list = (unsigned int*)(oh + 1);
    hipLaunchKernelGGL(pcl_trim_order_hdr_kernel, dim3(1), dim3(64), 0, s, oh, 0, 0, 0, 0);
    hipLaunchKernelGGL(pcl_trim_pose_setup_kernel, dim3((nslots + 255) / 256), dim3(256), 0, s, trans, rot, K, hdr, grs, ngroups, recs);
    hipLaunchKernelGGL(pcl_trim_centroid_kernel, dim3(so.nchunks), dim3(PCL_BLOCK), 0, s, so);
    const unsigned nb = (unsigned)((M + PCL_BLOCK - 1) / PCL_BLOCK);
    hipLaunchKernelGGL(pcl_trim_rowkey_kernel, dim3(nb), dim3(PCL_BLOCK), 0, s, so, k0, v0);
    PCL_LAUNCH_CHECK();
    // rank by row (16-bit keys, stable): v0 -> list
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, (const unsigned int*)k0, k1, (const unsigned int*)v0, list, M, 0, 16, s, false);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pcl_trim_bandkey_kernel, dim3(nb), dim3(PCL_BLOCK), 0, s, so, (const unsigned int*)list, k0);
    e = hipMemcpyAsync(v0, list, M * sizeof(int), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
    // (band, chunk)-major, stable: inside a cell the row ranking survives
    e = rocprim::radix_sort_pairs(temp, temp_bytes, (const unsigned int*)k0, k1, (const unsigned int*)v0, list, M, 0, 32, s, false);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pcl_trim_order_hdr_kernel, dim3(1), dim3(64), 0, s, oh, PCL_TRIM_ORDER_MAGIC, so.nchunks, nslots, so.bands);
    PCL_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t pcl_trim_loss_workspace_bytes(int64_t n, int K, int ngroups) { return trim_workspace_bytes(n, K, ngroups, 1); }
extern "C" size_t pcl_trim_loss_images_workspace_bytes(int64_t n, int K, int ngroups, int nimages)
{
    return nimages <= PCL_TRIM_MAX_IMAGES ? trim_workspace_bytes(n, K, ngroups, nimages) : 0;
}

extern "C" int pcl_trim_loss_images(const float* cloud, int64_t n, const void* const* panos_host, int nimages, int pano_format, int H, int W,
                                    const float* trans, int K, const float* rot, int R, const void* groups, int ngroups, const void* order,
                                    float* loss_tables, float* count_tables, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!cloud || !panos_host || !trans || !rot || !groups || !loss_tables || !workspace) return PCL_EINVAL;
    if (nimages <= 0 || nimages > PCL_TRIM_MAX_IMAGES) return PCL_EINVAL;
    for (int i = 0; i < nimages; i++)
        if (!panos_host[i]) return PCL_EINVAL;
    if (n <= 0 || n > PCL_MAX_POINTS || K <= 0 || R <= 0 || ngroups <= 0 || ngroups > R || H <= 0 || W <= 0) return PCL_EINVAL;
    if (pano_format != PCL_PANO_F32 && pano_format != PCL_PANO_U8 && pano_format != PCL_PANO_F16 && pano_format != PCL_PANO_U8P &&
        pano_format != PCL_PANO_U8V)
        return PCL_EINVAL;
    if ((int64_t)(H + 3) * (W + 2) * (pano_format == PCL_PANO_U8P ? 4 : pano_format == PCL_PANO_U8V ? 8 : pcl_texel_bytes(pano_format)) >= ((int64_t)1 << 31))
        return PCL_EINVAL;
    if ((int64_t)ngroups * K * nimages > (1 << 24)) return PCL_EINVAL;
    if (workspace_bytes < trim_workspace_bytes(n, K, ngroups, nimages)) return PCL_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const PclTrimHeader* hdr = (const PclTrimHeader*)groups;
    const PclTrimGroup* grs = (const PclTrimGroup*)(hdr + 1);
    const int nslots = ngroups * K;
    PclPoseRec* recs = (PclPoseRec*)workspace;
    float* partials = (float*)((char*)workspace + trim_align((size_t)nslots * sizeof(PclPoseRec)));
    // every entry starts as NaN (0xFFFFFFFF): what the finish kernel does not write — `ngroups` below the table's group count,
    // a blob built from another table — ranks last in the caller's selection instead of as whatever the buffer held
    hipError_t me = hipMemsetAsync(loss_tables, 0xFF, (size_t)nimages * K * R * sizeof(float), s);
    if (me == hipSuccess && count_tables) me = hipMemsetAsync(count_tables, 0, (size_t)nimages * K * R * sizeof(float), s);
    if (me != hipSuccess) return (int)me;
    hipLaunchKernelGGL(pcl_trim_pose_setup_kernel, dim3((nslots + 255) / 256), dim3(256), 0, s, trans, rot, K, hdr, grs, ngroups, recs);
    PclTrimArgs a;
    a.cloud = cloud; a.n = n; a.stride = pcl_cloud_stride(n);
    for (int i = 0; i < PCL_TRIM_MAX_IMAGES; i++) a.pano[i] = i < nimages ? panos_host[i] : nullptr;
    a.nimages = nimages;
    {
        // measured for 8 images per launch (tools/trim8.py, rows bit-identical): 167k points 0.776 -> 0.747 ms per image, 1M points
        // 3.094 -> 3.056.  PCL_TRIM_XCD_IMAGES=0 / 1 forces the mapping off / on (A/B).
        a.xcd_images = nimages % 8 == 0 && PCL_KNOB(TRIM_XCD_IMAGES, 1) != 0 ? 1 : 0;
    }
    a.dims = pcl_make_dims(H, W, pano_format == PCL_PANO_U8P || pano_format == PCL_PANO_U8V ? PCL_PANO_U8 : pano_format);       // (the same levels, the same constants)
    a.poses = recs; a.hdr = hdr; a.groups = grs; a.K = K; a.nslots = nslots; a.partials = partials;
    // the chunks of the SINGLE-image launch, whatever the number of images: per-image tables keep that launch's bits
    pcl_plan_for_groups(n, nslots, &a.nchunks, &a.seg_len, &a.steps_base, &a.steps_rem);
    const int64_t nblk = (int64_t)a.nchunks * nslots * nimages;
    if (nblk > 0x7fffffffll) return PCL_EINVAL;
    a.order = (const int*)order;
    if (pano_format == PCL_PANO_U8P) hipLaunchKernelGGL(pcl_trim_kernel<PCL_PANO_U8P>, dim3((unsigned)nblk), dim3(PCL_BLOCK), 0, s, a);
    else if (pano_format == PCL_PANO_U8V) hipLaunchKernelGGL(pcl_trim_kernel<PCL_PANO_U8V>, dim3((unsigned)nblk), dim3(PCL_BLOCK), 0, s, a);
    else if (pano_format == PCL_PANO_U8) hipLaunchKernelGGL(pcl_trim_kernel<PCL_PANO_U8>, dim3((unsigned)nblk), dim3(PCL_BLOCK), 0, s, a);
    else if (pano_format == PCL_PANO_F16) hipLaunchKernelGGL(pcl_trim_kernel<PCL_PANO_F16>, dim3((unsigned)nblk), dim3(PCL_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(pcl_trim_kernel<PCL_PANO_F32>, dim3((unsigned)nblk), dim3(PCL_BLOCK), 0, s, a);
    hipLaunchKernelGGL(pcl_trim_finish_kernel, dim3((nimages * nslots * PCL_TRIM_Y + 255) / 256), dim3(256), 0, s, partials, a.nchunks, nslots,
                       nimages, K, R, hdr, grs, loss_tables, count_tables);
    PCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcl_trim_loss(const float* cloud, int64_t n, const void* pano, int pano_format, int H, int W, const float* trans, int K,
                             const float* rot, int R, const void* groups, int ngroups, const void* order, float* loss_table, float* count_table,
                             void* workspace, size_t workspace_bytes, void* stream)
{
    return pcl_trim_loss_images(cloud, n, &pano, 1, pano_format, H, W, trans, K, rot, R, groups, ngroups, order, loss_table, count_table,
                                workspace, workspace_bytes, stream);
}
