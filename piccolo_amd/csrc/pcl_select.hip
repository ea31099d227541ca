// pcl_select.hip — the two selections of the initialisation stage, on the device and in one launch each:
//   utils.py:500-505  min_inds = loss_table.flatten().argsort()[:num_input];  trans[min_inds // len(rot)], rot[min_inds % len(rot)]
//   utils.py:583-586  min_inds = hist_intersect.flatten().argsort()[-num_input:], flipped;  trans[min_inds], rot[min_inds]
// The reference sorts the whole table (1320 - 1800 values) to keep 50, resp. 50 scores to keep 6.  Here one 1024-thread block per
// problem finds the n-th smallest of M 64-bit composites (value key, index) by an 8-bit radix select (exact, deterministic: the
// composites are distinct), collects the n winners, ranks them by counting and gathers the pose rows — replacing, per query image,
// torch.topk + sort + four index / arithmetic launches (~110 us of launches for 20 KB of data) by one ~10 us kernel.
//
// Order: ascending by value, ties by ascending index (a stable argsort); `largest`: descending by value, ties by DESCENDING index
// (= the last n of the stable ascending order, flipped: what utils.py:583-584 does with a stable sort).  NaN ranks last in both
// modes (the reference's scores hold no NaN, utils.py:579; a NaN loss — a pose that samples nothing — is never preferred); -0.0 and
// +0.0 are one value.  The torch.topk fallbacks for n_keep > 1024 (piccolo_amd/utils.py) are fed NaN-free copies (NaN -> +-inf on
// the losing side), so both paths share this order.
#include <stdint.h>

#include "pcl_device.h"

#define PCL_SEL_THREADS 1024
#define PCL_SEL_MAX_KEEP 1024

__device__ __forceinline__ unsigned long long pcl_sel_composite(float v, unsigned idx, int largest)
{
    unsigned u = __float_as_uint(v);
    if (v == 0.f) u = 0u;                                               // -0.0 == +0.0: one key, the tie goes to the index (as argsort / topk)
    unsigned key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // order-preserving: ascending floats -> ascending keys
    if (largest) key = ~key;
    if (v != v) key = 0xffffffffu;                                      // NaN: last
    return ((unsigned long long)key << 32) | (unsigned long long)(largest ? ~idx : idx);
}

// one block per problem.  values [nprob][M]; trans / rot rows of problem p start at p * pose_stride rows (0: shared tables).
__global__ void __launch_bounds__(PCL_SEL_THREADS) pcl_select_kernel(const float* __restrict__ values, int M, int n_keep, int largest,
                                                                     const float* __restrict__ trans, const float* __restrict__ rot,
                                                                     int rot_per_trans, long long pose_stride,
                                                                     float* __restrict__ out_trans, float* __restrict__ out_rot,
                                                                     int* __restrict__ out_idx)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned long long prefix_sh;
    __shared__ unsigned need_sh, nwin_sh, wave_tot[4];
    __shared__ unsigned long long win[PCL_SEL_MAX_KEEP];
    const int p = blockIdx.x, tid = threadIdx.x;
    const float* v = values + (long long)p * M;
    if (tid == 0) { prefix_sh = 0ull; need_sh = (unsigned)n_keep; nwin_sh = 0u; }
    __syncthreads();
    // radix select, most significant byte first: after pass b the n-th smallest composite is known down to byte 7 - b
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = prefix_sh;
        const unsigned long long high_mask = pass == 0 ? 0ull : (~0ull << (shift + 8));
        for (int i = tid; i < M; i += PCL_SEL_THREADS) {
            const unsigned long long c = pcl_sel_composite(v[i], (unsigned)i, largest);
            if ((c & high_mask) == prefix) atomicAdd(&hist[(unsigned)(c >> shift) & 255u], 1u);
        }
        __syncthreads();
        // the bucket that holds the wanted rank: inclusive scan of the 256 counts (wave scans + the four wave totals), then the
        // one thread whose bucket spans the rank publishes it (a serial walk by one thread cost 5 - 9 us per pass)
        const unsigned need = need_sh;
        unsigned h = 0u, incl = 0u;
        if (tid < 256) {
            h = hist[tid];
            incl = h;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned up = __shfl_up(incl, d, 64);
                if ((tid & 63) >= d) incl += up;
            }
            if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            unsigned before = 0u;
            for (int w = 0; w < (tid >> 6); w++) before += wave_tot[w];
            incl += before;
            const unsigned excl = incl - h;
            if (excl < need && need <= incl) {                        // exactly one bucket
                prefix_sh = prefix | ((unsigned long long)tid << shift);
                need_sh = need - excl;                                // rank of the wanted composite inside this bucket
            }
        }
        __syncthreads();
    }
    const unsigned long long cut = prefix_sh;                         // the n-th smallest composite itself
    for (int i = tid; i < M; i += PCL_SEL_THREADS) {
        const unsigned long long c = pcl_sel_composite(v[i], (unsigned)i, largest);
        if (c <= cut) win[atomicAdd(&nwin_sh, 1u)] = c;               // exactly n_keep of them (distinct composites)
    }
    __syncthreads();
    // rank by counting among the winners, then gather the rows
    if (tid < n_keep) {
        const unsigned long long c = win[tid];
        int rank = 0;
        for (int j = 0; j < n_keep; j++) rank += win[j] < c;
        unsigned idx = (unsigned)(c & 0xffffffffull);
        if (largest) idx = ~idx;
        const long long o = (long long)p * n_keep + rank;
        if (out_idx) out_idx[o] = (int)idx;
        const int it = rot_per_trans > 0 ? (int)idx / rot_per_trans : (int)idx;
        const int ir = rot_per_trans > 0 ? (int)idx % rot_per_trans : (int)idx;
        const float* t = trans + ((long long)p * pose_stride + it) * 3;
        const float* r = rot + ((long long)p * pose_stride + ir) * 3;
        for (int k = 0; k < 3; k++) { out_trans[o * 3 + k] = t[k]; out_rot[o * 3 + k] = r[k]; }
    }
}

extern "C" int pcl_select_poses(const float* values, int nprob, int M, int n_keep, int largest, const float* trans, const float* rot,
                                int rot_per_trans, int64_t pose_stride, float* out_trans, float* out_rot, int* out_idx, void* stream)
{
    if (!values || !trans || !rot || !out_trans || !out_rot) return PCL_EINVAL;
    if (nprob <= 0 || M <= 0 || n_keep <= 0 || n_keep > M || n_keep > PCL_SEL_MAX_KEEP || rot_per_trans < 0 || pose_stride < 0) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_select_kernel, dim3(nprob), dim3(PCL_SEL_THREADS), 0, (hipStream_t)stream, values, M, n_keep, largest ? 1 : 0, trans,
                       rot, rot_per_trans, (long long)pose_stride, out_trans, out_rot, out_idx);
    PCL_LAUNCH_CHECK();
    return 0;
}
