// pcl_color.hip — colour preprocessing of a query panorama against the point colours: color_match
// (color_utils.py:146-234, histogram matching with sin(latitude) weights) and color_mod (color_utils.py:7-65, joint
// luma equalisation in YCrCb).  Byte / histogram work: every kernel is a streaming pass (HBM-bound) with LDS-private
// integer histograms, so results do not depend on the order blocks run in.
//
// color_match, restated for 256-level panoramas (k/255, as decoded from an image file):
//   per channel: W[b] = sum of sin-weights of the non-black pixels at level b, x[i] = cumsum(W)[i] / total,
//   table[i] = interp(x[i]; template CDF), out(pixel at level b) = table[rank(b)] with rank(b) = #occupied levels < b
//   (the reference indexes the table by the rank among the DISTINCT pixel values, see oracle/color.py).
//   The template CDF is never materialised: the point colours are sorted once per cloud (rocPRIM radix sort, the one
//   library call in this file) and a quantile is located by binary search on the count c whose float32 quotient
//   c / n first exceeds x, exactly as the reference's float32 comparison `x < cumsum(counts) / n` decides it.
#include "pcl_device.h"

#include <rocprim/device/device_radix_sort.hpp>

#define PCL_CM_LEVELS 256
// Histogram passes end with one global atomic per occupied bin per block: with thousands of small blocks those atomics
// (all on the same few hundred addresses) cost more than the streaming pass itself (measured 48 us vs 8 us for a
// 1024 x 2048 panorama).  One 1024-thread block per CU keeps 16 waves per CU in flight with 8x fewer flushes.
#define PCL_HBLOCK 1024
#define PCL_HGRID 256
#define PCL_CM_FIX 68719476736.0   // 2^36: sin-weights accumulate as 64-bit fixed point (deterministic; <= 2^23+ pixels)

struct PclColorHist {                       // zeroed per call
    unsigned long long wsum[3][PCL_CM_LEVELS];
    unsigned int count[3][PCL_CM_LEVELS];
    unsigned int not_exact;                 // some non-black pixel channel is not k/255
    unsigned int pad[3];
};

static inline size_t color_align(size_t v) { return (v + 255) & ~(size_t)255; }

// (img * 255).long() per channel: truncation toward zero
__device__ __forceinline__ int pcl_level(float v) { return (int)(v * 255.f); }

// ------------------------------------------------------------------------------------- template (per cloud)

__global__ void __launch_bounds__(PCL_BLOCK) pcl_color_planes_kernel(const float* __restrict__ rgb, int64_t n, float* __restrict__ planes)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= n) return;
    planes[i] = rgb[3 * i];
    planes[n + i] = rgb[3 * i + 1];
    planes[2 * n + i] = rgb[3 * i + 2];
}

static size_t color_sort_temp_bytes(int64_t n)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys<rocprim::default_config, const float*, float*>(nullptr, bytes, nullptr, nullptr, (size_t)n, 0, 32,
                                                                                  nullptr, false);
    return bytes;
}

extern "C" size_t pcl_color_template_bytes(int64_t n) { return n > 0 ? (size_t)n * 3 * sizeof(float) : 0; }

extern "C" size_t pcl_color_template_workspace_bytes(int64_t n)
{
    return n > 0 ? color_align((size_t)n * 3 * sizeof(float)) + color_align(color_sort_temp_bytes(n)) : 0;
}

extern "C" int pcl_color_template_build(const float* rgb, int64_t n, float* tmpl, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!rgb || !tmpl || !workspace || n <= 0 || n > PCL_MAX_POINTS) return PCL_EINVAL;
    if (workspace_bytes < pcl_color_template_workspace_bytes(n)) return PCL_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* planes = (float*)workspace;
    void* temp = (char*)workspace + color_align((size_t)n * 3 * sizeof(float));
    size_t temp_bytes = color_sort_temp_bytes(n);
    hipLaunchKernelGGL(pcl_color_planes_kernel, dim3((unsigned)((n + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0, s, rgb, n, planes);
    PCL_LAUNCH_CHECK();
    for (int c = 0; c < 3; c++) {
        hipError_t e = rocprim::radix_sort_keys(temp, temp_bytes, (const float*)(planes + (size_t)c * n), tmpl + (size_t)c * n,
                                                (size_t)n, 0, 32, s, false);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------- color_match

// Pass 1: per-level sin-weight sums and pixel counts of the non-black pixels, per channel.
__global__ void __launch_bounds__(PCL_HBLOCK) pcl_cm_hist_kernel(const float* __restrict__ img, int H, int W, PclColorHist* hist)
{
    __shared__ unsigned long long wsum[3][PCL_CM_LEVELS];
    __shared__ unsigned int count[3][PCL_CM_LEVELS];
    for (int k = threadIdx.x; k < 3 * PCL_CM_LEVELS; k += PCL_HBLOCK) { (&wsum[0][0])[k] = 0ull; (&count[0][0])[k] = 0u; }
    __syncthreads();
    const int64_t npix = (int64_t)H * W;
    bool bad = false;
    for (int64_t p = (int64_t)blockIdx.x * PCL_HBLOCK + threadIdx.x; p < npix; p += (int64_t)gridDim.x * PCL_HBLOCK) {
        float v[3] = {img[3 * p], img[3 * p + 1], img[3 * p + 2]};
        int l[3] = {pcl_level(v[0]), pcl_level(v[1]), pcl_level(v[2])};
        if ((int64_t)l[0] + l[1] + l[2] <= 0) continue;                       // color_utils.py:223
        int h = (int)(p / W);
        float wgt = sinf(((float)h / (float)H) * 3.14159265358979323846f);    // color_utils.py:216-217 (fp32)
        unsigned long long fix = (unsigned long long)((double)wgt * PCL_CM_FIX + 0.5);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            bad = bad || l[c] < 0 || l[c] >= PCL_CM_LEVELS || __fdiv_rn((float)l[c], 255.f) != v[c];
            int b = min(max(l[c], 0), PCL_CM_LEVELS - 1);
            atomicAdd(&wsum[c][b], fix);
            atomicAdd(&count[c][b], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 3 * PCL_CM_LEVELS; k += PCL_HBLOCK) {
        unsigned int n = (&count[0][0])[k];
        if (n) {
            atomicAdd(&(&hist->count[0][0])[k], n);
            atomicAdd(&(&hist->wsum[0][0])[k], (&wsum[0][0])[k]);
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&hist->not_exact, 1u);
}

// fl32(fl32(c) / fl32(n)): the reference's float32 tensor cumsum(counts) / len(template) (color_utils.py:198)
__device__ __forceinline__ float pcl_cm_quot(int64_t c, float nf) { return __fdiv_rn((float)c, nf); }

// #(s < v) and #(s <= v) for a value v that is known to sit at sorted position `pos`: gallop outwards from pos, then
// bisect — 2-3 loads when the colours are continuous (runs of length 1), O(log run) when they are quantised.
__device__ inline int64_t pcl_lower_bound_at(const float* __restrict__ s, int64_t pos, float v)
{
    int64_t hi = pos, lo = pos - 1, step = 1;                    // invariant: s[hi] == v; answer in (lo, hi]
    while (lo >= 0 && s[lo] == v) { hi = lo; lo -= step; step <<= 1; }
    if (lo < -1) lo = -1;
    while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (s[mid] < v) lo = mid; else hi = mid; }
    return hi;
}

__device__ inline int64_t pcl_upper_bound_at(const float* __restrict__ s, int64_t n, int64_t pos, float v)
{
    int64_t lo = pos, hi = pos + 1, step = 1;                    // invariant: s[lo] == v; answer in (lo, hi]
    while (hi < n && s[hi] == v) { lo = hi; hi += step; step <<= 1; }
    if (hi > n) hi = n;
    while (hi - lo > 1) { int64_t mid = (lo + hi) >> 1; if (s[mid] <= v) lo = mid; else hi = mid; }
    return hi;
}

// Pass 2 (one block per channel, one thread per level): quantile of every level, its image under the template CDF with
// the reference's wrapped interpolation (color_utils.py:158-183, period 360), and the rank-indexed lookup table.
__global__ void __launch_bounds__(PCL_CM_LEVELS) pcl_cm_table_kernel(const PclColorHist* __restrict__ hist, const float* __restrict__ tmpl,
                                                                     int64_t n, float* __restrict__ lut)
{
    const int c = blockIdx.x, b = threadIdx.x;
    const float* __restrict__ s = tmpl + (size_t)c * n;
    __shared__ double cum[PCL_CM_LEVELS];
    __shared__ int rank[PCL_CM_LEVELS];
    __shared__ float table[PCL_CM_LEVELS];
    __shared__ int maxbin;
    cum[b] = (double)hist->wsum[c][b] * (1.0 / PCL_CM_FIX);
    rank[b] = hist->count[c][b] != 0u;
    __syncthreads();
    if (b == 0) {                                   // 256-entry scans in LDS: sequential, exact in double / int
        double acc = 0.0;
        int r = 0, mb = -1;
        for (int k = 0; k < PCL_CM_LEVELS; k++) {
            acc += cum[k];
            cum[k] = acc;
            int present = rank[k];
            rank[k] = r;
            if (present) { r++; mb = k; }
        }
        maxbin = mb;
    }
    __syncthreads();
    if (maxbin < 0) { lut[c * PCL_CM_LEVELS + b] = 0.f; return; }            // no non-black pixel at all
    const float nf = (float)n;
    // x = src_quantiles[b] = cumsum[b] / cumsum[-1] in float32 (color_utils.py:195-196)
    const float x = __fdiv_rn((float)cum[b], (float)cum[maxbin]);
    // smallest count c1 in [1, n] whose quotient exceeds x  ->  `big` = the distinct template value holding sorted
    // position c1 - 1; none -> the wrapped sample past the end
    float xb, fb, xs, fs;
    if (!(pcl_cm_quot(n, nf) > x)) {
        int64_t c_first = pcl_upper_bound_at(s, n, 0, s[0]);
        xb = __fadd_rn(pcl_cm_quot(c_first, nf), 360.f); fb = s[0];
        xs = pcl_cm_quot(n, nf); fs = s[n - 1];
    } else {
        int64_t lo = 1, hi = n;
        while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (pcl_cm_quot(mid, nf) > x) hi = mid; else lo = mid + 1; }
        fb = s[lo - 1];
        int64_t c_big = pcl_upper_bound_at(s, n, lo - 1, fb), c_small = pcl_lower_bound_at(s, lo - 1, fb);
        xb = pcl_cm_quot(c_big, nf);
        if (c_small > 0) { xs = pcl_cm_quot(c_small, nf); fs = s[c_small - 1]; }
        else { xs = __fsub_rn(pcl_cm_quot(n, nf), 360.f); fs = s[n - 1]; }   // wrapped sample before the start
    }
    // ((x - xs) * fb + (xb - x) * fs) / (xb - xs), float32, no contraction (color_utils.py:181)
    float num = __fadd_rn(__fmul_rn(__fsub_rn(x, xs), fb), __fmul_rn(__fsub_rn(xb, x), fs));
    table[b] = __fdiv_rn(num, __fsub_rn(xb, xs));
    __syncthreads();
    lut[c * PCL_CM_LEVELS + b] = table[rank[b]];
}

// Pass 3: remap the non-black pixels through the table.
__global__ void __launch_bounds__(PCL_BLOCK) pcl_cm_apply_kernel(const float* __restrict__ img, int64_t npix, const float* __restrict__ lut,
                                                                 float* __restrict__ out)
{
    __shared__ float t[3 * PCL_CM_LEVELS];
    for (int k = threadIdx.x; k < 3 * PCL_CM_LEVELS; k += PCL_BLOCK) t[k] = lut[k];
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x; p < npix; p += (int64_t)gridDim.x * PCL_BLOCK) {
        float v[3] = {img[3 * p], img[3 * p + 1], img[3 * p + 2]};
        int l[3] = {pcl_level(v[0]), pcl_level(v[1]), pcl_level(v[2])};
        if ((int64_t)l[0] + l[1] + l[2] > 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) v[c] = t[c * PCL_CM_LEVELS + min(max(l[c], 0), PCL_CM_LEVELS - 1)];
        }
        out[3 * p] = v[0]; out[3 * p + 1] = v[1]; out[3 * p + 2] = v[2];
    }
}

static unsigned color_hgrid(int64_t items)
{
    int64_t blocks = (items + PCL_HBLOCK - 1) / PCL_HBLOCK;
    return (unsigned)(blocks < PCL_HGRID ? (blocks > 0 ? blocks : 1) : PCL_HGRID);
}

static unsigned color_grid(int64_t items)
{
    int64_t blocks = (items + PCL_BLOCK - 1) / PCL_BLOCK;
    return (unsigned)(blocks < 2048 ? (blocks > 0 ? blocks : 1) : 2048);    // 8 blocks per CU, grid-stride beyond
}

// [PclColorHist][3 x 4096 words: color_match's 3 x 256 table, or color_mod's cumulative table + two luma histograms]
extern "C" size_t pcl_color_workspace_bytes(void) { return color_align(sizeof(PclColorHist)) + color_align(3 * 4096 * sizeof(float)); }

extern "C" int pcl_color_match(const float* img, int H, int W, const float* tmpl, int64_t n, float* out, int32_t* not_exact,
                               void* workspace, size_t workspace_bytes, void* stream)
{
    if (!img || !tmpl || !out || !workspace || H <= 0 || W <= 0 || n <= 0 || n > PCL_MAX_POINTS) return PCL_EINVAL;
    if (workspace_bytes < pcl_color_workspace_bytes()) return PCL_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    PclColorHist* hist = (PclColorHist*)workspace;
    float* lut = (float*)((char*)workspace + color_align(sizeof(PclColorHist)));
    const int64_t npix = (int64_t)H * W;
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(PclColorHist), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pcl_cm_hist_kernel, dim3(color_hgrid(npix)), dim3(PCL_HBLOCK), 0, s, img, H, W, hist);
    hipLaunchKernelGGL(pcl_cm_table_kernel, dim3(3), dim3(PCL_CM_LEVELS), 0, s, hist, tmpl, n, lut);
    hipLaunchKernelGGL(pcl_cm_apply_kernel, dim3(color_grid(npix)), dim3(PCL_BLOCK), 0, s, img, npix, lut, out);
    PCL_LAUNCH_CHECK();
    if (not_exact) {
        e = hipMemcpyAsync(not_exact, &hist->not_exact, sizeof(int32_t), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// --------------------------------------------------------------------------------------------- color_mod
// OpenCV's 8-bit RGB <-> YCrCb (imgproc color_yuv, 14-bit fixed point, rounded descale, saturation): the published
// algorithm behind cv2.cvtColor(COLOR_RGB2YCR_CB / COLOR_YCR_CB2RGB), color_utils.py:29-32, :48, :59.

#define PCL_YUV_SHIFT 14
__device__ __forceinline__ int pcl_descale(int v) { return (v + (1 << (PCL_YUV_SHIFT - 1))) >> PCL_YUV_SHIFT; }
__device__ __forceinline__ int pcl_sat8(int v) { return min(max(v, 0), 255); }

__device__ __forceinline__ void pcl_rgb2ycrcb(int r, int g, int b, int& y, int& cr, int& cb)
{
    y = pcl_descale(r * 4899 + g * 9617 + b * 1868);
    cr = pcl_sat8(pcl_descale((r - y) * 11682 + (128 << PCL_YUV_SHIFT)));
    cb = pcl_sat8(pcl_descale((b - y) * 9241 + (128 << PCL_YUV_SHIFT)));
    y = pcl_sat8(y);
}

__device__ __forceinline__ void pcl_ycrcb2rgb(int y, int cr, int cb, int& r, int& g, int& b)
{
    cr -= 128; cb -= 128;
    r = pcl_sat8(y + pcl_descale(cr * 22987));
    g = pcl_sat8(y + pcl_descale(cr * -11698 + cb * -5636));
    b = pcl_sat8(y + pcl_descale(cb * 29049));
}

// (x * 255.).astype(np.uint8) for x in [0, 1]: truncation (wraps modulo 256 outside, like the numpy cast)
__device__ __forceinline__ int pcl_to_u8(float v) { return (int)(v * 255.f) & 255; }

// luma level of a colour: (Y / 255 * (num_bins - 1)).long()  (color_utils.py:37-38)
__device__ __forceinline__ int pcl_luma_level(float r, float g, float b, int num_bins, int& cr, int& cb)
{
    int y;
    pcl_rgb2ycrcb(pcl_to_u8(r), pcl_to_u8(g), pcl_to_u8(b), y, cr, cb);
    return (int)(__fdiv_rn((float)y, 255.f) * (float)(num_bins - 1));
}

#define PCL_MOD_MAX_BINS 4096

// Luma histogram of colours [count][3]; `masked`: skip black pixels ((v * 255).long().sum() > 0, color_utils.py:26).
__global__ void __launch_bounds__(PCL_HBLOCK) pcl_mod_hist_kernel(const float* __restrict__ col, int64_t count, int num_bins, int masked,
                                                                 unsigned int* __restrict__ hist)
{
    __shared__ unsigned int h[PCL_MOD_MAX_BINS];
    for (int k = threadIdx.x; k < num_bins; k += PCL_HBLOCK) h[k] = 0u;
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * PCL_HBLOCK + threadIdx.x; p < count; p += (int64_t)gridDim.x * PCL_HBLOCK) {
        float r = col[3 * p], g = col[3 * p + 1], b = col[3 * p + 2];
        if (masked && (int64_t)pcl_level(r) + pcl_level(g) + pcl_level(b) <= 0) continue;
        int cr, cb;
        int lev = pcl_luma_level(r, g, b, num_bins, cr, cb);
        atomicAdd(&h[min(max(lev, 0), num_bins - 1)], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < num_bins; k += PCL_HBLOCK)
        if (h[k]) atomicAdd(&hist[k], h[k]);
}

// Joint cumulative table (color_utils.py:40-44): float32 histograms added, divided by their float32 sum, prefix sums
// accumulated in double and rounded per entry (torch.cumsum on a float32 CPU tensor).
__global__ void __launch_bounds__(PCL_BLOCK) pcl_mod_table_kernel(const unsigned int* __restrict__ hist_img, const unsigned int* __restrict__ hist_rgb,
                                                                  int num_bins, float* __restrict__ cdf)
{
    __shared__ float tot[PCL_MOD_MAX_BINS];
    __shared__ unsigned long long part[PCL_BLOCK / PCL_WAVE];
    unsigned long long cnt = 0;
    for (int k = threadIdx.x; k < num_bins; k += PCL_BLOCK) {
        unsigned int a = hist_img[k], b = hist_rgb[k];
        tot[k] = __fadd_rn((float)a, (float)b);
        cnt += (unsigned long long)a + b;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor((long long)cnt, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tf = (float)(part[0] + part[1] + part[2] + part[3]);
        double acc = 0.0;
        for (int k = 0; k < num_bins; k++) {                     // sequential: the prefix sums are order dependent
            acc += (double)__fdiv_rn(tot[k], tf);
            tot[k] = (float)acc;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < num_bins; k += PCL_BLOCK) cdf[k] = tot[k];
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_mod_apply_kernel(const float* __restrict__ col, int64_t count, int num_bins, int masked,
                                                                  const float* __restrict__ cdf, float* __restrict__ out)
{
    __shared__ float t[PCL_MOD_MAX_BINS];
    for (int k = threadIdx.x; k < num_bins; k += PCL_BLOCK) t[k] = cdf[k];
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x; p < count; p += (int64_t)gridDim.x * PCL_BLOCK) {
        float r = col[3 * p], g = col[3 * p + 1], b = col[3 * p + 2];
        if (!masked || (int64_t)pcl_level(r) + pcl_level(g) + pcl_level(b) > 0) {
            int cr, cb;
            int lev = pcl_luma_level(r, g, b, num_bins, cr, cb);
            // (ycrcb * 255.).astype(np.uint8): Y from the table, Cr / Cb through k / 255 * 255 (color_utils.py:48, :59)
            int y8 = pcl_to_u8(t[min(max(lev, 0), num_bins - 1)]);
            int cr8 = pcl_to_u8(__fdiv_rn((float)cr, 255.f)), cb8 = pcl_to_u8(__fdiv_rn((float)cb, 255.f));
            int ri, gi, bi;
            pcl_ycrcb2rgb(y8, cr8, cb8, ri, gi, bi);
            r = __fdiv_rn((float)ri, 255.f); g = __fdiv_rn((float)gi, 255.f); b = __fdiv_rn((float)bi, 255.f);
        }
        out[3 * p] = r; out[3 * p + 1] = g; out[3 * p + 2] = b;
    }
}

extern "C" int pcl_color_mod(const float* img, int H, int W, const float* rgb, int64_t n, int num_bins, float* out_img, float* out_rgb,
                             void* workspace, size_t workspace_bytes, void* stream)
{
    if (!img || !rgb || !out_img || !out_rgb || !workspace || H <= 0 || W <= 0 || n <= 0 || num_bins < 2 || num_bins > PCL_MOD_MAX_BINS)
        return PCL_EINVAL;
    if (workspace_bytes < pcl_color_workspace_bytes()) return PCL_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    // second workspace area: [cdf 4096][hist_img 4096][hist_rgb 4096]
    float* cdf = (float*)((char*)workspace + color_align(sizeof(PclColorHist)));
    unsigned int* hist_img = (unsigned int*)(cdf + PCL_MOD_MAX_BINS);
    unsigned int* hist_rgb = hist_img + PCL_MOD_MAX_BINS;
    const int64_t npix = (int64_t)H * W;
    hipError_t e = hipMemsetAsync(hist_img, 0, 2 * PCL_MOD_MAX_BINS * sizeof(unsigned int), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pcl_mod_hist_kernel, dim3(color_hgrid(npix)), dim3(PCL_HBLOCK), 0, s, img, npix, num_bins, 1, hist_img);
    hipLaunchKernelGGL(pcl_mod_hist_kernel, dim3(color_hgrid(n)), dim3(PCL_HBLOCK), 0, s, rgb, n, num_bins, 0, hist_rgb);
    hipLaunchKernelGGL(pcl_mod_table_kernel, dim3(1), dim3(PCL_BLOCK), 0, s, hist_img, hist_rgb, num_bins, cdf);
    hipLaunchKernelGGL(pcl_mod_apply_kernel, dim3(color_grid(npix)), dim3(PCL_BLOCK), 0, s, img, npix, num_bins, 1, cdf, out_img);
    hipLaunchKernelGGL(pcl_mod_apply_kernel, dim3(color_grid(n)), dim3(PCL_BLOCK), 0, s, rgb, n, num_bins, 0, cdf, out_rgb);
    PCL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------- histogram / histogram_intersection
// color_utils.py:68-118 (unbatched form; the batched form is the same per image with eps in the normalisation) and
// :122-144.  The fused trimming stage (pcl_hist.hip) does not call these; they back the stand-alone functions.

// order-preserving key of a float, for atomicMax
__device__ __forceinline__ unsigned int pcl_float_key(float v)
{
    unsigned int b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_img_max_kernel(const float* __restrict__ img, int64_t count, unsigned int* maxkey)
{
    unsigned int m = 0u;
    for (int64_t p = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x; p < count; p += (int64_t)gridDim.x * PCL_BLOCK)
        m = max(m, pcl_float_key(img[p]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(maxkey, m);
}

#define PCL_HIST_LDS_BINS 4096

template <bool LDS>
__global__ void __launch_bounds__(PCL_HBLOCK) pcl_histogram_kernel(const float* __restrict__ img, const uint8_t* __restrict__ mask, int64_t npix,
                                                                  int c0, int c1, int c2, const unsigned int* __restrict__ maxkey,
                                                                  unsigned int* __restrict__ hist)
{
    __shared__ unsigned int h[LDS ? PCL_HIST_LDS_BINS : 1];
    const int nbins = c0 * c1 * c2;
    if (LDS) {
        for (int k = threadIdx.x; k < nbins; k += PCL_HBLOCK) h[k] = 0u;
        __syncthreads();
    }
    // `if tgt_img.max() <= 1: tgt_img = (tgt_img * 255).long()` (color_utils.py:88-89)
    const float scale = (*maxkey <= pcl_float_key(1.0f)) ? 255.f : 1.f;
    // bin_size = ceil(255 / channels) (color_utils.py:86)
    const int s0 = (int)ceilf(255.f / (float)c0), s1 = (int)ceilf(255.f / (float)c1), s2 = (int)ceilf(255.f / (float)c2);
    for (int64_t p = (int64_t)blockIdx.x * PCL_HBLOCK + threadIdx.x; p < npix; p += (int64_t)gridDim.x * PCL_HBLOCK) {
        if (!mask[p]) continue;
        int q0 = (int)(img[3 * p] * scale) / s0, q1 = (int)(img[3 * p + 1] * scale) / s1, q2 = (int)(img[3 * p + 2] * scale) / s2;
        int code = q0 + c0 * q1 + c0 * c1 * q2;
        if (code < 0 || code >= nbins) continue;
        if (LDS) atomicAdd(&h[code], 1u); else atomicAdd(&hist[code], 1u);
    }
    if (LDS) {
        __syncthreads();
        for (int k = threadIdx.x; k < nbins; k += PCL_HBLOCK)
            if (h[k]) atomicAdd(&hist[k], h[k]);
    }
}

// counts -> float histogram; normalize: hist / (hist.sum() + eps) (eps = 0 in the unbatched form)
__global__ void __launch_bounds__(PCL_BLOCK) pcl_histogram_finish_kernel(const unsigned int* __restrict__ hist, int nbins, int normalize,
                                                                         float eps, float* __restrict__ out)
{
    __shared__ unsigned long long part[PCL_BLOCK / PCL_WAVE];
    unsigned long long s = 0;
    for (int k = threadIdx.x; k < nbins; k += PCL_BLOCK) s += hist[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor((long long)s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    unsigned long long total = part[0] + part[1] + part[2] + part[3];
    float denom = __fadd_rn((float)total, eps);
    for (int k = threadIdx.x; k < nbins; k += PCL_BLOCK) out[k] = normalize ? __fdiv_rn((float)hist[k], denom) : (float)hist[k];
}

extern "C" size_t pcl_histogram_workspace_bytes(int c0, int c1, int c2)
{
    if (c0 <= 0 || c1 <= 0 || c2 <= 0 || (int64_t)c0 * c1 * c2 > (1 << 24)) return 0;
    return color_align(16) + color_align((size_t)c0 * c1 * c2 * sizeof(unsigned int));
}

extern "C" int pcl_histogram(const float* img, const uint8_t* mask, int64_t npix, int c0, int c1, int c2, int normalize, float eps,
                             float* hist, void* workspace, size_t workspace_bytes, void* stream)
{
    size_t need = pcl_histogram_workspace_bytes(c0, c1, c2);
    if (!img || !mask || !hist || !workspace || npix <= 0 || need == 0) return PCL_EINVAL;
    if (workspace_bytes < need) return PCL_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nbins = c0 * c1 * c2;
    unsigned int* maxkey = (unsigned int*)workspace;
    unsigned int* counts = (unsigned int*)((char*)workspace + color_align(16));
    hipError_t e = hipMemsetAsync(workspace, 0, need, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pcl_img_max_kernel, dim3(color_grid(npix * 3)), dim3(PCL_BLOCK), 0, s, img, npix * 3, maxkey);
    if (nbins <= PCL_HIST_LDS_BINS)
        hipLaunchKernelGGL((pcl_histogram_kernel<true>), dim3(color_hgrid(npix)), dim3(PCL_HBLOCK), 0, s, img, mask, npix, c0, c1, c2, maxkey, counts);
    else
        hipLaunchKernelGGL((pcl_histogram_kernel<false>), dim3(color_hgrid(npix)), dim3(PCL_HBLOCK), 0, s, img, mask, npix, c0, c1, c2, maxkey, counts);
    hipLaunchKernelGGL(pcl_histogram_finish_kernel, dim3(1), dim3(PCL_BLOCK), 0, s, counts, nbins, normalize, eps, hist);
    PCL_LAUNCH_CHECK();
    return 0;
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_hist_intersection_kernel(const float* __restrict__ a, const float* __restrict__ b, int nbins,
                                                                          float* __restrict__ out)
{
    __shared__ double part[PCL_BLOCK / PCL_WAVE];
    const float* pa = a + (int64_t)blockIdx.x * nbins;
    const float* pb = b + (int64_t)blockIdx.x * nbins;
    double s = 0.0;
    for (int k = threadIdx.x; k < nbins; k += PCL_BLOCK) s += (double)fminf(pa[k], pb[k]);
    s = pcl_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(part[0] + part[1] + part[2] + part[3]);
}

// out[i] = sum_k min(a[i][k], b[i][k]) for `batch` histogram pairs of nbins entries (color_utils.py:122-144)
extern "C" int pcl_histogram_intersection(const float* a, const float* b, int batch, int nbins, float* out, void* stream)
{
    if (!a || !b || !out || batch <= 0 || nbins <= 0) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_hist_intersection_kernel, dim3(batch), dim3(PCL_BLOCK), 0, (hipStream_t)stream, a, b, nbins, out);
    PCL_LAUNCH_CHECK();
    return 0;
}
