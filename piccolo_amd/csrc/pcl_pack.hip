// pcl_pack.hip — one-time repacking of the reference's tensors into the layouts the loss kernel streams.
//   cloud : row-major (N,3) xyz + (N,3) rgb  (localize.py:159-164)  -> 6 SoA planes x,y,z,-r,-g,-b, optionally re-ordered
//   pano  : (H,W,3) float image              (localize.py:167-170)  -> zero-bordered (H+2, W+2) RGBA float4
#include "pcl_device.h"

#include <rocprim/device/device_radix_sort.hpp>

extern "C" int pcl_abi_version(void) { return PCL_ABI_VERSION; }

#ifndef PCL_SOURCE_HASH
#define PCL_SOURCE_HASH "unstamped"
#endif
extern "C" const char* pcl_source_hash(void) { return PCL_SOURCE_HASH; }
#ifndef PCL_LIBRARY_HASH
#define PCL_LIBRARY_HASH "unstamped"
#endif
extern "C" const char* pcl_library_hash(void) { return PCL_LIBRARY_HASH; }

extern "C" const char* pcl_error_string(int code)
{
    if (code == 0) return "success";
    if (code == PCL_EINVAL) return "piccolo_hip: invalid argument";
    if (code == PCL_EWORKSPACE) return "piccolo_hip: workspace too small";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "piccolo_hip: unknown error";
}

// plane stride: n rounded up to 256 floats (1 KiB) so every plane starts on a fresh 1-KiB wave-load boundary
extern "C" int64_t pcl_cloud_stride(int64_t n) { return n <= 0 ? 0 : ((n + 255) / 256) * 256; }
extern "C" size_t pcl_cloud_bytes(int64_t n) { return (size_t)pcl_cloud_stride(n) * 6 * sizeof(float); }
extern "C" size_t pcl_pano_bytes(int H, int W, int pano_format)
{
    if (H <= 0 || W <= 0) return 0;
    if (pano_format == PCL_PANO_U8P) return (size_t)((H + 3) >> 1) * (size_t)(W + 2) * 8;       // element rows of row PAIRS
    if (pano_format == PCL_PANO_U8V) return (size_t)(H + 2) * (size_t)(W + 2) * 8;              // every texel with the one below it
    if (pano_format != PCL_PANO_F32 && pano_format != PCL_PANO_U8 && pano_format != PCL_PANO_F16) return 0;
    return (size_t)(H + 2) * (size_t)(W + 2) * (size_t)pcl_texel_bytes(pano_format);
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_cloud_pack_kernel(const float* __restrict__ xyz, const float* __restrict__ rgb,
                                                                   const int64_t* __restrict__ order, int64_t n,
                                                                   int64_t stride, float* __restrict__ cloud)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= stride) return;
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        int64_t j = order ? order[i] : i;
        v[0] = xyz[3 * j]; v[1] = xyz[3 * j + 1]; v[2] = xyz[3 * j + 2];
        v[3] = -rgb[3 * j]; v[4] = -rgb[3 * j + 1]; v[5] = -rgb[3 * j + 2];   // the loss needs c - rgb: store -rgb
    }
#pragma unroll
    for (int k = 0; k < 6; k++) cloud[k * stride + i] = v[k];
}

extern "C" int pcl_cloud_pack(const float* xyz, const float* rgb, const int64_t* order, int64_t n, float* cloud, void* stream)
{
    if (!xyz || !rgb || !cloud || n <= 0) return PCL_EINVAL;
    int64_t stride = pcl_cloud_stride(n);
    hipLaunchKernelGGL(pcl_cloud_pack_kernel, dim3((unsigned)(stride / PCL_BLOCK)), dim3(PCL_BLOCK), 0, (hipStream_t)stream,
                       xyz, rgb, order, n, stride, cloud);
    PCL_LAUNCH_CHECK();
    return 0;
}

__device__ inline uint64_t pcl_spread21(uint32_t v)
{
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_morton_kernel(const float* __restrict__ xyz, int64_t n, float lx, float ly,
                                                               float lz, float sx, float sy, float sz,
                                                               int64_t* __restrict__ keys)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float top = 2097151.f;  // 2^21 - 1
    float fx = __builtin_amdgcn_fmed3f((xyz[3 * i] - lx) * sx, 0.f, top);
    float fy = __builtin_amdgcn_fmed3f((xyz[3 * i + 1] - ly) * sy, 0.f, top);
    float fz = __builtin_amdgcn_fmed3f((xyz[3 * i + 2] - lz) * sz, 0.f, top);
    uint64_t k = pcl_spread21((uint32_t)fx) | (pcl_spread21((uint32_t)fy) << 1) | (pcl_spread21((uint32_t)fz) << 2);
    keys[i] = (int64_t)k;
}

extern "C" int pcl_morton_keys(const float* xyz, int64_t n, const float* lo, const float* hi, int64_t* keys, void* stream)
{
    if (!xyz || !lo || !hi || !keys || n <= 0) return PCL_EINVAL;
    float s[3];
    for (int k = 0; k < 3; k++) s[k] = hi[k] > lo[k] ? 2097151.f / (hi[k] - lo[k]) : 0.f;
    hipLaunchKernelGGL(pcl_morton_kernel, dim3((unsigned)((n + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0,
                       (hipStream_t)stream, xyz, n, lo[0], lo[1], lo[2], s[0], s[1], s[2], keys);
    PCL_LAUNCH_CHECK();
    return 0;
}

// ---- Morton order of a cloud, entirely on the device: bounding box -> 63-bit keys -> radix sort of (key, index) pairs.
// workspace: [box: 6 ordered-uint words][keys n][keys sorted n][iota n][rocPRIM temp]
__device__ __forceinline__ unsigned int pcl_ordered_key(float v)
{
    unsigned int b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float pcl_ordered_value(unsigned int k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_bbox_kernel(const float* __restrict__ xyz, int64_t n, unsigned int* __restrict__ box)
{
    unsigned int lo[3] = {~0u, ~0u, ~0u}, hi[3] = {0u, 0u, 0u};
    for (int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PCL_BLOCK) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            unsigned int k = pcl_ordered_key(xyz[3 * i + c]);
            lo[c] = min(lo[c], k); hi[c] = max(hi[c], k);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[c] = min(lo[c], (unsigned int)__shfl_xor((int)lo[c], o, 64));
            hi[c] = max(hi[c], (unsigned int)__shfl_xor((int)hi[c], o, 64));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&box[c], lo[c]); atomicMax(&box[3 + c], hi[c]); }
    }
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_morton_box_kernel(const float* __restrict__ xyz, int64_t n, const unsigned int* __restrict__ box,
                                                                   unsigned long long* __restrict__ keys, int64_t* __restrict__ iota)
{
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float top = 2097151.f;  // 2^21 - 1
    unsigned long long k = 0ull;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float lo = pcl_ordered_value(box[c]), hi = pcl_ordered_value(box[3 + c]);
        float sc = hi > lo ? top / (hi - lo) : 0.f;
        float f = __builtin_amdgcn_fmed3f((xyz[3 * i + c] - lo) * sc, 0.f, top);
        k |= pcl_spread21((uint32_t)f) << c;
    }
    keys[i] = k;
    iota[i] = i;
}

static size_t order_align(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t order_sort_temp_bytes(int64_t n)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<rocprim::default_config, const unsigned long long*, unsigned long long*, const int64_t*, int64_t*>(
        nullptr, bytes, nullptr, nullptr, nullptr, nullptr, (size_t)n, 0, 63, nullptr, false);
    return bytes;
}

extern "C" size_t pcl_cloud_order_workspace_bytes(int64_t n)
{
    if (n <= 0) return 0;
    return order_align(64) + 3 * order_align((size_t)n * 8) + order_align(order_sort_temp_bytes(n));
}

// order[i] = index of the point that goes to packed slot i (Morton order of xyz inside its bounding box); feed it to
// pcl_cloud_pack.  Equal keys keep their original relative order (stable LSD radix sort).
extern "C" int pcl_cloud_order(const float* xyz, int64_t n, int64_t* order, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!xyz || !order || !workspace || n <= 0 || n > PCL_MAX_POINTS) return PCL_EINVAL;
    if (workspace_bytes < pcl_cloud_order_workspace_bytes(n)) return PCL_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    unsigned int* box = (unsigned int*)ws; ws += order_align(64);
    unsigned long long* keys = (unsigned long long*)ws; ws += order_align((size_t)n * 8);
    unsigned long long* keys_sorted = (unsigned long long*)ws; ws += order_align((size_t)n * 8);
    int64_t* iota = (int64_t*)ws; ws += order_align((size_t)n * 8);
    hipError_t e = hipMemsetAsync(box, 0xff, 12, s);                          // lo = max key
    if (e == hipSuccess) e = hipMemsetAsync(box + 3, 0, 12, s);               // hi = min key
    if (e != hipSuccess) return (int)e;
    int64_t blocks = (n + PCL_BLOCK - 1) / PCL_BLOCK;
    hipLaunchKernelGGL(pcl_bbox_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(PCL_BLOCK), 0, s, xyz, n, box);
    hipLaunchKernelGGL(pcl_morton_box_kernel, dim3((unsigned)blocks), dim3(PCL_BLOCK), 0, s, xyz, n, box, keys, iota);
    PCL_LAUNCH_CHECK();
    size_t temp_bytes = order_sort_temp_bytes(n);
    e = rocprim::radix_sort_pairs(ws, temp_bytes, (const unsigned long long*)keys, keys_sorted, (const int64_t*)iota, order, (size_t)n, 0, 63,
                                  s, false);
    return e == hipSuccess ? 0 : (int)e;
}

__global__ void __launch_bounds__(PCL_BLOCK) pcl_pano_pack_kernel(const float* __restrict__ img, int H, int W,
                                                                  pcl_f4* __restrict__ pano)
{
    int Wp = W + 2, Hp = H + 2;
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= (int64_t)Wp * Hp) return;
    int yp = (int)(i / Wp), xp = (int)(i - (int64_t)yp * Wp);
    pcl_f4 v = {0.f, 0.f, 0.f, 0.f};
    if (yp >= 1 && yp <= H && xp >= 1 && xp <= W) {
        const float* s = img + ((int64_t)(yp - 1) * W + (xp - 1)) * 3;
        v.x = s[0]; v.y = s[1]; v.z = s[2];
    }
    pano[i] = v;
}

extern "C" int pcl_pano_pack(const float* img_hwc, int H, int W, float* pano, void* stream)
{
    if (!img_hwc || !pano || H <= 0 || W <= 0) return PCL_EINVAL;
    int64_t total = (int64_t)(H + 2) * (W + 2);
    hipLaunchKernelGGL(pcl_pano_pack_kernel, dim3((unsigned)((total + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0,
                       (hipStream_t)stream, img_hwc, H, W, (pcl_f4*)pano);
    PCL_LAUNCH_CHECK();
    return 0;
}

// RGBA8 texels for images that are exactly k/255 (see include/piccolo_hip.h); flags anything else.
__global__ void __launch_bounds__(PCL_BLOCK) pcl_pano_pack_u8_kernel(const float* __restrict__ img, int H, int W,
                                                                     uint32_t* __restrict__ pano, int* __restrict__ not_exact)
{
    int Wp = W + 2, Hp = H + 2;
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= (int64_t)Wp * Hp) return;
    int yp = (int)(i / Wp), xp = (int)(i - (int64_t)yp * Wp);
    uint32_t v = 0u;
    if (yp >= 1 && yp <= H && xp >= 1 && xp <= W) {
        const float* s = img + ((int64_t)(yp - 1) * W + (xp - 1)) * 3;
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float f = s[c], k = rintf(f * 255.f);
            // uint8 -> .float() / 255. is an IEEE fp32 division: the value is exact iff dividing the level back gives f
            bad = bad || !(k >= 0.f && k <= 255.f) || __fdiv_rn(k, 255.f) != f;
            v |= ((uint32_t)k & 255u) << (8 * c);
        }
        if (bad) *not_exact = 1;
    }
    pano[i] = v;
}

// RGBA8 with the rows interleaved in pairs (PCL_PANO_U8P): the texel of bordered row yp, column xp goes to 32-bit word
// ((yp >> 1) * Wp + xp) * 2 + (yp & 1).  One thread per word of the padded layout (an odd H + 2 leaves a last half-pair: zeros).
__global__ void __launch_bounds__(PCL_BLOCK) pcl_pano_pack_u8p_kernel(const float* __restrict__ img, int H, int W,
                                                                      uint32_t* __restrict__ pano, int* __restrict__ not_exact)
{
    const int Wp = W + 2, pairs = (H + 3) >> 1;
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= (int64_t)Wp * pairs * 2) return;
    const int yp = (int)(i / Wp), xp = (int)(i - (int64_t)yp * Wp);
    uint32_t v = 0u;
    if (yp >= 1 && yp <= H && xp >= 1 && xp <= W) {
        const float* s = img + ((int64_t)(yp - 1) * W + (xp - 1)) * 3;
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float f = s[c], k = rintf(f * 255.f);
            bad = bad || !(k >= 0.f && k <= 255.f) || __fdiv_rn(k, 255.f) != f;
            v |= ((uint32_t)k & 255u) << (8 * c);
        }
        if (bad) *not_exact = 1;
    }
    pano[((int64_t)(yp >> 1) * Wp + xp) * 2 + (yp & 1)] = v;
}

// RGBA8 in vertical pairs (PCL_PANO_U8V): element (xp, yp) = texel (xp, yp), texel (xp, yp + 1) of the bordered image.
__global__ void __launch_bounds__(PCL_BLOCK) pcl_pano_pack_u8v_kernel(const float* __restrict__ img, int H, int W,
                                                                      pcl_i2* __restrict__ pano, int* __restrict__ not_exact)
{
    const int Wp = W + 2, Hp = H + 2;
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= (int64_t)Wp * Hp) return;
    const int yp = (int)(i / Wp), xp = (int)(i - (int64_t)yp * Wp);
    uint32_t v[2] = {0u, 0u};
    bool bad = false;
#pragma unroll
    for (int d = 0; d < 2; d++) {
        const int y = yp + d;
        if (y >= 1 && y <= H && xp >= 1 && xp <= W) {
            const float* s = img + ((int64_t)(y - 1) * W + (xp - 1)) * 3;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float f = s[c], k = rintf(f * 255.f);
                bad = bad || !(k >= 0.f && k <= 255.f) || __fdiv_rn(k, 255.f) != f;
                v[d] |= ((uint32_t)k & 255u) << (8 * c);
            }
        }
    }
    if (bad) *not_exact = 1;
    pano[i] = (pcl_i2){(int)v[0], (int)v[1]};
}

extern "C" int pcl_pano_pack_u8v(const float* img_hwc, int H, int W, uint32_t* pano, int* not_exact, void* stream)
{
    if (!img_hwc || !pano || !not_exact || H <= 0 || W <= 0) return PCL_EINVAL;
    int64_t total = (int64_t)(H + 2) * (W + 2);
    hipLaunchKernelGGL(pcl_pano_pack_u8v_kernel, dim3((unsigned)((total + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0,
                       (hipStream_t)stream, img_hwc, H, W, (pcl_i2*)pano, not_exact);
    PCL_LAUNCH_CHECK();
    return 0;
}

// half4 texels holding the levels 0..255 as fp16 (exact), for the same k/255 images as RGBA8.
__global__ void __launch_bounds__(PCL_BLOCK) pcl_pano_pack_f16_kernel(const float* __restrict__ img, int H, int W,
                                                                      pcl_i2* __restrict__ pano, int* __restrict__ not_exact)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    int Wp = W + 2, Hp = H + 2;
    int64_t i = (int64_t)blockIdx.x * PCL_BLOCK + threadIdx.x;
    if (i >= (int64_t)Wp * Hp) return;
    int yp = (int)(i / Wp), xp = (int)(i - (int64_t)yp * Wp);
    float lv[3] = {0.f, 0.f, 0.f};
    if (yp >= 1 && yp <= H && xp >= 1 && xp <= W) {
        const float* s = img + ((int64_t)(yp - 1) * W + (xp - 1)) * 3;
        bool bad = false;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float f = s[c], k = rintf(f * 255.f);
            bad = bad || !(k >= 0.f && k <= 255.f) || __fdiv_rn(k, 255.f) != f;
            lv[c] = k;
        }
        if (bad) *not_exact = 1;
    }
    h2 rg = {(_Float16)lv[0], (_Float16)lv[1]}, b0 = {(_Float16)lv[2], (_Float16)0.f};
    pano[i] = (pcl_i2){__builtin_bit_cast(int, rg), __builtin_bit_cast(int, b0)};
}

extern "C" int pcl_pano_pack_f16(const float* img_hwc, int H, int W, void* pano, int* not_exact, void* stream)
{
    if (!img_hwc || !pano || !not_exact || H <= 0 || W <= 0) return PCL_EINVAL;
    int64_t total = (int64_t)(H + 2) * (W + 2);
    hipLaunchKernelGGL(pcl_pano_pack_f16_kernel, dim3((unsigned)((total + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0,
                       (hipStream_t)stream, img_hwc, H, W, (pcl_i2*)pano, not_exact);
    PCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcl_pano_pack_u8p(const float* img_hwc, int H, int W, uint32_t* pano, int* not_exact, void* stream)
{
    if (!img_hwc || !pano || !not_exact || H <= 0 || W <= 0) return PCL_EINVAL;
    int64_t total = (int64_t)((H + 3) >> 1) * 2 * (W + 2);
    hipLaunchKernelGGL(pcl_pano_pack_u8p_kernel, dim3((unsigned)((total + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0,
                       (hipStream_t)stream, img_hwc, H, W, pano, not_exact);
    PCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcl_pano_pack_u8(const float* img_hwc, int H, int W, uint32_t* pano, int* not_exact, void* stream)
{
    if (!img_hwc || !pano || !not_exact || H <= 0 || W <= 0) return PCL_EINVAL;
    int64_t total = (int64_t)(H + 2) * (W + 2);
    hipLaunchKernelGGL(pcl_pano_pack_u8_kernel, dim3((unsigned)((total + PCL_BLOCK - 1) / PCL_BLOCK)), dim3(PCL_BLOCK), 0,
                       (hipStream_t)stream, img_hwc, H, W, pano, not_exact);
    PCL_LAUNCH_CHECK();
    return 0;
}
