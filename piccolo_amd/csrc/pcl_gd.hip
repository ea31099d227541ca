// pcl_gd.hip — second-stage reduction, chain rule to (t, yaw, pitch, roll) and the on-device optimiser epilogue.
//
// Replaces, per GD iteration of the reference: the tail of autograd (omniloc.py:47,254), B x Adam.step and
// B x ReduceLROnPlateau.step(float(loss)) — each a host sync — (omniloc.py:49-50, :256-258), the re-cat of the
// parameters (omniloc.py:260-263) and the clamp to the quantile box (omniloc.py:52-58, :265-269).
// One 256-thread block per candidate pose (four waves gather the partial sums, wave 0 runs the optimiser update in the
// same precision mix as the reference: fp32 tensors, python-double scalars).
#include <stdlib.h>

#include "pcl_gd_device.h"

#define PCL_GD_MAX_IMAGES 64     // panorama addresses per pcl_gd_set_pano_groups launch (they travel as kernel arguments)

int pcl_launch_loss(const float* cloud, int64_t n, const void* pano, int pano_format, int H, int W, const PclPoseRec* poses,
                    int B, bool grad, const uint8_t* visible, float* partials, hipStream_t s, int flip, const PclFuseArgs* fuse,
                    const PclDepthLook* depth);
size_t pcl_partials_bytes(int64_t n, int B);
int pcl_plan_nchunks(int64_t n, int B);
int pcl_plan_nblocks(int64_t n, int B);
int pcl_plan_G(int64_t n, int B);
size_t pcl_depth_zbuf_bytes(int B, int Hd, int Wd);
int pcl_launch_zbuffers(const float* cloud, int64_t n, const PclPoseRec* poses, int B, const PclDepthGrid& g, int zstride, uint32_t* zbuf, bool fill,
                        hipStream_t s);
extern "C" int pcl_depth_default(int64_t n, int H, int W, int stride_in, int* depth_h_host, int* depth_w_host, float* tau_host, int* stride_host);

// ---------------------------------------------------------------- stateless loss: pose setup + finish

__global__ void pcl_pose_setup_kernel(const float* __restrict__ trans, const float* __restrict__ rot, int B, PclPoseRec* recs)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float p[6] = {trans[3 * b], trans[3 * b + 1], trans[3 * b + 2], rot[3 * b], rot[3 * b + 1], rot[3 * b + 2]};
    pcl_write_pose_rec(&recs[b], p);
}

// one block per pose group: second-stage sums (the same fixed order as the GD epilogue), then thread g finishes pose g
template <int G>
__global__ void __launch_bounds__(PCL_GD_THREADS) pcl_finish_kernel(const float* __restrict__ partials, int nchunks, int B,
                                                                    const PclPoseRec* __restrict__ recs,
                                                                    const float* __restrict__ rot, int with_grad,
                                                                    float* __restrict__ result)
{
    __shared__ double rows_sh[(PCL_GD_THREADS / 16) * 2 * G][4];
    __shared__ double sums_sh[G][PCL_NACC];
    pcl_reduce_group<G>(partials, nchunks, blockIdx.x, threadIdx.x, rows_sh, sums_sh);
    if (threadIdx.x < G) {
        const int b = blockIdx.x * G + threadIdx.x;
        double s[PCL_NACC];
        for (int q = 0; q < PCL_NACC; q++) s[q] = sums_sh[threadIdx.x][q];
        float loss, g[6] = {0, 0, 0, 0, 0, 0};
        if (with_grad) {
            double sy, cy, sp, cp;
            sincos((double)rot[3 * b], &sy, &cy);
            sincos((double)rot[3 * b + 1], &sp, &cp);
            pcl_chain_rule(s, recs[b].R, sy, cy, sp, cp, loss, g);
        } else loss = (float)s[0] / (float)s[1];
        float* r = result + (int64_t)b * PCL_RESULT_STRIDE;
        r[0] = loss; r[1] = (float)s[1];
        for (int k = 0; k < 6; k++) r[2 + k] = g[k];
    }
}

extern "C" size_t pcl_loss_workspace_bytes(int64_t n, int B)
{
    if (n <= 0 || B <= 0) return 0;
    return (size_t)B * sizeof(PclPoseRec) + pcl_partials_bytes(n, B);
}

static int pcl_finish_launch(const float* partials, int64_t n, int B, const PclPoseRec* recs, const float* rot, int with_grad, float* result,
                             hipStream_t s)
{
    const int G = pcl_plan_G(n, B), nch = pcl_plan_nchunks(n, B);
    if (G == 4) hipLaunchKernelGGL(pcl_finish_kernel<4>, dim3(B / 4), dim3(PCL_GD_THREADS), 0, s, partials, nch, B, recs, rot, with_grad, result);
    else if (G == 2) hipLaunchKernelGGL(pcl_finish_kernel<2>, dim3(B / 2), dim3(PCL_GD_THREADS), 0, s, partials, nch, B, recs, rot, with_grad, result);
    else hipLaunchKernelGGL(pcl_finish_kernel<1>, dim3(B), dim3(PCL_GD_THREADS), 0, s, partials, nch, B, recs, rot, with_grad, result);
    PCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcl_sampling_loss(const float* cloud, int64_t n, const void* pano, int pano_format, int H, int W, const float* trans,
                                 const float* rot, int B, int with_grad, const uint8_t* visible, float* result,
                                 void* workspace, size_t workspace_bytes, void* stream)
{
    if (!cloud || !pano || !trans || !rot || !result || !workspace || n <= 0 || B <= 0 || H <= 0 || W <= 0) return PCL_EINVAL;
    if (n > PCL_MAX_POINTS) return PCL_EINVAL;               // (before anything is enqueued)
    if (workspace_bytes < pcl_loss_workspace_bytes(n, B)) return PCL_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    PclPoseRec* recs = (PclPoseRec*)workspace;
    float* partials = (float*)((char*)workspace + (size_t)B * sizeof(PclPoseRec));
    hipLaunchKernelGGL(pcl_pose_setup_kernel, dim3((B + 255) / 256), dim3(256), 0, s, trans, rot, B, recs);
    PCL_LAUNCH_CHECK();
    int rc = pcl_launch_loss(cloud, n, pano, pano_format, H, W, recs, B, with_grad != 0, visible, partials, s, 0, nullptr, nullptr);
    if (rc) return rc;
    return pcl_finish_launch(partials, n, B, recs, rot, with_grad, result, s);
}

// the grid and occluder stride a depth-masked call uses: the caller's, or pcl_depth_default's for 0 x 0 / stride 0 (a given grid
// without a stride: every point builds the z-buffer)
static int gd_depth_grid(int64_t n, int H, int W, int depth_h, int depth_w, float tau, int stride_in, PclDepthGrid* g, int* stride)
{
    if (depth_h < 0 || depth_w < 0 || (depth_h == 0) != (depth_w == 0) || !(tau >= 0.f) || stride_in < 0 || stride_in > 64) return PCL_EINVAL;
    int st = stride_in;
    if (depth_h == 0) {
        int rc = pcl_depth_default(n, H, W, stride_in, &depth_h, &depth_w, nullptr, &st);
        if (rc) return rc;
    } else if (st == 0) st = 1;
    if (stride) *stride = st;
    // (each side below 2^24: the in-kernel lookup addresses its cell with a 24-bit multiply, pcl_sample_device.h)
    if (depth_h < 2 || depth_w < 2 || depth_h >= (1 << 24) || depth_w >= (1 << 24) || (int64_t)depth_h * depth_w > ((int64_t)1 << 28)) return PCL_EINVAL;
    *g = pcl_make_depth_grid(depth_h, depth_w, tau);
    return 0;
}

static size_t gd_align(size_t v) { return (v + 255) & ~(size_t)255; }

// (the default grid follows the occluder stride — 1M points: 144 x 288 at the default stride 2, 200 x 400 at stride 1 — so the size query
//  takes the stride the call will be given: ADVICE r05, a 0 x 0 grid with an explicit stride overflowed a workspace sized without it)
extern "C" size_t pcl_loss_depth_workspace_bytes(int64_t n, int B, int H, int W, int depth_h, int depth_w, int depth_stride)
{
    PclDepthGrid g;
    if (n <= 0 || B <= 0 || gd_depth_grid(n, H, W, depth_h, depth_w, 0.f, depth_stride, &g, nullptr)) return 0;
    return gd_align(pcl_loss_workspace_bytes(n, B)) + pcl_depth_zbuf_bytes(B, g.Hd, g.Wd);
}

// pcl_sampling_loss with the scatter-min depth mask of the SAME poses multiplied into the mask: pose records, fill + z pass
// (pcl_depth.hip), loss launch that looks every point's cell up, finish.  Four launches, no byte mask.
extern "C" int pcl_sampling_loss_depth(const float* cloud, int64_t n, const void* pano, int pano_format, int H, int W, const float* trans,
                                       const float* rot, int B, int with_grad, int depth_h, int depth_w, float tau, int depth_stride, float* result,
                                       void* workspace, size_t workspace_bytes, void* stream)
{
    if (!cloud || !pano || !trans || !rot || !result || !workspace || n <= 0 || B <= 0 || H <= 0 || W <= 0) return PCL_EINVAL;
    if (n > PCL_MAX_POINTS) return PCL_EINVAL;
    PclDepthLook look;
    int zstride = 1;
    int rc = gd_depth_grid(n, H, W, depth_h, depth_w, tau, depth_stride, &look.grid, &zstride);
    if (rc) return rc;
    // sized from the grid this call RESOLVED (not from a second resolution of the arguments)
    if (workspace_bytes < gd_align(pcl_loss_workspace_bytes(n, B)) + pcl_depth_zbuf_bytes(B, look.grid.Hd, look.grid.Wd)) return PCL_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    PclPoseRec* recs = (PclPoseRec*)workspace;
    float* partials = (float*)((char*)workspace + (size_t)B * sizeof(PclPoseRec));
    uint32_t* zbuf = (uint32_t*)((char*)workspace + gd_align(pcl_loss_workspace_bytes(n, B)));
    look.zbuf = zbuf; look.zclear = nullptr; look.zclear_vec4 = 0;
    hipLaunchKernelGGL(pcl_pose_setup_kernel, dim3((B + 255) / 256), dim3(256), 0, s, trans, rot, B, recs);
    PCL_LAUNCH_CHECK();
    rc = pcl_launch_zbuffers(cloud, n, recs, B, look.grid, zstride, zbuf, true, s);
    if (rc) return rc;
    rc = pcl_launch_loss(cloud, n, pano, pano_format, H, W, recs, B, with_grad != 0, nullptr, partials, s, 0, nullptr, &look);
    if (rc) return rc;
    return pcl_finish_launch(partials, n, B, recs, rot, with_grad, result, s);
}

// ---------------------------------------------------------------- GD: state init, epilogue, run, result

// state blob = { PclGdPose[B], PclPoseRec[B] } x 2: copy 0 is the canonical one (what pcl_gd_init fills and pcl_gd_result reads);
// fused iterations ping-pong between the two (a block of iteration k + 1 reads iteration k's copy while the block of chunk 0
// writes iteration k + 1's).  The panorama addresses of the pose records are kept in both copies.
static inline size_t gd_copy_bytes(int B) { return (size_t)B * (sizeof(PclGdPose) + sizeof(PclPoseRec)); }
static inline PclGdPose* gd_poses(void* state, int B = 0, int copy = 0) { return (PclGdPose*)((char*)state + (size_t)copy * gd_copy_bytes(B)); }
static inline PclPoseRec* gd_recs(void* state, int B, int copy = 0)
{
    return (PclPoseRec*)((char*)state + (size_t)copy * gd_copy_bytes(B) + (size_t)B * sizeof(PclGdPose));
}

__global__ void pcl_gd_init_kernel(PclGdPose* st, PclPoseRec* recs, PclPoseRec* recs_shadow, const float* __restrict__ trans,
                                   const float* __restrict__ rot, int B, double lr)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    PclGdPose g;
    g.lr = lr;
    g.best = __builtin_inf();   // ReduceLROnPlateau: mode_worse = +inf
    for (int k = 0; k < 3; k++) { g.leaf[k] = trans[3 * b + k]; g.leaf[3 + k] = rot[3 * b + k]; }
    for (int k = 0; k < 6; k++) { g.fwd[k] = g.leaf[k]; g.m[k] = 0.f; g.v[k] = 0.f; }
    g.last_loss = 0.f; g.num_bad = 0; g.step = 0; g.pad = 0;
    g.beta1_pow = 1.0; g.beta2_pow = 1.0;
    recs[b].pano_lo = 0u; recs[b].pano_hi = 0u; recs[b].pad[0] = recs[b].pad[1] = 0.f;
    pcl_write_pose_rec_fast(&recs[b], g.fwd, g.sc);
    recs_shadow[b] = recs[b];
    st[b] = g;
}

template <int G>
__global__ void __launch_bounds__(PCL_GD_THREADS) pcl_gd_epilogue_kernel(const float* __restrict__ partials, int nchunks,
                                                                         const PclGdPose* st_in, const PclPoseRec* recs_in,
                                                                         PclGdPose* st_out, PclPoseRec* recs_out,
                                                                         const float* __restrict__ box, double factor,
                                                                         int patience, int mode, float* loss_out)
{
    __shared__ double rows_sh[(PCL_GD_THREADS / 16) * 2 * G][4];
    __shared__ double sums_sh[G][PCL_NACC];
    pcl_gd_finish_group<G, false>(partials, nchunks, blockIdx.x, threadIdx.x, st_in, recs_in, st_out, recs_out, true, box, factor, patience, mode,
                                  loss_out, rows_sh, sums_sh, nullptr);
}

// the epilogue's update path driven by a GIVEN loss and gradient per candidate (one block per candidate, G = 1)
__global__ void __launch_bounds__(PCL_GD_THREADS) pcl_gd_forced_kernel(const float* __restrict__ zero_partials, PclGdPose* st, PclPoseRec* recs,
                                                                       const float* __restrict__ box, double factor, int patience, int mode,
                                                                       const float* __restrict__ forced_loss, const float* __restrict__ forced_grad)
{
    __shared__ double rows_sh[(PCL_GD_THREADS / 16) * 2][4];
    __shared__ double sums_sh[1][PCL_NACC];
    pcl_gd_finish_group<1, false, true>(zero_partials, 1, blockIdx.x, threadIdx.x, st, recs, st, recs, true, box, factor, patience, mode, nullptr,
                                        rows_sh, sums_sh, nullptr, forced_loss, forced_grad);
}

__global__ void pcl_gd_result_kernel(const PclGdPose* __restrict__ st, int B, float* __restrict__ result)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float* r = result + (int64_t)b * PCL_GD_RESULT_STRIDE;
    for (int k = 0; k < 6; k++) { r[k] = st[b].fwd[k]; r[6 + k] = st[b].leaf[k]; }
    r[12] = st[b].last_loss;
    r[13] = (float)st[b].lr;
    r[14] = (float)st[b].num_bad;
    r[15] = (float)st[b].best;
}

extern "C" size_t pcl_gd_state_bytes(int B) { return B > 0 ? 2 * gd_copy_bytes(B) : 0; }

extern "C" size_t pcl_gd_workspace_bytes(int64_t n, int B, int H, int W, const pcl_gd_hyper* hyper_host)
{
    if (n <= 0 || B <= 0 || !hyper_host) return 0;
    size_t bytes = 2 * gd_align(pcl_partials_bytes(n, B));         // (two: fused iterations read one while they write the other)
    if (hyper_host->depth_mask) {
        PclDepthGrid g;
        if (gd_depth_grid(n, H, W, hyper_host->depth_h, hyper_host->depth_w, 0.f, hyper_host->depth_stride, &g, nullptr)) return 0;
        // two sets: iteration k reads set k & 1 and resets the other one for iteration k + 1's z pass (no fill launch between
        // iterations).  Scratch: every pcl_gd_run call fills set 0 itself before its first iteration, nothing persists in them.
        bytes += 2 * gd_align(pcl_depth_zbuf_bytes(B, g.Hd, g.Wd));
    }
    return bytes;
}

extern "C" int pcl_gd_init(void* state, const float* trans, const float* rot, int B, const pcl_gd_hyper* hyper_host, void* stream)
{
    if (!state || !trans || !rot || !hyper_host || B <= 0) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_gd_init_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, gd_poses(state),
                       gd_recs(state, B), gd_recs(state, B, 1), trans, rot, B, hyper_host->lr);
    PCL_LAUNCH_CHECK();
    return 0;
}

// ---- kernel timer: host-side pool of event pairs (include/piccolo_hip.h)
struct PclTimer {
    int capacity, used, stride;
    hipEvent_t* start;
    hipEvent_t* stop;
};

extern "C" void* pcl_timer_create(int capacity)
{
    if (capacity <= 0) return nullptr;
    PclTimer* t = new PclTimer;
    t->capacity = capacity; t->used = 0; t->stride = 1;
    t->start = new hipEvent_t[capacity];
    t->stop = new hipEvent_t[capacity];
    for (int i = 0; i < capacity; i++) {
        if (hipEventCreate(&t->start[i]) != hipSuccess || hipEventCreate(&t->stop[i]) != hipSuccess) {
            for (int j = 0; j <= i; j++) { (void)hipEventDestroy(t->start[j]); if (j < i) (void)hipEventDestroy(t->stop[j]); }
            delete[] t->start; delete[] t->stop; delete t;
            return nullptr;
        }
    }
    return t;
}

extern "C" void pcl_timer_destroy(void* timer)
{
    PclTimer* t = (PclTimer*)timer;
    if (!t) return;
    for (int i = 0; i < t->capacity; i++) { (void)hipEventDestroy(t->start[i]); (void)hipEventDestroy(t->stop[i]); }
    delete[] t->start; delete[] t->stop; delete t;
}

extern "C" void pcl_timer_reset(void* timer) { if (timer) ((PclTimer*)timer)->used = 0; }

extern "C" void pcl_timer_set_stride(void* timer, int stride) { if (timer && stride > 0) ((PclTimer*)timer)->stride = stride; }

extern "C" int pcl_timer_read(void* timer, double* total_ms_host, int* launches_host)
{
    PclTimer* t = (PclTimer*)timer;
    if (!t || !total_ms_host || !launches_host) return PCL_EINVAL;
    double total = 0.0;
    for (int i = 0; i < t->used; i++) {
        hipError_t e = hipEventSynchronize(t->stop[i]);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, t->start[i], t->stop[i]);
        if (e != hipSuccess) return (int)e;
        total += (double)ms;
    }
    *total_ms_host = total; *launches_host = t->used;
    return 0;
}

// largest grid that runs ONE launch per iteration: every block resident at once (256 CUs x 4 resident 256-thread blocks);
// pcl_gd_hyper.fuse < 0: never (the two-launch form, bit-identical: what the parity tests compare the fused form with)
static int gd_fuse_limit(const pcl_gd_hyper* hyper_host)
{
    if (hyper_host && hyper_host->fuse < 0) return 0;
    return PCL_KNOB(GD_FUSE_BLOCKS, 1024);                          // (experiments build: read per call)
}

extern "C" int pcl_gd_plan_hyper(int64_t n, int B, const pcl_gd_hyper* hyper_host, int* nchunks_host, int* poses_per_block_host, int* fused_host)
{
    if (n <= 0 || n > PCL_MAX_POINTS || B <= 0) return PCL_EINVAL;
    if (nchunks_host) *nchunks_host = pcl_plan_nchunks(n, B);
    if (poses_per_block_host) *poses_per_block_host = pcl_plan_G(n, B);
    // (the depth-masked loss pass reads a byte mask the fused prologue knows nothing about: pcl_gd_run keeps two launches there)
    const bool depth = hyper_host && hyper_host->depth_mask;
    if (fused_host) *fused_host = (!depth && pcl_plan_nblocks(n, B) <= gd_fuse_limit(hyper_host)) ? 1 : 0;
    return 0;
}

extern "C" int pcl_gd_plan(int64_t n, int B, int* nchunks_host, int* poses_per_block_host, int* fused_host)
{
    return pcl_gd_plan_hyper(n, B, nullptr, nchunks_host, poses_per_block_host, fused_host);
}

extern "C" int pcl_timer_calibrate(void* timer, int reps, double* pair_ms_host, void* stream)
{
    if (!timer || !pair_ms_host || reps <= 0 || reps > 4096) return PCL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t a, b;
    hipError_t e = hipEventCreate(&a);
    if (e != hipSuccess) return (int)e;
    e = hipEventCreate(&b);
    if (e != hipSuccess) { (void)hipEventDestroy(a); return (int)e; }
    float* ms = new float[reps];
    int got = 0;
    for (int i = 0; i < reps && e == hipSuccess; i++) {
        e = hipEventRecord(a, s);
        if (e == hipSuccess) e = hipEventRecord(b, s);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms[got], a, b);
        if (e == hipSuccess) got++;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (e == hipSuccess) {
        for (int i = 1; i < got; i++) {                      // insertion sort: a few dozen values
            float v = ms[i]; int j = i - 1;
            while (j >= 0 && ms[j] > v) { ms[j + 1] = ms[j]; j--; }
            ms[j + 1] = v;
        }
        *pair_ms_host = (double)ms[got / 2];
    }
    delete[] ms;
    return (int)e;
}

extern "C" int pcl_gd_run(const float* cloud, int64_t n, const void* pano, int pano_format, int H, int W, void* state, int B,
                          const float* box, const pcl_gd_hyper* hyper_host, int num_iter, float* loss_history,
                          void* workspace, size_t workspace_bytes, void* timer, void* stream)
{
    PclTimer* tm = (PclTimer*)timer;
    if (!cloud || !pano || !state || !box || !hyper_host || !workspace || n <= 0 || B <= 0 || H <= 0 || W <= 0 || num_iter < 0)
        return PCL_EINVAL;
    if (n > PCL_MAX_POINTS) return PCL_EINVAL;
    if (hyper_host->mode != PCL_GD_SEQUENTIAL && hyper_host->mode != PCL_GD_BATCH) return PCL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* partials2[2] = {(float*)workspace, (float*)((char*)workspace + gd_align(pcl_partials_bytes(n, B)))};
    PclDepthLook look;
    look.zbuf = nullptr;
    int zstride = 1;
    if (hyper_host->depth_mask) {
        int rcg = gd_depth_grid(n, H, W, hyper_host->depth_h, hyper_host->depth_w, hyper_host->depth_tau, hyper_host->depth_stride, &look.grid, &zstride);
        if (rcg) return rcg;
    }
    const size_t need = pcl_gd_workspace_bytes(n, B, H, W, hyper_host);
    if (need == 0 || workspace_bytes < need) return PCL_EWORKSPACE;
    uint32_t* zbuf2[2] = {nullptr, nullptr};
    if (hyper_host->depth_mask) {
        zbuf2[0] = (uint32_t*)((char*)workspace + 2 * gd_align(pcl_partials_bytes(n, B)));
        zbuf2[1] = (uint32_t*)((char*)zbuf2[0] + gd_align(pcl_depth_zbuf_bytes(B, look.grid.Hd, look.grid.Wd)));
        look.zclear_vec4 = (int64_t)(pcl_depth_zbuf_bytes(B, look.grid.Hd, look.grid.Wd) / 16);
    }
    const bool depth_on = zbuf2[0] != nullptr;
    static const int pingpong_env = PCL_KNOB(ZPINGPONG, 1);      // 0: a fill launch per iteration (A/B)
    const int nchunks = pcl_plan_nchunks(n, B);
    // odd iterations walk every XCD's chunks backwards: the first blocks of a launch then read the chunks the previous launch
    // finished with, still in that XCD's L2 (PCL_FLIP=0 turns it off; +0.3 % at cfg 2 in two A/B alternations on one box —
    // the first round of a launch stays 6 us slower than the later ones, so cold L2 is not what makes it slow)
    static const int flip_env = PCL_KNOB(FLIP, 1);
    // ONE launch per iteration for launches whose blocks are all resident at once (the reference's shipped 167k-point /
    // 6-candidate shape, cfg 1): there an iteration is two dependent launches of a few microseconds, and the 5 us the epilogue
    // launch costs are pure dispatch.  The loss launch of iteration k + 1 finishes iteration k in the prologue of every block
    // (PclFuseArgs) — no inter-block synchronisation, the kernel boundary is the only one; the last iteration is finished by
    // the stand-alone epilogue.  Same arithmetic in the same order: results are bit-identical to the two-launch form
    // (pcl_gd_hyper.fuse < 0: never — the form the parity tests compare this one with).
    // several panoramas in the launch AND a cloud small enough to live in every XCD's L2 next to a texture (6 MB: the reference's
    // shipped 167k points): the XCDs split the pose groups — the images — instead of the chunks (pcl_launch_loss).  Large clouds keep
    // the chunk mapping: measured at 1M points, 8 images per launch +0.3 % (3 523 -> 3 533), 5 images per launch (the driver's shape:
    // 10 groups per XCD straddling the images, every XCD walking the whole 24 MB cloud) -1.5 % (3 533 -> 3 479).
    const int xcd_bit = hyper_host->images > 1 && n * 24 <= ((int64_t)6 << 20) ? 2 : 0;
    const int fuse_blocks = gd_fuse_limit(hyper_host);
    const bool fused = !depth_on && pcl_plan_nblocks(n, B) <= fuse_blocks;
    const int G = pcl_plan_G(n, B);
    auto epilogue = [&](int it, int copy_in, float* partials) {
        const PclGdPose* si = gd_poses(state, B, copy_in);
        const PclPoseRec* ri = gd_recs(state, B, copy_in);
        PclGdPose* so = gd_poses(state, B, 0);
        PclPoseRec* ro = gd_recs(state, B, 0);
        float* lo = loss_history ? loss_history + (int64_t)it * B : nullptr;
        const double fac = hyper_host->factor;
        const int pat = (int)hyper_host->patience, mode = (int)hyper_host->mode;
        if (G == 4) hipLaunchKernelGGL(pcl_gd_epilogue_kernel<4>, dim3(B / 4), dim3(PCL_GD_THREADS), 0, s, partials, nchunks, si, ri, so, ro, box, fac, pat, mode, lo);
        else if (G == 2) hipLaunchKernelGGL(pcl_gd_epilogue_kernel<2>, dim3(B / 2), dim3(PCL_GD_THREADS), 0, s, partials, nchunks, si, ri, so, ro, box, fac, pat, mode, lo);
        else hipLaunchKernelGGL(pcl_gd_epilogue_kernel<1>, dim3(B), dim3(PCL_GD_THREADS), 0, s, partials, nchunks, si, ri, so, ro, box, fac, pat, mode, lo);
    };
    for (int it = 0; it < num_iter; it++) {
        const PclDepthLook* depth = nullptr;
        if (depth_on) {
            // the scatter-min z-buffers of the poses this iteration evaluates: z pass into set it & 1 (filled by this call before its
            // first iteration, reset by the previous iteration's loss launch afterwards); the loss launch looks them up
            const int set = pingpong_env ? (it & 1) : 0;
            int rcd = pcl_launch_zbuffers(cloud, n, gd_recs(state, B), B, look.grid, zstride, zbuf2[set], it == 0 || !pingpong_env, s);
            if (rcd) return rcd;
            look.zbuf = zbuf2[set];
            look.zclear = pingpong_env ? zbuf2[set ^ 1] : nullptr;
            depth = &look;
        }
        // time every `stride`-th launch only: an event pair costs a few microseconds of GPU timeline, which would
        // distort short kernels if it bracketed all of them
        const bool timed = tm && tm->used < tm->capacity && (it % tm->stride) == 0;
        if (timed) {
            hipError_t e = hipEventRecord(tm->start[tm->used], s);
            if (e != hipSuccess) return (int)e;
        }
        int rc;
        if (fused && it > 0) {
            // iteration `it` reads copy (it - 1) & 1 of state / pose records / partials and writes copy it & 1
            const int cin = (it - 1) & 1, cout = it & 1;
            PclFuseArgs f;
            f.partials_in = partials2[cin]; f.st_in = gd_poses(state, B, cin); f.recs_in = gd_recs(state, B, cin);
            f.st_out = gd_poses(state, B, cout); f.recs_out = gd_recs(state, B, cout);
            f.box = box; f.factor = hyper_host->factor; f.patience = (int)hyper_host->patience; f.mode = (int)hyper_host->mode;
            f.loss_out = loss_history ? loss_history + (int64_t)(it - 1) * B : nullptr;
            rc = pcl_launch_loss(cloud, n, pano, pano_format, H, W, f.recs_in, B, true, nullptr, partials2[cout], s, (flip_env ? (it & 1) : 0) | xcd_bit, &f,
                                 nullptr);
        } else {
            rc = pcl_launch_loss(cloud, n, pano, pano_format, H, W, gd_recs(state, B), B, true, nullptr, partials2[0], s, (flip_env ? (it & 1) : 0) | xcd_bit,
                                 nullptr, depth);
        }
        if (rc) return rc;
        if (timed) {
            // (only a completed start/stop pair counts as used: pcl_timer_read never sees a half-recorded slot)
            hipError_t e = hipEventRecord(tm->stop[tm->used], s);
            if (e != hipSuccess) return (int)e;
            tm->used++;
        }
        if (!fused) {
            epilogue(it, 0, partials2[0]);
            PCL_LAUNCH_CHECK();
        }
    }
    if (fused && num_iter > 0) {
        const int last = (num_iter - 1) & 1;                     // the copy the last launch wrote (copy 0 when it was the only one)
        epilogue(num_iter - 1, num_iter > 1 ? last : 0, partials2[num_iter > 1 ? last : 0]);
        PCL_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int pcl_gd_step_from_grads(void* state, int B, const float* loss, const float* grad, const float* box, const pcl_gd_hyper* hyper_host,
                                      float* scratch, void* stream)
{
    if (!state || !loss || !grad || !box || !hyper_host || !scratch || B <= 0) return PCL_EINVAL;
    if (hyper_host->mode != PCL_GD_SEQUENTIAL && hyper_host->mode != PCL_GD_BATCH) return PCL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, (size_t)B * PCL_NACC * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pcl_gd_forced_kernel, dim3(B), dim3(PCL_GD_THREADS), 0, s, scratch, gd_poses(state, B, 0), gd_recs(state, B, 0), box,
                       hyper_host->factor, (int)hyper_host->patience, (int)hyper_host->mode, loss, grad);
    PCL_LAUNCH_CHECK();
    return 0;
}

__global__ void pcl_gd_set_panos_kernel(PclPoseRec* recs, PclPoseRec* recs_shadow, const unsigned long long* __restrict__ panos, int B)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    unsigned long long p = panos ? panos[b] : 0ull;
    recs[b].pano_lo = recs_shadow[b].pano_lo = (uint32_t)(p & 0xffffffffull);
    recs[b].pano_hi = recs_shadow[b].pano_hi = (uint32_t)(p >> 32);
}

extern "C" int pcl_gd_set_panos(void* state, const uint64_t* panos, int B, void* stream)
{
    if (!state || B <= 0) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_gd_set_panos_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, gd_recs(state, B),
                       gd_recs(state, B, 1), (const unsigned long long*)panos, B);
    PCL_LAUNCH_CHECK();
    return 0;
}

// The panorama of every candidate from a SHORT host list: image i's address for candidates [i * per_image, (i + 1) * per_image).
// The addresses travel as kernel arguments — no device table, so no host-to-device copy in front of a refinement (a pageable
// H2D copy waits for everything the stream holds: it cost the shipped-shape pipeline its overlap of host and device work).
struct PclPanoList { unsigned long long p[PCL_GD_MAX_IMAGES]; };

__global__ void pcl_gd_set_pano_groups_kernel(PclPoseRec* recs, PclPoseRec* recs_shadow, PclPanoList list, int b0, int count, int per_image)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const unsigned long long p = list.p[j / per_image];
    const int b = b0 + j;
    recs[b].pano_lo = recs_shadow[b].pano_lo = (uint32_t)(p & 0xffffffffull);
    recs[b].pano_hi = recs_shadow[b].pano_hi = (uint32_t)(p >> 32);
}

extern "C" int pcl_gd_set_pano_groups(void* state, const uint64_t* panos_host, int nimages, int per_image, void* stream)
{
    if (!state || !panos_host || nimages <= 0 || per_image <= 0 || (int64_t)nimages * per_image > 0x7fffffff) return PCL_EINVAL;
    const int B = nimages * per_image;
    for (int i0 = 0; i0 < nimages; i0 += PCL_GD_MAX_IMAGES) {
        PclPanoList list;
        const int m = nimages - i0 < PCL_GD_MAX_IMAGES ? nimages - i0 : PCL_GD_MAX_IMAGES;
        for (int i = 0; i < PCL_GD_MAX_IMAGES; i++) list.p[i] = i < m ? (unsigned long long)panos_host[i0 + i] : 0ull;
        const int count = m * per_image;
        hipLaunchKernelGGL(pcl_gd_set_pano_groups_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, gd_recs(state, B),
                           gd_recs(state, B, 1), list, i0 * per_image, count, per_image);
        PCL_LAUNCH_CHECK();
    }
    return 0;
}

// The end of omniloc_batch (omniloc.py:271-277) for `nimages` images of `per_image` candidates each: the candidate whose LAST
// forward had the smallest loss (torch.argmin: the first of equal minima, and a NaN loss counts as the minimum), its post-step
// translation, R = RZ RY RX of its post-step angles, that loss and the angles: 16 floats per image.  Also hands the leaf
// parameters of all candidates back (the reference optimises views of the caller's tensors in place, omniloc.py:216-219).
// One wave per image: lane-strided scan of the candidates' last losses, a wave argmin with torch.argmin's rules (a NaN beats any
// number, among equals — or among NaNs — the smaller index wins), lane 0 writes the winner's 16 floats, all lanes the leaf rows.
__global__ void __launch_bounds__(PCL_WAVE) pcl_gd_winner_kernel(const PclGdPose* __restrict__ st, int nimages, int per_image, float* __restrict__ out,
                                                                 float* __restrict__ leaf_trans, float* __restrict__ leaf_rot)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    const PclGdPose* s = st + (int64_t)i * per_image;
    float best = 0.f;
    int k = 0x7fffffff;                                        // (no candidate yet)
    auto better = [](float la, int ia, float lb, int ib) {     // is (la, ia) ahead of (lb, ib)?
        if (ib == 0x7fffffff) return ia != 0x7fffffff;
        if (ia == 0x7fffffff) return false;
        const bool na = la != la, nb = lb != lb;
        if (na != nb) return na;
        if (na || la == lb) return ia < ib;
        return la < lb;
    };
    for (int b = lane; b < per_image; b += PCL_WAVE) {
        const float l = s[b].last_loss;
        if (better(l, b, best, k)) { best = l; k = b; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float lo = __shfl_xor(best, o, 64);
        const int ko = __shfl_xor(k, o, 64);
        if (better(lo, ko, best, k)) { best = lo; k = ko; }
    }
    if (lane == 0) {
        float* o = out + (int64_t)i * 16;
        float R[9];
        pcl_rot_from_ypr(s[k].fwd[3], s[k].fwd[4], s[k].fwd[5], R);
        for (int q = 0; q < 3; q++) { o[q] = s[k].fwd[q]; o[13 + q] = s[k].fwd[3 + q]; }
        for (int q = 0; q < 9; q++) o[3 + q] = R[q];
        o[12] = s[k].last_loss;
    }
    for (int b = lane; b < per_image; b += PCL_WAVE) {
        const int64_t row = ((int64_t)i * per_image + b) * 3;
        for (int q = 0; q < 3; q++) {
            if (leaf_trans) leaf_trans[row + q] = s[b].leaf[q];
            if (leaf_rot) leaf_rot[row + q] = s[b].leaf[3 + q];
        }
    }
}

extern "C" int pcl_gd_winner(const void* state, int nimages, int per_image, float* winners, float* leaf_trans, float* leaf_rot, void* stream)
{
    if (!state || !winners || nimages <= 0 || per_image <= 0) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_gd_winner_kernel, dim3(nimages), dim3(PCL_WAVE), 0, (hipStream_t)stream, (const PclGdPose*)state, nimages,
                       per_image, winners, leaf_trans, leaf_rot);
    PCL_LAUNCH_CHECK();
    return 0;
}

extern "C" int pcl_gd_result(const void* state, int B, float* result, void* stream)
{
    if (!state || !result || B <= 0) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_gd_result_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const PclGdPose*)state, B, result);
    PCL_LAUNCH_CHECK();
    return 0;
}
