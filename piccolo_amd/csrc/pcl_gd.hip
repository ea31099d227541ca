// pcl_gd.hip — second-stage reduction, chain rule to (t, yaw, pitch, roll) and the on-device optimiser epilogue.
//
// Replaces, per GD iteration of the reference: the tail of autograd (omniloc.py:47,254), B x Adam.step and
// B x ReduceLROnPlateau.step(float(loss)) — each a host sync — (omniloc.py:49-50, :256-258), the re-cat of the
// parameters (omniloc.py:260-263) and the clamp to the quantile box (omniloc.py:52-58, :265-269).
// One 256-thread block per candidate pose (four waves gather the partial sums, wave 0 runs the optimiser update in the
// same precision mix as the reference: fp32 tensors, python-double scalars).
#include <stdlib.h>

#include "pcl_gd_device.h"

int pcl_launch_loss(const float* cloud, int64_t n, const void* pano, int pano_format, int H, int W, const PclPoseRec* poses,
                    int B, bool grad, const uint8_t* visible, float* partials, hipStream_t s, int flip);
size_t pcl_partials_bytes(int64_t n, int B);
int pcl_plan_nchunks(int64_t n, int B);
size_t pcl_depth_zbuf_bytes(int B, int H, int W);
int pcl_launch_depth_mask(const float* cloud, int64_t n, const PclPoseRec* poses, int B, int H, int W, float tau,
                          uint32_t* zbuf, uint8_t* visible, hipStream_t s);

// ---------------------------------------------------------------- stateless loss: pose setup + finish

__global__ void pcl_pose_setup_kernel(const float* __restrict__ trans, const float* __restrict__ rot, int B, PclPoseRec* recs)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float p[6] = {trans[3 * b], trans[3 * b + 1], trans[3 * b + 2], rot[3 * b], rot[3 * b + 1], rot[3 * b + 2]};
    pcl_write_pose_rec(&recs[b], p);
}

__global__ void __launch_bounds__(PCL_GD_THREADS) pcl_finish_kernel(const float* __restrict__ partials, int nchunks, int B,
                                                                    const PclPoseRec* __restrict__ recs,
                                                                    const float* __restrict__ rot, int with_grad,
                                                                    float* __restrict__ result)
{
    int b = blockIdx.x;
    __shared__ double red[PCL_GD_THREADS / PCL_WAVE][PCL_NACC];
    double s[PCL_NACC];
    pcl_reduce_partials(partials, nchunks, B, b, threadIdx.x, PCL_GD_THREADS, s);
    if ((threadIdx.x & 63) == 0)
        for (int q = 0; q < PCL_NACC; q++) red[threadIdx.x >> 6][q] = s[q];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 0; q < PCL_NACC; q++) s[q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
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
    int rc = pcl_launch_loss(cloud, n, pano, pano_format, H, W, recs, B, with_grad != 0, visible, partials, s, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(pcl_finish_kernel, dim3(B), dim3(PCL_GD_THREADS), 0, s, partials, pcl_plan_nchunks(n, B), B, recs, rot,
                       with_grad, result);
    PCL_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- GD: state init, epilogue, run, result

// state blob = PclGdPose[B] followed by PclPoseRec[B]
static inline PclGdPose* gd_poses(void* state) { return (PclGdPose*)state; }
static inline PclPoseRec* gd_recs(void* state, int B) { return (PclPoseRec*)((char*)state + (size_t)B * sizeof(PclGdPose)); }

__global__ void pcl_gd_init_kernel(PclGdPose* st, PclPoseRec* recs, const float* __restrict__ trans,
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
    st[b] = g;
}

__global__ void __launch_bounds__(PCL_GD_THREADS) pcl_gd_epilogue_kernel(const float* __restrict__ partials, int nchunks, int B,
                                                                         PclGdPose* st, PclPoseRec* recs,
                                                                         const float* __restrict__ box, double factor,
                                                                         int patience, int mode, float* loss_out)
{
    pcl_gd_finish_pose(partials, nchunks, B, blockIdx.x, threadIdx.x, st, recs, box, factor, patience, mode, loss_out);
}

__global__ void pcl_gd_result_kernel(const PclGdPose* __restrict__ st, int B, float* __restrict__ result)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float* r = result + (int64_t)b * PCL_GD_RESULT_STRIDE;
    for (int k = 0; k < 6; k++) { r[k] = st[b].fwd[k]; r[6 + k] = st[b].leaf[k]; }
    r[12] = st[b].last_loss;
    r[13] = (float)st[b].lr;
}

extern "C" size_t pcl_gd_state_bytes(int B) { return B > 0 ? (size_t)B * (sizeof(PclGdPose) + sizeof(PclPoseRec)) : 0; }

static size_t gd_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t pcl_gd_workspace_bytes(int64_t n, int B, int H, int W, const pcl_gd_hyper* hyper_host)
{
    if (n <= 0 || B <= 0 || !hyper_host) return 0;
    size_t bytes = gd_align(pcl_partials_bytes(n, B));
    if (hyper_host->depth_mask) bytes += gd_align(pcl_depth_zbuf_bytes(B, H, W)) + gd_align((size_t)B * (size_t)n);
    return bytes;
}

extern "C" int pcl_gd_init(void* state, const float* trans, const float* rot, int B, const pcl_gd_hyper* hyper_host, void* stream)
{
    if (!state || !trans || !rot || !hyper_host || B <= 0) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_gd_init_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, gd_poses(state),
                       gd_recs(state, B), trans, rot, B, hyper_host->lr);
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
    if (workspace_bytes < pcl_gd_workspace_bytes(n, B, H, W, hyper_host)) return PCL_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* partials = (float*)workspace;
    uint32_t* zbuf = nullptr;
    uint8_t* visible = nullptr;
    if (hyper_host->depth_mask) {
        if (!(hyper_host->depth_tau >= 0.f) || B > 65535) return PCL_EINVAL;
        zbuf = (uint32_t*)((char*)workspace + gd_align(pcl_partials_bytes(n, B)));
        visible = (uint8_t*)zbuf + gd_align(pcl_depth_zbuf_bytes(B, H, W));
    }
    const int nchunks = pcl_plan_nchunks(n, B);
    // odd iterations walk every XCD's chunks backwards: the first blocks of a launch then read the chunks the previous launch
    // finished with, still in that XCD's L2 (PCL_FLIP=0 turns it off; +0.3 % at cfg 2 in two A/B alternations on one box —
    // the first round of a launch stays 6 us slower than the later ones, so cold L2 is not what makes it slow)
    static const int flip_env = getenv("PCL_FLIP") ? atoi(getenv("PCL_FLIP")) : 1;
    for (int it = 0; it < num_iter; it++) {
        if (visible) {
            int rcd = pcl_launch_depth_mask(cloud, n, gd_recs(state, B), B, H, W, hyper_host->depth_tau, zbuf, visible, s);
            if (rcd) return rcd;
        }
        // time every `stride`-th launch only: an event pair costs a few microseconds of GPU timeline, which would
        // distort short kernels if it bracketed all of them
        const bool timed = tm && tm->used < tm->capacity && (it % tm->stride) == 0;
        if (timed) {
            hipError_t e = hipEventRecord(tm->start[tm->used], s);
            if (e != hipSuccess) return (int)e;
        }
        int rc = pcl_launch_loss(cloud, n, pano, pano_format, H, W, gd_recs(state, B), B, true, visible, partials, s, flip_env ? (it & 1) : 0);
        if (rc) return rc;
        if (timed) {
            // (only a completed start/stop pair counts as used: pcl_timer_read never sees a half-recorded slot)
            hipError_t e = hipEventRecord(tm->stop[tm->used], s);
            if (e != hipSuccess) return (int)e;
            tm->used++;
        }
        hipLaunchKernelGGL(pcl_gd_epilogue_kernel, dim3(B), dim3(PCL_GD_THREADS), 0, s, partials, nchunks, B, gd_poses(state),
                           gd_recs(state, B), box, hyper_host->factor, (int)hyper_host->patience, (int)hyper_host->mode,
                           loss_history ? loss_history + (int64_t)it * B : nullptr);
        PCL_LAUNCH_CHECK();
    }
    return 0;
}

__global__ void pcl_gd_set_panos_kernel(PclPoseRec* recs, const unsigned long long* __restrict__ panos, int B)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    unsigned long long p = panos ? panos[b] : 0ull;
    recs[b].pano_lo = (uint32_t)(p & 0xffffffffull);
    recs[b].pano_hi = (uint32_t)(p >> 32);
}

extern "C" int pcl_gd_set_panos(void* state, const uint64_t* panos, int B, void* stream)
{
    if (!state || B <= 0) return PCL_EINVAL;
    hipLaunchKernelGGL(pcl_gd_set_panos_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, gd_recs(state, B),
                       (const unsigned long long*)panos, B);
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
