// pcl_gd_device.h — device-side pieces of the GD epilogue (second-stage reduction, chain rule, optimiser update).
//
// Measured and rejected (round 1): running pcl_gd_finish_pose inside pcl_loss_kernel, in the block that draws the last
// ticket of its pose group (write-through partial stores -> drain -> relaxed agent-scope ticket; last arriver: agent
// acquire -> reduce -> update), so that an iteration is ONE launch.  Bit-identical to the two-kernel path, but slower:
// cfg2 (B = 32) 119.2 vs 116.5 us per iteration, cfg1 10.9 vs 9.4 us; with a release fence per block in place of the
// sc1 stores 154.9 us (4096 L2 write-backs per launch).  The ticket round trip + acquire on the critical path of the
// last group cost more than the launch boundary they replace.
#pragma once
#include "pcl_device.h"

// Deterministic second-stage sum of the per-chunk partials of pose `b` (fixed lane->chunk assignment, double).
__device__ __forceinline__ void pcl_reduce_partials(const float* __restrict__ partials, int nchunks, int B, int b, int lane, double out[PCL_NACC])
{
    double s[PCL_NACC];
#pragma unroll
    for (int k = 0; k < PCL_NACC; k++) s[k] = 0.0;
    for (int c = lane; c < nchunks; c += PCL_WAVE) {
        const pcl_f4* p = reinterpret_cast<const pcl_f4*>(partials + ((int64_t)c * B + b) * PCL_NACC);
        pcl_f4 lo = p[0], hi = p[1];
        s[0] += lo.x; s[1] += lo.y; s[2] += lo.z; s[3] += lo.w;
        s[4] += hi.x; s[5] += hi.y; s[6] += hi.z; s[7] += hi.w;
    }
#pragma unroll
    for (int k = 0; k < PCL_NACC; k++) out[k] = pcl_wave_sum_d(s[k]);
}

// loss and gradient w.r.t. (t, yaw, pitch, roll) from the 8 sums, at pose p = (t, yaw, pitch, roll).
//   dL/dt = -R^T sum g / M ;  dL/dyaw = e_z . T/M ; dL/dpitch = (RZ e_y) . T/M ; dL/droll = (RZ RY e_x) . T/M
// with T = sum p x g (see pcl_loss.hip).  M = 0 gives NaN like the reference's 0/0.
__device__ __forceinline__ void pcl_chain_rule(const double s[PCL_NACC], const float R[9], double sy, double cy, double sp, double cp,
                                      float& loss, float grad[6])
{
    double M = s[1];
    loss = (float)s[0] / (float)M;
    double inv = 1.0 / M;
#pragma unroll
    for (int k = 0; k < 3; k++) grad[k] = (float)(-((double)R[k] * s[2] + (double)R[3 + k] * s[3] + (double)R[6 + k] * s[4]) * inv);
    grad[3] = (float)(s[7] * inv);
    grad[4] = (float)((-sy * s[5] + cy * s[6]) * inv);
    grad[5] = (float)((cy * cp * s[5] + sy * cp * s[6] - sp * s[7]) * inv);
}

// One wave (`lane` = 0..63) finishes pose b: deterministic reduction, chain rule, Adam, plateau scheduler, clamp, next
// pose record.
__device__ __forceinline__ void pcl_gd_finish_pose(const float* __restrict__ partials, int nchunks, int B, int b, int lane, PclGdPose* st,
                                          PclPoseRec* recs, const float* __restrict__ box, double factor, int patience, int mode,
                                          float* loss_out)
{
    double s[PCL_NACC];
    pcl_reduce_partials(partials, nchunks, B, b, lane, s);
    if (lane != 0) return;
    PclGdPose g = st[b];
    float loss, grad[6];
    pcl_chain_rule(s, recs[b].R, g.sc[0], g.sc[1], g.sc[2], g.sc[3], loss, grad);
    g.last_loss = loss;
    if (loss_out) loss_out[b] = loss;

    // torch.optim.Adam, single-tensor form (betas 0.9/0.999, eps 1e-8; call sites omniloc.py:33,235-236):
    // fp32 tensor math, python-double scalars
    const double beta1 = 0.9, beta2 = 0.999, eps = 1e-8;
    g.step += 1;
    g.beta1_pow *= beta1;                                     // beta ** step as a running product (python: pow)
    g.beta2_pow *= beta2;
    double bc1 = 1.0 - g.beta1_pow;
    double bc2 = 1.0 - g.beta2_pow;
    float step_size = (float)(-(g.lr / bc1));
    float bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2);
#pragma unroll
    for (int k = 0; k < 6; k++) {
        float gk = grad[k];
        g.m[k] = g.m[k] + w1 * (gk - g.m[k]);                 // exp_avg.lerp_(grad, 1 - beta1)
        g.v[k] = g.v[k] * b2 + w2 * gk * gk;                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        float denom = sqrtf(g.v[k]) / bc2_sqrt + (float)eps;  // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        g.leaf[k] = g.leaf[k] + step_size * g.m[k] / denom;   // param.addcdiv_(exp_avg, denom, value=-step_size)
    }

    // ReduceLROnPlateau(mode='min', threshold=1e-4 rel, cooldown=0, min_lr=0, eps=1e-8).step(float(loss))
    // (omniloc.py:37,50 / :237,258)
    double cur = (double)loss;
    if (cur < g.best * (1.0 - 1e-4)) { g.best = cur; g.num_bad = 0; }
    else g.num_bad += 1;
    if (g.num_bad > patience) {
        double new_lr = g.lr * factor;
        if (new_lr < 0.0) new_lr = 0.0;
        if (g.lr - new_lr > 1e-8) g.lr = new_lr;
        g.num_bad = 0;
    }

    // clamp t to the quantile box; batch mode forwards the pre-clamp copy (omniloc.py:260-269), sequential mode
    // clamps the very tensor the next forward reads (omniloc.py:56-58)
    if (mode == PCL_GD_BATCH) {
#pragma unroll
        for (int k = 0; k < 6; k++) g.fwd[k] = g.leaf[k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) g.leaf[k] = fminf(fmaxf(g.leaf[k], box[2 * k]), box[2 * k + 1]);
    if (mode != PCL_GD_BATCH) {
#pragma unroll
        for (int k = 0; k < 6; k++) g.fwd[k] = g.leaf[k];
    }
    pcl_write_pose_rec_fast(&recs[b], g.fwd, g.sc);
    st[b] = g;
}
