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

// Deterministic second-stage sum of the per-chunk partials of pose `b` over one WAVE's share of the chunks: thread `tid`
// of `nthreads` takes chunks tid, tid + nthreads, ... (fixed assignment, double), then the wave's lanes are summed.
// Two rows per trip so that a thread's loads are in flight together.
__device__ __forceinline__ void pcl_reduce_partials(const float* __restrict__ partials, int nchunks, int B, int b, int tid, int nthreads,
                                                    double out[PCL_NACC])
{
    double s[PCL_NACC];
#pragma unroll
    for (int k = 0; k < PCL_NACC; k++) s[k] = 0.0;
    for (int c = tid; c < nchunks; c += 2 * nthreads) {
        const int c2 = c + nthreads;
        const pcl_f4* p = reinterpret_cast<const pcl_f4*>(partials + ((int64_t)c * B + b) * PCL_NACC);
        const pcl_f4* q = reinterpret_cast<const pcl_f4*>(partials + ((int64_t)(c2 < nchunks ? c2 : c) * B + b) * PCL_NACC);
        pcl_f4 lo = p[0], hi = p[1], lo2 = q[0], hi2 = q[1];
        if (c2 >= nchunks) { lo2 = (pcl_f4){0.f, 0.f, 0.f, 0.f}; hi2 = lo2; }
        s[0] += lo.x; s[1] += lo.y; s[2] += lo.z; s[3] += lo.w;
        s[4] += hi.x; s[5] += hi.y; s[6] += hi.z; s[7] += hi.w;
        s[0] += lo2.x; s[1] += lo2.y; s[2] += lo2.z; s[3] += lo2.w;
        s[4] += hi2.x; s[5] += hi2.y; s[6] += hi2.z; s[7] += hi2.w;
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

// One 256-thread block finishes pose b: deterministic reduction, chain rule, Adam, plateau scheduler, clamp, next pose
// record.  An iteration of a small problem is two dependent launches of a few microseconds each, so this kernel is all
// latency (rocprofv3, round 2: 5.3 us of a 15 us iteration at the shipped 167k-point / 6-candidate shape):
//   - everything the update needs (optimiser state, pose record, clamp box) is requested BEFORE the partial sums, so
//     that one memory round trip covers both;
//   - the per-chunk partials are spread over four waves (one or two rows per thread: a single round of loads instead
//     of a loop of dependent ones per lane), wave sums by DPP, the four wave results through LDS;
//   - the per-parameter work then runs lane-parallel in wave 0 — lane k < 6 owns parameter k (t0, t1, t2, yaw, pitch,
//     roll): its gradient component, its Adam update, its clamp; lanes 3..5 take the sin/cos of the three angles at once.
// Every element goes through exactly the operations of the scalar form; the order of the fixed-order double sums is part
// of the build (same for eager launches, graph replay and the stateless pcl_finish_kernel).
#define PCL_GD_THREADS 256
__device__ __forceinline__ void pcl_gd_finish_pose(const float* __restrict__ partials, int nchunks, int B, int b, int tid, PclGdPose* st,
                                          PclPoseRec* recs, const float* __restrict__ box, double factor, int patience, int mode,
                                          float* loss_out)
{
    // torch evaluates the optimiser with separate, individually rounded tensor operations: no fused multiply-adds here
#pragma clang fp contract(off)
    PclGdPose* gp = st + b;
    PclPoseRec* rec = recs + b;
    const int lane = tid & 63, wave = tid >> 6;
    const bool owner = tid < 6;
    const int k = owner ? tid : 0;
    // ---- requests first (independent of the sums)
    const float sc0 = gp->sc[0], sc1 = gp->sc[1], sc2 = gp->sc[2], sc3 = gp->sc[3];
    const float Rk0 = rec->R[k < 3 ? k : 0], Rk1 = rec->R[3 + (k < 3 ? k : 0)], Rk2 = rec->R[6 + (k < 3 ? k : 0)];
    double lr = gp->lr, best = gp->best;
    int num_bad = gp->num_bad;
    const int step = gp->step + 1;
    const double beta1_pow_in = gp->beta1_pow, beta2_pow_in = gp->beta2_pow;
    float m = gp->m[k], v = gp->v[k], leaf = gp->leaf[k];
    const float box_lo = box[2 * (k < 3 ? k : 0)], box_hi = box[2 * (k < 3 ? k : 0) + 1];

    // ---- second-stage sums of the per-chunk partials (fixed thread -> chunk assignment, double)
    __shared__ double red[PCL_GD_THREADS / PCL_WAVE][PCL_NACC];
    double s[PCL_NACC];
    pcl_reduce_partials(partials, nchunks, B, b, tid, PCL_GD_THREADS, s);      // every lane holds its WAVE's eight sums
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < PCL_NACC; q++) red[wave][q] = s[q];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int q = 0; q < PCL_NACC; q++) s[q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);

    // loss and this lane's gradient component (pcl_chain_rule, one component per lane)
    const double M = s[1];
    const float loss = (float)s[0] / (float)M;
    const double inv = 1.0 / M;
    const double sy = sc0, cy = sc1, sp = sc2, cp = sc3;
    float gk;
    if (k < 3) gk = (float)(-((double)Rk0 * s[2] + (double)Rk1 * s[3] + (double)Rk2 * s[4]) * inv);
    else if (k == 3) gk = (float)(s[7] * inv);
    else if (k == 4) gk = (float)((-sy * s[5] + cy * s[6]) * inv);
    else gk = (float)((cy * cp * s[5] + sy * cp * s[6] - sp * s[7]) * inv);

    // torch.optim.Adam, single-tensor form (betas 0.9/0.999, eps 1e-8; call sites omniloc.py:33,235-236):
    // fp32 tensor math, python-double scalars
    const double beta1 = 0.9, beta2 = 0.999, eps = 1e-8;
    const double beta1_pow = beta1_pow_in * beta1;              // beta ** step as a running product (python: pow)
    const double beta2_pow = beta2_pow_in * beta2;
    const double bc1 = 1.0 - beta1_pow;
    const double bc2 = 1.0 - beta2_pow;
    const float step_size = (float)(-(lr / bc1));
    const float bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - beta1), b2 = (float)beta2, w2 = (float)(1.0 - beta2);
    m = m + w1 * (gk - m);                                      // exp_avg.lerp_(grad, 1 - beta1)
    v = v * b2 + w2 * gk * gk;                                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + (float)eps;       // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    leaf = leaf + step_size * m / denom;                        // param.addcdiv_(exp_avg, denom, value=-step_size)

    // ReduceLROnPlateau(mode='min', threshold=1e-4 rel, cooldown=0, min_lr=0, eps=1e-8).step(float(loss))
    // (omniloc.py:37,50 / :237,258) — wave-uniform
    const double cur = (double)loss;
    if (cur < best * (1.0 - 1e-4)) { best = cur; num_bad = 0; }
    else num_bad += 1;
    if (num_bad > patience) {
        double new_lr = lr * factor;
        if (new_lr < 0.0) new_lr = 0.0;
        if (lr - new_lr > 1e-8) lr = new_lr;
        num_bad = 0;
    }

    // clamp t to the quantile box; batch mode forwards the pre-clamp copy (omniloc.py:260-269), sequential mode
    // clamps the very tensor the next forward reads (omniloc.py:56-58)
    float fwd = leaf;
    if (k < 3) leaf = fminf(fmaxf(leaf, box_lo), box_hi);
    if (mode != PCL_GD_BATCH) fwd = leaf;

    // next pose record: lanes 3..5 hold yaw, pitch, roll (same fp32 sincosf + double products as pcl_write_pose_rec_fast)
    float sn, cs;
    sincosf(fwd, &sn, &cs);
    const double dsy = __shfl(sn, 3, 64), dcy = __shfl(cs, 3, 64), dsp = __shfl(sn, 4, 64), dcp = __shfl(cs, 4, 64);
    const double dsr = __shfl(sn, 5, 64), dcr = __shfl(cs, 5, 64);
    if (owner) {
        gp->leaf[k] = leaf; gp->fwd[k] = fwd; gp->m[k] = m; gp->v[k] = v;
        if (k < 3) rec->t[k] = fwd;                             // (pano_lo / pano_hi are left as they are)
    }
    if (lane == 0) {
        rec->R[0] = (float)(dcy * dcp); rec->R[1] = (float)(dcy * dsp * dsr - dsy * dcr); rec->R[2] = (float)(dcy * dsp * dcr + dsy * dsr);
        rec->R[3] = (float)(dsy * dcp); rec->R[4] = (float)(dsy * dsp * dsr + dcy * dcr); rec->R[5] = (float)(dsy * dsp * dcr - dcy * dsr);
        rec->R[6] = (float)(-dsp);      rec->R[7] = (float)(dcp * dsr);                   rec->R[8] = (float)(dcp * dcr);
        gp->sc[0] = (float)dsy; gp->sc[1] = (float)dcy; gp->sc[2] = (float)dsp; gp->sc[3] = (float)dcp;
        gp->lr = lr; gp->best = best; gp->num_bad = num_bad; gp->step = step;
        gp->beta1_pow = beta1_pow; gp->beta2_pow = beta2_pow; gp->last_loss = loss;
        if (loss_out) loss_out[b] = loss;
    }
}
