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

// Partial sums in HBM: partials[group][chunk][g][8] floats — the rows of one pose GROUP (the G poses a loss block evaluates
// together) are one contiguous run of nchunks x G x 32 bytes.  Whoever finishes a group (the stand-alone epilogue kernel, the
// fused prologue of the next iteration's loss blocks, the stateless finish kernel) reads that run with fully coalesced 16-byte
// loads: thread t takes float4 number t, t + 256, ... of the run, which is always the same QUARTER-ROW of the same pose
// (256 is a multiple of the 2 G float4s per chunk).  (Round 3, first fused version: partials[chunk][B][8] read one row per
// thread — every wave load touched 64 different cache lines, and with all 984 blocks of the shipped shape reading their
// 328 rows at once the prologue cost 5 us per iteration, more than the launch it replaced.)
#define PCL_GD_THREADS 256
__device__ __forceinline__ int64_t pcl_partials_row(int nchunks, int G, int group, int chunk, int g)
{
    return (((int64_t)group * nchunks + chunk) * G + g) * PCL_NACC;
}

template <int CTRL>
__device__ __forceinline__ double pcl_dpp_sum_d(double v)
{
    return v + pcl_dpp_d<CTRL>(v);
}

// Deterministic second-stage sums of pose group `grp`: sums[g][0..7] (LDS, double) for its G poses.  Fixed order: per thread
// over its float4s, then the lanes of a 16-lane row that hold the same quarter-row (DPP row rotations), then the 16 (wave, row)
// partial sums one after the other.  Needs all PCL_GD_THREADS threads; two barriers.
// rows_sh: 16 * 2 G entries of 4 doubles.
template <int G>
__device__ __forceinline__ void pcl_reduce_group(const float* __restrict__ partials, int nchunks, int grp, int tid, double (*rows_sh)[4],
                                                 double (*sums_sh)[PCL_NACC])
{
    constexpr int S = 2 * G;                                  // float4s per chunk
    static_assert(S == 2 || S == 4 || S == 8, "1, 2 or 4 poses per group");
    const pcl_f4* __restrict__ base = reinterpret_cast<const pcl_f4*>(partials + pcl_partials_row(nchunks, G, grp, 0, 0));
    const int total = nchunks * S;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    // eight loads in flight per trip (one trip for every grid that takes the fused path: <= 1024 chunk x group blocks); they are
    // added in the order of their addresses whatever the trip count
    for (int q = tid; q < total; q += 8 * PCL_GD_THREADS) {
        pcl_f4 u[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int qj = q + j * PCL_GD_THREADS;
            u[j] = base[qj < total ? qj : q];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (q + j * PCL_GD_THREADS < total) { a0 += u[j].x; a1 += u[j].y; a2 += u[j].z; a3 += u[j].w; }
        }
    }
    // lanes l, l + S, l + 2 S, ... of a row hold the same quarter-row: row_ror by S, 2 S, ... (0x120 + n)
    if (S <= 2) { a0 = pcl_dpp_sum_d<0x122>(a0); a1 = pcl_dpp_sum_d<0x122>(a1); a2 = pcl_dpp_sum_d<0x122>(a2); a3 = pcl_dpp_sum_d<0x122>(a3); }
    if (S <= 4) { a0 = pcl_dpp_sum_d<0x124>(a0); a1 = pcl_dpp_sum_d<0x124>(a1); a2 = pcl_dpp_sum_d<0x124>(a2); a3 = pcl_dpp_sum_d<0x124>(a3); }
    a0 = pcl_dpp_sum_d<0x128>(a0); a1 = pcl_dpp_sum_d<0x128>(a1); a2 = pcl_dpp_sum_d<0x128>(a2); a3 = pcl_dpp_sum_d<0x128>(a3);
    const int lane = tid & 63;
    if ((lane & 15) < S) {
        double* d = rows_sh[(tid >> 4) * S + (lane & 15)];     // (tid >> 4) = wave * 4 + row: 16 of them
        d[0] = a0; d[1] = a1; d[2] = a2; d[3] = a3;
    }
    __syncthreads();
    if (tid < 4 * S) {
        const int slot = tid >> 2, comp = tid & 3;
        double t = 0.0;
#pragma unroll
        for (int e = 0; e < PCL_GD_THREADS / 16; e++) t += rows_sh[e * S + slot][comp];
        sums_sh[slot >> 1][(slot & 1) * 4 + comp] = t;
    }
    __syncthreads();
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

// One 256-thread block finishes a pose GROUP: deterministic reduction, chain rule, Adam, plateau scheduler, clamp, next pose
// records.  An iteration of a small problem is a few microseconds, so this is all latency:
//   - everything the update needs (optimiser state, pose record, clamp box) is requested BEFORE the partial sums, so
//     that one memory round trip covers both;
//   - the group's partial rows are one contiguous run read with coalesced 16-byte loads by all four waves (pcl_reduce_group);
//   - the per-parameter work then runs lane-parallel in wave 0 — lane 8 g + k owns parameter k (t0, t1, t2, yaw, pitch, roll)
//     of pose g: its gradient component, its Adam update, its clamp; lanes 8 g + 3..5 take the sin/cos of the three angles.
// Every element goes through exactly the operations of the scalar form; the order of the fixed-order double sums is part
// of the build (same for the stand-alone epilogue, the fused prologue, graph replay and the stateless pcl_finish_kernel).

// What a FUSED iteration needs besides the loss pass (pcl_loss.hip, `FUSED` variant): every block of iteration k + 1 first
// finishes iteration k for its own poses — same reduction, chain rule, Adam, scheduler and clamp as the stand-alone epilogue,
// identical arithmetic in every block, so the blocks agree bit for bit without talking to each other — and only the block of
// chunk 0 stores the state.  Everything a block reads here is from iteration k's buffers, everything it writes goes to
// iteration k + 1's: the kernel boundary stays the only synchronisation.
struct PclFuseArgs {
    const float* partials_in;     // [nchunks][B][8] of the previous iteration
    const PclGdPose* st_in;       // optimiser state before the update
    const PclPoseRec* recs_in;    // pose the previous forward used (chain rule) + the panorama addresses
    PclGdPose* st_out;            // != st_in
    PclPoseRec* recs_out;         // != recs_in
    const float* box;
    double factor;
    int patience, mode;
    float* loss_out;              // nullable: loss history row of the previous iteration
};

// Finish pose GROUP `grp` (its G poses side by side in wave 0: lane 8 g + k owns parameter k of pose g).
// ALL: every wave stays to the end (fused prologue: the block goes on to the loss pass), else waves 1..3 leave after the
// reduction (stand-alone epilogue kernel).  `store`: write state, pose records and losses (the stand-alone kernel: always;
// fused: the block of chunk 0).  `pose_sh` (nullable): LDS [G][12], receives R[9], t[3] of the poses the next forward uses.
// in == out is allowed when a single block handles the group (stand-alone kernel).
// FORCED (pcl_gd_step_from_grads, the teacher-forcing hook of the parity tests): loss and gradient of every pose are READ from
// forced_loss [B] / forced_grad [B][6] (t0, t1, t2, yaw, pitch, roll) instead of coming out of the sums; everything behind them —
// Adam, scheduler, clamp, next pose record — is this very code.
template <int G, bool ALL, bool FORCED = false>
__device__ __forceinline__ void pcl_gd_finish_group(const float* __restrict__ partials, int nchunks, int grp, int tid,
                                                    const PclGdPose* st_in, const PclPoseRec* recs_in, PclGdPose* st_out, PclPoseRec* recs_out,
                                                    bool store, const float* __restrict__ box, double factor, int patience, int mode,
                                                    float* loss_out, double (*rows_sh)[4], double (*sums_sh)[PCL_NACC], float (*pose_sh)[12],
                                                    const float* __restrict__ forced_loss = nullptr, const float* __restrict__ forced_grad = nullptr)
{
    // torch evaluates the optimiser with separate, individually rounded tensor operations: no fused multiply-adds here
#pragma clang fp contract(off)
    const int g = (tid >> 3) < G ? (tid >> 3) : G - 1, k6 = tid & 7;
    const bool owner = tid < 8 * G && k6 < 6;
    const int k = owner ? k6 : 0;
    const int b = grp * G + g;
    const PclGdPose* gp = st_in + b;
    const PclPoseRec* rec = recs_in + b;
    PclGdPose* gq = st_out + b;
    PclPoseRec* rq = recs_out + b;
    // ---- requests first (independent of the sums; only wave 0 uses them)
    float sc0 = 0.f, sc1 = 0.f, sc2 = 0.f, sc3 = 0.f, Rk0 = 0.f, Rk1 = 0.f, Rk2 = 0.f, m = 0.f, v = 0.f, leaf = 0.f, box_lo = 0.f, box_hi = 0.f;
    double lr = 0.0, best = 0.0, beta1_pow_in = 1.0, beta2_pow_in = 1.0;
    int num_bad = 0, step = 0;
    if (tid < 64) {
        sc0 = gp->sc[0]; sc1 = gp->sc[1]; sc2 = gp->sc[2]; sc3 = gp->sc[3];
        Rk0 = rec->R[k < 3 ? k : 0]; Rk1 = rec->R[3 + (k < 3 ? k : 0)]; Rk2 = rec->R[6 + (k < 3 ? k : 0)];
        lr = gp->lr; best = gp->best;
        num_bad = gp->num_bad;
        step = gp->step + 1;
        beta1_pow_in = gp->beta1_pow; beta2_pow_in = gp->beta2_pow;
        m = gp->m[k]; v = gp->v[k]; leaf = gp->leaf[k];
        box_lo = box[2 * (k < 3 ? k : 0)]; box_hi = box[2 * (k < 3 ? k : 0) + 1];
    }

    // ---- second-stage sums of the per-chunk partials (fixed order, double)
    pcl_reduce_group<G>(partials, nchunks, grp, tid, rows_sh, sums_sh);
    if (!ALL && tid >= 64) return;
    if (tid < 64) {
        double s[PCL_NACC];
#pragma unroll
        for (int q = 0; q < PCL_NACC; q++) s[q] = sums_sh[g][q];

        // loss and this lane's gradient component (pcl_chain_rule, one component per lane)
        const double M = s[1];
        float loss = (float)s[0] / (float)M;
        const double inv = 1.0 / M;
        const double sy = sc0, cy = sc1, sp = sc2, cp = sc3;
        float gk;
        if (k < 3) gk = (float)(-((double)Rk0 * s[2] + (double)Rk1 * s[3] + (double)Rk2 * s[4]) * inv);
        else if (k == 3) gk = (float)(s[7] * inv);
        else if (k == 4) gk = (float)((-sy * s[5] + cy * s[6]) * inv);
        else gk = (float)((cy * cp * s[5] + sy * cp * s[6] - sp * s[7]) * inv);
        if constexpr (FORCED) { loss = forced_loss[b]; gk = forced_grad[6 * b + k]; }

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
        // (omniloc.py:37,50 / :237,258) — the same for all lanes of a pose
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

        // next pose record: lanes 8 g + 3..5 hold yaw, pitch, roll (same fp32 sincosf + double products as pcl_write_pose_rec_fast)
        float sn, cs;
        sincosf(fwd, &sn, &cs);
        const int l0 = tid & ~7;
        const double dsy = __shfl(sn, l0 + 3, 64), dcy = __shfl(cs, l0 + 3, 64), dsp = __shfl(sn, l0 + 4, 64), dcp = __shfl(cs, l0 + 4, 64);
        const double dsr = __shfl(sn, l0 + 5, 64), dcr = __shfl(cs, l0 + 5, 64);
        if (owner && pose_sh && k < 3) pose_sh[g][9 + k] = fwd;
        if (owner && store) {
            gq->leaf[k] = leaf; gq->fwd[k] = fwd; gq->m[k] = m; gq->v[k] = v;
            if (k < 3) rq->t[k] = fwd;                             // (pano_lo / pano_hi are left as they are)
        }
        if (owner && k6 == 0) {
            float Rn[9];
            Rn[0] = (float)(dcy * dcp); Rn[1] = (float)(dcy * dsp * dsr - dsy * dcr); Rn[2] = (float)(dcy * dsp * dcr + dsy * dsr);
            Rn[3] = (float)(dsy * dcp); Rn[4] = (float)(dsy * dsp * dsr + dcy * dcr); Rn[5] = (float)(dsy * dsp * dcr - dcy * dsr);
            Rn[6] = (float)(-dsp);      Rn[7] = (float)(dcp * dsr);                   Rn[8] = (float)(dcp * dcr);
            if (pose_sh) {
#pragma unroll
                for (int q = 0; q < 9; q++) pose_sh[g][q] = Rn[q];
            }
            if (store) {
#pragma unroll
                for (int q = 0; q < 9; q++) rq->R[q] = Rn[q];
                gq->sc[0] = (float)dsy; gq->sc[1] = (float)dcy; gq->sc[2] = (float)dsp; gq->sc[3] = (float)dcp;
                gq->lr = lr; gq->best = best; gq->num_bad = num_bad; gq->step = step;
                gq->beta1_pow = beta1_pow; gq->beta2_pow = beta2_pow; gq->last_loss = loss;
                if (loss_out) loss_out[b] = loss;
            }
        }
    }
}
