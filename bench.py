#!/usr/bin/env python3
"""bench.py — candidate-poses/s of the PICCOLO pose refinement on MI355X (BASELINE.json metric).

One STEP = the complete gradient-descent refinement of ONE query panorama: B candidate starting poses, num_iter = 100
iterations (configs/stanford_parallel.ini shape: lr 0.1, patience 5, factor 0.8), each iteration one fused
projection + bilinear sampling + loss + gradient pass over the whole cloud and one on-device optimiser epilogue.
candidate-poses/s = B * images / wall time.  Inputs (packed cloud, packed panoramas, starting poses) are resident in
HBM before the timed region starts.  Independent query images are sharded round-robin over the ranks (weak scaling:
every rank refines `steps` images); the only collective is the final all_gather of the results (RCCL).

    python bench.py                                   # 1 GPU, cfg2 (1M points, 2048x1024, 32 candidates)
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from piccolo_amd import _lib, ops, synth  # noqa: E402

WORKLOADS = {
    # name: (N points, H, W, B candidates, batch_mode)
    "cfg1": (100_000, 256, 512, 1, False),
    "cfg2": (1_000_000, 1024, 2048, 32, True),
    "cfg3": (1_000_000, 1024, 2048, 256, True),
    "cfg4": (1_000_000, 1024, 2048, 32, True),      # 64 query images in all, sharded over the ranks (steps = 64 / world)
    "cfg5": (10_000_000, 2048, 4096, 32, True),
}
NUM_ITER, LR, PATIENCE, FACTOR, QUANTILE = 100, 0.1, 5, 0.8, 0.05
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_POINT_POSE = 24      # xyz + rgb fp32 read once per pose evaluation


def usable_cores():
    """Host cores this process may actually use: CPU affinity, capped by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_cfg1_end_to_end(cores):
    """BASELINE config 0 (the reference's own CPU-runnable case) on the host: 100k points, 512x256, one candidate,
    sequential GD for all 100 iterations with the oracle's restatement of omniloc; reports the pose error."""
    from oracle import gd as ogd
    from oracle import oracle as orc
    n, H, W = 100_000, 256, 512
    xyz, rgb = synth.box_room(n, seed=0)
    t_gt, ypr_gt = synth.gt_pose(0)
    img = orc.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W)).astype(np.float32) / 255.0
    tr, ro = synth.start_poses(t_gt, ypr_gt, 1, seed=0)

    class Cfg:
        lr, num_iter, patience, factor, out_of_room_quantile = LR, NUM_ITER, PATIENCE, FACTOR, QUANTILE

    t0 = time.perf_counter()
    res = ogd.omniloc(img, xyz, rgb, tr, ro, 0, Cfg(), loss_grad=ogd.make_loss_grad(xyz, rgb, img, nthreads=cores))
    dt = time.perf_counter() - t0
    t_err, r_err = synth.pose_errors(res[0], res[1], t_gt, synth.rot_from_ypr_np(ypr_gt))
    return {"t_err_m": t_err, "r_err_deg": r_err, "seconds": dt, "candidate_poses_per_s": 1.0 / dt}


def cpu_baseline(xyz, rgb, img, trans, rot, budget_s=12.0):
    """The oracle (C restatement of the reference's loss + gradient, OpenMP on all host cores) timed on a bounded
    sample of the same workload: as many pose evaluations as fit in ~budget_s, scaled to candidate-poses/s."""
    from oracle import oracle as orc
    orc.build()
    cores = usable_cores()
    orc.sampling_loss(xyz, rgb, img, trans[:1], rot[:1], dtype=np.float32, grad=True, nthreads=cores)   # warm-up
    t0 = time.perf_counter()
    orc.sampling_loss(xyz, rgb, img, trans[:4], rot[:4], dtype=np.float32, grad=True, nthreads=cores)   # sizes the sample
    t1 = (time.perf_counter() - t0) / 4
    n_pose = int(max(4, budget_s / max(t1, 1e-4)))
    reps = np.arange(n_pose) % len(trans)                   # cycle through the candidate poses to fill the budget
    t0 = time.perf_counter()
    orc.sampling_loss(xyz, rgb, img, trans[reps], rot[reps], dtype=np.float32, grad=True, nthreads=cores)
    dt = time.perf_counter() - t0
    pose_evals_per_s = n_pose / dt
    plumbing = cpu_cfg1_end_to_end(cores)
    return {"value": pose_evals_per_s / NUM_ITER, "unit": "candidate-poses/s", "cores": cores, "kind": "port",
            "cfg1_end_to_end": plumbing,
            "sample": "%d fused loss+gradient pose evaluations over the full %d-point cloud (%.1f s), fp32 oracle/pcl_oracle.c "
                      "with OpenMP; one candidate = %d evaluations" % (n_pose, len(xyz), dt, NUM_ITER),
            "pose_evals_per_s": pose_evals_per_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--images-per-launch", type=int, default=0,
                    help="query images whose candidates share one launch chain (0 = auto: 256 // B, at most --steps); "
                         "they share the cloud, each candidate samples its own image's panorama")
    ap.add_argument("--timer-stride", type=int, default=10, help="HIP-event pair around every n-th loss-kernel launch")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("PCL_BENCH_STREAMS", "1")),
                    help="independent query images refined concurrently on this many HIP streams per GPU (measured: no gain, "
                         "2545 vs 2536 candidate-poses/s at 1 vs 2 streams; kept as a knob)")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed GPU activity before the W warm-up steps: a fresh box needs ~0.2 s of load to leave its idle "
                         "clocks (first run after boot measured 3137 vs 3300 candidate-poses/s with --warmup 2)")
    ap.add_argument("--traffic-json", default=os.path.join(REPO, "profiles", "traffic.json"),
                    help="per-launch HBM bytes from the rocprofv3 PMC passes (written by profiles/collect.sh)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    # (PCL_DIST_BACKEND=gloo lets several ranks share one GPU: used by the test-suite to run the N > 1 code path end to
    #  end on a single-GPU box; the driver's multi-GPU runs use the default, RCCL, one GPU per rank)
    backend = os.environ.get("PCL_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or os.environ.get("PCL_BENCH_FORCE_DIST") == "1":   # (the env knob exercises the RCCL path at world size 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # RCCL
        else:
            dist.init_process_group(backend)
    _lib.load()

    N, H, W, B, batch_mode = WORKLOADS[args.workload]
    if args.workload == "cfg4":
        args.steps = max(1, 64 // world)
    K, Wm = args.steps, args.warmup
    n_img = K            # warm-up refinements re-run the timed images (their results are overwritten by the timed pass)

    # ---- untimed setup: synthetic room, one panorama per query image, everything packed and resident in HBM
    xyz, rgb = synth.box_room(N, seed=0)                      # the shared cloud, replicated on every rank
    X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
    cloud = ops.Cloud(X, C)
    box = ops.quantile_box(X, QUANTILE)
    panos, starts, gts = [], [], []
    for i in range(n_img):
        image_id = rank + i * world                           # round-robin sharding of the query images
        t_gt, ypr_gt = synth.gt_pose(image_id)
        cam = ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt))
        img = synth.quantise_like_image_file(ops.make_pano(cam, C, (H, W)))      # uint8-quantised like a decoded image file
        panos.append(ops.Pano(img))
        tr, ro = synth.start_poses(t_gt, ypr_gt, B, seed=image_id)
        starts.append((torch.from_numpy(tr).to(dev), torch.from_numpy(ro).to(dev)))
        gts.append((t_gt, ypr_gt))
        if i == 0 and rank == 0:
            img0_host, start0_host = img.cpu().numpy(), (tr, ro)
        del cam, img
    results = torch.zeros(n_img, 16, device=dev)
    # HIP-event pairs around every TIMER_STRIDE-th loss launch of the timed region (each pair costs a few us of GPU
    # timeline; bracketing all 100 launches of a refinement slows cfg 1 by 2x and cfg 2 by ~2 %)
    timer = ops.KernelTimer((NUM_ITER // max(1, args.timer_stride) + 1) * K, stride=args.timer_stride)
    # Independent images go to separate HIP streams: one image's optimiser epilogue, kernel boundaries and the tail of
    # its loss kernel overlap with the other image's loss kernel (each GD loop is a strict launch-after-launch chain).
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]

    # Images are refined in groups of `ipl`: the group's ipl * B candidates go through ONE chain of launches (shared
    # cloud, per-candidate panorama pointer).  More poses per launch share each cloud chunk in L2 and amortise the
    # per-block costs; the candidates stay independent (own Adam / scheduler state), so per-image results are the same.
    ipl = args.images_per_launch if args.images_per_launch > 0 else max(1, 256 // B)
    ipl = max(1, min(ipl, K))
    timed_groups = [list(range(s0, min(s0 + ipl, n_img))) for s0 in range(0, n_img, ipl)]
    # W untimed warm-up steps, as whole launch groups of the sizes the timed pass uses (at least W image refinements)
    warm_groups, done = [], 0
    while done < Wm:
        g = timed_groups[len(warm_groups) % len(timed_groups)]
        warm_groups.append(g)
        done += len(g)
    groups = warm_groups + timed_groups
    gd_by_size, prepared = {}, []
    for grp in groups:
        m = len(grp)
        if m not in gd_by_size:
            gd_by_size[m] = ops.GradientDescent(cloud, panos[0], starts[0][0].repeat(m, 1), starts[0][1].repeat(m, 1), box, lr=LR,
                                                patience=PATIENCE, factor=FACTOR, batch_mode=batch_mode)
        tr = torch.cat([starts[i][0] for i in grp]).contiguous()
        ro = torch.cat([starts[i][1] for i in grp]).contiguous()
        table = torch.tensor([panos[i].data.data_ptr() for i in grp for _ in range(B)], dtype=torch.int64, device=dev)
        prepared.append((grp, tr, ro, table))
    cols = torch.tensor([0, 1, 2, 3, 4, 5, 12], device=dev)

    def refine(gi, tm=None):
        grp, tr, ro, table = prepared[gi]
        gd = gd_by_size[len(grp)]
        st = streams[gi % len(streams)]
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            gd.reset(tr, ro)
            gd.set_pano_table(table)
            gd.run(NUM_ITER, timer=tm)
            res = gd.result().reshape(len(grp), B, -1)
            k = torch.argmin(res[:, :, 12], dim=1)             # per image: smallest loss of the last forward
            win = torch.gather(res, 1, k.reshape(-1, 1, 1).expand(-1, 1, res.shape[2]))[:, 0]
            results[grp[0]:grp[-1] + 1, :7] = win.index_select(1, cols)

    def join_streams():
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)

    def barrier():
        if dist is not None:
            dist.barrier()

    n_warm_groups = len(warm_groups)
    if args.prewarm_ms > 0:                                    # part of the untimed setup, not of the W warm-up steps
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            refine(0)
            torch.cuda.synchronize()
    for gi in range(n_warm_groups):
        refine(gi)
    join_streams()
    if dist is not None:                                       # first use of the collective sets up its connections: untimed
        src = results if backend == "nccl" else results.cpu()
        dist.all_gather_into_tensor(torch.empty(world * n_img, 16, device=src.device), src)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for gi in range(n_warm_groups, len(groups)):
        refine(gi, timer)
    join_streams()
    if dist is not None:                                       # the path's only collective: gather the results
        src = results if backend == "nccl" else results.cpu()
        gathered = torch.empty(world * n_img, 16, device=src.device)
        dist.all_gather_into_tensor(gathered, src)
    else:
        gathered = results
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kernel_ms, launches = timer.read()
    # accuracy of this rank's images (localize.py:239-247 formulas)
    errs = []
    res_host = results.cpu().numpy()
    for i in range(n_img):
        R = ops.rot_from_ypr(torch.from_numpy(res_host[i, 3:6]))[0].cpu().numpy()
        errs.append(synth.pose_errors(res_host[i, :3], R, gts[i][0], synth.rot_from_ypr_np(gts[i][1])))
    errs = np.array(errs)

    if rank == 0:
        value = B * K * world / elapsed
        per_launch_ms = kernel_ms / max(launches, 1)
        # a launch of group g evaluates len(g) images x B candidates; every run times the same number of launches
        timed_per_run = len(range(0, NUM_ITER, max(1, args.timer_stride)))
        total_bytes = sum(BYTES_PER_POINT_POSE * N * B * len(g) * timed_per_run for g in timed_groups)
        assert launches == timed_per_run * len(timed_groups), (launches, timed_per_run, len(timed_groups))
        alg_bytes = total_bytes / max(launches, 1)              # mean algorithmic bytes per timed launch
        achieved = total_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        if os.path.exists(args.traffic_json):
            try:
                traffic = json.load(open(args.traffic_json)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "candidate-poses/s", "value": value, "unit": "candidate-poses/s", "n_gpus": world, "steps": K,
            "warmup": Wm, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d-point cloud, %dx%d panorama, %d candidate poses x %d GD iterations per query image"
                                   % (args.workload, N, W, H, B, NUM_ITER),
                       "images_per_gpu": K, "sharding": "query images round-robin over ranks, RCCL all_gather of results",
                       "mode": "omniloc_batch" if batch_mode else "omniloc", "streams_per_gpu": len(streams),
                       "images_per_launch": ipl},
            "pose_evals_per_s": value * NUM_ITER,
            "median_t_err_m": float(np.median(errs[:, 0])), "median_r_err_deg": float(np.median(errs[:, 1])),
            "roofline": {"bound": "hbm", "kernel": "pcl_loss_kernel<G=%d, GRAD, %s>" % (2 if (B * ipl) % 2 == 0 else 1, {_lib.PANO_U8: "RGBA8", _lib.PANO_F16: "F16x4", _lib.PANO_F32: "F32x4"}[panos[0].fmt]), "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_rate_6290GBs": achieved / 6290.0,
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": per_launch_ms, "launches_timed": launches},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(xyz, rgb, img0_host, start0_host[0], start0_host[1])
    # RCCL prints its version banner through C stdio, which is flushed at exit — after Python's own output.  Every rank
    # flushes it now, before the last barrier, so that rank 0's JSON line is the last thing the job writes to stdout.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
