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

The JSON line carries both launch modes and both roofs:
  value / ms_per_step         default mode: 256 // B query images share one launch chain (cfg 4's shape on one GPU)
  single_image {...}          the literal cfg-2 mode: ONE query image per launch chain (B poses per launch)
  roofline {...}              SURVEY.md 8(d): ALGORITHMIC bytes (24 B x points x poses) / kernel time vs the 8 TB/s HBM peak,
                              kernel time measured live with HIP events; `traffic` = memory-side bytes per launch from the
                              rocprofv3 PMC passes of THIS launch shape (profiles/roofs.json), null for any other shape;
    roofline.hbm_measured     those measured bytes over the measured kernel time: the bandwidth the kernel really draws
    roofline.valu             the roof that actually binds (the cloud and the panorama are cache resident): VALU instructions
                              per point-pose and the VALU-busy fraction, from the same PMC passes; issue_frac_live = those
                              instructions over this run's launch time against 1 instruction / SIMD / 4 cycles at 2400 MHz
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


def lookup_roofs(path, workload, poses_per_launch, fmt_name):
    """Counter-derived figures for exactly this launch shape (profiles/roofs.json, written by profiles/summarize.py from
    the rocprofv3 --pmc passes); None when the run's shape was never profiled."""
    workload = {"cfg4": "cfg2"}.get(workload, workload)      # cfg 4 = cfg 2's cloud, panorama size and candidates per image
    key = "%s/poses%d/%s" % (workload, poses_per_launch, fmt_name)
    try:
        return key, json.load(open(path)).get(key)
    except Exception:
        return key, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-image", action="store_true", help="skip the one-image-per-launch-chain pass")
    ap.add_argument("--images-per-launch", type=int, default=0,
                    help="query images whose candidates share one launch chain (0 = auto: 256 // B, at most --steps); "
                         "they share the cloud, each candidate samples its own image's panorama")
    ap.add_argument("--timer-stride", type=int, default=10, help="HIP-event pair around every n-th loss-kernel launch")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="repeat the whole timed pass (exactly --steps steps, barrier + synchronize on both sides) until this "
                         "much time has been measured and report the MEDIAN pass: the timed region of one cfg-2 pass is "
                         "0.08 s, too short for an outside observer (GPU-busy sampling) to corroborate; 0 = one pass")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed GPU activity before the W warm-up steps: a fresh box needs ~0.2 s of load to leave its idle "
                         "clocks (first run after boot measured 3137 vs 3300 candidate-poses/s with --warmup 2)")
    ap.add_argument("--roofs-json", default=os.path.join(REPO, "profiles", "roofs.json"),
                    help="per-launch-shape counter figures from the rocprofv3 PMC passes (profiles/collect.sh)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    # (PCL_DIST_BACKEND=gloo lets several ranks share one GPU: used by the test-suite to run the N > 1 code path end to
    #  end on a single-GPU box; the driver's multi-GPU runs use the default, RCCL, one GPU per rank)
    backend = os.environ.get("PCL_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs an MI355X: torch.cuda.device_count() == 0")
    if backend == "nccl" and n_dev < args.gpus:
        raise SystemExit("--gpus %d but only %d GPU(s) are visible to this process (torch.cuda.device_count()): one rank per "
                         "GPU is required for the RCCL run" % (args.gpus, n_dev))
    dev_index = local_rank % n_dev if backend != "nccl" else local_rank
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

    # Images are refined in groups of `ipl`: the group's ipl * B candidates go through ONE chain of launches (shared
    # cloud, per-candidate panorama pointer).  More poses per launch share each cloud chunk in L2 and amortise the
    # per-block costs; the candidates stay independent (own Adam / scheduler state), so per-image results are the same.
    if args.images_per_launch > 0:
        ipl = max(1, min(args.images_per_launch, K))
    else:
        # auto: about 256 poses per launch, in EQUAL groups when the step count allows it (every timed launch then has
        # one shape, the one the counter passes under profiles/ were taken for): the largest divisor of K that is at most
        # 256 // B and at least half of it; otherwise groups of 256 // B with a shorter last one
        # ... and no more images than keep their packed panoramas (8 B per texel) within ~half of the 256 MiB Infinity
        # Cache: cfg 5's 4096x2048 panoramas are 67 MB each (measured 330 / 328 / 312 candidate-poses/s at 1 / 2 / 4 images
        # per launch), cfg 2's 16.8 MB (3260 / 3314 / 3295 at 4 / 8 / 16)
        cache_cap = max(1, int(140e6 // ((H + 2) * (W + 2) * 8)))
        target = max(1, min(256 // B, K, cache_cap))
        divs = [d for d in range(target, 0, -1) if K % d == 0 and 2 * d >= target]
        ipl = divs[0] if divs else target
    # warm-up steps refine their OWN images (ids beyond every rank's timed ones), in whole launch groups of the timed size
    n_warm = ((Wm + ipl - 1) // ipl) * ipl if Wm > 0 else 0
    # (the untimed clock pre-warm needs a launch group of its own images even with --warmup 0)
    n_img = K + max(n_warm, ipl if args.prewarm_ms > 0 else 0)

    # ---- untimed setup: synthetic room, one panorama per query image, everything packed and resident in HBM
    xyz, rgb = synth.box_room(N, seed=0)                      # the shared cloud, replicated on every rank
    X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
    cloud = ops.Cloud(X, C)
    box = ops.quantile_box(X, QUANTILE)
    panos, starts, gts = [], [], []
    for i in range(n_img):
        # timed image i of this rank is query image rank + i * world (round-robin sharding); warm-up images follow
        image_id = rank + i * world if i < K else 1_000_000 + rank + (i - K) * world
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
    fmt_name = {_lib.PANO_U8: "u8", _lib.PANO_F16: "f16", _lib.PANO_F32: "f32"}[panos[0].fmt]
    results = torch.zeros(n_img, 16, device=dev)
    # HIP-event pairs around every TIMER_STRIDE-th loss launch of the timed region (each pair costs a few us of GPU
    # timeline; bracketing all 100 launches of a refinement slows cfg 1 by 2x and cfg 2 by ~2 %)
    timed_per_run = len(range(0, NUM_ITER, max(1, args.timer_stride)))
    timer = ops.KernelTimer(timed_per_run * (K + 1), stride=args.timer_stride)
    cols = torch.tensor([0, 1, 2, 3, 4, 5, 12], device=dev)

    def make_groups(images, per):
        return [images[s0:s0 + per] for s0 in range(0, len(images), per)]

    gd_by_size = {}

    def prepare(groups):
        out = []
        for grp in groups:
            m = len(grp)
            if m not in gd_by_size:
                gd_by_size[m] = ops.GradientDescent(cloud, panos[0], starts[0][0].repeat(m, 1), starts[0][1].repeat(m, 1), box, lr=LR,
                                                    patience=PATIENCE, factor=FACTOR, batch_mode=batch_mode)
            tr = torch.cat([starts[i][0] for i in grp]).contiguous()
            ro = torch.cat([starts[i][1] for i in grp]).contiguous()
            table = torch.tensor([panos[i].data.data_ptr() for i in grp for _ in range(B)], dtype=torch.int64, device=dev)
            out.append((grp, tr, ro, table))
        return out

    def refine(item, tm=None):
        grp, tr, ro, table = item
        gd = gd_by_size[len(grp)]
        gd.reset(tr, ro)
        gd.set_pano_table(table)
        gd.run(NUM_ITER, timer=tm)
        res = gd.result().reshape(len(grp), B, -1)
        k = torch.argmin(res[:, :, 12], dim=1)             # per image: smallest loss of the last forward
        win = torch.gather(res, 1, k.reshape(-1, 1, 1).expand(-1, 1, res.shape[2]))[:, 0]
        results[grp[0]:grp[-1] + 1, :7] = win.index_select(1, cols)

    def barrier():
        if dist is not None:
            dist.barrier()

    def gather_results():
        """the path's only collective: every rank's result rows (RCCL all_gather; gloo in the CPU tests)"""
        if dist is None:
            return results[:K]
        src = results[:K].contiguous() if backend == "nccl" else results[:K].cpu()
        out = torch.empty(world * K, 16, device=src.device)
        dist.all_gather_into_tensor(out, src)
        return out

    def timed_pass(items, tm, with_gather):
        """EXACTLY K steps: barrier + synchronize, refine every timed image once (+ the result gather), synchronize +
        barrier; returns the MAX over ranks of the elapsed wall time."""
        if tm is not None:
            tm.reset()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in items:
            refine(it, tm)
        if with_gather:
            gather_results()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed

    def repeated(items, tm, with_gather):
        """timed passes until --min-seconds of measurement (the count is agreed between ranks: rank 0's clock decides);
        returns the list of pass times and the kernel-timer reading of the median pass"""
        times, kernels, total = [], [], 0.0
        while True:
            dt = timed_pass(items, tm, with_gather)
            times.append(dt)
            kernels.append(tm.read() if tm is not None else (0.0, 0))
            total += dt
            go = torch.tensor([1 if (total < args.min_seconds and len(times) < 200) else 0], device=dev if backend == "nccl" else "cpu")
            if dist is not None:
                dist.broadcast(go, src=0)
            if not int(go.item()):
                break
        mid = int(np.argsort(times)[len(times) // 2])
        return times, times[mid], kernels[mid]

    timed_items = prepare(make_groups(list(range(K)), ipl))
    warm_items = prepare(make_groups(list(range(K, n_img)), ipl))
    if args.prewarm_ms > 0 and warm_items:                      # part of the untimed setup, not of the W warm-up steps
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            refine(warm_items[0])
            torch.cuda.synchronize()
    for it in warm_items[:n_warm // ipl]:                      # the W warm-up steps (distinct images)
        refine(it)
    if dist is not None:                                       # first use of the collective sets up its connections: untimed
        gather_results()
    results[:, 14] = float(rank)                                # stamp: which rank produced the row
    pass_times, elapsed, (kernel_ms, launches) = repeated(timed_items, timer, True)
    # proof that the collective really spanned N ranks: distinct rank stamps among the gathered rows
    ranks_seen = int(torch.unique(gather_results()[:, 14]).numel())

    # accuracy of this rank's images (localize.py:239-247 formulas)
    errs = []
    res_host = results[:K].cpu().numpy()
    for i in range(K):
        R = ops.rot_from_ypr(torch.from_numpy(res_host[i, 3:6]))[0].cpu().numpy()
        errs.append(synth.pose_errors(res_host[i, :3], R, gts[i][0], synth.rot_from_ypr_np(gts[i][1])))
    errs = np.array(errs)

    # ---- the literal cfg-2 mode: ONE query image per launch chain (B poses per launch), same images, same protocol
    single = None
    if not args.no_single_image and ipl > 1:
        single_items = prepare(make_groups(list(range(K)), 1))
        refine(prepare(make_groups([K], 1))[0] if n_img > K else single_items[0])       # untimed: first launch of this grid shape
        s_times, s_elapsed, (s_kernel_ms, s_launches) = repeated(single_items, timer, True)
        s_bytes = BYTES_PER_POINT_POSE * N * B * s_launches
        single = {"value": B * K * world / s_elapsed, "ms_per_step": s_elapsed / K * 1e3, "images_per_launch": 1,
                  "poses_per_launch": B, "passes": len(s_times),
                  "roofline_achieved_GBs": s_bytes / (s_kernel_ms * 1e-3) / 1e9,
                  "frac": s_bytes / (s_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "avg_launch_ms": s_kernel_ms / max(s_launches, 1), "launches_timed": s_launches}

    if rank == 0:
        value = B * K * world / elapsed
        per_launch_ms = kernel_ms / max(launches, 1)
        groups = [it[0] for it in timed_items]
        # a launch of group g evaluates len(g) images x B candidates; every run times the same number of launches
        total_bytes = sum(BYTES_PER_POINT_POSE * N * B * len(g) * timed_per_run for g in groups)
        assert launches == timed_per_run * len(groups), (launches, timed_per_run, len(groups))
        alg_bytes = total_bytes / max(launches, 1)              # mean algorithmic bytes per timed launch
        achieved = total_bytes / (kernel_ms * 1e-3) / 1e9
        # counter-derived figures exist per launch SHAPE (workload, poses per launch, texel format); a run of any other
        # shape reports null rather than another shape's numbers
        uniform = len({len(g) for g in groups}) == 1
        roofs_key, roofs = lookup_roofs(args.roofs_json, args.workload, len(groups[0]) * B, fmt_name) if uniform else ("mixed launch shapes", None)
        traffic = roofs.get("hbm_bytes_per_launch") if roofs else None
        hbm_measured = None
        if traffic:
            hbm_measured = {"bytes_per_launch": traffic, "GBs": traffic / (per_launch_ms * 1e-3) / 1e9,
                            "frac_of_peak": traffic / (per_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "note": "memory-side bytes (2 x FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included) of this launch "
                                    "shape over this run's kernel time: the cloud and the panorama are cache resident, HBM is idle"}
        valu = None
        if roofs and roofs.get("valu_instr_per_point_pose"):
            # live VALU-issue figure: this run's kernel time against one wave64 VALU instruction per SIMD every 4 cycles
            # at the guide's 2400 MHz maximum clock (256 CUs x 4 SIMDs) — counted instructions only: transcendentals,
            # s_nop slots between dependent packed ops and clocks below the maximum all lower it
            wave_instr = roofs["valu_instr_per_point_pose"] * N * B * len(groups[0]) / 64.0
            issue_peak = 1024 * 2.4e9 / 4.0 * (per_launch_ms * 1e-3)
            valu = {"instr_per_point_pose": roofs["valu_instr_per_point_pose"], "busy_frac": roofs["valu_busy_frac"],
                    "issue_frac_live": wave_instr / issue_peak,
                    "issue_frac_is": "profiled wave64 VALU instructions per launch / (1024 SIMDs x 2400 MHz / 4 cycles x this run's "
                                     "average launch time)",
                    "source": roofs.get("source"),
                    "note": "wave64 VALU instructions per 64 point-poses and 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): "
                            "the kernel is VALU-issue bound, this is the roof with headroom left"}
        line = {
            "metric": "candidate-poses/s", "value": value, "unit": "candidate-poses/s", "n_gpus": world, "steps": K,
            "warmup": Wm, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d-point cloud, %dx%d panorama, %d candidate poses x %d GD iterations per query image"
                                   % (args.workload, N, W, H, B, NUM_ITER),
                       "images_per_gpu": K, "sharding": "query images round-robin over ranks, RCCL all_gather of results",
                       "mode": "omniloc_batch" if batch_mode else "omniloc", "images_per_launch": ipl,
                       "poses_per_launch": ipl * B, "texels": fmt_name, "warmup_images": "distinct from the timed ones"},
            "passes": len(pass_times), "pass_ms": {"min": min(pass_times) * 1e3, "median": elapsed * 1e3, "max": max(pass_times) * 1e3},
            "ranks_seen": ranks_seen, "devices_visible": n_dev,
            "pose_evals_per_s": value * NUM_ITER,
            "median_t_err_m": float(np.median(errs[:, 0])), "median_r_err_deg": float(np.median(errs[:, 1])),
            "single_image": single,
            "roofline": {"bound": "hbm", "kernel": "pcl_loss_kernel<G=%d, GRAD, %s>" % (2 if (B * ipl) % 2 == 0 else 1, {"u8": "RGBA8", "f16": "F16x4", "f32": "F32x4"}[fmt_name]),
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_rate_6290GBs": achieved / 6290.0,
                         "achieved_is": "ALGORITHMIC bytes (24 B x points x poses per launch, SURVEY.md 8d) / measured kernel time; "
                                        "not a bandwidth: see hbm_measured and valu",
                         "traffic": traffic, "traffic_key": roofs_key,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": per_launch_ms, "launches_timed": launches,
                         "hbm_measured": hbm_measured, "valu": valu},
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
