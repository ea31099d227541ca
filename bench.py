#!/usr/bin/env python3
"""bench.py — candidate-poses/s of the PICCOLO pose refinement on MI355X (BASELINE.json metric).

One STEP = the complete gradient-descent refinement of ONE query panorama: B candidate starting poses, num_iter = 100
iterations (configs/stanford_parallel.ini shape: lr 0.1, patience 5, factor 0.8), each iteration one fused
projection + bilinear sampling + loss + gradient pass over the whole cloud and one on-device optimiser epilogue.
candidate-poses/s = B * images / wall time.  Inputs (packed cloud, packed panoramas, starting poses) are resident in
HBM before the timed region starts.  Independent query images are sharded round-robin over the ranks (weak scaling:
every rank refines `steps` images); the only collective is the final all_gather of the results (RCCL).

    python bench.py                                   # 1 GPU, cfg2 (1M points, 2048x1024, 32 candidates)
    python bench.py --gpus 8 --steps K --warmup W     # starts 8 fresh rank processes itself (self_launch) and relays rank 0's line
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --steps K --warmup W     # or under torchrun

Output.  The LAST line of stdout is ONE compact JSON record below 4 KB (compact_line: what the driver parses — BENCH_r05.parsed was null
for a 24.5 KB line); the COMPLETE record (the `also` block, per-kernel roofs, notes) goes to the side file --also-json (bench_also.json).
  value / ms_per_step         default mode: 256 // B query images share one launch chain (cfg 4's shape on one GPU)
  single_image {...}          the literal cfg-2 mode: ONE query image per launch chain (B poses per launch)
  roofline {...}              the roof that BINDS the dominant kernel: VALU issue.  The cloud and the panorama are L2 /
                              Infinity-Cache resident (memory-side traffic = 5 % of the HBM peak), so SURVEY.md 8(d)'s
                              algorithmic-bytes figure has saturated (> 1.0) and says nothing about the kernel any more;
                              it is kept as roofline.algorithmic_hbm_frac.
    bound "valu"              achieved = wave64 VALU instructions per second = (instructions per point-pose from the rocprofv3
                              PMC passes of THIS launch shape and THIS library build, profiles/roofs.json) x point-poses per
                              launch / 64 / (this run's average launch time, HIP events minus the measured cost of an empty
                              event pair); peak = 1024 SIMDs x 2400 MHz / 4 cycles per wave64 instruction (a convention);
                              mix_ceiling_cycles_per_instr / frac_of_mix_ceiling: the same rate against what the kernel's OWN
                              instruction mix can issue (tools/roof_mix.py: ISA x measured class costs); fp32_flop_frac: its
                              flops against the 157.3 TFLOP/s vector peak.
                              profiles/roofs.json entries carry pcl_source_hash() of the library they were collected from; if
                              the loaded library differs, valu is null and the line falls back to bound "hbm" (algorithmic).
    roofline.traffic          memory-side bytes per launch from the same PMC passes (2 x FETCH_SIZE + WRITE_SIZE)
  also_brief {...}            one number per side measurement; the side file's `also` {...} holds them whole (N = 1 only, --no-also
                              skips): cfg 3, cfg 5, the reference's shipped shape (167k points x 6 candidates; 1 and 8 images per
                              launch chain) and the whole per-image pipeline (make_input + refinement: what the reference's
                              `time (s)` column measures, localize.py:208,222-223) at cfg-2 size
  cpu_baseline {...}          the C oracle (oracle/pcl_oracle.c compiled with OpenMP: a PORT of the reference's loss +
                              autograd, pinned to the reference by tests/golden) on the host cores
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))


def self_launch(argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the LAUNCHER.  It starts
    N fresh rank processes (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
    bench.py <same arguments>`: what the driver's own N > 1 command line is), waits, relays rank 0's JSON line as the last line
    of its own stdout and exits with the children's status.  It runs BEFORE torch or the HIP library are imported — the
    launcher never touches the GPU (stdlib only) and nothing here replaces a running program (`subprocess`, never `os.exec*`).
    Returns None when this process is an ordinary rank (N == 1, or started by torch.distributed.run)."""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = argv[i + 1]
        elif a.startswith("--gpus="):
            n = a.split("=", 1)[1]
    try:
        n = int(n)
    except ValueError:
        return None                                          # argparse will complain
    if n <= 1 or "WORLD_SIZE" in os.environ or "-h" in argv or "--help" in argv:
        return None
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as s:                           # a free port on the loopback interface
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    # dmabuf IPC: on this pool's host driver RCCL between processes fails with `hipIpcGetMemHandle: invalid argument` under the
    # legacy IPC mode.  A DEFAULT only: a value already in the environment (the driver's, or PCL_HSA_IPC_MODE_LEGACY to force one
    # for an experiment) wins, and every rank prints the value it runs with and where it came from.
    if "PCL_HSA_IPC_MODE_LEGACY" in env:
        env["HSA_ENABLE_IPC_MODE_LEGACY"], env["PCL_IPC_MODE_FROM"] = env["PCL_HSA_IPC_MODE_LEGACY"], "override"
    elif "HSA_ENABLE_IPC_MODE_LEGACY" in env:
        env["PCL_IPC_MODE_FROM"] = "environment"
    else:
        env["HSA_ENABLE_IPC_MODE_LEGACY"], env["PCL_IPC_MODE_FROM"] = "0", "default"
    env.setdefault("OMP_NUM_THREADS", "1")                   # (torchrun would set it, with a warning; cpu_baseline sizes its own pool)
    rc, last_json = 1, None
    for attempt in (0, 1):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", port, "--", os.path.abspath(__file__)] + list(argv)
        print("bench.py: launching %d ranks (HSA_ENABLE_IPC_MODE_LEGACY=%s, %s): %s" % (n, env["HSA_ENABLE_IPC_MODE_LEGACY"], env["PCL_IPC_MODE_FROM"],
                                                                                       " ".join(cmd)), file=sys.stderr, flush=True)
        # one pipe for the ranks' stdout AND stderr (each line is one write below PIPE_BUF: never spliced): everything they print goes
        # to the launcher's stderr as it comes, rank 0's compact JSON line alone goes to stdout at the end
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=os.getcwd())
        last_json, collective_failed = None, False
        for ln in child.stdout:
            if ln.startswith("{") and ln.rstrip().endswith("}"):
                last_json = ln.rstrip("\n")
            else:
                sys.stderr.write(ln)
                collective_failed |= "bench.py rank" in ln and " FAILED: " in ln
        rc = child.wait()
        sys.stderr.flush()
        if rc == 0 or last_json is not None or not collective_failed or attempt == 1 or env["PCL_IPC_MODE_FROM"] == "override":
            break
        # A collective (or the process group's set-up) failed on some rank.  The one known cause on this pool is the IPC mode; the first
        # real multi-GPU run must still produce a line, so ONE more launch of fresh rank processes with the other mode (a new
        # subprocess from this launcher, which has never touched the GPU — never a re-exec of a rank).
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = "1" if env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" else "0"
        env["PCL_IPC_MODE_FROM"] = "retry"
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
        print("bench.py: a collective failed (rc %d, no JSON line): launching once more with HSA_ENABLE_IPC_MODE_LEGACY=%s"
              % (rc, env["HSA_ENABLE_IPC_MODE_LEGACY"]), file=sys.stderr, flush=True)
    if last_json is not None:                                # rank 0's line is the one line on stdout
        print(last_json, flush=True)
    if rc == 0 and last_json is None:
        print("bench.py: the ranks exited 0 without a JSON line", file=sys.stderr)
        rc = 1
    return rc


if __name__ == "__main__":
    _rc = self_launch(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, REPO)

from piccolo_amd import _lib, ops, synth  # noqa: E402

WORKLOADS = {
    # name: (N points, H, W, B candidates, batch_mode)
    "cfg1": (100_000, 256, 512, 1, False),
    "cfg2": (1_000_000, 1024, 2048, 32, True),
    "cfg3": (1_000_000, 1024, 2048, 256, True),
    "cfg4": (1_000_000, 1024, 2048, 32, True),      # 64 query images in all, sharded over the ranks (steps = 64 / world)
    "cfg5": (10_000_000, 2048, 4096, 32, True),
    "shipped": (166_667, 1024, 2048, 6, True),      # the reference's stanford_parallel.ini: 1M points / sample_rate 6, num_input 6
}
NUM_ITER, LR, PATIENCE, FACTOR, QUANTILE = 100, 0.1, 5, 0.8, 0.05
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_POINT_POSE = 24      # xyz + rgb fp32 read once per pose evaluation
VALU_PEAK_GINSTR = 1024 * 2.4e9 / 4.0 / 1e9     # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at 2400 MHz
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector (non-matrix) peak
FMT_NAMES = {_lib.PANO_U8: "u8", _lib.PANO_F16: "f16", _lib.PANO_F32: "f32"}
# the candidate grid of the reference's Stanford configs (75 translations x 24 rotations = 1800 poses on the box room)
STANFORD_INIT = dict(max_yaw=2 * np.pi, min_yaw=0, max_pitch=2 * np.pi, min_pitch=0, max_roll=2 * np.pi, min_roll=0, z_prior=None,
                     sample_rate_for_init=None, trans_init_mode="quantile", x_max=None, x_min=None, y_max=None, y_min=None,
                     z_max=None, z_min=None, num_split_h=4, num_split_w=4, xy_only=False, num_trans=50, yaw_only=False, num_yaw=4,
                     num_pitch=4, num_roll=4, dataset="Stanford2D-3D-S")


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
            "kind_is": "the oracle itself: oracle/pcl_oracle.c (scalar C restatement of the reference's loss + autograd, pinned to the "
                       "reference's outputs by tests/golden) compiled with OpenMP; not the reference's torch code",
            "cfg1_end_to_end": plumbing,
            "sample": "%d fused loss+gradient pose evaluations over the full %d-point cloud (%.1f s), fp32 oracle/pcl_oracle.c "
                      "with OpenMP; one candidate = %d evaluations" % (n_pose, len(xyz), dt, NUM_ITER),
            "pose_evals_per_s": pose_evals_per_s}


def pipeline_kernel_roofs(tag):
    """The binding roof of every kernel of a pipeline shape — trim: L1 line lookups per cycle per CU, bin / resolve / z pass: whichever
    of VALU issue, LDS, memory-side atomics or bytes explains them — from the rocprofv3 counter passes of that pipeline
    (profiles/pipeline_roofs.json, written by profiles/summarize.py: PIPELINE_TAG).  Stamped with pcl_library_hash() of the library
    they were collected from; another build's numbers are not reported."""
    path = os.path.join(REPO, "profiles", "pipeline_roofs.json")
    try:
        entry = json.load(open(path)).get(tag)
    except Exception as exc:                                    # noqa: BLE001
        return {"unavailable": "cannot read %s: %s" % (path, exc)}
    if not entry:
        return {"unavailable": "no entry %r in profiles/pipeline_roofs.json" % tag}
    loaded = _lib.load().pcl_library_hash().decode()
    if entry.get("library_hash") != loaded:
        return {"unavailable": "stale: profiled library %s, loaded library %s" % (entry.get("library_hash"), loaded)}
    return {"source": entry.get("source"), "command": entry.get("command"), "library_hash": loaded,
            "kernels": {k: {"us": v.get("avg_duration_us_kernel_trace"), "binding_roof": v.get("binding_roof"), "frac": v.get("binding_frac"),
                            "all": v.get("roof_fractions")} for k, v in sorted(entry.get("kernels", {}).items())}}


def lookup_roofs(path, workload, poses_per_launch, fmt_name, lib_hash):
    """Counter-derived figures for exactly this launch shape (profiles/roofs.json, written by profiles/summarize.py from
    the rocprofv3 --pmc passes).  -> (key, entry or None, why-not or None).  An entry collected from a library whose loss
    kernel sources differ from the loaded library's (pcl_source_hash) is NOT returned: a stale instruction count must not
    be scored."""
    workload = {"cfg4": "cfg2"}.get(workload, workload)      # cfg 4 = cfg 2's cloud, panorama size and candidates per image
    key = "%s/poses%d/%s" % (workload, poses_per_launch, fmt_name)
    try:
        entry = json.load(open(path)).get(key)
    except Exception:
        return key, None, "no roofs file"
    if entry is None:
        return key, None, "launch shape never profiled"
    if entry.get("source_hash") != lib_hash:
        return key, None, "stale: profiled library %s, loaded library %s" % (entry.get("source_hash"), lib_hash)
    return key, entry, None


class Ranks:
    """The process group of the run (None for one process): barrier, the result gather, MAX over ranks."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, self.world, args.gpus))
        # (PCL_DIST_BACKEND=gloo lets several ranks share one GPU: used by the test-suite to run the N > 1 code path end to
        #  end on a single-GPU box; the driver's multi-GPU runs use the default, RCCL, one GPU per rank)
        self.backend = os.environ.get("PCL_DIST_BACKEND", "nccl")
        # dmabuf IPC for RCCL / tensor sharing between the ranks' processes (the pool's driver supports nothing else): the HSA runtime
        # reads this when the process first touches the GPU, so it is set BEFORE torch.cuda.set_device below — setting it next to
        # init_process_group, as rounds 1-4 did, only worked because the boxes export it already
        self.ipc_from = os.environ.get("PCL_IPC_MODE_FROM")          # set by the launcher: default / environment / override / retry
        if self.ipc_from is None:
            if "PCL_HSA_IPC_MODE_LEGACY" in os.environ:             # (experiments: what happens without dmabuf IPC)
                os.environ["HSA_ENABLE_IPC_MODE_LEGACY"], self.ipc_from = os.environ["PCL_HSA_IPC_MODE_LEGACY"], "override"
            elif "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ:
                self.ipc_from = "environment"
            else:
                os.environ["HSA_ENABLE_IPC_MODE_LEGACY"], self.ipc_from = "0", "default"
        self.ipc_mode = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
        self.test_fail_nccl = os.environ.get("PCL_BENCH_TEST_FAIL_NCCL") == "1"     # (test hook: RCCL's set-up "fails" on every rank)
        self.n_dev = torch.cuda.device_count()
        if self.n_dev < 1:
            raise SystemExit("bench.py needs an MI355X: torch.cuda.device_count() == 0")
        if self.backend == "nccl" and self.n_dev < args.gpus and not self.test_fail_nccl:
            raise SystemExit("--gpus %d but only %d GPU(s) are visible to this process (torch.cuda.device_count()): one rank per "
                             "GPU is required for the RCCL run" % (args.gpus, self.n_dev))
        dev_index = local_rank % self.n_dev if (self.backend != "nccl" or self.test_fail_nccl) else local_rank
        torch.cuda.set_device(dev_index)
        self.dev = torch.device("cuda", dev_index)
        self.dist, self.fallback = None, None
        if self.world > 1 or os.environ.get("PCL_BENCH_FORCE_DIST") == "1":   # (the env knob exercises the RCCL path at world size 1)
            import datetime
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            # An explicit, SHORT rendezvous / collective timeout: a rank that cannot reach the others must end the job with a reason
            # inside the driver's own time limit, not hang in it (torch's default is 10 minutes for nccl, 30 for gloo).
            self.timeout_s = float(os.environ.get("PCL_DIST_TIMEOUT_S", "180"))
            tmo = datetime.timedelta(seconds=self.timeout_s)
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if self.backend == "nccl" else "-"
            except Exception as exc:                                # noqa: BLE001
                rccl = "unknown (%s)" % exc
            print("bench.py rank %d/%d: pid %d, device cuda:%d of %d visible (%s), backend %s, RCCL %s, HSA_ENABLE_IPC_MODE_LEGACY=%s (from: %s), "
                  "rendezvous %s:%s, timeout %.0f s" % (self.rank, self.world, os.getpid(), dev_index, self.n_dev, torch.cuda.get_device_name(dev_index),
                                                        self.backend, rccl, self.ipc_mode, self.ipc_from, os.environ["MASTER_ADDR"],
                                                        os.environ["MASTER_PORT"], self.timeout_s), file=sys.stderr, flush=True)
            if os.environ.get("PCL_BENCH_TEST_FAIL_IF_IPC") == self.ipc_mode:       # (test hook of the launcher's one retry)
                self.fail("init_process_group(%s)" % self.backend, RuntimeError("PCL_BENCH_TEST_FAIL_IF_IPC=%s" % self.ipc_mode))
            try:
                if self.backend == "nccl":
                    if self.test_fail_nccl:
                        raise RuntimeError("PCL_BENCH_TEST_FAIL_NCCL")
                    dist.init_process_group("nccl", device_id=self.dev, timeout=tmo)     # RCCL
                    probe = torch.ones(1, device=self.dev)                               # the communicator's first use, inside the try:
                    dist.all_reduce(probe)                                               # IPC handles are exchanged here
                    torch.cuda.synchronize()
                    assert int(probe.item()) == self.world, (float(probe.item()), self.world)
                else:
                    dist.init_process_group(self.backend, timeout=tmo)
            except Exception as exc:                                # noqa: BLE001
                if self.backend != "nccl" or os.environ.get("PCL_NO_GLOO_FALLBACK") == "1" or self.ipc_from in ("default", "environment") and "PCL_IPC_MODE_FROM" in os.environ:
                    # (under bench.py's own launcher the first failure ends the job with rc 3: the launcher tries the other IPC mode once)
                    self.fail("init_process_group(%s)" % self.backend, exc)
                # Started by someone else's torch.distributed.run (the driver's N > 1 command) or already on the launcher's second
                # attempt: there is no further launch to hope for.  The path's ONLY collective is the gather of 16 floats per image, so
                # the run goes on with it over host memory (gloo) and says so in the line — a measured value with a labelled
                # transport instead of no line at all.
                print("bench.py rank %d/%d: RCCL set-up FAILED (%s: %s): falling back to gloo for the result gather"
                      % (self.rank, self.world, type(exc).__name__, str(exc).replace("\n", " | ")[:300]), file=sys.stderr, flush=True)
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:                                   # noqa: BLE001
                    pass
                try:
                    # a store of its own for the new group: under torch.distributed.run the AGENT hosts the store on MASTER_PORT and every
                    # rank is a client of it (nobody would host another port); without an agent rank 0 hosts one on MASTER_PORT + 1
                    addr, port = os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"])
                    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True":
                        store = dist.TCPStore(addr, port, self.world, False, tmo)
                    else:
                        store = dist.TCPStore(addr, port + 1, self.world, self.rank == 0, tmo)
                    dist.init_process_group("gloo", store=dist.PrefixStore("pcl_gloo_fallback", store), rank=self.rank, world_size=self.world,
                                            timeout=tmo)
                except Exception as exc2:                           # noqa: BLE001
                    self.fail("init_process_group(gloo fallback)", exc2)
                self.backend = "gloo"
                self.fallback = "RCCL set-up failed (%s): result gather over gloo" % type(exc).__name__
            self.dist = dist
        self.coll_dev = self.dev if self.backend == "nccl" else torch.device("cpu")

    def fail(self, what, exc):
        """a collective failed: ONE line with the reason on stderr and a non-zero exit (no JSON line: the launcher / driver sees rc != 0)"""
        print("bench.py rank %d/%d: %s FAILED: %s: %s" % (self.rank, self.world, what, type(exc).__name__, str(exc).replace("\n", " | ")[:600]),
              file=sys.stderr, flush=True)
        os._exit(3)                                                 # (not sys.exit: a hung communicator must not block the interpreter's teardown)

    def barrier(self):
        if self.dist is not None:
            # (test hook, tests/test_hip_harness.py::test_bench_names_a_dead_rank: the named rank leaves before its first barrier — the
            #  others must end with a reason inside the timeout instead of hanging)
            if os.environ.get("PCL_BENCH_TEST_DIE_RANK") == str(self.rank):
                print("bench.py rank %d/%d: leaving before the barrier (PCL_BENCH_TEST_DIE_RANK)" % (self.rank, self.world), file=sys.stderr, flush=True)
                os._exit(7)
            try:
                self.dist.barrier()
            except Exception as exc:                                # noqa: BLE001
                self.fail("barrier", exc)

    def gather(self, rows):
        """the path's only collective: every rank's result rows (RCCL all_gather; gloo in the CPU tests)"""
        if self.dist is None:
            return rows
        src = rows.contiguous() if self.backend == "nccl" else rows.cpu()
        out = torch.empty(self.world * src.shape[0], src.shape[1], device=src.device)
        try:
            self.dist.all_gather_into_tensor(out, src)
            if self.backend == "nccl":
                torch.cuda.synchronize()                            # (an asynchronous RCCL error surfaces here, inside the try)
        except Exception as exc:                                    # noqa: BLE001
            self.fail("all_gather_into_tensor (%d x %d floats per rank)" % (src.shape[0], src.shape[1]), exc)
        return out

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], device=self.coll_dev, dtype=torch.float64)
        try:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        except Exception as exc:                                    # noqa: BLE001
            self.fail("all_reduce(MAX)", exc)
        return float(t.item())

    def all_ranks(self, value):
        """[value of rank 0, ..., value of rank world - 1] on every rank (one float per rank: the per-rank step times)"""
        if self.dist is None:
            return [float(value)]
        src = torch.tensor([float(value)], device=self.coll_dev, dtype=torch.float64)
        out = torch.empty(self.world, device=self.coll_dev, dtype=torch.float64)
        try:
            self.dist.all_gather_into_tensor(out, src)
        except Exception as exc:                                    # noqa: BLE001
            self.fail("all_gather_into_tensor (per-rank times)", exc)
        return [float(v) for v in out.cpu()]

    def agree(self, flag):
        """rank 0's decision, on every rank"""
        if self.dist is None:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], device=self.coll_dev)
        self.dist.broadcast(t, src=0)
        return bool(int(t.item()))


class Scene:
    """One synthetic room resident in HBM: the packed cloud (replicated on every rank), its quantile box, and packed query
    panoramas / starting poses per image id, made on demand (untimed set-up)."""

    def __init__(self, N, H, W, dev):
        self.N, self.H, self.W, self.dev = N, H, W, dev
        self.xyz, self.rgb = synth.box_room(N, seed=0)
        self.X, self.C = torch.from_numpy(self.xyz).to(dev), torch.from_numpy(self.rgb).to(dev)
        self.cloud = ops.Cloud(self.X, self.C)
        self.box = ops.quantile_box(self.X, QUANTILE)
        self._img = {}

    def image(self, image_id, keep_img=False):
        """-> dict(pano, gt=(t, ypr), img (device tensor, only if keep_img))"""
        e = self._img.get(image_id)
        if e is None or (keep_img and "img" not in e):
            t_gt, ypr_gt = synth.gt_pose(image_id)
            cam = ops.transform_cloud(self.X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt))
            img = synth.quantise_like_image_file(ops.make_pano(cam, self.C, (self.H, self.W)))   # uint8-quantised like a decoded image file
            # (the texel format the product's refinement takes for this cloud / panorama: fp16 levels, RGBA8 for sparse clouds)
            e = {"pano": ops.Pano(img, fmt=ops.refine_texels(self.N, self.H, self.W)), "gt": (t_gt, ypr_gt)}
            if keep_img:
                e["img"] = img
            self._img[image_id] = e
        return e

    def starts(self, image_id, B):
        t_gt, ypr_gt = self.image(image_id)["gt"]
        tr, ro = synth.start_poses(t_gt, ypr_gt, B, seed=image_id)
        return torch.from_numpy(tr).to(self.dev), torch.from_numpy(ro).to(self.dev), (tr, ro)


def equal_groups(K, B, H, W, requested=0):
    """Images per launch chain.  auto: about 256 poses per launch, in EQUAL groups when the step count allows it (every timed
    launch then has one shape, the one the counter passes under profiles/ were taken for): the largest divisor of K that is at
    most 256 // B and at least half of it; otherwise groups of 256 // B with a shorter last one — and no more images than keep
    their packed panoramas (8 B per texel) within ~half of the 256 MiB Infinity Cache: cfg 5's 4096x2048 panoramas are 67 MB each
    (measured 330 / 328 / 312 candidate-poses/s at 1 / 2 / 4 images per launch), cfg 2's 16.8 MB (3260 / 3314 / 3295 at 4 / 8 / 16)."""
    if requested > 0:
        return max(1, min(requested, K))
    cache_cap = max(1, int(140e6 // ((H + 2) * (W + 2) * 8)))
    target = max(1, min(256 // B, K, cache_cap))
    divs = [d for d in range(target, 0, -1) if K % d == 0 and 2 * d >= target]
    return divs[0] if divs else target


class Measure:
    """Timed refinement of image ids `timed` (this rank's) of `scene` with B candidates each, `ipl` images per launch chain."""

    def __init__(self, scene, B, batch_mode, ranks, timer_stride):
        self.scene, self.B, self.batch_mode, self.ranks = scene, B, batch_mode, ranks
        self.stride = max(1, timer_stride)
        self.timed_per_run = len(range(0, NUM_ITER, self.stride))
        self.gd_by_size = {}
        self.cols = torch.tensor([0, 1, 2, 3, 4, 5, 12], device=scene.dev)

    def prepare(self, image_ids, ipl, row0=0):
        """launch groups of `ipl` images: (result rows, image ids, trans, rot, panorama table)"""
        sc, B, out = self.scene, self.B, []
        for s0 in range(0, len(image_ids), ipl):
            grp = image_ids[s0:s0 + ipl]
            m = len(grp)
            if m not in self.gd_by_size:
                tr0, ro0, _ = sc.starts(grp[0], B)
                self.gd_by_size[m] = ops.GradientDescent(sc.cloud, sc.image(grp[0])["pano"], tr0.repeat(m, 1), ro0.repeat(m, 1), sc.box,
                                                         lr=LR, patience=PATIENCE, factor=FACTOR, batch_mode=self.batch_mode)
            st = [sc.starts(i, B) for i in grp]
            tr = torch.cat([s[0] for s in st]).contiguous()
            ro = torch.cat([s[1] for s in st]).contiguous()
            table = torch.tensor([sc.image(i)["pano"].data.data_ptr() for i in grp for _ in range(B)], dtype=torch.int64, device=sc.dev)
            out.append((list(range(row0 + s0, row0 + s0 + m)), grp, tr, ro, table))
        return out

    def refine(self, item, results, tm=None):
        rows, grp, tr, ro, table = item
        gd = self.gd_by_size[len(grp)]
        gd.reset(tr, ro)
        gd.set_pano_table(table, images=len(grp))
        gd.run(NUM_ITER, timer=tm)
        res = gd.result().reshape(len(grp), self.B, -1)
        k = torch.argmin(res[:, :, 12], dim=1)             # per image: smallest loss of the last forward
        win = torch.gather(res, 1, k.reshape(-1, 1, 1).expand(-1, 1, res.shape[2]))[:, 0]
        results[rows[0]:rows[-1] + 1, :7] = win.index_select(1, self.cols)

    def timed_pass(self, items, results, K, tm, with_gather):
        """EXACTLY K steps: barrier + synchronize, refine every timed image once (+ the result gather), synchronize +
        barrier; returns the MAX over ranks of the elapsed wall time."""
        if tm is not None:
            tm.reset()
        torch.cuda.synchronize()
        self.ranks.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in items:
            self.refine(it, results, tm)
        if with_gather:
            self.ranks.gather(results[:K])
        torch.cuda.synchronize()
        self.last_local_s = time.perf_counter() - t0           # this rank's own K steps (+ gather), before it waits for the others
        self.ranks.barrier()
        torch.cuda.synchronize()
        return self.ranks.max_over_ranks(time.perf_counter() - t0)

    def repeated(self, items, results, K, tm, min_seconds, with_gather=True):
        """timed passes until `min_seconds` of measurement (the count is agreed between ranks: rank 0's clock decides);
        returns the list of pass times, the median pass and the kernel-timer reading of that pass"""
        times, kernels, locals_, total = [], [], [], 0.0
        while True:
            dt = self.timed_pass(items, results, K, tm, with_gather)
            times.append(dt)
            locals_.append(self.last_local_s)
            kernels.append(tm.read() if tm is not None else (0.0, 0))
            total += dt
            if not self.ranks.agree(total < min_seconds and len(times) < 200):
                break
        mid = int(np.argsort(times)[len(times) // 2])
        self.local_s_of_median_pass = locals_[mid]             # (the pass times are identical on every rank: the same `mid` everywhere)
        return times, times[mid], kernels[mid]


def kernel_figures(N, B, groups, timed_per_run, kernel_ms, launches, pair_ms):
    """Live loss-kernel figures of one measurement: average launch time (HIP events on the launch stream, minus what an empty
    event pair reads) and SURVEY.md 8(d)'s algorithmic bytes over it."""
    raw = kernel_ms / max(launches, 1)
    avg = max(raw - pair_ms, 1e-6)
    total_bytes = sum(BYTES_PER_POINT_POSE * N * B * len(g) * timed_per_run for g in groups)
    alg_bytes = total_bytes / max(launches, 1)              # mean algorithmic bytes per timed launch
    return {"avg_launch_ms": avg, "avg_launch_ms_raw": raw, "event_pair_ms": pair_ms, "launches_timed": launches,
            "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_GBs": alg_bytes / (avg * 1e-3) / 1e9}


def loss_kernel_name(lib, N, poses_per_launch, fmt_name):
    """the dominant kernel of a launch chain as rocprofv3 lists it (pcl_gd_plan: poses per block, one or two launches per iteration)"""
    import ctypes
    G, fused = ctypes.c_int(0), ctypes.c_int(0)
    lib.pcl_gd_plan(N, poses_per_launch, None, ctypes.byref(G), ctypes.byref(fused))
    tex = {"u8": "RGBA8", "f16": "F16x4", "f32": "F32x4"}[fmt_name]
    if fused.value:
        return "pcl_loss_fused_kernel<G=%d, %s> (one launch per GD iteration: finishes the previous iteration in its prologue)" % (G.value, tex)
    return "pcl_loss_kernel<G=%d, GRAD, %s>" % (G.value, tex)


def valu_roof(roofs, N, poses_per_launch, avg_launch_ms):
    """The binding roof: wave64 VALU instructions per second of this run against one instruction per SIMD per 4 cycles."""
    wave_instr = roofs["valu_instr_per_point_pose"] * N * poses_per_launch / 64.0
    achieved = wave_instr / (avg_launch_ms * 1e-3) / 1e9
    out = {"achieved": achieved, "peak": VALU_PEAK_GINSTR, "unit": "G wave64 VALU instr/s", "frac": achieved / VALU_PEAK_GINSTR,
           "instr_per_point_pose": roofs["valu_instr_per_point_pose"], "wave_instr_per_launch": wave_instr,
           "busy_frac_profiled": roofs.get("valu_busy_frac"), "source": roofs.get("source"), "source_hash": roofs.get("source_hash")}
    # What `frac` is a fraction OF: the 4-cycle peak is a convention; the kernel's own instruction mix, priced with the measured issue
    # cost of every opcode class (tools/roof_mix.py: ISA of the loop x tools/micro/valu_rate.hip's table), needs `mix` cycles per
    # instruction — frac_of_mix_ceiling is the same instruction rate against 1024 SIMDs x 2.4 GHz / mix; fp32_flop_frac the flops of
    # that mix (fma = 2, packed = twice) against the 157.3 TFLOP/s fp32 vector peak.
    mix = roofs.get("mix_ceiling_cycles_per_instr")
    if mix:
        out["mix_ceiling_cycles_per_instr"] = mix
        out["frac_of_mix_ceiling"] = out["frac"] * mix / 4.0
    if roofs.get("fp32_flops_per_point_pose"):
        out["fp32_flops_per_point_pose"] = roofs["fp32_flops_per_point_pose"]
        out["fp32_flop_frac"] = roofs["fp32_flops_per_point_pose"] * N * poses_per_launch / (avg_launch_ms * 1e-3) / (FP32_VECTOR_PEAK_TFLOPS * 1e12)
    return out


def run_side(name, ranks, args, lib_hash, pair_ms, timer_stride, scenes, K, ipl, min_seconds=0.6):
    """One entry of the `also` block: workload `name` measured like the headline (distinct warm-up group, repeated timed
    passes, kernel timer), shorter.  -> dict"""
    N, H, W, B, batch_mode = WORKLOADS[name]
    key = (N, H, W)
    if key not in scenes:
        scenes[key] = Scene(N, H, W, ranks.dev)
    sc = scenes[key]
    m = Measure(sc, B, batch_mode, ranks, timer_stride)
    timed = list(range(K))
    warm = [1_000_000 + i for i in range(ipl)]
    items, witems = m.prepare(timed, ipl), m.prepare(warm, ipl, row0=K)
    results = torch.zeros(K + ipl, 16, device=sc.dev)
    timer = ops.KernelTimer(m.timed_per_run * (K + 1), stride=m.stride)
    m.refine(witems[0], results)
    torch.cuda.synchronize()
    # wall time WITHOUT the kernel timer (an event pair around every 10th launch costs ~0.7 us per iteration: 6 % of a 12 us
    # iteration at the shipped shape, nothing at cfg 2), then one timed pass for the launch time
    times, elapsed, _ = m.repeated(items, results, K, None, min_seconds, with_gather=False)
    m.timed_pass(items, results, K, timer, False)
    kernel_ms, launches = timer.read()
    groups = [it[1] for it in items]
    kf = kernel_figures(N, B, groups, m.timed_per_run, kernel_ms, launches, pair_ms)
    fmt = FMT_NAMES[sc.image(0)["pano"].fmt]
    rkey, roofs, why = lookup_roofs(args.roofs_json, name, ipl * B, fmt, lib_hash)
    res_host = results[:K].cpu().numpy()
    errs = []
    for i in range(K):
        R = ops.rot_from_ypr(torch.from_numpy(res_host[i, 3:6]))[0].cpu().numpy()
        gt = sc.image(i)["gt"]
        errs.append(synth.pose_errors(res_host[i, :3], R, gt[0], synth.rot_from_ypr_np(gt[1])))
    errs = np.array(errs)
    out = {"workload": "%d points, %dx%d, %d candidates, %d image(s) per launch chain" % (N, W, H, B, ipl),
           "kernel": loss_kernel_name(_lib.load(), N, ipl * B, fmt),
           "value": B * K / elapsed, "unit": "candidate-poses/s", "ms_per_step": elapsed / K * 1e3, "steps": K, "passes": len(times),
           "poses_per_launch": ipl * B, "texels": fmt, "avg_launch_ms": kf["avg_launch_ms"],
           "us_per_iteration": elapsed / (K / ipl) / NUM_ITER * 1e6,
           "algorithmic_hbm_frac": kf["algorithmic_GBs"] / HBM_PEAK_GBS,
           "valu_frac": valu_roof(roofs, N, ipl * B, kf["avg_launch_ms"])["frac"] if roofs else None,
           "roofs_key": rkey, "valu_unavailable": why,
           "median_t_err_m": float(np.median(errs[:, 0])), "median_r_err_deg": float(np.median(errs[:, 1]))}
    return out


def pipeline_block(sc, n_images=3, num_input=32, num_intermediate=64):
    """The whole per-image pipeline through the product's call surface — what the reference's `time (s)` CSV column measures
    (localize.py:208,222-223): make_input (1800-pose grid -> loss trim to `num_intermediate` -> histogram trim to `num_input`) +
    omniloc_batch (`num_input` candidates x 100 iterations); medians over `n_images` query images after one untimed image.
    cfg-2 size: 1M points, 64 -> 32; the reference's shipped stanford_parallel.ini: 166 667 points, 50 -> 6."""
    from piccolo_amd import omniloc as po
    from piccolo_amd import utils

    class Cfg:
        lr, num_iter, patience, factor, out_of_room_quantile = LR, NUM_ITER, PATIENCE, FACTOR, QUANTILE
    Cfg.num_input = num_input

    rows, totals = [], []
    imgs = []
    for j in range(n_images + 1):
        e = sc.image(2_000_000 + j, keep_img=True)
        imgs.append((e["img"], e["gt"]))
        del e["img"]
    for stage_sync in (True, False):
        for j, (img, gt) in enumerate(imgs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr, ro = utils.make_input(img, sc.X, sc.C, num_input, STANFORD_INIT, "loss_histogram", num_intermediate)
            if stage_sync:
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            res = po.omniloc_batch(img, sc.X, sc.C, tr, ro, Cfg(), {})
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if j == 0:
                continue                                     # untimed: first image of the pass
            if stage_sync:
                te, re = synth.pose_errors(res[0].numpy(), res[1].numpy(), gt[0], synth.rot_from_ypr_np(gt[1]))
                rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, te, re))
            else:
                totals.append((t2 - t0) * 1e3)
    rows = np.array(rows)
    return {"what": "per query image (%d points, %dx%d): make_input (75 x 24 = 1800-pose grid -> %d -> %d) + omniloc_batch (%d candidates x "
                    "100 iterations), medians over %d images" % (sc.N, sc.W, sc.H, num_intermediate, num_input, num_input, n_images),
            "total_ms": float(np.median(totals)),
            "total_is": "one clock around make_input + omniloc_batch, as the reference's `time (s)` column is taken (localize.py:208,222-223): "
                        "no device synchronisation between the two calls; the stage times below come from a second pass with one",
            "make_input_ms": float(np.median(rows[:, 0])), "refine_ms": float(np.median(rows[:, 1])),
            "total_with_stage_sync_ms": float(np.median(rows[:, 0] + rows[:, 1])),
            "median_t_err_m": float(np.median(rows[:, 2])), "median_r_err_deg": float(np.median(rows[:, 3]))}


def host_buffers_block(sc, B=32, n_images=3):
    """What `value` leaves out when the caller hands over HOST buffers (the product's Python surface accepts CPU tensors; the C ABI takes
    device pointers only): per query image at cfg-2 size, one clock around upload + omniloc_batch (its packing launches included),
    (a) image and start poses uploaded per call, the cloud tensors resident — a room's cloud is read once per room (localize.py:64-97),
    (b) the cloud uploaded per call as well (a fresh tensor: Morton order, pack and quantile box are redone).  Pageable host memory."""
    from piccolo_amd import omniloc as po

    class Cfg:
        lr, num_iter, patience, factor, out_of_room_quantile, num_input = LR, NUM_ITER, PATIENCE, FACTOR, QUANTILE, B
    xyz_h, rgb_h = torch.from_numpy(sc.xyz), torch.from_numpy(sc.rgb)
    host = []
    for j in range(n_images + 1):
        e = sc.image(3_000_000 + j, keep_img=True)
        host.append((e["img"].cpu(), sc.starts(3_000_000 + j, B)[2]))
        del e["img"]
    rows = {"image": [], "cloud_and_image": []}
    for mode in rows:
        for j, (img_h, (tr_h, ro_h)) in enumerate(host):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            X, C = (xyz_h.to(sc.dev), rgb_h.to(sc.dev)) if mode == "cloud_and_image" else (sc.X, sc.C)
            img = img_h.to(sc.dev)
            tr, ro = torch.from_numpy(tr_h).to(sc.dev), torch.from_numpy(ro_h).to(sc.dev)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            po.omniloc_batch(img, X, C, tr, ro, Cfg(), {})
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if j:
                rows[mode].append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    a, b = np.array(rows["image"]), np.array(rows["cloud_and_image"])
    mb = (xyz_h.numel() * xyz_h.element_size() + rgb_h.numel() * rgb_h.element_size()) / 1e6
    img_mb = host[0][0].numel() * host[0][0].element_size() / 1e6
    return {"what": "one image at a time, %d points, %dx%d, %d candidates x %d iterations; omniloc_batch from HOST buffers (pageable), medians over "
                    "%d images; `value` of the line is measured with everything resident and is not this" % (sc.N, sc.W, sc.H, B, NUM_ITER, n_images),
            "image_per_call": {"upload_MB": img_mb, "upload_ms": float(np.median(a[:, 0])), "total_ms": float(np.median(a[:, 1])),
                               "candidate_poses_per_s": B / (float(np.median(a[:, 1])) * 1e-3)},
            "cloud_and_image_per_call": {"upload_MB": mb + img_mb, "upload_ms": float(np.median(b[:, 0])), "total_ms": float(np.median(b[:, 1])),
                                         "candidate_poses_per_s": B / (float(np.median(b[:, 1])) * 1e-3)}}


def pipeline_images_block(sc, ipl=8, n_groups=3, num_input=6, num_intermediate=50):
    """The same pipeline for `ipl` query images of the room at a time, through the product's multi-image surface (what the dataset
    loops run with cfg images_per_launch): make_input_images (one trim launch over image x translation x rotation, one selection
    launch) + omniloc_batch_images (all ipl x num_input candidates in one launch chain).  ms per image, median over `n_groups`
    groups after one untimed group."""
    from piccolo_amd import omniloc as po
    from piccolo_amd import utils

    class Cfg:
        lr, num_iter, patience, factor, out_of_room_quantile = LR, NUM_ITER, PATIENCE, FACTOR, QUANTILE
    Cfg.num_input = num_input

    groups = []
    for g in range(n_groups + 1):
        imgs, gts = [], []
        for j in range(ipl):
            e = sc.image(3_000_000 + g * ipl + j, keep_img=True)
            imgs.append(e.pop("img"))
            gts.append(e["gt"])
        groups.append((imgs, gts))
    rows, errs = [], []
    for stage_sync in (True, False):
        for g, (imgs, gts) in enumerate(groups):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            starts = utils.make_input_images(imgs, sc.X, sc.C, num_input, STANFORD_INIT, "loss_histogram", num_intermediate)
            if stage_sync:
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            res = po.omniloc_batch_images(imgs, sc.X, sc.C, [s[0] for s in starts], [s[1] for s in starts], Cfg())
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if g == 0:
                continue
            rows.append((stage_sync, (t1 - t0) * 1e3 / ipl, (t2 - t1) * 1e3 / ipl, (t2 - t0) * 1e3 / ipl))
            if stage_sync:
                errs += [synth.pose_errors(r[0].numpy(), r[1].numpy(), gt[0], synth.rot_from_ypr_np(gt[1])) for r, gt in zip(res, gts)]
    rows, errs = np.array(rows), np.array(errs)
    staged, free = rows[rows[:, 0] == 1], rows[rows[:, 0] == 0]
    return {"what": "per query image, %d images of the room at a time (%d points, %dx%d): make_input_images (1800-pose grid -> %d -> %d per image) "
                    "+ omniloc_batch_images (%d x %d candidates x 100 iterations in one launch chain), medians over %d groups"
                    % (ipl, sc.N, sc.W, sc.H, num_intermediate, num_input, ipl, num_input, n_groups),
            "images_per_launch": ipl, "total_ms_per_image": float(np.median(free[:, 3])),
            "make_input_ms_per_image": float(np.median(staged[:, 1])), "refine_ms_per_image": float(np.median(staged[:, 2])),
            "median_t_err_m": float(np.median(errs[:, 0])), "median_r_err_deg": float(np.median(errs[:, 1]))}


def depth_mask_block(sc, B, n_images=2, panorama_grid=True):
    """north_star's scatter-min depth mask in the loop: the refinement of one query image's B candidates through the GradientDescent
    engine with cfg depth_mask — at EVERY iteration fill + z pass (LDS-tiled atomicMin into each candidate's own z-buffer) for the
    poses that iteration evaluates, then the loss launch that looks each point's cell up (no mark pass, no byte mask; DESIGN.md
    section 4.5).  `default_grid`: the product's default (pcl_depth_default: the z-buffer from every 2nd point of a 1M-point cloud
    on the grid that count calls for, >= 12 occluder samples per cell; every point tested); `every_point`: depth_stride = 1, every
    point builds the z-buffer (on its finer grid); `panorama_grid`: the z-buffer at the panorama's resolution with tau 0.02 — round 4's definition, kept as a priced comparison
    (tools/depth_recall.py: it finds a third of the occluded points); `plain`: the same refinements without the mask, the
    denominator of `x_plain`.  Whole 100-iteration refinements timed from the host; medians over `n_images` images."""
    dh, dw, dtau, dst = ops.default_depth(sc.N, sc.H, sc.W)
    h1, w1, t1, _ = ops.default_depth(sc.N, sc.H, sc.W, stride=1)
    out = {"workload": "%d points, %dx%d, %d candidates, 1 image per launch chain, 100 iterations" % (sc.N, sc.W, sc.H, B),
           "default_depth_res": [dh, dw], "default_depth_tau": dtau, "default_depth_stride": dst,
           "every_point_depth_res": [h1, w1], "every_point_depth_tau": t1}
    variants = [("plain", dict()), ("default_grid", dict(depth_mask=True)), ("every_point", dict(depth_mask=True, depth_stride=1))]
    if panorama_grid:
        variants.append(("panorama_grid", dict(depth_mask=True, depth_res=(sc.H, sc.W), depth_tau=0.02)))
    for name, kw in variants:
        times, errs = [], []
        gd = None
        for j in range(n_images + 1):
            image_id = 4_000_000 + j
            tr, ro, _ = sc.starts(image_id, B)
            pano = sc.image(image_id)["pano"]
            if gd is None:
                gd = ops.GradientDescent(sc.cloud, pano, tr, ro, sc.box, lr=LR, patience=PATIENCE, factor=FACTOR, batch_mode=True, **kw)
            else:
                gd.reset(tr, ro)
                gd.set_pano_table(torch.full((B,), pano.data.data_ptr(), dtype=torch.int64, device=sc.dev))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gd.run(NUM_ITER)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if j == 0:
                continue                                      # untimed: first refinement (allocations, first touch)
            res = gd.result().cpu().numpy()
            k = int(np.argmin(res[:, 12]))
            gt = sc.image(image_id)["gt"]
            R = ops.rot_from_ypr(torch.from_numpy(res[k:k + 1, 3:6]))[0].cpu().numpy()
            errs.append(synth.pose_errors(res[k, :3], R, gt[0], synth.rot_from_ypr_np(gt[1])))
            times.append(dt)
        errs = np.array(errs)
        out[name] = {"value": B / float(np.median(times)), "unit": "candidate-poses/s", "us_per_iteration": float(np.median(times)) / NUM_ITER * 1e6,
                     "median_t_err_m": float(np.median(errs[:, 0])), "median_r_err_deg": float(np.median(errs[:, 1]))}
        del gd
    for name in ("default_grid", "every_point", "panorama_grid"):
        if name in out:
            out[name]["x_plain"] = out[name]["us_per_iteration"] / out["plain"]["us_per_iteration"]
    return out


COMPACT_LIMIT = 4096            # bytes: the LAST stdout line must stay below this (the driver keeps an 8 KB tail of stdout)


def _r(v, sig=6):
    """floats rounded to `sig` significant digits (the compact line is for parsing, the full precision is in the side file)"""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    if v != v or v in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (sig, v))


def _pick(d, keys):
    return {k: _r(d.get(k)) for k in keys if d is not None and k in d}


def compact_line(full, also_path=None):
    """The ONE line the driver parses: headline, config, single_image, roofline, cpu_baseline, checks — numbers and short names only,
    always below COMPACT_LIMIT bytes.  Everything else of `full` (the `also` block, per-kernel roofs, the prose `*_is` strings) goes to
    the side file `also_path` (bench_also.json), never to the last stdout line: BENCH_r05.parsed was null because the line had grown
    to 24.5 KB."""
    cfg, roof, single, cpu = full.get("config") or {}, full.get("roofline") or {}, full.get("single_image"), full.get("cpu_baseline")
    valu = roof.get("valu") or {}
    alg, hbm = roof.get("algorithmic_hbm") or {}, roof.get("hbm_measured") or {}
    line = {k: _r(full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                         "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:160]}
    line["config"].update(_pick(cfg, ("images_per_launch", "poses_per_launch", "texels", "images_per_gpu", "mode")))
    if single:
        line["single_image"] = _pick(single, ("value", "ms_per_step", "poses_per_launch", "valu_frac", "avg_launch_ms"))
        line["single_image"]["workload"] = "literal %s: ONE query image per launch chain" % str(cfg.get("workload", "")).split(" ")[0]
    rl = _pick(roof, ("bound", "achieved", "peak", "unit", "frac"))
    rl["kernel"] = str(roof.get("kernel", ""))[:80]
    rl.update(_pick(roof, ("avg_launch_ms", "launches_timed", "traffic")))
    rl["algorithmic_hbm_frac"], rl["hbm_measured_frac"] = _r(alg.get("frac")), _r(hbm.get("frac_of_peak"))
    rl.update(_pick(valu, ("instr_per_point_pose", "busy_frac_profiled", "mix_ceiling_cycles_per_instr", "frac_of_mix_ceiling", "fp32_flop_frac")))
    rl["profile"] = str(valu.get("source", roof.get("traffic_key", "")))[:60]
    rl["source_hash"] = roof.get("source_hash_loaded_library")
    line["roofline"] = rl
    if cpu:
        line["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "kind", "pose_evals_per_s"))
        line["cpu_baseline"]["sample"] = str(cpu.get("sample", ""))[:150]
        e2e = cpu.get("cfg1_end_to_end")
        if e2e:
            line["cpu_baseline"]["cfg1_end_to_end"] = _pick(e2e, ("t_err_m", "r_err_deg", "seconds"))
    line["checks"] = _pick(full.get("checks") or {}, ("kernel_ms_per_step", "kernel_time_within_step", "kernel_share_of_step"))
    line.update(_pick(full, ("ranks_seen", "devices_visible", "dist_backend", "passes", "median_t_err_m", "median_r_err_deg")))
    pr = full.get("per_rank_ms_per_step")
    if pr:
        line["per_rank_ms_per_step"] = _pick(pr, ("min", "max"))
    n1 = full.get("n1_value_same_build")
    if n1:
        line["n1_value_same_build"] = _pick(n1, ("value", "ms_per_step"))
    line.update(_pick(full, ("pass_ms", "ipc_mode_legacy", "ipc_mode_from")))
    if full.get("dist_fallback"):
        line["dist_fallback"] = str(full["dist_fallback"])[:100]
    if "pass_ms" in line and isinstance(line["pass_ms"], dict):
        line["pass_ms"] = {k: _r(v) for k, v in line["pass_ms"].items()}
    also = full.get("also") or {}
    brief = {}
    for k, v in also.items():                                   # one number per side measurement; the blocks themselves are in the side file
        if isinstance(v, dict):
            for key in ("value", "total_ms", "total_ms_per_image"):
                if key in v:
                    brief[k] = _r(v[key], 5)
                    break
    if "depth_mask_cfg2" in also and isinstance(also["depth_mask_cfg2"], dict):
        for name in ("plain", "default_grid", "every_point"):
            if name in also["depth_mask_cfg2"]:
                brief["depth_mask_cfg2." + name] = _r(also["depth_mask_cfg2"][name].get("value"), 5)
    if also.get("error"):
        brief["error"] = str(also["error"])[:120]
    if brief:
        line["also_brief"] = brief
    if also_path:
        line["also_file"] = os.path.relpath(also_path, REPO) if also_path.startswith(REPO) else also_path
    # the size is a contract, not a hope: shed the optional blocks, least important first, until the line fits
    for drop in ("also_brief", "pass_ms", "n1_value_same_build", "per_rank_ms_per_step", "checks", "single_image"):
        if len(json.dumps(line)) < COMPACT_LIMIT - 64:
            break
        line.pop(drop, None)
    assert len(json.dumps(line)) < COMPACT_LIMIT, len(json.dumps(line))
    return line


def emit(full, also_path):
    """side file first (the complete record), then the compact line as the LAST line of stdout"""
    if also_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(also_path)), exist_ok=True)
            with open(also_path, "w") as f:
                json.dump(full, f, indent=1)
            print("bench.py: complete record (also / kernel_roofs / notes) written to %s" % also_path, file=sys.stderr, flush=True)
        except OSError as exc:
            print("bench.py: could not write %s: %s" % (also_path, exc), file=sys.stderr, flush=True)
            also_path = None
    print(json.dumps(compact_line(full, also_path)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0, help="size of the CPU sample (seconds of pose evaluations)")
    ap.add_argument("--no-single-image", action="store_true", help="skip the one-image-per-launch-chain pass")
    ap.add_argument("--no-also", action="store_true", help="skip the `also` block (cfg 3, cfg 5, shipped shape, per-image pipeline)")
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
    ap.add_argument("--also-json", default=os.path.join(REPO, "bench_also.json"),
                    help="side file for the COMPLETE record (the `also` block, per-kernel roofs, notes); the last stdout line is the compact "
                         "line only ('' = no side file)")
    args = ap.parse_args()

    ranks = Ranks(args)
    rank, world, dev = ranks.rank, ranks.world, ranks.dev
    lib = _lib.load()
    lib_hash = lib.pcl_source_hash().decode()

    N, H, W, B, batch_mode = WORKLOADS[args.workload]
    if args.workload == "cfg4":
        args.steps = max(1, 64 // world)
    K, Wm = args.steps, args.warmup

    # Images are refined in groups of `ipl`: the group's ipl * B candidates go through ONE chain of launches (shared
    # cloud, per-candidate panorama pointer).  More poses per launch share each cloud chunk in L2 and amortise the
    # per-block costs; the candidates stay independent (own Adam / scheduler state), so per-image results are the same.
    ipl = equal_groups(K, B, H, W, args.images_per_launch)
    # warm-up steps refine their OWN images (ids beyond every rank's timed ones), in whole launch groups of the timed size
    n_warm = ((Wm + ipl - 1) // ipl) * ipl if Wm > 0 else 0
    # (the untimed clock pre-warm needs a launch group of its own images even with --warmup 0)
    n_extra = max(n_warm, ipl if args.prewarm_ms > 0 else 0)

    # ---- untimed setup: synthetic room, one panorama per query image, everything packed and resident in HBM
    scenes = {(N, H, W): Scene(N, H, W, dev)}
    sc = scenes[(N, H, W)]
    # timed image i of this rank is query image rank + i * world (round-robin sharding); warm-up images follow
    timed_ids = [rank + i * world for i in range(K)]
    warm_ids = [1_000_000 + rank + i * world for i in range(n_extra)]
    if rank == 0:
        e0 = sc.image(timed_ids[0], keep_img=True)
        img0_host = e0.pop("img").cpu().numpy()
        start0_host = sc.starts(timed_ids[0], B)[2]
    for i in timed_ids + warm_ids:
        sc.image(i)
    fmt_name = FMT_NAMES[sc.image(timed_ids[0])["pano"].fmt]
    results = torch.zeros(K + n_extra, 16, device=dev)
    # HIP-event pairs around every TIMER_STRIDE-th loss launch of the timed region (each pair costs a few us of GPU
    # timeline; bracketing all 100 launches of a refinement slows cfg 1 by 2x and cfg 2 by ~2 %)
    m = Measure(sc, B, batch_mode, ranks, args.timer_stride)
    timer = ops.KernelTimer(m.timed_per_run * (K + 1), stride=m.stride)

    timed_items = m.prepare(timed_ids, ipl)
    warm_items = m.prepare(warm_ids, ipl, row0=K)
    if args.prewarm_ms > 0 and warm_items:                      # part of the untimed setup, not of the W warm-up steps
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            m.refine(warm_items[0], results)
            torch.cuda.synchronize()
    for it in warm_items[:n_warm // ipl]:                      # the W warm-up steps (distinct images)
        m.refine(it, results)
    if ranks.dist is not None:                                 # first use of the collective sets up its connections: untimed
        ranks.gather(results[:K])
    torch.cuda.synchronize()
    pair_ms = timer.calibrate(64)                              # what an event pair reads with nothing in between
    results[:, 14] = float(rank)                                # stamp: which rank produced the row
    pass_times, elapsed, (kernel_ms, launches) = m.repeated(timed_items, results, K, timer, args.min_seconds)
    # proof that the collective really spanned N ranks: distinct rank stamps among the gathered rows
    ranks_seen = int(torch.unique(ranks.gather(results[:K])[:, 14]).numel())
    # every rank's OWN time for the median pass's K steps (before it waited for the others): a straggler shows as max >> min
    per_rank_s = ranks.all_ranks(m.local_s_of_median_pass)

    # accuracy of this rank's images (localize.py:239-247 formulas)
    errs = []
    res_host = results[:K].cpu().numpy()
    for i in range(K):
        R = ops.rot_from_ypr(torch.from_numpy(res_host[i, 3:6]))[0].cpu().numpy()
        gt = sc.image(timed_ids[i])["gt"]
        errs.append(synth.pose_errors(res_host[i, :3], R, gt[0], synth.rot_from_ypr_np(gt[1])))
    errs = np.array(errs)

    # ---- the literal cfg-2 mode: ONE query image per launch chain (B poses per launch), same images, same protocol
    single = None
    if not args.no_single_image and ipl > 1:
        single_items = m.prepare(timed_ids, 1)
        m.refine(m.prepare(warm_ids[:1], 1, row0=K)[0] if warm_ids else single_items[0], results)   # untimed: first launch of this grid shape
        s_times, s_elapsed, (s_kernel_ms, s_launches) = m.repeated(single_items, results, K, timer, args.min_seconds)
        s_kf = kernel_figures(N, B, [it[1] for it in single_items], m.timed_per_run, s_kernel_ms, s_launches, pair_ms)
        s_key, s_roofs, s_why = lookup_roofs(args.roofs_json, args.workload, B, fmt_name, lib_hash)
        single = {"value": B * K * world / s_elapsed, "ms_per_step": s_elapsed / K * 1e3, "images_per_launch": 1,
                  "poses_per_launch": B, "passes": len(s_times),
                  "valu_frac": valu_roof(s_roofs, N, B, s_kf["avg_launch_ms"])["frac"] if s_roofs else None,
                  "valu_unavailable": s_why, "roofs_key": s_key,
                  "algorithmic_hbm_frac": s_kf["algorithmic_GBs"] / HBM_PEAK_GBS,
                  "avg_launch_ms": s_kf["avg_launch_ms"], "launches_timed": s_launches,
                  "kernel_ms_per_step": NUM_ITER * s_kf["avg_launch_ms"]}

    line = None
    if rank == 0:
        value = B * K * world / elapsed
        groups = [it[1] for it in timed_items]
        assert launches == m.timed_per_run * len(groups), (launches, m.timed_per_run, len(groups))
        kf = kernel_figures(N, B, groups, m.timed_per_run, kernel_ms, launches, pair_ms)
        # counter-derived figures exist per launch SHAPE (workload, poses per launch, texel format) and per library build;
        # a run of any other shape, or of a library with other loss-kernel sources, reports null rather than another's numbers
        uniform = len({len(g) for g in groups}) == 1
        if uniform:
            roofs_key, roofs, why = lookup_roofs(args.roofs_json, args.workload, len(groups[0]) * B, fmt_name, lib_hash)
        else:
            roofs_key, roofs, why = "mixed launch shapes", None, "mixed launch shapes"
        traffic = roofs.get("hbm_bytes_per_launch") if roofs else None
        hbm_measured = None
        if traffic:
            hbm_measured = {"bytes_per_launch": traffic, "GBs": traffic / (kf["avg_launch_ms"] * 1e-3) / 1e9,
                            "frac_of_peak": traffic / (kf["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "note": "memory-side bytes (2 x FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included) of this launch "
                                    "shape over this run's kernel time: the cloud and the panorama are cache resident, HBM is idle"}
        algorithmic = {"achieved": kf["algorithmic_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kf["algorithmic_GBs"] / HBM_PEAK_GBS,
                       "bytes_per_launch": kf["algorithmic_bytes_per_launch"], "saturated": kf["algorithmic_GBs"] / HBM_PEAK_GBS > 0.9,
                       "is": "SURVEY.md 8(d)'s ALGORITHMIC figure: 24 B x points x poses per launch / kernel time.  Not a bandwidth and no "
                             "longer a roof: a block reads its cloud chunk once for the two poses it evaluates, out of L2, so the figure "
                             "can pass 1.0 while the measured memory-side traffic is ~5 % of the HBM peak"}
        kernel_name = loss_kernel_name(lib, N, B * ipl, fmt_name)
        common = {"kernel": kernel_name, "traffic": traffic, "traffic_key": roofs_key,
                  "avg_launch_ms": kf["avg_launch_ms"], "avg_launch_ms_raw_events": kf["avg_launch_ms_raw"],
                  "event_pair_ms_subtracted": pair_ms, "launches_timed": launches,
                  "algorithmic_hbm": algorithmic, "hbm_measured": hbm_measured, "source_hash_loaded_library": lib_hash}
        if roofs and roofs.get("valu_instr_per_point_pose"):
            v = valu_roof(roofs, N, len(groups[0]) * B, kf["avg_launch_ms"])
            roofline = {"bound": "valu", "achieved": v["achieved"], "peak": v["peak"], "unit": v["unit"], "frac": v["frac"],
                        "bound_is": "VALU issue: wave64 VALU instructions per launch (rocprofv3 SQ_INSTS_VALU of this launch shape and this "
                                    "library build) / this run's launch time, against 1024 SIMDs x 2400 MHz / 4 cycles per instruction; "
                                    "transcendentals (8-cycle issue), the s_nop slots between dependent packed ops and clocks below 2400 MHz "
                                    "all lower it",
                        "valu": v}
        else:
            roofline = {"bound": "hbm", "achieved": algorithmic["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algorithmic["frac"],
                        "bound_is": "FALLBACK to the algorithmic-bytes figure: no VALU instruction count for this launch shape from this "
                                    "library build (%s); the kernel is VALU-issue bound, see profiles/" % why,
                        "valu": None}
        roofline.update(common)
        # consistency of the line with itself: the loss kernel's time inside one step cannot exceed the step
        kernel_ms_per_step = NUM_ITER * kf["avg_launch_ms"] / ipl
        checks = {"kernel_ms_per_step": kernel_ms_per_step, "ms_per_step": elapsed / K * 1e3,
                  "kernel_time_within_step": bool(kernel_ms_per_step <= elapsed / K * 1e3),
                  "kernel_share_of_step": kernel_ms_per_step / (elapsed / K * 1e3)}
        line = {
            "metric": "candidate-poses/s", "value": value, "unit": "candidate-poses/s", "n_gpus": world, "steps": K,
            "warmup": Wm, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s x %d images per launch chain%s: %d-pt cloud, %dx%d pano, %d candidates x %d GD iterations per image"
                                   % (args.workload, ipl, " (cfg 4's shape on one GPU)" if args.workload == "cfg2" and ipl > 1 else "",
                                      N, W, H, B, NUM_ITER),
                       "images_per_gpu": K, "sharding": "query images round-robin over ranks, RCCL all_gather of results",
                       "mode": "omniloc_batch" if batch_mode else "omniloc", "images_per_launch": ipl,
                       "poses_per_launch": ipl * B, "texels": fmt_name, "warmup_images": "distinct from the timed ones"},
            "passes": len(pass_times), "pass_ms": {"min": min(pass_times) * 1e3, "median": elapsed * 1e3, "max": max(pass_times) * 1e3},
            "ranks_seen": ranks_seen, "devices_visible": ranks.n_dev,
            "per_rank_ms_per_step": {"min": min(per_rank_s) / K * 1e3, "max": max(per_rank_s) / K * 1e3, "ranks": [v / K * 1e3 for v in per_rank_s],
                                     "is": "each rank's own wall time for the median pass's K steps (+ its side of the gather), before "
                                           "the closing barrier; ms_per_step is the barrier-to-barrier time, MAX over ranks"},
            "dist_backend": ranks.backend if ranks.dist is not None else None, "dist_fallback": ranks.fallback,
            "ipc_mode_legacy": ranks.ipc_mode, "ipc_mode_from": ranks.ipc_from,
            "pose_evals_per_s": value * NUM_ITER,
            "median_t_err_m": float(np.median(errs[:, 0])), "median_r_err_deg": float(np.median(errs[:, 1])),
            "single_image": single,
            "roofline": roofline,
            "checks": checks,
        }
    # RCCL prints its version banner through C stdio, which is flushed at exit — after Python's own output.  Every rank
    # flushes it now, before the last barrier, so that rank 0's JSON line is the last thing the job writes to stdout.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if ranks.dist is not None:
        ranks.barrier()
        ranks.dist.destroy_process_group()
        ranks.dist = None
    if rank != 0:
        return
    # ---- rank 0 alone from here on (the job's timed region and its final barrier are behind; the other ranks have left):
    # side measurements and the CPU baseline, so that an N > 1 line carries them too
    if world > 1:
        # the N = 1 value of THIS build on THIS node, for the driver's N = 1 record to be checked against: the same timed pass
        # (rank 0's K images, same launch shape) with no other rank running
        n1_times, n1_elapsed, _ = m.repeated(timed_items, results, K, None, min(args.min_seconds, 1.0), with_gather=False)
        line["n1_value_same_build"] = {"value": B * K / n1_elapsed, "ms_per_step": n1_elapsed / K * 1e3, "passes": len(n1_times),
                                       "is": "rank 0 alone after the job's final barrier: the same K steps at the same launch shape "
                                             "with the other ranks gone; value / n_gpus against this is the per-rank efficiency"}
    if not args.no_also:
        also = {"measured_by": "rank 0 alone, after the timed region%s" % (" and the job's final barrier (reduced set)" if world > 1 else "")}
        t_also = time.perf_counter()
        try:
            if args.workload != "cfg3" and world == 1:
                also["cfg3"] = run_side("cfg3", ranks, args, lib_hash, pair_ms, args.timer_stride, scenes, K=2, ipl=1)
            also["shipped_1_image_per_chain"] = run_side("shipped", ranks, args, lib_hash, pair_ms, args.timer_stride, scenes, K=8, ipl=1)
            also["shipped_8_images_per_chain"] = run_side("shipped", ranks, args, lib_hash, pair_ms, args.timer_stride, scenes, K=8, ipl=8)
            if (1_000_000, 1024, 2048) in scenes and world == 1:
                also["pipeline"] = pipeline_block(scenes[(1_000_000, 1024, 2048)])
                also["pipeline"]["kernel_roofs"] = pipeline_kernel_roofs("pipeline_cfg2")
                also["host_buffers_cfg2"] = host_buffers_block(scenes[(1_000_000, 1024, 2048)])
            if (166_667, 1024, 2048) in scenes:          # the reference's shipped config end to end (stanford_parallel.ini)
                also["pipeline_shipped"] = pipeline_block(scenes[(166_667, 1024, 2048)], num_input=6, num_intermediate=50)
                also["pipeline_shipped"]["kernel_roofs"] = pipeline_kernel_roofs("pipeline_shipped")
                also["pipeline_shipped_8_images"] = pipeline_images_block(scenes[(166_667, 1024, 2048)], ipl=8)
            if (1_000_000, 1024, 2048) in scenes and world == 1:
                also["depth_mask_cfg2"] = depth_mask_block(scenes[(1_000_000, 1024, 2048)], 32)
                also["depth_mask_cfg2"]["kernel_roofs"] = {"default_grid": pipeline_kernel_roofs("depth_mask_cfg2"),
                                                           "every_point": pipeline_kernel_roofs("depth_mask_cfg2_every_point")}
                also["depth_mask_cfg3"] = depth_mask_block(scenes[(1_000_000, 1024, 2048)], 256, n_images=1, panorama_grid=False)
            if args.workload != "cfg5" and world == 1:
                also["cfg5"] = run_side("cfg5", ranks, args, lib_hash, pair_ms, args.timer_stride, scenes, K=2, ipl=2)
        except Exception as exc:                        # the headline must survive a failing side measurement
            also["error"] = "%s: %s" % (type(exc).__name__, exc)
        also["seconds"] = time.perf_counter() - t_also
        line["also"] = also
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(sc.xyz, sc.rgb, img0_host, start0_host[0], start0_host[1], budget_s=args.cpu_baseline_seconds)
    emit(line, args.also_json)


if __name__ == "__main__":
    main()
