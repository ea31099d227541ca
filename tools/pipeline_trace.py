#!/usr/bin/env python3
"""The per-image pipeline (make_input + omniloc_batch) at the reference's shipped shape, for a kernel timeline:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ptrace -- python3 tools/pipeline_trace.py [n_images] [n_points]
then  python3 tools/pipeline_trace.py --report gpurun_out/ptrace   (per image: kernels in order with durations and the gaps between)"""
import csv
import glob
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def report(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"].split("(")[0].split("<")[0] for r in rows]
    # the last image: from the last pcl_pano_pack... of the u8 kind (first kernel of make_input) to the end
    starts = [i for i, n in enumerate(names) if "trim_pose_setup" in n]
    if len(starts) < 2:
        print("no image boundary found"); return
    a, b = starts[-2], starts[-1]
    seg = rows[a:b]
    t0 = int(seg[0]["Start_Timestamp"])
    busy, prev_end, agg = 0, None, {}
    for r, n in zip(seg, names[a:b]):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) if prev_end is not None else 0
        busy += e - s
        k = agg.setdefault(n, [0, 0, 0])
        k[0] += 1; k[1] += e - s; k[2] += max(gap, 0)
        prev_end = e
    wall = int(seg[-1]["End_Timestamp"]) - t0
    print("one image (trim_pose_setup to the next one): %d kernels, wall %.1f us, busy %.1f us, idle %.1f us" % (len(seg), wall / 1e3, busy / 1e3, (wall - busy) / 1e3))
    print("%-60s %6s %10s %12s" % ("kernel", "calls", "busy us", "gap-before us"))
    for n, (c, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
        print("%-60s %6d %10.1f %12.1f" % (n[:60], c, t / 1e3, g / 1e3))
    print("in order (first 60 kernels of the image):")
    prev_end = None
    for r, n in list(zip(seg, names[a:b]))[:60]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("  +%8.1f us  %-52s %8.1f us  (gap %.1f)" % ((s - t0) / 1e3, n[:52], (e - s) / 1e3, ((s - prev_end) / 1e3) if prev_end else 0.0))
        prev_end = e


if len(sys.argv) > 2 and sys.argv[1] == "--report":
    report(sys.argv[2])
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import omniloc as po, utils, synth  # noqa: E402

n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 6
N = int(sys.argv[2]) if len(sys.argv) > 2 else 166_667
NUM_INPUT = int(sys.argv[3]) if len(sys.argv) > 3 else 6
NUM_INTER = int(sys.argv[4]) if len(sys.argv) > 4 else 50
sc = bench.Scene(N, 1024, 2048, torch.device("cuda:0"))


class Cfg:
    lr, num_iter, patience, factor, out_of_room_quantile, num_input = 0.1, 100, 5, 0.8, 0.05, NUM_INPUT


for j in range(n_images):
    e = sc.image(2_000_000 + j, keep_img=True)
    img = e["img"]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr, ro = utils.make_input(img, sc.X, sc.C, NUM_INPUT, bench.STANFORD_INIT, "loss_histogram", NUM_INTER)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    res = po.omniloc_batch(img, sc.X, sc.C, tr, ro, Cfg(), {})
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("image %d: make_input %.3f ms, refine %.3f ms" % (j, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
