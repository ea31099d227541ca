#!/usr/bin/env python3
"""Times the fused loss+gradient kernel alone (HIP events around each launch) for one workload; tuning knobs come from
the environment (PCL_G, PCL_BLOCKS — read by the library once per process).
   python tools/kbench.py cfg2 [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS  # noqa: E402
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N, H, W, B, batch = WORKLOADS[wl]
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
sort = os.environ.get("KB_SORT", "1") == "1"
fmt = os.environ.get("KB_FMT", "auto")
cloud = ops.Cloud(X, C, sort=sort)
t_gt, ypr_gt = synth.gt_pose(0)
cam = ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt))
img = synth.quantise_like_image_file(ops.make_pano(cam, C, (H, W)))
pano = ops.Pano(img, fmt=fmt)
tr, ro = synth.start_poses(t_gt, ypr_gt, B, 0)
box = ops.quantile_box(X, 0.05)
gd = ops.GradientDescent(cloud, pano, torch.from_numpy(tr), torch.from_numpy(ro), box, lr=0.1, patience=5, factor=0.8, batch_mode=batch,
                         depth_mask=os.environ.get("KB_DEPTH", "0") == "1")
gd.run(5)
timer = ops.KernelTimer(iters)
gd.run(iters, timer=timer)
ms, n = timer.read()
per = ms / n
print("%s G=%s BLOCKS=%s sort=%d fmt=%s : %.1f us/launch  %.1f G point-pose/s  roofline %.3f  loss %.5f" % (
    wl, os.environ.get("PCL_G", "-"), os.environ.get("PCL_BLOCKS", "-"), sort, fmt, per * 1e3,
    N * B / per / 1e6, 24.0 * N * B / (per * 1e-3) / 8e12, float(gd.result()[:, 12].min())))
