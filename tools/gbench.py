#!/usr/bin/env python3
"""Eager launches vs hipGraph replay of the whole refinement (wall time per image, one GPU).
   python tools/gbench.py cfg1|cfg2"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS  # noqa: E402
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
N, H, W, B, batch = WORKLOADS[wl]
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
cloud = ops.Cloud(X, C)
t_gt, ypr_gt = synth.gt_pose(0)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
pano = ops.Pano(img)
tr, ro = synth.start_poses(t_gt, ypr_gt, B, 0)
TR, RO = torch.from_numpy(tr).to(dev), torch.from_numpy(ro).to(dev)
box = ops.quantile_box(X, 0.05)
gd = ops.GradientDescent(cloud, pano, TR, RO, box, lr=0.1, patience=5, factor=0.8, batch_mode=batch)
for mode in ("eager", "graph"):
    for rep in range(3):
        gd.reset(TR, RO)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            gd.reset(TR, RO)
            if mode == "eager":
                gd.run(100)
            else:
                gd.run_graph(100)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print("%s %s: %.3f ms per refinement (%.1f us per iteration), loss %.5f" % (wl, mode, dt * 1e3, dt * 1e4, float(gd.result()[:, 12].min())))
