#!/usr/bin/env python3
"""The depth passes alone for B candidate poses at cfg-2 size: pcl_depth_mask (pose setup + fill + z pass + mark) per call for a
list of (grid, occluder stride) combinations; the mark pass is the same in all of them (every point is tested), so differences are
the z pass's.   python tools/dbench.py [B]      (PCL_ZFORM / PCL_ZSECOND: see csrc/pcl_depth.hip)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N, H, W = 1_000_000, 1024, 2048
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
cloud = ops.Cloud(torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev))
t_gt, ypr_gt = synth.gt_pose(0)
tr, ro = synth.start_poses(t_gt, ypr_gt, B, 0)
TR, RO = torch.from_numpy(tr).to(dev), torch.from_numpy(ro).to(dev)
for res, stride in (((200, 400), 1), ((200, 400), 2), ((200, 400), 4), ((200, 400), 64), ((144, 288), 1), ((144, 288), 2), ((96, 192), 4), ((H, W), 1)):
    vis = ops.depth_mask(cloud, TR, RO, res, stride=stride)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        vis = ops.depth_mask(cloud, TR, RO, res, stride=stride)
    b.record()
    torch.cuda.synchronize()
    print("grid %4dx%-4d stride %2d B=%d form=%s: %.1f us per depth-mask call (fill + z pass + mark), visible fraction %.4f"
          % (res[1], res[0], stride, B, os.environ.get("PCL_ZFORM", "auto"), a.elapsed_time(b) * 100, float(vis.float().mean())))
