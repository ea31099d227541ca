#!/usr/bin/env python3
"""Depth-mask pass alone (z pass + mark pass) for B candidate poses: time per call.   python tools/dbench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from piccolo_amd import ops, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N, H, W = 1_000_000, 1024, 2048
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
cloud = ops.Cloud(torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev))
t_gt, ypr_gt = synth.gt_pose(0)
tr, ro = synth.start_poses(t_gt, ypr_gt, B, 0)
TR, RO = torch.from_numpy(tr).to(dev), torch.from_numpy(ro).to(dev)
vis = ops.depth_mask(cloud, TR, RO, (H, W))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    vis = ops.depth_mask(cloud, TR, RO, (H, W))
b.record()
torch.cuda.synchronize()
print("variant %s B=%d: %.1f us per depth-mask call, visible fraction %.4f" % (os.environ.get("PCL_ZPASS_VARIANT", "-"), B, a.elapsed_time(b) * 100,
                                                                               float(vis.float().mean())))
