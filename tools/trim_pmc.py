#!/usr/bin/env python3
"""The trim launch alone for the counter passes (profiles/collect.sh --script): `reps` launches of the 1800-pose Stanford grid in the layout
ops.trim_texels picks, in plain (chunk, slot) order or with the row-sorted work list.   python tools/trim_pmc.py [n_points] [plain|order] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import _knobs  # noqa: F401
from piccolo_amd import ops, synth, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "order"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
H, W = 1024, 2048
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(n, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
t_gt, ypr_gt = synth.gt_pose(3)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
rot = utils.generate_rot_points(bench.STANFORD_INIT, device=dev)
trans = utils.generate_trans_points(X, bench.STANFORD_INIT, device=dev)
groups, cloud = ops.TrimGroups(rot), ops.Cloud(X, C)
pano = ops.Pano(img, fmt=ops.trim_texels(n, H, W))
order = ops.TrimOrder(cloud, (pano.H, pano.W, pano.fmt), trans, groups) if mode == "order" else None
for _ in range(reps):
    t = ops.trim_loss_table(cloud, pano, trans, groups, order=order)
torch.cuda.synchronize()
print("n %d, %s, texels %s: %d launches, table checksum %.6f" % (n, mode, pano.fmt, reps, float(torch.nan_to_num(t).double().sum())))
