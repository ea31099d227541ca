#!/usr/bin/env python3
"""Trim launch for 8 query images of one room (pcl_trim_loss_images): ms per image.   python tools/trim8.py [n_points]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth, utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 166_667
H, W, I = 1024, 2048, 8
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(n, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
panos = []
for k in range(I):
    t_gt, ypr_gt = synth.gt_pose(10 + k)
    img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
    panos.append(ops.Pano(img, fmt=ops.trim_texels(n, H, W)))
rot = utils.generate_rot_points(bench.STANFORD_INIT, device=dev)
trans = utils.generate_trans_points(X, bench.STANFORD_INIT, device=dev)
groups, cloud = ops.TrimGroups(rot), ops.Cloud(X, C)
one = ops.trim_loss_table(cloud, panos[3], trans, groups)
for tag, order in (("plain order", None), ("row-sorted work list", ops.TrimOrder(cloud, (panos[0].H, panos[0].W, panos[0].fmt), trans, groups))) * 2:
    t = ops.trim_loss_tables(cloud, panos, trans, groups, order=order); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); t = ops.trim_loss_tables(cloud, panos, trans, groups, order=order); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3 / I)
    print("n %d, 8 images, %s, PCL_TRIM_XCD_IMAGES=%s: %.3f ms per image | rows equal the single-image launch: %s" % (
        n, tag, os.environ.get("PCL_TRIM_XCD_IMAGES"), float(np.median(ts)), bool(torch.equal(torch.nan_to_num(t[3], nan=-1.), torch.nan_to_num(one, nan=-1.)))), flush=True)
