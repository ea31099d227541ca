#!/usr/bin/env python3
"""Loss trim (utils.trim_input_loss's table) at cfg-2 size: the yaw-shared kernel (pcl_trim_loss) against the generic forward-only
kernel over the same K x R pairs, for the reference's two grid shapes.   python tools/trim_bench.py [n_points]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth, utils  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
fmts = sys.argv[2].split(",") if len(sys.argv) > 2 else ["u8"]
H, W = 1024, 2048
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(n, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
t_gt, ypr_gt = synth.gt_pose(3)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
base = dict(max_yaw=2 * np.pi, min_yaw=0, max_pitch=2 * np.pi, min_pitch=0, max_roll=2 * np.pi, min_roll=0, z_prior=None,
            sample_rate_for_init=None, trans_init_mode="quantile", x_max=None, x_min=None, y_max=None, y_min=None, z_max=None,
            z_min=None, num_split_h=4, num_split_w=4)
grids = {"stanford (75 x 24)": dict(base, xy_only=False, num_trans=50, yaw_only=False, num_yaw=4, num_pitch=4, num_roll=4, dataset="Stanford2D-3D-S"),
         "omniscenes (yaw only, 8)": dict(base, xy_only=True, num_trans=150, yaw_only=True, num_yaw=8, num_pitch=8, num_roll=8, dataset="OmniScenes", z_prior=0.0)}
cloud = ops.Cloud(X, C)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


for fmt in fmts:
    pano = ops.Pano(img, fmt=fmt)
    for name, init in grids.items():
        rot = utils.generate_rot_points(init, device=dev)
        trans = utils.generate_trans_points(X, init, device=dev)
        K, R = len(trans), len(rot)
        groups = ops.TrimGroups(rot)
        tt, rr = trans.repeat(R, 1), rot.repeat_interleave(K, dim=0)          # rotation-major, as the generic path launched it
        shared = timed(lambda: ops.trim_loss_table(cloud, pano, trans, groups))
        gpano = pano if fmt != "u8p" else ops.Pano(img, fmt="u8")           # (the generic kernel reads row-major texels only)
        generic = timed(lambda: ops.sampling_loss(cloud, gpano, tt, rr, with_grad=False))
        a = ops.trim_loss_table(cloud, pano, trans, groups).reshape(-1)
        b = ops.sampling_loss(cloud, gpano, tt, rr, with_grad=False)[:, 0].reshape(R, K).t().reshape(-1)
        print("%s %s: %d x %d pairs in %d groups | yaw-shared %.2f ms | generic %.2f ms | %.2fx | max rel diff %.1e | same top-64: %s" % (
            fmt, name, K, R, groups.ngroups, shared, generic, generic / shared, float((a - b).abs().max() / b.abs().max()),
            bool(set(torch.topk(a, 64, largest=False).indices.tolist()) == set(torch.topk(b, 64, largest=False).indices.tolist()))))
