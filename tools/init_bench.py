#!/usr/bin/env python3
"""Wall time of the initialisation stage (make_input: candidate grid -> sampling-loss trim -> histogram trim) and of the
refinement, per query image, on one GPU.   python tools/init_bench.py [n_points H W num_input num_intermediate]
(the reference's stanford_parallel.ini is 1M/6 points, 1024 2048, 6, 50)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import omniloc as po  # noqa: E402
from piccolo_amd import ops, synth, utils  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
W = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
NUM_INPUT = int(sys.argv[4]) if len(sys.argv) > 4 else 32
NUM_INTER = int(sys.argv[5]) if len(sys.argv) > 5 else 64
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(n, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
t_gt, ypr_gt = synth.gt_pose(3)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
init = dict(max_yaw=2 * np.pi, min_yaw=0, max_pitch=2 * np.pi, min_pitch=0, max_roll=2 * np.pi, min_roll=0, z_prior=None,
            sample_rate_for_init=None, trans_init_mode="quantile", x_max=None, x_min=None, y_max=None, y_min=None, z_max=None,
            z_min=None, num_split_h=4, num_split_w=4, xy_only=False, num_trans=50, yaw_only=False, num_yaw=4, num_pitch=4,
            num_roll=4, dataset="Stanford2D-3D-S")


class Cfg:
    lr, num_iter, patience, factor, out_of_room_quantile, num_input = 0.1, 100, 5, 0.8, 0.05, NUM_INPUT


def sync():
    torch.cuda.synchronize()


# candidate grids: once per cloud / config (make_input caches them; torch.quantile's sorts run here and never again)
sync(); t0 = time.perf_counter()
rot = utils.generate_rot_points(init, device=dev)
trans = utils.generate_trans_points(X, init, device=dev)
sync(); t1 = time.perf_counter()
print("candidates %dx%d: grids %.1f ms (once per cloud)" % (len(trans), len(rot), (t1 - t0) * 1e3))
tt, tr = utils.trim_input_loss(img, X, C, trans, rot, NUM_INTER)
it, ir = utils.trim_input_hist_secondary(img, X, C, tt, tr, NUM_INPUT, 4, 4)
it2, ir2 = utils.make_input(img, X, C, NUM_INPUT, init, "loss_histogram", NUM_INTER)
assert torch.equal(it, it2) and torch.equal(ir, ir2)            # the composed call = the two stages (checked once, outside the timed loop)
for rep in range(5):
    sync(); t1 = time.perf_counter()
    tt, tr = utils.trim_input_loss(img, X, C, trans, rot, NUM_INTER)
    sync(); t2 = time.perf_counter()
    it, ir = utils.trim_input_hist_secondary(img, X, C, tt, tr, NUM_INPUT, 4, 4)
    sync(); t3 = time.perf_counter()
    it2, ir2 = utils.make_input(img, X, C, NUM_INPUT, init, "loss_histogram", NUM_INTER)
    sync(); t3b = time.perf_counter()
    res = po.omniloc_batch(img, X, C, it2.clone(), ir2.clone(), Cfg(), {})
    sync(); t4 = time.perf_counter()
    te, re = synth.pose_errors(res[0].numpy(), res[1].numpy(), t_gt, synth.rot_from_ypr_np(ypr_gt))
    print("loss trim %.2f ms | hist trim %.2f ms | make_input (product call, no host sync inside) %.2f ms | GD %.2f ms | t_err %.3f m r_err %.2f deg"
          % ((t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3b - t3) * 1e3, (t4 - t3b) * 1e3, te, re))
