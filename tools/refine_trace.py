#!/usr/bin/env python3
"""Where the time of ONE omniloc_batch call goes at cfg-2 size (host wall clock with a device synchronisation after every stage;
the stages are the ones piccolo_amd.omniloc._refine runs).   python tools/refine_trace.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import omniloc as po  # noqa: E402
from piccolo_amd import ops, synth  # noqa: E402

N, H, W, B = 1_000_000, 1024, 2048, 32
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)


class Cfg:
    lr, num_iter, patience, factor, out_of_room_quantile, num_input = 0.1, 100, 5, 0.8, 0.05, B


def sync():
    torch.cuda.synchronize()


for rep in range(4):
    t_gt, ypr_gt = synth.gt_pose(10 + rep)
    img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
    tr, ro = synth.start_poses(t_gt, ypr_gt, B, seed=rep)
    TR, RO = torch.from_numpy(tr).to(dev), torch.from_numpy(ro).to(dev)
    sync(); t0 = time.perf_counter()
    res = po.omniloc_batch(img, X, C, TR.clone(), RO.clone(), Cfg(), {})
    sync(); whole = (time.perf_counter() - t0) * 1e3
    # the same call, stage by stage
    img2 = synth.mark_levels(img.clone())
    sync(); t = [time.perf_counter()]
    cloud = po.packed_cloud(X, C); sync(); t.append(time.perf_counter())
    pano = po.packed_pano(img2); sync(); t.append(time.perf_counter())
    box = po.quantile_box_of(X, 0.05); sync(); t.append(time.perf_counter())
    gd = ops.GradientDescent(cloud, pano, TR, RO, box, lr=0.1, patience=5, factor=0.8, batch_mode=True); sync(); t.append(time.perf_counter())
    gd.run(100); sync(); t.append(time.perf_counter())
    r = gd.result(); k = torch.argmin(r[:, 12]); win = r[k]; R = ops.rot_from_ypr(win[3:6].reshape(1, 3))[0]
    out = torch.cat([win[0:3], R.reshape(-1), win[12:13]]).cpu(); t.append(time.perf_counter())
    d = [(b - a) * 1e3 for a, b in zip(t[:-1], t[1:])]
    print("omniloc_batch %.2f ms | cloud (cached) %.3f | pano pack %.3f | box (cached) %.3f | engine %.3f | run(100) %.3f | result %.3f"
          % (whole, d[0], d[1], d[2], d[3], d[4], d[5]))
