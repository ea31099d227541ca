#!/usr/bin/env python3
"""Refinement of I query images of one cloud: one launch chain per image vs all images in one chain (omniloc_batch_images).
   python tools/chain_bench.py [n_points B I]      (defaults: 1M points, 6 candidates, 8 images — the shipped configs' shape)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import omniloc as po  # noqa: E402
from piccolo_amd import ops, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
I = int(sys.argv[3]) if len(sys.argv) > 3 else 8
H, W = 1024, 2048
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
imgs, trs, ros = [], [], []
for k in range(I):
    t_gt, ypr_gt = synth.gt_pose(k)
    imgs.append(synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W))))
    tr, ro = synth.start_poses(t_gt, ypr_gt, B, k)
    trs.append(torch.from_numpy(tr).to(dev)); ros.append(torch.from_numpy(ro).to(dev))


class Cfg:
    lr, num_iter, patience, factor, out_of_room_quantile, num_input = 0.1, 100, 5, 0.8, 0.05, B


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


one = timed(lambda: [po.omniloc_all(imgs[k], X, C, trs[k].clone(), ros[k].clone(), Cfg()) for k in range(I)])
chain = timed(lambda: po.omniloc_batch_images(imgs, X, C, [t.clone() for t in trs], [r.clone() for r in ros], Cfg(), batch_mode=False))
print("%d points, %d candidates, %d images: one chain per image %.2f ms per image | one chain for all %.2f ms per image (%.1fx)" % (
    N, B, I, one / I, chain / I, one / chain))
