#!/usr/bin/env python3
"""One query image, 32 candidates: one launch chain of 32 poses vs two independent chains of 16 poses on two HIP streams
(candidates never interact, so the batch can be cut anywhere).  With two chains in flight one chain's launch ramp,
tail and optimiser epilogue can overlap with the other chain's loss kernel.
   python tools/two_chain.py [cfg2] [splits]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS  # noqa: E402
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
splits = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N, H, W, B, batch = WORKLOADS[wl]
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
cloud = ops.Cloud(X, C)
box = ops.quantile_box(X, 0.05)
t_gt, ypr_gt = synth.gt_pose(0)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
pano = ops.Pano(img)
tr, ro = synth.start_poses(t_gt, ypr_gt, B, 0)
TR, RO = torch.from_numpy(tr).to(dev), torch.from_numpy(ro).to(dev)
kw = dict(lr=0.1, patience=5, factor=0.8, batch_mode=batch)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


one = ops.GradientDescent(cloud, pano, TR, RO, box, **kw)


def run_one():
    one.reset(TR, RO)
    one.run(100)


m = B // splits
parts = [ops.GradientDescent(cloud, pano, TR[i * m:(i + 1) * m].contiguous(), RO[i * m:(i + 1) * m].contiguous(), box, **kw) for i in range(splits)]
streams = [torch.cuda.Stream(device=dev) for _ in range(splits)]


def run_split(concurrent):
    cur = torch.cuda.current_stream()
    for i, gd in enumerate(parts):
        st = streams[i] if concurrent else cur
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            gd.reset(TR[i * m:(i + 1) * m].contiguous(), RO[i * m:(i + 1) * m].contiguous())
            gd.run(100)
    for st in streams:
        cur.wait_stream(st)


a = timed(run_one)
b = timed(lambda: run_split(False))
c = timed(lambda: run_split(True))
print("%s: one chain of %d poses %.2f ms | %d chains of %d poses back to back %.2f ms | on %d streams %.2f ms" % (wl, B, a, splits, m, b, splits, c))
r1 = one.result()[:, 12].min().item()
r2 = min(p.result()[:, 12].min().item() for p in parts)
print("best final loss: one chain %.6f, split %.6f" % (r1, r2))
