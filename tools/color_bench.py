#!/usr/bin/env python3
"""Colour preprocessing on one GPU: time per call (HIP events on the current stream) of the sorted-template build,
color_match and color_mod, and algorithmic GB/s.
   python tools/color_bench.py [n_points H W]      (defaults: cfg2 sizes, 1e6 colours, 1024 x 2048 panorama)

Algorithmic bytes (fp32 HWC / (n,3) tensors, every pass counted once):
   template build : 12 n read + 12 n write of the planes, radix sort 4 passes x (4 n read + 4 n write) per channel  = 120 n
   color_match    : histogram pass 12 B/pixel read, apply pass 12 read + 12 written                                 = 36 HW
   color_mod      : (histogram read + apply read + write) over pixels and over colours                              = 36 (HW + n)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 1_000_000
H = int(args[1]) if len(args) > 1 else 1024
W = int(args[2]) if len(args) > 2 else 2048
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(n, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
t_gt, ypr_gt = synth.gt_pose(3)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tmpl = ops.ColorTemplate(C)
rows = [("template build (once per cloud)", timed(lambda: ops.ColorTemplate(C)), 120 * n),
        ("color_match", timed(lambda: ops.color_match(img, tmpl)), 36 * H * W),
        ("color_mod", timed(lambda: ops.color_mod(img, C, 256)), 36 * (H * W + n))]
for name, ms, nbytes in rows:
    print("%-34s %8.3f ms  %7.1f GB/s algorithmic (%.1f MB)" % (name, ms, nbytes / ms / 1e6, nbytes / 1e6))
