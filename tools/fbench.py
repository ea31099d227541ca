#!/usr/bin/env python3
"""Forward-only sampling loss over a candidate grid (what trim_input_loss launches): time per call and point-poses/s.
   [PCL_G=1|2|4] python tools/fbench.py [n_poses] [with_grad]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1800
grad = len(sys.argv) > 2 and sys.argv[2] == "1"
N, H, W = 1_000_000, 1024, 2048
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
cloud = ops.Cloud(X, C)
t_gt, ypr_gt = synth.gt_pose(0)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
pano = ops.Pano(img, fmt=os.environ.get("FB_FMT", "u8"))
g = torch.Generator().manual_seed(0)
order = os.environ.get("FB_ORDER", "random")
if order == "random":
    tr = (torch.rand(P, 3, generator=g) - 0.5) * torch.tensor([6.0, 4.0, 2.0])
    ro = torch.rand(P, 3, generator=g) * 6.28
else:                                   # a K x 24 candidate grid like trim_input_loss: translation-major or rotation-major
    K, R = P // 24, 24
    P = K * R
    gx = torch.linspace(-3, 3, 5); gy = torch.linspace(-2, 2, 5); gz = torch.linspace(-1, 1, max(K // 25, 1))
    T = torch.stack(torch.meshgrid(gx, gy, gz, indexing="ij"), -1).reshape(-1, 3)[:K]
    K = T.shape[0]; P = K * R
    Rr = torch.rand(R, 3, generator=g) * 6.28
    if order == "trans":
        tr, ro = T.repeat_interleave(R, 0), Rr.repeat(K, 1)
    else:
        tr, ro = T.repeat(R, 1), Rr.repeat_interleave(K, 0)
TR, RO = tr.to(dev), ro.to(dev)
ops.sampling_loss(cloud, pano, TR, RO, with_grad=grad)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    out = ops.sampling_loss(cloud, pano, TR, RO, with_grad=grad)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print("G=%s poses %d grad=%d: %.3f ms per call, %.1f G point-pose/s, checksum %.6f" % (
    os.environ.get("PCL_G", "-"), P, grad, ms, N * P / ms / 1e6, float(torch.nan_to_num(out[:, 0]).sum())))
