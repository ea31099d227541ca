#!/usr/bin/env python3
"""A/B two builds of the library (PCL_SO=...) on the GD loop: run one refinement, store result + loss history, or compare
two stored runs bit for bit.   PCL_SO=a.so python tools/ab_gd.py run a [cfg1|cfg2] ; ... run b ; python tools/ab_gd.py cmp a b"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "cmp":
    a, b = (torch.load("gpurun_out/ab_%s.pt" % t) for t in sys.argv[2:4])
    ok = True
    for k in a:
        same = torch.equal(a[k].view(torch.int32), b[k].view(torch.int32))
        ok = ok and same
        print(k, "bit-identical" if same else "DIFFERENT max|d| = %g" % float((a[k] - b[k]).abs().max()))
    sys.exit(0 if ok else 1)

from bench import WORKLOADS  # noqa: E402
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth  # noqa: E402

tag, wl = sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "cfg2")
N, H, W, B, batch = WORKLOADS[wl]
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
cloud = ops.Cloud(X, C)
t_gt, ypr_gt = synth.gt_pose(0)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
pano = ops.Pano(img)
out = {}
for mode in (True, False):
    tr, ro = synth.start_poses(t_gt, ypr_gt, max(B, 4), 0)
    tr[0, 0] = 4.4                                          # one start outside the clamp box
    gd = ops.GradientDescent(cloud, pano, torch.from_numpy(tr).to(dev), torch.from_numpy(ro).to(dev), ops.quantile_box(X, 0.05),
                             lr=0.1, patience=5, factor=0.8, batch_mode=mode)
    hist = gd.run(100, history=True)
    out["result_batch%d" % mode], out["history_batch%d" % mode] = gd.result().cpu(), hist.cpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        gd.reset(torch.from_numpy(tr).to(dev), torch.from_numpy(ro).to(dev))
        gd.run(100)
    torch.cuda.synchronize()
    print("%s %s batch=%d: %.3f ms per refinement" % (tag, wl, mode, (time.perf_counter() - t0) / 5 * 1e3))
os.makedirs("gpurun_out", exist_ok=True)
torch.save(out, "gpurun_out/ab_%s.pt" % tag)
