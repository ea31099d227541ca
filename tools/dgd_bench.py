"""Depth-masked GD iteration (round 5 design: fill + z pass + loss launch with the in-kernel lookup) against the plain iteration,
whole 100-iteration refinements timed from the host.   python tools/dgd_bench.py [B] [n_points]
Env knobs of the z pass (experiments): PCL_ZFORM=1|2|3 (coarse-tile cache | LDS window | untiled scatter), PCL_ZSECOND=0 (no second window)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
import os
H, W = [int(v) for v in os.environ.get("PCL_TOOL_HW", "1024x2048").split("x")]
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
t, ypr = synth.gt_pose(0)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t), torch.from_numpy(ypr)), C, (H, W)))
tr, ro = synth.start_poses(t, ypr, B, seed=0)
cloud, pano = ops.Cloud(X, C), ops.Pano(img)
box = ops.quantile_box(X, 0.05)
dh, dw, dtau, dst = ops.default_depth(N, H, W)
h1, w1, t1, _ = ops.default_depth(N, H, W, stride=1)
rows = [("plain", dict(depth_mask=False)), ("mask, default: grid %dx%d tau %.3f stride %d" % (dw, dh, dtau, dst), dict(depth_mask=True)),
        ("mask, every point: grid %dx%d tau %.3f stride 1" % (w1, h1, t1), dict(depth_mask=True, depth_stride=1))]
if B <= 64:
    rows.append(("mask, panorama grid %dx%d tau 0.02 (round 4's)" % (W, H), dict(depth_mask=True, depth_res=(H, W), depth_tau=0.02)))
base = None
for name, kw in rows:
    ts = []
    for rep in range(4):
        gd = ops.GradientDescent(cloud, pano, torch.from_numpy(tr).cuda(), torch.from_numpy(ro).cuda(), box, lr=0.1, patience=5, factor=0.8, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter(); gd.run(100); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e4)
    res = gd.result().cpu().numpy(); k = int(np.argmin(res[:, 12]))
    te, re = synth.pose_errors(res[k, :3], ops.rot_from_ypr(torch.from_numpy(res[k:k + 1, 3:6]))[0].cpu().numpy(), t, synth.rot_from_ypr_np(ypr))
    us = float(np.median(ts[1:]))
    base = base or us
    print("%-52s %8.1f us per iteration  (%.2f x plain) | t_err %.4f m r_err %.3f deg" % (name, us, us / base, te, re), flush=True)
