#!/usr/bin/env python3
"""The second trimming stage alone (ops.hist_trim_scores), ms per call, for A/B of its knobs inside ONE process per setting.
   python tools/hist_stage_bench.py [n_points ncand reps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 166_667
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
H, W = [int(v) for v in os.environ.get("PCL_TOOL_HW", "1024x2048").split("x")]
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(n, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
t_gt, ypr_gt = synth.gt_pose(3)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
rng = np.random.default_rng(1)
tr = torch.from_numpy((t_gt[None] + rng.normal(0, 1.0, size=(K, 3))).astype(np.float32)).to(dev)
ro = torch.from_numpy((rng.integers(0, 4, size=(K, 3)) * (np.pi / 2)).astype(np.float32)).to(dev)
cloud = ops.Cloud(X, C)
s0 = ops.hist_trim_scores(img, cloud, tr, ro, 4, 4); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); s1 = ops.hist_trim_scores(img, cloud, tr, ro, 4, 4); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("n %d K %d | RESOLVE_THREADS=%s BIN_DEDUP=%s | %.3f ms per call (median of %d) | checksum %.9f" % (
    n, K, os.environ.get("PCL_RESOLVE_THREADS"), os.environ.get("PCL_BIN_DEDUP"), float(np.median(ts)), reps, float(s1.double().sum())))
