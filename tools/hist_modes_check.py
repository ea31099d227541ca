"""Scores of the second trimming stage at cfg-2 size (1M points, 64 candidates) — one process per setting, scores dumped and compared by
the caller:  PCL_BIN_EXACT=0|1 / PCL_HIST_SPLAT=1 python tools/hist_modes_check.py out.npy"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth
n, K, H, W = 1_000_000, 64, 1024, 2048
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(n, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
t_gt, ypr_gt = synth.gt_pose(3)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
rng = np.random.default_rng(1)
tr = torch.from_numpy((t_gt[None] + rng.normal(0, 1.0, size=(K, 3))).astype(np.float32)).to(dev)
ro = torch.from_numpy((rng.integers(0, 4, size=(K, 3)) * (np.pi / 2)).astype(np.float32)).to(dev)
cloud = ops.Cloud(X, C)
s, inter, nproj, nimg = ops.hist_trim_scores(img, cloud, tr, ro, 4, 4, return_parts=True, splat=os.environ.get("PCL_HIST_SPLAT") == "1")
np.save(sys.argv[1], np.concatenate([s.cpu().numpy().ravel(), nproj.cpu().numpy().ravel().astype(np.float64)]))
print(os.environ.get("PCL_BIN_EXACT"), os.environ.get("PCL_HIST_SPLAT"), float(s.double().sum()), int(nproj.sum()))
