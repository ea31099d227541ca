#!/usr/bin/env python3
"""Per-workgroup end times of pcl_tile_resolve_hist_kernel against the length of each tile's list (-DPCL_BLOCK_TRACE build).
   PCL_SO=piccolo_amd/lib/libpiccolo_trace.so python tools/hist_trace.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import _lib, ops, synth, utils  # noqa: E402

n, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000), 1024, 2048
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(n, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
t_gt, ypr_gt = synth.gt_pose(3)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
init = dict(max_yaw=2 * np.pi, min_yaw=0, max_pitch=2 * np.pi, min_pitch=0, max_roll=2 * np.pi, min_roll=0, z_prior=None,
            sample_rate_for_init=None, trans_init_mode="quantile", x_max=None, x_min=None, y_max=None, y_min=None, z_max=None,
            z_min=None, num_split_h=4, num_split_w=4, xy_only=False, num_trans=50, yaw_only=False, num_yaw=4, num_pitch=4,
            num_roll=4, dataset="Stanford2D-3D-S")
rot, trans = utils.generate_rot_points(init, device=dev), utils.generate_trans_points(X, init, device=dev)
t1, r1 = utils.trim_input_loss(img, X, C, trans, rot, NC)
cloud = ops.Cloud(X, C)
raw = ctypes.CDLL(_lib.so_path())
nt = (H // 64) * (W // 64)
buf = torch.zeros(NC * nt * 5, dtype=torch.int64, device=dev)
raw.pcl_debug_set_hist_trace.argtypes = [ctypes.c_void_p]
ops.hist_trim_scores(img, cloud, t1, r1, 4, 4)
assert raw.pcl_debug_set_hist_trace(ctypes.c_void_p(buf.data_ptr())) == 0
torch.cuda.synchronize()
ops.hist_trim_scores(img, cloud, t1, r1, 4, 4)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 5)
print('workgroups that returned at once (empty list):', int((t[:, 0] == 0).sum()))
t = t[t[:, 0] > 0]
end, cnt, beg = t[:, 0].astype(np.float64), t[:, 1], t[:, 2].astype(np.float64)
T0 = beg.min()
e = (end - T0) / 100.0
b = (beg - T0) / 100.0
print("blocks %d, span of end times %.1f us" % (e.size, e.max()))
order = np.argsort(e.ravel())
print("last 10 blocks to finish: end us", np.round(e[order[-10:]], 1), "start us", np.round(b[order[-10:]], 1), "list lengths", cnt[order[-10:]])
d = e - b
d1 = (t[:, 3] - t[:, 2]) / 100.0
d2 = (t[:, 4] - t[:, 3]) / 100.0
d3 = (t[:, 0] - t[:, 4]) / 100.0
for lo, hi in ((0, 100), (100, 1000), (1000, 4000), (4000, 16000), (16000, 40000), (40000, 10 ** 9)):
    m = (cnt >= lo) & (cnt < hi)
    if m.any():
        print("  blocks with %d..%d entries: mean %.1f us max %.1f us (%d blocks) | init + list walk %.1f us, winners -> histograms %.1f us, flush %.1f us"
              % (lo, hi, d[m].mean(), d[m].max(), m.sum(), d1[m].mean(), d2[m].mean(), d3[m].mean()))
print("list lengths: mean %.0f p50 %d p99 %d max %d" % (cnt.mean(), np.percentile(cnt, 50), np.percentile(cnt, 99), cnt.max()))
for lo, hi in ((0, 1000), (1000, 4000), (4000, 16000), (16000, 10 ** 9)):
    m = (cnt >= lo) & (cnt < hi)
    if m.any():
        print("  lists of %d..%d entries: %d blocks, end times %.1f .. %.1f us" % (lo, hi, m.sum(), e[m].min(), e[m].max()))
hist, edges = np.histogram(e.ravel(), bins=10)
print("blocks finished per 10 %% of the span:", hist.tolist())
