#!/usr/bin/env python3
"""Where does one launch of the loss kernel spend its time?  Per-block timeline from the -DPCL_BLOCK_TRACE build of the
library (every block records start / end-of-point-loop / end in 100 MHz ticks and the CU it ran on).

   hipcc ... -DPCL_BLOCK_TRACE -o piccolo_amd/lib/libpiccolo_trace.so piccolo_amd/csrc/*.hip
   PCL_SO=piccolo_amd/lib/libpiccolo_trace.so python tools/block_trace.py [cfg2] [images_per_launch]
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS  # noqa: E402
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import _lib, ops, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
ipl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N, H, W, B, batch = WORKLOADS[wl]
dev = torch.device("cuda:0")
lib = _lib.load()
raw = ctypes.CDLL(_lib.so_path())
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
cloud = ops.Cloud(X, C)
box = ops.quantile_box(X, 0.05)
panos, trs, ros = [], [], []
for k in range(ipl):
    t_gt, ypr_gt = synth.gt_pose(k)
    img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
    panos.append(ops.Pano(img))
    tr, ro = synth.start_poses(t_gt, ypr_gt, B, k)
    trs.append(torch.from_numpy(tr)); ros.append(torch.from_numpy(ro))
gd = ops.GradientDescent(cloud, panos[0], torch.cat(trs), torch.cat(ros), box, lr=0.1, patience=5, factor=0.8, batch_mode=batch)
gd.set_panos([panos[k] for k in range(ipl) for _ in range(B)])
gd.run(60)                                   # converge a bit: the timed launches of a refinement are mostly near the optimum
torch.cuda.synchronize()
nblk_max = 1 << 15
buf = torch.zeros(nblk_max * 4, dtype=torch.int64, device=dev)
raw.pcl_debug_set_block_trace.argtypes = [ctypes.c_void_p]
assert raw.pcl_debug_set_block_trace(ctypes.c_void_p(buf.data_ptr())) == 0
torch.cuda.synchronize()
raw.pcl_debug_stamp.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
stamp = torch.zeros(2, dtype=torch.int64, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
spans = []
for rep in range(5):
    buf.zero_()
    torch.cuda.synchronize()
    raw.pcl_debug_stamp(ctypes.c_void_p(stamp.data_ptr()), stream)          # a one-thread kernel right before the launch ...
    gd.run(1)
    raw.pcl_debug_stamp(ctypes.c_void_p(stamp.data_ptr() + 8), stream)      # ... and after its epilogue
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(-1, 4)
    t = t[t[:, 2] != 0]
    st = stamp.cpu().numpy()
    spans.append((t[:, 2].max() - st[0]) / 100.0)
nb = len(t)
te, hw = t[:, 2], t[:, 3]
T0, T1 = st[0], te.max()
span = (T1 - T0) / 100.0                     # us, from the stamp kernel before the launch (includes ~2 us of launch gap)
xcc = (hw >> 32) & 0xf
hwid = hw & 0xffffffff
cu = (hwid >> 8) & 0xf
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 20 + cu
ids = np.unique(cuid)
print("%s ipl=%d: %d blocks on %d CUs; stamp-to-last-block %.1f us (5 launches: %s); stamp-to-stamp incl. epilogue %.1f us" % (
    wl, ipl, nb, len(ids), span, " ".join("%.1f" % s for s in spans), (st[1] - st[0]) / 100.0))
per_cu = np.array([(cuid == c).sum() for c in ids])
print("blocks per CU: min %d mean %.2f max %d" % (per_cu.min(), per_cu.mean(), per_cu.max()))
# per CU: sorted end times; with S concurrent slots block k starts when block k-S ended (the first S at the launch)
for S in (4,):
    durs, firsts = [], []
    for c in ids:
        e = np.sort(te[cuid == c])
        starts = np.concatenate([np.full(min(S, len(e)), e[0] - 0), e[:-S]]) if len(e) > S else np.full(len(e), e[0])
        d = (e[S:] - e[:-S]) / 100.0 if len(e) > S else np.array([])
        durs.append(d)
        firsts.append((e[:S] - T0) / 100.0)
    durs, firsts = np.concatenate(durs), np.concatenate(firsts)
    print("assuming %d slots per CU: first-round blocks end %.1f .. %.1f us after the stamp (mean %.1f); later blocks take mean %.2f us (p5 %.2f p95 %.2f)" % (
        S, firsts.min(), firsts.max(), firsts.mean(), durs.mean(), np.percentile(durs, 5), np.percentile(durs, 95)))
last = np.array([te[cuid == c].max() for c in ids])
print("tail: a CU's last block ends %.2f .. %.2f us before the launch's last block (mean %.2f us = %.1f %% of the span)" % (
    (T1 - last.max()) / 100.0, (T1 - last.min()) / 100.0, (T1 - last.mean()) / 100.0, 100 * (T1 - last.mean()) / (T1 - T0)))
# throughput profile: blocks finished per 5 % of the span
edges = np.linspace(T0, T1, 21)
print("blocks finished per 5 %% time bin:", [int(((te > edges[i]) & (te <= edges[i + 1])).sum()) for i in range(20)])
# dispatch order: when do blocks of each blockIdx decile finish
order = np.arange(nb)
for q in range(10):
    sel = (order >= q * nb // 10) & (order < (q + 1) * nb // 10)
    print("  blockIdx decile %d: ends %.1f .. %.1f us" % (q, (te[sel].min() - T0) / 100.0, (te[sel].max() - T0) / 100.0))
# per XCD: when does each XCD finish
for x in np.unique(xcc):
    print("  XCD %d: %d blocks, last end %.1f us" % (x, (xcc == x).sum(), (te[xcc == x].max() - T0) / 100.0))
