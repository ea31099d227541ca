#!/usr/bin/env python3
"""us per GD iteration of ONE refinement at a given shape (graph replay, as the product runs small problems), for sweeps of the
launch-planning knobs (PCL_BLOCKS, PCL_GD_FUSE_BLOCKS, PCL_G ...).   python tools/iter_latency.py [n_points B reps spread]
spread=1: starting poses all over the room (what make_input hands over), 0: near the ground truth (bench.py's)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 166_667
if N < 0:                  # sweep of the spread modes in one process: python tools/iter_latency.py -166667 6
    import subprocess
    for sp in (0, 1, 2, 3):
        subprocess.run([sys.executable, __file__, str(-N)] + sys.argv[2:3] + ["10", str(sp)])
    sys.exit(0)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
spread = int(sys.argv[4]) if len(sys.argv) > 4 else 1
H, W = [int(v) for v in os.environ.get("PCL_TOOL_HW", "1024x2048").split("x")]          # PCL_TOOL_HW=2048x4096: another panorama size
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
t_gt, ypr_gt = synth.gt_pose(0)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
tr, ro = synth.start_poses(t_gt, ypr_gt, B, 0)
if spread:
    rng = np.random.default_rng(0)
    tr = (t_gt[None] + rng.normal(0, 1.0, size=(B, 3))).astype(np.float32)
    ro = (rng.integers(0, 4, size=(B, 3)) * (np.pi / 2)).astype(np.float32)
    tr[0], ro[0] = t_gt + 0.2, ypr_gt + 0.1
    if spread == 2:        # the candidates of a pose group (G = 2: rows 2k, 2k + 1) look the same way from nearby places
        for k in range(0, B - 1, 2):
            ro[k + 1] = ro[k]; tr[k + 1] = tr[k] + np.float32(0.2)
    if spread == 3:        # pairs share the translation, rotations differ
        for k in range(0, B - 1, 2):
            tr[k + 1] = tr[k]
FMT = ops.EXPERIMENT.pano_fmt or ops.refine_texels(N, H, W)      # the product's texel format for this cloud / panorama unless PCL_PANO_FMT forces one
cloud, pano, box = ops.Cloud(X, C), ops.Pano(img, fmt=FMT), ops.quantile_box(X, 0.05)
gd = ops.GradientDescent(cloud, pano, torch.from_numpy(tr).to(dev), torch.from_numpy(ro).to(dev), box, lr=0.1, patience=5, factor=0.8, batch_mode=True)
NIMG = int(os.environ.get("ITER_IMAGES", "1"))           # ITER_IMAGES=8: the B candidates split over 8 query images (own panoramas)
if NIMG > 1:
    panos = []
    for k in range(NIMG):
        tg, yg = synth.gt_pose(100 + k)
        panos.append(ops.Pano(synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(tg), torch.from_numpy(yg)), C, (H, W))), fmt=FMT))
    gd.set_pano_groups(panos)
gd.run_graph(100); torch.cuda.synchronize()
ts = []
for _ in range(reps):
    gd.reset(torch.from_numpy(tr).to(dev), torch.from_numpy(ro).to(dev))
    if NIMG > 1:
        gd.set_pano_groups(panos)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); gd.run_graph(100); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e4)
import ctypes
from piccolo_amd import _lib
nch, G, fz = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
_lib.load().pcl_gd_plan(N, B, ctypes.byref(nch), ctypes.byref(G), ctypes.byref(fz))
print("N %d B %d spread %d | PCL_BLOCKS=%s FUSE=%s | chunks %d G %d fused %d | %.2f us per iteration (median of %d), min %.2f" % (
    N, B, spread, os.environ.get("PCL_BLOCKS"), os.environ.get("PCL_GD_FUSE_BLOCKS"), nch.value, G.value, fz.value, float(np.median(ts)), reps, min(ts)))
