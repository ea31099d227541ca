#!/usr/bin/env python3
"""the LDS-staged panorama window of the trim launch, MEASURED — in isolation (tools/micro/trim_window.hip: the texture
path only, on the product's real footprints), at the shape users run (166 667 points, 2048 x 1024, the Stanford grid's poses).
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/libtrimwin.so tools/micro/trim_window.hip
    python tools/trim_window_bench.py [n_points] [n_poses]"""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth, utils  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 166_667
P = int(sys.argv[2]) if len(sys.argv) > 2 else 96
H, W = 1024, 2048
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(REPO, "tools", "micro", "libtrimwin.so"))
lib.tw_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
xyz, rgb = synth.box_room(n, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
cloud = ops.Cloud(X, C)
t_gt, ypr_gt = synth.gt_pose(3)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
pano = ops.Pano(img, fmt="u8")
init = dict(max_yaw=2 * np.pi, min_yaw=0, max_pitch=2 * np.pi, min_pitch=0, max_roll=2 * np.pi, min_roll=0, z_prior=None, sample_rate_for_init=None,
            trans_init_mode="quantile", x_max=None, x_min=None, y_max=None, y_min=None, z_max=None, z_min=None, num_split_h=4, num_split_w=4, xy_only=False,
            num_trans=50, yaw_only=False, num_yaw=4, num_pitch=4, num_roll=4, dataset="Stanford2D-3D-S")
rot = utils.generate_rot_points(init, device=dev)
trans = utils.generate_trans_points(X, init, device=dev)
rng = np.random.default_rng(5)
pairs = [(int(rng.integers(len(trans))), j % len(rot)) for j in range(P)]          # every rotation of the grid, random translations
Xp = X[cloud.order]                                                               # the packed (Morton) order the kernels walk
xy = torch.empty(P, n, 2, dtype=torch.int32, device=dev)
for p, (i, j) in enumerate(pairs):
    R = ops.rot_from_ypr(rot[j:j + 1])[0]
    cam = (Xp - trans[i][None, :]) @ R.T
    g = utils.cloud2idx(cam).clamp(-0.99, 0.99)
    ix, iy = ((g[:, 0] + 1) * W - 1) / 2, ((g[:, 1] + 1) * H - 1) / 2
    xy[p, :, 0] = torch.floor(ix).to(torch.int32) + 1                              # + 1: the zero border
    xy[p, :, 1] = torch.floor(iy).to(torch.int32) + 1
assert int(xy[..., 0].min()) >= 0 and int(xy[..., 0].max()) <= W and int(xy[..., 1].min()) >= 0 and int(xy[..., 1].max()) <= H
out = torch.zeros(2, P, dtype=torch.int64, device=dev)
stats = torch.zeros(2, dtype=torch.int64, device=dev)


def run(staged, nchunks, reps=20):
    o = out[1 if staged else 0]
    o.zero_(); stats.zero_()
    args = (staged, pano.data.data_ptr(), W + 2, H + 2, xy.data_ptr(), n, P, nchunks, o.data_ptr(), stats.data_ptr(), None)
    assert lib.tw_run(*args) == 0
    torch.cuda.synchronize()
    first, st = o.clone(), stats.clone()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        lib.tw_run(*args)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3, first, st


print("%d points x %d poses (the Stanford grid's 24 rotations, random grid translations), 2048 x 1024 RGBA8 panorama, 512 points per block step" % (n, P))
for nchunks in (41, 82, 163, 326):
    tg, og, _ = run(0, nchunks)
    ts, os_, st = run(1, nchunks)
    assert torch.equal(og, os_), "the two kernels fetched different texels"
    lds, glb = int(st[0]), int(st[1])
    print("chunks %3d (%4.1f steps per block): gather %7.1f us   staged window %7.1f us   (%.2f x)   footprints served from LDS %.1f %%"
          % (nchunks, n / 512 / nchunks, tg, ts, ts / tg, 100.0 * lds / max(lds + glb, 1)))
