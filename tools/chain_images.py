import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth
N, H, W, B = 166_667, 1024, 2048, 6
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
cloud, box = ops.Cloud(X, C), ops.quantile_box(X, 0.05)
panos, TR, RO = [], [], []
for j in range(32):
    t, ypr = synth.gt_pose(100 + j)
    img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t), torch.from_numpy(ypr)), C, (H, W)))
    panos.append(ops.Pano(img, fmt=ops.refine_texels(N, H, W)))
    tr, ro = synth.start_poses(t, ypr, B, seed=j)
    TR.append(torch.from_numpy(tr).to(dev)); RO.append(torch.from_numpy(ro).to(dev))
for I in (1, 2, 4, 8, 16, 24, 32):
    tr = torch.cat(TR[:I]).contiguous(); ro = torch.cat(RO[:I]).contiguous()
    gd = ops.GradientDescent(cloud, panos[0], tr, ro, box, lr=0.1, patience=5, factor=0.8, batch_mode=True)
    ts = []
    for rep in range(6):
        gd.reset(tr, ro); gd.set_pano_groups(panos[:I])
        torch.cuda.synchronize(); t0 = time.perf_counter(); gd.run(100); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e4)
    m = float(np.median(ts[1:]))
    print("%2d images per chain (%3d poses): %.1f us per iteration = %.2f us per image-iteration" % (I, I * B, m, m / I), flush=True)
