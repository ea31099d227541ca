"""A few depth-masked refinements at cfg-2 / cfg-3 size for the profiler
(rocprofv3 --kernel-trace --stats -- python3 tools/dgd_profile.py [B] [iterations] [depth_stride, 0 = default])."""
import sys

import torch

sys.path.insert(0, '.')
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 100
STRIDE = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N, H, W = 1_000_000, 1024, 2048
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
t, ypr = synth.gt_pose(0)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t), torch.from_numpy(ypr)), C, (H, W)))
tr, ro = synth.start_poses(t, ypr, B, seed=0)
cloud, pano = ops.Cloud(X, C), ops.Pano(img)
box = ops.quantile_box(X, 0.05)
for rep in range(3):
    gd = ops.GradientDescent(cloud, pano, torch.from_numpy(tr).cuda(), torch.from_numpy(ro).cuda(), box, lr=0.1, patience=5, factor=0.8, depth_mask=True,
                             depth_stride=STRIDE or None)
    gd.run(ITERS)
    torch.cuda.synchronize()
print("done")
