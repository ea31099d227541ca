"""One (grid, stride) combination of the depth passes, a few calls, for the profiler's counter passes:
rocprofv3 --pmc ... -- python3 tools/zcount.py depth_h depth_w stride [B]"""
import sys
import torch
sys.path.insert(0, '.')
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth  # noqa: E402
dh, dw, stride = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
N = 1_000_000
xyz, rgb = synth.box_room(N, 0)
cloud = ops.Cloud(torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda())
t_gt, ypr_gt = synth.gt_pose(0)
tr, ro = synth.start_poses(t_gt, ypr_gt, B, 0)
for _ in range(3):
    ops.depth_mask(cloud, torch.from_numpy(tr).cuda(), torch.from_numpy(ro).cuda(), (dh, dw), stride=stride)
torch.cuda.synchronize()
