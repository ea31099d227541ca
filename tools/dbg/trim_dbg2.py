import sys, os
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import numpy as np, torch
from piccolo_amd import ops, synth, utils
from test_hip_harness import STANFORD
from oracle import oracle
n, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(n, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
t_gt, ypr_gt = synth.gt_pose(3)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
init = dict(STANFORD)
rot = utils.generate_rot_points(init, device=dev)
trans = utils.generate_trans_points(X, init, device=dev)
K, R = len(trans), len(rot)
cloud = ops.Cloud(X, C); pano = ops.Pano(img, fmt="u8")
g = ops.TrimGroups(rot)
a, ca = ops.trim_loss_table(cloud, pano, trans, g, return_count=True)
tt, rr = trans.repeat_interleave(R, 0), rot.repeat(K, 1)
gen = ops.sampling_loss(cloud, pano, tt, rr, with_grad=False)
a, ca, b, cb = a.reshape(-1).cpu().numpy(), ca.reshape(-1).cpu().numpy(), gen[:, 0].cpu().numpy(), gen[:, 1].cpu().numpy()
d = np.abs(a - b) / np.abs(b).max()
idx = np.argsort(-d)[:8]
for i in idx:
    print("pair", i, "k", i // R, "r", i % R, rot[i % R].cpu().numpy(), trans[i // R].cpu().numpy(), "trim", a[i], ca[i], "gen", b[i], cb[i])
ref = oracle.sampling_loss(xyz, rgb, img.cpu().numpy(), tt.cpu().numpy()[idx], rr.cpu().numpy()[idx], dtype=np.float64, grad=False)
print("oracle", ref["loss"], ref["count"])
