import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from piccolo_amd import ops, synth, utils
from test_hip_harness import STANFORD
from oracle import oracle
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
n, H, W = 60_000, 128, 256
xyz, rgb = synth.box_room(n, 5)
xyz[:3] = [[0.3, -0.2, 0.9], [0.3, -0.2, -1.1], [0.3 + 2e-5, -0.2, 1.0]]
X, C = T(xyz), T(rgb)
t_gt, ypr_gt = synth.gt_pose(5)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, T(t_gt), T(ypr_gt)), C, (H, W)))
trans = np.array([[0.3, -0.2, 0.1]], np.float32)
rot = utils.generate_rot_points(dict(STANFORD), device=X.device).cpu().numpy()
print(rot)
pano = ops.Pano(img, fmt="u8")
def run(sel):
    cloud = ops.Cloud(T(xyz[sel]), T(rgb[sel]), sort=False)
    g = ops.TrimGroups(T(rot))
    tab, cnt = ops.trim_loss_table(cloud, pano, T(trans), g, return_count=True)
    gen = ops.sampling_loss(cloud, pano, T(np.repeat(trans, len(rot), 0)), T(rot), with_grad=False).cpu().numpy()
    ref = oracle.sampling_loss(xyz[sel], rgb[sel], img.cpu().numpy(), np.repeat(trans, len(rot), 0), rot, dtype=np.float64, grad=False)
    return tab.cpu().numpy()[0], cnt.cpu().numpy()[0], gen, ref
tab, cnt, gen, ref = run(slice(0, n))
d = np.abs(tab - gen[:, 0]); print("full: worst", d.max(), "at rot", d.argmax(), rot[d.argmax()], "vs oracle trim", np.abs(tab - ref["loss"]).max(), "gen", np.abs(gen[:, 0] - ref["loss"]).max())
for i in range(3):
    tab, cnt, gen, ref = run(slice(i, i + 1))
    print("point", i, "trim", tab[:8], "\n   gen", gen[:8, 0], "\n   ref", ref["loss"][:8])
    print("  max diff trim-ref", np.nanmax(np.abs(tab - ref["loss"])), "gen-ref", np.nanmax(np.abs(gen[:, 0] - ref["loss"])), "argmax", np.nanargmax(np.abs(tab - ref["loss"])))
tab, cnt, gen, ref = run(slice(3, n))
d = np.abs(tab - gen[:, 0]); print("without specials: worst", d.max(), np.abs(tab - ref["loss"]).max(), np.abs(gen[:, 0] - ref["loss"]).max())
