"""Does the depth mask change the refinement at 256 candidates as it does at 32?  Winner pose and a digest of all rows for plain /
mask every iteration / mask every 4th, at both sizes."""
import sys, hashlib
import numpy as np, torch
sys.path.insert(0, '.')
from piccolo_amd import ops, synth
N, H, W = 1_000_000, 1024, 2048
xyz, rgb = synth.box_room(N, 0); X, C = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
cloud = ops.Cloud(X, C); box = ops.quantile_box(X, 0.05)
image_id = 4_000_001
t, ypr = synth.gt_pose(image_id)
img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t), torch.from_numpy(ypr)), C, (H, W)))
pano = ops.Pano(img)
for B in (32, 256):
    tr, ro = synth.start_poses(t, ypr, B, seed=image_id)
    for name, kw in (("plain", dict()), ("every 1", dict(depth_mask=True)), ("every 4", dict(depth_mask=True, depth_every=4))):
        gd = ops.GradientDescent(cloud, pano, torch.from_numpy(tr).cuda(), torch.from_numpy(ro).cuda(), box, lr=0.1, patience=5, factor=0.8, **kw)
        gd.run(100)
        res = gd.result().cpu().numpy()
        k = int(np.argmin(res[:, 12]))
        print(B, name, "winner", k, res[k, :6], "loss", res[k, 12], "digest", hashlib.sha1(res.tobytes()).hexdigest()[:12],
              "masks", float(gd.depth_refresh_counts().float().mean()))
