"""What north_star's scatter-min depth mask buys on a room that is NOT convex (synth.furnished_room: a pillar, a cabinet, a low
block): loss at the ground-truth pose with and without the mask, share of the points it hides, recall / precision against analytic
occlusion, and the refinement's pose error from the bench's starting poses (32 candidates x 100 iterations) — plain, mask on the
default grid (pcl_depth_default), mask on the panorama's own grid (round 4's definition)."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth
N, H, W, B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 1024, 2048, 32
xyz, rgb = synth.furnished_room(N, 0); X, C = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
cloud = ops.Cloud(X, C); box = ops.quantile_box(X, 0.05)
rows = []
ids = [i for i in range(40) if not synth.inside_furniture(synth.gt_pose(i)[0])][:12]
for image_id in ids:
    t, ypr = synth.gt_pose(image_id)
    img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t), torch.from_numpy(ypr)), C, (H, W)))
    pano = ops.Pano(img)
    tg, rg = torch.from_numpy(t).cuda().reshape(1, 3), torch.from_numpy(ypr).cuda().reshape(1, 3)
    vis = ops.depth_mask(cloud, tg, rg, ops.default_depth_res(N, H, W), stride=ops.default_depth(N, H, W)[3])
    l_plain = float(ops.sampling_loss(cloud, pano, tg, rg, with_grad=False)[0, 0])
    l_mask = float(ops.sampling_loss(cloud, pano, tg, rg, with_grad=False, depth=True)[0, 0])
    hidden = 1.0 - float(vis.float().mean())
    occ = synth.occluded_by_furniture(xyz, t)
    hid = np.empty(N, bool); hid[cloud.order.cpu().numpy()] = ~vis.cpu().numpy()[0].astype(bool)
    rec, prec = float((hid & occ).sum()) / max(occ.sum(), 1), float((hid & occ).sum()) / max(hid.sum(), 1)
    tr, ro = synth.start_poses(t, ypr, B, seed=image_id)
    errs = []
    for kw in (dict(), dict(depth_mask=True), dict(depth_mask=True, depth_res=(H, W), depth_tau=0.02)):
        gd = ops.GradientDescent(cloud, pano, torch.from_numpy(tr).cuda(), torch.from_numpy(ro).cuda(), box, lr=0.1, patience=5, factor=0.8, **kw)
        gd.run(100)
        res = gd.result().cpu().numpy(); k = int(np.argmin(res[:, 12]))
        errs.append(synth.pose_errors(res[k, :3], ops.rot_from_ypr(torch.from_numpy(res[k:k + 1, 3:6]))[0].cpu().numpy(), t, synth.rot_from_ypr_np(ypr)))
    rows.append((l_plain, l_mask, hidden) + tuple(e[0] for e in errs) + tuple(e[1] for e in errs) + (rec, prec, float(occ.mean())))
    print("image %2d  loss at GT plain %.5f masked %.5f  hidden %.3f | t_err mm plain %.1f default-grid %.1f panorama-grid %.1f | r_err deg %.3f %.3f %.3f"
          % ((image_id,) + rows[-1][:3] + tuple(1e3 * v for v in rows[-1][3:6]) + rows[-1][6:9]), flush=True)
r = np.array(rows)
print("default grid %s tau %.3f stride %d: recall %.3f precision %.3f vs analytic occlusion (truly occluded share %.3f), medians" % (
    (ops.default_depth(N, H, W)[:2],) + ops.default_depth(N, H, W)[2:4] + tuple(np.median(np.array(rows)[:, k]) for k in (9, 10, 11))))
print("MEDIAN over %d images: loss at GT plain %.5f masked %.5f hidden %.3f | t_err mm %.1f / %.1f / %.1f | r_err deg %.3f / %.3f / %.3f"
      % (len(r), np.median(r[:, 0]), np.median(r[:, 1]), np.median(r[:, 2]), 1e3 * np.median(r[:, 3]), 1e3 * np.median(r[:, 4]), 1e3 * np.median(r[:, 5]),
         np.median(r[:, 6]), np.median(r[:, 7]), np.median(r[:, 8])))
