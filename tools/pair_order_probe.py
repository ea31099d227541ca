"""Do the survivors make_input hands to the refinement pair up?  For 8 query images at the shipped shape: the 6 survivors' rotations /
translations, and the time of the 8-image launch chain (48 candidates) with the candidates of each image (a) as handed over,
(b) ordered so that the two poses of a block look the same way where possible (greedy nearest pairs), (c) adversarially mixed."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth, utils
from piccolo_amd import omniloc as po
N, H, W, B, I = 166_667, 1024, 2048, 6, 8
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
imgs = []
for j in range(I):
    t, ypr = synth.gt_pose(3_000_000 + j)
    imgs.append(synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t), torch.from_numpy(ypr)), C, (H, W))))
starts = utils.make_input_images(imgs, X, C, B, bench.STANFORD_INIT, "loss_histogram", 50)


def pose_dist(t, r):
    """(B, B) distance: rotation angle between the two poses (rad) + translation distance / 2 m"""
    R = ops.rot_from_ypr(r).cpu().numpy().astype(np.float64)
    t = t.cpu().numpy().astype(np.float64)
    d = np.zeros((len(t), len(t)))
    for a in range(len(t)):
        for b in range(len(t)):
            c = (np.trace(R[a].T @ R[b]) - 1) / 2
            d[a, b] = np.arccos(np.clip(c, -1, 1)) + np.linalg.norm(t[a] - t[b]) / 2.0
    return d


def greedy_pairs(d):
    left, order = list(range(len(d))), []
    while left:
        a = left.pop(0)
        if not left:
            order.append(a); break
        b = min(left, key=lambda x: d[a, x])
        left.remove(b)
        order += [a, b]
    return order


def worst_pairs(d):
    left, order = list(range(len(d))), []
    while left:
        a = left.pop(0)
        if not left:
            order.append(a); break
        b = max(left, key=lambda x: d[a, x])
        left.remove(b)
        order += [a, b]
    return order


orders = {"as handed over": [], "nearest pairs": [], "farthest pairs": []}
for j, (t, r) in enumerate(starts):
    d = pose_dist(t, r)
    o = greedy_pairs(d)
    orders["as handed over"].append(list(range(B))); orders["nearest pairs"].append(o); orders["farthest pairs"].append(worst_pairs(d))
    pd = lambda oo: [round(float(d[oo[k], oo[k + 1]]), 2) for k in range(0, B, 2)]
    print("image %d: rot (deg) %s | pair distances as handed over %s, nearest %s" % (
        j, np.rad2deg(r.cpu().numpy()).round(0).astype(int).tolist(), pd(list(range(B))), pd(o)))

cloud = po.packed_cloud(X, C)
panos = [po.packed_pano(im, n_points=N) for im in imgs]
box = po.quantile_box_of(X, 0.05)
for name, oo in orders.items():
    tr = torch.cat([starts[j][0][oo[j]] for j in range(I)]).contiguous()
    ro = torch.cat([starts[j][1][oo[j]] for j in range(I)]).contiguous()
    gd = ops.GradientDescent(cloud, panos[0], tr, ro, box, lr=0.1, patience=5, factor=0.8, batch_mode=True)
    ts = []
    for rep in range(5):
        gd.reset(tr, ro); gd.set_pano_groups(panos)
        torch.cuda.synchronize(); t0 = time.perf_counter(); gd.run(100); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e4)
    res = gd.result().cpu().numpy()
    print("%-16s %.1f us per iteration (48 candidates of 8 images), digest of sorted losses %s" % (name, float(np.median(ts)), np.sort(res[:, 12])[:4]))
# one image at a time (fused launch, 6 candidates)
for name, oo in orders.items():
    ts = []
    for j in range(I):
        tr, ro = starts[j][0][oo[j]].contiguous(), starts[j][1][oo[j]].contiguous()
        gd = ops.GradientDescent(cloud, panos[j], tr, ro, box, lr=0.1, patience=5, factor=0.8, batch_mode=True)
        for rep in range(3):
            gd.reset(tr, ro)
            torch.cuda.synchronize(); t0 = time.perf_counter(); gd.run(100); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e4
        ts.append(dt)
    print("%-16s one image per chain: %.2f us per iteration (median over the 8 images; per image %s)" % (name, float(np.median(ts)), np.round(ts, 1).tolist()))
