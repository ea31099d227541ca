#!/usr/bin/env python3
"""Trim launch with row-major RGBA8 texels vs rows interleaved in pairs (PCL_PANO_U8P) vs vertical pairs (PCL_PANO_U8V): tables must agree
bit for bit; ms per launch.
   python tools/trim_u8p.py [n_points ...]        (PCL_TOOL_HW=2048x4096 for another panorama size)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth, utils
H, W = [int(v) for v in os.environ.get("PCL_TOOL_HW", "1024x2048").split("x")]
dev = torch.device("cuda:0")
for n in [int(a) for a in sys.argv[1:]] or [166_667, 1_000_000]:
    xyz, rgb = synth.box_room(n, 0)
    X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
    t_gt, ypr_gt = synth.gt_pose(3)
    img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
    rot = utils.generate_rot_points(bench.STANFORD_INIT, device=dev)
    trans = utils.generate_trans_points(X, bench.STANFORD_INIT, device=dev)
    groups, cloud = ops.TrimGroups(rot), ops.Cloud(X, C)
    out = {}
    # every layout without (plain (chunk, slot) order) and with the row-sorted work list (ops.TrimOrder, round 6), alternating twice
    for fmt in ("u8", "u8p", "u8v", "u8", "u8p", "u8v"):
        pano = ops.Pano(img, fmt=fmt)
        for tag, order in (("", None), ("+order", ops.TrimOrder(cloud, (pano.H, pano.W, pano.fmt), trans, groups))):
            t = ops.trim_loss_table(cloud, pano, trans, groups, order=order); torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                t0 = time.perf_counter(); t = ops.trim_loss_table(cloud, pano, trans, groups, order=order); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            out.setdefault(fmt + tag, []).append(round(float(np.median(ts)), 3))
            out[fmt + tag + "_table"] = t
    same = lambda a, b: bool(torch.equal(torch.nan_to_num(out[a + "_table"], nan=-1.0), torch.nan_to_num(out[b + "_table"], nan=-1.0)))
    print("n %d: u8 %s +order %s ms | u8p %s +order %s ms | u8v %s +order %s ms | tables equal (NaN-aware): u8p %s, u8v %s, with order: %s" % (
        n, out["u8"], out["u8+order"], out["u8p"], out["u8p+order"], out["u8v"], out["u8v+order"], same("u8", "u8p"), same("u8", "u8v"),
        all(same(f, f + "+order") for f in ("u8", "u8p", "u8v"))), flush=True)
