#!/usr/bin/env python3
"""Achieved parity of the fused loss+gradient kernel, as a table: HIP (through the C ABI) vs the fp64 answer, next to the
fp32-vs-fp64 gap of the reference itself (goldens) or of the fp32 oracle (synthetic scenes) — the yardstick for what any
fp32 evaluation can deliver.  rel(a, b) = max|a - b| / max|b| over all poses and components of a block.

   python tools/grad_error.py [--full]         (--full adds the BASELINE cfg-2 / cfg-3 sizes; GPU box only)
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import oracle as orc  # noqa: E402
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ok = ~np.isnan(b)
    return float(np.abs(a[ok] - b[ok]).max() / max(np.abs(b[ok]).max(), 1e-30))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def hip(xyz, rgb, img, trans, rot, fmt="auto", sort=True):
    cloud, pano = ops.Cloud(T(xyz), T(rgb), sort=sort), ops.Pano(T(img), fmt=fmt)
    return ops.sampling_loss(cloud, pano, T(trans), T(rot), with_grad=True).cpu().numpy()


def row(name, out, l64, gt64, gr64, l32, gt32, gr32, cnt=None):
    dc = "" if cnt is None else " dcount %d" % int(np.abs(out[:, 1] - cnt).max())
    print("| %-46s | %.1e | %.1e | %.1e | %.1e | %.1e | %.1e |%s" % (
        name, rel(out[:, 0], l64), rel(out[:, 2:5], gt64), rel(out[:, 5:8], gr64), rel(l32, l64), rel(gt32, gt64), rel(gr32, gr64), dc))


def main():
    full = "--full" in sys.argv
    orc.build()
    print("| case | HIP loss | HIP grad_t | HIP grad_ypr | fp32 ref loss | fp32 ref grad_t | fp32 ref grad_ypr |")
    print("|---|---|---|---|---|---|---|")
    g = np.load(os.path.join(GOLDEN, "g3_sampling_loss.npz"))
    for fmt in ("auto", "u8", "f32"):
        for sort in (False, True):
            out = hip(g["xyz"], g["rgb"], g["img"], g["trans"], g["rot"], fmt=fmt, sort=sort)
            row("G3 (reference autograd) fmt=%s sort=%d" % (fmt, sort), out, g["loss_f64"], g["grad_t_f64"], g["grad_ypr_f64"],
                g["loss_f32"], g["grad_t_f32"], g["grad_ypr_f32"])
    g4 = np.load(os.path.join(GOLDEN, "g4_batch_sampling_loss.npz"))
    out = hip(g["xyz"], g["rgb"], g["img"], g4["trans"], g4["rot"])
    row("G4 (reference autograd, batched)", out, g4["loss_list_f64"], g4["grad_t_f64"], g4["grad_ypr_f64"],
        g4["loss_list_f32"], g4["grad_t_f32"], g4["grad_ypr_f32"])
    cases = [(255, 64, 128, 3), (10_000, 128, 256, 5), (50_021, 101, 203, 5), (100_000, 256, 512, 1), (200_003, 256, 512, 8)]
    if full:
        cases += [(1_000_000, 1024, 2048, 32), (1_000_000, 1024, 2048, 256)]
    for n, H, W, B in cases:
        xyz, rgb = synth.box_room(n, seed=n)
        t_gt, ypr_gt = synth.gt_pose(n % 97)
        if n >= 1_000_000:
            img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(T(xyz), T(t_gt), T(ypr_gt)), T(rgb), (H, W))).cpu().numpy()
        else:
            img = orc.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W)).astype(np.float32) / 255
        trans, rot = synth.start_poses(t_gt, ypr_gt, B, seed=n)
        r64 = orc.sampling_loss(xyz, rgb, img, trans, rot, dtype=np.float64)
        r32 = orc.sampling_loss(xyz, rgb, img, trans, rot, dtype=np.float32)
        for fmt in (("auto", "f32") if n < 1_000_000 else ("auto",)):
            out = hip(xyz, rgb, img, trans, rot, fmt=fmt)
            row("oracle n=%d %dx%d B=%d fmt=%s" % (n, W, H, B, fmt), out, r64["loss"], r64["grad_t"], r64["grad_ypr"],
                r32["loss"], r32["grad_t"], r32["grad_ypr"], cnt=r64["count"])
        # per-pose worst case (each pose's error relative to that pose's own largest component)
        gh = np.concatenate([out[:, 2:5], out[:, 5:8]], 1)
        g64 = np.concatenate([r64["grad_t"], r64["grad_ypr"]], 1)
        per = np.abs(gh - g64).max(1) / np.abs(g64).max(1)
        print("|   per-pose rel: median %.1e max %.1e | | | | | | |" % (np.median(per), per.max()))


if __name__ == "__main__":
    main()
