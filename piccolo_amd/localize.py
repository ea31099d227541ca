"""Thin counterpart of the reference's dataset harness (localize.py) for the part that is on the hot path:
the per-image body — starting poses -> refinement -> pose error (localize.py:208-258) — plus a synthetic-scene
driver that needs no dataset.  Dataset IO, TensorBoard and result images stay with the
reference's own localize.py, which runs unchanged on top of piccolo_amd's omniloc/utils.
"""
import csv
import os
import time

import numpy as np
import torch

from . import dist as pdist
from . import ops, synth
from .color_utils import color_match, color_mod
from .omniloc import omniloc_all, omniloc_batch


def preprocess_colors(img, rgb, cfg):
    """The colour modulation step of the per-image body (localize.py:173-179 for Stanford2D3DS, :395-409 for
    OmniScenes): cfg.match_color -> color_match(img, rgb); cfg.sharpen_color -> color_mod(img, rgb, cfg.num_bins).
    As in the reference, both start from the ORIGINAL image (when both are set, color_mod's result is the one kept) and
    the result is re-quantised to uint8 levels (`(255 * new_img).astype(np.uint8)`, :404, :410).  -> (img, rgb)."""
    new_img = img
    if getattr(cfg, "match_color", False):
        new_img = color_match(img, rgb)
    if getattr(cfg, "sharpen_color", False):
        new_img, rgb = color_mod(img, rgb, int(getattr(cfg, "num_bins", 256)))
    if new_img is not img:
        new_img = synth.quantise_like_image_file(new_img * 255.0)
    return new_img, rgb


def refine_image(img, xyz, rgb, input_trans, input_rot, cfg, scalar_summaries=None):
    """localize.py:215-233: run the refinement the config asks for and pick the min-loss candidate.
    Returns (t (3,1), R (3,3), loss) as cpu tensors."""
    summaries = scalar_summaries if scalar_summaries is not None else {}
    if getattr(cfg, "parallel", False):
        results = [omniloc_batch(img, xyz, rgb, input_trans, input_rot, cfg, summaries)]
    else:
        # the reference loops omniloc() over the starting points (localize.py:219-220); same results, one launch chain
        results = omniloc_all(img, xyz, rgb, input_trans, input_rot, cfg, summaries)
    best = min(range(len(results)), key=lambda i: float(results[i][2]))
    return results[best][0], results[best][1], results[best][2]


def pose_errors(t, R, gt_trans, gt_rot):
    """t-error (m) / R-error (deg), localize.py:239-247."""
    return synth.pose_errors(np.asarray(t), np.asarray(R), np.asarray(gt_trans), np.asarray(gt_rot))


def localize_synthetic(cfg, writer=None, log_dir=None):
    """Localise `num_images` synthetic panoramas of the box room (SURVEY.md §8d recipe) and report per-image errors.

    Query images are sharded over the ranks of an initialised process group (one process per GPU); every rank
    returns the full (num_images, 16) result table [t(3), R(9), loss, t_err, r_err, seconds]."""
    dev = ops.device()
    n = int(getattr(cfg, "num_points", 100_000))
    H, W = int(getattr(cfg, "pano_height", 256)), int(getattr(cfg, "pano_width", 512))
    n_img = int(getattr(cfg, "num_images", 4))
    B = int(getattr(cfg, "num_input", 6))
    xyz_np, rgb_np = synth.box_room(n, seed=0)
    xyz, rgb = torch.from_numpy(xyz_np).to(dev), torch.from_numpy(rgb_np).to(dev)

    def body(k):
        t_gt, ypr_gt = synth.gt_pose(k)
        cam = ops.transform_cloud(xyz, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt))
        img = synth.quantise_like_image_file(ops.make_pano(cam, rgb, (H, W)))
        img, rgb_k = preprocess_colors(img, rgb, cfg)
        tr, ro = synth.start_poses(t_gt, ypr_gt, B, seed=k, sigma_t=float(getattr(cfg, "start_sigma_t", 0.3)),
                                   sigma_r=float(getattr(cfg, "start_sigma_r", 0.15)))
        torch.cuda.synchronize()
        t0 = time.time()
        t, R, loss = refine_image(img, xyz, rgb_k, torch.from_numpy(tr).to(dev), torch.from_numpy(ro).to(dev), cfg)
        dt = time.time() - t0
        t_err, r_err = pose_errors(t, R, t_gt, synth.rot_from_ypr_np(ypr_gt))
        return torch.cat([t.reshape(3), R.reshape(9), loss.reshape(1), torch.tensor([t_err, r_err, dt])])

    table = pdist.localize_sharded(n_img, body, dev)
    rank, _ = pdist.world()
    if rank == 0 and log_dir is not None:
        os.makedirs(log_dir, exist_ok=True)
        with open(os.path.join(log_dir, "synthetic_results.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["image", "t_error (m)", "r_error (degrees)", "loss", "time (s)"])
            for k, row in enumerate(table.cpu().numpy()):
                w.writerow([k, row[13], row[14], row[12], row[15]])
    return table
