"""Recall / precision of the scatter-min depth mask against ANALYTIC occlusion, as a function of the z-buffer's grid, of tau and of
the occluder stride (CPU only: the oracle's pixels and scatter-min; nothing of the product runs here).

synth.furnished_room is a box room with three axis-aligned boxes.  From a camera at t a surface point P is truly occluded iff the
open segment t -> P passes through the interior of a box (synth.occluded_by_furniture: the walls / floor behind furniture AND the
faces of a box that point away from the camera).  The mask hides P iff ||P - t|| > zmin(cell of P) * (1 + tau), cells = make_pano's
pixels (utils.py:158-165) on the depth grid, zmin = scatter-min over every stride-th point of the Morton-ordered cloud.

    recall    = hidden and occluded / occluded         (what the mask is for)
    precision = hidden and occluded / hidden           (1 - precision = visible points the mask throws away: at a coarse grid a
                                                        surface seen at a grazing angle spans more than tau in depth inside one
                                                        cell and hides its own far side)

Part 1: every point builds the z-buffer, grids from 96 to 3 points per cell plus the panorama's own 2048 x 1024 (round 4's z-buffer).
Part 2: strides 1, 2, 4, each on the grid pcl_depth_default gives ITS sample count.  `*` marks pcl_depth_default's choice.

usage: python tools/depth_recall.py [n_points] [n_images]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle                      # noqa: E402  (a measurement tool, not the product)
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import synth                  # noqa: E402


def morton_order(xyz):
    """the packed cloud's order restated on the CPU (21 bits per axis inside the bounding box, csrc/pcl_pack.hip)"""
    lo, hi = xyz.min(0), xyz.max(0)
    q = np.clip(((xyz - lo) / (hi - lo) * (2 ** 21 - 1)).astype(np.uint64), 0, 2 ** 21 - 1)

    def spread(v):
        v = v.astype(np.uint64)
        for sh, m in ((32, 0x1f00000000ffff), (16, 0x1f0000ff0000ff), (8, 0x100f00f00f00f00f), (4, 0x10c30c30c30c30c3), (2, 0x1249249249249249)):
            v = (v | (v << np.uint64(sh))) & np.uint64(m)
        return v
    key = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
    return np.argsort(key, kind="stable")


def grid_h(m):
    """pcl_depth_default's grid for m occluder samples: >= 12 per cell"""
    h = 16
    while (h + 8) * (h + 8) * 24 <= m:
        h += 8
    return h


def default_choice(n):
    stride = 1
    for c in (2, 4):
        if grid_h(n // c) >= 128:
            stride = c
    h = grid_h((n + stride - 1) // stride)
    return stride, h, round(min(max(3.5 * np.pi / h, 0.02), 0.15), 3)


def scores(cam, occ, res, tau, stride):
    zmin, _ = oracle.scatter_min_depth(cam[::stride], res)
    zmin = np.where(zmin == 0, np.inf, zmin)                  # torch_scatter's 0 for an empty cell: nothing in front
    row, col = oracle.pano_pixels(cam, res)
    d = np.linalg.norm(cam.astype(np.float64), axis=1)
    hid = d > zmin[row.astype(np.int64) * res[1] + col].astype(np.float64) * (1 + tau)
    tp = float((hid & occ).sum())
    return tp / max(occ.sum(), 1), tp / max(hid.sum(), 1), hid.mean()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n_images = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    xyz, _ = synth.furnished_room(n, 0)
    P = xyz[morton_order(xyz)]
    ids = [i for i in range(60) if not synth.inside_furniture(synth.gt_pose(i)[0])][:n_images]
    d_stride, d_h, d_tau = default_choice(n)
    cams = []
    for image_id in ids:
        t, ypr = synth.gt_pose(image_id)
        cams.append((synth.transform_cloud(P, t, ypr), synth.occluded_by_furniture(P, t)))
    print("%d points, %d poses; truly occluded share: median %.4f; pcl_depth_default: stride %d, grid %dx%d, tau %.3f"
          % (n, len(ids), np.median([o.mean() for _, o in cams]), d_stride, 2 * d_h, d_h, d_tau))
    print("part 1 — every point builds the z-buffer; per tau: recall / precision / hidden share (medians over the poses)")
    taus = [0.02, 0.03, 0.05, 0.07, 0.1]
    grids = [(max(8, int(np.sqrt(n / (2.0 * ppp))) // 8 * 8),) * 1 for ppp in (96, 48, 24, 12, 6, 3)]
    grids = [(g[0], 2 * g[0]) for g in grids] + [(1024, 2048)]
    for g in grids:
        cells = ["%4dx%-4d %6.2f pts/cell" % (g[1], g[0], n / (g[0] * g[1]))]
        for tau in taus:
            a = np.array([scores(cam, occ, g, tau, 1) for cam, occ in cams])
            cells.append("tau %.2f: %.3f / %.3f / %.3f" % (tau, np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 2])))
        print(" | ".join(cells), flush=True)
    print("part 2 — occluder stride s: z-buffer from every s-th point on the grid its sample count calls for (>= 12 samples per cell)")
    for stride in (1, 2, 4):
        h = grid_h((n + stride - 1) // stride)
        rule_tau = round(min(max(3.5 * np.pi / h, 0.02), 0.15), 3)
        cells = ["stride %d %4dx%-4d" % (stride, 2 * h, h)]
        for tau in sorted({rule_tau, 0.05, 0.07, 0.1}):
            a = np.array([scores(cam, occ, (h, 2 * h), tau, stride) for cam, occ in cams])
            mark = " *" if (stride, tau) == (d_stride, d_tau) else (" r" if tau == rule_tau else "")
            cells.append("tau %.3f%s: %.3f / %.3f" % (tau, mark, np.median(a[:, 0]), np.median(a[:, 1])))
        print(" | ".join(cells), flush=True)
    print("(* = pcl_depth_default's choice, r = the rule's tau for that grid)")


if __name__ == "__main__":
    main()
