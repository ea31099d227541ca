"""CPU restatement of the second trimming stage, trim_input_hist_secondary (utils.py:510-588) with color_utils.histogram
(color_utils.py:68-118) and histogram_intersection (color_utils.py:122-144) — TEST INFRASTRUCTURE.

Per candidate pose: render the cloud with make_pano at the image's resolution, and for every block of the middle block
rows (h = 1 .. num_split_h - 2, utils.py:556) intersect the normalised 8x8x8 colour histogram of the rendered pixels
(where both the render and the query image are non-black) with the histogram of the query image's non-black pixels;
score = sum of the block intersections / (num_split_h * num_split_w).  Candidates are ranked by score, best first.

Quirk reproduced (pinned by G19): the reference keeps ONE slot vector `hist_intersect_split` for all candidates
(utils.py:539).  A block with no rendered or no query pixel writes 0 into its slot and `break`s out of its block row
(utils.py:568-571); the remaining slots of that row keep whatever the last candidate that reached them left there, so a
candidate's score can include an earlier candidate's intersections.  `inter` returned here is that slot vector as it
stands after each candidate (NaNs cleaned in place, utils.py:579).
"""
import math

import numpy as np

from . import oracle as orc

BINS = 8


def _codes(img255):
    """8x8x8 bin index per pixel: value.long() // ceil(255 / 8) per channel (color_utils.py:86-95)."""
    q = np.floor(np.asarray(img255, np.float64)).astype(np.int64) // int(math.ceil(255 / BINS))
    return q[..., 0] + BINS * q[..., 1] + BINS * BINS * q[..., 2]


def block_hist(code, mask, h, w, bh, bw):
    m = mask[h * bh:(h + 1) * bh, w * bw:(w + 1) * bw]
    c = code[h * bh:(h + 1) * bh, w * bw:(w + 1) * bw][m]
    hist = np.bincount(c, minlength=BINS ** 3).astype(np.float32)
    n = int(m.sum())
    return (hist / np.float32(hist.sum())) if n else hist, n


def hist_scores(img, xyz, rgb, trans, rot, num_split_h, num_split_w):
    """(scores (K,), inter (K, num_split_h*num_split_w)) for K candidate poses."""
    img255 = np.asarray(img, np.float32) * np.float32(255)
    H, W, _ = img255.shape
    bh, bw = H // num_split_h, W // num_split_w
    img_mask = ~(img255 == 0).all(axis=2)
    img_code = _codes(img255)
    K = len(trans)
    inter = np.zeros((K, num_split_h * num_split_w), np.float64)
    slots = np.zeros(num_split_h * num_split_w, np.float64)          # hist_intersect_split: allocated once (utils.py:539)
    for i in range(K):
        R = orc.rot_from_ypr(rot[i], np.float32)
        cam = ((np.asarray(xyz, np.float32) - np.asarray(trans[i], np.float32)[None, :]) @ R.T).astype(np.float32)
        proj = orc.make_pano(cam, rgb, (H, W))
        both = ~(proj == 0).all(axis=2) & img_mask
        proj_code = _codes(proj)
        for h in range(1, num_split_h - 1):
            for w in range(num_split_w):
                hp, n_p = block_hist(proj_code, both, h, w, bh, bw)
                hq, n_q = block_hist(img_code, img_mask, h, w, bh, bw)
                if n_p == 0 or n_q == 0:
                    slots[h * num_split_w + w] = 0.0                 # utils.py:569-571: zero this slot, leave the row
                    break
                slots[h * num_split_w + w] = float(np.minimum(hp, hq).sum())
        slots[np.isnan(slots)] = 0.0                                 # utils.py:579, in place
        inter[i] = slots
    scores = inter.sum(1) / (num_split_h * num_split_w)
    return scores, inter


def trim_input_hist_secondary(img, xyz, rgb, trans, rot, num_input, num_split_h, num_split_w):
    scores, _ = hist_scores(img, xyz, rgb, trans, rot, num_split_h, num_split_w)
    order = np.argsort(scores, kind="stable")[-num_input:][::-1]
    return np.asarray(trans)[order], np.asarray(rot)[order], scores
