"""Drop-in for the reference's color_utils.py (imported at localize.py:12 and utils.py:7): same function names, argument
meaning and return types; every computation is a HIP kernel behind include/piccolo_hip.h (no CPU path).

    color_mod(img, rgb, num_bins)      color_utils.py:7-65     joint luma equalisation (cfg `sharpen_color`)
    color_match(img, rgb)              color_utils.py:146-234  histogram matching to the point colours (cfg `match_color`)
    histogram(img, mask, channels, normalize), histogram_intersection(h1, h2)      color_utils.py:68-144

Results come back on the device of `img` (CPU tensors in -> CPU tensors out, like the other stand-alone ops).
"""
import weakref

import torch

from . import ops

_templates = {}


def _template(rgb):
    """ColorTemplate (point colours sorted per channel) cached per rgb tensor: the harness matches many query images
    against the colours of one cloud (localize.py:357-403)."""
    key = (rgb.data_ptr(), tuple(rgb.shape), rgb._version, str(rgb.device))
    hit = _templates.get(key)
    if hit is not None and hit[0]() is rgb:
        return hit[1]
    t = ops.ColorTemplate(rgb)
    if len(_templates) > 4:
        _templates.clear()
    _templates[key] = (weakref.ref(rgb), t)
    return t


def color_match(img, rgb):
    """Match the colour distribution of the panorama to that of the point cloud (color_utils.py:146-234).
    img (H,W,3) in [0,1] holding levels k/255, rgb (N,3) -> img (H,W,3); black pixels stay black."""
    out = ops.color_match(img, _template(rgb))
    return out.to(img.device)


def color_mod(img, rgb, num_bins):
    """Joint histogram equalisation of the luma of the panorama and of the point colours (color_utils.py:7-65).
    -> (img (H,W,3), rgb (N,3)).  Unlike the reference, `img` is not modified in place."""
    out_img, out_rgb = ops.color_mod(img, rgb, num_bins)
    return out_img.to(img.device), out_rgb.to(img.device)     # the reference returns both on img's device (:53, :62)


def histogram(img, mask, channels=[32, 32, 32], normalize=True):
    """Colour histogram of the masked pixels (color_utils.py:68-118): (H,W,3)/(H,W) -> (*channels);
    batched (B,H,W,3)/(B,H,W) -> (B,*channels) with the batched form's eps in the normalisation."""
    if img.dim() == 3:
        return ops.histogram(img, mask, channels, normalize, 0.0).reshape(*channels).to(img.device)
    hists = [ops.histogram(img[b], mask[b], channels, normalize, 1e-6) for b in range(img.shape[0])]
    return torch.stack(hists).reshape(img.shape[0], *channels).to(img.device)


def histogram_intersection(hist_1, hist_2):
    """Sum of the element-wise minimum (color_utils.py:122-144): scalar tensor, or (B,) for batched histograms."""
    assert hist_1.shape == hist_2.shape
    if hist_1.dim() == 3:
        return ops.histogram_intersection(hist_1.reshape(1, -1), hist_2.reshape(1, -1))[0].to(hist_1.device)
    B = hist_1.shape[0]
    return ops.histogram_intersection(hist_1.reshape(B, -1), hist_2.reshape(B, -1)).to(hist_1.device)
