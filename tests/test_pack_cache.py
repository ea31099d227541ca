"""Host logic of the pack cache (piccolo_amd/omniloc.py): per-kind LRU keyed by tensor identity.  No GPU needed — the
cached objects are stand-ins."""
import gc

import torch

from piccolo_amd import omniloc as po


def test_lru_per_kind_keeps_the_room_while_images_stream_through():
    po._cache.clear()
    xyz, rgb = torch.zeros(10, 3), torch.zeros(10, 3)
    made = []

    def make(tag):
        def f():
            made.append(tag)
            return object()
        return f
    cloud = po._cached("cloud", (xyz, rgb), make("cloud"))
    box = po._cached("box", (xyz,), make("box"), sub=0.05)
    grid = po._cached("grid", (xyz,), make("grid"), sub="cfg-a")
    imgs = [torch.zeros(4, 8, 3) for _ in range(40)]          # far more query images than any capacity
    for im in imgs:
        po._cached("pano", (im,), make("pano"))
        po._cached("pano_u8", (im,), make("pano_u8"))
    # the cloud-side entries survived the stream of panoramas
    assert po._cached("cloud", (xyz, rgb), make("again")) is cloud
    assert po._cached("box", (xyz,), make("again"), sub=0.05) is box
    assert po._cached("grid", (xyz,), make("again"), sub="cfg-a") is grid
    assert "again" not in made
    assert len(po._cache.kinds["pano"]) == po._CAPACITY["pano"] and len(po._cache.kinds["pano_u8"]) == po._CAPACITY["pano_u8"]
    # least recently used goes first: the last image is still there, the first is not
    n = len(made)
    po._cached("pano", (imgs[-1],), make("hit"))
    assert len(made) == n
    po._cached("pano", (imgs[0],), make("miss"))
    assert made[-1] == "miss"


def test_sub_keys_versions_and_dead_tensors():
    po._cache.clear()
    xyz = torch.zeros(10, 3)
    a = po._cached("box", (xyz,), object, sub=0.05)
    b = po._cached("box", (xyz,), object, sub=0.1)
    assert a is not b and po._cached("box", (xyz,), object, sub=0.05) is a
    xyz.add_(1.0)                                              # in-place edit bumps the version: a new entry
    assert po._cached("box", (xyz,), object, sub=0.05) is not a
    # entries of dead tensors are purged on the next insertion (their address may be reused)
    tmp = torch.zeros(5, 3)
    po._cached("box", (tmp,), object, sub=0.05)
    del tmp
    gc.collect()
    keep = torch.ones(7, 3)
    po._cached("box", (keep,), object, sub=0.05)
    shapes = [k[1][1] for k in po._cache.kinds["box"]]
    assert (5, 3) not in shapes and (7, 3) in shapes
    assert len(po._cache.kinds["box"]) <= po._CAPACITY["box"]


def test_debug_visualize_host_helper(monkeypatch):
    """utils.debug_visualize (utils.py:641-699): accepts (H,W), (H,W,C), (B,H,W,C) tensors / arrays, rejects anything else."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    import numpy as np
    import pytest
    from piccolo_amd import utils
    shown = []
    monkeypatch.setattr(plt, "show", lambda *a, **k: shown.append(len(plt.gcf().axes)))
    for t in (torch.rand(8, 16), torch.rand(8, 16, 3), torch.rand(8, 16, 1), torch.rand(2, 8, 16, 4), np.random.rand(8, 16, 2) * 255):
        utils.debug_visualize(t)
        plt.close("all")
    assert shown == [1, 1, 1, 4, 2]
    with pytest.raises(ValueError):
        utils.debug_visualize([1, 2, 3])


def test_levels_tag_follows_the_tensor_version():
    """synth.mark_levels tags an image as 'every value exactly k/255 by construction' so that packing it never reads the
    device-side exactness flag back; an in-place change afterwards voids the tag, ops.EXPERIMENT.verify_levels ignores it (host logic)."""
    import os
    import torch
    from piccolo_amd import ops, synth
    img = synth.mark_levels(torch.randint(0, 256, (4, 8, 3)).float() / 255.0)
    assert ops._known_levels(img)
    assert not ops._known_levels(img.clone()) and not ops._known_levels(img.numpy())      # the tag belongs to this very tensor object
    img[0, 0, 0] = 0.123
    assert not ops._known_levels(img)                                                       # changed in place after tagging
    img2 = synth.quantise_like_image_file(torch.rand(4, 8, 3) * 255)
    assert ops._known_levels(img2) and bool((img2 * 255 == torch.round(img2 * 255)).all())
    ops.EXPERIMENT.verify_levels = True
    try:
        assert not ops._known_levels(img2)
    finally:
        ops.EXPERIMENT.verify_levels = False
