"""GPU integration test of the dataset harness: a synthetic room written to disk in the
Stanford2D-3D-S and OmniScenes directory layouts (text cloud, PNG panoramas, pose files) is localised end to end through
main.py's code path — native text parser -> colour preprocessing -> make_input -> refinement -> errors -> CSV."""
import csv
import json
import os

import numpy as np
import pytest
import torch

from conftest import Cfg

pytestmark = pytest.mark.gpu

H, W, N = 256, 512, 120_000
COMMON = dict(num_trans=30, xy_only=False, yaw_only=False, num_yaw=4, num_pitch=4, num_roll=4, criterion="loss_histogram",
              num_intermediate=20, num_input=8, num_split_h=4, num_split_w=4, lr=0.1, num_iter=100, patience=5, factor=0.8,
              out_of_room_quantile=0.05, sample_rate=1, parallel=True, num_bins=256)


def _euler_for_stanford(R_gt):
    """final_camera_rotation such that data_utils.obtain_gt_stanford returns R_gt (inverse of data_utils.py:78-90)."""
    flip = np.diag([-1.0, -1.0, 1.0])
    rot = (flip @ R_gt).T
    r = np.stack([rot[:, 1], rot[:, 2], rot[:, 0]], axis=1)
    b = -np.arcsin(r[2, 0])
    a = np.arctan2(r[2, 1], r[2, 2])
    c = np.arctan2(r[1, 0], r[0, 0])
    return [float(a), float(b), float(c)]


def _scene():
    from piccolo_amd import synth
    xyz, rgb = synth.box_room(N, 21)
    rgb8 = np.clip(np.round(rgb * 255), 0, 255).astype(np.uint8)           # dataset files hold integer colours
    return xyz.astype(np.float32), rgb8


def _render(xyz, rgb8, t, ypr):
    from piccolo_amd import ops, synth
    X = torch.from_numpy(xyz).cuda()
    C = torch.from_numpy(rgb8.astype(np.float32) / np.float32(255)).cuda()
    cam = ops.transform_cloud(X, torch.from_numpy(t), torch.from_numpy(ypr))
    return ops.make_pano(cam, C, (H, W)).cpu().numpy().astype(np.uint8), synth.rot_from_ypr_np(ypr)


def _write_cloud(path, xyz, rgb8):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        for p, c in zip(xyz, rgb8):
            f.write("%.4f %.4f %.4f %d %d %d\n" % (p[0], p[1], p[2], c[0], c[1], c[2]))


POSES = [(np.array([0.8, -0.5, 0.1], np.float32), np.array([0.7, 0.03, -0.02], np.float32)),
         (np.array([-1.1, 0.9, -0.2], np.float32), np.array([3.9, -0.04, 0.05], np.float32))]


def test_localize_stanford_layout(tmp_path):
    from PIL import Image
    from piccolo_amd import localize
    root = tmp_path / "stanford"
    xyz, rgb8 = _scene()
    _write_cloud(str(root / "pcd_not_aligned/area_3/office_1.txt"), xyz, rgb8)
    os.makedirs(root / "pano/area_3")
    os.makedirs(root / "pose/area_3")
    names = []
    for k, (t, ypr) in enumerate(POSES + [(np.array([4.5, 0.0, 0.0], np.float32), np.zeros(3, np.float32))]):
        pano, R = _render(xyz, rgb8, t, ypr)
        stem = "camera_c%03d_office_1_frame_equirectangular_domain" % k
        Image.fromarray(pano).save(root / "pano/area_3" / (stem + "_rgb.png"))
        with open(root / "pose/area_3" / (stem + "_pose.json"), "w") as f:
            json.dump({"camera_location": [float(v) for v in t], "final_camera_rotation": _euler_for_stanford(R.astype(np.float64))}, f)
        names.append(stem + "_rgb.png")
    log = tmp_path / "log"
    cfg = Cfg(dataset="Stanford2D-3D-S", area=3, sharpen_color=True, **COMMON)
    table = localize.localize_stanford(cfg, None, str(log), root=str(root)).cpu().numpy()
    assert table.shape == (3, 16)
    # localised by the reference's own criterion (t < 0.2 m, R < 0.2 rad, localize.py:250) with a wide margin; how close a
    # free-running refinement gets on this sparse 120k-point room moves by centimetres with last-bit changes (the
    # equalised colours vs the un-equalised main image, localize.py:175-213, bias it too) — accuracy itself is pinned in
    # test_hip_parity.py
    assert (table[:2, 13] < 0.15).all() and (table[:2, 14] < 2.0).all(), table[:, 13:15]
    assert np.isnan(table[2]).all()                                         # camera outside the quantile box of the cloud: skipped
    with open(log / "stanford_results.csv") as f:
        rows = list(csv.reader(f))
    assert rows[0] == ["area_num", "pano_name", "gt_trans", "gt_rot", "skipped?", "OmniLoc_trans", "OmniLoc_rot", "t_error (m)",
                       "r_error (degrees)", "time (s)"]
    assert [r[1] for r in rows[1:]] == names and [r[4] for r in rows[1:]] == ["0", "0", "1"]
    assert abs(float(rows[1][7]) - table[0, 13]) < 1e-6 and len(rows[3]) == 5
    est = np.array(rows[1][5].split(), np.float64)
    assert np.abs(est - POSES[0][0]).max() < 0.15
    img = Image.open(log / "results/area_3" / names[0])
    assert img.size == (W // 2, 2 * (H // 2))                               # GT panorama over the render, half resolution
    # images_per_launch with sharpen_color: every image has its own equalised cloud colours, so the batcher falls back to
    # one image per launch — same table as above
    again = localize.localize_stanford(Cfg(**{**cfg.__dict__, "images_per_launch": 4}), None, None, root=str(root)).cpu().numpy()
    assert np.array_equal(np.isnan(again), np.isnan(table)) and np.allclose(again[:2, :13], table[:2, :13], atol=0, rtol=0)


def test_localize_omniscenes_layout(tmp_path):
    from PIL import Image
    from piccolo_amd import localize
    root = tmp_path / "omniscenes"
    xyz, rgb8 = _scene()
    _write_cloud(str(root / "pcd/room_1.txt"), xyz, rgb8)
    video = "handheld_room_1_scene_2"
    os.makedirs(root / "extreme_pano" / video)
    os.makedirs(root / "extreme_pose" / video)
    poses = POSES
    for k, (t, ypr) in enumerate(poses):
        pano, R = _render(xyz, rgb8, t, ypr)
        # 2048 x 1024 like the dataset's frames (pixel-replicated, so that the holes of the sparse render stay black), stored
        # losslessly under the dataset's .jpg name
        big = np.repeat(np.repeat(pano, 4, axis=0), 4, axis=1)
        Image.fromarray(big).save(root / "extreme_pano" / video / ("%06d.jpg" % k), format="PNG")
        np.savetxt(root / "extreme_pose" / video / ("%06d.txt" % k), np.hstack([R.astype(np.float64), t.reshape(3, 1).astype(np.float64)]))
    log = tmp_path / "log"
    # init_downsample 8 // 2 = 4: the initialisation runs on 512 x 256 (the resolution the panoramas were rendered at)
    base = dict(dataset="OmniScenes", init_downsample_h=8, init_downsample_w=8, main_downsample_h=2, main_downsample_w=2, scene_number=2,
                **{**COMMON, "num_intermediate": 40, "parallel": False})
    table = localize.localize_omniscenes(Cfg(**base), None, str(log), root=str(root)).cpu().numpy()
    assert table.shape == (2, 16) and np.isfinite(table).all()
    # Frame 1 is localised with every texel format and every build; frame 0 sits next to a second basin in this sparse
    # 120k-point room and the free-running refinement lands in either depending on last-bit differences (texel format,
    # summation order of a build: 0.03 m or the neighbour 1.7 m away) — the plumbing is what is under test here, so frame
    # 0 is only required to stay inside the room.
    assert table[1, 13] < 0.08 and table[1, 14] < 1.5, table[:, 13:15]
    assert table[0, 13] < 4.0, table[:, 13:15]
    with open(log / "omniscenes_results.csv") as f:
        rows = list(csv.reader(f))
    assert rows[0][0] == "pano_name" and rows[1][0] == video + "/000000.jpg" and rows[1][3] == "0"
    # the shipped OmniScenes config: match_color.  In this synthetic room colour encodes position, so remapping the
    # panorama's colours to the cloud's distribution (visibility-weighted vs uniform) biases the pose — the loop may
    # return the room's 180-degree twin; only the plumbing is asserted: finite results, one CSV row per frame.
    table = localize.localize_omniscenes(Cfg(match_color=True, synth_gamma=1.1, **base), None, str(tmp_path / "log2"), root=str(root)).cpu().numpy()
    assert table.shape == (2, 16) and np.isfinite(table).all()
    with open(tmp_path / "log2" / "omniscenes_results.csv") as f:
        assert len(list(csv.reader(f))) == 3
    # images_per_launch: both frames of the room refined in one launch chain (same cloud tensors, same image size); every
    # frame still gets its own row; frame 1 again localised
    both = localize.localize_omniscenes(Cfg(images_per_launch=4, save_starting_point=True, **base), None, str(tmp_path / "log3"),
                                        root=str(root)).cpu().numpy()
    assert both.shape == (2, 16) and np.isfinite(both).all() and both[1, 13] < 0.08 and both[1, 14] < 1.5
    with open(tmp_path / "log3" / "omniscenes_results.csv") as f:
        assert len(list(csv.reader(f))) == 3
    assert (tmp_path / "log3" / "results" / video / "000001.png").exists()
    # cfg.save_starting_point (localize.py:457-471): one stacked query / render image per starting pose and frame, at half the
    # 2048 x 1024 frame's resolution
    for frame in ("000000", "000001"):
        for idx in range(COMMON["num_input"]):
            pth = tmp_path / "log3" / "starting_points" / video / ("%s_%d.png" % (frame, idx))
            assert pth.exists(), pth
    assert Image.open(tmp_path / "log3" / "starting_points" / video / "000001_0.png").size == (1024, 1024)
    # gravity_aligned = False: the reference's own path stops at an undefined helper; refused up front here
    with pytest.raises(NotImplementedError):
        localize.localize_omniscenes(Cfg(gravity_aligned=False, **base), None, str(tmp_path / "log4"), root=str(root))
    # filters of the loop
    none = localize.localize_omniscenes(Cfg(**{**base, "scene_number": 7}), None, None, root=str(root))
    assert tuple(none.shape) == (0, 16)


def test_resize_image_geometry():
    from piccolo_amd.utils import resize_image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (8, 16, 3)).astype(np.uint8)
    assert resize_image(img, 16, 8) is img
    half = resize_image(img, 8, 4)                                         # exact 2x reduction = 2x2 box mean with cv2's geometry
    box = img.reshape(4, 2, 8, 2, 3).astype(np.float64).mean(axis=(1, 3))
    assert np.abs(half.astype(np.float64) - box).max() <= 0.5 + 1e-9
    up = resize_image(img, 32, 16)
    assert up.shape == (16, 32, 3) and up.min() >= img.min() and up.max() <= img.max()


def test_resize_image_agrees_with_torch_bilinear_within_one_level():
    """cv2.resize (INTER_LINEAR: pixel centres at k + 0.5, edge clamp, no antialiasing) is absent; torch's
    F.interpolate(mode="bilinear", align_corners=False, antialias=False) is an independent implementation of the same
    geometry.  The restatement must agree with it to one 8-bit level (rounding of .5 ties) for down- and up-scaling by integer
    and non-integer factors — the factors the configs use (init / main_downsample_h, _w) and the 2048 x 1024 normalisation
    of the OmniScenes loop."""
    import torch.nn.functional as Fn
    from piccolo_amd.utils import resize_image
    rng = np.random.default_rng(5)
    for (H, W), (h, w) in (((64, 128), (32, 64)), ((64, 128), (16, 32)), ((60, 100), (45, 80)), ((48, 96), (96, 192)),
                           ((50, 70), (128, 256)), ((96, 200), (32, 50))):
        img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        mine = resize_image(img, w, h).astype(np.int16)
        ref = Fn.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(h, w), mode="bilinear", align_corners=False,
                             antialias=False)[0].permute(1, 2, 0).numpy()
        assert mine.shape == (h, w, 3)
        assert np.abs(mine - ref).max() <= 0.5 + 1e-6, ((H, W), (h, w), np.abs(mine - ref).max())
