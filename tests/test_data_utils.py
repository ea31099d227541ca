"""Dataset text clouds and ground-truth pose conventions: the native host parser and the pose helpers
against what the reference's data_utils.py returned for the same files (tests/golden/g15_data_utils.npz; the fixture
holds the text / pose inputs and the reference's outputs).  The parser and the pose helpers are host code and are tested
without a GPU; the `gpu`-marked tests at the end take the parsed file all the way into the loss kernel."""
import json
import os

import numpy as np
import pytest

from conftest import load_golden


@pytest.fixture(scope="module")
def du():
    from piccolo_amd import build, data_utils
    build.build()                      # hipcc cross-compiles without a GPU; the parser is host code in the same library
    return data_utils


@pytest.fixture(scope="module")
def cloud_file(tmp_path_factory):
    g = load_golden("g15_data_utils.npz")
    path = tmp_path_factory.mktemp("cloud") / "cloud.txt"
    path.write_bytes(g["cloud_txt"].tobytes())
    return str(path)


def test_read_cloud_matches_reference_bits(du, cloud_file):
    g = load_golden("g15_data_utils.npz")
    for fn in (du.read_stanford, du.read_omniscenes):
        xyz, rgb = fn(cloud_file)
        assert xyz.dtype == np.float64 and rgb.dtype == np.float64
        assert np.array_equal(xyz, g["xyz"])               # same doubles as pandas' parser, bit for bit
        assert np.array_equal(rgb, g["rgb"])


def test_read_cloud_subsample_uses_numpy_global_rng(du, cloud_file):
    g = load_golden("g15_data_utils.npz")
    np.random.seed(7)
    xyz, rgb = du.read_stanford(cloud_file, sample_rate=4)
    assert np.array_equal(xyz, g["xyz_s4"]) and np.array_equal(rgb, g["rgb_s4"])


def test_read_cloud_threads_agree_on_large_file(du, tmp_path):
    """> 1 MiB so that the file is cut into per-thread slices: same table as a single-threaded parse and as numpy."""
    from piccolo_amd import _lib
    rng = np.random.default_rng(0)
    n = 60_000
    table = np.hstack([rng.normal(0, 10, (n, 3)).round(4), rng.integers(0, 256, (n, 3)).astype(np.float64)])
    path = tmp_path / "big.txt"
    with open(path, "w") as f:
        for i, row in enumerate(table):
            f.write("%.4f %.4f %.4f %d %d %d\n" % tuple(row))
            if i % 1000 == 0:
                f.write("\n")
    assert os.path.getsize(path) > (1 << 20)
    lib = _lib.load()
    rows = lib.pcl_cloud_txt_rows(os.fsencode(str(path)))
    assert rows == n
    multi, single = np.empty((n, 6)), np.empty((n, 6))
    assert lib.pcl_cloud_txt_read(os.fsencode(str(path)), n, 6, multi.ctypes.data, 0) == 0
    assert lib.pcl_cloud_txt_read(os.fsencode(str(path)), n, 6, single.ctypes.data, 1) == 0
    assert np.array_equal(multi, single) and np.array_equal(multi, table)
    xyz, rgb = du.read_stanford(str(path))
    assert np.array_equal(xyz, table[:, :3]) and np.array_equal(rgb, table[:, 3:] / 255.)


@pytest.mark.parametrize("bad,line", [("1 2 3 4 5\n", 1), ("1 2 3 4 5 6\n1 2 3 4 5 6 7\n", 2), ("1 2 3 4 5 6\n\n1 2 x 4 5 6\n", 3),
                                      ("1 2 3 4 5 185\x100\n", 1)])
def test_read_cloud_reports_malformed_line(du, tmp_path, bad, line):
    path = tmp_path / "bad.txt"
    path.write_text(bad)
    with pytest.raises(ValueError, match="line %d" % line):
        du.read_stanford(str(path))


def test_read_cloud_missing_and_empty(du, tmp_path):
    with pytest.raises(OSError):
        du.read_stanford(str(tmp_path / "nope.txt"))
    empty = tmp_path / "empty.txt"
    empty.write_text("\n\n")
    xyz, rgb = du.read_stanford(str(empty))
    assert xyz.shape == (0, 3) and rgb.shape == (0, 3)


def test_special_values_match_reference(du, tmp_path):
    """nan / inf spellings, underflow, 17+ digit mantissas, leading zeros, exponents: the doubles pandas' default
    converter produced for the reference (incl. its quirks: digits beyond the 17th only shift the exponent, leading
    zeros count as digits)."""
    g = load_golden("g15_data_utils.npz")
    path = tmp_path / "special.txt"
    path.write_bytes(g["special_txt"].tobytes())
    xyz, rgb = du.read_stanford(str(path))
    assert np.array_equal(xyz.view(np.uint64), g["special"][:, :3].copy().view(np.uint64))     # bit patterns: nan, -0.0
    assert np.array_equal(rgb.view(np.uint64), g["special_rgb"].copy().view(np.uint64))


def test_out_of_range_number_is_malformed(du, tmp_path):
    """1e400 does not fit a double: pandas leaves such a column as strings and the reference then fails on `/ 255.`;
    here the line is reported."""
    path = tmp_path / "o.txt"
    path.write_text("1 2 3 4 5 6\n1 2 1e400 4 5 6\n")
    with pytest.raises(ValueError, match="line 2"):
        du.read_stanford(str(path))


def test_gt_poses(du, tmp_path):
    g = load_golden("g15_data_utils.npz")
    root = tmp_path / "pose"
    (root / "area_3").mkdir(parents=True)
    (root / "area_30").mkdir(parents=True)
    name = "camera_abc123_office_7_frame_equirectangular_domain_rgb.png"
    with open(root / "area_3" / "camera_abc123_office_7_frame_equirectangular_domain_pose.json", "w") as f:
        json.dump({"camera_location": g["pose_loc"].tolist(), "final_camera_rotation": g["pose_rot"].tolist()}, f)
    np.savetxt(root / "area_30" / "office_7.txt", g["align"])
    t3, r3 = du.obtain_gt_stanford(3, name, root=str(root))
    assert t3.shape == (3, 1) and np.array_equal(t3, g["gt3_t"])
    assert np.abs(r3 - g["gt3_r"]).max() <= 1e-15          # scipy's quaternion route vs three plane rotations
    t30, r30 = du.obtain_gt_stanford(30, name, root=str(root))
    assert np.abs(t30 - g["gt30_t"]).max() <= 1e-15 and np.abs(r30 - g["gt30_r"]).max() <= 1e-15
    assert np.abs(r3 @ r3.T - np.eye(3)).max() <= 1e-15 and abs(np.linalg.det(r3) - 1) <= 1e-15
    (tmp_path / "omni" / "pano").mkdir(parents=True)
    (tmp_path / "omni" / "pose").mkdir(parents=True)
    np.savetxt(tmp_path / "omni" / "pose" / "room_1.txt", g["omni"])
    to, ro = du.obtain_gt_omniscenes(str(tmp_path / "omni" / "pano" / "room_1.jpg"))
    assert np.array_equal(to, g["omni_t"]) and np.array_equal(ro, g["omni_r"])


# ------------------------------------------------------------------------------------------------ file -> GPU (-m gpu)
@pytest.mark.gpu
def test_load_cloud_gives_the_reference_tensors_on_the_gpu(du, cloud_file):
    """load_cloud: dataset text file -> float32 CUDA tensors.  The reference gets there by read_stanford (pandas, float64,
    rgb / 255. in float64) and torch.from_numpy(...).float().to(device) (localize.py:159-162): the tensors must hold exactly
    the float32 roundings of G15's doubles — with and without the binary side-car cache, and with subsampling."""
    import torch
    g = load_golden("g15_data_utils.npz")
    side = cloud_file + ".pcl.npy"
    if os.path.exists(side):
        os.remove(side)
    for attempt in ("parse", "side-car"):
        xyz, rgb = du.load_cloud(cloud_file)
        assert xyz.is_cuda and rgb.is_cuda and xyz.dtype == torch.float32 and rgb.dtype == torch.float32
        assert np.array_equal(xyz.cpu().numpy(), g["xyz"].astype(np.float32)), attempt
        assert np.array_equal(rgb.cpu().numpy(), g["rgb"].astype(np.float32)), attempt
        assert os.path.exists(side)
    np.random.seed(7)
    xs, cs = du.load_cloud(cloud_file, sample_rate=4)
    assert np.array_equal(xs.cpu().numpy(), g["xyz_s4"].astype(np.float32)) and np.array_equal(cs.cpu().numpy(), g["rgb_s4"].astype(np.float32))
    xyz2, _ = du.load_cloud(cloud_file, cache=False)
    assert torch.equal(xyz2, xyz)


@pytest.mark.gpu
def test_text_cloud_through_the_loss_kernel(du, oracle, parity, tmp_path):
    """A dataset-format text cloud written to disk, parsed by the native reader, uploaded, packed and evaluated by the HIP
    loss kernel == the fp64 oracle fed the doubles numpy parses from the same text (rounded to float32 like the
    reference's .float())."""
    import torch
    from parity_helpers import T, _check_vs_oracle, _oracle_pair
    from piccolo_amd import ops, synth
    n, H, W, B = 30_011, 96, 192, 4
    xyz, rgb = synth.box_room(n, 77)
    lv = np.rint(rgb * 255).astype(np.int64)                       # dataset colours are 8-bit levels
    path = tmp_path / "room.txt"
    with open(path, "w") as f:
        for p, c in zip(xyz.astype(np.float64), lv):
            f.write("%.6f %.6f %.6f %d %d %d\n" % (p[0], p[1], p[2], c[0], c[1], c[2]))
    X, C = du.load_cloud(str(path))
    table = np.loadtxt(path)                                        # an independent parser
    x32, c32 = table[:, :3].astype(np.float32), (table[:, 3:] / 255.).astype(np.float32)
    assert np.array_equal(X.cpu().numpy(), x32) and np.array_equal(C.cpu().numpy(), c32)
    t_gt, ypr_gt = synth.gt_pose(77)
    img = oracle.make_pano_u8(synth.transform_cloud(x32, t_gt, ypr_gt), c32, (H, W)).astype(np.float32) / 255
    trans, rot = synth.start_poses(t_gt, ypr_gt, B, seed=77)
    out = ops.sampling_loss(ops.Cloud(X, C), ops.Pano(T(img)), T(trans), T(rot)).cpu().numpy()
    r64, r32 = _oracle_pair(oracle, x32, c32, img, trans, rot)
    _check_vs_oracle(parity, out, r64, r32, n)
