"""Pins the CPU oracle (oracle/) against golden vectors produced by running the reference
(tests/golden/gen_goldens.py).  CPU only."""
import json

import numpy as np
import pytest

from conftest import Cfg, load_golden


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# G1 ------------------------------------------------------------------------------------------
def test_cloud2idx_f32_f64(oracle):
    g = load_golden("g1_cloud2idx.npz")
    out = oracle.cloud2idx(g["xyz"])
    # libm atan2f vs ATen's (sleef) atan2f differ by <= 1-2 ulp; coordinates are O(1)
    assert np.abs(out - g["coord"]).max() <= 5e-7
    out64 = oracle.cloud2idx(g["xyz"].astype(np.float64))
    assert np.abs(out64 - g["coord_f64"]).max() <= 1e-14
    outb = oracle.cloud2idx(g["xyz_b"])
    assert outb.shape == g["coord_b"].shape
    assert np.abs(outb - g["coord_b"]).max() <= 5e-7


# G2 ------------------------------------------------------------------------------------------
def test_sample_from_img(oracle):
    g = load_golden("g2_sample_from_img.npz")
    out = oracle.sample_from_img(g["img"], g["coord"])
    assert np.abs(out - g["rgb"]).max() <= 2e-6
    # exact-zero pattern (drives the loss mask) must be identical
    assert np.array_equal(out == 0, g["rgb"] == 0)
    out64 = oracle.sample_from_img(g["img"].astype(np.float64), g["coord"].astype(np.float64))
    assert np.abs(out64 - g["rgb_f64"]).max() <= 1e-13
    outb = oracle.sample_from_img(g["img"], g["coord_b"])
    assert np.abs(outb - g["rgb_b"]).max() <= 2e-6


# G3 / G4 -------------------------------------------------------------------------------------
def test_sampling_loss_and_grad_f64(oracle):
    g = load_golden("g3_sampling_loss.npz")
    o = oracle.sampling_loss(g["xyz"], g["rgb"], g["img"], g["trans"], g["rot"], dtype=np.float64)
    assert rel(o["loss"], g["loss_f64"]) <= 1e-13
    assert rel(o["grad_t"], g["grad_t_f64"]) <= 1e-11
    assert rel(o["grad_ypr"], g["grad_ypr_f64"]) <= 1e-11


def test_sampling_loss_and_grad_f32(oracle):
    g = load_golden("g3_sampling_loss.npz")
    o = oracle.sampling_loss(g["xyz"], g["rgb"], g["img"], g["trans"], g["rot"], dtype=np.float32)
    assert rel(o["loss"], g["loss_f32"]) <= 2e-6
    # fp32 per-point rounding differs from ATen's op order: the reference's own fp32-vs-fp64 gap is the yardstick
    gap_t = rel(g["grad_t_f32"], g["grad_t_f64"])
    gap_r = rel(g["grad_ypr_f32"], g["grad_ypr_f64"])
    assert rel(o["grad_t"], g["grad_t_f64"]) <= max(3 * gap_t, 1e-4)
    assert rel(o["grad_ypr"], g["grad_ypr_f64"]) <= max(3 * gap_r, 1e-4)
    assert rel(o["grad_t"], g["grad_t_f32"]) <= 3e-4
    assert rel(o["grad_ypr"], g["grad_ypr_f32"]) <= 3e-4


def test_batch_sampling_loss(oracle):
    s = load_golden("g3_sampling_loss.npz")
    g = load_golden("g4_batch_sampling_loss.npz")
    o = oracle.sampling_loss(s["xyz"], s["rgb"], s["img"], g["trans"], g["rot"], dtype=np.float64)
    assert rel(o["loss"], g["loss_list_f64"]) <= 1e-13
    assert abs(o["loss"].sum() - g["loss_f64"]) <= 1e-12
    assert rel(o["grad_t"], g["grad_t_f64"]) <= 1e-11
    assert rel(o["grad_ypr"], g["grad_ypr_f64"]) <= 1e-11
    o32 = oracle.sampling_loss(s["xyz"], s["rgb"], s["img"], g["trans"], g["rot"], dtype=np.float32)
    assert rel(o32["loss"], g["loss_list_f32"]) <= 2e-6


# G5 ------------------------------------------------------------------------------------------
def _teacher(grads_by_iter, losses_by_iter):
    """loss_grad that replays the reference's recorded losses/gradients (teacher forcing)."""
    state = {"it": 0}

    def fn(trans, rot):
        it = state["it"]
        state["it"] += 1
        g = grads_by_iter[it]            # (B, 6) in Adam order [t(3), yaw, roll, pitch]
        gt = g[:, :3]
        gr = np.stack([g[:, 3], g[:, 5], g[:, 4]], 1)   # -> yaw, pitch, roll
        return losses_by_iter[it].astype(np.float32), gt.astype(np.float32), gr.astype(np.float32)
    return fn


@pytest.mark.parametrize("sp", [0, 1])
def test_omniloc_trajectory_teacher_forced(oracle, sp):
    """Adam + plateau + clamp restatement reproduces the reference's sequential trajectory step by step."""
    from oracle import gd
    g = load_golden("g5_trajectories.npz")
    cfg = Cfg(**json.loads(str(g["cfg"])))
    pre = "seq%d_" % sp
    trace = []
    it, ir = g["trans0"].copy(), g["rot0"].copy()
    res = gd.omniloc(g["img"], g["xyz"], g["rgb"], it, ir, sp, cfg,
                     loss_grad=_teacher(g[pre + "adam_grad"], g[pre + "fwd_loss"]), trace=trace)
    ref_after = g[pre + "adam_param_after"][:, 0]        # Adam order t, yaw, roll, pitch — BEFORE the clamp
    for k, tr in enumerate(trace):
        fwd = np.concatenate([g[pre + "fwd_trans"][k, 0], g[pre + "fwd_rot"][k, 0]])
        assert np.abs(tr["param"] - fwd).max() <= 2e-6, k
        assert abs(tr["lr_after"] - g[pre + "sched_lr"][k, 0]) <= 1e-15, k
        assert tr["num_bad"] == g[pre + "sched_num_bad"][k, 0], k
        assert tr["lr"] == g[pre + "adam_lr"][k, 0], k
    assert np.abs(res[0] - g[pre + "ret_t"]).max() <= 2e-6
    assert np.abs(res[1] - g[pre + "ret_R"]).max() <= 2e-6
    assert abs(res[2] - g[pre + "ret_loss"]) <= 1e-7
    assert np.abs(it - g[pre + "input_trans_after"]).max() <= 2e-6      # caller's rows mutated like the reference
    assert np.abs(ir - g[pre + "input_rot_after"]).max() <= 2e-6
    assert ref_after.shape == (100, 6)


@pytest.mark.parametrize("tag", ["bat_", "bat1_", "bat2_"])
def test_omniloc_batch_trajectory_teacher_forced(oracle, tag):
    """... and the batch path incl. the one-iteration clamp lag and the pre-clamp return value."""
    from oracle import gd
    g = load_golden("g5_trajectories.npz")
    d = json.loads(str(g["cfg"]))
    d["num_iter"] = g[tag + "fwd_loss"].shape[0]
    cfg = Cfg(**d)
    trace = []
    it, ir = g["trans0"].copy(), g["rot0"].copy()
    res = gd.omniloc_batch(g["img"], g["xyz"], g["rgb"], it, ir, cfg,
                           loss_grad=_teacher(g[tag + "adam_grad"], g[tag + "fwd_loss"]), trace=trace)
    for k, tr in enumerate(trace):
        assert np.abs(tr["fwd"][:, :3] - g[tag + "fwd_trans"][k]).max() <= 2e-6, k
        assert np.abs(tr["fwd"][:, 3:] - g[tag + "fwd_rot"][k]).max() <= 2e-6, k
        # the leaf Adam sees (clamped) — Adam order [t, yaw, roll, pitch]
        pb = g[tag + "adam_param_before"][k]
        leaf_ref = np.stack([pb[:, 0], pb[:, 1], pb[:, 2], pb[:, 3], pb[:, 5], pb[:, 4]], 1)
        assert np.abs(tr["leaf"] - leaf_ref).max() <= 2e-6, k
        assert np.abs(tr["lr_after"] - g[tag + "sched_lr"][k]).max() <= 1e-15, k
        assert np.array_equal(tr["num_bad"], g[tag + "sched_num_bad"][k]), k
    assert np.abs(res[0] - g[tag + "ret_t"]).max() <= 2e-6
    assert np.abs(res[1] - g[tag + "ret_R"]).max() <= 2e-6
    assert abs(res[2] - g[tag + "ret_loss"]) <= 1e-7
    assert np.abs(it - g[tag + "input_trans_after"]).max() <= 2e-6
    assert np.abs(ir - g[tag + "input_rot_after"]).max() <= 2e-6


def test_batch_clamp_lag_is_pinned():
    """The golden itself shows the quirk: start x=4.4 is outside the box (x_max = 4.0, the wall); after one
    iteration the batch path forwards the unclamped 4.5 while the leaf Adam updates holds the clamped 4.0."""
    g = load_golden("g5_trajectories.npz")
    assert g["bat2_fwd_trans"][1, 1, 0] > g["bat2_adam_param_before"][1, 1, 0] + 1e-3   # forward saw unclamped x
    assert g["bat1_input_trans_after"][1, 0] == 4.0                                      # leaf was clamped
    assert g["seq1_fwd_trans"][1, 0, 0] == 4.0                                           # sequential path: no lag


def test_omniloc_free_running_first_iterations(oracle):
    """Free-running (oracle loss+grad, not teacher-forced): parity holds for the first iterations (SURVEY §8c:
    the trajectory is chaotic; <=1e-4 is only achievable for 2-3 iterations)."""
    from oracle import gd
    g = load_golden("g5_trajectories.npz")
    d = json.loads(str(g["cfg"]))
    d["num_iter"] = 3
    trace = []
    gd.omniloc(g["img"], g["xyz"], g["rgb"], g["trans0"].copy(), g["rot0"].copy(), 0, Cfg(**d), trace=trace)
    for k, tr in enumerate(trace):
        fwd = np.concatenate([g["seq0_fwd_trans"][k, 0], g["seq0_fwd_rot"][k, 0]])
        assert np.abs(tr["param"] - fwd).max() <= 1e-4, k
        assert abs(tr["loss"] - g["seq0_fwd_loss"][k, 0]) <= 1e-5, k


# G6 ------------------------------------------------------------------------------------------
def test_quantile(oracle):
    g = load_golden("g6_quantile.npz")
    for n in (1, 2, 19, 20, 21, 1000, 1001, 4096):
        for q in (0.05, 0.1, 0.25):
            lo, hi = oracle.quantile(g["x_%d" % n], q)
            ref = g["q_%d_%g" % (n, q)]
            assert lo == ref[0] and hi == ref[1], (n, q)


# G7 ------------------------------------------------------------------------------------------
def test_trim_input_loss(oracle):
    from oracle import gd
    g = load_golden("g7_trim_input_loss.npz")
    tt, tr, table = gd.trim_input_loss(g["img"], g["xyz"], g["rgb"], g["trans"], g["rot"], 7)
    assert rel(table, g["loss_table"]) <= 2e-6
    assert np.array_equal(tt, g["trimmed_trans"])
    assert np.array_equal(tr, g["trimmed_rot"])


# G8 ------------------------------------------------------------------------------------------
def test_make_pano(oracle):
    """The reference's per-pixel winner inside one index_put_ pass is undefined (duplicate indices); what IS
    defined is which pass paints a pixel last and which points that pass writes there.  So: every reference
    pixel must carry the colour of one of those candidates, empty pixels must agree, and the oracle's own
    choice (the nearest candidate) must be one of them too."""
    g = load_golden("g8_make_pano.npz")
    H, W = [int(v) for v in g["resolution"]]
    img, owner, contested = oracle.make_pano(g["xyz_cam"], g["rgb"], (H, W), return_aux=True)
    cands = oracle.make_pano_candidates(g["xyz_cam"], (H, W))
    ref = g["pano_f32"].reshape(-1, 3)
    rgb255 = g["rgb"] * np.float32(255)
    d = np.linalg.norm(g["xyz_cam"].astype(np.float64), axis=1)
    n_multi = 0
    for k, c in enumerate(cands):
        if not c:
            assert (ref[k] == 0).all() and owner.ravel()[k] == -1
            continue
        assert any(np.array_equal(rgb255[i], ref[k]) for i in c), k
        assert owner.ravel()[k] in c
        assert d[owner.ravel()[k]] <= min(d[i] for i in c) + 1e-6          # oracle = nearest candidate
        n_multi += len(c) > 1
    assert n_multi > 100                                                       # the ambiguity is exercised
    agree = (np.abs(img.reshape(-1, 3) - ref).max(-1) <= 1e-4).mean()
    assert agree > 0.95                                                        # measured 0.974
    # uint8 conversion: astype(uint8) truncation (utils.py:203)
    same = np.abs(img - g["pano_f32"]).max(-1) <= 1e-4
    assert np.array_equal(img.astype(np.uint8)[same], g["pano_u8"][same])


def test_scatter_min_is_consistent_with_make_pano_centre_pass(oracle):
    """scatter-min has no reference call site (parity unpinned); check it against the make_pano owner where the
    centre pass decides: the nearest point of a pixel owns it."""
    g = load_golden("g8_make_pano.npz")
    H, W = [int(v) for v in g["resolution"]]
    _, owner, contested = oracle.make_pano(g["xyz_cam"], g["rgb"], (H, W), return_aux=True)
    zmin, arg = oracle.scatter_min_depth(g["xyz_cam"], (H, W))
    n = len(g["xyz_cam"])
    filled = arg.reshape(H, W) < n
    same = (arg.reshape(H, W) == owner) | contested.reshape(H, W) | ~filled
    assert same.all()
    d = np.linalg.norm(g["xyz_cam"].astype(np.float64), axis=1)
    assert np.allclose(zmin[filled.ravel()], d[arg[filled.ravel()]], rtol=1e-6)
    assert (zmin[~filled.ravel()] == 0).all()


def test_scatter_min_pinned_to_torch_scatter_reduce(oracle):
    """The reference imports torch_scatter.scatter_min (utils.py:6) and never calls it, so the op's semantics here are
    build-defined on make_pano's pixel / depth definitions (utils.py:152-165).  They are pinned to a THIRD-PARTY
    definition of the same reduction: torch.Tensor.scatter_reduce_(0, pix, depth, 'amin',
    include_self=False) on the CPU — minimum depth per pixel, untouched pixels left at their fill value — plus
    torch_scatter's documented conventions for what scatter_reduce has no notion of: out = 0 and arg = N where nothing
    landed; arg = an index attaining the minimum."""
    import torch
    g = load_golden("g8_make_pano.npz")
    H, W = [int(v) for v in g["resolution"]]
    cam = g["xyz_cam"]
    n = len(cam)
    for xyz in (cam, np.concatenate([cam, cam[:500] * np.float32(1.5), cam[:300]])):      # second: exact depth ties + occlusion
        n = len(xyz)
        zmin, arg = oracle.scatter_min_depth(xyz, (H, W))
        row, col = oracle.pano_pixels(xyz, (H, W))
        pix = torch.from_numpy(row.astype(np.int64) * W + col.astype(np.int64))
        # depth exactly as the oracle / make_pano define it: ||p|| in fp32 (utils.py:152, torch.norm)
        depth = torch.linalg.vector_norm(torch.from_numpy(xyz), dim=1)
        ref = torch.full((H * W,), float("inf")).scatter_reduce_(0, pix, depth, "amin", include_self=False)
        hit = torch.zeros(H * W, dtype=torch.bool).index_fill_(0, pix, True).numpy()
        ref = ref.numpy()
        # (torch.norm vs sqrtf(x*x+y*y+z*z): <= 1 ulp apart)
        assert np.allclose(zmin[hit], ref[hit], rtol=2.5e-7, atol=0)
        assert (zmin[~hit] == 0).all() and (arg[~hit] == n).all()                      # torch_scatter: empty -> 0 / N
        assert (arg[hit] < n).all()
        a = arg[hit]
        assert np.array_equal(row[a].astype(np.int64) * W + col[a], np.nonzero(hit)[0])   # the winner really is in that pixel
        assert np.allclose(depth.numpy()[a], zmin[hit], rtol=2.5e-7, atol=0)           # and attains the minimum


# G11 -----------------------------------------------------------------------------------------
def test_end_to_end_within_reference_self_noise(oracle):
    from oracle import gd
    from piccolo_amd import synth
    g = load_golden("g11_end_to_end.npz")
    N, H, W, seed = int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed"])
    xyz, rgb = synth.box_room(N, seed)
    cam = synth.transform_cloud(xyz, g["t_gt"], g["ypr_gt"])
    img_u8 = oracle.make_pano_u8(cam, rgb, (H, W))
    assert (img_u8 != g["img_u8"]).any(axis=-1).mean() < 0.03      # same panorama up to index_put_ duplicate-index ambiguity (measured 1.0 %)
    img = g["img_u8"].astype(np.float32) / 255.0
    cfg = Cfg(lr=0.1, num_iter=100, patience=5, factor=0.8, out_of_room_quantile=0.05)
    res = gd.omniloc(img, xyz, rgb, g["trans0"].copy(), g["rot0"].copy(), 0, cfg)
    t_err, r_err = synth.pose_errors(res[0], res[1], g["t_gt"], synth.rot_from_ypr_np(g["ypr_gt"]))
    band_t = np.concatenate([g["self_noise"][:, 0], [float(g["t_err"])]])
    band_r = np.concatenate([g["self_noise"][:, 1], [float(g["r_err"])]])
    # inside the band spanned by the reference's own permuted-order reruns, with 50 % slack on its width
    wt, wr = band_t.max() - band_t.min(), band_r.max() - band_r.min()
    assert band_t.min() - 0.5 * wt - 2e-3 <= t_err <= band_t.max() + 0.5 * wt + 2e-3, (t_err, band_t)
    assert band_r.min() - 0.5 * wr - 0.05 <= r_err <= band_r.max() + 0.5 * wr + 0.05, (r_err, band_r)


# G18 -----------------------------------------------------------------------------------------
def g18_scene(oracle, g, s):
    """Scene s of G18, regenerated from its seed (the panorama comes from the deterministic oracle renderer; its checksum
    is in the fixture)."""
    from piccolo_amd import synth
    N, H, W, seed = int(g["N"]), int(g["H"]), int(g["W"]), int(g["seed0"]) + s
    xyz, rgb = synth.box_room(N, seed)
    t_gt, ypr_gt = synth.gt_pose(seed)
    img_u8 = oracle.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W))
    assert int(img_u8.astype(np.int64).sum()) == int(g["img_sum"][s]), "G18 panorama %d differs from the generator's" % s
    trans, rot = synth.start_poses(t_gt, ypr_gt, 4, seed=seed)
    return xyz, rgb, img_u8.astype(np.float32) / 255.0, trans, rot, t_gt, synth.rot_from_ypr_np(ypr_gt)


def g18_compare(rows, ref, record):
    """rows (S, 15) of this implementation vs ref (S, 2, 15) = the reference's run and its rerun with permuted points
    (columns t(3) R(9) loss t_err r_err).  The implementation must be as close to the reference as the reference is to
    itself: the distance distributions are compared at their median and maximum."""
    self_t, self_r = np.abs(ref[:, 0, 13] - ref[:, 1, 13]), np.abs(ref[:, 0, 14] - ref[:, 1, 14])
    self_p = np.abs(ref[:, 0, :3] - ref[:, 1, :3]).max(1)
    d_t, d_r = np.abs(rows[:, 13] - ref[:, 0, 13]), np.abs(rows[:, 14] - ref[:, 0, 14])
    d_p = np.abs(rows[:, :3] - ref[:, 0, :3]).max(1)
    # (with only 8 scenes the median of the self-noise is itself noisy: 2e-4 for the batch runs against 6e-4 over 32 scenes)
    floor_t = 2e-4 if len(rows) >= 16 else 6e-4
    record("t-err distance to the reference, median over seeds (m)", np.median(d_t), 2.5 * np.median(self_t) + floor_t, np.median(self_t))
    record("R-err distance to the reference, median over seeds (deg)", np.median(d_r), 2.5 * np.median(self_r) + 5e-3, np.median(self_r))
    record("recovered translation vs the reference's, median over seeds (m)", np.median(d_p), 2.5 * np.median(self_p) + 2e-4, np.median(self_p))
    record("t-err distance to the reference, worst seed (m)", d_t.max(), 2.5 * self_t.max(), self_t.max())
    record("R-err distance to the reference, worst seed (deg)", d_r.max(), 2.5 * self_r.max(), self_r.max())
    # Medians over the scenes: the reference's two runs differ in their medians by an amount that is itself one draw of a
    # noisy statistic (seq: 1.2e-4 m observed, while a paired bootstrap over the 32 scenes gives that difference a standard
    # deviation of 5.8e-4 m).  Bound = the observed difference + 3 bootstrap standard deviations of it (seeded resampling of the
    # scenes, both runs resampled together) — round 2's 2.5 x observed + 5e-4 passed at 76-86 % on one box by luck of the draw.
    rng = np.random.default_rng(0)
    idx = rng.integers(0, len(ref), size=(4000, len(ref)))
    for col, what, unit in ((13, "t-err", "m"), (14, "R-err", "deg")):
        a, b = ref[:, 0, col], ref[:, 1, col]
        spread = (np.median(a[idx], 1) - np.median(b[idx], 1)).std()
        record("median %s over seeds vs the reference's median (%s)" % (what, unit), abs(np.median(rows[:, col]) - np.median(a)),
               abs(np.median(a) - np.median(b)) + 3.0 * spread, spread)


def test_oracle_end_to_end_32_seeds_within_reference_self_noise(oracle, parity):
    """The oracle's restatement of omniloc (loss + gradient in C, Adam / plateau / clamp in oracle/gd.py), free-running
    for 100 iterations on the 32 scenes of G18, lands as close to the reference as the reference lands to itself."""
    from oracle import gd
    from piccolo_amd import synth
    g = load_golden("g18_end_to_end_seeds.npz")
    cfg = Cfg(lr=0.1, num_iter=100, patience=5, factor=0.8, out_of_room_quantile=0.05, num_input=4)
    rows = []
    for s in range(g["seq"].shape[0]):
        xyz, rgb, img, trans, rot, t_gt, R_gt = g18_scene(oracle, g, s)
        r = gd.omniloc(img, xyz, rgb, trans.copy(), rot.copy(), 0, cfg)
        t, R = r[0].reshape(3), r[1]
        rows.append(np.concatenate([t, R.reshape(-1), [float(r[2])], synth.pose_errors(t, R, t_gt, R_gt)]))
    g18_compare(np.array(rows), g["seq"], parity)


# G21 -----------------------------------------------------------------------------------------
def g21_scene(oracle, g):
    """The scene of G21 = the bench's cfg-2 scene (box_room(1M, seed 0), ground truth and 32 starting poses of image 0), the
    panorama from the deterministic oracle renderer; checked against the fixture's checksums."""
    from piccolo_amd import synth
    N, H, W, B = int(g["N"]), int(g["H"]), int(g["W"]), int(g["B"])
    xyz, rgb = synth.box_room(N, seed=0)
    t_gt, ypr_gt = synth.gt_pose(0)
    img_u8 = oracle.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W))
    assert int(img_u8.astype(np.int64).sum()) == int(g["img_sum"]), "G21 panorama differs from the generator's"
    assert float(xyz.astype(np.float64).sum()) == float(g["xyz_sum"]) and float(rgb.astype(np.float64).sum()) == float(g["rgb_sum"])
    trans, rot = synth.start_poses(t_gt, ypr_gt, B, seed=0)
    assert np.array_equal(trans, g["trans"]) and np.array_equal(rot, g["rot"])
    return xyz, rgb, img_u8.astype(np.float32) / 255.0, trans, rot


def test_oracle_at_full_cfg2_size_equals_the_reference(oracle, parity):
    """G21: the reference's own BatchSamplingLoss + autograd at BASELINE config 2's full size (1M points, 2048x1024, 32 poses),
    fp32 and fp64.  The fp64 oracle must reproduce the reference's fp64 numbers to rounding; the fp32 oracle's distance from
    fp64 is the reference's own fp32 distance (both ~1.2e-3 on the gradient: texel-cell flips of ~100 of the 1M points)."""
    g = load_golden("g21_full_size_batch_loss.npz")
    xyz, rgb, img, trans, rot = g21_scene(oracle, g)
    r64 = oracle.sampling_loss(xyz, rgb, img, trans, rot, dtype=np.float64)
    parity("G21 oracle fp64 vs reference fp64: loss_list", rel(r64["loss"], g["loss_list_f64"]), 1e-12)
    parity("G21 oracle fp64 vs reference fp64: grad_t", rel(r64["grad_t"], g["grad_t_f64"]), 1e-11)
    parity("G21 oracle fp64 vs reference fp64: grad_ypr", rel(r64["grad_ypr"], g["grad_ypr_f64"]), 1e-11)
    r32 = oracle.sampling_loss(xyz, rgb, img, trans, rot, dtype=np.float32)
    for key, gk in (("loss", "loss_list"), ("grad_t", "grad_t"), ("grad_ypr", "grad_ypr")):
        gap_ref = rel(g[gk + "_f32"], g[gk + "_f64"])
        parity("G21 fp32 oracle vs reference fp64: %s (yardstick: the reference's fp32)" % key, rel(r32[key], g[gk + "_f64"]),
               2 * gap_ref, gap_ref)


# G22 -----------------------------------------------------------------------------------------
def g22_scene(oracle, g, s):
    """Scene s of G22 (the reference's SHIPPED shape: 166 667 points, 2048x1024, 6 candidates), regenerated from its seed."""
    from piccolo_amd import synth
    N, H, W, B, seed = int(g["N"]), int(g["H"]), int(g["W"]), int(g["B"]), int(g["seed0"]) + s
    xyz, rgb = synth.box_room(N, seed)
    t_gt, ypr_gt = synth.gt_pose(seed)
    img_u8 = oracle.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W))
    assert int(img_u8.astype(np.int64).sum()) == int(g["img_sum"][s]), "G22 panorama %d differs from the generator's" % s
    trans, rot = synth.start_poses(t_gt, ypr_gt, B, seed=seed)
    return xyz, rgb, img_u8.astype(np.float32) / 255.0, trans, rot, t_gt, synth.rot_from_ypr_np(ypr_gt)


def g22_compare(rows, ref, record):
    """rows (S, 15) of this implementation vs ref (S, 2, 15): the reference's omniloc_batch and its permuted-order rerun.  Four
    scenes: every scene is held to the reference's OWN run-to-run distance (worst scene of the reference, x 2.5, + a floor)."""
    self_t, self_r = np.abs(ref[:, 0, 13] - ref[:, 1, 13]), np.abs(ref[:, 0, 14] - ref[:, 1, 14])
    self_p = np.abs(ref[:, 0, :3] - ref[:, 1, :3]).max(1)
    d_t, d_r = np.abs(rows[:, 13] - ref[:, 0, 13]), np.abs(rows[:, 14] - ref[:, 0, 14])
    d_p = np.abs(rows[:, :3] - ref[:, 0, :3]).max(1)
    record("t-err distance to the reference, worst of 4 scenes (m)", d_t.max(), 2.5 * self_t.max() + 5e-4, self_t.max())
    record("R-err distance to the reference, worst of 4 scenes (deg)", d_r.max(), 2.5 * self_r.max() + 1e-2, self_r.max())
    record("recovered translation vs the reference's, worst of 4 scenes (m)", d_p.max(), 2.5 * self_p.max() + 5e-4, self_p.max())


def test_oracle_at_the_shipped_shape_within_reference_self_noise(oracle, parity):
    """G22: the reference's omniloc_batch at the sizes of its shipped config, 100 iterations, 4 scenes x (original, permuted).
    The oracle's restatement, free-running, lands as close to the reference as the reference lands to itself."""
    from oracle import gd
    from piccolo_amd import synth
    g = load_golden("g22_shipped_shape.npz")
    cfg = Cfg(lr=0.1, num_iter=100, patience=5, factor=0.8, out_of_room_quantile=0.05, num_input=int(g["B"]))
    rows = []
    for s in range(g["batch"].shape[0]):
        xyz, rgb, img, trans, rot, t_gt, R_gt = g22_scene(oracle, g, s)
        r = gd.omniloc_batch(img, xyz, rgb, trans.copy(), rot.copy(), cfg)
        t, R = r[0].reshape(3), r[1]
        rows.append(np.concatenate([t, R.reshape(-1), [float(r[2])], synth.pose_errors(t, R, t_gt, R_gt)]))
    g22_compare(np.array(rows), g["batch"], parity)


# G23 -----------------------------------------------------------------------------------------
G23_CONFIGS = ("stanford", "stanford_parallel", "omniscenes")


def g23_case(g, tag):
    """One config of G23: the cloud as the reference's loader would hand it over (subsampled by sample_rate), the image, and the
    reference's make_input intermediates."""
    import json
    num_input, num_intermediate, rate, n = [int(v) for v in g[tag + "_num"]]
    xyz, rgb = g["xyz"], g["rgb"]
    if rate > 1:
        k = int(len(xyz) / rate)
        xyz, rgb = np.ascontiguousarray(xyz[::rate][:k]), np.ascontiguousarray(rgb[::rate][:k])
    assert len(xyz) == n
    d = {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + "_") and k[len(tag) + 1:len(tag) + 5] in ("loss", "hist", "inpu")}
    if tag == "stanford":        # (stanford_parallel_* keys also start with "stanford_")
        d = {k: v for k, v in d.items() if not k.startswith("parallel_")}
    return xyz, rgb, g["img"], json.loads(str(g[tag + "_init_dict"])), num_input, num_intermediate, d


def got_of(table, n):
    return np.argsort(np.asarray(table).reshape(-1), kind="stable")[:n]


def g23_check_table(record, who, tag, table, ref, n_points, n_keep, got=None):
    """Loss table against the reference's.  All entries agree to fp32 rounding except where ONE point changed sides of the
    exact-zero mask (quarter-turn rotations put whole walls of the room on pixel boundaries): such an entry moves by ~0.6 / n.
    Then the survivors, as far as the reference's table decides them: every entry known to 10 x the rounding tolerance, plus
    the measured difference where a point flipped."""
    from parity_helpers import check_selection
    e = np.abs(np.asarray(table, np.float64) - ref).reshape(-1)
    flipped = e > 4e-7
    record("G23 %s: %s loss table vs the reference's, entries without a mask flip (abs)" % (tag, who), e[~flipped].max(), 4e-7)
    # (measured: oracle 0-2 of 1800 entries, device 9-10 — its yaw-shared projection rounds differently on the seam planes — each
    #  of them ONE point: the largest move x points is 0.13 ... 0.72, a single point's share of the mean)
    record("G23 %s: %s loss table, entries where a point changed its mask side or bilinear cell (of %d)" % (tag, who, e.size), flipped.sum(), 0.01 * e.size)
    record("G23 %s: %s loss table, largest such move x points" % (tag, who), (e.max() if flipped.any() else 0.0) * n_points, 2.0)
    got = got_of(table, n_keep) if got is None else got
    must, ranked = check_selection(got, ref, n_keep, 10 * 4e-7 + np.where(flipped, e, 0.0))
    assert must >= n_keep - 4 and ranked >= n_keep // 3, (tag, must, ranked)
    return must, ranked


def test_g23_fixture_is_self_consistent():
    """What the fixture holds is the reference's own composition: its survivors are its table's argsort through the
    `// len(rot)`, `% len(rot)` decode (utils.py:500-505), its final poses its scores' ranking (utils.py:583-586)."""
    g = load_golden("g23_make_input.npz")
    for tag in G23_CONFIGS:
        xyz, rgb, img, init, n_in, n_mid, d = g23_case(g, tag)
        K, Rn = d["loss_loss_table"].shape
        mi = d["loss_min_inds"]
        assert np.array_equal(d["loss_trimmed_trans"], d["loss_trans"][mi // Rn]) and np.array_equal(d["loss_trimmed_rot"], d["loss_rot"][mi % Rn])
        assert np.array_equal(np.sort(d["loss_loss_table"].reshape(-1)[mi]), np.sort(d["loss_loss_table"].reshape(-1))[:n_mid])
        assert np.array_equal(d["hist_trans"], d["loss_trimmed_trans"]) and np.array_equal(d["hist_rot"], d["loss_trimmed_rot"])
        hi = d["hist_min_inds"]
        assert np.array_equal(d["input_trans"], d["hist_trans"][hi]) and np.array_equal(d["input_rot"], d["hist_rot"][hi])
        assert np.array_equal(np.sort(d["hist_hist_intersect"][hi])[::-1], np.sort(d["hist_hist_intersect"])[::-1][:n_in])


def test_make_input_composed_matches_the_reference(oracle, parity):
    """G23: the oracle's composition of the initialisation stage (loss table -> survivors -> histogram scores -> final starting
    poses) against the reference's make_input run on the same scene, for its three shipped configs (utils.py:591-629)."""
    from oracle import gd, hist
    from parity_helpers import check_selection
    g = load_golden("g23_make_input.npz")
    for tag in G23_CONFIGS:
        xyz, rgb, img, init, n_in, n_mid, d = g23_case(g, tag)
        K, Rn = d["loss_loss_table"].shape
        tt, tr, table = gd.trim_input_loss(img, xyz, rgb, d["loss_trans"], d["loss_rot"], n_mid)
        g23_check_table(parity, "oracle", tag, table, d["loss_loss_table"], len(xyz), n_mid)
        assert np.array_equal(tt, d["loss_trans"][got_of(table, n_mid) // Rn]) and np.array_equal(tr, d["loss_rot"][got_of(table, n_mid) % Rn])  # the decode
        # second stage on the REFERENCE's survivors: scores up to the render's duplicate-index ambiguity (G12), selection as far
        # as the reference's scores decide it at that tolerance
        ft, fr, scores = hist.trim_input_hist_secondary(img, xyz, rgb, d["hist_trans"], d["hist_rot"], n_in, init["num_split_h"], init["num_split_w"])
        serr = np.abs(scores - d["hist_hist_intersect"]).max()
        self_noise = np.abs(d["hist_hist_intersect_permuted"] - d["hist_hist_intersect"]).max()       # the reference vs itself, points permuted
        parity("G23 %s: oracle histogram scores vs the reference's (%d renders, abs; yardstick: the reference's own rerun)" % (tag, n_mid),
               serr, 2.5 * self_noise + 1e-3, self_noise)
        sel = np.argsort(scores, kind="stable")[-n_in:][::-1]
        check_selection(sel, d["hist_hist_intersect"], n_in, np.abs(scores - d["hist_hist_intersect"]) + 1e-6, largest=True)
        assert np.array_equal(ft, d["hist_trans"][sel]) and np.array_equal(fr, d["hist_rot"][sel])


# G22b ----------------------------------------------------------------------------------------
def g22b_reference_rows(gb, s, k):
    """The reference's fp32 loss_list / autograd gradients of recorded iteration k of scene s (Adam's parameter order is
    [t, yaw, roll, pitch], omniloc.py:235-236) and its own fp64 evaluation at the same poses."""
    g = gb["adam_grad"][s, k]
    return (gb["fwd_loss"][s, k], g[:, :3], g[:, [3, 5, 4]]), (gb["loss_f64"][s, k], gb["grad_t_f64"][s, k], gb["grad_ypr_f64"][s, k])


def g22b_teacher_forced(record, who, evaluate, gb, s):
    """evaluate(trans, rot) -> (loss, grad_t, grad_ypr) at the reference's recorded forward poses of iterations 0-4, 10, 50, 99 of
    scene s: every evaluation within 2 x the reference's own fp32 distance from its fp64 value."""
    names = ("loss_list", "grad_t", "grad_ypr")
    # the yardstick is pooled over the scene's eight recorded iterations: an evaluation's fp32 error is a handful of discrete
    # events (a point changing its bilinear cell or its side of the zero mask: 1 / 166 667 of the loss scale each), so single
    # evaluations scatter around the scene's level (the reference's own: 1.2e-5 ... 1.3e-4 on the loss, 9e-4 ... 1.2e-2 on grad_t)
    gaps = np.array([[rel(a, b) for a, b in zip(*g22b_reference_rows(gb, s, k))] for k in range(len(gb["iters"]))])
    pooled = gaps.max(0)
    worst = np.zeros(3)
    for k, it in enumerate(gb["iters"]):
        out = evaluate(gb["fwd_trans"][s, k], gb["fwd_rot"][s, k])
        _, r64 = g22b_reference_rows(gb, s, k)
        for j, name in enumerate(names):
            err = rel(out[j], r64[j])
            record("G22b scene %d iteration %d: %s %s vs the reference's fp64 (yardstick: its fp32 run, worst of the scene's 8 evaluations)"
                   % (s, it, who, name), err, 2 * pooled[j] + 1e-6, pooled[j])
            worst[j] = max(worst[j], err / pooled[j])
    return worst


def g22b_free_running(record, who, loss_hist, fwd, gb, s):
    """Free-running first iterations at the shipped shape against the reference's recorded trajectory of scene s.
    loss_hist (>= 2, B): loss_list of iterations 0, 1, ...; fwd {1: (B, 6), 2: (B, 6)}: forward poses [t, yaw, pitch, roll] of
    iterations 1 and 2.  What can be asserted here is less than G5's 1e-4 over three iterations, and the fixture says why: at
    this shape the reference's OWN fp32 gradient is 0.4 - 1.2 % from its fp64 gradient, and Adam's first step is lr x sign(g)
    whatever the magnitude — a component below that noise can step the other way in any fp32 evaluation (0.2 apart after one
    iteration), and the second step divides two noisy moments.  So: iteration 0's loss_list; iteration 1's pose bit-near for
    every parameter whose gradient sign the reference's fp32 and fp64 runs agree on by a margin of 3 x its fp32 error; the
    candidates all of whose parameters are such: loss of iteration 1, and the pose of iteration 2 within lr x 4 x the
    reference's relative gradient error."""
    lr = 0.1
    g32 = np.concatenate(g22b_reference_rows(gb, s, 0)[0][1:], 1)                  # (B, 6) [t, yaw, pitch, roll]
    g64 = np.concatenate(g22b_reference_rows(gb, s, 0)[1][1:], 1)
    noise = np.concatenate([np.full(3, np.abs(g32[:, :3] - g64[:, :3]).max()), np.full(3, np.abs(g32[:, 3:] - g64[:, 3:]).max())])
    sure = (np.abs(g64) > 3 * noise[None, :]) & (np.sign(g32) == np.sign(g64))     # (B, 6)
    ref1 = np.concatenate([gb["fwd_trans"][s, 1], gb["fwd_rot"][s, 1]], 1)
    ref2 = np.concatenate([gb["fwd_trans"][s, 2], gb["fwd_rot"][s, 2]], 1)
    record("G22b scene %d: %s free-running, loss_list of iteration 0 (abs)" % (s, who), np.abs(loss_hist[0] - gb["fwd_loss"][s, 0]).max(), 2e-5)
    assert sure.sum() >= 24, (s, sure.sum())                                        # the conditioning leaves most of the 36 parameters
    record("G22b scene %d: %s free-running, pose of iteration 1 where the reference's gradient sign is certain (%d of 36 parameters, abs)"
           % (s, who, sure.sum()), np.abs(fwd[1] - ref1)[sure].max(), 1e-6)
    cand = sure.all(1)
    if cand.any():
        record("G22b scene %d: %s free-running, loss_list of iteration 1, candidates with all six signs certain (%d of 6, abs)" % (s, who, cand.sum()),
               np.abs(loss_hist[1] - gb["fwd_loss"][s, 1])[cand].max(), 2e-5)
        gap = max(rel(g32[:, :3], g64[:, :3]), rel(g32[:, 3:], g64[:, 3:]))
        record("G22b scene %d: %s free-running, pose of iteration 2, same candidates (abs; yardstick: lr x the reference's relative gradient error)"
               % (s, who), np.abs(fwd[2] - ref2)[cand].max(), 4 * lr * gap + 1e-5, lr * gap)
    return int(sure.sum()), int(cand.sum())


def g22b_final_candidates(record, who, final_param, final_loss, gb):
    """All six candidates at the end of the 100 iterations (S, 6, 6) / (S, 6) against the reference's, per candidate; yardstick:
    the reference's own rerun with the points permuted (its candidates are unconverged at this shape and move by centimetres)."""
    rp, rl = gb["final_param"], gb["final_loss"]
    self_t, self_a = np.abs(rp[:, 0, :, :3] - rp[:, 1, :, :3]).max(-1), np.abs(rp[:, 0, :, 3:] - rp[:, 1, :, 3:]).max(-1)
    d_t, d_a = np.abs(final_param[:, :, :3] - rp[:, 0, :, :3]).max(-1), np.abs(final_param[:, :, 3:] - rp[:, 0, :, 3:]).max(-1)
    self_l, d_l = np.abs(rl[:, 0] - rl[:, 1]), np.abs(final_loss - rl[:, 0])
    for name, d, sn, floor in (("translation (m)", d_t, self_t, 5e-4), ("yaw/pitch/roll (rad)", d_a, self_a, 1e-4), ("last loss (abs)", d_l, self_l, 1e-4)):
        record("G22b final candidates, %s %s: median over the 24 candidates of the distance to the reference's" % (who, name),
               np.median(d), 2.5 * np.median(sn) + floor, np.median(sn))
        record("G22b final candidates, %s %s: worst candidate" % (who, name), d.max(), 2.5 * sn.max() + floor, sn.max())


def test_oracle_per_evaluation_at_the_shipped_shape(oracle, parity):
    """G22b: the reference's omniloc_batch at its SHIPPED shape (166 667 points, 2048x1024, 6 candidates) per evaluation: the
    forward poses, loss_list and autograd gradients of iterations 0-4, 10, 50, 99 and its own fp64 evaluation at those poses.
    The fp64 oracle reproduces the fp64 values; the fp32 oracle is as far from them as the reference's fp32 run; the oracle's
    free-running loop follows the recorded trajectory for the first iterations (omniloc.py:249-269, 311-356)."""
    from oracle import gd
    g, gb = load_golden("g22_shipped_shape.npz"), load_golden("g22b_shipped_iterations.npz")
    assert list(gb["iters"]) == [0, 1, 2, 3, 4, 10, 50, 99]
    cfg = Cfg(lr=0.1, num_iter=3, patience=5, factor=0.8, out_of_room_quantile=0.05, num_input=int(g["B"]))
    for s in range(2):                                        # two of the four scenes on the CPU (the GPU test takes all four)
        xyz, rgb, img, trans, rot, t_gt, R_gt = g22_scene(oracle, g, s)
        assert np.array_equal(gb["fwd_trans"][s, 0], trans) and np.array_equal(gb["fwd_rot"][s, 0], rot)     # iteration 0 = the starts

        def ev(dtype):
            def f(t, r):
                o = oracle.sampling_loss(xyz, rgb, img, t, r, dtype=dtype)
                return o["loss"], o["grad_t"], o["grad_ypr"]
            return f

        for k in (0, 5, 7):
            o64 = ev(np.float64)(gb["fwd_trans"][s, k], gb["fwd_rot"][s, k])
            _, r64 = g22b_reference_rows(gb, s, k)
            for j, name in enumerate(("loss_list", "grad_t", "grad_ypr")):
                parity("G22b scene %d iteration %d: fp64 oracle %s vs the reference's fp64" % (s, gb["iters"][k], name), rel(o64[j], r64[j]), 1e-10)
        g22b_teacher_forced(parity, "fp32 oracle", ev(np.float32), gb, s)
        trace = []
        gd.omniloc_batch(img, xyz, rgb, trans.copy(), rot.copy(), cfg, trace=trace)
        g22b_free_running(parity, "oracle", np.stack([t["loss"] for t in trace]), {k: trace[k]["fwd"] for k in (1, 2)}, gb, s)


# G12 -----------------------------------------------------------------------------------------
def test_trim_input_hist_secondary(oracle):
    """Oracle vs the reference's block-histogram scores.  The rendered panoramas differ from the reference's in the
    ~2-3 % of pixels where its duplicate-index index_put_ picked another point of the same pass (see test_make_pano),
    which moves a normalised-histogram intersection by a few 1e-3; the ranking of the candidates is unaffected."""
    from oracle import hist
    g = load_golden("g12_trim_input_hist.npz")
    nh, nw = [int(v) for v in g["num_split"]]
    scores, inter = hist.hist_scores(g["img"], g["xyz"], g["rgb"], g["trans"], g["rot"], nh, nw)
    assert np.abs(scores - g["scores"]).max() <= 1e-2
    assert np.abs(inter - g["inter"]).max() <= 5e-2      # single blocks (few hundred pixels each); measured 3.3e-2
    assert np.array_equal(np.argsort(scores)[::-1][:4], np.argsort(g["scores"])[::-1][:4])
    tt, tr, _ = hist.trim_input_hist_secondary(g["img"], g["xyz"], g["rgb"], g["trans"], g["rot"], 4, nh, nw)
    assert np.array_equal(tt, g["selected_trans"]) and np.array_equal(tr, g["selected_rot"])
    assert np.argmax(scores) == 0                         # candidate 0 is the ground-truth pose


# G19 -----------------------------------------------------------------------------------------
def test_trim_input_hist_empty_blocks_carry_over(oracle):
    """G19: a partial cloud leaves blocks of the candidates' renders empty.  The reference breaks out of the block row and
    the rest of the row keeps EARLIER candidates' intersections (one slot vector for all candidates, utils.py:539,568-571).
    The fixture holds the slot vector after every candidate, read out of the running reference function."""
    from oracle import hist
    g = load_golden("g19_trim_hist_empty_blocks.npz")
    nh, nw = [int(v) for v in g["num_split"]]
    scores, inter = hist.hist_scores(g["img"], g["xyz"], g["rgb"], g["trans"], g["rot"], nh, nw)
    ref = g["split"]
    # the fixture really exercises the quirk: some candidate carries a non-zero value it did not compute
    assert (ref[6, nw + 1:nw + 3] == ref[5, nw + 1:nw + 3]).all() and ref[6, nw] == 0 and ref[5, nw + 1] > 0
    # same pattern of zero / carried / computed slots, values up to the render's duplicate-index ambiguity (see G12)
    assert np.array_equal(inter == 0, ref == 0)
    assert np.abs(inter - ref).max() <= 5e-2
    carried = ref[6, nw + 1:nw + 3]
    assert np.array_equal(inter[6, nw + 1:nw + 3], inter[5, nw + 1:nw + 3]) and np.abs(inter[6, nw + 1:nw + 3] - carried).max() <= 5e-2
    assert np.abs(scores - g["scores"]).max() <= 1e-2
    tt, tr, _ = hist.trim_input_hist_secondary(g["img"], g["xyz"], g["rgb"], g["trans"], g["rot"], 4, nh, nw)
    assert np.array_equal(tt, g["ranked_trans"][:4]) and np.array_equal(tr, g["ranked_rot"][:4])


# G20 -----------------------------------------------------------------------------------------
def g20_regular(g):
    """Points of G20 where cloud2idx's Jacobian is regular: at a = x + 1e-6 ~ 0 with y = 0 (or on the vertical axis) the
    reference's own fp32 and fp64 gradients differ by ten orders of magnitude (4.6e5 vs 2.7e14): nothing to pin there."""
    x = g["xyz"].astype(np.float64)
    s1 = (x[:, 0] + 1e-6) ** 2 + x[:, 1] ** 2
    s2 = x[:, 0] ** 2 + x[:, 1] ** 2 + (x[:, 2] + 1e-6) ** 2
    return (s1 > 1e-6) & (s2 > 1e-6)


def test_standalone_backward_matches_reference_autograd(oracle):
    """G20: the oracle's backward of cloud2idx and sample_from_img against the reference's autograd, fp64 to rounding, fp32
    as close to fp64 as the reference's own fp32 run."""
    g = load_golden("g20_standalone_backward.npz")
    ok = g20_regular(g)
    assert ok.sum() > 900
    ref = g["grad_xyz_f64"]
    scale = np.maximum(np.abs(ref), 1.0)
    assert (np.abs(oracle.cloud2idx_backward(g["xyz"], g["grad_coord_in"], np.float64) - ref) / scale)[ok].max() <= 1e-12
    gap = (np.abs(g["grad_xyz_f32"] - ref) / scale)[ok].max()
    assert (np.abs(oracle.cloud2idx_backward(g["xyz"], g["grad_coord_in"], np.float32) - ref) / scale)[ok].max() <= 2 * gap + 1e-6
    gc, gi = oracle.sample_from_img_backward(g["img"], g["coord"], g["grad_rgb_in"], np.float64)
    assert np.abs(gc - g["grad_coord_f64"]).max() <= 1e-12 and np.abs(gi - g["grad_img_f64"]).max() <= 1e-12
    assert np.array_equal(gc == 0, g["grad_coord_f64"] == 0)          # clipped coordinates get exactly no gradient
    gc32, gi32 = oracle.sample_from_img_backward(g["img"], g["coord"], g["grad_rgb_in"], np.float32)
    assert np.array_equal(gc32 == 0, g["grad_coord_f32"] == 0)
    assert np.abs(gc32 - g["grad_coord_f32"]).max() <= 2e-4 and np.abs(gi32 - g["grad_img_f32"]).max() <= 2e-5
