"""GPU tests of the callers either side of the hot path: candidate generation, the two trimming
stages of make_input, and the synthetic harness."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import Cfg, load_golden

pytestmark = pytest.mark.gpu

BASE = dict(max_yaw=2 * np.pi, min_yaw=0, max_pitch=2 * np.pi, min_pitch=0, max_roll=2 * np.pi, min_roll=0,
            z_prior=None, sample_rate_for_init=None, trans_init_mode="quantile",
            x_max=None, x_min=None, y_max=None, y_min=None, z_max=None, z_min=None, num_split_h=4, num_split_w=4)
STANFORD = dict(BASE, xy_only=False, num_trans=50, yaw_only=False, num_yaw=4, num_pitch=4, num_roll=4, dataset="Stanford2D-3D-S")
OMNI = dict(BASE, xy_only=True, num_trans=150, yaw_only=True, num_yaw=8, num_pitch=8, num_roll=8, dataset="OmniScenes", z_prior=1.5)


def _rows(a):
    a = np.asarray(a, np.float64)
    return a[np.lexsort(a.T[::-1])]


def test_candidate_generation_matches_reference():
    """G10: the reference's own generate_rot_points / generate_trans_points on the same cloud.  The rotation set is
    compared as a set (the reference orders it by iterating a Python set of strings)."""
    from piccolo_amd import utils
    g = load_golden("g10_candidates.npz")
    xyz = torch.from_numpy(g["xyz"]).cuda()
    for tag, d in (("stanford", STANFORD), ("omniscenes", OMNI)):
        rot = utils.generate_rot_points(d, device=xyz.device).cpu().numpy()
        tr = utils.generate_trans_points(xyz, d, device=xyz.device).cpu().numpy()
        assert rot.shape == g[tag + "_rot"].shape and tr.shape == g[tag + "_trans"].shape
        assert np.abs(_rows(rot) - _rows(g[tag + "_rot"])).max() <= 1e-6
        assert np.abs(tr - g[tag + "_trans"]).max() <= 1e-5


def test_make_input_finds_the_neighbourhood_of_the_true_pose(oracle):
    """make_input = candidate grid -> sampling-loss trim -> histogram trim.  On a synthetic room the best surviving
    candidate must be the grid pose nearest to the ground truth (coarse grid: ~1 m / 90 deg spacing)."""
    from piccolo_amd import synth, utils
    n, H, W = 50_000, 128, 256
    xyz, rgb = synth.box_room(n, 3)
    t_gt = np.array([0.9, -0.6, 0.0], np.float32)
    ypr_gt = np.array([np.pi / 2, 0.0, 0.0], np.float32)
    img = oracle.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W)).astype(np.float32) / 255
    X, C, I = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda(), torch.from_numpy(img).cuda()
    trans, rot = utils.make_input(I, X, C, 6, dict(STANFORD), "loss_histogram", 20)
    assert trans.shape == (6, 3) and rot.shape == (6, 3)
    R_gt = synth.rot_from_ypr_np(ypr_gt)
    errs = [synth.pose_errors(trans[i].cpu().numpy(), synth.rot_from_ypr_np(rot[i].cpu().numpy()), t_gt, R_gt) for i in range(6)]
    assert min(e[0] for e in errs) < 1.2 and min(e[1] for e in errs) < 5.0, errs
    with pytest.raises(UnboundLocalError):
        utils.make_input(I, X, C, 6, dict(STANFORD), "histogram", 20)


def test_utils_surface(oracle):
    from piccolo_amd import synth, utils
    g = load_golden("g6_quantile.npz")
    x = torch.from_numpy(g["x_1001"])                       # CPU tensor in, CPU tensors out
    lo, hi = utils.quantile(x, 0.05)
    assert lo.device.type == "cpu" and lo.item() == g["q_1001_0.05"][0] and hi.item() == g["q_1001_0.05"][1]
    xyz, rgb = synth.box_room(2000, 5)
    X = torch.from_numpy(xyz).cuda()
    inside = torch.tensor([[0.0], [0.0], [0.0]])
    outside = torch.tensor([[4.5], [0.0], [0.0]])
    assert utils.out_of_room(X, inside) is False and utils.out_of_room(X, outside) is True
    pano = utils.make_pano(X, torch.from_numpy(rgb).cuda(), resolution=(32, 64))
    assert pano.dtype == np.uint8 and pano.shape == (32, 64, 3)
    R = utils.rot_from_ypr(torch.tensor([0.3, -0.2, 0.1]))
    assert np.abs(R.numpy() - oracle.rot_from_ypr([0.3, -0.2, 0.1], np.float64)).max() <= 2e-7
    with pytest.raises(NotImplementedError):
        utils.sample_from_img(torch.zeros(4, 8, 3), torch.zeros(3, 2), padding="border")


def test_localize_synthetic_converges():
    from piccolo_amd.localize import localize_synthetic
    cfg = Cfg(num_points=100_000, pano_height=256, pano_width=512, num_images=3, num_input=8, parallel=True,
              lr=0.1, num_iter=100, patience=5, factor=0.8, out_of_room_quantile=0.05)
    table = localize_synthetic(cfg).cpu().numpy()
    assert table.shape == (3, 16)
    assert np.median(table[:, 13]) < 0.05 and np.median(table[:, 14]) < 1.0          # ~0.01 m / 0.3 deg expected
    seq = Cfg(**{**cfg.__dict__, "parallel": False, "num_images": 1, "num_input": 2})
    t2 = localize_synthetic(seq).cpu().numpy()
    assert t2[0, 13] < 0.1


def test_integration_md_stub_runs(oracle):
    """The ctypes stub printed in INTEGRATION.md is executable documentation: extract it, run it, check it."""
    import os
    import re
    from conftest import REPO
    from piccolo_amd import _lib, synth
    _lib.load()
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# piccolo_hip_stub\.py.*?)```", text, re.S).group(1)
    ns = {}
    cwd = os.getcwd()
    os.chdir(REPO)
    try:
        exec(compile(code, "piccolo_hip_stub.py", "exec"), ns)
    finally:
        os.chdir(cwd)
    g = load_golden("g3_sampling_loss.npz")
    dev = torch.device("cuda")
    fused = ns["FusedSamplingLoss"](torch.from_numpy(g["xyz"]).to(dev), torch.from_numpy(g["rgb"]).to(dev),
                                    torch.from_numpy(g["img"]).to(dev))
    out = fused(torch.from_numpy(g["trans"]).to(dev), torch.from_numpy(g["rot"]).to(dev)).cpu().numpy()
    assert np.abs(out[:, 0] - g["loss_f64"]).max() <= 2e-6
    assert np.abs(out[:, 2:5] - g["grad_t_f64"]).max() / np.abs(g["grad_t_f64"]).max() <= 1e-4


def test_omniloc_batch_images_equals_per_image_calls(oracle):
    from piccolo_amd import omniloc as po
    from piccolo_amd import synth
    n, H, W, B, I = 4096, 64, 128, 4, 3
    xyz, rgb = synth.box_room(n, 29)
    X, C = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
    cfg = Cfg(lr=0.1, num_iter=30, patience=5, factor=0.8, out_of_room_quantile=0.05, num_input=B)
    imgs, trs, ros = [], [], []
    for k in range(I):
        t_gt, ypr_gt = synth.gt_pose(60 + k)
        img = oracle.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W)).astype(np.float32) / 255
        imgs.append(torch.from_numpy(img).cuda())
        tr, ro = synth.start_poses(t_gt, ypr_gt, B, seed=60 + k)
        trs.append(torch.from_numpy(tr).cuda())
        ros.append(torch.from_numpy(ro).cuda())
    single = [po.omniloc_batch(imgs[k], X, C, trs[k].clone(), ros[k].clone(), cfg, {}) for k in range(I)]
    multi = po.omniloc_batch_images(imgs, X, C, [t.clone() for t in trs], [r.clone() for r in ros], cfg)
    for a, b in zip(single, multi):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    # one image that is not k/255 forces a common texel format (float4) for the whole launch: still one result per image,
    # the k/255 images within the lerp-rounding distance of their level-texel results
    # (compared after 5 iterations: later the two texel formats' trajectories drift apart chaotically, like any two fp32
    # evaluations of this loop)
    cfg5 = Cfg(lr=0.1, num_iter=5, patience=5, factor=0.8, out_of_room_quantile=0.05, num_input=B)
    mixed = [imgs[0], imgs[1] * 0.9, imgs[2]]
    out = po.omniloc_batch_images(mixed, X, C, [t.clone() for t in trs], [r.clone() for r in ros], cfg5)
    ref0 = po.omniloc_batch(imgs[0], X, C, trs[0].clone(), ros[0].clone(), cfg5, {})
    assert len(out) == I and all(torch.isfinite(o[2]) for o in out)
    assert abs(float(out[0][2]) - float(ref0[2])) <= 1e-4 * abs(float(ref0[2])) + 1e-6


def test_hist_trim_scores_vs_oracle_and_reference_golden(oracle):
    """The fused histogram-trim kernels (csrc/pcl_hist.hip) against the oracle restatement (same nearest-wins rendering:
    equal up to pixel-boundary flips) and against the reference's own scores (G12; its renders differ in the pixels
    where index_put_'s duplicate-index choice is undefined)."""
    from oracle import hist
    from piccolo_amd import ops, utils
    g = load_golden("g12_trim_input_hist.npz")
    nh, nw = [int(v) for v in g["num_split"]]
    dev = torch.device("cuda")
    I, X, C = [torch.from_numpy(g[k]).to(dev) for k in ("img", "xyz", "rgb")]
    tr, ro = torch.from_numpy(g["trans"]).to(dev), torch.from_numpy(g["rot"]).to(dev)
    cloud = ops.Cloud(X, C)
    scores = ops.hist_trim_scores(I, cloud, tr, ro, nh, nw, batch=4).cpu().numpy()     # 10 candidates in batches of 4, 4, 2
    ref, _ = hist.hist_scores(g["img"], g["xyz"], g["rgb"], g["trans"], g["rot"], nh, nw)
    assert np.abs(scores - ref).max() <= 2e-3, np.abs(scores - ref).max()
    assert np.abs(scores - g["scores"]).max() <= 1e-2
    assert np.array_equal(np.argsort(scores)[::-1][:4], np.argsort(g["scores"])[::-1][:4])
    tt, trr = utils.trim_input_hist_secondary(I, X, C, tr, ro, 4, nh, nw)
    assert np.array_equal(tt.cpu().numpy(), g["selected_trans"]) and np.array_equal(trr.cpu().numpy(), g["selected_rot"])
    # an all-black query image: every block is empty -> all scores 0 (no NaN)
    z = ops.hist_trim_scores(torch.zeros_like(I), cloud, tr, ro, nh, nw).cpu().numpy()
    assert (z == 0).all()


def test_hist_trim_empty_blocks_carry_over_like_the_reference(oracle):
    """G19: candidates whose render leaves a block empty.  The reference's score then includes what earlier candidates left
    in the rest of that block row (utils.py:539,568-571); the fixture holds the reference's per-candidate slot vectors.
    Scores and ranking must follow the reference, also when the candidates are rendered in several batches."""
    from oracle import hist
    from piccolo_amd import ops, utils
    g = load_golden("g19_trim_hist_empty_blocks.npz")
    nh, nw = [int(v) for v in g["num_split"]]
    dev = torch.device("cuda")
    I, X, C = [torch.from_numpy(g[k]).to(dev) for k in ("img", "xyz", "rgb")]
    tr, ro = torch.from_numpy(g["trans"]).to(dev), torch.from_numpy(g["rot"]).to(dev)
    cloud = ops.Cloud(X, C)
    ref, _ = hist.hist_scores(g["img"], g["xyz"], g["rgb"], g["trans"], g["rot"], nh, nw)
    for batch in (16, 5):
        scores = ops.hist_trim_scores(I, cloud, tr, ro, nh, nw, batch=batch).cpu().numpy()
        assert np.abs(scores - ref).max() <= 2e-3, (batch, np.abs(scores - ref).max())
        assert np.abs(scores - g["scores"]).max() <= 1e-2
    # without the carry-over the scores would be visibly different: the quirk is really exercised
    no_carry = g["split"].copy()
    no_carry[6, nw + 1:nw + 3] = 0
    assert abs(no_carry[6].sum() / (nh * nw) - g["scores"][6]) > 1e-2
    tt, trr = utils.trim_input_hist_secondary(I, X, C, tr, ro, 4, nh, nw)
    assert np.array_equal(tt.cpu().numpy(), g["ranked_trans"][:4]) and np.array_equal(trr.cpu().numpy(), g["ranked_rot"][:4])


def test_omniloc_all_equals_sequential_calls():
    """omniloc_all = the reference's `for i: omniloc(..., i, ...)` loop in one launch chain, result by result."""
    from piccolo_amd import omniloc as po
    g = load_golden("g5_trajectories.npz")
    import json
    cfg = Cfg(**json.loads(str(g["cfg"])))
    dev = torch.device("cuda")
    I, X, C = [torch.from_numpy(g[k]).to(dev) for k in ("img", "xyz", "rgb")]
    t_all, r_all = torch.from_numpy(g["trans0"].copy()).to(dev), torch.from_numpy(g["rot0"].copy()).to(dev)
    together = po.omniloc_all(I, X, C, t_all, r_all, cfg)
    t_seq, r_seq = torch.from_numpy(g["trans0"].copy()).to(dev), torch.from_numpy(g["rot0"].copy()).to(dev)
    for i in range(4):
        single = po.omniloc(I, X, C, t_seq, r_seq, i, cfg, {})
        assert all(torch.equal(a, b) for a, b in zip(single, together[i])), i
    assert torch.equal(t_all, t_seq) and torch.equal(r_all, r_seq)      # the callers' rows end up identical too


def test_make_input_composed_matches_the_reference(oracle, parity):
    """G23: the reference's make_input (utils.py:591-629) run on one scene for its three shipped configs, intermediates read out
    of the running functions.  Stage by stage on the reference's own candidate tables — the loss table of the yaw-shared trim
    launch, the survivors through the `// len(rot)`, `% len(rot)` decode (utils.py:500-505), the histogram scores of those
    survivors, the final starting poses — and then the product's make_input from nothing but the image, the cloud and the
    init_dict: its own candidate grids must be the reference's, and its final poses the reference's as far as the reference's
    scores decide them (its own rerun with permuted points moves them by 0.4e-2 ... 2e-2)."""
    from oracle import hist
    from parity_helpers import check_selection, match_rows
    from piccolo_amd import ops, utils
    from test_oracle_golden import G23_CONFIGS, g23_case, g23_check_table
    g = load_golden("g23_make_input.npz")
    dev = torch.device("cuda")
    for tag in G23_CONFIGS:
        xyz, rgb, img, init, n_in, n_mid, d = g23_case(g, tag)
        X, C, I = [torch.from_numpy(a).to(dev) for a in (xyz, rgb, img)]
        K, Rn = d["loss_loss_table"].shape
        tr, ro = torch.from_numpy(d["loss_trans"]).to(dev), torch.from_numpy(d["loss_rot"]).to(dev)
        # stage 1 on the reference's tables (its rotation order is that of a Python set of strings: arbitrary per process)
        table = ops.trim_loss_table(ops.Cloud(X, C), ops.Pano(I, fmt="u8"), tr, ops.TrimGroups(ro)).cpu().numpy()
        t1, r1 = utils.trim_input_loss(I, X, C, tr, ro, n_mid)
        got = match_rows(t1.cpu().numpy(), d["loss_trans"]) * Rn + match_rows(r1.cpu().numpy(), d["loss_rot"])
        assert np.array_equal(np.sort(table.reshape(-1)[got]), np.sort(table.reshape(-1))[:n_mid])      # it returns its own table's best
        g23_check_table(parity, "device", tag, table, d["loss_loss_table"], len(xyz), n_mid, got=got)
        # stage 2 on the reference's survivors
        ht, hr = torch.from_numpy(d["hist_trans"]).to(dev), torch.from_numpy(d["hist_rot"]).to(dev)
        scores = ops.hist_trim_scores(I, ops.Cloud(X, C), ht, hr, init["num_split_h"], init["num_split_w"]).cpu().numpy()
        self_noise = np.abs(d["hist_hist_intersect_permuted"] - d["hist_hist_intersect"]).max()
        parity("G23 %s: device histogram scores vs the reference's (yardstick: the reference's own rerun)" % tag,
               np.abs(scores - d["hist_hist_intersect"]).max(), 2.5 * self_noise + 1e-3, self_noise)
        oscores, _ = hist.hist_scores(img, xyz, rgb, d["hist_trans"], d["hist_rot"], init["num_split_h"], init["num_split_w"])
        parity("G23 %s: device histogram scores vs the oracle's (same nearest-wins render)" % tag, np.abs(scores - oscores).max(), 2e-3)
        ft, fr = utils.trim_input_hist_secondary(I, X, C, ht, hr, n_in, init["num_split_h"], init["num_split_w"])
        sel = match_rows(np.concatenate([ft.cpu().numpy(), fr.cpu().numpy()], 1), np.concatenate([d["hist_trans"], d["hist_rot"]], 1))
        check_selection(sel, d["hist_hist_intersect"], n_in, np.abs(scores - d["hist_hist_intersect"]) + 1e-6, largest=True)
        # the composition from scratch: own grids, own trims
        utils._ROT_GRIDS.clear()
        it, ir = utils.make_input(I, X, C, n_in, init, "loss_histogram", n_mid)
        own_t = utils.generate_trans_points(X, init, device=dev).cpu().numpy()
        own_r = utils.generate_rot_points(init, device=dev).cpu().numpy()
        assert own_t.shape == d["loss_trans"].shape and np.abs(own_t - d["loss_trans"]).max() <= 1e-5          # same grid, same order
        assert own_r.shape == d["loss_rot"].shape
        match_rows(own_r, d["loss_rot"], atol=1e-5)                                                          # same rotations, as a set
        fin = np.concatenate([it.cpu().numpy(), ir.cpu().numpy()], 1)
        assert fin.shape == (n_in, 6)
        # every final pose is one of the reference's stage-1 survivors or ranks within the stage-1 tie band, and as a selection
        # among the reference's survivors it is admissible at the measured score tolerance
        pool = np.concatenate([d["hist_trans"], d["hist_rot"]], 1)
        dist = np.abs(fin[:, None, :] - pool[None, :, :]).max(-1)
        inside = dist.min(1) <= 1e-5
        parity("G23 %s: make_input from scratch, final poses outside the reference's 50 survivors (stage-1 ties)" % tag, (~inside).sum(), 1)
        if inside.all():
            check_selection(dist.argmin(1), d["hist_hist_intersect"], n_in, np.abs(scores - d["hist_hist_intersect"]) + 1e-6, largest=True)


def test_xcd_per_image_mapping_gives_the_same_bits():
    """pcl_gd_hyper.images > 1 (set by set_pano_groups / set_panos): the XCDs split the pose groups — i.e. the query images — of a
    multi-image launch chain instead of the chunks of the cloud.  A mapping hint only: every (group, chunk) partial sum, hence every
    result, loss history and optimiser state is the same bit pattern as with the hint off; fused and two-launch shapes."""
    from piccolo_amd import ops, synth
    H, W, I, per = 64, 128, 8, 2
    for n in (20_000, 240_000):                               # 8 groups, clouds below the 6 MB limit of the mapping: one launch per iteration / two
        xyz, rgb = synth.box_room(n, 33)
        dev = torch.device("cuda")
        X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
        cloud, box = ops.Cloud(X, C), ops.quantile_box(X, 0.05)
        panos, tr, ro = [], [], []
        for i in range(I):
            t_gt, ypr_gt = synth.gt_pose(70 + i)
            img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
            panos.append(ops.Pano(img))
            a, b = synth.start_poses(t_gt, ypr_gt, per, seed=i)
            tr.append(a); ro.append(b)
        T_, R_ = torch.from_numpy(np.concatenate(tr)).to(dev), torch.from_numpy(np.concatenate(ro)).to(dev)
        out = []
        for hint in (True, False):
            gd = ops.GradientDescent(cloud, panos[0], T_, R_, box, lr=0.1, patience=5, factor=0.8, batch_mode=True)
            gd.set_pano_groups(panos)
            assert gd.hyper.images == I
            if not hint:
                gd.hyper.images = 0
            hist = gd.run(12, history=True)
            out.append((hist.clone(), gd.result().clone(), gd.state.clone()[: I * per * (160 + 64)]))
        for a, b in zip(*out):
            assert torch.equal(a, b), n


def test_select_poses_is_the_reference_argsort_and_decode():
    """pcl_select_poses against numpy's stable argsort: utils.py:500-505 (ascending, `// len(rot)`, `% len(rot)`) and
    utils.py:583-586 (the flipped tail of the ascending argsort), with ties, NaNs (ranked last), -0.0 / +0.0, negative values,
    n_keep = 1, = M and the 1024 limit, several problems per launch with shared and with per-problem pose tables."""
    from piccolo_amd import ops
    rng = np.random.default_rng(11)
    dev = torch.device("cuda")

    def ref_order(v, n, largest):
        key = np.where(np.isnan(v), np.inf, v).astype(np.float64)
        if not largest:
            return np.argsort(key, kind="stable")[:n]
        key = np.where(np.isnan(v), -np.inf, v).astype(np.float64)
        return np.argsort(key, kind="stable")[-n:][::-1]

    for K, R, n in ((75, 24, 50), (165, 8, 50), (7, 3, 21), (1, 1, 1), (300, 5, 1024), (2000, 1, 6)):
        M = K * R
        v = rng.normal(size=M).astype(np.float32)
        v[rng.integers(0, M, M // 10)] = np.float32(0.25)              # ties
        v[rng.integers(0, M, max(1, M // 50))] = np.nan
        if M > 4:
            v[1], v[3] = np.float32(0.0), np.float32(-0.0)             # one value: the tie goes to the index (-0.0 must not rank first)
        n = min(n, M)
        trans, rot = rng.normal(size=(K, 3)).astype(np.float32), rng.normal(size=(R, 3)).astype(np.float32)
        tt, tr, idx = ops.select_poses(torch.from_numpy(v).to(dev), n, torch.from_numpy(trans).to(dev), torch.from_numpy(rot).to(dev),
                                       largest=False, rot_per_trans=R, return_idx=True)
        want = ref_order(v, n, False)
        assert np.array_equal(idx.cpu().numpy(), want), (K, R, n)
        assert np.array_equal(tt.cpu().numpy(), trans[want // R]) and np.array_equal(tr.cpu().numpy(), rot[want % R])
        # the second stage's form: one pose row per value, best first
        pt, pr = rng.normal(size=(M, 3)).astype(np.float32), rng.normal(size=(M, 3)).astype(np.float32)
        tt, tr, idx = ops.select_poses(torch.from_numpy(v).to(dev), n, torch.from_numpy(pt).to(dev), torch.from_numpy(pr).to(dev),
                                       largest=True, return_idx=True)
        want = ref_order(v, n, True)
        assert np.array_equal(idx.cpu().numpy(), want), (K, R, n, "largest")
        assert np.array_equal(tt.cpu().numpy(), pt[want]) and np.array_equal(tr.cpu().numpy(), pr[want])
    # several problems per launch
    P, M, n = 5, 1800, 50
    v = rng.normal(size=(P, M)).astype(np.float32)
    trans, rot = rng.normal(size=(75, 3)).astype(np.float32), rng.normal(size=(24, 3)).astype(np.float32)
    tt, tr, idx = ops.select_poses(torch.from_numpy(v).to(dev), n, torch.from_numpy(trans).to(dev), torch.from_numpy(rot).to(dev),
                                   rot_per_trans=24, return_idx=True)
    for p in range(P):
        want = ref_order(v[p], n, False)
        assert np.array_equal(idx[p].cpu().numpy(), want)
        assert np.array_equal(tt[p].cpu().numpy(), trans[want // 24]) and np.array_equal(tr[p].cpu().numpy(), rot[want % 24])
    pt, pr = rng.normal(size=(P, 64, 3)).astype(np.float32), rng.normal(size=(P, 64, 3)).astype(np.float32)
    sc = rng.uniform(size=(P, 64)).astype(np.float32)
    tt, tr = ops.select_poses(torch.from_numpy(sc).to(dev), 6, torch.from_numpy(pt).to(dev), torch.from_numpy(pr).to(dev), largest=True)
    for p in range(P):
        want = ref_order(sc[p], 6, True)
        assert np.array_equal(tt[p].cpu().numpy(), pt[p][want]) and np.array_equal(tr[p].cpu().numpy(), pr[p][want])
    # a table far larger than the block (every thread strides through it), all values distinct or all equal
    M = 300_007
    v = rng.permutation(M).astype(np.float32)
    pt = rng.normal(size=(M, 3)).astype(np.float32)
    tt, tr, idx = ops.select_poses(torch.from_numpy(v).to(dev), 1000, torch.from_numpy(pt).to(dev), torch.from_numpy(pt).to(dev), return_idx=True)
    assert np.array_equal(idx.cpu().numpy(), ref_order(v, 1000, False)) and np.array_equal(tt.cpu().numpy(), pt[ref_order(v, 1000, False)])
    same = torch.full((5000,), 0.5, device=dev)
    _, _, idx = ops.select_poses(same, 7, torch.zeros(5000, 3, device=dev), torch.zeros(5000, 3, device=dev), return_idx=True)
    assert idx.cpu().tolist() == list(range(7))                   # ties: ascending index
    _, _, idx = ops.select_poses(same, 7, torch.zeros(5000, 3, device=dev), torch.zeros(5000, 3, device=dev), largest=True, return_idx=True)
    assert idx.cpu().tolist() == list(range(4999, 4992, -1))      # largest: descending index (the flipped tail of the stable argsort)
    with pytest.raises(Exception):
        ops.select_poses(torch.zeros(10, device=dev), 11, torch.zeros(10, 3, device=dev), torch.zeros(10, 3, device=dev))


def test_gd_winner_is_argmin_of_the_last_losses_with_its_rotation():
    """pcl_gd_winner (omniloc.py:271-277 on the device) against the host-side form it replaces: torch.argmin over the last
    losses, rot_from_ypr of the winner, leaf parameters of all candidates — for one image and for 3 images x 4 candidates."""
    from piccolo_amd import ops, synth
    n, H, W = 20_000, 64, 128
    xyz, rgb = synth.box_room(n, 2)
    dev = torch.device("cuda")
    X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
    cloud = ops.Cloud(X, C)
    panos, tr, ro = [], [], []
    for i in range(3):
        t_gt, ypr_gt = synth.gt_pose(20 + i)
        img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))
        panos.append(ops.Pano(img))
        a, b = synth.start_poses(t_gt, ypr_gt, 4, seed=i)
        tr.append(a); ro.append(b)
    box = ops.quantile_box(X, 0.05)
    for nimg in (1, 3):
        T_, R_ = torch.from_numpy(np.concatenate(tr[:nimg])).to(dev), torch.from_numpy(np.concatenate(ro[:nimg])).to(dev)
        gd = ops.GradientDescent(cloud, panos[0], T_, R_, box, lr=0.1, patience=5, factor=0.8, batch_mode=True)
        gd.set_pano_groups(panos[:nimg])
        gd.run(7)
        res = gd.result()
        # the same chain with the device-table form of the panorama list gives the same bits
        gd2 = ops.GradientDescent(cloud, panos[0], T_, R_, box, lr=0.1, patience=5, factor=0.8, batch_mode=True)
        gd2.set_panos([panos[i] for i in range(nimg) for _ in range(4)])
        gd2.run(7)
        assert torch.equal(res, gd2.result())
        lt, lr_ = torch.empty(4 * nimg, 3, device=dev), torch.empty(4 * nimg, 3, device=dev)
        win = gd.winner(nimg, lt, lr_)
        assert torch.equal(lt, res[:, 6:9]) and torch.equal(lr_, res[:, 9:12])
        for i in range(nimg):
            blk = res[4 * i:4 * i + 4]
            k = int(torch.argmin(blk[:, 12]))
            assert torch.equal(win[i, 0:3], blk[k, 0:3]) and torch.equal(win[i, 13:16], blk[k, 3:6]) and win[i, 12] == blk[k, 12]
            assert torch.equal(win[i, 3:12], ops.rot_from_ypr(blk[k:k + 1, 3:6])[0].reshape(-1))


def test_make_input_images_equals_per_image_make_input():
    """make_input_images (one trim launch over image x translation x rotation, one selection launch for all images) returns, for
    every image, the very tensors make_input returns for it: the multi-image launch cuts the cloud into the single-image launch's
    chunks, so the loss tables agree bit for bit (also checked directly), and the selections are deterministic."""
    from piccolo_amd import ops, synth, utils
    from piccolo_amd.omniloc import packed_cloud, packed_pano
    n, H, W, I = 40_000, 128, 256, 5
    xyz, rgb = synth.box_room(n, 4)
    dev = torch.device("cuda")
    X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
    imgs = []
    for i in range(I):
        t_gt, ypr_gt = synth.gt_pose(40 + i)
        imgs.append(synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W))))
    for init in (STANFORD, OMNI):
        d = dict(init)
        if d.get("z_prior") is not None:
            d["z_prior"] = 0.0
        single = [utils.make_input(im, X, C, 6, d, "loss_histogram", 50) for im in imgs]
        multi = utils.make_input_images(imgs, X, C, 6, d, "loss_histogram", 50)
        assert len(multi) == I
        for (a, b), (c, e) in zip(single, multi):
            assert torch.equal(a, c) and torch.equal(b, e)
        rot = utils.generate_rot_points(d, device=dev)
        trans = utils.generate_trans_points(X, d, device=dev)
        groups = ops.TrimGroups(rot)
        cloud = packed_cloud(X, C)
        panos = [packed_pano(im, many_poses=True) for im in imgs]
        tabs, cnts = ops.trim_loss_tables(cloud, panos, trans, groups, return_count=True)
        for i in range(I):
            t1, c1 = ops.trim_loss_table(cloud, panos[i], trans, groups, return_count=True)
            assert torch.equal(tabs[i], t1) and torch.equal(cnts[i], c1), i
        # the second stage for all images at once: row i = the single-image call, bit for bit (non-contiguous slices included)
        t1, r1 = ops.select_poses(tabs.reshape(I, -1), 50, trans, rot, rot_per_trans=len(rot))
        both = ops.hist_trim_scores_images(imgs, cloud, t1, r1, d["num_split_h"], d["num_split_w"])
        for i in range(I):
            assert torch.equal(both[i], ops.hist_trim_scores(imgs[i], cloud, t1[i], r1[i], d["num_split_h"], d["num_split_w"])), i
    # eight images in one launch: the XCDs split the IMAGES instead of the chunks (pcl_trim.hip, xcd_images) — the same rows
    eight = [panos[i % I] for i in range(8)]
    tabs8 = ops.trim_loss_tables(cloud, eight, trans, groups)
    for i in range(8):
        assert torch.equal(tabs8[i], tabs[i % I]), i
    # more images than one launch takes (32 for the trim launch; the second stage is forced to groups of two here): the chunked
    # calls give the same rows
    many = [panos[i % I] for i in range(ops.TRIM_MAX_IMAGES + 3)]
    tabs_many = ops.trim_loss_tables(cloud, many, trans, groups)
    for i in range(len(many)):
        assert torch.equal(tabs_many[i], tabs[i % I]), i
    old_max = ops.HIST_MAX_IMAGES
    try:
        ops.HIST_MAX_IMAGES = 2
        assert torch.equal(ops.hist_trim_scores_images(imgs, cloud, t1, r1, d["num_split_h"], d["num_split_w"]), both)
    finally:
        ops.HIST_MAX_IMAGES = old_max
    with pytest.raises(ValueError):
        ops.trim_loss_tables(cloud, [panos[0], ops.Pano(imgs[1], fmt="f32")], trans, groups)


REQUIRED_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                      "dtype", "data", "config", "roofline", "checks", "ranks_seen", "per_rank_ms_per_step")


def _run_bench(cmd, env_extra, timeout=900, base_env=None):
    """-> (the compact LAST stdout line, the complete record of the side file).  The line is what the driver parses: one line, the last
    thing on stdout, below 4 KB (BENCH_r05.parsed was null for a 24.5 KB line), with every required key."""
    import json as js
    import os
    import subprocess
    import tempfile
    from conftest import REPO
    env = dict(os.environ if base_env is None else base_env, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    with tempfile.TemporaryDirectory() as tmp:
        side = os.path.join(tmp, "bench_also.json")
        out = subprocess.run(cmd + ["--also-json", side], env=env, cwd=REPO, capture_output=True, text=True, timeout=timeout)
        assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
        full = js.load(open(side))
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]             # rank 0 prints exactly one JSON line ...
    assert out.stdout.rstrip().endswith(lines[0])          # ... and it is the last thing on stdout (RCCL banner flushed before)
    assert len(lines[0]) < 4096, len(lines[0])
    d = js.loads(lines[0])
    assert not [k for k in REQUIRED_LINE_KEYS if k not in d], [k for k in REQUIRED_LINE_KEYS if k not in d]
    assert "also" not in d and set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"}
    for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup"):            # the line is the record, rounded
        assert abs(d[k] - full[k]) <= 1e-5 * abs(full[k]), k
    return d, full


def test_bench_two_ranks_end_to_end(tmp_path):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per rank), here
    with two ranks sharing the one GPU over gloo: exercises the sharding of query images, the barrier-bracketed timing,
    the max-over-ranks reduction and the result gather on real kernels."""
    import os
    import sys
    from conftest import REPO
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "cfg1", "--no-cpu-baseline", "--min-seconds", "0.2"]
    d, full = _run_bench(cmd, {"PCL_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["unit"] == "candidate-poses/s"
    assert d["ranks_seen"] == 2                              # the gathered rows carry both ranks' stamps
    assert d["value"] > 0 and abs(d["value"] - 1 * 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-4
    assert full["value"] > 0 and abs(full["value"] - 1 * 2 * 2 / (full["ms_per_step"] * 2 / 1e3)) / full["value"] < 1e-6
    assert d["passes"] >= 1 and d["pass_ms"]["min"] <= d["pass_ms"]["median"] <= d["pass_ms"]["max"]
    assert d["median_t_err_m"] < 0.1 and "roofline" in d and d["vs_baseline"] is None
    assert d["per_rank_ms_per_step"]["min"] <= d["per_rank_ms_per_step"]["max"] and len(full["per_rank_ms_per_step"]["ranks"]) == 2
    s = full["single_image"]                                 # cfg 1: 2 images per launch chain by default, 1 in this pass
    assert s["images_per_launch"] == 1 and s["value"] > 0 and s["poses_per_launch"] == 1
    assert d["single_image"]["poses_per_launch"] == 1 and d["single_image"]["value"] > 0


def test_bench_rccl_path_at_world_size_one():
    """N > 1 readiness on a one-GPU box: PCL_BENCH_FORCE_DIST=1 takes bench.py through everything an N-GPU run does —
    RCCL init bound to the device, the warm-up all_gather, barriers, all_reduce(MAX) of the time, the result
    all_gather_into_tensor, the banner flush — with a world of one.  Exactly one JSON line, the last thing on stdout."""
    import os
    import sys
    from conftest import REPO
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "cfg1",
           "--no-cpu-baseline", "--no-also", "--min-seconds", "0.2"]
    d, full = _run_bench(cmd, {"PCL_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29579"})
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["devices_visible"] >= 1 and d["value"] > 0
    assert d["dist_backend"] == "nccl" and d["roofline"]["bound"] == "hbm" and d["roofline"]["traffic"] is None
    r = full["roofline"]
    # cfg 1 at 2 poses per launch was never profiled with counters: no VALU instruction count for this shape, so the line falls
    # back to the algorithmic-bytes figure and says so instead of borrowing another shape's numbers
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "FALLBACK" in r["bound_is"]
    assert r["traffic"] is None and r["hbm_measured"] is None and r["valu"] is None and "cfg1" in r["traffic_key"]
    assert abs(r["algorithmic_hbm"]["frac"] - r["frac"]) < 1e-12 and r["event_pair_ms_subtracted"] >= 0
    assert r["avg_launch_ms"] <= r["avg_launch_ms_raw_events"] and len(r["source_hash_loaded_library"]) == 16
    assert d["checks"]["kernel_time_within_step"] and "also" not in full and "also_brief" not in d


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` as the driver calls the N = 1 bench, no torchrun around it (localize.py:143,357: the query
    images are what is sharded): the launcher process starts two fresh ranks before it has touched the GPU, the ranks share the
    one GPU over gloo, and the launcher's stdout is exactly rank 0's JSON line — a COMPLETE one: cpu_baseline, a reduced `also`
    block and the N = 1 value of the same build are measured by rank 0 after the job's final barrier."""
    import os
    import sys
    from conftest import REPO
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "cfg1", "--steps", "2", "--warmup", "1",
           "--min-seconds", "0.2", "--cpu-baseline-seconds", "2"]
    env = {"PCL_DIST_BACKEND": "gloo"}
    assert "WORLD_SIZE" not in os.environ
    d, full = _run_bench(cmd, env)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["unit"] == "candidate-poses/s" and cb["sample"]
    n1 = d["n1_value_same_build"]
    assert n1["value"] > 0 and d["value"] > 0
    assert d["also_brief"]["pipeline_shipped"] > 0 and d["ipc_mode_legacy"] == "0" and d["ipc_mode_from"] in ("environment", "default")
    also = full["also"]
    assert "error" not in also, also
    assert also["pipeline_shipped"]["total_ms"] > 0 and also["shipped_8_images_per_chain"]["value"] > 0
    assert "cfg5" not in also and "cfg3" not in also          # the reduced set of an N > 1 line
    assert d["roofline"]["frac"] > 0 and d["single_image"]["value"] > 0


def test_bench_falls_back_to_gloo_when_rccl_cannot_set_up():
    """the first real multi-GPU run must produce a line either way.  Started the way the DRIVER starts N > 1
    (its own torch.distributed.run: no launcher of ours to retry), with RCCL's set-up "failing" on every rank (test hook): the ranks
    re-form the group over gloo, the only collective (16 floats per image) goes over host memory, and the line says so."""
    import os
    import sys
    from conftest import REPO
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29587", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "cfg1", "--no-cpu-baseline", "--no-also", "--min-seconds", "0.2"]
    env = {k: v for k, v in os.environ.items() if k != "PCL_DIST_BACKEND"}
    d, full = _run_bench(cmd, {"PCL_BENCH_TEST_FAIL_NCCL": "1"}, base_env=env)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["dist_backend"] == "gloo" and "RCCL set-up failed" in d["dist_fallback"]
    assert d["value"] > 0 and d["ipc_mode_from"] == "environment"


def test_bench_launcher_retry_end_to_end():
    """The launcher's one retry on real ranks: two self-launched gloo ranks whose process-group set-up "fails" under
    HSA_ENABLE_IPC_MODE_LEGACY=0 (test hook) — the second launch runs with 1 and its line is relayed."""
    import os
    import sys
    from conftest import REPO
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "cfg1", "--steps", "2", "--warmup", "1",
           "--min-seconds", "0.2", "--no-cpu-baseline", "--no-also"]
    d, full = _run_bench(cmd, {"PCL_DIST_BACKEND": "gloo", "PCL_BENCH_TEST_FAIL_IF_IPC": "0"})
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["ipc_mode_legacy"] == "1" and d["ipc_mode_from"] == "retry"


def test_bench_names_a_dead_rank():
    """a rank that dies must not leave the job hanging.  Two self-launched gloo ranks, rank 1 leaves before its
    first barrier (test hook), PCL_DIST_TIMEOUT_S = 20: the job ends within seconds, non-zero, without a JSON line — rank 0's barrier
    fails inside the timeout with ONE line naming the collective and the reason (or torch.distributed.run, seeing rank 1's exit code,
    ends rank 0 first and reports the failed rank itself)."""
    import os
    import subprocess
    import sys
    import time
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PCL_DIST_BACKEND="gloo", PCL_BENCH_TEST_DIE_RANK="1", PCL_DIST_TIMEOUT_S="20")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "cfg1", "--steps", "2", "--warmup", "1",
                          "--no-also", "--no-cpu-baseline"], env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0, out.stderr[-2000:]
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "leaving before the barrier" in out.stderr
    assert "bench.py rank 0/2" in out.stderr and "FAILED" in out.stderr, out.stderr[-2000:]
    assert time.time() - t0 < 300                                   # ended by the 20 s collective timeout, not by torch's 30 minutes


def test_bench_launcher_relays_the_ranks_refusal():
    """The same command line with the default back end (RCCL: one GPU per rank) on a box with fewer GPUs than ranks: every
    rank refuses with the "visible" message, the launcher exits non-zero and prints no JSON line."""
    import os
    import subprocess
    import sys
    from conftest import REPO
    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PCL_DIST_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n + 1), "--workload", "cfg1", "--steps", "2",
                          "--warmup", "1"], env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "visible" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_more_ranks_than_gpus():
    """--gpus N with fewer than N visible devices (RCCL needs one GPU per rank): a clear message, no hang."""
    import os
    import subprocess
    import sys
    from conftest import REPO
    n = torch.cuda.device_count()
    env = dict(os.environ, WORLD_SIZE=str(n + 1), RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29581")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                         env=env, cwd=REPO, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "visible" in (out.stderr + out.stdout)


def test_warp_from_img_matches_golden_samples():
    """warp_from_img == sample_from_img on a grid of coordinates (G2's reference samples, reshaped)."""
    from piccolo_amd import utils
    g = load_golden("g2_sample_from_img.npz")
    coord = torch.from_numpy(g["coord"][:1200].reshape(30, 40, 2)).cuda()
    out = utils.warp_from_img(torch.from_numpy(g["img"]).cuda(), coord).cpu().numpy()
    assert out.shape == (30, 40, 3) and np.abs(out.reshape(-1, 3) - g["rgb"][:1200]).max() <= 2e-6
    # reshape_img_tensor (utils.py:632-638): uint8 round trip + bilinear resize to size = (X, Y), on the input's device
    small = utils.reshape_img_tensor(torch.full((4, 8, 3), 0.5), (4, 2))
    assert tuple(small.shape) == (2, 4, 3) and torch.allclose(small, torch.full((2, 4, 3), 127 / 255.))


def test_small_utils_match_reference():
    """G17: create_coordinate, compute_sampling_grid, adaptive_trans_num, out_of_room, get_bound as the reference returned
    them for the same cloud."""
    from piccolo_amd import utils
    g = load_golden("g17_small_utils.npz")
    X = torch.from_numpy(g["xyz"]).cuda()
    assert np.abs(utils.create_coordinate(4, 8).numpy() - g["coord_4x8"]).max() <= 1e-6
    for k, ypr in enumerate(g["yprs"]):
        y = torch.from_numpy(ypr).cuda()
        assert np.abs(utils.compute_sampling_grid(y, 4, 4).cpu().numpy() - g["grids_4x4"][k]).max() <= 2e-6
        assert np.abs(utils.compute_sampling_grid(y, 2, 4).cpu().numpy() - g["grids_2x4"][k]).max() <= 2e-6
    assert tuple(utils.adaptive_trans_num(X, 50, xy_only=False)) == tuple(g["adaptive_xyz_50"])
    assert tuple(utils.adaptive_trans_num(X, 150, xy_only=True)) == tuple(g["adaptive_xy_150"])
    for q, key in ((0.05, "out_of_room_q05"), (0.2, "out_of_room_q20")):
        got = [utils.out_of_room(X, torch.from_numpy(p).reshape(3, 1), q) for p in g["probes"]]
        assert got == [bool(v) for v in g[key]]
    b = utils.get_bound(X, Cfg(out_of_room_quantile=0.1, max_yaw=3.0))
    got = np.array([b[k] for k in ("x", "y", "z", "yaw", "pitch", "roll")], np.float64)
    assert np.abs(got - g["bound_q10"]).max() <= 1e-6


def test_fused_iterations_are_bit_identical_to_the_two_launch_form(oracle):
    """Launches whose blocks are all resident at once (the reference's shipped 167k-point / 6-candidate shape, cfg 1) run ONE
    launch per GD iteration: every block of iteration k + 1 finishes iteration k for its own poses in its prologue (same
    reduction order, chain rule, Adam, scheduler, clamp as the stand-alone epilogue kernel; the block of chunk 0 stores the
    state), the last iteration is finished by the stand-alone epilogue.  fuse=False (pcl_gd_hyper.fuse = -1) selects the two-launch form:
    optimiser state, poses, per-iteration loss history and the caller-visible results must agree BIT FOR BIT — both modes,
    even and odd candidate counts, one candidate, per-candidate panoramas, 1 / 2 / 3 / 100 iterations, run() called in pieces."""
    from piccolo_amd import ops, synth
    H, W = 64, 128
    cases = [(20_000, 6, True), (20_000, 5, True), (20_000, 1, False), (20_000, 4, False), (166_667, 6, True), (100_000, 1, False)]
    if True:
        for n, B, batch_mode in cases:
            xyz, rgb = synth.box_room(n, 90 + B)
            X, C = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
            cloud = ops.Cloud(X, C)
            box = ops.quantile_box(X, 0.05)
            panos, tr_all, ro_all = [], [], []
            for k in range(2):
                t_gt, ypr_gt = synth.gt_pose(90 + k)
                img = oracle.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W)).astype(np.float32) / 255
                panos.append(ops.Pano(torch.from_numpy(img).cuda()))
                tr, ro = synth.start_poses(t_gt, ypr_gt, B, seed=90 + k)
                tr[0, 0] = 4.4                                   # one start outside the clamp box (batch mode's clamp lag)
                tr_all.append(tr)
                ro_all.append(ro)
            tr, ro = torch.from_numpy(tr_all[0]).cuda(), torch.from_numpy(ro_all[0]).cuda()
            table = [panos[b % 2] for b in range(B)]             # candidates alternate between two panoramas
            out = {}
            for mode in ("two", "fused"):
                fuse = False if mode == "two" else None
                fz = ctypes.c_int(-1)
                hy = ops._lib.GdHyper(0.1, 0.8, 5, 1 if batch_mode else 0, 0, 0.0, 0, 0, 0, -1 if fuse is False else 0, 0)
                assert ops._lib.load().pcl_gd_plan_hyper(n, B, ctypes.byref(hy), None, None, ctypes.byref(fz)) == 0
                assert fz.value == (0 if mode == "two" else 1), (n, B, mode)        # every case here fuses by the rule
                runs = []
                for num_iter in (1, 2, 3, 100):
                    gd = ops.GradientDescent(cloud, panos[0], tr, ro, box, lr=0.1, patience=5, factor=0.8, batch_mode=batch_mode, fuse=fuse)
                    gd.set_panos(table)
                    hist = gd.run(num_iter, history=True)
                    runs.append((hist.clone(), gd.result().clone(), gd.state.clone()[: B * (160 + 64)]))
                # in pieces: 2 + 1 + 4 iterations continue one another exactly like 7 in one call
                gd = ops.GradientDescent(cloud, panos[0], tr, ro, box, lr=0.1, patience=5, factor=0.8, batch_mode=batch_mode, fuse=fuse)
                gd.set_panos(table)
                pieces = torch.cat([gd.run(2, history=True), gd.run(1, history=True), gd.run(4, history=True)])
                gd7 = ops.GradientDescent(cloud, panos[0], tr, ro, box, lr=0.1, patience=5, factor=0.8, batch_mode=batch_mode, fuse=fuse)
                gd7.set_panos(table)
                whole = gd7.run(7, history=True)
                assert torch.equal(pieces, whole) and torch.equal(gd.result(), gd7.result()), (n, B, mode)
                runs.append((pieces.clone(), gd.result().clone(), gd.state.clone()[: B * (160 + 64)]))
                out[mode] = runs
            for (h2, r2, s2), (hf, rf, sf) in zip(out["two"], out["fused"]):
                assert torch.equal(h2, hf), (n, B, "loss history", (h2 - hf).abs().max())
                assert torch.equal(r2, rf), (n, B, "result")
                assert torch.equal(s2, sf), (n, B, "optimiser state")          # (canonical copy of the state blob — B x (160 B state + 64 B pose record) — byte for byte)


def test_small_problems_replay_a_cached_graph_and_stay_bit_identical(oracle):
    """Small refinements (points x candidates <= GRAPH_POINT_POSES) go through a GradientDescent object cached per cloud and
    launch shape whose launch chain is captured into a hipGraph once and replayed for every later image.  Results must be
    those of fresh eager launches, bit for bit — across images, across repeated calls and for omniloc / omniloc_batch /
    omniloc_batch_images alike; cfg.gd_graph=False forces the eager path."""
    from piccolo_amd import omniloc as po
    from piccolo_amd import synth
    n, H, W, B = 20_000, 64, 128, 4
    xyz, rgb = synth.box_room(n, 71)
    X, C = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
    base = dict(lr=0.1, num_iter=20, patience=5, factor=0.8, out_of_room_quantile=0.05, num_input=B)
    imgs, trs, ros = [], [], []
    for k in range(3):
        t_gt, ypr_gt = synth.gt_pose(80 + k)
        img = oracle.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W)).astype(np.float32) / 255
        imgs.append(torch.from_numpy(img).cuda())
        tr, ro = synth.start_poses(t_gt, ypr_gt, B, seed=80 + k)
        trs.append(torch.from_numpy(tr).cuda())
        ros.append(torch.from_numpy(ro).cuda())
    assert n * B <= po.GRAPH_POINT_POSES
    po._cache.clear()
    eager = [po.omniloc_batch(imgs[k], X, C, trs[k].clone(), ros[k].clone(), Cfg(gd_graph=False, **base), {}) for k in range(3)]
    assert "gd" not in po._cache.kinds or len(po._cache.kinds["gd"]) == 0
    for rep in range(2):                                     # second round: the graph captured in the first is replayed
        for k in range(3):
            it, ir = trs[k].clone(), ros[k].clone()
            got = po.omniloc_batch(imgs[k], X, C, it, ir, Cfg(**base), {})
            assert all(torch.equal(a, b) for a, b in zip(eager[k], got)), (rep, k)
    assert len(po._cache.kinds["gd"]) == 1                   # one engine served all six refinements
    eng = next(iter(po._cache.kinds["gd"].values()))[1]
    assert len(eng._graphs) == 1
    # sequential mode, one starting point at a time (the reference's non-parallel branch): its own engine (B = 1)
    s_eager = po.omniloc(imgs[1], X, C, trs[1].clone(), ros[1].clone(), 2, Cfg(gd_graph=False, **base), {})
    for rep in range(2):
        s_graph = po.omniloc(imgs[1], X, C, trs[1].clone(), ros[1].clone(), 2, Cfg(**base), {})
        assert all(torch.equal(a, b) for a, b in zip(s_eager, s_graph))
    # several images per launch chain
    m_eager = po.omniloc_batch_images(imgs, X, C, [t.clone() for t in trs], [r.clone() for r in ros], Cfg(gd_graph=False, **base))
    m_graph = po.omniloc_batch_images(imgs, X, C, [t.clone() for t in trs], [r.clone() for r in ros], Cfg(**base))
    for a, b in zip(m_eager, m_graph):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    # colours that change with every query image (color_mod / match_color hand each image its own rgb): still ONE engine per
    # point set and launch shape — the new colours are copied into the engine's private packed cloud, whose address the captured
    # graph holds — and the results are those of fresh eager launches on that cloud
    n_eng = len(po._cache.kinds["gd"])
    for scale in (0.9, 0.8, 0.9):
        C2 = (C * scale).contiguous()
        e2 = po.omniloc_batch(imgs[0], X, C2, trs[0].clone(), ros[0].clone(), Cfg(gd_graph=False, **base), {})
        g2 = po.omniloc_batch(imgs[0], X, C2, trs[0].clone(), ros[0].clone(), Cfg(**base), {})
        assert all(torch.equal(a, b) for a, b in zip(e2, g2)), scale
        assert not torch.equal(e2[0], eager[0][0])                   # (other colours: another answer)
    assert len(po._cache.kinds["gd"]) == n_eng
    back = po.omniloc_batch(imgs[0], X, C, trs[0].clone(), ros[0].clone(), Cfg(**base), {})
    assert all(torch.equal(a, b) for a, b in zip(eager[0], back))      # and the original colours again


def test_hist_trim_tile_binned_equals_the_zbuffer_path(oracle):
    """The tile-binned render + histogram (no z-buffer in HBM) uses the same 64-bit priority keys as the z-buffer splat:
    same winners, same integer histogram counts, hence BIT-identical intersections, counts and scores — on the golden
    scenes (tiles larger than the histogram blocks: the direct-to-global branch) and on a 512 x 1024 panorama with 200k
    points (tiles inside histogram blocks, several tiles per block, points on tile borders, ragged image edges)."""
    import os
    from piccolo_amd import ops, synth
    cases = []
    g = load_golden("g12_trim_input_hist.npz")
    cases.append((g["img"], g["xyz"], g["rgb"], g["trans"], g["rot"], 4, 4))
    g = load_golden("g19_trim_hist_empty_blocks.npz")
    cases.append((g["img"], g["xyz"], g["rgb"], g["trans"], g["rot"], 4, 4))
    # (also: fewer points than a binning block, an image smaller than one tile, image edges that cut tiles and histogram
    # blocks, one candidate, many thin histogram rows)
    for (n, H, W, nh, nw) in ((200_000, 512, 1024, 4, 4), (60_000, 200, 330, 5, 3), (100, 40, 50, 3, 1), (3_000, 63, 129, 3, 2),
                              (20_000, 130, 257, 8, 5), (5_000, 64, 64, 4, 4)):
        xyz, rgb = synth.box_room(n, 90)
        t_gt, ypr_gt = synth.gt_pose(90)
        X, C = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
        img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W))).cpu().numpy()
        tr, ro = synth.start_poses(t_gt, ypr_gt, 1 if n == 5_000 else 9, seed=90, sigma_t=0.5, sigma_r=0.4)
        cases.append((img, xyz, rgb, tr, ro, nh, nw))
    # One tile fed by more bin blocks than the resolve workgroup has threads (its run table goes through in batches): 2.3M points — 1124
    # blocks of 2048 — in a sheet that the first candidate, the sheet's own camera, sees in the middle of ONE 64 x 64 tile of a 256 x 128
    # panorama (the other candidates cut it across tiles), on top of a room.
    H, W, n_blob = 128, 256, 2_300_000
    rng = np.random.default_rng(7)
    t_gt, ypr_gt = synth.gt_pose(91)
    for _ in range(200):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        r0, c0 = oracle.pano_pixels((5.0 * d)[None].astype(np.float32), (H, W))
        if 44 <= int(r0[0]) <= 52 and 12 <= int(c0[0]) % 64 <= 52:          # a scored block row (32 .. 95), mid-tile
            break
    else:
        raise AssertionError("no direction found")
    e1 = np.cross(d, [0.0, 0.0, 1.0]); e1 /= np.linalg.norm(e1); e2 = np.cross(d, e1)
    # (a SHEET facing the camera, 20 x 20 pixels: its ~400 winners are spread over all of the tile's runs, the late ones included)
    cam_blob = (5.0 * d)[None] + rng.uniform(-1.2, 1.2, size=(n_blob, 1)) * e1[None] + rng.uniform(-1.2, 1.2, size=(n_blob, 1)) * e2[None] \
        + rng.normal(0, 0.002, size=(n_blob, 1)) * d[None]
    Rgt = synth.rot_from_ypr_np(ypr_gt.astype(np.float64))
    world_blob = (cam_blob @ Rgt + t_gt[None]).astype(np.float32)            # x = R^T p + t   (p = R (x - t))
    room_xyz, room_rgb = synth.box_room(50_000, 91)
    xyz = np.concatenate([room_xyz, world_blob]).astype(np.float32)
    rgb = np.concatenate([room_rgb, rng.integers(0, 256, size=(n_blob, 3)).astype(np.float32) / 255.0]).astype(np.float32)
    X, C = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()
    img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W))).cpu().numpy()
    tr, ro = synth.start_poses(t_gt, ypr_gt, 3, seed=91, sigma_t=0.05, sigma_r=0.05)
    tr[0], ro[0] = t_gt, ypr_gt
    cases.append((img, xyz, rgb, tr, ro, 4, 4))
    del X, C
    dev = torch.device("cuda")
    try:
        for img, xyz, rgb, tr, ro, nh, nw in cases:
            I, X, C = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (img, xyz, rgb)]
            T_, R_ = torch.from_numpy(tr).to(dev), torch.from_numpy(ro).to(dev)
            cloud = ops.Cloud(X, C)
            out = {}
            for mode in ("1", "0"):
                out[mode] = ops.hist_trim_scores(I, cloud, T_, R_, nh, nw, batch=4, return_parts=True, splat=mode == "1")
            for a, b in zip(out["1"], out["0"]):
                assert torch.equal(a, b), (img.shape, len(xyz))
            assert float(out["0"][0].abs().sum()) > 0
    finally:
        pass


@pytest.mark.gpu
def test_results_do_not_depend_on_the_block_to_xcd_mapping():
    """Which XCD evaluates which chunk, and in which order (contiguous ranges, interleaved chunks, odd iterations walking
    backwards), is scheduling only: every chunk's partial sums land in its own slot and the second-stage sum runs over the
    slots in index order, so a multi-round refinement must come out bit for bit the same under every mapping.  The knobs exist
    in the EXPERIMENTS build only (lib/libpiccolo_hip_exp.so, -DPCL_EXPERIMENTS; the shipped library reads no environment variable)
    and are read once per process: one child process per setting, plus one with the shipped library — same digest."""
    import hashlib
    import os
    import subprocess
    import sys
    from conftest import REPO
    code = (
        "import hashlib, numpy as np, torch\n"
        "from piccolo_amd import ops, synth\n"
        "n, H, W, B = 300_000, 256, 512, 8\n"
        "xyz, rgb = synth.box_room(n, 5)\n"
        "X, C = torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()\n"
        "t_gt, ypr_gt = synth.gt_pose(5)\n"
        "img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t_gt), torch.from_numpy(ypr_gt)), C, (H, W)))\n"
        "tr, ro = synth.start_poses(t_gt, ypr_gt, B, seed=5)\n"
        "cloud, pano = ops.Cloud(X, C), ops.Pano(img)\n"
        "assert ops._lib.load().pcl_loss_workspace_bytes(n, B) // (B * 32) * (B // 2) > 1024      # chunks x groups: several rounds\n"
        "gd = ops.GradientDescent(cloud, pano, torch.from_numpy(tr).cuda(), torch.from_numpy(ro).cuda(), ops.quantile_box(X, 0.05), lr=0.1, factor=0.9, patience=5)\n"
        "gd.run(25)\n"
        "print(hashlib.sha256(gd.result().cpu().numpy().tobytes()).hexdigest())\n")
    from piccolo_amd import build as hip_build
    exp_so = hip_build.build_experiments()                  # (prebuilt by __graft_entry__.build(); compiled here if missing or stale)
    digests = {}
    for runs, flip in (("0", "1"), ("1", "0"), ("4", "1"), ("0", "0"), (None, None)):
        env = dict(os.environ, PCL_XCD_RUNS=runs, PCL_FLIP=flip, PCL_SO=exp_so) if runs is not None else {k: v for k, v in os.environ.items() if k != "PCL_SO"}
        out = subprocess.run([sys.executable, "-c", code], env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        digests[(runs, flip)] = out.stdout.strip().splitlines()[-1]
    assert len(set(digests.values())) == 1, digests
    assert len(hashlib.sha256(b"").hexdigest()) == len(next(iter(digests.values())))


@pytest.mark.gpu
def test_cached_engines_survive_alternating_clouds_sizes_and_eviction(oracle):
    """The cached GradientDescent engines (one per cloud and launch shape, graph replay for small problems) hold device
    addresses inside captured graphs.  Alternate between two clouds, two image sizes and two candidate counts for long
    enough that engines are evicted and rebuilt (8 combinations cycle through a cache of 6), and check every result against an eager, uncached refinement of the same inputs, bit for bit."""
    from piccolo_amd import omniloc as po
    from piccolo_amd import synth
    base = dict(lr=0.1, num_iter=12, patience=3, factor=0.8, out_of_room_quantile=0.05)
    clouds = []
    for seed, n in ((101, 15_000), (102, 22_001)):
        xyz, rgb = synth.box_room(n, seed)
        clouds.append((xyz, rgb, torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda()))
    po._cache.clear()
    combos = [(c, hw, B) for c in (0, 1) for hw in ((48, 96), (64, 128)) for B in (2, 4)]
    assert len(combos) > po._CAPACITY["gd"]
    for rep in range(3):
        for k, (c, (H, W), B) in enumerate(combos):
            xyz, rgb, X, C = clouds[c]
            t_gt, ypr_gt = synth.gt_pose(200 + 7 * rep + k)
            img = torch.from_numpy(oracle.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W)).astype(np.float32) / 255).cuda()
            tr, ro = synth.start_poses(t_gt, ypr_gt, B, seed=300 + 7 * rep + k)
            tr, ro = torch.from_numpy(tr).cuda(), torch.from_numpy(ro).cuda()
            want = po.omniloc_batch(img, X, C, tr.clone(), ro.clone(), Cfg(gd_graph=False, num_input=B, **base), {})
            got = po.omniloc_batch(img, X, C, tr.clone(), ro.clone(), Cfg(num_input=B, **base), {})
            assert all(torch.equal(a, b) for a, b in zip(want, got)), (rep, k)
    assert len(po._cache.kinds["gd"]) <= po._CAPACITY["gd"]
