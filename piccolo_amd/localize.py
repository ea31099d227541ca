"""Counterpart of the reference's dataset harness (localize.py): the per-image body — colour preprocessing ->
starting poses -> refinement -> pose error (localize.py:173-258) —, a synthetic-scene driver that needs no dataset, and
the two dataset loops `localize_stanford` / `localize_omniscenes` over the reference's directory layout and CSV format.
The reference's own localize.py also runs unchanged on top of piccolo_amd's modules; the loops here
exist so that `main.py` works without OpenCV / TensorBoard (images through PIL, an optional writer) and shards the query
images over the ranks of a process group.
"""
import csv
import glob
import os
import random
import time

import numpy as np
import torch

from . import data_utils
from . import dist as pdist
from . import ops, synth
from .color_utils import color_match, color_mod
from .omniloc import omniloc_all, omniloc_batch, omniloc_batch_images
from .utils import make_input, make_input_images, make_pano, out_of_room, resize_image, write_summaries


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


# ------------------------------------------------------------------------------------------ dataset harness
def get_init_dict(cfg):
    """localize.py:18-73: the initialisation settings make_input reads, with the reference's defaults."""
    g = lambda k, d: getattr(cfg, k, d)  # noqa: E731
    return {"xy_only": g("xy_only", True), "num_trans": g("num_trans", 50), "yaw_only": g("yaw_only", True),
            "num_yaw": g("num_yaw", 4), "num_pitch": g("num_pitch", 0), "num_roll": g("num_roll", 0),
            "max_yaw": g("max_yaw", 2 * np.pi), "min_yaw": g("min_yaw", 0), "max_pitch": g("max_pitch", 2 * np.pi),
            "min_pitch": g("min_pitch", 0), "max_roll": g("max_roll", 2 * np.pi), "min_roll": g("min_roll", 0),
            "z_prior": g("z_prior", None), "dataset": cfg.dataset, "sample_rate_for_init": g("sample_rate_for_init", None),
            "trans_init_mode": g("trans_init_mode", "quantile"), "x_max": g("x_max", None), "x_min": g("x_min", None),
            "y_max": g("y_max", None), "y_min": g("y_min", None), "z_max": g("z_max", None), "z_min": g("z_min", None),
            "num_split_h": g("num_split_h", 2), "num_split_w": g("num_split_w", 4)}


def read_image(filename):
    """RGB uint8 (H,W,3) array of an image file (the reference: cv2.imread + BGR2RGB, localize.py:167)."""
    from PIL import Image
    with Image.open(filename) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8).copy()


def _to_img(img8, dev):
    # host division, like the reference (localize.py:169); exactly k/255 by construction: tagged, so that packing it never waits
    # for the device-side exactness check
    if img8.dtype != np.uint8:                    # the tag below promises levels k/255: only a decoded 8-bit image keeps it
        raise TypeError("_to_img: expected a uint8 image, got %s" % img8.dtype)
    return synth.mark_levels((torch.from_numpy(img8).float() / 255.).to(dev))


def _fmt(a):
    return str(np.asarray(a).flatten())[1:-1].replace("\n", "")


class _NullWriter:
    def add_text(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass


def _save_result_image(path, gt_img8, xyz, rgb, t, R, resolution):
    """localize.py:264-279: ground-truth panorama stacked over the cloud rendered from the estimated pose."""
    from PIL import Image
    new_xyz = torch.matmul(R.to(xyz.device), (xyz - t.reshape(1, 3).to(xyz.device)).t()).t()
    render = make_pano(new_xyz, rgb, resolution=resolution)
    gt = resize_image(gt_img8, render.shape[1], render.shape[0])
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(np.concatenate([gt, np.asarray(render, np.uint8)], axis=0)).save(path)


def _save_starting_points(dirname, stem, gt_img8, xyz, rgb, input_trans, input_rot):
    """cfg.save_starting_point (localize.py:457-471): for every starting pose of the refinement, the query panorama stacked over
    the cloud rendered from that pose at half the panorama's resolution, as <dirname>/<stem>_<idx>.png."""
    from .utils import rot_from_ypr
    for idx in range(int(input_trans.shape[0])):
        R = rot_from_ypr(input_rot[idx].detach().cpu()).float()
        _save_result_image(os.path.join(dirname, "{}_{}.png".format(stem, idx)), gt_img8, xyz, rgb,
                           input_trans[idx].detach().cpu().float(), R, (gt_img8.shape[0] // 2, gt_img8.shape[1] // 2))


def _require_gravity_aligned(cfg):
    """localize.py:141,155-157 / :355,370-372: with gravity_aligned = False the reference calls data_utils.obtain_align_matrix,
    which its data_utils.py does not define (AttributeError on the first room); refused here before anything is loaded."""
    if not getattr(cfg, "gravity_aligned", True):
        raise NotImplementedError("gravity_aligned = False: the reference's own path stops at the undefined "
                                  "data_utils.obtain_align_matrix (localize.py:155); align the cloud beforehand")


def stanford_success(t_err, r_err):
    """localize.py:250: `(t_error < 0.2) and (r_error < np.rad2deg(0.2))` — 0.2 m and 0.2 rad (11.46 degrees)."""
    return bool(t_err < 0.2 and r_err < np.rad2deg(0.2))


def omniscenes_success(t_err, r_err):
    """localize.py:513: `(t_error < 0.1) and (r_error < 5)` — 0.1 m and 5 degrees."""
    return bool(t_err < 0.1 and r_err < 5)


LAST_RUN = {}          # rank 0's summary of the latest dataset loop: accuracy, failed / skipped file names (what the reference prints)


def write_results(table, gts, filenames, writer, log_dir, csv_name, header, row_prefix, success):
    """Rank 0's tail of the dataset loops (localize.py:250-297 / :513-530): the CSV in the reference's columns, the running
    accuracy under the DATASET's own success rule (`success(t_err, r_err)`), failed / skipped rooms.  Pure host code.
    -> {"accuracy", "well_posed", "total", "failed", "skipped"}."""
    import contextlib
    writer = writer if writer is not None else _NullWriter()
    accuracy, well_posed, total = 0.0, 0, 0
    failed, skipped_list = [], []
    scalar_summaries = {"current_accuracy": []}
    if log_dir is not None:
        os.makedirs(log_dir, exist_ok=True)
    with (open(os.path.join(log_dir, csv_name), "w", encoding="utf-8", newline="") if log_dir is not None
          else contextlib.nullcontext(open(os.devnull, "w"))) as f:
        w = csv.writer(f)
        w.writerow(header)
        for k, row in enumerate(np.asarray(table)):
            gt_t, gt_r, skipped = gts[k]
            if skipped:
                skipped_list.append(filenames[k])
                writer.add_text("skipped rooms", filenames[k])
                w.writerow(row_prefix(filenames[k]) + [_fmt(gt_t), _fmt(gt_r), 1])
                continue
            t_err, r_err = float(row[13]), float(row[14])
            if success(t_err, r_err):
                well_posed += 1
            else:
                failed.append(filenames[k])
                writer.add_text("failed rooms", filenames[k])
            total += 1
            accuracy = well_posed / total
            scalar_summaries["current_accuracy"] = [accuracy]
            write_summaries(writer, scalar_summaries, k)                     # localize.py:295
            w.writerow(row_prefix(filenames[k]) + [_fmt(gt_t), _fmt(gt_r), 0, _fmt(row[0:3]), _fmt(row[3:12]), t_err, r_err,
                                                   float(row[15])])
    writer.add_scalar("final accuracy", accuracy)
    print("Final Accuracy : {}".format(accuracy))
    print("failed {} rooms : {}\n".format(len(failed), failed))
    print("skipped {} rooms : {}".format(len(skipped_list), skipped_list))
    return {"accuracy": accuracy, "well_posed": well_posed, "total": total, "failed": failed, "skipped": skipped_list}


def _run_dataset(cfg, writer, log_dir, filenames, per_image, csv_name, header, row_prefix, success):
    """Shared loop of the two dataset harnesses: shard the query images over the ranks, run `per_image(k)` ->
    (RESULT_WIDTH row, gt_trans, gt_rot, skipped), gather, and let rank 0 write the reference's CSV and the accuracy under
    the dataset's own success rule (`success`: stanford_success / omniscenes_success)."""
    dev = ops.device()
    gts = {}

    def body(k):
        row, gt_t, gt_r, skipped = per_image(k)
        gts[k] = (gt_t, gt_r, skipped)
        return row

    per_launch = int(getattr(cfg, "images_per_launch", 1))
    if per_launch > 1:
        rank, world = pdist.world()
        mine = pdist.shard(len(filenames), rank, world)
        rows = torch.full((len(mine), pdist.RESULT_WIDTH), float("nan"), dtype=torch.float32)
        batcher = _Batcher(cfg, per_launch)
        for j, k in enumerate(mine):
            def store(row, j=j):
                rows[j] = row
            _, gt_t, gt_r, skipped = per_image(k, batcher=batcher, done=store)
            gts[k] = (gt_t, gt_r, skipped)
        batcher.flush()
        table = pdist.gather_rows(rows.to(dev), len(filenames), rank, world)
    else:
        table = pdist.localize_sharded(len(filenames), body, dev)
    rank, world = pdist.world()
    if world > 1 and rank == 0:                     # ground truths of the other ranks' images, for the CSV
        for k in range(len(filenames)):
            if k not in gts:
                gts[k] = per_image(k, gt_only=True)
    LAST_RUN.clear()
    if rank == 0:
        LAST_RUN.update(write_results(table.cpu().numpy(), gts, filenames, writer, log_dir, csv_name, header, row_prefix, success))
    return table


def _seed_all():
    torch.manual_seed(2)                            # localize.py:95-98
    if torch.cuda.is_available():
        torch.cuda.manual_seed(2)
    np.random.seed(2)
    random.seed(2)


def _nan_row():
    return torch.full((pdist.RESULT_WIDTH,), float("nan"))


def _refine_and_score(img_init, img_main, xyz, rgb, cfg, gt_trans, gt_rot, summaries, batcher=None, finish=None, on_start=None):
    """localize.py:199-247: make_input on the initialisation image, refinement on the main image, errors.  With a
    `batcher` BOTH stages are deferred: images of one cloud are initialised together (make_input_images) and refined together
    (omniloc_batch_images); `finish(t, R, row)` is called then."""
    if batcher is not None:
        batcher.submit(dict(img_init=img_init, img=img_main, xyz=xyz, rgb=rgb, gt=(gt_trans, gt_rot), finish=finish, on_start=on_start))
        return None
    init_dict = get_init_dict(cfg)
    torch.cuda.synchronize()
    t0 = time.time()
    input_trans, input_rot = make_input(img_init, xyz, rgb, getattr(cfg, "num_input", 6), init_dict,
                                        getattr(cfg, "criterion", "histogram"), getattr(cfg, "num_intermediate", 20))
    if on_start is not None:
        on_start(input_trans, input_rot)
    t, R, loss = refine_image(img_main, xyz, rgb, input_trans, input_rot, cfg, summaries)
    dt = time.time() - t0
    return (t, R) + (_result_row(t, R, loss, gt_trans, gt_rot, dt),)


def _result_row(t, R, loss, gt_trans, gt_rot, seconds):
    t_err, r_err = pose_errors(t, R, gt_trans, gt_rot)
    return torch.cat([t.reshape(3), R.reshape(9), loss.reshape(1), torch.tensor([t_err, r_err, seconds], dtype=torch.float32)])


class _Batcher:
    """cfg.images_per_launch > 1: query images that share the cloud tensors and the image size are refined in one launch
    chain (omniloc_batch_images) — at the shipped 6 candidates per image a launch is latency-bound, eight images cost
    little more than one.  Images whose cloud colours were changed per image (sharpen_color) cannot share a launch and
    go one by one."""

    def __init__(self, cfg, size):
        self.cfg, self.size, self.jobs = cfg, size, []

    def submit(self, job):
        if self.jobs and not (job["xyz"] is self.jobs[0]["xyz"] and job["rgb"] is self.jobs[0]["rgb"] and
                              job["img"].shape == self.jobs[0]["img"].shape and job["img_init"].shape == self.jobs[0]["img_init"].shape):
            self.flush()
        self.jobs.append(job)
        if len(self.jobs) >= self.size:
            self.flush()

    def flush(self):
        jobs, self.jobs = self.jobs, []
        if not jobs:
            return
        cfg = self.cfg
        torch.cuda.synchronize()
        t0 = time.time()
        starts = make_input_images([j["img_init"] for j in jobs], jobs[0]["xyz"], jobs[0]["rgb"], getattr(cfg, "num_input", 6), get_init_dict(cfg),
                                   getattr(cfg, "criterion", "histogram"), getattr(cfg, "num_intermediate", 20))
        for j, (tr, ro) in zip(jobs, starts):
            j["trans"], j["rot"] = tr, ro
            if j["on_start"] is not None:
                j["on_start"](tr, ro)
        if len(jobs) == 1:
            j = jobs[0]
            results = [refine_image(j["img"], j["xyz"], j["rgb"], j["trans"], j["rot"], cfg)]
        else:
            results = omniloc_batch_images([j["img"] for j in jobs], jobs[0]["xyz"], jobs[0]["rgb"], [j["trans"] for j in jobs],
                                           [j["rot"] for j in jobs], cfg, batch_mode=bool(getattr(cfg, "parallel", False)))
        share = (time.time() - t0) / len(jobs)               # the group's wall time, shared equally (localize.py:208,222-223 per image)
        for j, (t, R, loss) in zip(jobs, results):
            j["finish"](t, R, _result_row(t, R, loss, j["gt"][0], j["gt"][1], share))


def localize_stanford(cfg, writer=None, log_dir="./log", root="./data/stanford"):
    """Stanford2D-3D-S loop (localize.py:76-297) over `root`/pano/area_*/ *.png, pcd_not_aligned/area_*/<room>.txt and
    pose/area_*/ *.json; writes `stanford_results.csv` with the reference's columns and result images under results/."""
    _require_gravity_aligned(cfg)
    _seed_all()
    dev = ops.device()
    area_num = getattr(cfg, "area", None)
    # the reference's sort key (room type, room number) leaves the cameras of one room in glob order; names break the tie here
    key = lambda x: (x.split("/")[-1].split("_")[2], int(x.split("/")[-1].split("_")[3]))  # noqa: E731
    if area_num is not None:
        areas = area_num if isinstance(area_num, list) else [area_num]
        filenames = []
        for a in areas:
            filenames += sorted(sorted(glob.glob(os.path.join(root, "pano/area_{}/*.png".format(a)))), key=key)
    else:
        filenames = sorted(sorted(glob.glob(os.path.join(root, "pano/area_*/*.png"))),
                           key=lambda x: (int(x.split("/")[-2].replace("area_", "")),) + key(x))
    room_name = getattr(cfg, "room_name", None)
    if room_name is not None:
        filenames = [f for f in filenames if room_name in f]
    sample_rate = getattr(cfg, "sample_rate", 1)
    quant = getattr(cfg, "out_of_room_quantile", 0.05)
    dh, dw = getattr(cfg, "init_downsample_h", 1), getattr(cfg, "init_downsample_w", 1)
    mh, mw = getattr(cfg, "main_downsample_h", 1), getattr(cfg, "main_downsample_w", 1)
    cache = {}
    summaries = {}

    def per_image(k, gt_only=False, batcher=None, done=None):
        filename = filenames[k]
        area = int(filename.split("/")[-2].split("_")[-1])
        img_name = filename.split("/")[-1]
        room_type, room_no = img_name.split("_")[2], img_name.split("_")[3]
        gt_trans, gt_rot = data_utils.obtain_gt_stanford(area, img_name, root=os.path.join(root, "pose"))
        gt_trans, gt_rot = gt_trans.astype(np.float32), gt_rot.astype(np.float32)
        pcd_name = os.path.join(root, "pcd_not_aligned/area_{}/{}_{}.txt".format(area, room_type, room_no))
        if cache.get("name") != pcd_name:
            xyz_np, rgb_np = data_utils.read_stanford(pcd_name, sample_rate)
            cache.update(name=pcd_name, xyz=torch.from_numpy(xyz_np).float().to(dev), rgb=torch.from_numpy(rgb_np).float().to(dev))
        xyz, rgb = cache["xyz"], cache["rgb"]
        skipped = bool(out_of_room(xyz, torch.from_numpy(gt_trans), quant)) and not getattr(cfg, "eval_full", False)
        if gt_only:
            return gt_trans, gt_rot, skipped
        if skipped:
            print("corrupted file : {}, gt_trans is out of the room\n".format(filename))
            return _nan_row(), gt_trans, gt_rot, True
        orig = read_image(filename)
        img = _to_img(resize_image(orig, orig.shape[1] // dw, orig.shape[0] // dh), dev)
        rgb_k = rgb
        if getattr(cfg, "sharpen_color", False):        # localize.py:175-179: only the INITIALISATION image is equalised
            img, rgb_k = color_mod(img, rgb, int(getattr(cfg, "num_bins", 256)))
        img_main = _to_img(resize_image(orig, orig.shape[1] // mw, orig.shape[0] // mh), dev)      # localize.py:211-213
        def report(t, R, row):
            print("\n{}\ntranslation error : {}\nrotation error : {}\n".format(img_name, float(row[13]), float(row[14])))
            if log_dir is not None:
                _save_result_image(os.path.join(log_dir, "results", "area_{}".format(area), img_name), orig, xyz, rgb, t, R,
                                   (img_main.shape[0] // 2, img_main.shape[1] // 2))
            if done is not None:
                done(row)

        if batcher is not None:
            _refine_and_score(img, img_main, xyz, rgb_k, cfg, gt_trans, gt_rot, summaries, batcher=batcher, finish=report)
            return None, gt_trans, gt_rot, False
        t, R, row = _refine_and_score(img, img_main, xyz, rgb_k, cfg, gt_trans, gt_rot, summaries)
        report(t, R, row)
        return row, gt_trans, gt_rot, False

    return _run_dataset(cfg, writer, log_dir, filenames, per_image, "stanford_results.csv",
                        ["area_num", "pano_name", "gt_trans", "gt_rot", "skipped?", "OmniLoc_trans", "OmniLoc_rot", "t_error (m)",
                         "r_error (degrees)", "time (s)"],
                        lambda f: [int(f.split("/")[-2].split("_")[-1]), f.split("/")[-1]], stanford_success)


def localize_omniscenes(cfg, writer=None, log_dir="./log", root="./data/omniscenes"):
    """OmniScenes loop (localize.py:300-530) over `root`/<split>_pano/<video>/<frame>, pcd/<room>.txt and <split>_pose;
    writes `omniscenes_results.csv`.  Includes the synthetic illumination changes (synth_const / synth_gamma / synth_wb)
    and the colour preprocessing of the whole image (match_color / sharpen_color)."""
    _require_gravity_aligned(cfg)
    _seed_all()
    dev = ops.device()
    split = getattr(cfg, "split_name", "extreme")
    filenames = sorted(glob.glob(os.path.join(root, "{}_pano/*/*".format(split))))
    room_name, scene = getattr(cfg, "room_name", None), getattr(cfg, "scene_number", None)
    if isinstance(room_name, str):
        filenames = [f for f in filenames if room_name in f]
    elif isinstance(room_name, list):
        filenames = [f for f in filenames if any(rm in f for rm in room_name)]
    if scene is not None:
        filenames = [f for f in filenames if "scene_{}".format(scene) in f]
    sample_rate = getattr(cfg, "sample_rate", 1)
    quant = getattr(cfg, "out_of_room_quantile", 0.05)
    dh = max(getattr(cfg, "init_downsample_h", 1) // 2, 1)         # "match resolution with stanford" (localize.py:349-350)
    dw = max(getattr(cfg, "init_downsample_w", 1) // 2, 1)
    mh, mw = getattr(cfg, "main_downsample_h", 1), getattr(cfg, "main_downsample_w", 1)
    cache = {}
    summaries = {}

    def per_image(k, gt_only=False, batcher=None, done=None):
        filename = filenames[k]
        video = filename.split("/")[-2]
        room_type, room_no = video.split("_")[1], video.split("_")[2]
        gt_trans, gt_rot = data_utils.obtain_gt_omniscenes(filename)
        gt_trans, gt_rot = gt_trans.astype(np.float32), gt_rot.astype(np.float32)
        pcd_name = os.path.join(root, "pcd/{}_{}.txt".format(room_type, room_no))
        if cache.get("name") != pcd_name:
            xyz_np, rgb_np = data_utils.read_omniscenes(pcd_name, sample_rate)
            cache.update(name=pcd_name, xyz=torch.from_numpy(xyz_np).float().to(dev), rgb=torch.from_numpy(rgb_np).float().to(dev))
        xyz, rgb = cache["xyz"], cache["rgb"]
        skipped = bool(out_of_room(xyz, torch.from_numpy(gt_trans), quant))
        if gt_only:
            return gt_trans, gt_rot, skipped
        if skipped:
            print("corrupted file : {}, gt_trans is out of the room\n".format(filename))
            return _nan_row(), gt_trans, gt_rot, True
        orig = resize_image(read_image(filename), 2048, 1024)       # localize.py:372
        if getattr(cfg, "synth_const", None) is not None:           # synthetic illumination changes, localize.py:375-385
            orig = orig // cfg.synth_const
        if getattr(cfg, "synth_gamma", None) is not None:
            orig = (((orig / 255.) ** cfg.synth_gamma) * 255).astype(np.uint8)
        if getattr(cfg, "synth_wb", None):
            for c, gain in enumerate((cfg.synth_r, cfg.synth_g, cfg.synth_b)):
                orig[..., c] = (((orig[..., c] / 255.) * gain) * 255).astype(np.uint8)
        new_img, rgb_k = preprocess_colors(_to_img(orig, dev), rgb, cfg)
        orig = (255 * new_img.cpu().numpy()).astype(np.uint8)
        img = _to_img(resize_image(orig, orig.shape[1] // dw, orig.shape[0] // dh), dev)
        img_main = _to_img(resize_image(orig, orig.shape[1] // mw, orig.shape[0] // mh), dev)
        def report(t, R, row):
            print("\n{}/{}\ntranslation error : {}\nrotation error : {}\n".format(video, filename.split("/")[-1], float(row[13]), float(row[14])))
            if log_dir is not None:
                _save_result_image(os.path.join(log_dir, "results", video, os.path.splitext(filename.split("/")[-1])[0] + ".png"), orig, xyz,
                                   rgb_k, t, R, (img_main.shape[0] // 2, img_main.shape[1] // 2))
            if done is not None:
                done(row)

        on_start = None
        if getattr(cfg, "save_starting_point", False) and log_dir is not None:
            def on_start(input_trans, input_rot):
                _save_starting_points(os.path.join(log_dir, "starting_points", video), os.path.splitext(filename.split("/")[-1])[0],
                                      orig, xyz, rgb_k, input_trans, input_rot)
        if batcher is not None:
            _refine_and_score(img, img_main, xyz, rgb_k, cfg, gt_trans, gt_rot, summaries, batcher=batcher, finish=report,
                              on_start=on_start)
            return None, gt_trans, gt_rot, False
        t, R, row = _refine_and_score(img, img_main, xyz, rgb_k, cfg, gt_trans, gt_rot, summaries, on_start=on_start)
        report(t, R, row)
        return row, gt_trans, gt_rot, False

    return _run_dataset(cfg, writer, log_dir, filenames, per_image, "omniscenes_results.csv",
                        ["pano_name", "gt_trans", "gt_rot", "skipped?", "OmniLoc_trans", "OmniLoc_rot", "t_error (m)", "r_error (degrees)",
                         "time (s)"],
                        lambda f: ["{}/{}".format(f.split("/")[-2], f.split("/")[-1])], omniscenes_success)
