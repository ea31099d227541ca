#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE (imported from /root/reference).

Runs only in the build container (the reference does not exist on the GPU box);
the outputs are committed as small .npz/.json fixtures next to this script.
The reference needs three modules this image lacks (cv2, torch_scatter, open3d):
none is called on the paths captured here, so empty stubs are enough
(SURVEY.md §8c).  Nothing from the reference is copied: the fixtures hold only
inputs and the outputs the reference computed for them.

    python tests/golden/gen_goldens.py            # rewrites tests/golden/*.npz, *.json
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

for name in ("cv2", "torch_scatter", "open3d"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["torch_scatter"].scatter_min = None
sys.path.insert(0, REF)
sys.path.insert(1, REPO)

import torch  # noqa: E402

torch.set_num_threads(4)

import utils as ref_utils  # noqa: E402  (the reference's)
import omniloc as ref_omniloc  # noqa: E402
import parse_utils as ref_parse  # noqa: E402
from torch.optim.lr_scheduler import ReduceLROnPlateau  # noqa: E402

from piccolo_amd import synth  # noqa: E402


class Cfg:
    """Attribute bag standing in for the namedtuple parse_ini returns."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", name, {k: np.asarray(v).shape for k, v in arrays.items()})


def small_scene(n=4096, H=64, W=128, seed=3, black_patch=True):
    """Box room + a panorama rendered by the reference's make_pano from a GT pose."""
    xyz, rgb = synth.box_room(n, seed)
    t_gt, ypr_gt = synth.gt_pose(seed)
    cam = synth.transform_cloud(xyz, t_gt, ypr_gt)
    img = ref_utils.make_pano(torch.from_numpy(cam), torch.from_numpy(rgb), resolution=(H, W))
    img = img.astype(np.float32) / 255.0
    if black_patch:
        img[H // 4:H // 4 + 6, W // 3:W // 3 + 9, :] = 0.0
    return xyz, rgb, img, t_gt, ypr_gt


# ----------------------------------------------------------------------------- G1
def g1_cloud2idx():
    rng = np.random.default_rng(1)
    pts = rng.normal(0, 2.0, size=(1000, 3)).astype(np.float32)
    special = np.array([
        [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1],
        [0, 0, 0], [-1, 1e-7, 0], [-1, -1e-7, 0], [1e-3, 0, 5], [0, 2, 0], [3, 0, 0],
        [-1e-6, 0, 1], [-1e-6, 0, -1e-6], [2, -2, 0], [-2, -2, 0.5],
    ], dtype=np.float32)
    pts = np.concatenate([special, pts], 0)
    out = ref_utils.cloud2idx(torch.from_numpy(pts)).numpy()
    out64 = ref_utils.cloud2idx(torch.from_numpy(pts).double()).numpy()
    ptsb = np.stack([pts, pts[::-1].copy(), pts * 0.5], 0)
    outb = ref_utils.cloud2idx(torch.from_numpy(ptsb), batched=True).numpy()
    save("g1_cloud2idx.npz", xyz=pts, coord=out, coord_f64=out64, xyz_b=ptsb, coord_b=outb)


# ----------------------------------------------------------------------------- G2
def g2_sample_from_img():
    rng = np.random.default_rng(2)
    H, W = 16, 32
    img = (rng.integers(0, 256, size=(H, W, 3)) / 255.0).astype(np.float32)
    img[4:8, 10:15, :] = 0.0
    # pixel centres, edges, beyond the +-0.99 clip, random
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    centres = np.stack([(2 * xs.ravel() + 1) / W - 1, (2 * ys.ravel() + 1) / H - 1], -1)
    edges = np.array([[-1, -1], [1, 1], [-1, 1], [1, -1], [-0.99, 0.99], [0.99, -0.99],
                      [-0.995, 0], [0.995, 0], [0, -1.5], [0, 1.5], [0, 0], [-0.99, -0.99]])
    rand = rng.uniform(-1.05, 1.05, size=(700, 2))
    coord = np.concatenate([centres, edges, rand], 0).astype(np.float32)
    out = ref_utils.sample_from_img(torch.from_numpy(img), torch.from_numpy(coord)).numpy()
    out64 = ref_utils.sample_from_img(torch.from_numpy(img).double(), torch.from_numpy(coord).double()).numpy()
    coordb = np.stack([coord, coord[::-1].copy()], 0)
    outb = ref_utils.sample_from_img(torch.from_numpy(img), torch.from_numpy(coordb), batched=True).numpy()
    save("g2_sample_from_img.npz", img=img, coord=coord, rgb=out, rgb_f64=out64, coord_b=coordb, rgb_b=outb)


# ----------------------------------------------------------------------------- G3 / G4
def loss_and_grads(xyz, rgb, img, trans, rot, dtype):
    """Reference SamplingLoss + autograd for each pose row: loss (B,), grad_t (B,3), grad_ypr (B,3), count (B,)."""
    x = torch.from_numpy(xyz).to(dtype)
    c = torch.from_numpy(rgb).to(dtype)
    im = torch.from_numpy(img).to(dtype)
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        mod = ref_omniloc.SamplingLoss(x, c, im, x.device, Cfg())
        losses, gts, grs = [], [], []
        for b in range(trans.shape[0]):
            t = torch.tensor(trans[b], dtype=dtype).reshape(3, 1).requires_grad_()
            yaw = torch.tensor([rot[b, 0]], dtype=dtype).requires_grad_()
            pitch = torch.tensor([rot[b, 1]], dtype=dtype).requires_grad_()
            roll = torch.tensor([rot[b, 2]], dtype=dtype).requires_grad_()
            loss = mod(t, yaw, pitch, roll)
            loss.backward()
            losses.append(loss.item())
            gts.append(t.grad.reshape(3).numpy().copy())
            grs.append(np.array([yaw.grad.item(), pitch.grad.item(), roll.grad.item()]))
    finally:
        torch.set_default_dtype(old)
    return np.array(losses), np.stack(gts), np.stack(grs)


def g3_g4_loss_grad():
    xyz, rgb, img, t_gt, ypr_gt = small_scene()
    trans, rot = synth.start_poses(t_gt, ypr_gt, 6, seed=5)
    trans[0], rot[0] = t_gt, ypr_gt  # the exact GT pose too
    l32, gt32, gr32 = loss_and_grads(xyz, rgb, img, trans, rot, torch.float32)
    l64, gt64, gr64 = loss_and_grads(xyz, rgb, img, trans, rot, torch.float64)
    save("g3_sampling_loss.npz", xyz=xyz, rgb=rgb, img=img, trans=trans, rot=rot,
         loss_f32=l32, grad_t_f32=gt32, grad_ypr_f32=gr32,
         loss_f64=l64, grad_t_f64=gt64, grad_ypr_f64=gr64)

    # G4: BatchSamplingLoss, B=4 (same scene, first 4 poses)
    B = 4
    res = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            mod = ref_omniloc.BatchSamplingLoss(torch.from_numpy(xyz).to(dtype), torch.from_numpy(rgb).to(dtype),
                                                torch.from_numpy(img).to(dtype), torch.device("cpu"), Cfg(num_input=B))
            t = torch.tensor(trans[:B], dtype=dtype).unsqueeze(-1).requires_grad_()
            yaw = torch.tensor(rot[:B, 0:1], dtype=dtype).requires_grad_()
            pitch = torch.tensor(rot[:B, 1:2], dtype=dtype).requires_grad_()
            roll = torch.tensor(rot[:B, 2:3], dtype=dtype).requires_grad_()
            loss, loss_list = mod(t, yaw, pitch, roll)
            loss.backward()
            res["loss_" + tag] = loss.item()
            res["loss_list_" + tag] = loss_list.detach().numpy()
            res["grad_t_" + tag] = t.grad.squeeze(-1).numpy()
            res["grad_ypr_" + tag] = torch.cat([yaw.grad, pitch.grad, roll.grad], 1).numpy()
        finally:
            torch.set_default_dtype(old)
    save("g4_batch_sampling_loss.npz", trans=trans[:B], rot=rot[:B], **res)  # scene: g3 file


# ----------------------------------------------------------------------------- G5
class Recorder:
    """Wraps Adam.step / ReduceLROnPlateau.step / loss forwards to capture per-iteration state."""

    def __init__(self):
        self.opt_ids = {}
        self.adam = []   # (opt_idx, lr, [param_before...], [grad...], [param_after...])
        self.sched = []  # (opt_idx, metric, num_bad_after, best_after, lr_after)
        self.fwd = []    # (t, yaw, pitch, roll) as seen by the loss forward, + output loss(es)

    def install(self):
        rec = self
        self._adam_step = torch.optim.Adam.step
        self._sched_step = ReduceLROnPlateau.step
        self._fwd_s = ref_omniloc.SamplingLoss.forward
        self._fwd_b = ref_omniloc.BatchSamplingLoss.forward

        def adam_step(opt, *a, **k):
            idx = rec.opt_ids.setdefault(id(opt), len(rec.opt_ids))
            ps = opt.param_groups[0]["params"]
            before = [p.detach().clone().reshape(-1).numpy() for p in ps]
            grads = [p.grad.detach().clone().reshape(-1).numpy() for p in ps]
            lr = float(opt.param_groups[0]["lr"])
            out = rec._adam_step(opt, *a, **k)
            after = [p.detach().clone().reshape(-1).numpy() for p in ps]
            rec.adam.append((idx, lr, np.concatenate(before), np.concatenate(grads), np.concatenate(after)))
            return out

        def sched_step(s, metrics, *a, **k):
            idx = rec.opt_ids.setdefault(id(s.optimizer), len(rec.opt_ids))
            out = rec._sched_step(s, metrics, *a, **k)
            rec.sched.append((idx, float(metrics), int(s.num_bad_epochs), float(s.best),
                              float(s.optimizer.param_groups[0]["lr"])))
            return out

        def fwd_s(m, translation, yaw, pitch, roll):
            out = rec._fwd_s(m, translation, yaw, pitch, roll)
            rec.fwd.append((translation.detach().reshape(1, 3).numpy().copy(),
                            np.array([[yaw.item(), pitch.item(), roll.item()]]), np.array([out.item()])))
            return out

        def fwd_b(m, translation, yaw, pitch, roll):
            out = rec._fwd_b(m, translation, yaw, pitch, roll)
            rec.fwd.append((translation.detach().squeeze(-1).numpy().copy(),
                            torch.cat([yaw, pitch, roll], 1).detach().numpy().copy(),
                            out[1].detach().numpy().copy()))
            return out

        torch.optim.Adam.step = adam_step
        ReduceLROnPlateau.step = sched_step
        ref_omniloc.SamplingLoss.forward = fwd_s
        ref_omniloc.BatchSamplingLoss.forward = fwd_b

    def remove(self):
        torch.optim.Adam.step = self._adam_step
        ReduceLROnPlateau.step = self._sched_step
        ref_omniloc.SamplingLoss.forward = self._fwd_s
        ref_omniloc.BatchSamplingLoss.forward = self._fwd_b

    def arrays(self, prefix):
        B = len(self.opt_ids)
        n_it = len(self.adam) // B
        d = {}
        # Adam param order in the reference is [translation(3), yaw, roll, pitch] (omniloc.py:33,235-236)
        d["adam_lr"] = np.array([a[1] for a in self.adam]).reshape(n_it, B)
        d["adam_param_before"] = np.stack([a[2] for a in self.adam]).reshape(n_it, B, 6)
        d["adam_grad"] = np.stack([a[3] for a in self.adam]).reshape(n_it, B, 6)
        d["adam_param_after"] = np.stack([a[4] for a in self.adam]).reshape(n_it, B, 6)
        d["sched_metric"] = np.array([s[1] for s in self.sched]).reshape(n_it, B)
        d["sched_num_bad"] = np.array([s[2] for s in self.sched]).reshape(n_it, B)
        d["sched_best"] = np.array([s[3] for s in self.sched]).reshape(n_it, B)
        d["sched_lr"] = np.array([s[4] for s in self.sched]).reshape(n_it, B)
        d["fwd_trans"] = np.stack([f[0] for f in self.fwd])   # (n_it, B, 3)
        d["fwd_rot"] = np.stack([f[1] for f in self.fwd])     # (n_it, B, 3) [yaw, pitch, roll]
        d["fwd_loss"] = np.stack([f[2] for f in self.fwd])    # (n_it, B)
        return {prefix + k: v for k, v in d.items()}


def g5_trajectories():
    xyz, rgb, img, t_gt, ypr_gt = small_scene()
    cfg = Cfg(lr=0.1, num_iter=100, patience=5, factor=0.8, out_of_room_quantile=0.05, visualize=False, num_input=4)
    trans, rot = synth.start_poses(t_gt, ypr_gt, 4, seed=7)
    # one start outside the 5-95 % box in x (the box equals the room: >5 % of the points lie on each wall)
    trans[1, 0] = 4.4
    out = dict(xyz=xyz, rgb=rgb, img=img, trans0=trans.copy(), rot0=rot.copy(), t_gt=t_gt, ypr_gt=ypr_gt,
               cfg=json.dumps(cfg.__dict__))

    # sequential omniloc, starting points 0 and 1
    for sp in (0, 1):
        rec = Recorder()
        rec.install()
        try:
            it, ir = torch.from_numpy(trans.copy()), torch.from_numpy(rot.copy())
            res = ref_omniloc.omniloc(torch.from_numpy(img), torch.from_numpy(xyz), torch.from_numpy(rgb),
                                      it, ir, sp, cfg, {})
        finally:
            rec.remove()
        out.update(rec.arrays("seq%d_" % sp))
        out["seq%d_ret_t" % sp] = res[0].detach().numpy()
        out["seq%d_ret_R" % sp] = res[1].detach().numpy()
        out["seq%d_ret_loss" % sp] = res[2].detach().numpy()
        out["seq%d_input_trans_after" % sp] = it.detach().numpy()
        out["seq%d_input_rot_after" % sp] = ir.detach().numpy()

    # batched omniloc_batch over all 4 starts, plus short runs (1 and 2 iterations) that expose the clamp lag
    for n_it, tag in ((100, "bat_"), (1, "bat1_"), (2, "bat2_")):
        cfg_b = Cfg(**{**cfg.__dict__, "num_iter": n_it})
        rec = Recorder()
        rec.install()
        try:
            it, ir = torch.from_numpy(trans.copy()), torch.from_numpy(rot.copy())
            res = ref_omniloc.omniloc_batch(torch.from_numpy(img), torch.from_numpy(xyz), torch.from_numpy(rgb),
                                            it, ir, cfg_b, {})
        finally:
            rec.remove()
        out.update(rec.arrays(tag))
        out[tag + "ret_t"] = res[0].detach().numpy()
        out[tag + "ret_R"] = res[1].detach().numpy()
        out[tag + "ret_loss"] = res[2].detach().numpy()
        out[tag + "input_trans_after"] = it.detach().numpy()
        out[tag + "input_rot_after"] = ir.detach().numpy()
    save("g5_trajectories.npz", **out)


# ----------------------------------------------------------------------------- G6
def g6_quantile():
    rng = np.random.default_rng(6)
    out = {}
    for n in (1, 2, 19, 20, 21, 1000, 1001, 4096):
        x = rng.normal(size=n).astype(np.float32)
        for q in (0.05, 0.1, 0.25):
            lo, hi = ref_utils.quantile(torch.from_numpy(x), q)
            out["x_%d" % n] = x
            out["q_%d_%g" % (n, q)] = np.array([lo.item(), hi.item()], dtype=np.float32)
    save("g6_quantile.npz", **out)


# ----------------------------------------------------------------------------- G7
def g7_trim_input_loss():
    xyz, rgb, img, t_gt, ypr_gt = small_scene(n=2048)
    rng = np.random.default_rng(8)
    trans = (t_gt[None] + rng.normal(0, 0.5, size=(5, 3))).astype(np.float32)
    rot = np.zeros((4, 3), np.float32)
    rot[:, 0] = ypr_gt[0] + np.array([0.0, 0.4, -0.4, np.pi])
    rot[:, 1:] = ypr_gt[1:]
    # capture the full loss table by patching argsort-free: recompute exactly as the reference loop does
    tb = torch.zeros(5, 4)
    X, C, I = torch.from_numpy(xyz), torch.from_numpy(rgb), torch.from_numpy(img)
    for i in range(5):
        for j in range(4):
            R = ref_utils.rot_from_ypr(torch.from_numpy(rot[j]))
            p = (torch.matmul(R, X.t() - torch.from_numpy(trans[i]).reshape(3, -1))).t()
            s = ref_utils.sample_from_img(I, ref_utils.cloud2idx(p))
            m = torch.sum(s == 0, dim=1) != 3
            tb[i, j] = torch.norm(s[m] - C[m], dim=-1).mean()
    tt, tr = ref_utils.trim_input_loss(I, X, C, torch.from_numpy(trans), torch.from_numpy(rot), 7)
    save("g7_trim_input_loss.npz", xyz=xyz, rgb=rgb, img=img, trans=trans, rot=rot, loss_table=tb.numpy(),
         trimmed_trans=tt.numpy(), trimmed_rot=tr.numpy())


# ----------------------------------------------------------------------------- G8
def g8_make_pano():
    xyz, rgb = synth.box_room(3000, seed=11)
    t_gt, ypr_gt = synth.gt_pose(11)
    cam = synth.transform_cloud(xyz, t_gt, ypr_gt)
    H, W = 48, 96
    img = ref_utils.make_pano(torch.from_numpy(cam), torch.from_numpy(rgb), resolution=(H, W))
    imgf = ref_utils.make_pano(torch.from_numpy(cam), torch.from_numpy(rgb), resolution=(H, W), return_torch=True).numpy()
    save("g8_make_pano.npz", xyz_cam=cam, rgb=rgb, pano_u8=img, pano_f32=imgf, resolution=np.array([H, W]))


# ----------------------------------------------------------------------------- G9
def g9_parse():
    out = {}
    for name in ("stanford.ini", "stanford_parallel.ini", "omniscenes.ini"):
        cfg = ref_parse.parse_ini(os.path.join(REF, "configs", name))
        out[name] = cfg._asdict()
    vals = ["1", "0.1", "1e-3", "-2", "+3", "True", "False", "true", "None", "1,2,3", "a,b", "abc", "1.5,2.5",
            "4.", "1e5", "-1e-2", "area_1", "3,"]
    out["parse_value"] = {v: ref_parse.parse_value(v) for v in vals}
    with open(os.path.join(HERE, "g9_parse.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote g9_parse.json")


# ----------------------------------------------------------------------------- G10
def g10_candidates():
    xyz, _ = synth.box_room(20000, seed=12)
    X = torch.from_numpy(xyz)
    out = {"xyz": xyz}
    base = dict(max_yaw=2 * np.pi, min_yaw=0, max_pitch=2 * np.pi, min_pitch=0, max_roll=2 * np.pi, min_roll=0,
                z_prior=None, sample_rate_for_init=None, trans_init_mode="quantile",
                x_max=None, x_min=None, y_max=None, y_min=None, z_max=None, z_min=None, num_split_h=4, num_split_w=4)
    cases = {
        "stanford": dict(base, xy_only=False, num_trans=50, yaw_only=False, num_yaw=4, num_pitch=4, num_roll=4,
                         dataset="Stanford2D-3D-S"),
        "omniscenes": dict(base, xy_only=True, num_trans=150, yaw_only=True, num_yaw=8, num_pitch=8, num_roll=8,
                           dataset="OmniScenes", z_prior=1.5),
    }
    for tag, d in cases.items():
        rot = ref_utils.generate_rot_points(d)
        tr = ref_utils.generate_trans_points(X, d)
        out[tag + "_rot"] = rot.numpy()
        out[tag + "_trans"] = tr.numpy()
        print(tag, "rot", tuple(rot.shape), "trans", tuple(tr.shape))
    save("g10_candidates.npz", **out)


# ----------------------------------------------------------------------------- G11
def g11_end_to_end():
    """cfg-1 shape but smaller so the fixture stays small: N=20k, 128x256, one candidate, sequential, 100 iters."""
    N, H, W = 20000, 128, 256
    xyz, rgb = synth.box_room(N, seed=21)
    t_gt, ypr_gt = synth.gt_pose(21)
    cam = synth.transform_cloud(xyz, t_gt, ypr_gt)
    img = ref_utils.make_pano(torch.from_numpy(cam), torch.from_numpy(rgb), resolution=(H, W)).astype(np.float32) / 255.0
    trans, rot = synth.start_poses(t_gt, ypr_gt, 2, seed=21)
    cfg = Cfg(lr=0.1, num_iter=100, patience=5, factor=0.8, out_of_room_quantile=0.05, visualize=False, num_input=2)
    it, ir = torch.from_numpy(trans.copy()), torch.from_numpy(rot.copy())
    res = ref_omniloc.omniloc(torch.from_numpy(img), torch.from_numpy(xyz), torch.from_numpy(rgb), it, ir, 0, cfg, {})
    t, R = res[0].detach().numpy(), res[1].detach().numpy()
    t_err, r_err = synth.pose_errors(t, R, t_gt, synth.rot_from_ypr_np(ypr_gt))
    # self-noise band: the same run with the point order permuted (fp32 summation order changes)
    band = []
    for s in range(3):
        perm = np.random.default_rng(100 + s).permutation(N)
        it2, ir2 = torch.from_numpy(trans.copy()), torch.from_numpy(rot.copy())
        r2 = ref_omniloc.omniloc(torch.from_numpy(img), torch.from_numpy(xyz[perm]), torch.from_numpy(rgb[perm]),
                                 it2, ir2, 0, cfg, {})
        te, re = synth.pose_errors(r2[0].detach().numpy(), r2[1].detach().numpy(), t_gt, synth.rot_from_ypr_np(ypr_gt))
        band.append([te, re, float(np.abs(r2[0].detach().numpy() - t).max())])
    print("G11 t_err %.5f r_err %.4f band %s" % (t_err, r_err, band))
    # the image is regenerated in the test from the seed by the oracle's make_pano and compared with this one
    save("g11_end_to_end.npz", seed=21, N=N, H=H, W=W, img_u8=(img * 255 + 0.5).astype(np.uint8), trans0=trans, rot0=rot,
         t_gt=t_gt, ypr_gt=ypr_gt, ret_t=t, ret_R=R, ret_loss=res[2].detach().numpy(),
         t_err=t_err, r_err=r_err, self_noise=np.array(band))


# ----------------------------------------------------------------------------- G12
def g12_trim_input_hist():
    """Second trimming stage (utils.py:510-588) on a small scene: per-candidate histogram-intersection scores and the
    selected candidates.  The scores are captured by re-running the reference's own loop body (make_pano + color_utils
    histogram / histogram_intersection); the selection by calling the function itself."""
    import color_utils as ref_color
    xyz, rgb, img, t_gt, ypr_gt = small_scene(n=6000, H=64, W=128, seed=9, black_patch=True)
    rng = np.random.default_rng(12)
    K = 10
    trans = (t_gt[None] + rng.normal(0, 0.6, size=(K, 3))).astype(np.float32)
    rot = (ypr_gt[None] + rng.normal(0, 0.5, size=(K, 3))).astype(np.float32)
    trans[0], rot[0] = t_gt, ypr_gt
    X, C, I = torch.from_numpy(xyz), torch.from_numpy(rgb), torch.from_numpy(img)
    nh, nw = 4, 4
    sel_t, sel_r = ref_utils.trim_input_hist_secondary(I, X, C, torch.from_numpy(trans), torch.from_numpy(rot), 4, nh, nw)
    # scores, block by block, with the reference's own primitives
    img255 = I.clone() * 255
    H, W, _ = img255.shape
    img_mask = torch.zeros([H, W], dtype=torch.bool)
    img_mask[torch.sum(img255 == 0, dim=2) != 3] = True
    bh, bw = H // nh, W // nw
    scores = np.zeros(K)
    inter_all = np.zeros((K, nh * nw))
    for i in range(K):
        R = ref_utils.rot_from_ypr(torch.from_numpy(rot[i]))
        cam = torch.transpose(torch.matmul(R, torch.transpose(X - torch.from_numpy(trans[i]), 0, 1)), 0, 1)
        proj = ref_utils.make_pano(cam, C, resolution=(H, W), return_torch=True)
        proj_mask = torch.zeros([H, W], dtype=torch.bool)
        proj_mask[torch.sum(proj == 0, dim=2) != 3] = True
        for h in range(1, nh - 1):
            for w in range(nw):
                block = torch.zeros([H, W], dtype=torch.bool)
                block[h * bh:(h + 1) * bh, w * bw:(w + 1) * bw] = True
                fm = proj_mask & img_mask & block
                fim = img_mask & block
                if fm.sum() == 0 or fim.sum() == 0:
                    continue
                ph = ref_color.histogram(proj, fm, [8, 8, 8])
                ih = ref_color.histogram(img255, fim, [8, 8, 8])
                inter_all[i, h * nw + w] = float(ref_color.histogram_intersection(ih, ph))
        scores[i] = np.nan_to_num(inter_all[i]).sum() / (nh * nw)
    save("g12_trim_input_hist.npz", xyz=xyz, rgb=rgb, img=img, trans=trans, rot=rot, scores=scores, inter=inter_all,
         selected_trans=sel_t.numpy(), selected_rot=sel_r.numpy(), num_split=np.array([nh, nw]))
    print("G12 scores", np.round(scores, 4))


def _colour_case(seed, H, W, n, levels, template):
    """A quantised panorama (k/255) with a black patch, whose channels use only `levels`, and point colours that are
    either quantised (k/255, like the datasets' integer RGB columns) or continuous (like the synthetic room)."""
    rng = np.random.default_rng(seed)
    img8 = rng.choice(np.asarray(levels, np.uint8), size=(H, W, 3))
    img8[H // 4:H // 4 + 3, W // 8:W // 2] = 0                  # black (unrendered) pixels stay untouched
    img8[0, :5] = (0, 0, 7)                                     # non-black through one channel only
    img = img8.astype(np.float32) / np.float32(255)
    if template == "quantised":
        rgb = rng.integers(16, 250, size=(n, 3)).astype(np.float32) / np.float32(255)
    else:
        rgb = (rng.random((n, 3)) ** 0.7).astype(np.float32)
    return img, rgb


def g13_color_match():
    """color_utils.color_match (pure torch: runs here).  Cases: every level present / gaps in the levels (pins the
    rank-vs-level indexing of _match_cumulative_cdf), quantised and continuous templates."""
    import color_utils as ref_color
    out = {}
    cases = [("full_q", list(range(0, 200)), "quantised"), ("gaps_q", list(range(3, 256, 5)), "quantised"),
             ("full_c", list(range(0, 256)), "continuous"), ("gaps_c", [0, 1, 2, 40, 41, 90, 200, 255], "continuous")]
    for i, (name, levels, template) in enumerate(cases):
        img, rgb = _colour_case(100 + i, 24, 48, 3000, levels, template)
        res = ref_color.color_match(torch.from_numpy(img.copy()), torch.from_numpy(rgb)).numpy()
        out[name + "_img"], out[name + "_rgb"], out[name + "_out"] = img, rgb, res
    save("g13_color_match.npz", **out)


def g14_color_mod():
    """color_utils.color_mod.  It calls cv2.cvtColor, and OpenCV is absent from this image: the two uint8 conversions
    are injected from oracle/color.py (a restatement of OpenCV's published fixed-point formulas), so this fixture pins
    everything in color_mod EXCEPT those conversions (masking, uint8 truncation, joint histogram, cumulative table,
    write-back).  Stated as such in oracle/color.py and DESIGN.md."""
    import color_utils as ref_color
    from oracle import color as ocolor
    cv2 = sys.modules["cv2"]
    cv2.COLOR_RGB2YCR_CB, cv2.COLOR_YCR_CB2RGB = 36, 38
    cv2.cvtColor = lambda a, code: (ocolor.rgb2ycrcb_u8 if code == 36 else ocolor.ycrcb2rgb_u8)(a)
    ref_color.cv2 = cv2
    out = {}
    for i, (name, bins) in enumerate((("b256", 256), ("b64", 64))):
        img, rgb = _colour_case(200 + i, 24, 48, 3000, list(range(0, 256)), "quantised")
        res_img, res_rgb = ref_color.color_mod(torch.from_numpy(img.copy()), torch.from_numpy(rgb.copy()), bins)
        out[name + "_img"], out[name + "_rgb"] = img, rgb
        out[name + "_out_img"], out[name + "_out_rgb"] = res_img.numpy(), res_rgb.numpy()
        out[name + "_bins"] = np.array(bins)
    save("g14_color_mod.npz", **out)


def g15_data_utils():
    """data_utils.read_stanford / read_omniscenes on a small text cloud written here (integers, decimals, exponents,
    tabs, blank lines, CRLF, long mantissas) and obtain_gt_stanford / obtain_gt_omniscenes on pose files written here.
    The fixture holds the text/JSON inputs and what the reference returned for them."""
    import tempfile
    import data_utils as ref_data
    rng = np.random.default_rng(15)
    lines = []
    for i in range(300):
        x, y, z = rng.normal(0, 5, 3)
        r, g, b = rng.integers(0, 256, 3)
        style = i % 6
        if style == 0:
            lines.append("%.3f %.3f %.3f %d %d %d" % (x, y, z, r, g, b))
        elif style == 1:
            lines.append("%.6f\t%.6f\t%.6f\t%d\t%d\t%d" % (x, y, z, r, g, b))
        elif style == 2:
            lines.append("  %.8e %.8e %.8e %d %d %d  " % (x, y, z, r, g, b))
        elif style == 3:
            lines.append("%.17g %.17g %.17g %d.0 %d.0 %d.0" % (x, y, z, r, g, b))
        elif style == 4:
            lines.append("%d %d %d %d %d %d\r" % (int(x), int(y), int(z), r, g, b))
        else:
            lines.append("%+.4f -0.0 .5 %d %d %d" % (x, r, g, b))
        if i % 50 == 49:
            lines.append("")
    text = "\n".join(lines) + "\n"
    out = {"cloud_txt": np.frombuffer(text.encode(), np.uint8)}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "cloud.txt")
        with open(path, "w", newline="") as f:
            f.write(text)
        xyz, rgb = ref_data.read_stanford(path)
        out["xyz"], out["rgb"] = xyz, rgb
        xyz2, rgb2 = ref_data.read_omniscenes(path)
        assert np.array_equal(xyz, xyz2) and np.array_equal(rgb, rgb2)
        special = ("nan inf -inf 1e-400 12345678901234567.5 0.1000000000000000055511151231257827\n"
                   "9007199254740993 1e22 1e23 4.9e-324 0 -0.0\n"
                   "0.000000000000000000001234567890123456789 1.7976931348623157e308 2.2250738585072014e-308 5e-1 5E+2 00012.50\n")
        spath = os.path.join(tmp, "special.txt")
        with open(spath, "w", newline="") as f:
            f.write(special)
        sx, sr = ref_data.read_stanford(spath)
        assert sx.dtype == np.float64 and sr.dtype == np.float64
        out["special_txt"] = np.frombuffer(special.encode(), np.uint8)
        out["special"] = np.hstack([sx, sr * 255.])            # (the reference divides the last three columns by 255)
        out["special_rgb"] = sr
        np.random.seed(7)
        xs, rs = ref_data.read_stanford(path, sample_rate=4)
        out["xyz_s4"], out["rgb_s4"] = xs, rs
        # ground-truth poses
        os.makedirs(os.path.join(tmp, "data/stanford/pose/area_3"))
        os.makedirs(os.path.join(tmp, "data/stanford/pose/area_30"))
        pose = {"camera_location": [1.25, -3.5, 1.6], "final_camera_rotation": [1.4835, 0.0321, -2.2143]}
        name = "camera_abc123_office_7_frame_equirectangular_domain_rgb.png"
        with open(os.path.join(tmp, "data/stanford/pose/area_3/camera_abc123_office_7_frame_equirectangular_domain_pose.json"), "w") as f:
            json.dump(pose, f)
        align = np.array([[0.8, -0.6, 0.0, 2.0], [0.6, 0.8, 0.0, -1.0], [0.0, 0.0, 1.0, 0.25]])
        np.savetxt(os.path.join(tmp, "data/stanford/pose/area_30/office_7.txt"), align)
        os.makedirs(os.path.join(tmp, "omni/pano"))
        os.makedirs(os.path.join(tmp, "omni/pose"))
        omni = np.hstack([np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]), np.array([[0.5], [1.5], [-0.25]])])
        np.savetxt(os.path.join(tmp, "omni/pose/room_1.txt"), omni)
        os.chdir(tmp)
        try:
            t3, r3 = ref_data.obtain_gt_stanford(3, name)
            t30, r30 = ref_data.obtain_gt_stanford(30, name)
            to, ro = ref_data.obtain_gt_omniscenes(os.path.join(tmp, "omni/pano/room_1.jpg"))
        finally:
            os.chdir(cwd)
    out.update(pose_loc=np.array(pose["camera_location"]), pose_rot=np.array(pose["final_camera_rotation"]), align=align, omni=omni,
               gt3_t=t3, gt3_r=r3, gt30_t=t30, gt30_r=r30, omni_t=to, omni_r=ro)
    save("g15_data_utils.npz", **out)


def g16_histogram():
    """color_utils.histogram / histogram_intersection (pure torch): unit-range and 0..255 images, [32,32,32] (default),
    [8,8,8] and uneven bins, unbatched and batched forms."""
    import color_utils as ref_color
    rng = np.random.default_rng(16)
    img255 = rng.integers(0, 256, size=(2, 20, 30, 3)).astype(np.float32)
    mask = rng.random((2, 20, 30)) < 0.6
    out = {"img255": img255, "mask": mask}
    for tag, ch in (("c32", [32, 32, 32]), ("c8", [8, 8, 8]), ("cu", [4, 16, 6])):
        h_unit = ref_color.histogram(torch.from_numpy(img255[0] / np.float32(255)), torch.from_numpy(mask[0]), ch)
        h_255 = ref_color.histogram(torch.from_numpy(img255[1]), torch.from_numpy(mask[1]), ch)
        h_raw = ref_color.histogram(torch.from_numpy(img255[1]), torch.from_numpy(mask[1]), ch, normalize=False)
        hb = ref_color.histogram(torch.from_numpy(img255), torch.from_numpy(mask), ch)
        out[tag + "_unit"], out[tag + "_255"], out[tag + "_raw"], out[tag + "_batched"] = h_unit.numpy(), h_255.numpy(), h_raw.numpy(), hb.numpy()
        out[tag + "_inter"] = np.array(float(ref_color.histogram_intersection(h_unit, h_255)))
        out[tag + "_inter_batched"] = ref_color.histogram_intersection(hb, hb.flip(0)).numpy()
    save("g16_histogram.npz", **out)


def g17_small_utils():
    """The small helpers around the candidate grids: create_coordinate, compute_sampling_grid, adaptive_trans_num,
    out_of_room, get_bound (utils.py:232-318, :688-755)."""
    xyz, _ = synth.box_room(20_000, 17)
    X = torch.from_numpy(xyz)
    out = {"xyz": xyz}
    out["coord_4x8"] = ref_utils.create_coordinate(4, 8).numpy()
    yprs = np.array([[0.0, 0.0, 0.0], [0.7, -0.3, 0.2], [3.0, 1.2, -2.0]], np.float32)
    out["yprs"] = yprs
    out["grids_4x4"] = np.stack([ref_utils.compute_sampling_grid(torch.from_numpy(y), 4, 4).numpy() for y in yprs])
    out["grids_2x4"] = np.stack([ref_utils.compute_sampling_grid(torch.from_numpy(y), 2, 4).numpy() for y in yprs])
    out["adaptive_xyz_50"] = np.array(ref_utils.adaptive_trans_num(X, 50, xy_only=False))
    out["adaptive_xy_150"] = np.array(ref_utils.adaptive_trans_num(X, 150, xy_only=True))
    probes = np.array([[0.0, 0.0, 0.0], [3.9, 0.0, 0.0], [4.1, 0.0, 0.0], [0.0, -3.2, 0.0], [0.0, 0.0, 1.6], [-3.99, 2.99, -1.49]], np.float32)
    out["probes"] = probes
    out["out_of_room_q05"] = np.array([ref_utils.out_of_room(X, torch.from_numpy(p).reshape(3, 1), 0.05) for p in probes])
    out["out_of_room_q20"] = np.array([ref_utils.out_of_room(X, torch.from_numpy(p).reshape(3, 1), 0.2) for p in probes])
    b = ref_utils.get_bound(X, Cfg(out_of_room_quantile=0.1, max_yaw=3.0))
    out["bound_q10"] = np.array([b[k] for k in ("x", "y", "z", "yaw", "pitch", "roll")], np.float64)
    save("g17_small_utils.npz", **out)


# ----------------------------------------------------------------------------- G18
def g18_end_to_end_many_seeds():
    """The reference's full 100-iteration refinements over 32 scenes, each run twice (original and permuted point order:
    the second run is the reference's own fp32 self-noise) — sequential omniloc from one start, and for 8 scenes also
    omniloc_batch with 4 starts.  The query panoramas come from the deterministic C oracle renderer (oracle.make_pano_u8;
    the reference's own make_pano is set-valued, see G8), so the test regenerates them from the seed; a checksum is kept."""
    from oracle import oracle as orc
    N, H, W, S = 20000, 128, 256, 32
    cfg = Cfg(lr=0.1, num_iter=100, patience=5, factor=0.8, out_of_room_quantile=0.05, visualize=False, num_input=4)
    rows, rows_b, sums = [], [], []
    for s in range(S):
        seed = 300 + s
        xyz, rgb = synth.box_room(N, seed=seed)
        t_gt, ypr_gt = synth.gt_pose(seed)
        img_u8 = orc.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W))
        sums.append(int(img_u8.astype(np.int64).sum()))
        img = img_u8.astype(np.float32) / 255.0
        trans, rot = synth.start_poses(t_gt, ypr_gt, 4, seed=seed)
        R_gt = synth.rot_from_ypr_np(ypr_gt)
        perm = np.random.default_rng(1000 + s).permutation(N)
        out = []
        for x, c in ((xyz, rgb), (xyz[perm], rgb[perm])):
            r = ref_omniloc.omniloc(torch.from_numpy(img), torch.from_numpy(x), torch.from_numpy(c), torch.from_numpy(trans.copy()),
                                    torch.from_numpy(rot.copy()), 0, cfg, {})
            t, R = r[0].detach().numpy().reshape(3), r[1].detach().numpy()
            out.append(np.concatenate([t, R.reshape(-1), [float(r[2])], synth.pose_errors(t, R, t_gt, R_gt)]))
        rows.append(np.stack(out))
        if s < 8:
            out = []
            for x, c in ((xyz, rgb), (xyz[perm], rgb[perm])):
                r = ref_omniloc.omniloc_batch(torch.from_numpy(img), torch.from_numpy(x), torch.from_numpy(c),
                                              torch.from_numpy(trans.copy()), torch.from_numpy(rot.copy()), cfg, {})
                t, R = r[0].detach().numpy().reshape(3), r[1].detach().numpy()
                out.append(np.concatenate([t, R.reshape(-1), [float(r[2])], synth.pose_errors(t, R, t_gt, R_gt)]))
            rows_b.append(np.stack(out))
        print("G18 seed %d seq t_err %.4f / %.4f (perm)  r_err %.3f / %.3f" % (seed, rows[-1][0, 13], rows[-1][1, 13], rows[-1][0, 14], rows[-1][1, 14]), flush=True)
    # columns: t(3) R(9) loss t_err r_err ; axis 1: original / permuted point order
    save("g18_end_to_end_seeds.npz", N=N, H=H, W=W, seed0=300, seq=np.stack(rows), batch=np.stack(rows_b), img_sum=np.array(sums))


# ----------------------------------------------------------------------------- G19
def g19_trim_hist_empty_blocks():
    """trim_input_hist_secondary (utils.py:510-588) on a PARTIAL cloud (floor + one wall panel), so that for most
    candidates some blocks of the middle block rows are empty.  The reference then `break`s out of the block row
    (utils.py:568-571) and the remaining slots of `hist_intersect_split` — allocated once, outside the candidate loop
    (utils.py:539) — keep what EARLIER candidates left there: a candidate's score includes stale intersections.
    Captured from the function itself: the per-candidate `hist_intersect_split` (read from the function's frame each time
    its progress bar is updated, i.e. after the NaN clean-up at utils.py:579) and the full ranking (num_input = K)."""
    full_xyz, full_rgb = synth.box_room(30000, seed=19)
    keep = (full_xyz[:, 2] < -1.49) | ((full_xyz[:, 0] > 3.99) & (np.abs(full_xyz[:, 1]) < 1.5))
    xyz, rgb = full_xyz[keep], full_rgb[keep]
    t_gt, ypr_gt = synth.gt_pose(19)
    H, W, nh, nw = 64, 128, 4, 4
    # the query image shows the WHOLE room (no empty block on its side); the candidates render the partial cloud
    img = ref_utils.make_pano(torch.from_numpy(synth.transform_cloud(full_xyz, t_gt, ypr_gt)), torch.from_numpy(full_rgb),
                              resolution=(H, W)).astype(np.float32) / 255.0
    rng = np.random.default_rng(19)
    K = 12
    trans = (t_gt[None] + rng.normal(0, 0.4, size=(K, 3))).astype(np.float32)
    rot = np.stack([ypr_gt[0] + np.arange(K) * (2 * np.pi / K) * 1.7, ypr_gt[1] + rng.normal(0, 0.1, K), ypr_gt[2] + rng.normal(0, 0.1, K)], 1).astype(np.float32)
    trans[3], rot[3] = t_gt, ypr_gt
    captured = []

    class Bar:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def update(self, n):
            captured.append(sys._getframe(1).f_locals["hist_intersect_split"].clone().numpy())

    real = ref_utils.tqdm
    ref_utils.tqdm = Bar
    try:
        sel_t, sel_r = ref_utils.trim_input_hist_secondary(torch.from_numpy(img), torch.from_numpy(xyz), torch.from_numpy(rgb),
                                                           torch.from_numpy(trans), torch.from_numpy(rot), K, nh, nw)
    finally:
        ref_utils.tqdm = real
    split = np.stack(captured)                                  # (K, nh*nw) effective slots per candidate
    scores = split.sum(1) / (nh * nw)
    print("G19 split rows:\n", np.round(split[:, nw:3 * nw], 3), "\nscores", np.round(scores, 4))
    save("g19_trim_hist_empty_blocks.npz", xyz=xyz, rgb=rgb, img=img, trans=trans, rot=rot, split=split, scores=scores,
         ranked_trans=sel_t.numpy(), ranked_rot=sel_r.numpy(), num_split=np.array([nh, nw]))


# ----------------------------------------------------------------------------- G20
def g20_standalone_backward():
    """Autograd of the reference's stand-alone utils.cloud2idx / utils.sample_from_img (fp32 and fp64): gradients w.r.t. the
    points, the coordinates and the image, for random incoming gradients.  Inputs include the G1 special points (axes,
    rho = 0, the wrap seam), coordinates beyond +-0.99 (clip) and footprints that touch the zero padding."""
    g1 = np.load(os.path.join(HERE, "g1_cloud2idx.npz"))
    xyz = g1["xyz"].astype(np.float32)
    rng = np.random.default_rng(20)
    G = rng.normal(size=(len(xyz), 2)).astype(np.float32)
    out = {"xyz": xyz, "grad_coord_in": G}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        x = torch.from_numpy(xyz).to(dt).requires_grad_()
        ref_utils.cloud2idx(x).backward(torch.from_numpy(G).to(dt))
        out["grad_xyz_" + tag] = x.grad.numpy()
        xb = torch.from_numpy(xyz[:900].reshape(3, 300, 3)).to(dt).requires_grad_()
        ref_utils.cloud2idx(xb, batched=True).backward(torch.from_numpy(G[:900].reshape(3, 300, 2)).to(dt))
        out["grad_xyz_b_" + tag] = xb.grad.numpy()
    g2 = np.load(os.path.join(HERE, "g2_sample_from_img.npz"))
    img, coord = g2["img"].astype(np.float32), g2["coord"].astype(np.float32)
    Go = rng.normal(size=(len(coord), 3)).astype(np.float32)
    out.update(img=img, coord=coord, grad_rgb_in=Go)
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        I = torch.from_numpy(img).to(dt).requires_grad_()
        c = torch.from_numpy(coord).to(dt).requires_grad_()
        ref_utils.sample_from_img(I, c).backward(torch.from_numpy(Go).to(dt))
        out["grad_coord_" + tag] = c.grad.numpy()
        out["grad_img_" + tag] = I.grad.numpy()
    # composed, the way a caller would chain them: points -> cloud2idx -> sample_from_img -> sum of squares
    pts = synth.box_room(2000, seed=20)[0]
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        p = torch.from_numpy(pts).to(dt).requires_grad_()
        col = ref_utils.sample_from_img(torch.from_numpy(img).to(dt), ref_utils.cloud2idx(p))
        (col ** 2).sum().backward()
        out["chain_grad_" + tag] = p.grad.numpy()
    out["chain_pts"] = pts
    save("g20_standalone_backward.npz", **out)


# ----------------------------------------------------------------------------- G21
def g21_full_size_batch_loss():
    """The reference's own BatchSamplingLoss (omniloc.py:299-356) + autograd at BASELINE config 2's FULL size: 1M points,
    2048x1024 panorama, 32 candidate poses — in fp32 (what the reference computes) and in fp64.  The scene is the bench's
    (synth.box_room(1M, seed 0), image 0's ground truth and starting poses) with the panorama rendered by the deterministic
    C oracle renderer, so the GPU test regenerates the inputs from the seeds; only outputs and input checksums are stored.
    ~1 min in fp32 and ~2 min in fp64 on 8 cores, peak RSS ~8 GB."""
    from oracle import oracle as orc
    N, H, W, B = 1_000_000, 1024, 2048, 32
    xyz, rgb = synth.box_room(N, seed=0)
    t_gt, ypr_gt = synth.gt_pose(0)
    img_u8 = orc.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W))
    img = img_u8.astype(np.float32) / 255.0
    trans, rot = synth.start_poses(t_gt, ypr_gt, B, seed=0)
    res = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            mod = ref_omniloc.BatchSamplingLoss(torch.from_numpy(xyz).to(dtype), torch.from_numpy(rgb).to(dtype),
                                                torch.from_numpy(img).to(dtype), torch.device("cpu"), Cfg(num_input=B))
            t = torch.tensor(trans, dtype=dtype).unsqueeze(-1).requires_grad_()
            yaw = torch.tensor(rot[:, 0:1], dtype=dtype).requires_grad_()
            pitch = torch.tensor(rot[:, 1:2], dtype=dtype).requires_grad_()
            roll = torch.tensor(rot[:, 2:3], dtype=dtype).requires_grad_()
            loss, loss_list = mod(t, yaw, pitch, roll)
            loss.backward()
            res["loss_list_" + tag] = loss_list.detach().numpy()
            res["grad_t_" + tag] = t.grad.squeeze(-1).numpy()
            res["grad_ypr_" + tag] = torch.cat([yaw.grad, pitch.grad, roll.grad], 1).numpy()
            print("G21", tag, "done", flush=True)
            del mod, loss, loss_list, t, yaw, pitch, roll
        finally:
            torch.set_default_dtype(old)
    save("g21_full_size_batch_loss.npz", N=N, H=H, W=W, B=B, trans=trans, rot=rot, img_sum=int(img_u8.astype(np.int64).sum()),
         xyz_sum=float(xyz.astype(np.float64).sum()), rgb_sum=float(rgb.astype(np.float64).sum()), **res)


# ----------------------------------------------------------------------------- G22
G22B_ITERS = [0, 1, 2, 3, 4, 10, 50, 99]


def g22_shipped_shape_end_to_end():
    """The reference's omniloc_batch, full 100 iterations, at the sizes of its SHIPPED config (configs/stanford_parallel.ini:
    1M points / sample_rate 6 = 166 667 points, 2048x1024 panorama, num_input 6, lr 0.1, patience 5, factor 0.8) on 4 scenes,
    each run twice (original and permuted point order: the reference's own fp32 self-noise).  On the device this shape runs ONE
    launch per GD iteration (fused prologue, csrc/pcl_loss.hip).  Panoramas from the deterministic oracle renderer (checksums kept).
    ~45 s per run on 4 threads."""
    from oracle import oracle as orc
    N, H, W, S, B = 166_667, 1024, 2048, 4, 6
    cfg = Cfg(lr=0.1, num_iter=100, patience=5, factor=0.8, out_of_room_quantile=0.05, visualize=False, num_input=B)
    rows, sums = [], []
    it_rec = {k: [] for k in ("fwd_trans", "fwd_rot", "fwd_loss", "adam_grad", "adam_lr", "adam_param_after", "loss_f64", "grad_t_f64", "grad_ypr_f64",
                               "perm_fwd_trans", "perm_fwd_rot", "perm_fwd_loss")}
    fin_p, fin_l = ([], []), ([], [])
    for s in range(S):
        seed = 700 + s
        xyz, rgb = synth.box_room(N, seed=seed)
        t_gt, ypr_gt = synth.gt_pose(seed)
        img_u8 = orc.make_pano_u8(synth.transform_cloud(xyz, t_gt, ypr_gt), rgb, (H, W))
        sums.append(int(img_u8.astype(np.int64).sum()))
        img = img_u8.astype(np.float32) / 255.0
        trans, rot = synth.start_poses(t_gt, ypr_gt, B, seed=seed)
        R_gt = synth.rot_from_ypr_np(ypr_gt)
        perm = np.random.default_rng(2000 + s).permutation(N)
        out = []
        for run, (x, c) in enumerate(((xyz, rgb), (xyz[perm], rgb[perm]))):
            rec = Recorder()                     # G22b: every forward, gradient and optimiser step of the run (as G5)
            rec.install()
            try:
                r = ref_omniloc.omniloc_batch(torch.from_numpy(img), torch.from_numpy(x), torch.from_numpy(c), torch.from_numpy(trans.copy()),
                                              torch.from_numpy(rot.copy()), cfg, {})
            finally:
                rec.remove()
            t, R = r[0].detach().numpy().reshape(3), r[1].detach().numpy()
            out.append(np.concatenate([t, R.reshape(-1), [float(r[2])], synth.pose_errors(t, R, t_gt, R_gt)]))
            a = rec.arrays("")
            # all six candidates at the end: the post-step, pre-clamp parameters the function picks its return value from
            # (omniloc.py:260-263,271-277; Adam's order [t, yaw, roll, pitch]) and the losses of the last forward
            fin_p[run].append(a["adam_param_after"][-1])
            fin_l[run].append(a["fwd_loss"][-1])
            if run == 1:                         # the reference's own rerun (points permuted), same iterations: the yardstick of
                for k in ("fwd_trans", "fwd_rot", "fwd_loss"):         # every free-running comparison
                    it_rec["perm_" + k].append(a[k][G22B_ITERS])
            if run == 0:
                for k in ("fwd_trans", "fwd_rot", "fwd_loss", "adam_grad", "adam_lr", "adam_param_after"):
                    it_rec[k].append(a[k][G22B_ITERS])
                # the reference's own fp64 evaluation AT THE RECORDED fp32 POSES of those iterations: the yardstick of every
                # per-evaluation comparison (its fp32 run's distance from it)
                l64, gt64, gy64 = [], [], []
                for k in G22B_ITERS:
                    l, gt, gy = loss_and_grads(x, c, img, a["fwd_trans"][k].astype(np.float64), a["fwd_rot"][k].astype(np.float64), torch.float64)
                    l64.append(l); gt64.append(gt); gy64.append(gy)
                it_rec["loss_f64"].append(np.stack(l64)); it_rec["grad_t_f64"].append(np.stack(gt64)); it_rec["grad_ypr_f64"].append(np.stack(gy64))
        rows.append(np.stack(out))
        print("G22 seed %d t_err %.4f / %.4f (perm)  r_err %.3f / %.3f  loss %.5f / %.5f" % (
            seed, rows[-1][0, 13], rows[-1][1, 13], rows[-1][0, 14], rows[-1][1, 14], rows[-1][0, 12], rows[-1][1, 12]), flush=True)
    # columns: t(3) R(9) loss t_err r_err ; axis 1: original / permuted point order
    save("g22_shipped_shape.npz", N=N, H=H, W=W, B=B, seed0=700, batch=np.stack(rows), img_sum=np.array(sums))
    # G22b: per scene, iterations G22B_ITERS of the original-order run: forward poses (S, 8, 6, 3), loss_list (S, 8, 6), autograd
    # gradients in Adam's order [t, yaw, roll, pitch] (S, 8, 6, 6), lr, post-step parameters; the reference's fp64 loss / gradients
    # at those poses; and all six final candidates of both runs
    save("g22b_shipped_iterations.npz", iters=np.array(G22B_ITERS), **{k: np.stack(v) for k, v in it_rec.items()},
         final_param=np.stack([np.stack(fin_p[0]), np.stack(fin_p[1])], 1), final_loss=np.stack([np.stack(fin_l[0]), np.stack(fin_l[1])], 1))


# ----------------------------------------------------------------------------- G23
def g23_make_input_composed():
    """The reference's make_input (utils.py:591-629) COMPOSED, for its three shipped configs: candidate grid -> trim_input_loss
    (loss table, argsort, the `// len(rot)` / `% len(rot)` decode at utils.py:504-505) -> trim_input_hist_secondary -> final
    starting poses.  The configs go through the reference's own parse_ini and localize.get_init_dict (localize.py:18-73; the
    module imports with a stub for tensorboard).  Intermediates are READ OUT OF THE RUNNING FUNCTIONS: a profile hook copies the
    locals of trim_input_loss / trim_input_hist_secondary (loss_table, min_inds, hist_intersect, their arguments) at their return.
    Scene: 30 000-point room, 64x128 panorama with a black patch; the cloud is subsampled by the config's sample_rate the way
    data_utils.read_stanford does it at load time (stanford_parallel.ini: 6)."""
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    sys.modules.setdefault("torch.utils.tensorboard", tb)
    if not hasattr(torch.utils, "tensorboard"):
        torch.utils.tensorboard = tb
    import localize as ref_localize
    xyz0, rgb0, img, t_gt, ypr_gt = small_scene(n=30000, H=64, W=128, seed=23, black_patch=True)
    out = dict(xyz=xyz0, rgb=rgb0, img=img, t_gt=t_gt, ypr_gt=ypr_gt)
    codes = {ref_utils.trim_input_loss.__code__: ("loss", ("loss_table", "min_inds", "trans", "rot", "trimmed_trans", "trimmed_rot")),
             ref_utils.trim_input_hist_secondary.__code__: ("hist", ("hist_intersect", "min_inds", "trans", "rot"))}
    for name in ("stanford.ini", "stanford_parallel.ini", "omniscenes.ini"):
        cfg = ref_parse.parse_ini(os.path.join(REF, "configs", name))
        init_dict = ref_localize.get_init_dict(cfg)
        rate = getattr(cfg, "sample_rate", 1)
        xyz, rgb = xyz0, rgb0
        if rate > 1:                                           # data_utils.py:36-41: every sample_rate-th point, int(N / rate) of them
            k = int(xyz0.shape[0] / rate)
            xyz, rgb = xyz0[::rate][:k], rgb0[::rate][:k]
        got = {}

        def hook(frame, event, arg):
            if event == "return" and frame.f_code in codes:
                tag, names = codes[frame.f_code]
                for nm in names:
                    got[tag + "_" + nm] = frame.f_locals[nm].detach().clone().numpy()

        torch.manual_seed(0)
        sys.setprofile(hook)
        try:
            it, ir = ref_utils.make_input(torch.from_numpy(img), torch.from_numpy(xyz), torch.from_numpy(rgb), cfg.num_input, init_dict,
                                          cfg.criterion, cfg.num_intermediate)
        finally:
            sys.setprofile(None)
        # the reference against ITSELF: the second stage again on the same survivors with the points in another order.  Its
        # render is set-valued (duplicate indices in index_put_, argsort ties), so its own scores move — the yardstick for
        # every comparison of histogram scores
        first = dict(got)
        perm = np.random.default_rng(23).permutation(len(xyz))
        sys.setprofile(hook)
        try:
            ref_utils.trim_input_hist_secondary(torch.from_numpy(img), torch.from_numpy(xyz[perm]), torch.from_numpy(rgb[perm]),
                                                torch.from_numpy(first["hist_trans"]), torch.from_numpy(first["hist_rot"]), cfg.num_input,
                                                init_dict["num_split_h"], init_dict["num_split_w"])
        finally:
            sys.setprofile(None)
        perm_scores = got["hist_hist_intersect"]
        got = first
        got["hist_hist_intersect_permuted"] = perm_scores
        print("    the reference's scores move by %.2e when its points are permuted" % np.abs(perm_scores - got["hist_hist_intersect"]).max())
        tag = name.replace(".ini", "")
        out[tag + "_init_dict"] = json.dumps(init_dict)
        out[tag + "_num"] = np.array([cfg.num_input, cfg.num_intermediate, rate, len(xyz)])
        for k, v in got.items():
            out[tag + "_" + k] = v
        out[tag + "_input_trans"], out[tag + "_input_rot"] = it.numpy(), ir.numpy()
        tbl = np.sort(got["loss_loss_table"].reshape(-1))
        print("G23", name, "grid", got["loss_loss_table"].shape, "table gap at the cut %.3e (table %.3f..%.3f)" % (
            tbl[cfg.num_intermediate] - tbl[cfg.num_intermediate - 1], tbl[0], tbl[-1]),
            "hist scores %.4f..%.4f" % (got["hist_hist_intersect"].min(), got["hist_hist_intersect"].max()), flush=True)
    save("g23_make_input.npz", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    todo = [g1_cloud2idx, g2_sample_from_img, g3_g4_loss_grad, g5_trajectories, g6_quantile, g7_trim_input_loss,
            g8_make_pano, g9_parse, g10_candidates, g11_end_to_end, g12_trim_input_hist, g13_color_match, g14_color_mod, g15_data_utils, g16_histogram, g17_small_utils,
            g18_end_to_end_many_seeds, g19_trim_hist_empty_blocks,
            g20_standalone_backward, g21_full_size_batch_loss, g22_shipped_shape_end_to_end, g23_make_input_composed]
    for fn in todo:
        if only and not any(fn.__name__.startswith(o) for o in only):
            continue
        torch.manual_seed(0)
        fn()
