"""MI355X counterpart of the reference's utils.py for the sampling-loss path (what localize.py pulls in through
`from utils import *`, localize.py:13).  Geometry / sampling / rendering run as HIP kernels through piccolo_amd.ops;
candidate generation is small host logic kept behaviour-compatible with the reference.

    cloud2idx            utils.py:16-61        sample_from_img     utils.py:64-103
    make_pano            utils.py:134-205      quantile            utils.py:208-229
    out_of_room          utils.py:232-254      rot_from_ypr        utils.py:425-453
    trim_input_loss      utils.py:462-507      trim_input_hist_secondary utils.py:510-588
    make_input           utils.py:591-629      generate_rot_points / generate_trans_points / adaptive_trans_num
    compute_sampling_grid / create_coordinate  utils.py:702-755      write_summaries utils.py:455-459
"""
import math
from collections import defaultdict
from math import ceil

import numpy as np
import torch

from . import ops
from .omniloc import packed_cloud, packed_pano

__all__ = ["cloud2idx", "sample_from_img", "warp_from_img", "reshape_img_tensor", "make_pano", "quantile", "out_of_room", "rot_from_ypr", "trim_input_loss",
           "trim_input_hist_secondary", "make_input", "make_input_images", "generate_rot_points", "generate_trans_points", "adaptive_trans_num",
           "compute_sampling_grid", "create_coordinate", "write_summaries", "debug_visualize", "resize_image", "get_bound", "defaultdict", "torch", "np"]


def _like(out, ref):
    """Results live where the caller's tensors live (the reference computes on the inputs' device)."""
    return out if (torch.is_tensor(ref) and ref.is_cuda) else out.cpu()


# ------------------------------------------------------------------------------------------------ geometry ops
class _Cloud2Idx(torch.autograd.Function):
    """cloud2idx as an autograd op: forward and backward are one HIP kernel each (csrc/pcl_ops.hip)."""

    @staticmethod
    def forward(ctx, xyz):
        ctx.save_for_backward(xyz)
        return _like(ops.cloud2idx(xyz), xyz)

    @staticmethod
    def backward(ctx, grad_coord):
        xyz, = ctx.saved_tensors
        return _like(ops.cloud2idx_backward(xyz, grad_coord), xyz).to(xyz.dtype)


class _SampleFromImg(torch.autograd.Function):
    """sample_from_img as an autograd op: gradient w.r.t. the coordinates and, if asked for, the image."""

    @staticmethod
    def forward(ctx, img, coord_arr):
        pano = packed_pano(img)
        ctx.pano = pano
        ctx.save_for_backward(img, coord_arr)
        return _like(ops.sample_from_img(pano, coord_arr), coord_arr)

    @staticmethod
    def backward(ctx, grad_rgb):
        img, coord = ctx.saved_tensors
        want_img, want_coord = ctx.needs_input_grad
        gc, gi = ops.sample_from_img_backward(ctx.pano, coord, grad_rgb, want_coord=want_coord, want_img=want_img)
        return (_like(gi, img).to(img.dtype) if want_img else None), (_like(gc, coord).to(coord.dtype) if want_coord else None)


def cloud2idx(xyz, batched=False):
    """(N,3) or (B,N,3) camera-frame points -> equirectangular coordinates in [-1,1]^2 (x = column, y = row).
    Differentiable like the reference's (a plain autograd op there, utils.py:16-61): if xyz requires grad the result
    carries a grad_fn whose backward is the HIP kernel pcl_cloud2idx_backward."""
    if torch.is_tensor(xyz) and xyz.requires_grad and torch.is_grad_enabled():
        return _Cloud2Idx.apply(xyz)
    return _like(ops.cloud2idx(xyz), xyz)


def sample_from_img(img, coord_arr, padding="zeros", mode="bilinear", batched=False):
    """Bilinear lookup of (N,2) / (B,N,2) coordinates in an (H,W,3) image, clipped to +-0.99, zero padding.
    Differentiable w.r.t. the coordinates and the image like the reference's (torch.clip + F.grid_sample, utils.py:64-103)."""
    if padding != "zeros" or mode != "bilinear":
        raise NotImplementedError("sample_from_img: only padding='zeros', mode='bilinear' (all the reference uses)")
    if batched and coord_arr.shape[0] == 1:
        # the reference's batched path squeezes the batch away for B == 1 (utils.py:88) and then fails
        raise RuntimeError("sample_from_img(batched=True) needs B > 1, like the reference")
    if torch.is_grad_enabled() and ((torch.is_tensor(coord_arr) and coord_arr.requires_grad) or (torch.is_tensor(img) and img.requires_grad)):
        return _SampleFromImg.apply(img, coord_arr)
    return _like(ops.sample_from_img(packed_pano(img), coord_arr), coord_arr)


def warp_from_img(img, coord_arr, padding="zeros", mode="bilinear"):
    """(H,W,3) image sampled at an (H',W',2) grid of coordinates -> (H',W',3) (utils.py:106-131; unused by the reference's
    own flows, kept for the `from utils import *` surface)."""
    if padding != "zeros" or mode != "bilinear" or img.shape[-1] != 3:
        raise NotImplementedError("warp_from_img: bilinear / zeros / 3 channels only")
    shp = coord_arr.shape
    return sample_from_img(img, coord_arr.reshape(-1, 2)).reshape(shp[0], shp[1], 3)


def resize_image(img8, width, height):
    """uint8 (H,W,3) -> (height,width,3), bilinear with OpenCV's INTER_LINEAR geometry (pixel centres at k + 0.5, edge
    clamp, no antialiasing; cv2.resize at localize.py:168,211,372).  Identity when the size is unchanged, which is the
    case for all shipped configs on 2048 x 1024 panoramas.  cv2 interpolates in 11-bit fixed point: results may differ
    from it by one level — parity unpinned against OpenCV (absent from the build image); torch's F.interpolate(bilinear,
    align_corners=False, antialias=False), the same geometry, agrees within half a level (tests/test_dataset_harness.py)."""
    H, W = img8.shape[:2]
    if (W, H) == (width, height):
        return img8
    fy = (np.arange(height, dtype=np.float64) + 0.5) * (H / height) - 0.5
    fx = (np.arange(width, dtype=np.float64) + 0.5) * (W / width) - 0.5
    y0, x0 = np.floor(fy).astype(np.int64), np.floor(fx).astype(np.int64)
    wy, wx = (fy - y0)[:, None, None], (fx - x0)[None, :, None]
    y0c, y1c = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)
    x0c, x1c = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
    a = img8.astype(np.float64)
    top = a[y0c][:, x0c] * (1 - wx) + a[y0c][:, x1c] * wx
    bot = a[y1c][:, x0c] * (1 - wx) + a[y1c][:, x1c] * wx
    return np.clip(np.rint(top * (1 - wy) + bot * wy), 0, 255).astype(np.uint8)


def reshape_img_tensor(img, size):
    """utils.py:632-638: (H,W,3) float image in [0,1] -> uint8 -> resized to size = (X, Y) -> float / 255, on img's device
    (host round trip like the reference; `resize_image` stands in for cv2.resize)."""
    cv_img = (img.detach().cpu().numpy() * 255).astype(np.uint8)
    cv_img = resize_image(cv_img, int(size[0]), int(size[1])) / 255.
    return torch.from_numpy(cv_img).float().to(img.device)


def rot_from_ypr(ypr_array):
    """(3,) [yaw, pitch, roll] -> R = RZ(yaw) RY(pitch) RX(roll), (3,3)."""
    return _like(ops.rot_from_ypr(ypr_array.reshape(1, 3))[0], ypr_array)


def quantile(x, q):
    """(x_sorted[int(n q)], x_sorted[int(n (1-q))]) — order statistics without interpolation."""
    col = ops._dev(x).reshape(-1, 1).expand(-1, 3).contiguous()
    box = _like(ops.quantile_box(col, q), x)
    return box[0], box[1]


def out_of_room(xyz, trans, out_quantile=0.05):
    """True if `trans` (3,1) is outside the open [q, 1-q] quantile box of the cloud."""
    from .omniloc import _cached, quantile_box_of    # the box is loop invariant: shared with omniloc / omniloc_batch
    # (its host copy too: the dataset loops ask once per query image)
    box = _cached("box", (xyz,), lambda: quantile_box_of(xyz, out_quantile).cpu(), sub=("host", float(out_quantile)))
    t = torch.as_tensor(trans).detach().cpu().reshape(3)
    inside = all(box[2 * k] < t[k] < box[2 * k + 1] for k in range(3))
    return not inside


def get_bound(xyz, cfg, return_brute=False):
    box = ops.quantile_box(xyz, getattr(cfg, "out_of_room_quantile", 0.05)).cpu().tolist()
    rng = [(getattr(cfg, "min_yaw", 0), getattr(cfg, "max_yaw", 2 * np.pi)),
           (getattr(cfg, "min_pitch", 0), getattr(cfg, "max_pitch", np.pi)),
           (getattr(cfg, "min_roll", 0), getattr(cfg, "max_roll", 2 * np.pi))]
    pairs = [(box[0], box[1]), (box[2], box[3]), (box[4], box[5])] + rng
    if return_brute:
        return tuple(slice(a, b) for a, b in pairs)
    return dict(zip(["x", "y", "z", "yaw", "pitch", "roll"], pairs))


def make_pano(xyz, rgb, resolution=(200, 400), return_torch=False):
    """Render camera-frame points into an (H,W,3) panorama: 3x3 splats, nearest point wins (z-buffered on the GPU
    instead of the reference's argsort + nine index_put_ passes).  uint8 numpy by default, float tensor*255 if
    return_torch."""
    img = ops.make_pano(xyz, rgb, resolution)
    if return_torch:
        return _like(img, xyz)
    return img.cpu().numpy().astype(np.uint8)


# ------------------------------------------------------------------------------------------------ initialisation
def _trim_order(xyz, cloud, pano, trans, rot, groups):
    """The trim launch's work list (ops.TrimOrder), cached per (cloud points, candidate grid, panorama size / texel layout): it does not
    depend on the query image or on the colours.  None where a cache key cannot be formed (grids that are not tensors)."""
    from .omniloc import _cached
    if not (torch.is_tensor(trans) and torch.is_tensor(rot) and torch.is_tensor(xyz)) or not ops.trim_order_pays(cloud.n, pano.H, pano.W, pano.fmt):
        return None
    return _cached("trimorder", (xyz, trans, rot), lambda: ops.TrimOrder(cloud, (pano.H, pano.W, pano.fmt), trans, groups), sub=(pano.H, pano.W, pano.fmt))


def trim_input_loss(img, xyz, rgb, trans, rot, num_input):
    """Keep the `num_input` (translation, rotation) pairs with the smallest sampling loss out of all K x R pairs.
    The reference loops K*R forwards in Python (utils.py:484-499); here all pairs go through ONE launch in which the rotations
    that differ only in yaw share the projection of every point (csrc/pcl_trim.hip): the candidate rotations are a yaw x pitch x
    roll grid (utils.py:321-360), and for R = RZ(yaw) RY RX neither the panorama row nor the distance depends on yaw."""
    from .omniloc import _cached
    K, Rn = len(trans), len(rot)
    cloud = packed_cloud(xyz, rgb)
    # (the generic kernel of the R > 1024 fallback reads row-major texels only)
    pano = packed_pano(img, many_poses=True, n_points=xyz.shape[0] if Rn <= ops.TRIM_MAX_ROT else None)
    # the (pitch, roll) classes of the rotation table: once per table (the grid is cached per config in make_input)
    if Rn <= ops.TRIM_MAX_ROT:
        groups = _cached("trimgroups", (rot,), lambda: ops.TrimGroups(rot)) if torch.is_tensor(rot) else ops.TrimGroups(rot)
        table = ops.trim_loss_table(cloud, pano, trans, groups, order=_trim_order(xyz, cloud, pano, trans, rot, groups)).reshape(-1)      # row-major (K, R) like the reference's loss_table
    else:
        # more rotations than the yaw-sharing launch classifies (its table of classes lives in LDS): the generic forward-only
        # kernel over all pairs, a slice of translations at a time
        tr, ro = ops._dev(trans).reshape(-1, 3), ops._dev(rot).reshape(-1, 3)
        rows = max(1, 65536 // Rn)
        table = torch.cat([ops.sampling_loss(cloud, pano, tr[k0:k0 + rows].repeat_interleave(Rn, 0), ro.repeat(min(rows, K - k0), 1),
                                             with_grad=False)[:, 0] for k0 in range(0, K, rows)])
    num_input = min(num_input, K * Rn)
    # loss_table.flatten().argsort()[:num_input] and the `// len(rot)`, `% len(rot)` decode (utils.py:500-505) in ONE launch
    # (pcl_select_poses: radix select + rank + gather; torch.topk + sort + four index kernels took 110 us per image for 7 KB of
    # data); NaN losses (nothing sampled) rank last, as in the reference's argsort
    if num_input <= ops.SELECT_MAX_KEEP:
        tt, tr = ops.select_poses(table, num_input, trans, rot, largest=False, rot_per_trans=Rn)
        return _like(tt, trans), _like(tr, rot)
    # (more than SELECT_MAX_KEEP survivors: torch.topk, with NaN made to lose as in pcl_select_poses — topk itself ranks NaN first)
    min_inds = torch.topk(torch.nan_to_num(table, nan=float("inf")), num_input, largest=False, sorted=True).indices.to(trans.device)
    return trans[torch.div(min_inds, Rn, rounding_mode="floor")], rot[min_inds % Rn]


def trim_input_hist_secondary(img, xyz, rgb, trans, rot, num_input, num_split_h, num_split_w):
    """Second trimming stage (utils.py:510-588): render a panorama per candidate and rank candidates by the mean
    block-wise colour-histogram intersection with the query image.  All candidates go through three fused kernels
    (csrc/pcl_hist.hip): batched z-buffer splat, query histograms, per-(candidate, block) LDS histogram + intersection."""
    scores = ops.hist_trim_scores(img, packed_cloud(xyz, rgb), trans, rot, num_split_h, num_split_w)
    n = min(num_input, scores.numel())
    if n <= ops.SELECT_MAX_KEEP:                    # flip(argsort()[-n:]) and the two gathers (utils.py:583-586) in one launch
        tt, tr = ops.select_poses(scores, n, trans, rot, largest=True)
        return _like(tt, trans), _like(tr, rot)
    order = torch.topk(torch.nan_to_num(scores, nan=float("-inf")), n, largest=True, sorted=True).indices.to(trans.device)   # best first, NaN last
    return trans[order], rot[order]


def adaptive_trans_num(xyz, max_trans_num, xy_only=False):
    ext = torch.quantile(xyz, 0.90, dim=0) - torch.quantile(xyz, 0.10, dim=0)
    lx, ly, lz = [float(v) for v in ext]
    if xy_only:
        return ceil((lx * max_trans_num / ly) ** 0.5), ceil((ly * max_trans_num / lx) ** 0.5)
    nums = [ceil((lx ** 2 * max_trans_num / (ly * lz)) ** (1 / 3)), ceil((ly ** 2 * max_trans_num / (lx * lz)) ** (1 / 3)),
            ceil((lz ** 2 * max_trans_num / (lx * ly)) ** (1 / 3))]
    return tuple(n - 1 if n % 2 == 0 else n for n in nums)


def create_coordinate(h_out, w_out, device=torch.device("cpu")):
    """(h_out, w_out, 2) grid of (longitude pi - 2 pi x / w, latitude pi y / h)."""
    lon = np.pi - torch.arange(w_out, device=device, dtype=torch.float32) * (2 * math.pi / w_out)
    lat = torch.arange(h_out, device=device, dtype=torch.float32) * (math.pi / h_out)
    return torch.stack([lon[None, :].expand(h_out, -1), lat[:, None].expand(-1, w_out)], dim=-1)


def _sampling_dirs(num_split_h, num_split_w, device):
    a = create_coordinate(num_split_h, num_split_w, device)
    lon = a[..., 0] - np.pi / num_split_w
    lat = a[..., 1] + np.pi / (num_split_h * 2)
    return torch.stack([torch.sin(lat) * torch.cos(lon), torch.sin(lat) * torch.sin(lon), torch.cos(lat)], dim=-1)


def compute_sampling_grid(ypr, num_split_h, num_split_w):
    """Where the centres of a num_split_h x num_split_w block grid land after rotating by R(ypr)^T (utils.py:719-755)."""
    R = rot_from_ypr(ypr).T
    dirs = _sampling_dirs(num_split_h, num_split_w, ypr.device)
    rotated = (R @ dirs.unsqueeze(3)).squeeze(3)
    return cloud2idx(rotated.reshape(-1, 3)).reshape(num_split_h, num_split_w, 2)


def generate_rot_points(init_dict=None, device="cpu"):
    """(R,3) [yaw, pitch, roll] starting rotations (utils.py:321-360).  In the 3-DoF case rotations whose block
    sampling grids coincide (to 3 decimals) are dropped; the survivors are returned in first-occurrence order (the
    reference's order is that of a Python set of strings, i.e. arbitrary per process)."""
    d = init_dict
    if d["yaw_only"]:
        rot = torch.zeros(d["num_yaw"], 3, device=device)
        rot[:, 0] = torch.arange(d["num_yaw"], dtype=torch.float, device=device) * 2 * np.pi / d["num_yaw"]
        return rot
    fr = [torch.arange(d[k], device=device).float() / d[k] for k in ("num_yaw", "num_pitch", "num_roll")]
    grid = torch.stack(torch.meshgrid(*fr, indexing="ij"), dim=-1).reshape(-1, 3)
    lo = torch.tensor([d["min_yaw"], d["min_pitch"], d["min_roll"]], device=device, dtype=torch.float)
    hi = torch.tensor([d["max_yaw"], d["max_pitch"], d["max_roll"]], device=device, dtype=torch.float)
    rot = grid * (hi - lo) + lo
    # all sampling grids in one projection launch (the reference projects one rotation at a time)
    Rt = ops.rot_from_ypr(rot).transpose(1, 2).to(device)
    dirs = _sampling_dirs(d["num_yaw"], d["num_pitch"], device).reshape(-1, 3)
    rotated = torch.einsum("bij,nj->bni", Rt, dirs)
    grids = cloud2idx(rotated.reshape(-1, 3)).reshape(len(rot), d["num_yaw"], d["num_pitch"], 2).cpu().numpy()
    seen, keep = set(), []
    for i, g in enumerate(grids):
        key = str(np.around(g, 3))
        if key not in seen:
            seen.add(key)
            keep.append(i)
    return rot[keep]


def generate_trans_points(xyz, init_dict=None, device="cpu"):
    """(K,3) starting translations on a grid inside the cloud (utils.py:363-422)."""
    d = init_dict

    def axis_points(k, num):
        col = xyz[:, k]
        mode = d["trans_init_mode"]
        if mode == "uniform":
            return (torch.arange(num, device=device) + 1) / (num + 1) * (col.max() - col.min()) + col.min()
        if mode == "manual":
            lo, hi = d["xyz"[k] + "_min"], d["xyz"[k] + "_max"]
            return torch.arange(num, device=device) / (num - 1) * (hi - lo) + lo
        if 1 / (num + 1) > 0.1:
            split = (torch.arange(num, device=device) + 1) / (num + 1)
        else:
            split = torch.linspace(0.1, 0.9, num, device=device)
        return torch.quantile(col, split)

    if d["xy_only"]:
        if d["dataset"] not in ("Stanford2D-3D-S", "OmniScenes"):
            raise NotImplementedError("Other datasets not supported")
        nx, ny = adaptive_trans_num(xyz, d["num_trans"], xy_only=True)
        gx, gy = torch.meshgrid(axis_points(0, nx), axis_points(1, ny), indexing="ij")
        out = torch.zeros(nx * ny, 3, device=device)
        out[:, 0], out[:, 1] = gx.reshape(-1), gy.reshape(-1)
        out[:, 2] = d["z_prior"] if d["z_prior"] is not None else xyz[:, 2].mean()
        return out
    nx, ny, nz = adaptive_trans_num(xyz, d["num_trans"], xy_only=False)
    g = torch.meshgrid(axis_points(0, nx), axis_points(1, ny), axis_points(2, nz), indexing="ij")
    return torch.stack([c.reshape(-1) for c in g], dim=1)


_ROT_GRIDS = {}


_INIT_KEYS = {}


def _init_key(init_dict, dev):
    """The cache key of a candidate-grid config on a device: repr of its sorted items (as rounds 1-5), memoised by the items themselves —
    building the string was 15 us of every make_input call, in front of its first launch."""
    try:
        items = (tuple(init_dict.items()), dev)
        key = _INIT_KEYS.get(items)
        if key is None:
            if len(_INIT_KEYS) > 64:
                _INIT_KEYS.clear()
            key = _INIT_KEYS[items] = repr(sorted((k, str(v)) for k, v in init_dict.items())) + str(dev)
        return key
    except TypeError:                                 # an unhashable value (a list): no memo
        return repr(sorted((k, str(v)) for k, v in init_dict.items())) + str(dev)


def make_input(img, xyz, rgb, num_input, init_dict=None, criterion="histogram", num_intermediate=None):
    """Starting poses for the refinement (utils.py:591-629): candidate grid -> sampling-loss trim -> histogram trim.
    Only criterion == 'loss_histogram' exists in the reference (anything else hits an unbound local there)."""
    # the candidate grids depend on the cloud and the config only, not on the query image: build them once per cloud
    # (the reference rebuilds them for every image; ~10 ms of small tensor ops, torch.quantile's NaN scans included)
    from .omniloc import _cached
    key = _init_key(init_dict, img.device)
    rot = _ROT_GRIDS.get(key)                       # the rotation grid depends on the config alone: once per process
    if rot is None:
        rot = _ROT_GRIDS[key] = generate_rot_points(init_dict, device=img.device)
    trans = _cached("grid", (xyz,), lambda: generate_trans_points(xyz, init_dict, device=img.device), sub=key)
    if init_dict["sample_rate_for_init"] is not None:
        raise NotImplementedError("sample_rate_for_init: broken in the reference too (utils.py:618-620)")
    if criterion != "loss_histogram":
        raise UnboundLocalError("make_input: only criterion='loss_histogram' is implemented (as in the reference)")
    t1, r1 = trim_input_loss(img, xyz, rgb, trans, rot, num_intermediate)
    return trim_input_hist_secondary(img, xyz, rgb, t1, r1, num_input, init_dict["num_split_h"], init_dict["num_split_w"])


def make_input_images(imgs, xyz, rgb, num_input, init_dict=None, criterion="histogram", num_intermediate=None):
    """make_input for SEVERAL query images of one room (throughput extension; the reference's image loop, localize.py:143-223,
    calls make_input once per image although the candidate grid depends on the cloud only, utils.py:613-616): ONE trim launch
    over image x translation x rotation, one selection launch for all images, the second stage for all images' survivors in one set
    of launches, one final selection.
    Returns [(input_trans, input_rot)] per image — the tensors make_input returns for that image, bit for bit (the trim launch
    cuts the cloud into the single-image launch's chunks; tests/test_hip_harness.py)."""
    from .omniloc import _cached
    if init_dict["sample_rate_for_init"] is not None:
        raise NotImplementedError("sample_rate_for_init: broken in the reference too (utils.py:618-620)")
    if criterion != "loss_histogram":
        raise UnboundLocalError("make_input: only criterion='loss_histogram' is implemented (as in the reference)")
    I = len(imgs)
    dev = imgs[0].device
    key = _init_key(init_dict, dev)
    rot = _ROT_GRIDS.get(key)
    if rot is None:
        rot = _ROT_GRIDS[key] = generate_rot_points(init_dict, device=dev)
    trans = _cached("grid", (xyz,), lambda: generate_trans_points(xyz, init_dict, device=dev), sub=key)
    K, Rn = len(trans), len(rot)
    n_mid = min(num_intermediate, K * Rn)
    if I == 1 or Rn > ops.TRIM_MAX_ROT or n_mid > ops.SELECT_MAX_KEEP:
        return [make_input(im, xyz, rgb, num_input, init_dict, criterion, num_intermediate) for im in imgs]
    cloud = packed_cloud(xyz, rgb)
    panos = [packed_pano(im, many_poses=True, n_points=xyz.shape[0]) for im in imgs]
    if len({p.fmt for p in panos}) > 1:                       # a launch needs one texel format: float4 holds any image
        panos = [ops.Pano(im, fmt="f32") for im in imgs]
    groups = _cached("trimgroups", (rot,), lambda: ops.TrimGroups(rot))
    tables = ops.trim_loss_tables(cloud, panos, trans, groups, order=_trim_order(xyz, cloud, panos[0], trans, rot, groups)).reshape(I, K * Rn)
    t1, r1 = ops.select_poses(tables, n_mid, trans, rot, largest=False, rot_per_trans=Rn)               # (I, n_mid, 3)
    scores = ops.hist_trim_scores_images(imgs, cloud, t1, r1, init_dict["num_split_h"], init_dict["num_split_w"])        # (I, n_mid)
    ft, fr = ops.select_poses(scores, min(num_input, n_mid), t1, r1, largest=True)                      # (I, num_input, 3)
    return [(ft[i], fr[i]) for i in range(I)]


def debug_visualize(tgt_tensor):
    """utils.py:641-699: show a tensor / array of shape (H,W), (H,W,C) or (B,H,W,C) with matplotlib (first batch item;
    3 channels as RGB, otherwise one grey panel per channel; values above 2 are taken as 0..255).  Host-side helper kept for
    the `from utils import *` surface.  (The reference's numpy branch uses `np.float`, removed in numpy 1.24.)"""
    import matplotlib.pyplot as plt
    if torch.is_tensor(tgt_tensor):
        vis_tgt = tgt_tensor.detach().cpu().float().numpy()
    elif isinstance(tgt_tensor, np.ndarray):
        vis_tgt = tgt_tensor.astype(np.float64)
    else:
        raise ValueError("Invalid input!")
    if vis_tgt.max() > 2.0:
        vis_tgt = vis_tgt / 255.
    if vis_tgt.ndim == 4:
        vis_tgt = vis_tgt[0]
    if vis_tgt.ndim == 2:
        vis_tgt = vis_tgt[..., None]
    if vis_tgt.ndim != 3:
        return
    C = vis_tgt.shape[2]
    if C == 3:
        plt.imshow(vis_tgt)
    elif C == 1:
        plt.imshow(vis_tgt[..., 0], cmap="gray", vmin=vis_tgt.min(), vmax=vis_tgt.max())
    else:
        fig = plt.figure(figsize=(50, 50))
        for i in range(C):
            fig.add_subplot(max(C // 2, 1), 2, i + 1)
            plt.imshow(vis_tgt[..., i], cmap="gray", vmin=vis_tgt[..., i].min(), vmax=vis_tgt[..., i].max())
    plt.show()


def write_summaries(writer, scalar_summaries, step):
    for k, v in scalar_summaries.items():
        writer.add_scalar(k, float(np.array(v).mean()), step)
