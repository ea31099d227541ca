"""MI355X counterpart of the reference's omniloc.py — same names, arguments, return values and error behaviour.

    omniloc(img, xyz, rgb, input_trans, input_rot, starting_point, cfg, scalar_summaries)   omniloc.py:11-102
    omniloc_batch(img, xyz, rgb, input_trans, input_rot, cfg, scalar_summaries)             omniloc.py:205-296
    sampling_loss(img, xyz, rgb, input_trans, input_rot, starting_point, cfg, return_list)  omniloc.py:105-157
    SamplingLoss / BatchSamplingLoss (nn.Module, differentiable w.r.t. the pose)            omniloc.py:160-202, :299-356

The reference builds the loss from ~40 ATen ops and lets autograd differentiate it; here one HIP kernel computes
loss and gradient together (csrc/pcl_loss.hip) and the whole Adam / ReduceLROnPlateau / clamp loop runs on the
device (csrc/pcl_gd.hip) without a host round trip per iteration.  Differences a caller can observe:
  * results come back as DETACHED cpu float32 tensors (the reference's still require grad, which breaks its own
    localize.py:227 on numpy >= 2);
  * omniloc_batch also accepts a single candidate (the reference asserts num_input > 1, omniloc.py:208; the assert
    is kept because callers may rely on it, see `strict_reference_asserts`);
  * extra, optional cfg keys: depth_mask (default False = reference behaviour) multiplies the north star's
    scatter-min visibility (csrc/pcl_depth.hip) of the CURRENT poses into the loss mask at every iteration; depth_res =
    (depth_h, depth_w) is the z-buffer's grid (default: by point density, pcl_depth_default — not the panorama's size),
    depth_tau its tolerance (default: the rule's value for the grid), depth_stride = s builds the z-buffer from every s-th
    point of the Morton-ordered cloud (default: pcl_depth_default's choice; every point is still tested against it);
  * cfg.visualize: the reference's frame capture is broken (`new_xyz` undefined, omniloc.py:61 -> NameError); here
    omniloc returns the frame list that code means to build (query image over the cloud rendered at the current pose,
    per iteration) as 4th element.
"""
import weakref
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops

strict_reference_asserts = True

# ------------------------------------------------------------------------------------------------ pack caches
# The harness calls omniloc() once per starting point with the same img/xyz/rgb (localize.py:219-220): pack once.
# One small LRU per KIND of packed object, so a stream of query images can never evict the room's Morton order, packed
# cloud, quantile box or translation grid (a dataset loop touches 4 cloud-side entries per room and 2 per image).
# An entry is keyed by the identity of the tensors it was made from (address, shape, in-place version) and holds weak
# references to them: a hit needs the very same live tensor, and entries whose tensors died are purged.
_CAPACITY = {"cloud": 2, "order": 2, "box": 8, "grid": 4, "pano": 16, "pano_u8": 16, "pano_u8p": 16, "pano_u8v": 16, "gd": 6, "trimgroups": 4}


class _PackCache:
    def __init__(self):
        self.kinds = {}                     # kind -> OrderedDict(key -> ([weakrefs], object)), least recent first

    def _lru(self, kind):
        lru = self.kinds.get(kind)
        if lru is None:
            lru = self.kinds[kind] = OrderedDict()
        return lru

    def get(self, kind, key, tensors):
        lru = self._lru(kind)
        hit = lru.get(key)
        if hit is not None and all(r() is t for r, t in zip(hit[0], tensors)):
            lru.move_to_end(key)
            return hit[1]
        return None

    def put(self, kind, key, tensors, obj):
        lru = self._lru(kind)
        for k in [k for k, (refs, _) in lru.items() if any(r() is None for r in refs)]:
            del lru[k]                      # the tensors are gone: the address may be reused by another tensor
        lru[key] = ([weakref.ref(t) for t in tensors], obj)
        lru.move_to_end(key)
        while len(lru) > _CAPACITY.get(kind, 4):
            lru.popitem(last=False)

    def clear(self):
        self.kinds.clear()

    def __len__(self):
        return sum(len(v) for v in self.kinds.values())


_cache = _PackCache()


def _key(*tensors):
    return tuple((t.data_ptr(), tuple(t.shape), t._version, str(t.device), t.dtype) for t in tensors)


def _cached(kind, tensors, make, sub=None):
    """`kind` selects the LRU (and its capacity), `sub` distinguishes entries of one kind made from the same tensors
    (the quantile of a box, the config of a candidate grid)."""
    k = (sub,) + _key(*tensors)
    obj = _cache.get(kind, k, tensors)
    if obj is None:
        obj = make()
        _cache.put(kind, k, tensors, obj)
    return obj


def quantile_box_of(xyz, out_quantile):
    """The clamp box of omniloc.py:53-55 / :245-247, computed once per (cloud, quantile)."""
    return _cached("box", (xyz,), lambda: ops.quantile_box(xyz, out_quantile), sub=float(out_quantile))


def packed_cloud(xyz, rgb):
    """Packed cloud cached per (xyz, rgb); the Morton order is cached per xyz alone, so a cloud whose colours change with
    every query image (color_mod, localize.py:175-179) is re-packed without being re-sorted."""
    def make():
        order = _cache.get("order", (None,) + _key(xyz), (xyz,))
        if order is not None:
            return ops.Cloud(xyz, rgb, order=order)
        c = ops.Cloud(xyz, rgb)
        if c.order is not None:
            _cache.put("order", (None,) + _key(xyz), (xyz,), c.order)
        return c
    return _cached("cloud", (xyz, rgb), make)


def packed_pano(img, many_poses=False, n_points=None):
    """Packed panorama of `img`, cached per tensor.  RGBA8 texels (half the footprint of the fp16-level default) when the launch
    evaluates hundreds of candidate poses all over the room (`many_poses`, trim_input_loss: with 1800 poses the fp16 texture
    thrashes L2, 8.7 vs 5.3 ms per launch at cfg-2 size) or when the cloud to be refined is sparse against the panorama
    (`n_points`: ops.refine_texels); fp16-level texels otherwise (the refinement's nearby poses on a dense cloud run 5 % faster on
    them).  The trim launch of a SPARSE cloud takes its own layout ('u8p', rows interleaved in pairs, cache 'pano_u8p'; a DENSE one
    'u8v', vertical pairs), so at the
    shipped 167k-point shape an image whose initialisation and refinement use the same tensor is packed twice (8 MB each, two
    caches): the two stages share a packing only for dense clouds' trim ('u8') and a sparse cloud's refinement ('u8').  Images that
    are not k/255 get float4 texels either way."""
    rgba8 = many_poses or (n_points is not None and ops.refine_texels(n_points, img.shape[0], img.shape[1]) == "u8")
    if ops.EXPERIMENT.pano_fmt in ("f16", "f32") and not many_poses:             # experiments: force the refinement's format
        rgba8 = False
    if not rgba8:
        return _cached("pano", (img,), lambda: ops.Pano(img))
    # the trim launch picks its own layout by point density (ops.trim_texels: rows interleaved in pairs / plain rows / vertical pairs);
    # nothing else reads the paired layouts
    fmt = ops.trim_texels(n_points, img.shape[0], img.shape[1]) if many_poses and n_points is not None else "u8"
    if ops.EXPERIMENT.trim_fmt in ("u8", "u8p", "u8v") and many_poses:            # experiments
        fmt = ops.EXPERIMENT.trim_fmt

    def make():
        try:
            return ops.Pano(img, fmt=fmt)
        except ValueError:
            return ops.Pano(img, fmt="f32")
    return _cached("pano_" + fmt, (img,), make)


def _cfg(cfg, key, default):
    return getattr(cfg, key, default)


def _rot_matrix(ypr):
    return ops.rot_from_ypr(ypr.reshape(1, 3))[0]


# ------------------------------------------------------------------------------------------------ GD drivers
# A refinement of a SMALL problem is launch-latency bound: 2 x num_iter dependent launches of a few microseconds each
# (the reference's shipped configs: 167k points x 6 candidates, 15 us per iteration on the GPU).  For those the whole
# launch chain is captured once into a hipGraph and replayed for every later refinement of the same cloud and shape —
# pcl_gd_run neither allocates nor synchronises, every candidate reads its panorama through its pose record
# (pcl_gd_set_panos), so a new image only needs pcl_gd_init + pcl_gd_set_panos + one graph launch.  Replay is
# bit-identical to the eager launches (tests).  Measured at cfg 1: 0.65 vs 0.73 ms per refinement; nothing at cfg 2
# (110 us kernels), hence the size limit.  Round 5: between 4M and 16M point-poses (1M points x 6 candidates, 400k x 32: 15-30 us
# launches) the medians are the same, but an eager chain now and then loses a millisecond to the host thread (3.3 -> 4.4 ms, 5.2 -> 6.2 ms
# seen at 1M / 2M points x 6 candidates; never with replay): the limit went from 4M to 16M.
GRAPH_POINT_POSES = 16_000_000         # use graph replay when points x candidates is at most this (cfg key gd_graph overrides)


def _refine(xyz, rgb, panos, trans, rot, box, cfg, batch_mode, vis_hook=None):
    """Run the on-device GD for the rows of trans / rot and return the GradientDescent object (read gd.result() / gd.winner()).
    `panos`: one packed panorama per query image; the B rows split evenly over them, image by image.
    The GradientDescent object (state, workspace, captured graph) is cached per cloud and launch shape."""
    cloud = packed_cloud(xyz, rgb)
    trans, rot = ops._dev(trans).reshape(-1, 3), ops._dev(rot).reshape(-1, 3)
    B = int(trans.shape[0])
    p0 = panos[0]
    num_iter = _cfg(cfg, "num_iter", 100)
    depth = bool(_cfg(cfg, "depth_mask", False))
    d_tau, d_res, d_st = _cfg(cfg, "depth_tau", None), _cfg(cfg, "depth_res", None), _cfg(cfg, "depth_stride", None)
    hyper = (float(_cfg(cfg, "lr", 0.1)), int(_cfg(cfg, "patience", 5)), float(_cfg(cfg, "factor", 0.9)), bool(batch_mode), depth,
             None if d_tau is None else float(d_tau), None if d_res is None else (int(d_res[0]), int(d_res[1])), None if d_st is None else int(d_st))
    use_graph = _cfg(cfg, "gd_graph", None)
    if use_graph is None and ops.EXPERIMENT.gd_graph is not None:                   # experiments
        use_graph = bool(ops.EXPERIMENT.gd_graph)
    fuse = None if _cfg(cfg, "gd_fuse", True) else False       # (cfg gd_fuse = False: two launches per iteration, bit-identical)
    if use_graph is None:
        use_graph = cloud.n * B <= GRAPH_POINT_POSES
    use_graph = bool(use_graph) and vis_hook is None and not depth

    def make(c=cloud, bx=box):
        return ops.GradientDescent(c, p0, trans, rot, bx, lr=hyper[0], patience=hyper[1], factor=hyper[2], batch_mode=hyper[3],
                                   depth_mask=hyper[4], depth_tau=hyper[5], depth_res=hyper[6], depth_stride=hyper[7], fuse=fuse)
    if not use_graph:
        gd = make()                                        # (fresh buffers: nothing worth keeping for a long eager chain)
    else:
        # One engine (state, workspace, captured graph) per POINT SET and launch shape.  The colours may change with every query
        # image (color_mod / match_color give each image its own rgb): the engine owns a private copy of the packed cloud whose
        # address the captured graph holds, and a cloud with other colours is copied into it (24 bytes per point on the device)
        # instead of capturing a new graph per image.
        def make_private():
            g = make(ops.Cloud.private_copy(cloud), ops._dev(box).reshape(6).clone())     # (its own box buffer: updated in place below)
            g._cloud_src = weakref.ref(cloud)                # the copy just made IS this cloud: nothing to copy on first use
            g._box_src, g._fresh = box, True
            return g
        # (one or two launches per iteration is frozen into a captured graph: part of the key)
        gd = _cached("gd", (xyz,), make_private, sub=(B, len(panos), p0.H, p0.W, p0.fmt, fuse) + hyper)
        fresh, gd._fresh = gd._fresh, False                  # (a new engine was initialised with these very poses)
        if gd._cloud_src() is not cloud:                     # weak: the engine must not keep packed clouds of past images alive
            gd.cloud.data.copy_(cloud.data)
            gd._cloud_src = weakref.ref(cloud)
        if gd._box_src is not box:                           # in place: the captured graph holds this buffer's address
            gd.box.copy_(ops._dev(box).reshape(6))
            gd._box_src = box                                # (the cached box tensor of this cloud: identity is enough)
        if not fresh:
            gd.reset(trans, rot)
    if len(panos) > 1 or use_graph:
        gd.set_pano_groups(list(panos))                      # addresses as kernel arguments: no H2D copy, nothing waits
    if vis_hook is not None:
        vis_hook(gd, num_iter)
    elif use_graph:
        gd.run_graph(num_iter)
    else:
        gd.run(num_iter)
    return gd


def _leaf_buffers(input_trans, input_rot, B):
    """The caller's starting-pose tensors as write-back targets of pcl_gd_winner when they are contiguous float32 GPU tensors of B
    rows (the harness's are); else fresh buffers plus a copy afterwards.  -> (buf_t, buf_r, after)"""
    def usable(t):
        return torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == 3 * B and not t.requires_grad
    bt = input_trans if usable(input_trans) else torch.empty(B, 3, dtype=torch.float32, device=ops.device())
    br = input_rot if usable(input_rot) else torch.empty(B, 3, dtype=torch.float32, device=ops.device())

    def after():
        with torch.no_grad():
            if bt is not input_trans:
                input_trans.copy_(bt.reshape(input_trans.shape).to(input_trans.device))
            if br is not input_rot:
                input_rot.copy_(br.reshape(input_rot.shape).to(input_rot.device))
    return bt, br, after


def omniloc(img, xyz, rgb, input_trans, input_rot, starting_point, cfg, scalar_summaries):
    """Sequential refinement of ONE starting pose.  Returns [t (3,1), R (3,3), loss ()] (+ frames if cfg.visualize).

    `loss` is the loss of the last forward, i.e. at the pose before the final update, like the reference
    (omniloc.py:46,102).  Row `starting_point` of input_trans / input_rot ends up holding the final pose, as in the
    reference where the optimised tensors are views of those rows (omniloc.py:15-19).
    """
    vis = _cfg(cfg, "visualize", False)
    out_quantile = _cfg(cfg, "out_of_room_quantile", 0.05)

    pano = packed_pano(img, n_points=xyz.shape[0])
    # the reference recomputes these three quantiles every iteration (omniloc.py:53-55); they are loop invariant
    box = quantile_box_of(xyz, out_quantile)
    frames = []
    hook = (lambda gd, n: frames.extend(_run_with_frames(gd, img, xyz, rgb, n))) if vis else None
    res = _refine(xyz, rgb, [pano], input_trans[starting_point], input_rot[starting_point], box, cfg, False, vis_hook=hook).result()[0]
    R = _rot_matrix(res[3:6])
    out = torch.cat([res[0:3], R.reshape(-1), res[12:13]]).cpu()
    with torch.no_grad():
        input_trans[starting_point] = res[6:9].to(input_trans.device)
        input_rot[starting_point] = res[9:12].to(input_rot.device)
    ret = [out[0:3].reshape(3, 1).clone(), out[3:12].reshape(3, 3).clone(), out[12].clone()]
    if vis:
        ret.append(frames)
    return ret


def _run_with_frames(gd, img, xyz, rgb, num_iter):
    """cfg.visualize: the frame list the reference means to build (omniloc.py:59-69, :93-100; its own code stops at the
    undefined `new_xyz`): per iteration one PIL frame, the query image on top of the cloud rendered (make_pano, half
    resolution) at the pose that iteration's forward used; the first frame 5 times in all, the last one 10 more times, then
    5 frames with a black lower half.  The harness saves them as a GIF (localize.py:285-288).  Same on-device GD, one
    iteration per call; the renders are the z-buffer kernel."""
    import numpy as np
    from PIL import Image
    h, w = int(img.shape[0]) // 2, int(img.shape[1]) // 2
    gt_img = Image.fromarray(np.uint8(ops._dev(img).cpu().numpy() * 255), "RGB").resize((w, h))
    frames, new_frame = [], None
    for it in range(num_iter):
        pose = gd.result()[0]                               # the parameters this iteration's forward sees
        gd.run(1)
        cur = ops.make_pano(ops.transform_cloud(xyz, pose[0:3], pose[3:6]), rgb, (h, w)).cpu().numpy().astype(np.uint8)
        new_frame = Image.new("RGB", (w, 2 * h))
        new_frame.paste(gt_img, (0, 0))
        new_frame.paste(Image.fromarray(cur), (0, h))
        frames.extend([new_frame] * (5 if it == 0 else 1))
    if new_frame is not None:
        last_frame = Image.new("RGB", (w, 2 * h))
        last_frame.paste(gt_img, (0, 0))
        frames.extend([new_frame] * 10 + [last_frame] * 5)
    return frames


def omniloc_all(img, xyz, rgb, input_trans, input_rot, cfg, scalar_summaries=None):
    """Throughput extension: what the reference's non-parallel branch computes with
    `for i in range(num_input): omniloc(..., i, ...)` (localize.py:219-220), for ALL starting points in one launch chain.
    Every starting point keeps omniloc's SEQUENTIAL semantics (its own Adam / scheduler, clamp applied to the parameters
    the next forward reads) and the points never interact, so the list returned equals the K separate calls."""
    box = quantile_box_of(xyz, _cfg(cfg, "out_of_room_quantile", 0.05))
    res = _refine(xyz, rgb, [packed_pano(img, n_points=xyz.shape[0])], input_trans, input_rot, box, cfg, False).result()
    K = res.shape[0]
    R = ops.rot_from_ypr(res[:, 3:6])
    host = torch.cat([res[:, 0:3], R.reshape(K, 9), res[:, 12:13]], dim=1).cpu()
    with torch.no_grad():
        input_trans.copy_(res[:, 6:9].to(input_trans.device))
        input_rot.copy_(res[:, 9:12].to(input_rot.device))
    return [[host[i, 0:3].reshape(3, 1).clone(), host[i, 3:12].reshape(3, 3).clone(), host[i, 12].clone()] for i in range(K)]


def omniloc_batch(img, xyz, rgb, input_trans, input_rot, cfg, scalar_summaries):
    """Parallel refinement of all starting poses; returns [t (3,1), R (3,3), loss ()] of the candidate whose LAST
    forward had the smallest loss (omniloc.py:271).  Keeps the reference's clamp lag (omniloc.py:260-269): the
    returned translation is the post-step, pre-clamp value (omniloc.py:272)."""
    if strict_reference_asserts:
        assert cfg.num_input > 1
    box = quantile_box_of(xyz, _cfg(cfg, "out_of_room_quantile", 0.05))
    gd = _refine(xyz, rgb, [packed_pano(img, n_points=xyz.shape[0])], input_trans, input_rot, box, cfg, True)
    # loss_list.argmin() of the last forward, R of the winner and the write-back of the leaves: one kernel, then the one D2H copy
    # of the whole refinement (64 bytes)
    bt, br, after = _leaf_buffers(input_trans, input_rot, gd.B)
    out = gd.winner(1, bt, br)[0].cpu()
    after()
    return [out[0:3].reshape(3, 1).clone(), out[3:12].reshape(3, 3).clone(), out[12].clone()]


def omniloc_batch_images(imgs, xyz, rgb, input_trans_list, input_rot_list, cfg, scalar_summaries=None, batch_mode=True):
    """Throughput extension (not in the reference): omniloc_batch for SEVERAL query images of the same cloud at once.

    imgs: list of (H,W,3) images of one size; input_trans_list / input_rot_list: per image (B,3) starting poses (same B).
    The I * B candidates run through one chain of launches (shared cloud in L2, per-candidate panorama pointer), each with
    its own Adam / scheduler state, so every image gets the result omniloc_batch would give it (bit for bit when the
    cloud is cut into the same chunks, else up to the summation order of the partial sums); at 32 candidates per image,
    8 images per launch are ~25 % faster than 8 separate refinements.  Returns a list of [t, R, loss].
    batch_mode=False gives every candidate omniloc's SEQUENTIAL semantics instead (what omniloc_all computes per image)."""
    if strict_reference_asserts and batch_mode:
        assert cfg.num_input > 1
    I = len(imgs)
    B = int(input_trans_list[0].shape[0])
    fmt = ops.refine_texels(xyz.shape[0], imgs[0].shape[0], imgs[0].shape[1])
    panos = [packed_pano(im, n_points=xyz.shape[0]) if I <= 8 else ops.Pano(im, fmt=fmt if ops._known_levels(im) else "auto") for im in imgs]
    if len({p.fmt for p in panos}) > 1:          # a launch needs ONE texel format: float4 holds any image
        panos = [ops.Pano(im, fmt="f32") for im in imgs]
    box = quantile_box_of(xyz, _cfg(cfg, "out_of_room_quantile", 0.05))
    tr = torch.cat([ops._dev(t).reshape(B, 3) for t in input_trans_list])
    ro = torch.cat([ops._dev(r).reshape(B, 3) for r in input_rot_list])
    gd = _refine(xyz, rgb, panos, tr, ro, box, cfg, batch_mode)
    leaf_t, leaf_r = torch.empty(I * B, 3, dtype=torch.float32, device=tr.device), torch.empty(I * B, 3, dtype=torch.float32, device=tr.device)
    host = gd.winner(I, leaf_t, leaf_r).cpu()                # per image: the smallest last loss (omniloc.py:271), one D2H copy
    with torch.no_grad():
        for i in range(I):
            input_trans_list[i].copy_(leaf_t[i * B:(i + 1) * B].to(input_trans_list[i].device))
            input_rot_list[i].copy_(leaf_r[i * B:(i + 1) * B].to(input_rot_list[i].device))
    return [[host[i, 0:3].reshape(3, 1).clone(), host[i, 3:12].reshape(3, 3).clone(), host[i, 12].clone()] for i in range(I)]


def sampling_loss(img, xyz, rgb, input_trans, input_rot, starting_point, cfg, return_list=True):
    """Forward-only loss of one starting pose — omniloc.py:105-157."""
    cloud, pano = packed_cloud(xyz, rgb), packed_pano(img)
    t, r = input_trans[starting_point], input_rot[starting_point]
    res = ops.sampling_loss(cloud, pano, t, r, with_grad=False, depth=_depth_cfg(cfg))[0]
    loss = res[0].cpu()
    if return_list:
        return [t.detach().reshape(3, 1).cpu().clone(), _rot_matrix(ops._dev(r)).cpu(), loss]
    return loss


def _depth_cfg(cfg):
    """None (the reference's loss), or the depth-mask arguments of ops.sampling_loss from cfg.depth_mask / depth_res / depth_tau —
    the same mask one iteration of the GD loop uses for these poses."""
    if not bool(_cfg(cfg, "depth_mask", False)):
        return None
    return {"depth_res": _cfg(cfg, "depth_res", None), "depth_tau": _cfg(cfg, "depth_tau", None), "depth_stride": _cfg(cfg, "depth_stride", None),
            "on": True}


# ------------------------------------------------------------------------------------------------ nn.Modules
class _LossFn(torch.autograd.Function):
    """loss_b(t_b, ypr_b) for B poses; the fused kernel returns loss and gradient together, backward just scales."""

    @staticmethod
    def forward(ctx, cloud, pano, trans, rot, depth=None):
        res = ops.sampling_loss(cloud, pano, trans, rot, with_grad=True, depth=depth)
        ctx.save_for_backward(res[:, 2:5], res[:, 5:8])
        ctx.devs = (trans.device, rot.device)
        return res[:, 0].to(trans.device)

    @staticmethod
    def backward(ctx, grad_loss):
        gt, gr = ctx.saved_tensors
        g = grad_loss.to(gt.device).reshape(-1, 1)
        return None, None, (g * gt).to(ctx.devs[0]), (g * gr).to(ctx.devs[1]), None


class SamplingLoss(nn.Module):
    """omniloc.py:160-202.  forward(translation (3,1), yaw (1,), pitch (1,), roll (1,)) -> scalar loss."""

    def __init__(self, xyz, rgb, img, device, cfg):
        super().__init__()
        self.xyz, self.rgb, self.img, self.cfg = xyz, rgb, img, cfg
        self._cloud, self._pano = ops.Cloud(xyz, rgb), ops.Pano(img)

    def forward(self, translation, yaw, pitch, roll):
        trans = translation.reshape(1, 3)
        rot = torch.cat([yaw.reshape(1), pitch.reshape(1), roll.reshape(1)]).reshape(1, 3)
        return _LossFn.apply(self._cloud, self._pano, trans, rot, _depth_cfg(self.cfg))[0]


class BatchSamplingLoss(nn.Module):
    """omniloc.py:299-356.  forward(translation (B,3,1), yaw (B,1), pitch (B,1), roll (B,1)) -> (sum, (B,) list)."""

    def __init__(self, xyz, rgb, img, device, cfg):
        super().__init__()
        self.xyz, self.rgb, self.img, self.cfg = xyz, rgb, img, cfg
        self.num_input = cfg.num_input
        self._cloud, self._pano = ops.Cloud(xyz, rgb), ops.Pano(img)

    def forward(self, translation, yaw, pitch, roll):
        B = translation.shape[0]
        if B != self.num_input:
            # the reference's (num_input,1) constant tensors make any other batch size a shape error (omniloc.py:307-318)
            raise RuntimeError("BatchSamplingLoss: batch size %d != cfg.num_input %d" % (B, self.num_input))
        trans = translation.reshape(B, 3)
        rot = torch.cat([yaw.reshape(B, 1), pitch.reshape(B, 1), roll.reshape(B, 1)], dim=1)
        loss_list = _LossFn.apply(self._cloud, self._pano, trans, rot, _depth_cfg(self.cfg))
        return loss_list.sum(), loss_list
