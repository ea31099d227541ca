"""Torch-tensor front end of the C ABI: device memory and streams come from PyTorch-ROCm (plumbing), every
computation is a HIP kernel in libpiccolo_hip.so.

Inputs may live on the CPU (the reference's harness hands over whatever `device` it picked, localize.py:124):
they are uploaded to cuda:0.  Without a GPU or without the library every call raises — there is no fallback.
"""
import ctypes

import torch

from . import _lib


class _Experiment:
    """A/B switches of the measured-and-rejected log (profiles/EXPERIMENTS.md).  The product reads NO environment variable: these are
    plain attributes, all None / False by default, set by a tool or a test in its own process (tools/_knobs.py maps the old PCL_*
    variables onto them for the sweep scripts).
      pano_fmt      "f16" | "f32": what Pano(fmt="auto") tries first / forces for the refinement's panorama
      trim_fmt      "u8" | "u8p" | "u8v": the trim launch's texel layout instead of ops.trim_texels' choice
      gd_graph      True | False: hipGraph replay of the refinement chain on / off whatever the problem size
      verify_levels True: ignore synth.mark_levels' tag (the device-side k/255 check is read back as for any tensor)"""
    pano_fmt = None
    trim_fmt = None
    gd_graph = None
    verify_levels = False


EXPERIMENT = _Experiment()

F32 = torch.float32


_GPU_SEEN = False        # torch.cuda.is_available() answered True once (it is re-asked until then: the product must fail loudly without a GPU)


def device():
    global _GPU_SEEN
    if not _GPU_SEEN:
        if not torch.cuda.is_available():
            raise _lib.PiccoloHipError("piccolo_amd needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
        _GPU_SEEN = True
    return torch.device("cuda", torch.cuda.current_device())


def _dev(t, dtype=F32):
    """contiguous tensor of `dtype` on the GPU (detached).  A tensor that already is one is returned as it is (the per-image path of
    make_input passes ~20 of them per call: the detach / to / contiguous round trip was a third of its host time)."""
    if torch.is_tensor(t) and t.is_cuda and t.dtype == dtype and not t.requires_grad and t.is_contiguous() and t.device.index == torch.cuda.current_device():
        return t
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    return t.detach().to(device=device(), dtype=dtype).contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _bytes(nbytes):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device())


class Cloud:
    """Point cloud packed for the loss kernel: 6 SoA planes, by default in Morton order of xyz.

    `order` maps packed slot -> original point index (None if the original order was kept)."""

    def __init__(self, xyz, rgb, sort=True, order=None):
        """`order`: a Morton order computed before for the same xyz (Cloud(...).order): skips the sort, e.g. when only
        the colours of a cloud changed (color_mod gives every query image its own rgb)."""
        lib = _lib.load()
        xyz, rgb = _dev(xyz), _dev(rgb)
        if xyz.dim() != 2 or xyz.shape[1] != 3 or rgb.shape != xyz.shape:
            raise ValueError("xyz and rgb must both be (N, 3)")
        self.n = int(xyz.shape[0])
        if self.n <= 0:
            raise ValueError("empty point cloud")
        self.order = None
        if order is not None:
            if order.dtype != torch.int64 or order.numel() != self.n or not order.is_cuda:
                raise ValueError("order must be a CUDA int64 tensor with one entry per point")
            self.order = order
        elif sort and self.n > 1:
            self.order = torch.empty(self.n, dtype=torch.int64, device=xyz.device)
            nws = lib.pcl_cloud_order_workspace_bytes(self.n)
            ws = _bytes(nws)
            _lib.check(lib.pcl_cloud_order(_ptr(xyz), self.n, _ptr(self.order), _ptr(ws), nws, _stream()), "pcl_cloud_order")
        self.data = _bytes(lib.pcl_cloud_bytes(self.n))
        _lib.check(lib.pcl_cloud_pack(_ptr(xyz), _ptr(rgb), _ptr(self.order), self.n, _ptr(self.data), _stream()),
                   "pcl_cloud_pack")
        self.xyz = xyz          # kept for quantile_box (reads the reference's AoS layout)

    @classmethod
    def private_copy(cls, other):
        """A Cloud with its own packed buffer holding `other`'s contents (same point order): for an engine whose captured graph
        must keep one cloud address while the colours change from image to image."""
        c = cls.__new__(cls)
        c.n, c.order, c.xyz = other.n, other.order, other.xyz
        c.data = other.data.clone()
        return c


class Pano:
    """Query panorama (H,W,3) float packed as zero-bordered texels.

    fmt="auto": fp16-level texels (half4, 8 B) when every value is exactly k/255 in fp32 (what an 8-bit image file
    divided by 255 gives, i.e. everything the reference's harness produces), float4 texels otherwise.  "u8" packs the
    same k/255 images as RGBA8 (4 B/texel: half the footprint, ~5 % slower loss kernel), "f32" forces float4.  The
    exactness test is one kernel and one 4-byte D2H read per image, outside the GD loop."""

    _PACK = {"f16": ("pcl_pano_pack_f16", _lib.PANO_F16), "u8": ("pcl_pano_pack_u8", _lib.PANO_U8),
             "u8p": ("pcl_pano_pack_u8p", _lib.PANO_U8P),         # u8p: rows interleaved in pairs, u8v: vertical pairs — trim launch only
             "u8v": ("pcl_pano_pack_u8v", _lib.PANO_U8V)}

    def __init__(self, img, fmt="auto"):
        lib = _lib.load()
        src = img
        img = _dev(img)
        if img.dim() != 3 or img.shape[2] != 3:
            raise ValueError("img must be (H, W, 3)")
        self.H, self.W = int(img.shape[0]), int(img.shape[1])
        self.fmt = None
        if fmt not in ("auto", "f16", "u8", "u8p", "u8v", "f32"):
            raise ValueError("unknown texel format %r" % (fmt,))
        prefer = "f16"
        if fmt == "auto":                                     # (experiments: what "auto" tries first)
            prefer = EXPERIMENT.pano_fmt or "f16"
            if prefer == "f32":
                fmt = "f32"
        if fmt != "f32":
            fn, code = self._PACK[prefer if fmt == "auto" else fmt]
            data = _bytes(lib.pcl_pano_bytes(self.H, self.W, code))
            # an image tagged as k/255 by construction (synth.mark_levels: the harness's decoded image files) is not waited for:
            # reading the flag is a blocking D2H copy per query image in front of a millisecond of work — and a flag nobody reads
            # need not be zeroed first (a scratch word per device instead of a fill launch per image)
            known = _known_levels(src)
            flag = _scratch_flag(img.device) if known else torch.zeros(1, dtype=torch.int32, device=img.device)
            _lib.check(getattr(lib, fn)(_ptr(img), self.H, self.W, _ptr(data), _ptr(flag), _stream()), fn)
            if known or int(flag.item()) == 0:
                self.fmt, self.data = code, data
            elif fmt != "auto":
                raise ValueError("image is not exactly k/255: cannot use %s texels" % fmt)
        if self.fmt is None:
            self.fmt = _lib.PANO_F32
            self.data = _bytes(lib.pcl_pano_bytes(self.H, self.W, _lib.PANO_F32))
            _lib.check(lib.pcl_pano_pack(_ptr(img), self.H, self.W, _ptr(self.data), _stream()), "pcl_pano_pack")


_SCRATCH_FLAGS = {}


def _scratch_flag(dev):
    """a device int32 the pack kernels may write their `not_exact` answer to when nobody will read it"""
    f = _SCRATCH_FLAGS.get(dev)
    if f is None:
        f = _SCRATCH_FLAGS[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    return f


def refine_texels(n, H, W):
    """Level-texel format ("f16" | "u8") for the REFINEMENT of an n-point cloud against an H x W panorama.  fp16-level texels (8 B)
    save 6 VALU instructions per point-pose and win where the loss kernel is VALU-bound (cfg 2: +4-5 % for starting poses near each
    other); a sparse cloud is bound by texture lines, latency and the L2 residency of the texture instead (the 2 x 2 footprints of a
    wave's 128 Morton neighbours share no cache line), and RGBA8 texels (4 B: half the texture in L2) win.  Measured per GD iteration,
    32 candidates all over the room (what make_input hands over; tools/refine_fmt_sweep.sh), f16 -> u8:
        2048 x 1024: 167k points x 6 candidates 14.1 -> 12.0 us; 32 candidates: 400k 66.3 -> 57.2, 700k 86.5 -> 83.1, 1M 112.4 -> 109.6,
                     2M 195.5 -> 199.4 (and with the candidates make_input really trims to at 1M: 11.4 -> 11.55 ms per refinement)
        4096 x 2048: 2M 344.7 -> 226.7, 3M 348.0 -> 318.9, 4M 410.7 -> 407.3, 4.5M 455.1 -> 456.7, 6M 581.6 -> 589.3, 8M 760.1 -> 766.9
    i.e. RGBA8 up to ~0.5 points per pixel for spread poses at both sizes; the threshold sits at 0.45 so that cfg 2 (0.48 points per
    pixel, where poses near each other — bench.py's — run 4-5 % faster on fp16 levels) stays on fp16.  (Round 4's threshold was 1/3:
    it left 8 % on the table at 3M points on 4096 x 2048 — 20 % before the chunks of large clouds were made smaller, pcl_plan.)  The formats give the same bits
    (tests/test_hip_parity.py::test_pano_format_selection_and_float_image)."""
    return "u8" if 20 * int(n) < 9 * int(H) * int(W) else "f16"


def trim_texels(n, H, W):
    """Level-texel layout ("u8p" | "u8" | "u8v") for the TRIM launch of an n-point cloud against an H x W panorama.  Three layouts of the
    same RGBA8 texels, bit-identical tables (tools/trim_u8p.py, tests/test_hip_parity.py::test_trim_loss_table_yaw_shared_vs_generic_
    kernel_and_oracle): plain rows `u8` (two 8-byte accesses per 2 x 2 footprint), rows interleaved in pairs `u8p` (1.5 accesses, same
    bytes, four selects per sample), vertical pairs `u8v` (ONE access, twice the texture).  Which one is fastest depends on what the
    launch is bound by — texture lines (sparse clouds), L2 residency of the texture under hundreds of concurrent views (large
    panoramas), VALU issue (dense clouds) — measured per 1800-pose launch, ms (u8 / u8p / u8v), chunks of at most 16k points
    (pcl_plan_for_groups):
        1024 x  512, u8v = 4 MB : 100k points 0.51 / 0.45 / 0.40, 250k 0.84 / 0.87 / 0.73, 500k 1.47 / 1.62 / 1.35    -> u8v always
        2048 x 1024, u8v = 17 MB: 167k 1.02 / 0.85 / 1.24, 400k 1.72 / 1.51 / 1.94, 700k 2.5 / 2.5 / 2.5, 850k 2.84 / 2.88 / 2.80,
                                  1M 3.18 / 3.34 / 3.06, 2M 6.05 / 6.47 / 5.31, 3M 8.29 / 9.60 / 7.78
                                                                                 -> u8p below 1/3 point per pixel, u8v from 5/12
        4096 x 2048, u8v = 67 MB: 3M 11.1 / 10.2 / 13.6, 4M 13.1 / 13.4 / 14.7, 6M 17.7 / 19.8 / 16.9, 8M 22.8 / 26.1 / 21.2,
                                  10M 28.0 / 32.5 / 26.2                         -> u8p below 0.45 points per pixel, u8v from 0.6
    (With the 64 chunks per launch of rounds 3-4 the large panorama looked different — 10M points 30.4 / 32.0 / 33.4: every chunk was
    a large piece of the room, and the doubled texture lost.)  Sizes in between take the rule of the nearer measured class (by the
    bytes of the doubled texture: up to 6 MB it lives in one XCD's L2, up to 24 MB it is cfg 2's class)."""
    n, px = int(n), int(H) * int(W)
    doubled = 8 * (int(H) + 2) * (int(W) + 2)
    if doubled <= 6_000_000:
        return "u8v"
    if doubled <= 24_000_000:
        return "u8p" if 3 * n < px else "u8v" if 12 * n >= 5 * px else "u8"
    return "u8p" if 20 * n < 9 * px else "u8v" if 5 * n >= 3 * px else "u8"


def trim_order_pays(n, H, W, fmt):
    """Does the trim launch of an n-point cloud against an H x W panorama in texel layout `fmt` (code) take the row-sorted work list
    (TrimOrder)?  Measured per 1800-pose launch, plain (chunk, slot) order -> with the list (tools/trim_u8p.py, round 6; tables bit-identical):
        2048 x 1024: 167k u8p 0.83 -> 0.74 ms, 400k u8p 1.51 -> 1.44, 1M u8v 3.17 -> 3.07 (memory-side 16.4 -> 6.6 GB, L2 hit 0.70 -> 0.88),
                     2M u8v 5.78 -> 5.52; 8 images per launch: 167k 0.747 -> 0.685 per image, 1M 3.10 -> 3.05
        1024 x  512: 100k u8v 0.38 -> 0.38, 500k 1.36 -> 1.35                                      (the texture lives in one L2 either way)
        4096 x 2048: 3M u8p 10.19 -> 10.48 (u8 11.2 -> 11.8, u8v 13.6 -> 13.1), 10M u8v 26.18 -> 26.46 (u8 27.8 -> 28.6)
    i.e. always up to cfg 2's texture class and never above it: for every layout ops.trim_texels picks on a 4096 x 2048 panorama the list
    loses 1-3 % (a band of a 67 MB texture is several L2s wide whatever the order; cfg 5's launch stays at round 5's 26.2 ms)."""
    doubled = 8 * (int(H) + 2) * (int(W) + 2)
    return doubled <= 24_000_000


def _known_levels(img):
    """True for a tensor tagged by synth.mark_levels (every value exactly k/255 by construction); EXPERIMENT.verify_levels ignores
    the tag (the device-side check is then read back as for any other tensor)."""
    tag = getattr(img, "_pcl_levels", None)
    return tag is not None and torch.is_tensor(img) and tag == img._version and not EXPERIMENT.verify_levels


def default_depth(n, H, W, stride=0):
    """(depth_h, depth_w, tau, stride) of pcl_depth_default: the scatter-min depth mask's grid, tolerance and occluder stride for an
    n-point cloud seen in an H x W panorama (>= 12 occluder samples per cell, never finer than the panorama; tau = 3.5 pi / depth_h
    in [0.02, 0.15]; stride 0: the largest of 1, 2, 4 that keeps depth_h >= 128)."""
    dh, dw, tau, st = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_float(0), ctypes.c_int(0)
    _lib.check(_lib.load().pcl_depth_default(int(n), int(H), int(W), int(stride), ctypes.byref(dh), ctypes.byref(dw), ctypes.byref(tau),
                                             ctypes.byref(st)), "pcl_depth_default")
    return dh.value, dw.value, tau.value, st.value


def default_depth_res(n, H, W, stride=0):
    return default_depth(n, H, W, stride)[:2]


def depth_tau_rule(depth_h):
    """the tolerance pcl_depth_default attaches to a grid of depth_h rows"""
    return min(max(3.5 * 3.141592653589793 / max(int(depth_h), 1), 0.02), 0.15)


def _depth_args(n, H, W, depth_res, depth_tau, depth_stride=None):
    """(depth_h, depth_w, tau, stride) from the optional cfg values: missing pieces come from pcl_depth_default; a given grid
    without a tolerance gets the rule's tolerance FOR THAT GRID, without a stride every point builds the z-buffer."""
    if depth_res is None:
        dh, dw, tau, st = default_depth(n, H, W, int(depth_stride or 0))
    else:
        dh, dw = int(depth_res[0]), int(depth_res[1])
        tau, st = depth_tau_rule(dh), int(depth_stride or 1)
    if depth_tau is not None:
        tau = float(depth_tau)
    return dh, dw, float(tau), st


def sampling_loss(cloud, pano, trans, rot, with_grad=True, visible=None, depth=None):
    """(B, 8) float tensor on the GPU: loss, count, dL/dt(3), dL/d(yaw, pitch, roll).
    visible: (B, n) uint8 mask in packed point order.  depth: True, or a dict with optional depth_res / depth_tau / depth_stride — the
    scatter-min depth mask of the SAME poses is built and looked up inside the launch (pcl_sampling_loss_depth)."""
    lib = _lib.load()
    trans, rot = _dev(trans).reshape(-1, 3), _dev(rot).reshape(-1, 3)
    B = int(trans.shape[0])
    if rot.shape[0] != B:
        raise ValueError("trans and rot must have the same number of rows")
    out = torch.empty(B, _lib.RESULT_STRIDE, dtype=F32, device=trans.device)
    if depth:
        if visible is not None:
            raise ValueError("sampling_loss: pass either a byte mask (visible) or depth, not both")
        d = depth if isinstance(depth, dict) else {}
        dh, dw, tau, st = _depth_args(cloud.n, pano.H, pano.W, d.get("depth_res"), d.get("depth_tau"), d.get("depth_stride"))
        ws_bytes = lib.pcl_loss_depth_workspace_bytes(cloud.n, B, pano.H, pano.W, dh, dw, st)
        if ws_bytes == 0:
            raise _lib.PiccoloHipError("pcl_loss_depth_workspace_bytes: invalid depth grid %dx%d" % (dh, dw))
        ws = _bytes(ws_bytes)
        _lib.check(lib.pcl_sampling_loss_depth(_ptr(cloud.data), cloud.n, _ptr(pano.data), pano.fmt, pano.H, pano.W, _ptr(trans), _ptr(rot), B,
                                               1 if with_grad else 0, dh, dw, tau, st, _ptr(out), _ptr(ws), ws_bytes, _stream()),
                   "pcl_sampling_loss_depth")
        return out
    ws_bytes = lib.pcl_loss_workspace_bytes(cloud.n, B)
    ws = _bytes(ws_bytes)
    vis = None
    if visible is not None:
        vis = _dev(visible, torch.uint8).reshape(B, cloud.n)
    _lib.check(lib.pcl_sampling_loss(_ptr(cloud.data), cloud.n, _ptr(pano.data), pano.fmt, pano.H, pano.W, _ptr(trans), _ptr(rot), B,
                                     1 if with_grad else 0, _ptr(vis), _ptr(out), _ptr(ws), ws_bytes, _stream()),
               "pcl_sampling_loss")
    return out


TRIM_MAX_ROT = 1024        # pcl_trim_groups: rotations per table (include/piccolo_hip.h)


class TrimGroups:
    """Classes of equal (pitch, roll) of an (R, 3) rotation table, built on the device (pcl_trim_groups) — what trim_loss_table
    needs from the rotation grid.  Built once per grid: the group count is read back here (one 4-byte D2H copy), so that the
    per-image launches are sized exactly."""

    def __init__(self, rot):
        lib = _lib.load()
        self.rot = _dev(rot).reshape(-1, 3)
        self.R = int(self.rot.shape[0])
        if self.R <= 0:
            raise ValueError("empty rotation table")
        self.data = _bytes(lib.pcl_trim_groups_bytes(self.R))
        _lib.check(lib.pcl_trim_groups(_ptr(self.rot), self.R, _ptr(self.data), _stream()), "pcl_trim_groups")
        self.ngroups = int(self.data[:4].view(torch.int32).item())


class TrimOrder:
    """The trim launch's row-sorted work list (pcl_trim_order): which (cloud chunk, slot) item each block evaluates, ranked by the
    panorama row the chunk lands in so that an XCD's L2 holds a band of the texture for all the poses.  Depends on the cloud, the
    candidate grid and the panorama's size / texel layout — not on the query image: build once per room, pass to trim_loss_table[s]
    (scheduling only: tables are bit-identical with or without it)."""

    def __init__(self, cloud, pano_shape, trans, groups):
        """pano_shape = (H, W, texel format code) of the panoramas the launches will read"""
        lib = _lib.load()
        trans = _dev(trans).reshape(-1, 3)
        H, W, fmt = pano_shape
        self.key = (cloud.n, int(trans.shape[0]), groups.ngroups, int(H), int(W), int(fmt))
        self.data = _bytes(lib.pcl_trim_order_bytes(cloud.n, self.key[1], groups.ngroups))
        nws = lib.pcl_trim_order_workspace_bytes(cloud.n, self.key[1], groups.ngroups)
        ws = _bytes(nws)
        _lib.check(lib.pcl_trim_order(_ptr(cloud.data), cloud.n, int(fmt), int(H), int(W), _ptr(trans), self.key[1], _ptr(groups.rot), groups.R,
                                      _ptr(groups.data), groups.ngroups, _ptr(self.data), _ptr(ws), nws, _stream()), "pcl_trim_order")


def trim_loss_table(cloud, pano, trans, groups, return_count=False, order=None):
    """utils.py:484-499 for all pairs: (K, R) float GPU tensor loss_table[i, j] = forward-only sampling loss of (trans[i], rot[j]),
    rotations of one (pitch, roll) class sharing the projection (csrc/pcl_trim.hip).  order: a TrimOrder of this cloud / grid."""
    lib = _lib.load()
    trans = _dev(trans).reshape(-1, 3)
    K = int(trans.shape[0])
    table = torch.empty(K, groups.R, dtype=F32, device=trans.device)
    count = torch.empty(K, groups.R, dtype=F32, device=trans.device) if return_count else None
    nws = lib.pcl_trim_loss_workspace_bytes(cloud.n, K, groups.ngroups)
    ws = _bytes(nws)
    _lib.check(lib.pcl_trim_loss(_ptr(cloud.data), cloud.n, _ptr(pano.data), pano.fmt, pano.H, pano.W, _ptr(trans), K, _ptr(groups.rot),
                                 groups.R, _ptr(groups.data), groups.ngroups, _ptr(order.data) if order is not None else None, _ptr(table), _ptr(count),
                                 _ptr(ws), nws, _stream()),
               "pcl_trim_loss")
    return (table, count) if return_count else table


TRIM_MAX_IMAGES = 32       # pcl_trim_loss_images: query images per launch


def trim_loss_tables(cloud, panos, trans, groups, return_count=False, order=None):
    """trim_loss_table for several query images of one room in ONE launch: (I, K, R) float GPU tensor; image i's table has the
    bits of trim_loss_table(cloud, panos[i], ...) (same chunks of the cloud).  panos: list of Pano of one size / texel format."""
    lib = _lib.load()
    trans = _dev(trans).reshape(-1, 3)
    K, I = int(trans.shape[0]), len(panos)
    p0 = panos[0]
    if any((p.H, p.W, p.fmt) != (p0.H, p0.W, p0.fmt) for p in panos):
        raise ValueError("all panoramas of a launch must share size and texel format")
    table = torch.empty(I, K, groups.R, dtype=F32, device=trans.device)
    count = torch.empty(I, K, groups.R, dtype=F32, device=trans.device) if return_count else None
    for i0 in range(0, I, TRIM_MAX_IMAGES):
        part = panos[i0:i0 + TRIM_MAX_IMAGES]
        nws = lib.pcl_trim_loss_images_workspace_bytes(cloud.n, K, groups.ngroups, len(part))
        ws = _bytes(nws)
        arr = (ctypes.c_void_p * len(part))(*[p.data.data_ptr() for p in part])
        _lib.check(lib.pcl_trim_loss_images(_ptr(cloud.data), cloud.n, arr, len(part), p0.fmt, p0.H, p0.W, _ptr(trans), K, _ptr(groups.rot),
                                            groups.R, _ptr(groups.data), groups.ngroups, _ptr(order.data) if order is not None else None, _ptr(table[i0:]),
                                            _ptr(count[i0:]) if return_count else None, _ptr(ws), nws, _stream()), "pcl_trim_loss_images")
    return (table, count) if return_count else table


def select_poses(values, n_keep, trans, rot, largest=False, rot_per_trans=0, return_idx=False):
    """The selections of the initialisation stage (utils.py:500-505 / :583-586) in one launch: values (M,) or (P, M) -> the
    n_keep best rows (trans[idx // rot_per_trans], rot[idx % rot_per_trans]) — or (trans[idx], rot[idx]) when rot_per_trans is 0
    — in rank order, as ((P,) n_keep, 3) tensors.  Stable order; NaN ranks last.  With (P, M) values, trans / rot are either shared
    2-D tables or (P, rows, 3) stacks."""
    lib = _lib.load()
    values = _dev(values)
    single = values.dim() == 1
    v = values.reshape(1, -1) if single else values.reshape(values.shape[0], -1)
    P, M = int(v.shape[0]), int(v.shape[1])
    trans, rot = _dev(trans), _dev(rot)
    stride = 0
    if trans.dim() == 3:
        if rot.dim() != 3 or trans.shape[0] != P or rot.shape[:2] != trans.shape[:2]:
            raise ValueError("select_poses: per-problem pose tables must be (P, rows, 3) for both trans and rot")
        stride = int(trans.shape[1])
    trans, rot = trans.reshape(-1, 3), rot.reshape(-1, 3)
    rows_t = trans.shape[0] if stride == 0 else stride
    rows_r = rot.shape[0] if stride == 0 else stride
    need_t = (M + rot_per_trans - 1) // rot_per_trans if rot_per_trans > 0 else M
    need_r = rot_per_trans if rot_per_trans > 0 else M
    if rows_t < need_t or rows_r < need_r:
        raise ValueError("select_poses: %d values need %d translations and %d rotations" % (M, need_t, need_r))
    n_keep = int(n_keep)
    ot = torch.empty(P, n_keep, 3, dtype=F32, device=v.device)
    orr = torch.empty(P, n_keep, 3, dtype=F32, device=v.device)
    oi = torch.empty(P, n_keep, dtype=torch.int32, device=v.device) if return_idx else None
    _lib.check(lib.pcl_select_poses(_ptr(v), P, M, n_keep, 1 if largest else 0, _ptr(trans), _ptr(rot), int(rot_per_trans), stride,
                                    _ptr(ot), _ptr(orr), _ptr(oi), _stream()), "pcl_select_poses")
    if single:
        ot, orr, oi = ot[0], orr[0], (oi[0] if oi is not None else None)
    return (ot, orr, oi) if return_idx else (ot, orr)


SELECT_MAX_KEEP = 1024      # pcl_select_poses: winners per problem (include/piccolo_hip.h)


def hist_trim_scores(img, cloud, trans, rot, num_split_h, num_split_w, batch=64, return_parts=False, splat=False):
    """Histogram-intersection score of every candidate pose (utils.py:510-588): (K,) GPU tensor, higher is better.
    `cloud` is a packed Cloud.  Candidates are processed `batch` at a time.  Workspace per candidate: the point lists of the
    tile-binned render (48 bytes per point in the worst case, plus the tiles' run tables — 2 bytes per point for a 2048 x 1024 panorama: 3.2 GB for 64
    candidates at 1M points; HBM is there to be used),
    or, where that path does not apply (more than 4096 image tiles, ...) or does not fit, H * W * 8 bytes for the z-buffer of the
    splat path.  If the allocation fails the batch is halved, and the last resort is the splat path's small workspace.
    return_parts: (scores, inter, nproj, nimg).  splat=True: the z-buffer splat path on purpose (its small workspace selects it in
    pcl_hist_trim_scores; the tests compare the two renderers bit for bit)."""
    lib = _lib.load()
    img = _dev(img)
    trans, rot = _dev(trans).reshape(-1, 3), _dev(rot).reshape(-1, 3)
    K, (H, W) = int(trans.shape[0]), (int(img.shape[0]), int(img.shape[1]))
    if splat:
        size_of = lambda b: lib.pcl_hist_trim_workspace_bytes(b, H, W, num_split_h, num_split_w)          # noqa: E731
    else:
        size_of = lambda b: lib.pcl_hist_trim_workspace_bytes_n(cloud.n, b, H, W, num_split_h, num_split_w)  # noqa: E731
    nblk = (num_split_h - 2) * num_split_w
    inter = torch.empty(K, nblk, dtype=F32, device=img.device)
    nproj = torch.empty(K, nblk, dtype=torch.int32, device=img.device)
    nimg = torch.empty(nblk, dtype=torch.int32, device=img.device)
    # one batch for the 64 survivors of the loss trim at 1M points (four batches of 16: 2.0 instead of 1.7 ms; 0.8 instead of
    # 0.5 ms at 167k points); a batch is kept within ~8 GB (10M points: 16 candidates at a time).
    # pcl_hist_trim_workspace_bytes_n is the binned path's size where that path will be taken, the splat path's otherwise.
    per_cand = max(size_of(1), 1)
    batch = max(1, min(batch, K, int(HIST_BATCH_BYTES // per_cand)))
    if size_of(batch) == 0:
        raise ValueError("hist_trim_scores: need num_split_h >= 3 and blocks of at least one pixel")
    ws = None
    while ws is None:
        nws = size_of(batch)
        try:
            ws = _bytes(nws)
        except torch.cuda.OutOfMemoryError:
            torch.cuda.empty_cache()
            if batch > 1:
                batch = (batch + 1) // 2
                continue
            nws = lib.pcl_hist_trim_workspace_bytes(1, H, W, num_split_h, num_split_w)      # the z-buffer splat path
            ws = _bytes(nws)
    for k0 in range(0, K, batch):
        k1 = min(k0 + batch, K)
        _lib.check(lib.pcl_hist_trim_scores(_ptr(cloud.data), cloud.n, _ptr(img), H, W, _ptr(trans[k0:k1]),
                                            _ptr(rot[k0:k1]), k1 - k0, num_split_h, num_split_w, _ptr(inter[k0:k1]),
                                            _ptr(nproj[k0:k1]), _ptr(nimg), _ptr(ws), nws, _stream()), "pcl_hist_trim_scores")
    # a block with no pixels ends its block row (the reference `break`s there, utils.py:568-571)
    scores = torch.empty(K, dtype=F32, device=img.device)
    _lib.check(lib.pcl_hist_trim_reduce(_ptr(inter), _ptr(nproj), _ptr(nimg), K, num_split_h, num_split_w, _ptr(scores), _stream()),
               "pcl_hist_trim_reduce")
    if return_parts:
        return scores, inter, nproj, nimg
    return scores


HIST_MAX_IMAGES = 32       # pcl_hist_trim_scores_images: query images per call
# bytes of point lists one histogram-trim call may hold
HIST_BATCH_BYTES = 8e9


def hist_trim_scores_images(imgs, cloud, trans, rot, num_split_h, num_split_w):
    """hist_trim_scores for several query images of one room in ONE set of launches: imgs = list of I (H, W, 3) float GPU images of
    one size, trans / rot (I, K, 3): image i's K candidates.  -> (I, K) scores, row i what hist_trim_scores(imgs[i], ...) returns
    (bit for bit: same keys, same integer counts, one carry-over chain per image).  Images go through in groups that keep the
    point lists within ~8 GB."""
    lib = _lib.load()
    imgs = [_dev(im) for im in imgs]
    trans, rot = _dev(trans), _dev(rot)
    I, K = int(trans.shape[0]), int(trans.shape[1])
    H, W = int(imgs[0].shape[0]), int(imgs[0].shape[1])
    if len(imgs) != I or any(tuple(im.shape) != (H, W, 3) or not im.is_contiguous() for im in imgs):
        raise ValueError("hist_trim_scores_images: one contiguous (H, W, 3) image per row of candidates, all of one size")
    nblk = (num_split_h - 2) * num_split_w
    dev = imgs[0].device
    inter = torch.empty(I * K, nblk, dtype=F32, device=dev)
    nproj = torch.empty(I * K, nblk, dtype=torch.int32, device=dev)
    nimg = torch.empty(I, nblk, dtype=torch.int32, device=dev)
    scores = torch.empty(I, K, dtype=F32, device=dev)
    per_image = lib.pcl_hist_trim_images_workspace_bytes(cloud.n, 1, K, H, W, num_split_h, num_split_w)
    if per_image == 0:
        raise ValueError("hist_trim_scores_images: need num_split_h >= 3 and blocks of at least one pixel")
    group = max(1, min(HIST_MAX_IMAGES, I, int(HIST_BATCH_BYTES // per_image)))
    t2, r2 = trans.reshape(I * K, 3).contiguous(), rot.reshape(I * K, 3).contiguous()
    i0 = 0
    while i0 < I:
        m = min(group, I - i0)
        nws = lib.pcl_hist_trim_images_workspace_bytes(cloud.n, m, K, H, W, num_split_h, num_split_w)
        try:
            ws = _bytes(nws)
        except torch.cuda.OutOfMemoryError:
            # like hist_trim_scores: fewer images at a time, and in the end the per-image path with its own fallbacks
            torch.cuda.empty_cache()
            if group > 1:
                group = (group + 1) // 2
                continue
            scores[i0] = hist_trim_scores(imgs[i0], cloud, trans[i0], rot[i0], num_split_h, num_split_w)
            i0 += 1
            continue
        arr = (ctypes.c_void_p * m)(*[im.data_ptr() for im in imgs[i0:i0 + m]])
        _lib.check(lib.pcl_hist_trim_scores_images(_ptr(cloud.data), cloud.n, arr, m, K, H, W, _ptr(t2[i0 * K:]), _ptr(r2[i0 * K:]), num_split_h,
                                                   num_split_w, _ptr(inter[i0 * K:]), _ptr(nproj[i0 * K:]), _ptr(nimg[i0:]), _ptr(ws), nws, _stream()),
                   "pcl_hist_trim_scores_images")
        _lib.check(lib.pcl_hist_trim_reduce_images(_ptr(inter[i0 * K:]), _ptr(nproj[i0 * K:]), _ptr(nimg[i0:]), m, K, num_split_h, num_split_w,
                                                   _ptr(scores[i0:]), _stream()), "pcl_hist_trim_reduce_images")
        i0 += m
    return scores


def depth_mask(cloud, trans, rot, resolution, tau=None, stride=1):
    """(B, n) uint8 GPU tensor in PACKED point order: scatter-min visibility of every point for every pose on a z-buffer grid of
    `resolution` = (depth_h, depth_w) cells (the DEPTH grid — see default_depth — not the panorama's size) built from every
    stride-th packed point; tau None: the rule's tolerance for that grid."""
    lib = _lib.load()
    trans, rot = _dev(trans).reshape(-1, 3), _dev(rot).reshape(-1, 3)
    B, (H, W) = int(trans.shape[0]), (int(resolution[0]), int(resolution[1]))
    if tau is None:
        tau = depth_tau_rule(H)
    vis = torch.empty(B, cloud.n, dtype=torch.uint8, device=trans.device)
    nws = lib.pcl_depth_workspace_bytes(B, H, W)
    ws = _bytes(nws)
    _lib.check(lib.pcl_depth_mask(_ptr(cloud.data), cloud.n, _ptr(trans), _ptr(rot), B, H, W, float(tau), int(stride), _ptr(vis), _ptr(ws), nws,
                                  _stream()), "pcl_depth_mask")
    return vis


class ColorTemplate:
    """The point colours sorted per channel (3, n): what color_match needs from the cloud, built once per cloud."""

    def __init__(self, rgb):
        lib = _lib.load()
        rgb = _dev(rgb).reshape(-1, 3)
        self.n = int(rgb.shape[0])
        self.data = torch.empty(3, self.n, dtype=F32, device=rgb.device)
        nws = lib.pcl_color_template_workspace_bytes(self.n)
        ws = _bytes(nws)
        _lib.check(lib.pcl_color_template_build(_ptr(rgb), self.n, _ptr(self.data), _ptr(ws), nws, _stream()),
                   "pcl_color_template_build")


def color_match(img, template):
    """color_utils.color_match (color_utils.py:146-234) on the GPU: img (H,W,3) with levels k/255 -> matched (H,W,3).
    `template` is a ColorTemplate of the point colours."""
    lib = _lib.load()
    known = _known_levels(img)
    img = _dev(img)
    H, W = int(img.shape[0]), int(img.shape[1])
    out = torch.empty_like(img)
    flag = torch.zeros(1, dtype=torch.int32, device=img.device)
    nws = lib.pcl_color_workspace_bytes()
    ws = _bytes(nws)
    _lib.check(lib.pcl_color_match(_ptr(img), H, W, _ptr(template.data), template.n, _ptr(out), _ptr(flag), _ptr(ws), nws,
                                   _stream()), "pcl_color_match")
    if not known and int(flag.item()):
        raise ValueError("color_match: the panorama must hold levels k/255 (an image file's uint8 / 255); "
                         "found a non-black pixel channel in between")
    return out


def color_mod(img, rgb, num_bins=256):
    """color_utils.color_mod (color_utils.py:7-65) on the GPU: -> (img (H,W,3), rgb (n,3)) after joint luma equalisation."""
    lib = _lib.load()
    img, rgb = _dev(img), _dev(rgb).reshape(-1, 3)
    H, W = int(img.shape[0]), int(img.shape[1])
    out_img, out_rgb = torch.empty_like(img), torch.empty_like(rgb)
    nws = lib.pcl_color_workspace_bytes()
    ws = _bytes(nws)
    _lib.check(lib.pcl_color_mod(_ptr(img), H, W, _ptr(rgb), int(rgb.shape[0]), int(num_bins), _ptr(out_img), _ptr(out_rgb),
                                 _ptr(ws), nws, _stream()), "pcl_color_mod")
    return out_img, out_rgb


def histogram(img, mask, channels, normalize=True, eps=0.0):
    """color_utils.histogram of one image (color_utils.py:68-103): flat float histogram of c0*c1*c2 bins on the GPU."""
    lib = _lib.load()
    img = _dev(img).reshape(-1, 3)
    mask = _dev(mask != 0, torch.uint8).reshape(-1)
    c0, c1, c2 = (int(c) for c in channels)
    nws = lib.pcl_histogram_workspace_bytes(c0, c1, c2)
    if nws == 0:
        raise ValueError("histogram: bad bin counts %r" % (channels,))
    ws = _bytes(nws)
    hist = torch.empty(c0 * c1 * c2, dtype=F32, device=img.device)
    _lib.check(lib.pcl_histogram(_ptr(img), _ptr(mask), int(img.shape[0]), c0, c1, c2, int(bool(normalize)), float(eps),
                                 _ptr(hist), _ptr(ws), nws, _stream()), "pcl_histogram")
    return hist


def histogram_intersection(h1, h2):
    """(B, nbins) x (B, nbins) -> (B,) sums of element-wise minima (color_utils.py:122-144)."""
    lib = _lib.load()
    h1, h2 = _dev(h1), _dev(h2)
    B, nbins = int(h1.shape[0]), int(h1.shape[1])
    out = torch.empty(B, dtype=F32, device=h1.device)
    _lib.check(lib.pcl_histogram_intersection(_ptr(h1), _ptr(h2), B, nbins, _ptr(out), _stream()), "pcl_histogram_intersection")
    return out


def quantile_box(xyz, q):
    """(6,) GPU tensor: x_lo, x_hi, y_lo, y_hi, z_lo, z_hi — utils.py:208-229 on the three columns."""
    lib = _lib.load()
    xyz = _dev(xyz)
    box = torch.empty(6, dtype=F32, device=xyz.device)
    ws = _bytes(lib.pcl_quantile_workspace_bytes())
    _lib.check(lib.pcl_quantile_box(_ptr(xyz), int(xyz.shape[0]), float(q), _ptr(box), _ptr(ws), _stream()), "pcl_quantile_box")
    return box


class GradientDescent:
    """On-device GD refinement of B candidates (Adam + ReduceLROnPlateau + clamp), pcl_gd_* of the C ABI."""

    def __init__(self, cloud, pano, trans, rot, box, lr=0.1, patience=5, factor=0.9, batch_mode=True, depth_mask=False,
                 depth_tau=None, depth_res=None, depth_stride=None, fuse=None):
        """fuse None: pcl_gd_plan's rule (one launch per iteration for launches whose blocks are all resident); False: always the
        two-launch form (bit-identical; tests and measurements)."""
        lib = _lib.load()
        self.cloud, self.pano = cloud, pano
        trans, rot = _dev(trans).reshape(-1, 3), _dev(rot).reshape(-1, 3)
        self.B = int(trans.shape[0])
        self.box = _dev(box).reshape(6)
        dh, dw, tau, st = _depth_args(cloud.n, pano.H, pano.W, depth_res, depth_tau, depth_stride) if depth_mask else (0, 0, 0.0, 0)
        self.hyper = _lib.GdHyper(float(lr), float(factor), int(patience), _lib.GD_BATCH if batch_mode else _lib.GD_SEQUENTIAL,
                                  1 if depth_mask else 0, float(tau), int(dh), int(dw), int(st), -1 if fuse is False else 0, 0)
        self.state = _bytes(lib.pcl_gd_state_bytes(self.B))
        self.ws_bytes = lib.pcl_gd_workspace_bytes(cloud.n, self.B, pano.H, pano.W, ctypes.byref(self.hyper))
        if self.ws_bytes == 0:
            raise _lib.PiccoloHipError("pcl_gd_workspace_bytes: invalid arguments (depth grid %dx%d?)" % (dh, dw))
        self.ws = _bytes(self.ws_bytes)
        _lib.check(lib.pcl_gd_init(_ptr(self.state), _ptr(trans), _ptr(rot), self.B, ctypes.byref(self.hyper), _stream()),
                   "pcl_gd_init")

    def run(self, num_iter, history=False, timer=None):
        lib = _lib.load()
        hist = torch.empty(num_iter, self.B, dtype=F32, device=self.state.device) if history else None
        _lib.check(lib.pcl_gd_run(_ptr(self.cloud.data), self.cloud.n, _ptr(self.pano.data), self.pano.fmt, self.pano.H, self.pano.W,
                                  _ptr(self.state), self.B, _ptr(self.box), ctypes.byref(self.hyper), int(num_iter),
                                  _ptr(hist), _ptr(self.ws), self.ws_bytes, timer.handle if timer else None, _stream()),
                   "pcl_gd_run")
        return hist

    def run_graph(self, num_iter):
        """Same as run(num_iter) but the 2 * num_iter launches are captured into one hipGraph and replayed: the host
        enqueues one graph instead of 200 kernels per refinement (pcl_gd_run neither allocates nor synchronises, so it
        is capture-safe).  The instantiated graph is cached per num_iter; replaying it continues from the current
        state, exactly like calling run() again."""
        cache = self.__dict__.setdefault("_graphs", {})
        g = cache.get(num_iter)
        if g is None:
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=self.state.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    self.run(num_iter)
            torch.cuda.current_stream().wait_stream(side)
            cache[num_iter] = g
            # capture does not execute: fall through to the first replay
        g.replay()

    def reset(self, trans, rot):
        """Re-initialise the optimiser state for new starting poses (same cloud / panorama / B): lets one captured
        graph serve many refinements."""
        lib = _lib.load()
        trans, rot = _dev(trans).reshape(-1, 3), _dev(rot).reshape(-1, 3)
        assert trans.shape[0] == self.B
        _lib.check(lib.pcl_gd_init(_ptr(self.state), _ptr(trans), _ptr(rot), self.B, ctypes.byref(self.hyper), _stream()),
                   "pcl_gd_init")

    def set_panos(self, panos):
        """Candidate b samples panos[b] (a list of B Pano objects, all the size / texel format of self.pano): lets the
        candidates of several query images share one launch chain.  Call after __init__ / reset()."""
        lib = _lib.load()
        assert len(panos) == self.B
        for p in panos:
            if (p.H, p.W, p.fmt) != (self.pano.H, self.pano.W, self.pano.fmt):
                raise ValueError("all panoramas of a launch must share size and texel format")
        self._panos = list(panos)                      # keep them alive
        self.set_pano_table(torch.tensor([p.data.data_ptr() for p in panos], dtype=torch.int64, device=self.state.device),
                            images=len({id(p) for p in panos}))

    def set_pano_table(self, table, images=0):
        """Same with a ready-made device tensor of B packed-panorama addresses (int64); the caller keeps the Pano
        objects alive.  No host work besides the launch.  `images`: how many query images the table names (image i's candidates
        a contiguous range) — a hint for the block -> XCD mapping of the launches, results do not depend on it."""
        self.hyper.images = int(images)
        assert table.dtype == torch.int64 and table.numel() == self.B and table.is_cuda
        _lib.check(_lib.load().pcl_gd_set_panos(_ptr(self.state), _ptr(table), self.B, _stream()), "pcl_gd_set_panos")
        self._pano_table = table

    def set_pano_groups(self, panos):
        """Candidates [i * B / I, (i + 1) * B / I) sample panos[i] (I Pano objects of the size / texel format of self.pano, B
        divisible by I).  Unlike set_panos / set_pano_table nothing is copied to the device: the addresses are kernel arguments."""
        lib = _lib.load()
        I = len(panos)
        if I <= 0 or self.B % I:
            raise ValueError("set_pano_groups: %d candidates do not split into %d images" % (self.B, I))
        for p in panos:
            if (p.H, p.W, p.fmt) != (self.pano.H, self.pano.W, self.pano.fmt):
                raise ValueError("all panoramas of a launch must share size and texel format")
        self._panos = list(panos)                      # keep them alive
        self.hyper.images = I                          # (mapping hint for pcl_gd_run: the XCDs split the images)
        arr = (ctypes.c_uint64 * I)(*[p.data.data_ptr() for p in panos])
        _lib.check(lib.pcl_gd_set_pano_groups(_ptr(self.state), arr, I, self.B // I, _stream()), "pcl_gd_set_pano_groups")

    def winner(self, nimages=1, leaf_trans=None, leaf_rot=None):
        """(nimages, 16) GPU tensor, per image of B / nimages candidates the one omniloc_batch returns (omniloc.py:271-277):
        post-step t (3), R (9), last loss, yaw / pitch / roll.  leaf_trans / leaf_rot: contiguous float32 GPU tensors of B x 3
        that receive every candidate's leaf parameters (the reference optimises the caller's rows in place)."""
        lib = _lib.load()
        if nimages <= 0 or self.B % nimages:
            raise ValueError("winner: %d candidates do not split into %d images" % (self.B, nimages))
        out = torch.empty(nimages, 16, dtype=F32, device=self.state.device)
        for t in (leaf_trans, leaf_rot):
            if t is not None and not (t.is_cuda and t.dtype == F32 and t.is_contiguous() and t.numel() == 3 * self.B):
                raise ValueError("winner: leaf buffers must be contiguous float32 GPU tensors of B x 3")
        _lib.check(lib.pcl_gd_winner(_ptr(self.state), nimages, self.B // nimages, _ptr(out), _ptr(leaf_trans), _ptr(leaf_rot), _stream()),
                   "pcl_gd_winner")
        return out

    def step_from_grads(self, loss, grad):
        """Teacher-forcing hook (tests): ONE optimiser step of every candidate from a GIVEN loss (B,) and gradient (B, 6) =
        dL/d(t0, t1, t2, yaw, pitch, roll) through the epilogue's own Adam / ReduceLROnPlateau / clamp code (pcl_gd_step_from_grads)."""
        loss, grad = _dev(loss).reshape(self.B), _dev(grad).reshape(self.B, 6)
        scratch = torch.empty(self.B * 8, dtype=F32, device=self.state.device)
        _lib.check(_lib.load().pcl_gd_step_from_grads(_ptr(self.state), self.B, _ptr(loss), _ptr(grad), _ptr(self.box), ctypes.byref(self.hyper),
                                                      _ptr(scratch), _stream()), "pcl_gd_step_from_grads")

    def result(self):
        """(B, 16): fwd t(3), fwd ypr(3), leaf t(3), leaf ypr(3), last loss, lr, scheduler num_bad_epochs, scheduler best."""
        lib = _lib.load()
        out = torch.empty(self.B, _lib.GD_RESULT_STRIDE, dtype=F32, device=self.state.device)
        _lib.check(lib.pcl_gd_result(_ptr(self.state), self.B, _ptr(out), _stream()), "pcl_gd_result")
        return out


class KernelTimer:
    """HIP-event pairs around every fused loss+gradient launch of GradientDescent.run (measurement aid)."""

    def __init__(self, capacity, stride=1):
        self.handle = ctypes.c_void_p(_lib.load().pcl_timer_create(int(capacity)))
        if not self.handle:
            raise _lib.PiccoloHipError("pcl_timer_create failed")
        _lib.load().pcl_timer_set_stride(self.handle, int(stride))

    def reset(self):
        _lib.load().pcl_timer_reset(self.handle)

    def read(self):
        """(total kernel ms, launches) since the last reset; synchronises on the recorded events."""
        ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
        _lib.check(_lib.load().pcl_timer_read(self.handle, ctypes.byref(ms), ctypes.byref(cnt)), "pcl_timer_read")
        return ms.value, cnt.value

    def calibrate(self, reps=64):
        """Median reading (ms) of an event pair with NOTHING between its two records, on the current stream: what a pair around
        a kernel over-reads.  Synchronises."""
        ms = ctypes.c_double(0)
        _lib.check(_lib.load().pcl_timer_calibrate(self.handle, int(reps), ctypes.byref(ms), _stream()), "pcl_timer_calibrate")
        return ms.value

    def __del__(self):
        try:
            if self.handle:
                _lib.load().pcl_timer_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def cloud2idx(xyz):
    lib = _lib.load()
    x = _dev(xyz)
    shp = x.shape
    flat = x.reshape(-1, 3)
    out = torch.empty(flat.shape[0], 2, dtype=F32, device=x.device)
    if flat.shape[0]:
        _lib.check(lib.pcl_cloud2idx(_ptr(flat), int(flat.shape[0]), _ptr(out), _stream()), "pcl_cloud2idx")
    return out.reshape(shp[:-1] + (2,))


def sample_from_img(pano, coord):
    lib = _lib.load()
    c = _dev(coord)
    shp = c.shape
    flat = c.reshape(-1, 2)
    out = torch.empty(flat.shape[0], 3, dtype=F32, device=c.device)
    if flat.shape[0]:
        _lib.check(lib.pcl_sample_from_img(_ptr(pano.data), pano.fmt, pano.H, pano.W, _ptr(flat), int(flat.shape[0]), _ptr(out), _stream()),
                   "pcl_sample_from_img")
    return out.reshape(shp[:-1] + (3,))


def cloud2idx_backward(xyz, grad_coord):
    """grad w.r.t. xyz (.., 3) of cloud2idx for the incoming gradient grad_coord (.., 2)."""
    lib = _lib.load()
    x, g = _dev(xyz), _dev(grad_coord)
    flat, gflat = x.reshape(-1, 3), g.reshape(-1, 2)
    if gflat.shape[0] != flat.shape[0]:
        raise ValueError("grad_coord must have one (gx, gy) per point")
    out = torch.empty_like(flat)
    if flat.shape[0]:
        _lib.check(lib.pcl_cloud2idx_backward(_ptr(flat), _ptr(gflat), int(flat.shape[0]), _ptr(out), _stream()), "pcl_cloud2idx_backward")
    return out.reshape(x.shape)


def sample_from_img_backward(pano, coord, grad_rgb, want_coord=True, want_img=False):
    """(grad_coord (.., 2) or None, grad_img (H, W, 3) or None) of sample_from_img for the incoming gradient grad_rgb (.., 3)."""
    lib = _lib.load()
    c, g = _dev(coord), _dev(grad_rgb)
    flat, gflat = c.reshape(-1, 2), g.reshape(-1, 3)
    if gflat.shape[0] != flat.shape[0]:
        raise ValueError("grad_rgb must have one colour per coordinate")
    gc = torch.empty_like(flat) if want_coord else None
    gi = torch.zeros(pano.H, pano.W, 3, dtype=F32, device=c.device) if want_img else None
    if flat.shape[0] and (want_coord or want_img):
        _lib.check(lib.pcl_sample_from_img_backward(_ptr(pano.data), pano.fmt, pano.H, pano.W, _ptr(flat), _ptr(gflat), int(flat.shape[0]),
                                                    _ptr(gc), _ptr(gi), _stream()), "pcl_sample_from_img_backward")
    elif gc is not None:
        gc.zero_()
    return (gc.reshape(c.shape) if gc is not None else None), gi


def rot_from_ypr(rot):
    lib = _lib.load()
    r = _dev(rot).reshape(-1, 3)
    out = torch.empty(r.shape[0], 9, dtype=F32, device=r.device)
    _lib.check(lib.pcl_rot_from_ypr(_ptr(r), int(r.shape[0]), _ptr(out), _stream()), "pcl_rot_from_ypr")
    return out.reshape(-1, 3, 3)


def transform_cloud(xyz, trans, rot):
    lib = _lib.load()
    x = _dev(xyz)
    t, r = _dev(trans).reshape(3), _dev(rot).reshape(3)
    out = torch.empty_like(x)
    _lib.check(lib.pcl_transform_cloud(_ptr(x), int(x.shape[0]), _ptr(t), _ptr(r), _ptr(out), _stream()), "pcl_transform_cloud")
    return out


def make_pano(xyz_cam, rgb, resolution):
    """(H, W, 3) float GPU tensor = rgb*255 of the winning point per pixel (utils.py:134-205 semantics)."""
    lib = _lib.load()
    x, c = _dev(xyz_cam), _dev(rgb)
    H, W = int(resolution[0]), int(resolution[1])
    img = torch.empty(H, W, 3, dtype=F32, device=x.device)
    ws = _bytes(H * W * 8)
    _lib.check(lib.pcl_make_pano(_ptr(x), _ptr(c), int(x.shape[0]), H, W, _ptr(img), _ptr(ws), _stream()), "pcl_make_pano")
    return img


def scatter_min_depth(xyz_cam, resolution):
    """torch_scatter-style (zmin (H*W,), argmin (H*W,)) of point depth per make_pano pixel; empty -> (0, n)."""
    lib = _lib.load()
    x = _dev(xyz_cam)
    H, W = int(resolution[0]), int(resolution[1])
    n = int(x.shape[0])
    zbuf = _bytes(H * W * 8)
    zmin = torch.empty(H * W, dtype=F32, device=x.device)
    arg = torch.empty(H * W, dtype=torch.int64, device=x.device)
    _lib.check(lib.pcl_scatter_min_depth(_ptr(x), n, H, W, _ptr(zbuf), _stream()), "pcl_scatter_min_depth")
    _lib.check(lib.pcl_scatter_min_unpack(_ptr(zbuf), n, H, W, _ptr(zmin), _ptr(arg), _stream()), "pcl_scatter_min_unpack")
    return zmin, arg
