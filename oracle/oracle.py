"""ctypes front-end of oracle/pcl_oracle.c (TEST INFRASTRUCTURE — see oracle/__init__.py).

numpy in, numpy out; `dtype` picks the f32 or f64 instantiation of the C restatement.
The reference file:line each entry point follows is cited in pcl_oracle_impl.inc.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpcl_oracle.so")
_lib = None

_c_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
            for f in ("pcl_oracle.c", "pcl_oracle_impl.inc")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_max_threads.restype = _int
    return _lib


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32", np.float32
    if dtype == np.float64:
        return "_f64", np.float64
    raise TypeError(dtype)


def _arr(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return a.ctypes.data_as(_c_p) if a is not None else None


def max_threads():
    return int(lib().orc_max_threads())


def rot_from_ypr(ypr, dtype=np.float32):
    s, dt = _sfx(dtype)
    y = _arr(ypr, dt).reshape(3)
    R = np.empty(9, dt)
    getattr(lib(), "orc_rot_from_ypr" + s)(_p(y), _p(R))
    return R.reshape(3, 3)


def cloud2idx(xyz, dtype=None):
    s, dt = _sfx(dtype or np.asarray(xyz).dtype)
    x = _arr(xyz, dt)
    shp = x.shape
    x = x.reshape(-1, 3)
    out = np.empty((x.shape[0], 2), dt)
    getattr(lib(), "orc_cloud2idx" + s)(_p(x), _i64(x.shape[0]), _p(out))
    return out.reshape(shp[:-1] + (2,))


def sample_from_img(img, coord, dtype=None):
    s, dt = _sfx(dtype or np.asarray(img).dtype)
    im = _arr(img, dt)
    c = _arr(coord, dt)
    shp = c.shape
    c = c.reshape(-1, 2)
    H, W, _ = im.shape
    out = np.empty((c.shape[0], 3), dt)
    getattr(lib(), "orc_sample_from_img" + s)(_p(im), _int(H), _int(W), _p(c), _i64(c.shape[0]), _p(out))
    return out.reshape(shp[:-1] + (3,))


def cloud2idx_backward(xyz, grad_coord, dtype=None):
    """d(sum(grad_coord * cloud2idx(xyz))) / d xyz — what autograd gives the reference's utils.cloud2idx."""
    s, dt = _sfx(dtype or np.asarray(xyz).dtype)
    x = _arr(xyz, dt)
    shp = x.shape
    x = x.reshape(-1, 3)
    g = _arr(grad_coord, dt).reshape(-1, 2)
    out = np.empty_like(x)
    getattr(lib(), "orc_cloud2idx_backward" + s)(_p(x), _p(g), _i64(x.shape[0]), _p(out))
    return out.reshape(shp)


def sample_from_img_backward(img, coord, grad_out, dtype=None, want_img=True):
    """(grad_coord, grad_img) of utils.sample_from_img for the incoming gradient grad_out (.., 3)."""
    s, dt = _sfx(dtype or np.asarray(img).dtype)
    im = _arr(img, dt)
    c = _arr(coord, dt)
    shp = c.shape
    c = c.reshape(-1, 2)
    g = _arr(grad_out, dt).reshape(-1, 3)
    H, W, _ = im.shape
    gc = np.empty_like(c)
    gi = np.zeros_like(im) if want_img else None
    getattr(lib(), "orc_sample_from_img_backward" + s)(_p(im), _int(H), _int(W), _p(c), _p(g), _i64(c.shape[0]), _p(gc), _p(gi))
    return gc.reshape(shp), gi


def sampling_loss(xyz, rgb, img, trans, rot, dtype=np.float32, grad=True, visible=None, nthreads=0):
    """Loss (+ gradient) of B candidate poses.  trans (B,3), rot (B,3)=[yaw,pitch,roll].

    Returns dict(loss (B,), count (B,), grad_t (B,3), grad_ypr (B,3)); grads omitted if grad=False.
    """
    s, dt = _sfx(dtype)
    x, c, im = _arr(xyz, dt), _arr(rgb, dt), _arr(img, dt)
    t = _arr(trans, dt).reshape(-1, 3)
    r = _arr(rot, dt).reshape(-1, 3)
    B, n = t.shape[0], x.shape[0]
    H, W, _ = im.shape
    loss = np.empty(B, dt)
    count = np.empty(B, np.int64)
    gt = np.empty((B, 3), dt) if grad else None
    gr = np.empty((B, 3), dt) if grad else None
    vis = None
    if visible is not None:
        vis = np.ascontiguousarray(visible, dtype=np.uint8).reshape(B, n)
    getattr(lib(), "orc_sampling_loss" + s)(_p(x), _p(c), _i64(n), _p(im), _int(H), _int(W), _p(t), _p(r), _int(B),
                                            _p(vis), _p(loss), _p(count), _p(gt), _p(gr), _int(nthreads))
    out = dict(loss=loss, count=count)
    if grad:
        out.update(grad_t=gt, grad_ypr=gr)
    return out


def quantile(x, q, dtype=None):
    """(x[int(n*q)], x[int(n*(1-q))]) of sorted x — utils.py:208-229."""
    s, dt = _sfx(dtype or np.asarray(x).dtype)
    v = _arr(x, dt).reshape(-1)
    out = np.empty(2, dt)
    getattr(lib(), "orc_quantile" + s)(_p(v), _i64(v.shape[0]), _i64(1), ctypes.c_double(q), _p(out))
    return out[0], out[1]


def quantile_box(xyz, q, dtype=np.float32):
    """[[x_lo,x_hi],[y_lo,y_hi],[z_lo,z_hi]] — the clamp box of omniloc.py:53-55 / :245-247."""
    x = np.asarray(xyz)
    return np.array([quantile(x[:, k], q, dtype) for k in range(3)], dtype=dtype)


def make_pano(xyz_cam, rgb, resolution=(200, 400), dtype=np.float32, return_aux=False):
    """image*255 as float (H,W,3) — utils.py:134-205; aux = (owner (H,W), contested (H,W))."""
    s, dt = _sfx(dtype)
    x, c = _arr(xyz_cam, dt), _arr(rgb, dt)
    H, W = int(resolution[0]), int(resolution[1])
    img = np.empty((H, W, 3), dt)
    owner = np.empty((H, W), np.int64)
    cont = np.empty((H, W), np.uint8)
    getattr(lib(), "orc_make_pano" + s)(_p(x), _p(c), _i64(x.shape[0]), _int(H), _int(W), _p(img), _p(owner), _p(cont))
    if return_aux:
        return img, owner, cont.astype(bool)
    return img


def pano_pixels(xyz_cam, resolution, dtype=np.float32):
    """(row, col) int32 of each camera-frame point in make_pano's pixel grid (utils.py:158-165)."""
    s, dt = _sfx(dtype)
    x = _arr(xyz_cam, dt)
    H, W = int(resolution[0]), int(resolution[1])
    row = np.empty(x.shape[0], np.int32)
    col = np.empty(x.shape[0], np.int32)
    getattr(lib(), "orc_pano_pixels" + s)(_p(x), _i64(x.shape[0]), _int(H), _int(W), _p(row), _p(col))
    return row, col


# make_pano pass order idx8,7,6,5,4,3,2,1,centre (utils.py:190-198) as (drow, dcol)
PANO_PASSES = [(0, -1), (0, 1), (-1, -1), (-1, 0), (-1, 1), (1, -1), (1, 0), (1, 1), (0, 0)]


def make_pano_candidates(xyz_cam, resolution):
    """Per pixel, the set of points written by the LAST pass that touches it (any of them may win in the
    reference: index_put_ with duplicate indices is undefined).  Returns list-of-arrays indexed by pixel."""
    H, W = int(resolution[0]), int(resolution[1])
    row, col = pano_pixels(xyz_cam, resolution)
    last_pass = np.full(H * W, -1, np.int32)
    pix_of = []
    for p, (dr, dc) in enumerate(PANO_PASSES):
        pix = np.clip(row + dr, 0, H - 1).astype(np.int64) * W + np.clip(col + dc, 0, W - 1)
        pix_of.append(pix)
        last_pass[pix] = p
    cands = [[] for _ in range(H * W)]
    for p in range(9):
        sel = np.nonzero(last_pass[pix_of[p]] == p)[0]
        for i in sel:
            cands[pix_of[p][i]].append(i)
    return cands


def make_pano_u8(xyz_cam, rgb, resolution=(200, 400)):
    """uint8 image exactly as make_pano(return_torch=False) returns it (astype(uint8) truncation, utils.py:203)."""
    return make_pano(xyz_cam, rgb, resolution, np.float32).astype(np.uint8)


def scatter_min_depth(xyz_cam, resolution, dtype=np.float32):
    s, dt = _sfx(dtype)
    x = _arr(xyz_cam, dt)
    H, W = int(resolution[0]), int(resolution[1])
    zmin = np.empty(H * W, dt)
    arg = np.empty(H * W, np.int64)
    getattr(lib(), "orc_scatter_min_depth" + s)(_p(x), _i64(x.shape[0]), _int(H), _int(W), _p(zmin), _p(arg))
    return zmin, arg
