"""Seeded synthetic scenes for tests and bench (SURVEY.md §8d recipe).

There are no datasets on the GPU box, so every measurement runs on a coloured
box room: 8 x 6 x 3 m centred at the origin, points drawn uniformly by area on
the six faces, colour a smooth function of position.  A query panorama is the
room rendered from a ground-truth pose; the renderer is passed in by the caller
(the HIP ``make_pano`` in bench.py / smoke, the oracle's in CPU tests, the
reference's own in tests/golden/gen_goldens.py) so this module depends on
numpy only.

Pose convention follows the reference throughout: ``p = R (x - t)`` with
``R = RZ(yaw) RY(pitch) RX(roll)`` (/root/reference/utils.py:425-453).
"""
import numpy as np

ROOM = np.array([8.0, 6.0, 3.0])
_K = np.array([[1.3, 0.7, 2.1], [0.9, 1.9, 0.5], [2.3, 1.1, 1.7]])
_PHI = np.array([0.3, 1.1, 2.0])


def rot_from_ypr_np(ypr):
    """R = RZ(yaw) RY(pitch) RX(roll), float64 (utils.py:425-453 convention)."""
    y, p, r = [float(v) for v in ypr]
    cy, sy, cp, sp, cr, sr = np.cos(y), np.sin(y), np.cos(p), np.sin(p), np.cos(r), np.sin(r)
    RX = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    RY = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    RZ = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return RZ @ RY @ RX


def box_room(n_points, seed=0):
    """(xyz, rgb) float32 (N,3): coloured points on the faces of the box room."""
    rng = np.random.default_rng(seed)
    lx, ly, lz = ROOM
    areas = np.array([ly * lz, ly * lz, lx * lz, lx * lz, lx * ly, lx * ly])
    face = rng.choice(6, size=n_points, p=areas / areas.sum())
    u = rng.random(n_points)
    v = rng.random(n_points)
    xyz = np.empty((n_points, 3), dtype=np.float64)
    # faces 0/1: x = -+lx/2 ; 2/3: y = -+ly/2 ; 4/5: z = -+lz/2
    for f in range(6):
        m = face == f
        axis = f // 2
        sign = -1.0 if f % 2 == 0 else 1.0
        a, b = [k for k in range(3) if k != axis]
        xyz[m, axis] = sign * ROOM[axis] / 2
        xyz[m, a] = (u[m] - 0.5) * ROOM[a]
        xyz[m, b] = (v[m] - 0.5) * ROOM[b]
    rgb = 0.5 + 0.45 * np.sin(xyz @ _K.T + _PHI)
    return xyz.astype(np.float32), rgb.astype(np.float32)


# interior boxes of `furnished_room`: (centre, size) in metres — a pillar, a cabinet against a wall, a table-high block
_FURNITURE = (((1.2, 0.6, 0.0), (0.5, 0.5, 3.0)), ((-2.6, -1.8, -0.5), (1.2, 0.6, 2.0)), ((0.3, -1.4, -1.0), (1.6, 0.9, 1.0)))


def furnished_room(n_points, seed=0):
    """The box room with three interior boxes (a floor-to-ceiling pillar, a cabinet, a low block): NOT convex, so from any pose part
    of the walls / floor is hidden behind furniture — the case north_star's scatter-min depth mask exists for (a point that the
    query panorama does not show still projects into it and samples the colour of whatever is in front).  Points by area over the
    room's faces and the boxes' outer faces, same colour field as box_room."""
    rng = np.random.default_rng(seed)
    surfaces = []                                             # (axis, coordinate, (lo_a, hi_a), (lo_b, hi_b))
    for axis in range(3):
        a, b = [k for k in range(3) if k != axis]
        for sign in (-1.0, 1.0):
            surfaces.append((axis, sign * ROOM[axis] / 2, (-ROOM[a] / 2, ROOM[a] / 2), (-ROOM[b] / 2, ROOM[b] / 2)))
    for c, sz in _FURNITURE:
        for axis in range(3):
            a, b = [k for k in range(3) if k != axis]
            for sign in (-1.0, 1.0):
                surfaces.append((axis, c[axis] + sign * sz[axis] / 2, (c[a] - sz[a] / 2, c[a] + sz[a] / 2), (c[b] - sz[b] / 2, c[b] + sz[b] / 2)))
    areas = np.array([(s[2][1] - s[2][0]) * (s[3][1] - s[3][0]) for s in surfaces])
    which = rng.choice(len(surfaces), size=n_points, p=areas / areas.sum())
    u, v = rng.random(n_points), rng.random(n_points)
    xyz = np.empty((n_points, 3), dtype=np.float64)
    for i, (axis, coord, ra, rb) in enumerate(surfaces):
        m = which == i
        a, b = [k for k in range(3) if k != axis]
        xyz[m, axis] = coord
        xyz[m, a] = ra[0] + u[m] * (ra[1] - ra[0])
        xyz[m, b] = rb[0] + v[m] * (rb[1] - rb[0])
    rgb = 0.5 + 0.45 * np.sin(xyz @ _K.T + _PHI)
    return xyz.astype(np.float32), rgb.astype(np.float32)


def inside_furniture(t, margin=0.3):
    """True if position t is within `margin` of one of furnished_room's boxes (no camera there)."""
    return any(all(abs(float(t[k]) - c[k]) <= sz[k] / 2 + margin for k in range(3)) for c, sz in _FURNITURE)


def occluded_by_furniture(xyz, t, eps=1e-4):
    """(n,) bool, float64 slab test: the open segment from camera position t to xyz[i] crosses the interior of one of
    furnished_room's boxes — the ANALYTIC occlusion a depth mask is measured against (walls / floor behind furniture and the
    faces of a box that point away from the camera; the room itself is convex and hides nothing)."""
    P = np.asarray(xyz, np.float64)
    t = np.asarray(t, np.float64).reshape(3)
    d = P - t[None, :]
    out = np.zeros(len(P), bool)
    for c, sz in _FURNITURE:
        lo, hi = np.array(c) - np.array(sz) / 2, np.array(c) + np.array(sz) / 2
        with np.errstate(divide="ignore", invalid="ignore"):
            s0, s1 = (lo[None, :] - t[None, :]) / d, (hi[None, :] - t[None, :]) / d
        smin, smax = np.minimum(s0, s1), np.maximum(s0, s1)
        par = d == 0                                                  # parallel to a slab: inside it for every s, or never
        inside = (t[None, :] > lo[None, :]) & (t[None, :] < hi[None, :])
        smin = np.where(par, np.where(inside, -np.inf, np.inf), smin)
        smax = np.where(par, np.where(inside, np.inf, -np.inf), smax)
        enter, leave = smin.max(axis=1), smax.min(axis=1)
        out |= np.maximum(enter, 0.0) < np.minimum(leave, 1.0 - eps) - eps       # a piece of positive length strictly before the point
    return out


def gt_pose(seed):
    """Ground-truth (t (3,), ypr (3,)) for query image `seed`: t in the central half of the room."""
    rng = np.random.default_rng(10_000 + seed)
    t = (rng.random(3) - 0.5) * 0.5 * ROOM
    ypr = np.array([rng.random() * 2 * np.pi, rng.normal(0, 0.05), rng.normal(0, 0.05)])
    return t.astype(np.float32), ypr.astype(np.float32)


def start_poses(t_gt, ypr_gt, n_start, seed, sigma_t=0.3, sigma_r=0.15):
    """`n_start` perturbed starting poses around the ground truth: (trans (B,3), rot (B,3)=[yaw,pitch,roll])."""
    rng = np.random.default_rng(20_000 + seed)
    trans = t_gt[None, :] + rng.normal(0, sigma_t, size=(n_start, 3))
    rot = ypr_gt[None, :] + rng.normal(0, sigma_r, size=(n_start, 3))
    return trans.astype(np.float32), rot.astype(np.float32)


def transform_cloud(xyz, t, ypr):
    """R (x - t) in float64, returned float32 (N,3)."""
    R = rot_from_ypr_np(ypr)
    return ((xyz.astype(np.float64) - np.asarray(t, np.float64)[None, :]) @ R.T).astype(np.float32)


def pose_errors(t, R, t_gt, R_gt):
    """t-error (m) and R-error (deg) exactly as /root/reference/localize.py:239-247 computes them."""
    t_err = float(np.linalg.norm(np.asarray(t_gt, np.float64).reshape(3) - np.asarray(t, np.float64).reshape(3)))
    tr = float(np.trace(np.asarray(R, np.float64).T @ np.asarray(R_gt, np.float64)))
    if tr < -1:
        tr = -2 - tr
    elif tr > 3:
        tr = 6 - tr
    r_err = float(np.rad2deg(np.abs(np.arccos((tr - 1) / 2))))
    return t_err, r_err


def quantise_like_image_file(pano255):
    """torch (H,W,3) float panorama in [0,255] -> float image exactly as `uint8 image / 255.` gives it on the host
    (the reference divides on the CPU, localize.py:167-170).  A GPU division is not IEEE-exact on ROCm, so the levels
    are mapped through a 256-entry table computed on the CPU."""
    import torch
    lut = (torch.arange(256, dtype=torch.float32) / 255.0).to(pano255.device)
    return mark_levels(lut[pano255.clamp(0, 255).to(torch.int64)])


def mark_levels(img):
    """Tag a float image tensor whose every value is exactly k/255 BY CONSTRUCTION (a decoded 8-bit image divided by 255 on the
    host, a table lookup of such levels): the packers then take the level formats (fp16 / RGBA8 texels) without reading the
    device-side exactness flag back — one blocking D2H copy per query image less.  Untagged tensors are checked as before."""
    img._pcl_levels = img._version          # (an in-place change of the tensor afterwards voids the tag)
    return img
