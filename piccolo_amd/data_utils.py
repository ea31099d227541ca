"""Drop-in for the reference's data_utils.py (imported as a module at localize.py:11): dataset text clouds and ground-truth
pose conventions — the data formats on the input side of the hot path.

    read_stanford(filepath, sample_rate=1), read_omniscenes(filepath, sample_rate=1)      data_utils.py:16-43, :138-163
    obtain_gt_stanford(area_num, img_name), obtain_gt_omniscenes(full_img_path)           data_utils.py:46-135, :166-182

The text parser is native (piccolo_amd/csrc/pcl_io.hip, host threads over an mmap; `pcl_cloud_txt_*` in
include/piccolo_hip.h) and returns the same float64 (N,3) arrays as the reference's pandas call.  `load_cloud` is the
build's extension: float32 tensors on the GPU with a binary side-car cache, so that a multi-million-point cloud is parsed
once per dataset, not once per run.
"""
import json
import os

import numpy as np

from . import _lib


def _read_table(filepath, cols=6):
    """(N, cols) float64 array of a whitespace-separated text file (pandas.read_table(...).values)."""
    lib = _lib.load()
    path = os.fsencode(filepath)
    rows = lib.pcl_cloud_txt_rows(path)
    if rows < 0:
        raise OSError(-rows, os.strerror(-rows), filepath)
    data = np.empty((rows, cols), np.float64)
    rc = lib.pcl_cloud_txt_read(path, rows, cols, data.ctypes.data, 0)
    if rc <= -1000:
        raise ValueError("%s: line %d does not hold %d numbers" % (filepath, -rc - 1000, cols))
    if rc == -1:
        raise ValueError("%s: the file changed while it was read" % filepath)
    if rc < 0:
        raise OSError(-rc, os.strerror(-rc), filepath)
    return data


def _read_cloud(filepath, sample_rate):
    data = _read_table(filepath)
    xyz = data[:, :3]
    rgb = data[:, 3:] / 255.
    if sample_rate > 1.0:                                   # data_utils.py:36-41: the reference draws from numpy's global RNG
        perm = np.random.permutation(xyz.shape[0])
        num_samples = int(xyz.shape[0] / sample_rate)
        idx = perm[:num_samples]
        xyz = xyz[idx]
        rgb = rgb[idx]
    return xyz, rgb


def read_stanford(filepath, sample_rate=1):
    """Stanford2D-3D-S point cloud: xyz (N,3), rgb (N,3) in [0,1], float64 (data_utils.py:16-43)."""
    return _read_cloud(filepath, sample_rate)


def read_omniscenes(filepath, sample_rate=1):
    """OmniScenes point cloud, same format (data_utils.py:138-163)."""
    return _read_cloud(filepath, sample_rate)


def load_cloud(filepath, sample_rate=1, cache=True):
    """Extension: (xyz, rgb) float32 CUDA tensors of a dataset cloud.  With `cache`, the parsed float64 table is kept next
    to the text file as `<file>.pcl.npy` and reused while it is newer than the text."""
    import torch
    from . import ops
    side = filepath + ".pcl.npy"
    data = None
    if cache and os.path.exists(side) and os.path.getmtime(side) >= os.path.getmtime(filepath):
        data = np.load(side, mmap_mode="r")
    if data is None:
        data = _read_table(filepath)
        if cache:
            try:
                np.save(side, data)
            except OSError:
                pass                                        # read-only dataset directory: parse again next time
    xyz, rgb = data[:, :3], data[:, 3:] / 255.
    if sample_rate > 1.0:
        idx = np.random.permutation(xyz.shape[0])[:int(xyz.shape[0] / sample_rate)]
        xyz, rgb = xyz[idx], rgb[idx]
    dev = ops.device()
    return (torch.from_numpy(np.ascontiguousarray(xyz)).float().to(dev),
            torch.from_numpy(np.ascontiguousarray(rgb)).float().to(dev))


def _euler_xyz_matrix(angles):
    """scipy Rotation.from_euler('xyz', angles).as_matrix(): extrinsic rotations about x, then y, then z,
    R = Rz(c) Ry(b) Rx(a) (data_utils.py:78-79)."""
    a, b, c = (float(v) for v in angles)
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
    ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
    rz = np.array([[cc, -sc, 0], [sc, cc, 0], [0, 0, 1]])
    return rz @ ry @ rx


_FLIP = np.array([[-1, 0, 0], [0, -1, 0], [0, 0, 1]])


def _camera_rot(cam_rot):
    """data_utils.py:78-86: camera axes (z, x, y) -> columns, inverted."""
    r = _euler_xyz_matrix(cam_rot)
    rot = np.zeros([3, 3])
    rot[:, 0] = r[:, 2]
    rot[:, 1] = r[:, 0]
    rot[:, 2] = r[:, 1]
    return np.linalg.inv(rot)


def obtain_gt_stanford(area_num, img_name, root="./data/stanford/pose"):
    """Ground-truth translation (3,1) and rotation (3,3) of a Stanford2D-3D-S panorama (data_utils.py:46-135).
    area_num < 10: pose json of the area; otherwise (`area_num` = 10 x area, aligned rooms) the room's 3x4 alignment
    file is applied on top.  `root` defaults to the reference's relative path."""
    splits = img_name.split('_')
    camera_id, room_type, room_id = splits[1], splits[2], splits[3]
    area = area_num if area_num < 10 else area_num // 10
    pose_file = os.path.join(root, 'area_{}'.format(area),
                             'camera_{}_{}_{}_frame_equirectangular_domain_pose.json'.format(camera_id, room_type, room_id))
    with open(pose_file) as f:
        pose = json.load(f)
    cam_loc = np.array(pose['camera_location'])
    gt_trans = np.array([[cam_loc[0]], [cam_loc[1]], [cam_loc[2]]])
    rot = _camera_rot(pose['final_camera_rotation'])
    if area_num < 10:
        return gt_trans, np.matmul(_FLIP, rot)              # always 180 degrees about z (data_utils.py:88-90)
    transformation_mat = np.loadtxt(os.path.join(root, 'area_{}'.format(area_num), '{}_{}.txt'.format(room_type, room_id)))
    rot_mat, trans_mat = transformation_mat[:, :3], transformation_mat[:, 3:]
    gt_rot = np.matmul(_FLIP, np.matmul(rot, np.linalg.inv(rot_mat)))
    gt_trans = np.matmul(rot_mat, gt_trans - trans_mat)
    return gt_trans, gt_rot


def obtain_gt_omniscenes(full_img_path):
    """OmniScenes: the 3x4 [R | t] text file next to the panorama (data_utils.py:166-182)."""
    pose_file = full_img_path.replace('pano', 'pose').replace('.jpg', '.txt')
    gt_mat = np.loadtxt(pose_file)
    return gt_mat[:, 3:], gt_mat[:, :3]
