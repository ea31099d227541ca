"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/piccolo_hip.h
declares, the ctypes binding covers exactly that set, and size queries behave (no compute calls: there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import REPO

HEADER = os.path.join(REPO, "include", "piccolo_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcl_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from piccolo_amd import _lib, build
    build.build()                      # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_header_declares_the_path():
    names = declared_symbols()
    for must in ("pcl_cloud_pack", "pcl_pano_pack", "pcl_pano_pack_u8", "pcl_sampling_loss", "pcl_gd_init", "pcl_gd_run",
                 "pcl_gd_result", "pcl_cloud2idx", "pcl_sample_from_img", "pcl_quantile_box", "pcl_scatter_min_depth",
                 "pcl_make_pano", "pcl_rot_from_ypr"):
        assert must in names


def test_binding_covers_exactly_the_header(lib):
    from piccolo_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_library_exports_every_declared_symbol(lib):
    from piccolo_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.so_path()], text=True)
    exported = set(re.findall(r"\bT (pcl_[a-z0-9_]+)", out))
    assert set(declared_symbols()) <= exported
    for name in declared_symbols():
        assert getattr(lib, name) is not None


def test_library_is_a_gfx950_code_object(lib):
    from piccolo_amd import _lib
    blob = open(_lib.so_path(), "rb").read()
    assert b"gfx950" in blob and b"pcl_loss_kernel" in blob


def test_size_queries_and_argument_checks(lib):
    from piccolo_amd import _lib
    assert lib.pcl_abi_version() == _lib.ABI_VERSION
    assert lib.pcl_cloud_stride(1) == 256 and lib.pcl_cloud_stride(256) == 256 and lib.pcl_cloud_stride(257) == 512
    assert lib.pcl_cloud_bytes(1000) == 1024 * 6 * 4
    assert lib.pcl_pano_bytes(4, 8, _lib.PANO_F32) == 6 * 10 * 16 and lib.pcl_pano_bytes(4, 8, _lib.PANO_U8) == 6 * 10 * 4 and lib.pcl_pano_bytes(4, 8, _lib.PANO_F16) == 6 * 10 * 8
    assert lib.pcl_pano_bytes(4, 8, 7) == 0 and lib.pcl_pano_bytes(0, 8, 0) == 0
    # two copies (fused iterations read one while they write the other); the depth mask keeps nothing in the state (ABI 7)
    assert lib.pcl_gd_state_bytes(32) == 2 * 32 * (160 + 64)
    ws1, ws2 = lib.pcl_loss_workspace_bytes(1_000_000, 32), lib.pcl_loss_workspace_bytes(1_000_000, 256)
    assert 0 < ws1 < ws2 < 64 << 20
    assert lib.pcl_loss_workspace_bytes(0, 32) == 0
    # bad arguments are rejected before anything touches a device
    assert lib.pcl_sampling_loss(None, 10, None, 0, 4, 8, None, None, 1, 1, None, None, None, 0, None) == -1
    assert lib.pcl_gd_run(None, 10, None, 0, 4, 8, None, 1, None, None, 1, None, None, 0, None, None) == -1
    assert lib.pcl_quantile_box(None, 10, 0.05, None, None, None) == -1
    assert b"invalid argument" in lib.pcl_error_string(-1) and b"workspace" in lib.pcl_error_string(-2)


def test_product_fails_loudly_without_gpu_or_library(monkeypatch, tmp_path):
    """No CPU fallback: without a GPU every op raises; without the .so the loader raises."""
    import torch
    from piccolo_amd import _lib, ops
    if not torch.cuda.is_available():
        with pytest.raises(_lib.PiccoloHipError):
            ops.cloud2idx(torch.zeros(4, 3))
        with pytest.raises(_lib.PiccoloHipError):
            ops.Cloud(torch.zeros(4, 3), torch.zeros(4, 3))
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "so_path", lambda: str(tmp_path / "missing.so"))
    with pytest.raises(_lib.PiccoloHipError):
        _lib.load()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under piccolo_amd/, dropin/ or main.py may reference it."""
    bad = []
    for root in ("piccolo_amd", "dropin"):
        for dp, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    text = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|oracle\.|pcl_oracle", text, re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
    assert "oracle" not in open(os.path.join(REPO, "main.py")).read()


def test_every_entry_point_rejects_null_arguments_before_touching_a_device():
    """Error behaviour of the whole boundary: every status-returning function called with null pointers and zero sizes
    answers PCL_EINVAL (-1) — no crash, no HIP call (this runs without a GPU) — and every *_bytes sizing function answers 0
    for an empty problem."""
    import ctypes
    from piccolo_amd import _lib
    lib = _lib.load()
    skipped, wrong = [], []
    for name, (res, args) in sorted(_lib.SIGNATURES.items()):
        if name in ("pcl_abi_version", "pcl_error_string", "pcl_source_hash", "pcl_library_hash", "pcl_color_workspace_bytes", "pcl_quantile_workspace_bytes"):
            continue                                        # no arguments to get wrong / constant
        zero = []
        for a in args:
            if a in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(a, "contents"):
                zero.append(None)
            elif a in (ctypes.c_float, ctypes.c_double):
                zero.append(0.0)
            else:
                zero.append(0)
        rc = getattr(lib, name)(*zero)
        if res is ctypes.c_int:
            if rc != -1:
                wrong.append((name, rc))
        elif res is ctypes.c_size_t:
            if rc != 0:
                wrong.append((name, rc))
        elif res is ctypes.c_int64:
            if rc > 0:                                      # cloud_stride(0) = 0; text readers: negative status
                wrong.append((name, rc))
        else:
            skipped.append(name)
    assert not wrong, wrong
    # handle-returning / void functions: called with nulls above without a crash, nothing to compare
    assert set(skipped) <= {"pcl_timer_create", "pcl_timer_destroy", "pcl_timer_reset", "pcl_timer_set_stride"}, skipped


def test_depth_default_grid_and_tolerance_host_only():
    """pcl_depth_default (host-only): >= 12 occluder samples per cell, depth_w = 2 depth_h, a multiple of 8, never finer than the
    panorama; tau = 3.5 pi / depth_h in [0.02, 0.15]; the occluder stride: the largest of 1, 2, 4 that keeps depth_h >= 128
    (tools/depth_recall.py: what recall / precision that buys); and the workspace sizes that follow from it."""
    import ctypes
    import math
    from piccolo_amd import _lib
    lib = _lib.load()

    def dflt(n, H, W, stride=0):
        h, w, t, st = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_float(0), ctypes.c_int(0)
        assert lib.pcl_depth_default(n, H, W, stride, ctypes.byref(h), ctypes.byref(w), ctypes.byref(t), ctypes.byref(st)) == 0
        return h.value, w.value, t.value, st.value
    for n, H, W in ((1_000_000, 1024, 2048), (166_667, 1024, 2048), (10_000_000, 2048, 4096), (40_000, 128, 256), (100_000, 256, 512)):
        for stride in (0, 1, 2):
            h, w, t, st = dflt(n, H, W, stride)
            assert st == stride or stride == 0
            m = (n + st - 1) // st                                     # occluder samples
            assert w == 2 * h and h % 8 == 0 and h <= H and w <= W
            assert m / (h * w) >= 12 and m / ((h + 8) * 2 * (h + 8)) < 12, (n, st, h, w)
            assert abs(t - min(max(3.5 * math.pi / h, 0.02), 0.15)) < 1e-6
            if stride == 0:
                assert st in (1, 2, 4) and (st == 1 or h >= 128) and (st == 4 or dflt(n, H, W, 2 * st)[0] < 128), (n, st, h)
    assert dflt(1_000_000, 1024, 2048, 1)[:2] == (200, 400)         # every point: make_pano's own default resolution (utils.py:134)
    assert dflt(1_000_000, 1024, 2048) == (144, 288, dflt(1_000_000, 1024, 2048)[2], 2)
    assert dflt(166_667, 1024, 2048)[3] == 1 and dflt(10_000_000, 2048, 4096)[3] == 4
    assert dflt(1_000_000, 64, 128)[:2] == (64, 128)                # never finer than the panorama
    assert lib.pcl_depth_default(1000, 64, 128, 0, None, None, None, None) == 0 and lib.pcl_depth_default(0, 64, 128, 0, None, None, None, None) == -1
    hyper = _lib.GdHyper(0.1, 0.8, 5, _lib.GD_BATCH, 0, 0.0, 0, 0, 0, 0)
    plain = lib.pcl_gd_workspace_bytes(1_000_000, 32, 1024, 2048, ctypes.byref(hyper))
    hyper.depth_mask = 1
    masked = lib.pcl_gd_workspace_bytes(1_000_000, 32, 1024, 2048, ctypes.byref(hyper))
    assert 0 < plain < masked and masked - plain == 2 * 32 * 144 * 288 * 4       # two sets of 5.3 MB of z-buffers (round 4: 268 MB + 32 MB of byte masks)
    hyper.depth_h, hyper.depth_w = 1024, 2048
    assert lib.pcl_gd_workspace_bytes(1_000_000, 32, 1024, 2048, ctypes.byref(hyper)) - plain == 2 * 32 * 1024 * 2048 * 4
    hyper.depth_h, hyper.depth_w = 200, 0                           # half a grid: invalid
    assert lib.pcl_gd_workspace_bytes(1_000_000, 32, 1024, 2048, ctypes.byref(hyper)) == 0
    plain_loss = lib.pcl_loss_workspace_bytes(1_000_000, 32)
    assert lib.pcl_loss_depth_workspace_bytes(1_000_000, 32, 1024, 2048, 0, 0, 0) > plain_loss
    # the DEFAULT grid follows the occluder stride, so the size query takes the stride the call will be given —
    # 0 x 0 with stride 1 is 200 x 400 cells per pose, not the 144 x 288 of the default stride 2
    align = lambda v: (v + 255) & ~255                                # noqa: E731
    for stride, (dh, dw) in ((0, (144, 288)), (2, (144, 288)), (1, (200, 400)), (4, dflt(1_000_000, 1024, 2048, 4)[:2])):
        got = lib.pcl_loss_depth_workspace_bytes(1_000_000, 32, 1024, 2048, 0, 0, stride)
        assert got == align(plain_loss) + 32 * dh * dw * 4, (stride, got)
    assert lib.pcl_loss_depth_workspace_bytes(1_000_000, 32, 1024, 2048, 300, 600, 1) == align(plain_loss) + 32 * 300 * 600 * 4
    assert lib.pcl_loss_depth_workspace_bytes(1_000_000, 32, 1024, 2048, 0, 0, 65) == 0                    # stride out of range
    assert lib.pcl_loss_depth_workspace_bytes(1_000_000, 1, 1024, 2048, 2, 1 << 24, 1) == 0                # a side of 2^24 cells: rejected
    hyper.depth_h, hyper.depth_w = 2, 1 << 24
    assert lib.pcl_gd_workspace_bytes(1_000_000, 1, 1024, 2048, ctypes.byref(hyper)) == 0


def test_shipped_library_reads_no_environment_variable(lib):
    """seventeen getenv knobs lived in the product library; `PCL_G` / `PCL_BLOCKS` changed the chunking — hence
    the rounding of every loss — and the workspace sizes of a C ABI that otherwise has no hidden state.  They now exist only in the
    EXPERIMENTS build (-DPCL_EXPERIMENTS, lib/libpiccolo_hip_exp.so): the shipped library holds no PCL_* string,
    calls no getenv of its own, and its size queries ignore the environment."""
    from piccolo_amd import _lib, build
    blob = subprocess.check_output(["strings", _lib.so_path()], text=True).splitlines()
    assert not [ln for ln in blob if ln.startswith("PCL_")], [ln for ln in blob if ln.startswith("PCL_")][:5]
    # (the library still IMPORTS getenv: rocprim's radix-sort headers, used by pcl_pack.hip / pcl_color.hip, read their own variables;
    #  no translation unit of csrc/ calls it outside PCL_EXPERIMENTS — checked on the sources)
    import glob
    for path in glob.glob(os.path.join(REPO, "piccolo_amd", "csrc", "*")):
        text = open(path).read()
        if "getenv" in text:
            assert os.path.basename(path) == "pcl_device.h" and text.index("#ifdef PCL_EXPERIMENTS") < text.index("getenv") < text.index("#else"), path
    before = (lib.pcl_loss_workspace_bytes(1_000_000, 32), lib.pcl_trim_loss_workspace_bytes(1_000_000, 75, 6))
    knobs = dict(PCL_G="1", PCL_BLOCKS="512", PCL_XCD_RUNS="4", PCL_TRIM_CHUNKS="128", PCL_GD_FUSE_BLOCKS="0")
    code = ("from piccolo_amd import _lib; lib = _lib.load(); "
            "print(lib.pcl_loss_workspace_bytes(1000000, 32), lib.pcl_trim_loss_workspace_bytes(1000000, 75, 6))")
    out = subprocess.check_output([os.sys.executable, "-c", code], env=dict(os.environ, **knobs), cwd=REPO, text=True)
    assert tuple(int(v) for v in out.split()[-2:]) == before
    # the experiments build is the one that listens (same sources, one more -D)
    exp = build.build_experiments()
    exp_strings = subprocess.check_output(["strings", exp], text=True).splitlines()
    assert {"PCL_G", "PCL_BLOCKS", "PCL_GD_FUSE_BLOCKS"} <= set(exp_strings)
    out = subprocess.check_output([os.sys.executable, "-c", code], env=dict(os.environ, PCL_SO=exp, **knobs), cwd=REPO, text=True)
    assert tuple(int(v) for v in out.split()[-2:]) != before
    # ... and nothing under piccolo_amd/ reads PCL_* from the environment except PCL_SO (which library to load) and the build's flags
    import glob
    hits = []
    for path in glob.glob(os.path.join(REPO, "piccolo_amd", "*.py")):
        for i, ln in enumerate(open(path), 1):
            if "environ" in ln and "PCL_" in ln and "PCL_SO" not in ln and "PCL_HIPCC_FLAGS" not in ln:
                hits.append((os.path.basename(path), i))
    assert not hits, hits


def test_gd_plan_reports_the_decomposition_host_only():
    """pcl_gd_plan (host-only query): chunks, poses per block and whether an iteration is one launch (all chunk x group blocks
    resident at once: the reference's shipped 167k-point / 6-candidate shape, cfg 1) or two (cfg 2 and larger)."""
    import ctypes
    from piccolo_amd import _lib
    lib = _lib.load()

    def plan(n, B):
        c, g, f = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
        assert lib.pcl_gd_plan(n, B, ctypes.byref(c), ctypes.byref(g), ctypes.byref(f)) == 0
        return c.value, g.value, f.value
    if True:
        c, g, f = plan(166_667, 6)
        assert g == 2 and f == 1 and c % 8 == 0 and c * 3 <= 1024            # 3 pose groups x chunks: all resident
        assert plan(100_000, 1)[1:] == (1, 1)                                # cfg 1: one pose per block, fused
        c, g, f = plan(1_000_000, 32)
        assert (c, g, f) == (256, 2, 0)                                      # cfg 2: 4096 blocks, two launches per iteration
        assert plan(1_000_000, 256)[2] == 0 and plan(10_000_000, 64)[2] == 0
        # pcl_gd_hyper.fuse = -1: never one launch per iteration (the form the parity tests compare the fused one with); the
        # environment does not reach the shipped library (no getenv: test_shipped_library_reads_no_environment_variable)
        fz = ctypes.c_int(-1)
        hy = _lib.GdHyper(0.1, 0.8, 5, 1, 0, 0.0, 0, 0, 0, -1, 0)
        assert lib.pcl_gd_plan_hyper(166_667, 6, ctypes.byref(hy), None, None, ctypes.byref(fz)) == 0 and fz.value == 0
        os.environ["PCL_GD_FUSE_BLOCKS"] = "0"
        try:
            assert plan(166_667, 6)[2] == 1
        finally:
            del os.environ["PCL_GD_FUSE_BLOCKS"]
    assert lib.pcl_gd_plan(0, 6, None, None, None) == -1 and lib.pcl_gd_plan(1000, 4, None, None, None) == 0
