"""Colour preprocessing: color_match / color_mod / histogram.

CPU part  : oracle/color.py against the goldens produced by running the reference (g13: color_match, pure torch;
            g14: color_mod with the two OpenCV conversions injected from the oracle — those conversions themselves are
            parity-unpinned, OpenCV is absent from the build image) and known answers of the fixed-point YCrCb pair.
GPU part  : the HIP path through the C ABI against the same goldens and against the oracle on larger seeded inputs,
            plus size-independent properties at full panorama size."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import color as ocolor

MATCH_CASES = ("full_q", "gaps_q", "full_c", "gaps_c")
# float32 interpolation of O(1) values; sin() of the pixel weights differs by an ulp between numpy / ATen / ocml
MATCH_TOL = 1e-6


# ----------------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("case", MATCH_CASES)
def test_oracle_color_match_golden(case):
    g = load_golden("g13_color_match.npz")
    out = ocolor.color_match(g[case + "_img"], g[case + "_rgb"])
    assert np.abs(out - g[case + "_out"]).max() <= MATCH_TOL
    black = ~ocolor.nonblack_mask(g[case + "_img"])
    assert np.array_equal(out[black], g[case + "_img"][black])            # black pixels are copied through


@pytest.mark.parametrize("case", ("b256", "b64"))
def test_oracle_color_mod_golden(case):
    g = load_golden("g14_color_mod.npz")
    img, rgb = ocolor.color_mod(g[case + "_img"], g[case + "_rgb"], int(g[case + "_bins"]))
    assert np.array_equal(img, g[case + "_out_img"])                        # byte arithmetic: exact
    assert np.array_equal(rgb, g[case + "_out_rgb"])


def test_ycrcb_known_answers():
    """Primaries and greys under OpenCV's YCrCb definition (Y = .299R+.587G+.114B, Cr = (R-Y).713+128 saturated,
    Cb = (B-Y).564+128), and the round-trip error bound."""
    rgb = np.array([[0, 0, 0], [255, 255, 255], [128, 128, 128], [255, 0, 0], [0, 255, 0], [0, 0, 255]], np.uint8)
    ycc = ocolor.rgb2ycrcb_u8(rgb)
    assert ycc.tolist() == [[0, 128, 128], [255, 128, 128], [128, 128, 128], [76, 255, 85], [150, 21, 43], [29, 107, 255]]
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, size=(20000, 3)).astype(np.uint8)
    back = ocolor.ycrcb2rgb_u8(ocolor.rgb2ycrcb_u8(x)).astype(int)
    assert np.abs(back - x.astype(int)).max() <= 2                            # 8-bit quantisation of Cr/Cb
    grey = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1)
    assert np.array_equal(ocolor.ycrcb2rgb_u8(ocolor.rgb2ycrcb_u8(grey)), grey)


# ----------------------------------------------------------------------------------------------------- GPU
def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def cu():
    from piccolo_amd import color_utils as c
    from piccolo_amd import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return c


def scene(seed, H, W, n, levels=None, continuous=True):
    rng = np.random.default_rng(seed)
    levels = np.arange(256) if levels is None else np.asarray(levels)
    img8 = rng.choice(levels.astype(np.uint8), size=(H, W, 3))
    img8[H // 3:H // 3 + max(H // 16, 1), W // 5:W // 2] = 0
    img = img8.astype(np.float32) / np.float32(255)
    if continuous:
        rgb = (rng.random((n, 3)) ** 0.6).astype(np.float32)
    else:
        rgb = rng.integers(5, 256, size=(n, 3)).astype(np.float32) / np.float32(255)
    return img, rgb


@pytest.mark.gpu
@pytest.mark.parametrize("case", MATCH_CASES)
def test_color_match_golden(cu, case):
    g = load_golden("g13_color_match.npz")
    out = cu.color_match(T(g[case + "_img"]), T(g[case + "_rgb"])).cpu().numpy()
    assert np.abs(out - g[case + "_out"]).max() <= MATCH_TOL
    black = ~ocolor.nonblack_mask(g[case + "_img"])
    assert np.array_equal(out[black], g[case + "_img"][black])


@pytest.mark.gpu
@pytest.mark.parametrize("continuous", [True, False])
@pytest.mark.parametrize("levels", [None, list(range(7, 250, 3))])
def test_color_match_oracle(cu, continuous, levels):
    img, rgb = scene(5, 128, 256, 200_000, levels, continuous)
    out = cu.color_match(T(img), T(rgb)).cpu().numpy()
    ref = ocolor.color_match(img, rgb)
    assert np.abs(out - ref).max() <= MATCH_TOL


@pytest.mark.gpu
def test_color_match_cpu_tensors_and_cache(cu):
    """CPU tensors in -> CPU tensor out (the reference's harness may run on either device); a second image against
    the same colours reuses the sorted template."""
    img, rgb = scene(6, 32, 64, 5000)
    rgb_t = torch.from_numpy(rgb)
    a = cu.color_match(torch.from_numpy(img), rgb_t)
    assert a.device.type == "cpu" and a.dtype == torch.float32
    assert len(cu._templates) >= 1
    tmpl = cu._template(rgb_t)
    img2, _ = scene(7, 32, 64, 10)
    b = cu.color_match(torch.from_numpy(img2), rgb_t)
    assert cu._template(rgb_t) is tmpl
    assert np.abs(b.numpy() - ocolor.color_match(img2, rgb)).max() <= MATCH_TOL


@pytest.mark.gpu
def test_color_match_rejects_unquantised(cu):
    img, rgb = scene(8, 16, 32, 1000)
    img = img.copy()
    img[3, 3, 1] = 0.5003
    with pytest.raises(ValueError):
        cu.color_match(T(img), T(rgb))


@pytest.mark.gpu
def test_color_match_properties_full_size(cu):
    """1024 x 2048 panorama, 1e6 colours (BASELINE cfg2 sizes): black pixels untouched, every output value is attained
    between the template's extremes, the map is monotone per channel, and it is idempotent in distribution: matching an
    already matched (re-quantised) image moves its levels by at most one quantisation step almost everywhere."""
    img, rgb = scene(9, 1024, 2048, 1_000_000)
    out = cu.color_match(T(img), T(rgb))
    o = out.cpu().numpy()
    black = ~ocolor.nonblack_mask(img)
    assert np.array_equal(o[black], img[black])
    nb = ~black
    for c in range(3):
        lo, hi = rgb[:, c].min(), rgb[:, c].max()
        v = o[..., c][nb]
        assert v.min() >= lo - 1e-6 and v.max() <= hi + 1e-6
        src = img[..., c][nb]
        order = np.argsort(src, kind="stable")
        assert (np.diff(v[order]) >= -1e-6).all()                           # monotone in the source level
    # sin-weighted CDF of the result tracks the template CDF
    H = img.shape[0]
    w = np.sin(np.arange(H, dtype=np.float64) / H * np.pi)[:, None].repeat(img.shape[1], 1)[nb]
    for c in range(3):
        v = o[..., c][nb]
        for q in (0.1, 0.5, 0.9):
            tq = np.quantile(rgb[:, c], q)
            frac = w[v <= tq].sum() / w.sum()
            assert abs(frac - q) <= 0.02


@pytest.mark.gpu
@pytest.mark.parametrize("case", ("b256", "b64"))
def test_color_mod_golden(cu, case):
    g = load_golden("g14_color_mod.npz")
    img, rgb = cu.color_mod(T(g[case + "_img"]), T(g[case + "_rgb"]), int(g[case + "_bins"]))
    assert np.array_equal(img.cpu().numpy(), g[case + "_out_img"])
    assert np.array_equal(rgb.cpu().numpy(), g[case + "_out_rgb"])


@pytest.mark.gpu
@pytest.mark.parametrize("bins", [256, 32, 1000])
def test_color_mod_oracle(cu, bins):
    img, rgb = scene(11, 256, 512, 300_000)
    oi, orgb = cu.color_mod(T(img), T(rgb), bins)
    ri, rrgb = ocolor.color_mod(img, rgb, bins)
    assert np.array_equal(oi.cpu().numpy(), ri)                              # byte arithmetic: exact
    assert np.array_equal(orgb.cpu().numpy(), rrgb)


@pytest.mark.gpu
def test_color_mod_properties_full_size(cu):
    """cfg2 sizes: outputs are levels k/255, black pixels untouched, the input is not modified, and the joint luma
    histogram of the result is flatter than the input's (that is what equalisation does)."""
    img, rgb = scene(12, 1024, 2048, 1_000_000)
    rgb = (rgb * 0.5 + 0.1).astype(np.float32)                               # a low-contrast cloud
    ti, tr = T(img), T(rgb)
    oi, orgb = cu.color_mod(ti, tr, 256)
    assert np.array_equal(ti.cpu().numpy(), img) and np.array_equal(tr.cpu().numpy(), rgb)
    o, r = oi.cpu().numpy(), orgb.cpu().numpy()
    lut = (np.arange(256, dtype=np.float32) / np.float32(255))
    assert np.isin(o, lut).all() and np.isin(r, lut).all()
    black = ~ocolor.nonblack_mask(img)
    assert np.array_equal(o[black], img[black])

    def luma_hist(a):
        y = ocolor.rgb2ycrcb_u8((a * np.float32(255)).astype(np.uint8))[..., 0]
        return np.bincount(y.reshape(-1), minlength=256) / y.size
    before, after = luma_hist(rgb), luma_hist(r)
    assert (after ** 2).sum() < (before ** 2).sum()                         # lower collision probability = flatter


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [[8, 8, 8], [32, 32, 32], [4, 16, 6]])
@pytest.mark.parametrize("unit_range", [True, False])
def test_histogram(cu, channels, unit_range):
    """color_utils.histogram / histogram_intersection against a direct numpy restatement (color_utils.py:86-103)."""
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(64, 96, 3)).astype(np.float32)
    if unit_range:
        img = img / np.float32(255)
    mask = rng.random((64, 96)) < 0.6

    def ref_hist(im, m):
        v = (im * np.float32(255)).astype(np.int64) if im.max() <= 1 else im.astype(np.int64)
        size = np.ceil(255.0 / np.asarray(channels, np.float32)).astype(np.int64)
        q = v[m] // size
        code = q[:, 0] + channels[0] * q[:, 1] + channels[0] * channels[1] * q[:, 2]
        h = np.bincount(code, minlength=int(np.prod(channels))).astype(np.float32)
        return h / h.sum(dtype=np.float32)
    h1 = cu.histogram(T(img), T(mask), channels)
    assert tuple(h1.shape) == tuple(channels)
    # reference layout: flat index r + c0 g + c0 c1 b, reshaped to (*channels)
    assert np.array_equal(h1.cpu().numpy().reshape(-1), ref_hist(img, mask))
    mask2 = rng.random((64, 96)) < 0.3
    h2 = cu.histogram(T(img[::-1].copy()), T(mask2), channels)
    inter = float(cu.histogram_intersection(h1, h2))
    want = float(np.minimum(h1.cpu().numpy(), h2.cpu().numpy()).astype(np.float64).sum())
    assert abs(inter - want) <= 1e-6
    # batched form: eps in the normalisation (color_utils.py:104-116)
    hb = cu.histogram(T(np.stack([img, img[::-1]])), T(np.stack([mask, mask2])), channels)
    assert tuple(hb.shape) == (2, *channels)
    assert np.abs(hb[0].cpu().numpy() - h1.cpu().numpy()).max() <= 1e-6
    ib = cu.histogram_intersection(hb, hb.flip(0))
    assert ib.shape == (2,) and abs(float(ib[0]) - inter) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("tag,channels", [("c32", [32, 32, 32]), ("c8", [8, 8, 8]), ("cu", [4, 16, 6])])
def test_histogram_reference_golden(cu, tag, channels):
    """G16: what the reference's color_utils.histogram / histogram_intersection returned (unit-range and 0..255 images,
    normalised and raw counts, unbatched and batched forms)."""
    g = load_golden("g16_histogram.npz")
    img255, mask = g["img255"], g["mask"]
    h_unit = cu.histogram(T(img255[0] / np.float32(255)), T(mask[0]), channels)
    h_255 = cu.histogram(T(img255[1]), T(mask[1]), channels)
    h_raw = cu.histogram(T(img255[1]), T(mask[1]), channels, normalize=False)
    assert np.array_equal(h_raw.cpu().numpy(), g[tag + "_raw"])                       # counts: exact
    assert np.abs(h_unit.cpu().numpy() - g[tag + "_unit"]).max() <= 1e-9            # count / total, one fp32 division
    assert np.abs(h_255.cpu().numpy() - g[tag + "_255"]).max() <= 1e-9
    hb = cu.histogram(T(img255), T(mask), channels)
    assert tuple(hb.shape) == (2, *channels) and np.abs(hb.cpu().numpy() - g[tag + "_batched"]).max() <= 1e-7
    assert abs(float(cu.histogram_intersection(h_unit, h_255)) - float(g[tag + "_inter"])) <= 1e-6
    ib = cu.histogram_intersection(hb, hb.flip(0)).cpu().numpy()
    assert np.abs(ib - g[tag + "_inter_batched"]).max() <= 1e-6


def test_ycrcb_pair_agrees_with_pillow_within_one_level_over_the_whole_cube():
    """OpenCV is absent, so cv2.cvtColor itself cannot be the pin — but Pillow's RGB <-> YCbCr (the JPEG / BT.601 full-range
    transform) is the same map up to OpenCV's rounded constants (.713 ~ .5 / (1 - .299), .564 ~ .5 / (1 - .114)) and its
    fixed-point rounding: the restatement must stay within ONE 8-bit level of that independent implementation on every one of
    the 2^24 colours, in both directions.  (A wrong coefficient, a swapped Cr / Cb or a missing +128 is tens of levels.)"""
    from PIL import Image
    v = np.arange(256, dtype=np.uint8)
    for c0 in range(0, 256, 32):                                    # 8 slabs of 32 x 256 x 256 colours
        cube = np.stack(np.meshgrid(v[c0:c0 + 32], v, v, indexing="ij"), -1).reshape(-1, 256, 3)
        mine = ocolor.rgb2ycrcb_u8(cube).astype(np.int16)           # Y, Cr, Cb
        pil = np.asarray(Image.fromarray(cube, "RGB").convert("YCbCr")).astype(np.int16)      # Y, Cb, Cr
        assert np.abs(mine[..., 0] - pil[..., 0]).max() <= 1
        assert np.abs(mine[..., 1] - pil[..., 2]).max() <= 1 and np.abs(mine[..., 2] - pil[..., 1]).max() <= 1
        back = ocolor.ycrcb2rgb_u8(cube).astype(np.int16)           # the same triples read as (Y, Cr, Cb)
        pil_back = np.asarray(Image.fromarray(np.ascontiguousarray(cube[..., [0, 2, 1]]), "YCbCr").convert("RGB")).astype(np.int16)
        assert np.abs(back - pil_back).max() <= 1
