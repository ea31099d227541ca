"""Host-side selection rules (no GPU): the texel layout / format a launch takes for a cloud and panorama size.  The rules encode
measurements (docstrings of ops.trim_texels / ops.refine_texels); these are their anchor points, so that an edit that moves one of the
shapes the bench line and the profiles were taken at shows up here."""
from piccolo_amd import ops


def test_refinement_format_by_density():
    assert ops.refine_texels(1_000_000, 1024, 2048) == "f16"        # cfg 2 / 3 / 4: what bench.py's headline runs on
    assert ops.refine_texels(10_000_000, 2048, 4096) == "f16"       # cfg 5
    assert ops.refine_texels(100_000, 256, 512) == "f16"            # cfg 1
    assert ops.refine_texels(166_667, 1024, 2048) == "u8"           # the reference's shipped shape
    assert ops.refine_texels(700_000, 1024, 2048) == "u8" and ops.refine_texels(3_000_000, 2048, 4096) == "u8"
    assert ops.refine_texels(4_500_000, 2048, 4096) == "f16"


def test_trim_layout_by_density_and_texture_size():
    assert [ops.trim_texels(n, 1024, 2048) for n in (166_667, 400_000, 700_000, 850_000, 1_000_000, 2_000_000)] == \
        ["u8p", "u8p", "u8", "u8", "u8v", "u8v"]
    assert [ops.trim_texels(n, 2048, 4096) for n in (3_000_000, 4_000_000, 6_000_000, 10_000_000)] == ["u8p", "u8", "u8v", "u8v"]
    assert [ops.trim_texels(n, 512, 1024) for n in (100_000, 500_000)] == ["u8v", "u8v"]
    assert ops.trim_texels(100_000, 256, 512) == "u8v"
