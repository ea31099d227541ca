"""Config interface: piccolo_amd.parse_utils vs what the reference's parser returned for its own shipped configs
(tests/golden/g9_parse.json, produced by tests/golden/gen_goldens.py), plus the override grammar of main.py."""
import glob
import json
import os

from conftest import GOLDEN, REPO

from piccolo_amd import parse_utils

G9 = json.load(open(os.path.join(GOLDEN, "g9_parse.json")))

# the three configs the reference ships, restated as key/value text (data, not code) so the parser can be run on them
REFERENCE_CONFIG_TEXT = {
    name: "[All]\n" + "\n".join("%s = %s" % (k, ", ".join(map(str, v)) if isinstance(v, list) else v) for k, v in d.items())
    for name, d in G9.items() if name.endswith(".ini")
}


def test_parse_ini_matches_reference_on_its_configs(tmp_path):
    for name, text in REFERENCE_CONFIG_TEXT.items():
        p = tmp_path / name
        p.write_text(text)
        cfg = parse_utils.parse_ini(str(p))
        got = cfg._asdict()
        assert got == G9[name], name
        for k, v in G9[name].items():
            assert type(got[k]) is type(v), (name, k)


def test_parse_value_matches_reference():
    for text, want in G9["parse_value"].items():
        got = parse_utils.parse_value(text)
        assert got == want and type(got) is type(want), text


def test_shipped_configs_parse_and_carry_the_hot_path_keys():
    files = sorted(glob.glob(os.path.join(REPO, "configs", "*.ini")))
    assert len(files) >= 4
    for f in files:
        cfg = parse_utils.parse_ini(f)
        for key in ("lr", "num_iter", "patience", "factor", "num_input", "out_of_room_quantile"):
            assert hasattr(cfg, key), (f, key)
        assert cfg.lr == 0.1 and cfg.num_iter == 100 and cfg.patience == 5 and cfg.factor == 0.8
    b32 = parse_utils.parse_ini(os.path.join(REPO, "configs", "stanford_mi355x_b32.ini"))
    assert b32.num_input == 32 and b32.parallel is True and b32.area is None and b32.dataset == "Stanford2D-3D-S"


def test_override_grammar(tmp_path):
    cfg = parse_utils.parse_ini(os.path.join(REPO, "configs", "stanford_mi355x_b32.ini"))
    one = parse_utils.apply_override(cfg, "num_input=256")
    assert one.num_input == 256 and one.lr == 0.1
    many = parse_utils.apply_override(cfg, "lr=0.05,area=[1,3],room_name=office_1,parallel=False,extra_key=7")
    assert many.lr == 0.05 and many.area == [1, 3] and many.room_name == "office_1" and many.parallel is False
    assert many.extra_key == 7 and many.num_input == 32
    parse_utils.save_ini(os.path.join(REPO, "configs", "synthetic.ini"), str(tmp_path))
    assert parse_utils.parse_ini(str(tmp_path / "config.ini")).dataset == "Synthetic"


def test_dropin_modules_resolve_to_the_mi355x_implementation():
    """What the reference's localize.py imports (localize.py:11-15) must exist under the drop-in names."""
    import importlib
    import sys
    sys.path.insert(0, os.path.join(REPO, "dropin"))
    try:
        for name in ("omniloc", "utils", "parse_utils", "color_utils", "data_utils"):
            sys.modules.pop(name, None)
        om = importlib.import_module("omniloc")
        ut = importlib.import_module("utils")
        pu = importlib.import_module("parse_utils")
        assert om.omniloc.__module__ == "piccolo_amd.omniloc" and om.omniloc_batch and om.sampling_loss
        for fn in ("make_input", "out_of_room", "make_pano", "write_summaries", "cloud2idx", "sample_from_img", "quantile",
                   "rot_from_ypr", "trim_input_loss", "generate_rot_points", "generate_trans_points"):
            assert callable(getattr(ut, fn)), fn
        assert pu.parse_ini and pu.parse_value and pu.save_ini
        cu = importlib.import_module("color_utils")                   # `from color_utils import color_mod, color_match`
        du = importlib.import_module("data_utils")                    # `import data_utils`
        assert cu.color_mod.__module__ == "piccolo_amd.color_utils" and cu.color_match and cu.histogram and cu.histogram_intersection
        for fn in ("read_stanford", "read_omniscenes", "obtain_gt_stanford", "obtain_gt_omniscenes"):
            assert getattr(du, fn).__module__ == "piccolo_amd.data_utils", fn
    finally:
        sys.path.remove(os.path.join(REPO, "dropin"))
        for name in ("omniloc", "utils", "parse_utils", "color_utils", "data_utils"):
            sys.modules.pop(name, None)
