"""Host logic of the dataset loops' tail (CPU): the per-dataset success rule and the CSV / accuracy bookkeeping of
piccolo_amd.localize.write_results — the reference's localize.py:250-297 (Stanford2D-3D-S) and :513-530 (OmniScenes)."""
import csv

import numpy as np

from piccolo_amd import localize


def _table(errs):
    t = np.zeros((len(errs), 16), np.float32)
    for k, (te, re) in enumerate(errs):
        t[k, 13], t[k, 14], t[k, 15] = te, re, 0.5
        t[k, 3:12] = np.eye(3).reshape(-1)
    return t


def test_success_rules_are_per_dataset():
    # localize.py:250  t < 0.2 m and r < rad2deg(0.2) = 11.459 deg ;  localize.py:513  t < 0.1 m and r < 5 deg
    assert localize.stanford_success(0.15, 3.0) and not localize.omniscenes_success(0.15, 3.0)          # between the two in t
    assert localize.stanford_success(0.05, 8.0) and not localize.omniscenes_success(0.05, 8.0)          # between the two in R
    assert localize.stanford_success(0.05, 3.0) and localize.omniscenes_success(0.05, 3.0)
    assert not localize.stanford_success(0.25, 3.0) and not localize.stanford_success(0.05, 11.5)
    assert not localize.omniscenes_success(0.1, 1.0) and not localize.omniscenes_success(0.01, 5.0)     # strict inequalities


def test_write_results_accuracy_per_dataset(tmp_path, capsys):
    errs = [(0.05, 3.0), (0.15, 3.0), (0.05, 8.0), (0.30, 1.0), (float("nan"), float("nan"))]
    files = ["a/f%d.png" % k for k in range(5)]
    gts = {k: (np.zeros(3, np.float32), np.eye(3, dtype=np.float32), k == 4) for k in range(5)}
    header = ["pano_name", "gt_trans", "gt_rot", "skipped?", "OmniLoc_trans", "OmniLoc_rot", "t_error (m)", "r_error (degrees)", "time (s)"]
    out = {}
    for name, rule in (("stanford", localize.stanford_success), ("omniscenes", localize.omniscenes_success)):
        out[name] = localize.write_results(_table(errs), gts, files, None, str(tmp_path / name), name + ".csv", header,
                                           lambda f: [f], rule)
        printed = capsys.readouterr().out
        assert "Final Accuracy : {}".format(out[name]["accuracy"]) in printed and "skipped 1 rooms" in printed
        with open(tmp_path / name / (name + ".csv")) as f:
            rows = list(csv.reader(f))
        assert rows[0] == header and len(rows) == 6 and rows[5][3] == "1" and len(rows[5]) == 4
        assert abs(float(rows[2][6]) - 0.15) < 1e-6 and float(rows[2][8]) == 0.5
    assert out["stanford"]["accuracy"] == 3 / 4 and out["stanford"]["failed"] == ["a/f3.png"]
    assert out["omniscenes"]["accuracy"] == 1 / 4 and out["omniscenes"]["failed"] == ["a/f1.png", "a/f2.png", "a/f3.png"]
    assert out["stanford"]["skipped"] == out["omniscenes"]["skipped"] == ["a/f4.png"]
