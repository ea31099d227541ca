#!/usr/bin/env python3
"""Text-cloud parser: wall time of piccolo_amd.data_utils.read_stanford on a synthetic "x y z r g b" file, beside
pandas.read_table (what the reference calls, data_utils.py:30).   python tools/io_bench.py [n_points]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import data_utils  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(0)
xyz = rng.normal(0, 8, (n, 3))
rgb = rng.integers(0, 256, (n, 3))
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "cloud.txt")
    t0 = time.perf_counter()
    np.savetxt(path, np.hstack([xyz, rgb]), fmt="%.3f %.3f %.3f %d %d %d")
    size = os.path.getsize(path)
    print("wrote %d points, %.1f MB in %.1f s" % (n, size / 1e6, time.perf_counter() - t0))
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        a, b = data_utils.read_stanford(path)
        best = min(best, time.perf_counter() - t0)
    print("native parser (%d threads): %.3f s  = %.0f MB/s, %.1f M points/s" % (os.cpu_count(), best, size / best / 1e6, n / best / 1e6))
    try:
        import warnings
        from pandas import read_table
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            ref = read_table(path, header=None, delim_whitespace=True).values
            dt = time.perf_counter() - t0
        print("pandas.read_table (reference): %.3f s = %.0f MB/s; same bits: %s" % (
            dt, size / dt / 1e6, np.array_equal(ref[:, :3], a) and np.array_equal(ref[:, 3:] / 255., b)))
    except ImportError:
        print("pandas not available")
