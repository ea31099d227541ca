#!/usr/bin/env python3
"""Where the host time of one make_input + omniloc_batch call goes (shipped shape by default): per call, the time until the Python call
RETURNS (launches enqueued) against the time until the device is idle, plus a cProfile of the host side.
   python tools/host_overhead.py [n_points num_input num_intermediate]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import _knobs  # noqa: F401,E402
from piccolo_amd import omniloc as po, utils  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 166_667
NUM_INPUT = int(sys.argv[2]) if len(sys.argv) > 2 else 6
NUM_INTER = int(sys.argv[3]) if len(sys.argv) > 3 else 50
sc = bench.Scene(N, 1024, 2048, torch.device("cuda:0"))


class Cfg:
    lr, num_iter, patience, factor, out_of_room_quantile, num_input = 0.1, 100, 5, 0.8, 0.05, NUM_INPUT


imgs = []
for j in range(12):
    e = sc.image(2_000_000 + j, keep_img=True)
    imgs.append(e.pop("img"))
rows = []
for j, img in enumerate(imgs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr, ro = utils.make_input(img, sc.X, sc.C, NUM_INPUT, bench.STANFORD_INIT, "loss_histogram", NUM_INTER)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    res = po.omniloc_batch(img, sc.X, sc.C, tr, ro, Cfg(), {})
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    if j >= 2:
        rows.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t2) * 1e3))
r = np.median(np.array(rows), axis=0)
print("make_input: host returns after %.3f ms, device idle after %.3f ms | omniloc_batch: returns (incl. its D2H copy) %.3f ms, idle %.3f ms" % tuple(r))
pr = cProfile.Profile()
for img in imgs[2:]:
    torch.cuda.synchronize()
    pr.enable()
    tr, ro = utils.make_input(img, sc.X, sc.C, NUM_INPUT, bench.STANFORD_INIT, "loss_histogram", NUM_INTER)
    res = po.omniloc_batch(img, sc.X, sc.C, tr, ro, Cfg(), {})
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
