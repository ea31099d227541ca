#!/usr/bin/env python3
"""cProfile of the HOST side of utils.make_input at the shipped shape (what runs before and between the launches).   python tools/host_profile_make_input.py"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import _knobs  # noqa: F401
from piccolo_amd import utils
sc = bench.Scene(166_667, 1024, 2048, torch.device("cuda:0"))
imgs = []
for j in range(40):
    e = sc.image(2_000_000 + j, keep_img=True); imgs.append(e.pop("img"))
for img in imgs[:4]:
    utils.make_input(img, sc.X, sc.C, 6, bench.STANFORD_INIT, "loss_histogram", 50)
torch.cuda.synchronize()
ts = []
for img in imgs[4:20]:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    utils.make_input(img, sc.X, sc.C, 6, bench.STANFORD_INIT, "loss_histogram", 50)
    ts.append((time.perf_counter() - t0) * 1e6)
print("host time of make_input (launches enqueued, not waited for): median %.0f us" % np.median(ts))
pr = cProfile.Profile()
for img in imgs[20:]:
    torch.cuda.synchronize()
    pr.enable(); utils.make_input(img, sc.X, sc.C, 6, bench.STANFORD_INIT, "loss_histogram", 50); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
