#!/usr/bin/env python3
"""8 query images x 6 candidates at the shipped shape: ONE launch chain of 48 poses (two launches per iteration) vs 2 / 4 / 8 independent
chains (24 / 12 / 6 poses; 6 poses = the fused one-launch iteration) replayed as hipGraphs on as many HIP streams — do the chains' launch
gaps and ramps overlap?   python tools/multi_stream_chains.py [n_points]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 166_667
H, W, B, I = 1024, 2048, 6, 8
dev = torch.device("cuda:0")
xyz, rgb = synth.box_room(N, 0)
X, C = torch.from_numpy(xyz).to(dev), torch.from_numpy(rgb).to(dev)
cloud, box = ops.Cloud(X, C), ops.quantile_box(X, 0.05)
panos, TR, RO = [], [], []
for j in range(I):
    t, ypr = synth.gt_pose(100 + j)
    img = synth.quantise_like_image_file(ops.make_pano(ops.transform_cloud(X, torch.from_numpy(t), torch.from_numpy(ypr)), C, (H, W)))
    panos.append(ops.Pano(img, fmt=ops.refine_texels(N, H, W)))
    tr, ro = synth.start_poses(t, ypr, B, seed=j)
    TR.append(torch.from_numpy(tr).to(dev)); RO.append(torch.from_numpy(ro).to(dev))
kw = dict(lr=0.1, patience=5, factor=0.8, batch_mode=True)
ref = None
for nchains in (1, 2, 4, 8):
    per = I // nchains
    engines, streams = [], []
    for c in range(nchains):
        tr = torch.cat(TR[c * per:(c + 1) * per]).contiguous(); ro = torch.cat(RO[c * per:(c + 1) * per]).contiguous()
        gd = ops.GradientDescent(cloud, panos[c * per], tr, ro, box, **kw)
        gd.set_pano_groups(panos[c * per:(c + 1) * per])
        engines.append((gd, tr, ro)); streams.append(torch.cuda.Stream(device=dev))
    for use_graph in (False, True):
        ts = []
        for rep in range(6):
            for c, (gd, tr, ro) in enumerate(engines):
                gd.reset(tr, ro); gd.set_pano_groups(panos[c * per:(c + 1) * per])
            torch.cuda.synchronize()
            cur = torch.cuda.current_stream()
            t0 = time.perf_counter()
            for c, (gd, tr, ro) in enumerate(engines):
                st = streams[c] if nchains > 1 else cur
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    gd.run_graph(100) if use_graph else gd.run(100)
            for st in streams:
                cur.wait_stream(st)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res = torch.cat([e[0].result() for e in engines]).cpu().numpy()
        if ref is None:
            ref = res
        print("%d chain(s) x %2d poses, %s: %.3f ms per 8 images x 100 iterations = %.1f us per iteration of all 48 | results identical to one chain: %s"
              % (nchains, per * B, "graph replay" if use_graph else "eager       ", float(np.median(ts[1:])), float(np.median(ts[1:])) * 10, bool(np.array_equal(res, ref))), flush=True)
