import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench
from piccolo_amd import ops, utils, omniloc as po
N = int(sys.argv[1]) if len(sys.argv) > 1 else 166_667
NI, NM = (6, 50) if N < 500_000 else (32, 64)
sc = bench.Scene(N, 1024, 2048, torch.device("cuda:0"))
imgs = []
for j in range(10):
    e = sc.image(2_000_000 + j, keep_img=True); imgs.append(e.pop("img"))
pays = ops.trim_order_pays
for mode in ("order", "plain", "order", "plain"):
    ops.trim_order_pays = pays if mode == "order" else (lambda *a: False)
    po._cache.kinds.pop("trimorder", None)
    ts = []
    outs = []
    for j, img in enumerate(imgs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr, ro = utils.make_input(img, sc.X, sc.C, NI, bench.STANFORD_INIT, "loss_histogram", NM)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        outs.append((tr.clone(), ro.clone()))
    print(N, mode, "make_input ms: median %.3f (first %.3f)" % (float(np.median(ts[2:])), ts[0]), flush=True)
    if mode == "order": ref = outs
    else: print("  same starting poses as with the list:", all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(ref, outs)))
