"""Distribution of the device's G23 loss-table differences from the reference's (experiment aid)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import _knobs  # noqa: F401  (PCL_* experiment variables -> experiments build / ops.EXPERIMENT; tools/_knobs.py)
from piccolo_amd import ops
from test_oracle_golden import G23_CONFIGS, g23_case
g = np.load(os.path.join(REPO, "tests/golden/g23_make_input.npz"))
dev = torch.device("cuda")
for tag in G23_CONFIGS:
    xyz, rgb, img, init, n_in, n_mid, d = g23_case(g, tag)
    X, C, I = [torch.from_numpy(a).to(dev) for a in (xyz, rgb, img)]
    tr, ro = torch.from_numpy(d["loss_trans"]).to(dev), torch.from_numpy(d["loss_rot"]).to(dev)
    for fmt in ("u8", "f32"):
        table, cnt = ops.trim_loss_table(ops.Cloud(X, C), ops.Pano(I, fmt=fmt), tr, ops.TrimGroups(ro), return_count=True)
        K, Rn = table.shape
        gen = ops.sampling_loss(ops.Cloud(X, C), ops.Pano(I, fmt=fmt), tr.repeat_interleave(Rn, 0), ro.repeat(K, 1), with_grad=False).cpu().numpy()
        e = np.abs(table.cpu().numpy() - d["loss_loss_table"]).reshape(-1)
        eg = np.abs(gen[:, 0] - d["loss_loss_table"].reshape(-1))
        n = len(xyz)
        print(tag, fmt, "n", n, "yaw-shared: max %.2e  >4e-7: %d  >1e-6: %d  max*n %.2f | generic: max %.2e >4e-7: %d max*n %.2f | count diff shared-vs-generic max %d" % (
            e.max(), (e > 4e-7).sum(), (e > 1e-6).sum(), e.max() * n, eg.max(), (eg > 4e-7).sum(), eg.max() * n, np.abs(cnt.cpu().numpy().reshape(-1) - gen[:, 1]).max()))
        bad = np.argsort(e)[-12:]
        print("   worst:", ["%.1e@r%d" % (e[b], b % Rn) for b in bad])
