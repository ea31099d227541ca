"""Shared helpers of the GPU parity tests (not a test module)."""
import numpy as np
import torch


def rel(a, b):
    """max |a-b| / max |b|; NaNs must sit at the same places (0/0 losses of fully masked poses) and are then ignored."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b)), (a, b)
    ok = ~np.isnan(b)
    if not ok.any():
        return 0.0
    return np.abs(a[ok] - b[ok]).max() / max(np.abs(b[ok]).max(), 1e-30)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _oracle_pair(oracle, xyz, rgb, img, trans, rot, **kw):
    """The exact answer (fp64 oracle) and the yardstick: the SAME formulas evaluated in plain fp32 with libm (fp32 oracle),
    i.e. what the reference's fp32 tensors deliver.  Its distance from fp64 is what an fp32 evaluation of this scene can
    achieve: tiny at small N (G3: 3e-6), but growing with N and resolution, because the gradient of a piecewise-bilinear
    image is discontinuous at texel boundaries — a point whose pixel coordinate (up to 2048, ulp 1.2e-4 px) rounds into the
    neighbouring cell changes its own gradient term by O(1), and the terms of a near-converged pose cancel almost
    completely (measured at 1M points, 2048x1024: fp32 oracle 2.7e-3 / 1.1e-3 from fp64, HIP 1.6e-3 / 1.0e-3)."""
    r64 = oracle.sampling_loss(xyz, rgb, img, trans, rot, dtype=np.float64, **kw)
    r32 = oracle.sampling_loss(xyz, rgb, img, trans, rot, dtype=np.float32, **kw)
    return r64, r32


def _check_vs_oracle(parity, out, r64, r32, n, tag=""):
    """loss / grad_t / grad_ypr of `out` (B,8) against the fp64 oracle.  Bounds: loss 3e-7 + mask flips (measured
    4e-8..2.4e-7); gradients 2 x the fp32 oracle's own gap + 5e-6 + mask flips (measured: 0.9..1.3 x that gap)."""
    dcount = float(np.abs(out[:, 1] - r64["count"]).max())
    parity(tag + "count (points)", dcount, max(2, 2e-5 * n))
    # a point that flips between masked and kept moves the mean loss by ~1/n of its scale, the gradient by more
    parity(tag + "loss vs fp64", rel(out[:, 0], r64["loss"]), 3e-7 + 2.0 * dcount / n, rel(r32["loss"], r64["loss"]))
    gap_t, gap_r = rel(r32["grad_t"], r64["grad_t"]), rel(r32["grad_ypr"], r64["grad_ypr"])
    parity(tag + "grad_t vs fp64", rel(out[:, 2:5], r64["grad_t"]), 2 * gap_t + 5e-6 + 20.0 * dcount / n, gap_t)
    parity(tag + "grad_ypr vs fp64", rel(out[:, 5:8], r64["grad_ypr"]), 2 * gap_r + 5e-6 + 20.0 * dcount / n, gap_r)


def check_selection(got_idx, ref_values, n, tol, largest=False):
    """A top-n selection against the reference's values, as far as those values can decide it.  Every value is known to +-tol
    (a scalar, or one tolerance per entry: e.g. 10 x the table tolerance, plus the measured difference where a mask flip moved
    an entry).  An entry MUST be selected when fewer than n others can possibly rank ahead of it, MAY be selected only when
    fewer than n others certainly rank ahead of it, and where its rank is certain (as many certainly ahead as possibly ahead)
    it must hold exactly that RANK in the selection.  -> (number of must-haves, number of rank-checked positions)."""
    v = np.asarray(ref_values, np.float64).reshape(-1)
    v = -v if largest else v
    tol = np.broadcast_to(np.asarray(tol, np.float64).reshape(-1), v.shape)
    lo, hi = v - tol, v + tol
    got = [int(i) for i in np.asarray(got_idx).reshape(-1)]
    assert len(got) == n == len(set(got)), (len(got), n)
    possibly_ahead = (lo[None, :] < hi[:, None]).sum(1) - 1          # j != i with lo_j < hi_i (i itself always counts once)
    certainly_ahead = (hi[None, :] < lo[:, None]).sum(1)
    must = {int(i) for i in np.nonzero(possibly_ahead < n)[0]}
    may = {int(i) for i in np.nonzero(certainly_ahead < n)[0]}
    assert must <= set(got), ("selection misses", sorted(must - set(got)))
    assert set(got) <= may, ("selection includes", sorted(set(got) - may))
    ranked = 0
    for i in must:
        if possibly_ahead[i] == certainly_ahead[i]:                  # the reference's rank of i is unambiguous at this tolerance
            assert got[int(certainly_ahead[i])] == i, ("rank", int(certainly_ahead[i]), got[int(certainly_ahead[i])], i)
            ranked += 1
    return len(must), ranked


def match_rows(rows, table, atol=1e-6):
    """index in `table` of every row of `rows` (each must match exactly one)"""
    rows, table = np.asarray(rows, np.float64), np.asarray(table, np.float64)
    d = np.abs(rows[:, None, :] - table[None, :, :]).max(-1)
    idx = d.argmin(1)
    assert (d[np.arange(len(rows)), idx] <= atol).all() and ((d <= atol).sum(1) == 1).all()
    return idx
