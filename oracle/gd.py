"""CPU restatement of the reference's gradient-descent refinement (TEST INFRASTRUCTURE).

omniloc (sequential, omniloc.py:11-102) and omniloc_batch (parallel, omniloc.py:205-296) with
torch.optim.Adam and ReduceLROnPlateau restated as scalar numpy code.  Adam / the scheduler are
third-party (torch, pinned torch==1.7.0 in requirements.txt:1; the semantics used here are those of
the single-tensor, non-capturable Adam and of ReduceLROnPlateau(mode='min', threshold=1e-4 'rel',
cooldown=0, min_lr=0, eps=1e-8), unchanged through 2.10) — pinned by the per-iteration trajectories
in tests/golden/g5_trajectories.npz.

`loss_grad(trans (B,3), rot (B,3)) -> (loss (B,), grad_t (B,3), grad_ypr (B,3))` is injectable so the
tests can teacher-force recorded reference gradients or plug the HIP path.
"""
import math

import numpy as np

from . import oracle as orc

F32 = np.float32


class Adam:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) on one float32 vector; call sites omniloc.py:33,235."""

    def __init__(self, n, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = float(lr), beta1, beta2, eps
        self.m = np.zeros(n, F32)
        self.v = np.zeros(n, F32)
        self.t = 0

    def step(self, param, grad):
        g = grad.astype(F32)
        self.t += 1
        w = F32(1 - self.b1)
        self.m = self.m + w * (g - self.m)                                  # exp_avg.lerp_(grad, 1 - beta1)
        self.v = self.v * F32(self.b2) + F32(1 - self.b2) * g * g           # mul_(beta2).addcmul_(g, g, 1 - beta2)
        bc1 = 1 - self.b1 ** self.t                                         # python doubles
        bc2 = 1 - self.b2 ** self.t
        step_size = self.lr / bc1
        denom = np.sqrt(self.v) / F32(math.sqrt(bc2)) + F32(self.eps)
        return (param + F32(-step_size) * self.m / denom).astype(F32)       # addcdiv_(m, denom, value=-step_size)


class Plateau:
    """ReduceLROnPlateau(opt, mode='min', patience, factor); call sites omniloc.py:37,50,237,258."""

    def __init__(self, opt, patience, factor, threshold=1e-4, eps=1e-8):
        self.opt, self.patience, self.factor, self.threshold, self.eps = opt, patience, factor, threshold, eps
        self.best = math.inf
        self.num_bad = 0

    def step(self, metric):
        cur = float(metric)
        if cur < self.best * (1.0 - self.threshold):
            self.best, self.num_bad = cur, 0
        else:
            self.num_bad += 1
        if self.num_bad > self.patience:
            new_lr = max(self.opt.lr * self.factor, 0.0)
            if self.opt.lr - new_lr > self.eps:
                self.opt.lr = new_lr
            self.num_bad = 0


def _cfg(cfg, k, d):
    return getattr(cfg, k, d)


def make_loss_grad(xyz, rgb, img, dtype=np.float32, nthreads=0):
    def fn(trans, rot):
        o = orc.sampling_loss(xyz, rgb, img, trans, rot, dtype=dtype, grad=True, nthreads=nthreads)
        return o["loss"].astype(F32), o["grad_t"].astype(F32), o["grad_ypr"].astype(F32)
    return fn


def omniloc(img, xyz, rgb, input_trans, input_rot, starting_point, cfg, loss_grad=None, trace=None):
    """Sequential GD for one start — omniloc.py:11-102.  Returns [t (3,1), R (3,3), loss ()] float32.

    The returned loss is the one of the LAST forward, i.e. at the pose before the final update (:46,:102).
    The reference's parameters are views of input_trans/input_rot rows (:15-19), so those rows end up
    holding the final pose; this restatement writes them back the same way.
    """
    lr, num_iter = _cfg(cfg, "lr", 0.1), _cfg(cfg, "num_iter", 100)
    patience, factor = _cfg(cfg, "patience", 5), _cfg(cfg, "factor", 0.9)
    q = _cfg(cfg, "out_of_room_quantile", 0.05)
    loss_grad = loss_grad or make_loss_grad(xyz, rgb, img)
    box = orc.quantile_box(xyz, q)                                  # recomputed every iteration at :53-55; invariant
    # Adam parameter order is [translation(3), yaw, roll, pitch] (:33); per-element Adam, order irrelevant
    p = np.concatenate([input_trans[starting_point], input_rot[starting_point]]).astype(F32)   # t, yaw, pitch, roll
    opt = Adam(6, lr)
    sched = Plateau(opt, patience, factor)
    loss = F32(0)
    for it in range(num_iter):
        l, gt, gr = loss_grad(p[None, :3], p[None, 3:])
        loss = l[0]
        g = np.concatenate([gt[0], gr[0]])
        if trace is not None:
            trace.append(dict(param=p.copy(), loss=loss, grad=g.copy(), lr=opt.lr))
        p = opt.step(p, g)
        sched.step(loss)
        p[:3] = np.minimum(np.maximum(p[:3], box[:, 0]), box[:, 1])  # clamp in place on the leaf (:56-58)
        if trace is not None:
            trace[-1].update(param_after=p.copy(), lr_after=opt.lr, num_bad=sched.num_bad, best=sched.best)
    input_trans[starting_point] = p[:3]
    input_rot[starting_point] = p[3:]
    R = orc.rot_from_ypr(p[3:], np.float32)
    return [p[:3].reshape(3, 1).copy(), R, F32(loss)]


def omniloc_batch(img, xyz, rgb, input_trans, input_rot, cfg, loss_grad=None, trace=None):
    """Parallel GD over all starts — omniloc.py:205-296.  Returns [t (3,1), R (3,3), loss ()] of the winner.

    Quirk kept: parameters are re-concatenated (copied) at :260-263 BEFORE the in-place
    clamp of the leaves at :265-269, so the next forward sees the unclamped translation while Adam keeps
    updating the clamped leaf; the returned translation is that unclamped copy (:272).  Winner = argmin of the
    last forward's loss_list (:271).
    """
    lr, num_iter = _cfg(cfg, "lr", 0.1), _cfg(cfg, "num_iter", 100)
    patience, factor = _cfg(cfg, "patience", 5), _cfg(cfg, "factor", 0.9)
    q = _cfg(cfg, "out_of_room_quantile", 0.05)
    loss_grad = loss_grad or make_loss_grad(xyz, rgb, img)
    B = input_trans.shape[0]
    box = orc.quantile_box(xyz, q)                                  # once, :244-247
    leaf = np.concatenate([input_trans, input_rot], 1).astype(F32)  # (B,6): t, yaw, pitch, roll
    fwd = leaf.copy()                                               # the torch.cat copies (:239-242)
    opts = [Adam(6, lr) for _ in range(B)]
    scheds = [Plateau(o, patience, factor) for o in opts]
    loss_list = np.zeros(B, F32)
    for it in range(num_iter):
        loss_list, gt, gr = loss_grad(fwd[:, :3], fwd[:, 3:])
        g = np.concatenate([gt, gr], 1)
        if trace is not None:
            trace.append(dict(fwd=fwd.copy(), leaf=leaf.copy(), loss=loss_list.copy(), grad=g.copy(),
                              lr=np.array([o.lr for o in opts])))
        for b in range(B):
            leaf[b] = opts[b].step(leaf[b], g[b])
            scheds[b].step(loss_list[b])
        fwd = leaf.copy()                                           # re-cat, :260-263
        leaf[:, :3] = np.minimum(np.maximum(leaf[:, :3], box[None, :, 0]), box[None, :, 1])   # :265-269
        if trace is not None:
            trace[-1].update(leaf_after=leaf.copy(), lr_after=np.array([o.lr for o in opts]),
                             num_bad=np.array([s.num_bad for s in scheds]), best=np.array([s.best for s in scheds]))
    input_trans[:] = leaf[:, :3]
    input_rot[:] = leaf[:, 3:]
    k = int(np.argmin(loss_list))                                   # NaN-free case; torch argmin :271
    R = orc.rot_from_ypr(fwd[k, 3:], np.float32)
    return [fwd[k, :3].reshape(3, 1).copy(), R, F32(loss_list[k])]


def trim_input_loss(img, xyz, rgb, trans, rot, num_input, dtype=np.float32, nthreads=0):
    """utils.py:462-507: loss table over all (trans[i], rot[j]), keep the num_input smallest.

    Returns (trimmed_trans, trimmed_rot, loss_table)."""
    K, Rn = len(trans), len(rot)
    tt = np.repeat(np.asarray(trans, F32), Rn, 0)
    rr = np.tile(np.asarray(rot, F32), (K, 1))
    o = orc.sampling_loss(xyz, rgb, img, tt, rr, dtype=dtype, grad=False, nthreads=nthreads)
    table = o["loss"].astype(F32).reshape(K, Rn)
    num_input = min(num_input, K * Rn)
    inds = np.argsort(table.reshape(-1), kind="stable")[:num_input]
    return np.asarray(trans)[inds // Rn], np.asarray(rot)[inds % Rn], table
