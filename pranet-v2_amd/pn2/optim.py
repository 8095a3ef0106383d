"""clip_gradient + torch.optim.Adam.step of MyTrain_med.py:85-86 on the nn.Module surface, without editing the script.

The script builds `torch.optim.Adam(params, lr)` itself (MyTrain_med.py:149) and calls `clip_gradient(optimizer, clip)` from utils/utils.py - the one
function on that path the drop-in directory supplies.  Stock behaviour costs ~10 multi-tensor passes over 478 parameter tensors per step (2.4 ms of host
time, ~1 ms of GPU time at 30.5 M parameters).  Here:

  * the module surface keeps a call site's trained parameters in ONE flat fp32 arena (`flat_params`) - the layout of the flat gradient buffer its
    backward hands to autograd (pn2/graph.py: `_Site.hand_out`), so parameter i and its .grad sit at the same offset of two flat tensors;
  * `clip_flat` clamps that one gradient tensor in place (one launch; `.grad` is clamped when clip_gradient returns, as in the reference);
  * `fuse_adam`, called from clip_gradient, gives THIS optimizer instance a `step` that runs `pn2_clamp_adam` over (flat parameters, flat gradient,
    flat moments) - the Trainer's optimizer kernel, same update formula as torch's - whenever the layout still holds, and the stock step otherwise
    (closure, other hyper-parameters, gradients that are not the site's flat views, a parameter re-allocated by .to() / load ...).
    `optimizer.state[p]` holds views of the flat moments and a shared step counter, so `optimizer.state_dict()` keeps working.
PN2_FUSED_OPT=0 leaves the optimizer alone."""
import os
import weakref

import torch

from .capi import call
from .core import _p, _stream

FUSED_OPT = os.environ.get("PN2_FUSED_OPT", "1") == "1"


def _r4(n):
    return (n + 3) // 4 * 4


class FlatParams:
    """One flat fp32 arena behind a list of parameters (offsets rounded to 4 elements: the layout of _Site.gflat and of Trainer.flat)."""

    def __init__(self, params):
        self.off, o = {}, 0
        for p in params:
            self.off[id(p)] = (o, p.numel())
            o += _r4(p.numel())
        self.n = o
        p0 = params[0]
        st0 = p0.data.untyped_storage()
        if (all(p.data.untyped_storage().data_ptr() == st0.data_ptr() and p.data.storage_offset() == p0.data.storage_offset() + self.off[id(p)][0] and p.data.is_contiguous()
                for p in params) and (p0.data.storage_offset() + o) * 4 <= st0.nbytes()):
            # the parameters already live in one arena with this layout (a pn2.trainer.Trainer over the same model re-homed them): share it, do not move them
            self.flat = torch.empty(0, dtype=torch.float32, device=p0.device).set_(st0, p0.data.storage_offset(), (o,))
        else:
            self.flat = torch.zeros(o, dtype=torch.float32, device=p0.device)
        self.rehome(params)

    def rehome(self, params):
        with torch.no_grad():
            for p in params:
                o, n = self.off[id(p)]
                v = self.flat[o:o + n].view(p.shape)
                if p.data_ptr() != v.data_ptr():
                    v.copy_(p.data)
                    p.data = v

    def holds(self, p):
        e = self.off.get(id(p))
        return e is not None and p.data_ptr() == self.flat.data_ptr() + 4 * e[0]


def flat_params(params):
    """The arena of a module's trained parameters (hangs off the first parameter, like the pack cache); (re-)homes parameters that are not in it."""
    if not FUSED_OPT or not params or any(p.dtype != torch.float32 or not p.is_cuda for p in params):
        return None
    slot = params[0].__dict__
    fp = slot.get("_pn2_flat")
    if fp is None or len(fp.off) != len(params) or any(id(p) not in fp.off for p in params) or fp.flat.device != params[0].device:
        fp = slot["_pn2_flat"] = FlatParams(params)
    elif not all(fp.holds(p) for p in params):
        fp.rehome(params)
    return fp


def _grad_base(grads):
    """A 1-D fp32 alias of the ONE buffer all these gradients are contiguous slices of (the module surface's hand-out: views of a fresh flat tensor, which autograd
    detaches when it adopts them as .grad - so the test is on the storage, not on ._base), or None."""
    if not grads:
        return None
    g0 = grads[0]
    if g0.dtype != torch.float32 or not g0.is_cuda:
        return None
    st = g0.untyped_storage()
    sp = st.data_ptr()
    for g in grads:
        if g.dtype != torch.float32 or g.untyped_storage().data_ptr() != sp or not g.is_contiguous():
            return None
    return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, 0, (st.nbytes() // 4,))


def clip_flat(optimizer, grad_clip):
    """clip_gradient over gradients that are views of one flat buffer (the module surface's hand-out): ONE in-place clamp.  False: not that layout."""
    if not FUSED_OPT:
        return False
    grads = [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
    b = _grad_base(grads)
    if b is None or sum(_r4(g.numel()) for g in grads) != b.numel():          # (the views tile the buffer: nothing else lives in it)
        return False
    b.clamp_(-grad_clip, grad_clip)
    return True


class _FusedAdam:
    def __init__(self, opt):
        self.opt = weakref.ref(opt)
        self.stock = opt.step                     # the bound (hook-wrapped) torch step
        self.fp = None
        self.m = self.v = self.bc = None
        self.t = None                             # shared step counter (CPU tensor, as torch keeps it for non-capturable Adam)
        self.lr = None
        self.used = 0

    def _layout(self, opt):
        """-> (FlatParams, gradient base) when every parameter that has a gradient sits at the same offset of the flat arena and of one flat gradient buffer."""
        if len(opt.param_groups) != 1 or opt._optimizer_step_pre_hooks or opt._optimizer_step_post_hooks:
            return None          # (step hooks are run by the stock step's wrapper: leave such an optimizer alone)
        g = opt.param_groups[0]
        if (g.get("amsgrad") or g.get("maximize") or g.get("weight_decay", 0) != 0 or g.get("capturable") or g.get("differentiable") or g.get("fused")
                or not isinstance(g["lr"], float)):
            return None
        ps = [p for p in g["params"] if p.grad is not None]
        if not ps:
            return None
        fp = ps[0].__dict__.get("_pn2_flat")
        if fp is None:
            for p in g["params"]:
                fp = p.__dict__.get("_pn2_flat")
                if fp is not None:
                    break
        if fp is None or len(ps) != len(fp.off):
            return None
        b = _grad_base([p.grad for p in ps])
        if b is None or b.numel() != fp.n or b.device != fp.flat.device:
            return None
        for p in ps:
            e = fp.off.get(id(p))
            if e is None or p.grad.storage_offset() != e[0] or p.data_ptr() != fp.flat.data_ptr() + 4 * e[0]:
                return None
        return fp, b, g, ps

    def _adopt(self, opt, fp, ps, g):
        """First fused step on this arena: flat moments (taking over what stock steps have accumulated), the device-side hyper-parameter block."""
        dev = fp.flat.device
        self.fp = fp
        self.m, self.v = torch.zeros_like(fp.flat), torch.zeros_like(fp.flat)
        steps = set()
        for p in ps:
            s = opt.state.get(p)
            if s:
                o, n = fp.off[id(p)]
                self.m[o:o + n].copy_(s["exp_avg"].reshape(-1))
                self.v[o:o + n].copy_(s["exp_avg_sq"].reshape(-1))
                steps.add(int(s["step"]))
            else:
                steps.add(0)
        if len(steps) != 1:
            return False
        k = steps.pop()
        b1, b2 = g["betas"]
        self.t = torch.tensor(float(k))
        self.lr = g["lr"]
        self.bc = torch.tensor([1.0 - b1 ** k, 1.0 - b2 ** k, b1 ** k, b2 ** k, self.lr, 3.0e38, 0.0, 1.0], dtype=torch.float32, device=dev)
        for p in ps:
            o, n = fp.off[id(p)]
            opt.state[p] = {"step": self.t, "exp_avg": self.m[o:o + n].view(p.shape), "exp_avg_sq": self.v[o:o + n].view(p.shape)}
        return True

    def step(self, closure=None):
        opt = self.opt()
        lay = self._layout(opt) if (closure is None and FUSED_OPT) else None
        if lay is not None and self.fp is not lay[0]:
            if self.fp is not None or not self._adopt(opt, lay[0], lay[3], lay[2]):
                lay = None
        if lay is None:
            if self.fp is not None:          # back to the stock step for good: its state entries are views of our moments, which stay valid tensors
                for s in opt.state.values():
                    if s.get("step") is self.t:
                        s["step"] = self.t.clone()          # (the stock foreach step increments every entry: they must not share one tensor)
                self.fp = None
                opt.step = self.stock
                opt._pn2_fused = None
            return self.stock(closure) if closure is not None else self.stock()
        fp, b, g, ps = lay
        if g["lr"] != self.lr:               # utils.adjust_lr (MyTrain_med.py:155): the kernel reads lr from the device block
            self.lr = g["lr"]
            self.bc[4] = self.lr
        b1, b2 = g["betas"]
        st = _stream()
        call.pn2_adam_tick(_p(self.bc), b1, b2, st)
        call.pn2_clamp_adam(_p(fp.flat), _p(b), _p(self.m), _p(self.v), fp.n, self.lr, b1, b2, g["eps"], 3.0e38, 1.0, _p(self.bc), 0.0, st)
        self.t += 1
        self.used += 1
        return None


def fuse_adam(optimizer):
    """Called by utils.clip_gradient: give a plain torch.optim.Adam instance the fused step (once).  Anything else is left alone."""
    if (not FUSED_OPT or type(optimizer) is not torch.optim.Adam or getattr(optimizer, "_pn2_fused", None) is not None
            or getattr(optimizer.step, "_wrapped_by_lr_sched", False)):          # (a torch lr_scheduler counts calls of the step it wrapped)
        return
    f = _FusedAdam(optimizer)
    if f._layout(optimizer) is None:
        return
    optimizer._pn2_fused = f
    optimizer.step = f.step
