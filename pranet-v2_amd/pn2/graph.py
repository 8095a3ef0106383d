"""Bridge between the nn.Module surface (NCHW fp32 torch tensors, torch autograd) and the engine.

One torch.autograd.Function spans a whole module call: forward builds and runs the engine graph,
backward seeds the output gradients, replays the tape and hands parameter gradients back to autograd.
So `loss.backward()` / `optimizer.step()` of the reference's MyTrain_med.py work unchanged.
"""
import os

import torch

from .capi import F32, BF16
from .engine import Engine, PackCache, TUNER

_DT = {"bf16": BF16, "bfloat16": BF16, "fp32": F32, "float32": F32, "f32": F32}
_compute_dtype = _DT[os.environ.get("PN2_DTYPE", "bf16").lower()]


def set_compute_dtype(name):
    """'bf16' (default: bf16 storage + MFMA, fp32 accumulate) or 'fp32' (exact-fp32 MFMA, parity runs)."""
    global _compute_dtype
    _compute_dtype = _DT[name.lower()] if isinstance(name, str) else name


def get_compute_dtype():
    return _compute_dtype


def _pack_cache(dtype, params):
    """Packed-panel cache of one module call site.  The packed weight panels of a call site persist across calls (the cache hangs off the module's
    first parameter, so it lives and dies with the module) and are ALL rebuilt from the current fp32 weights by one table-driven launch
    at the start of every call - never reused unrefreshed, whatever touched the weights in between.  Conv tiles come from the
    process-wide per-shape tuner, as in pn2.trainer.Trainer."""
    pc = None
    if params and os.environ.get("PN2_MODULE_PACK_CACHE", "1") == "1":
        slot = params[0].__dict__.setdefault("_pn2_pack", {})
        pc = slot.get(dtype)
        if pc is None:
            pc = slot[dtype] = PackCache()
        if pc.keep and any(j.w != w.data_ptr() for j, w in zip(pc.jobs, pc.keep)):
            pc = slot[dtype] = PackCache()            # a weight was re-allocated (p.data = ...): start over
        pc.refresh()
    return pc


def _module_engine(dtype, training, need_grad, pc):
    tuner = TUNER if os.environ.get("PN2_AUTOTUNE", "1") == "1" else None
    return Engine(dtype, training, need_grad=need_grad, pack_cache=pc, tuner=tuner)


def _seed_grad(act, g):
    """Install d(loss)/d(output) (NCHW fp32 from autograd, or None) as the output activation's gradient."""
    N, H, W, C = act.N, act.H, act.W, act.C
    if g is None:
        buf = act.grad_buf()
        buf.zero_()
    elif act.Cp == C and act.t.dtype == torch.float32:
        gg = g.permute(0, 2, 3, 1)
        # always a private copy: another consumer of this activation accumulates into the buffer in place, and autograd's
        # incoming gradient tensors must not be modified
        act.grad = gg.clone(memory_format=torch.contiguous_format)
    else:
        buf = act.grad_buf()
        buf.zero_()
        gg = g.permute(0, 2, 3, 1).to(buf.dtype)
        if act.gw == C or act.Cp == C:
            buf[..., :C].copy_(gg)
        else:
            for c in range(C):
                buf[..., (c // act.gw) * act.gwp + c % act.gw].copy_(gg[..., c])
    act.grad_written = True


class _GraphFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, build, training, dtype, pc, n_in, *tensors):
        inputs, params = tensors[:n_in], tensors[n_in:]
        eng = _module_engine(dtype, training, True, pc)
        acts = [eng.from_nchw(x, requires_grad=x.requires_grad) for x in inputs]
        outs = build(eng, *acts)
        eng.finish_forward()
        ctx.eng, ctx.acts, ctx.outs, ctx.params, ctx.n_in = eng, acts, outs, params, n_in
        return tuple(eng.to_nchw(o) for o in outs)

    @staticmethod
    def backward(ctx, *gouts):
        eng = ctx.eng
        for o, g in zip(ctx.outs, gouts):
            _seed_grad(o, g)
        eng.backward()
        gin = []
        for a in ctx.acts:
            if a.requires_grad and a.grad is not None:
                gin.append(a.grad[..., :a.C].float().permute(0, 3, 1, 2))
            else:
                gin.append(None)
        gpar = [eng.pgrads.get(p) for p in ctx.params]
        ctx.eng = ctx.acts = ctx.outs = None
        return (None, None, None, None, None, *gin, *gpar)


def run_module(build, inputs, params, training, dtype=None):
    """Run `build(eng, *acts) -> [Act]` on NCHW inputs; returns a tuple of NCHW fp32 tensors."""
    dtype = _compute_dtype if dtype is None else dtype
    params = [p for p in params]
    need = torch.is_grad_enabled() and (any(p.requires_grad for p in params) or any(x.requires_grad for x in inputs))
    pc = _pack_cache(dtype, params)
    if not need:
        eng = _module_engine(dtype, training, False, pc)
        acts = [eng.from_nchw(x) for x in inputs]
        outs = build(eng, *acts)
        eng.finish_forward()
        return tuple(eng.to_nchw(o) for o in outs)
    return _GraphFn.apply(build, training, dtype, pc, len(inputs), *inputs, *params)
