"""Bridge between the nn.Module surface (NCHW fp32 torch tensors, torch autograd) and the engine.

One torch.autograd.Function spans a whole module call: forward builds and runs the engine graph,
backward seeds the output gradients, replays the tape and hands parameter gradients back to autograd.
So `loss.backward()` / `optimizer.step()` of the reference's MyTrain_med.py work unchanged.
"""
import os

import torch

import weakref

from .capi import F32, BF16
from .engine import Engine, GradQueue, PackCache, StepArena, TUNER

_DT = {"bf16": BF16, "bfloat16": BF16, "fp32": F32, "float32": F32, "f32": F32}
_compute_dtype = _DT[os.environ.get("PN2_DTYPE", "bf16").lower()]


def set_compute_dtype(name):
    """'bf16' (default: bf16 storage + MFMA, fp32 accumulate), 'fp32' (fp32 storage, conv contractions in DOUBLE on the f64 matrix pipe: the parity path) or
    'fp32fast' (fp32 storage, fp32 products and sums on the f32 matrix pipe - the reference's own arithmetic, MyTrain_med.py runs without autocast - at twice
    the pipe rate; everything except the conv GEMM / wgrad kernels is the 'fp32' path)."""
    global _compute_dtype
    from . import capi
    fast = isinstance(name, str) and name.lower() in ("fp32fast", "f32fast", "fp32_fast")
    capi.set_f32_mma(fast)
    _compute_dtype = F32 if fast else (_DT[name.lower()] if isinstance(name, str) else name)


def get_compute_mode():
    """'bf16' | 'fp32' | 'fp32fast'"""
    from . import capi
    return "bf16" if _compute_dtype == BF16 else ("fp32fast" if capi.F32_MMA == capi.F32F else "fp32")


def get_compute_dtype():
    return _compute_dtype


def _pack_cache(dtype, params, refresh=True):
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
        if refresh:
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


# ------------------------------------------------------------------------------------------------------------------------------------------
# Training call sites replayed from hipGraphs.  `model(images) ... loss.backward()` of MyTrain_med.py:59-86 launches ~1 900 kernels from Python
# per step (~57 ms of host time against ~15 ms of GPU work at bs = 32).  A call site - one module called in train mode with one input shape -
# therefore goes through three stages: its first PLAIN_CALLS calls are the plain eager pass above; the next two run the same pass on the
# trainer's machinery (step arena, lock-step tables, deferred weight-gradient tables: pn2/trainer.py) so that every buffer and job table has
# its final address; from then on the forward is ONE hipGraph and the backward another (sharing a memory pool), and a call costs two
# hipGraphLaunch plus the copies of the inputs, outputs and output gradients.  Calls the graphs cannot serve - an input that requires grad, a
# second forward before the backward of the previous one, eval mode - fall back to the plain pass, so the semantics of the surface do not change.
# ------------------------------------------------------------------------------------------------------------------------------------------
MODULE_GRAPH = os.environ.get("PN2_MODULE_GRAPH", "1") == "1"
PLAIN_CALLS = 2
MAX_SITES = 4            # input shapes kept per module (the 0.75x / 1x / 1.25x schedule of MyTrain_med.py:55 needs 3)


def set_module_graph(on):
    global MODULE_GRAPH
    MODULE_GRAPH = bool(on)


class _Token:
    pass


class _Sites(dict):
    """Hangs off the module's first parameter next to the pack cache: never pickled / deep-copied with it."""

    def __reduce__(self):
        return (_Sites, ())

    def __deepcopy__(self, memo):
        return _Sites()


class _Site:
    def __init__(self, params, pc):
        self.pc, self.calls, self.arena_steps, self.gen = pc, 0, 0, 0
        self.arena, self.lock, self.rq = StepArena(), {}, GradQueue(defer_wgrad=True)
        self.off, o = {}, 0
        for p in params:
            self.off[id(p)] = (o, p.numel())
            o += (p.numel() + 3) // 4 * 4
        self.gflat = torch.zeros(o, dtype=torch.float32, device=params[0].device)     # parameter gradients at fixed addresses (the reduce tables hold them)
        self.graph_f = self.graph_b = None
        self.no_graph = False
        self.live, self.bwd_done = None, True

    def busy(self):
        """A forward of this site is waiting for its backward: its saved activations live in the arena the next pass would overwrite."""
        return not self.bwd_done and self.live is not None and self.live() is not None

    def grad_view(self, p):
        if id(p) not in self.off:
            return torch.empty_like(p, dtype=torch.float32)
        o, n = self.off[id(p)]
        return self.gflat[o:o + n].view(p.shape)

    def hand_out(self, grads):
        """The parameter gradients of a pass as autograd gets them: views of ONE fresh copy of the flat buffer (one launch instead of one clone per parameter;
        nothing else references the views, so autograd adopts them as .grad where a parameter has none).  Gradients that do not live in the flat buffer are cloned."""
        flat = self.gflat.clone()
        out = []
        for g in grads:
            if g is None:
                out.append(None)
            elif g.untyped_storage().data_ptr() == self.gflat.untyped_storage().data_ptr():
                o = g.storage_offset()
                out.append(flat[o:o + g.numel()].view(g.shape))
            else:
                out.append(g.clone())
        return out

    def begin(self, ctx):
        self.gen += 1
        ctx.site, ctx.gen, ctx.token = self, self.gen, _Token()
        self.live, self.bwd_done = weakref.ref(ctx.token), False

    def check(self, ctx):
        if ctx.gen != self.gen:
            raise RuntimeError("this call site ran again before the backward of an earlier call (its activations are gone); set PN2_MODULE_GRAPH=0")

    def forward(self, build, training, dtype, inputs):
        self.arena.begin_step(inputs[0].device)
        eng = Engine(dtype, training, grad_provider=self.grad_view, need_grad=True, pack_cache=self.pc, grad_queue=self.rq, arena=self.arena,
                     tuner=TUNER if os.environ.get("PN2_AUTOTUNE", "1") == "1" else None,
                     lock_cache=self.lock if self.arena.buf is not None else None)
        acts = [eng.from_nchw(x, requires_grad=False) for x in inputs]
        outs = build(eng, *acts)
        eng.finish_forward()
        return eng, outs, tuple(eng.to_nchw(o) for o in outs)

    def backward(self, eng, outs, gouts):
        for o, g in zip(outs, gouts):
            _seed_grad(o, g)
        self.rq.begin_step()
        eng.backward()
        self.rq.flush()

    def capture(self, build, training, dtype, inputs, params):
        import gc
        from .trainer import CAPTURE_MODE
        self.s_in = [x.detach().clone() for x in inputs]
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        gc_was = gc.isenabled()
        gc.collect()
        gc.disable()               # no collector inside a capture (see Trainer.capture)
        try:
            with torch.no_grad():
                with torch.cuda.graph(gf, pool=pool, capture_error_mode=CAPTURE_MODE):
                    self.pc.refresh()
                    eng, outs, nchw = self.forward(build, training, dtype, self.s_in)
                self.s_out = nchw
                self.s_gout = [torch.zeros_like(o) for o in nchw]
                with torch.cuda.graph(gb, pool=pool, capture_error_mode=CAPTURE_MODE):
                    self.backward(eng, outs, self.s_gout)
            self.s_pgrads = [eng.pgrads.get(p) for p in params]
            eng.pgrads.provider = None
            eng.lock_cache = eng.grad_queue = eng.arena = eng.pack_cache = None
        finally:
            if gc_was:
                gc.enable()
        self.graph_f, self.graph_b = gf, gb


def _hand_out(outs):
    """Private copies of a call site's outputs (its arena / static buffers are rewritten by the next call).  Outputs that are slices of ONE storage - the eight
    full-resolution maps of PraNet-V2 are views of one block - are copied by ONE launch and handed out as the same views of the copy: a loss that takes two of them
    (MyTrain_med.py:78-81) finds them at a fixed stride and reads them in place (pn2.loss)."""
    if len(outs) > 1 and all(o.dtype == torch.float32 and o.is_cuda for o in outs):
        sp = outs[0].untyped_storage().data_ptr()
        if all(o.untyped_storage().data_ptr() == sp for o in outs):
            lo = min(o.storage_offset() for o in outs)
            hi = max(o.storage_offset() + (sum((n - 1) * st for n, st in zip(o.shape, o.stride())) + 1 if o.numel() else 0) for o in outs)
            if hi - lo <= 2 * sum(o.numel() for o in outs):
                flat = torch.empty(0, dtype=torch.float32, device=outs[0].device).set_(outs[0].untyped_storage(), lo, (hi - lo,)).clone()
                return tuple(flat.as_strided(o.shape, o.stride(), o.storage_offset() - lo) for o in outs)
    return tuple(o.clone() for o in outs)


class _SiteFn(torch.autograd.Function):
    """One eager pass of a call site on its arena / tables (the two passes before the capture)."""

    @staticmethod
    def forward(ctx, site, build, training, dtype, n_in, *tensors):
        eng, outs, nchw = site.forward(build, training, dtype, tensors[:n_in])
        site.begin(ctx)
        ctx.eng, ctx.outs, ctx.params, ctx.n_in = eng, outs, tensors[n_in:], n_in
        return _hand_out(nchw)         # the arena is reused by the next call: hand out copies

    @staticmethod
    def backward(ctx, *gouts):
        site, eng = ctx.site, ctx.eng
        site.check(ctx)
        site.backward(eng, ctx.outs, gouts)
        gpar = site.hand_out([eng.pgrads.get(p) for p in ctx.params])
        site.bwd_done = True
        eng.pgrads.provider = None
        eng.lock_cache = eng.grad_queue = eng.arena = eng.pack_cache = None
        ctx.eng = ctx.outs = None
        return (None, None, None, None, None, *([None] * ctx.n_in), *gpar)


class _ReplayFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, site, n_in, *tensors):
        for s_, x in zip(site.s_in, tensors[:n_in]):
            s_.copy_(x, non_blocking=True)
        site.graph_f.replay()
        site.begin(ctx)
        ctx.n_in = n_in
        return _hand_out(site.s_out)

    @staticmethod
    def backward(ctx, *gouts):
        site = ctx.site
        site.check(ctx)
        for s_, g in zip(site.s_gout, gouts):
            if g is None:
                s_.zero_()
            else:
                s_.copy_(g, non_blocking=True)
        site.graph_b.replay()
        site.bwd_done = True
        return (None, None, *([None] * ctx.n_in), *site.hand_out(site.s_pgrads))


def _site(build, inputs, params, training, dtype, pc):
    fn = getattr(build, "__func__", build)
    key = (fn.__code__, len(params), tuple(tuple(x.shape) for x in inputs), dtype)
    sites = params[0].__dict__.setdefault("_pn2_sites", _Sites())
    st = sites.get(key)
    if st is not None and st.pc is not pc:            # the pack cache was rebuilt (a weight was re-allocated): the graphs hold the old pointers
        st = None
    sites.pop(key, None)
    if st is None:
        while len(sites) >= MAX_SITES:
            sites.pop(next(iter(sites)))          # the least recently used input shape
        st = _Site(params, pc)
    sites[key] = st                               # (re-inserted: most recently used last)
    return st


def run_module(build, inputs, params, training, dtype=None):
    """Run `build(eng, *acts) -> [Act]` on NCHW inputs; returns a tuple of NCHW fp32 tensors."""
    dtype = _compute_dtype if dtype is None else dtype
    params = [p for p in params]
    need = torch.is_grad_enabled() and (any(p.requires_grad for p in params) or any(x.requires_grad for x in inputs))
    if need and training and MODULE_GRAPH and params and not any(x.requires_grad for x in inputs) and os.environ.get("PN2_MODULE_PACK_CACHE", "1") == "1":
        from .optim import flat_params
        flat_params(params)          # the trained parameters in ONE flat arena, laid out like the site's flat gradient buffer (pn2/optim.py: one-launch clip + Adam)
        pc = _pack_cache(dtype, params, refresh=False)
        site = _site(build, inputs, params, training, dtype, pc)
        if site.calls >= PLAIN_CALLS and not site.busy():
            if site.graph_f is None and site.arena_steps >= 2 and not site.no_graph:
                try:
                    site.capture(build, training, dtype, inputs, params)
                except Exception as e:          # noqa: BLE001   (the context manager has ended the capture: this site stays on eager launches)
                    import warnings
                    warnings.warn(f"capturing this call site failed ({type(e).__name__}: {e}); it keeps running eager launches")
                    site.no_graph, site.graph_f, site.graph_b = True, None, None
                    torch.cuda.synchronize()
            if site.graph_f is not None:
                return _ReplayFn.apply(site, len(inputs), *inputs, *params)
            pc.refresh()
            site.arena_steps += 1
            return _SiteFn.apply(site, build, training, dtype, len(inputs), *inputs, *params)
        site.calls += 1
    pc = _pack_cache(dtype, params)
    if not need:
        eng = _module_engine(dtype, training, False, pc)
        acts = [eng.from_nchw(x) for x in inputs]
        outs = build(eng, *acts)
        eng.finish_forward()
        return tuple(eng.to_nchw(o) for o in outs)
    return _GraphFn.apply(build, training, dtype, pc, len(inputs), *inputs, *params)
