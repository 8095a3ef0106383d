"""Host engine: the Engine class = core state (allocation, tape, lock-step regions) + the op mix-ins.

pn2/core.py       activations, caches, tuning table, behaviour switches, deferred weight-gradient queue
pn2/ops_conv.py   packing, tuners, conv + BatchNorm ops (forward and backward)
pn2/ops_encoder.py  PVTv2 / EMCAD ops
pn2/ops_spatial.py  pooling, resampling, element-wise, DSRA ops
Everything importable from pn2.engine before the split still is (the switches live in pn2.core: patch them there)."""
import ctypes as C
import os

import torch

from . import capi
from . import core
from .capi import call, F32, BF16
from .core import *            # noqa: F401,F403
from .core import _p, _stream, _job_table, _thrash, _w4, _LinearAsConv, _PERMS, _THRASH      # noqa: F401
from .ops_conv import ConvOps
from .ops_encoder import EncoderOps
from .ops_spatial import SpatialOps


class Engine(ConvOps, EncoderOps, SpatialOps):
    def __init__(self, dtype=BF16, training=True, grad_provider=None, need_grad=True, pack_cache=None, tuner=None, grad_queue=None, arena=None, lock_cache=None,
                 bn_fold=None):
        self.bn_fold = bn_fold          # BnFoldCache: eval-mode BatchNorm rows kept across forwards (the caller refreshes it once per forward); None = one prepare launch per layer
        self.lock_cache = lock_cache    # dict shared across steps: device job tables of the lock-step regions (Engine.lockstep); None = no lock-step batching
        self.pack_cache = pack_cache
        self.grad_queue = grad_queue
        self.arena = arena
        self.tuner = tuner              # dict shared across steps: conv shape -> tuned kernel/tile code (bf16 only)
        if not torch.cuda.is_available():
            raise RuntimeError("pranet-v2_amd runs on MI355X only: no GPU visible and there is no CPU fallback")
        capi.load()
        self.dt = dtype
        self.tdt = TORCH_DT[dtype]
        self.training = training
        self.need_grad = need_grad      # decided by the caller (grad mode is off inside autograd.Function.forward)
        self.tape = []
        self.cjobs, self.cin, self.ckeep, self.cseg, self.ctables = [], [], [], 0, []      # queued pn2_colsum_finalize jobs (see colsum_finalize)
        self.pgrads = ParamGrads(grad_provider)
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.bn_modules = []            # for num_batches_tracked bookkeeping
        self._lat = None                # contiguous block of the model's full-resolution output maps
        self.fuse_tail = False          # trainer: leave the lateral up-sampling to the fused DSRA tail kernels (K = 1)
        self.tail = {}                  # lateral slot -> (low-res source Act, align_corners, rh, rw) when fuse_tail
        self._keep = []
        self._pending_pool = []          # activations with a deferred AvgPool2d(2, 2) backward (Act.pool_prior)

    # ------------------------------------------------------------------ allocation / layout
    def alloc(self, shape, dtype):
        if self.arena is not None:
            return self.arena.alloc(tuple(shape), dtype, self.dev)
        return torch.empty(tuple(shape), dtype=dtype, device=self.dev)

    def empty(self, N, H, W, Cp, dt=None):
        return self.alloc((N, H, W, Cp), TORCH_DT[self.dt if dt is None else dt])

    def new_act(self, N, H, W, C_, gw=None, gwp=None, dt=None, zero=False):
        gw = C_ if gw is None else gw
        gwp = rup(gw, 8) if gwp is None else gwp
        Cp = (C_ + gw - 1) // gw * gwp
        t = self.empty(N, H, W, Cp, dt)
        if zero:
            t.zero_()
        return Act(self, t, C_, gw, gwp, self.dt if dt is None else dt)

    def lateral_out(self, j, nmaps, N, OH, OW, K):
        """j-th full-resolution fp32 output map, carved from one [nmaps][N][OH][OW][K] block so the fused
        structure-loss kernels can walk all supervision pairs with a single base pointer + stride."""
        if self._lat is None:
            self._lat = self.alloc((nmaps, N, OH, OW, K), torch.float32)
        a = Act(self, self._lat[j], K, K, K, F32)
        a.lat = j
        return a

    def lateral_block(self):
        return self._lat

    def fbuf(self, *shape):
        return self.alloc(shape, torch.float32)

    def from_nchw(self, x, requires_grad=False, dt=None):
        """fp32 NCHW module input -> NHWC compute dtype (or `dt`), channels zero-padded to a multiple of 8."""
        if not x.is_cuda:
            raise RuntimeError("pranet-v2_amd ops need GPU tensors (no CPU fallback)")
        x = x.contiguous().float()
        N, Cc, H, W = x.shape
        a = self.new_act(N, H, W, Cc, dt=dt)
        call.pn2_nchw_to_nhwc(a.dt, _p(x), a.ptr, a.ld, N, Cc, H * W, a.Cp, _stream())
        a.requires_grad = requires_grad and self.need_grad
        return a

    def cast(self, a, dt):
        """Same activation in another storage dtype (no gradient: used on the resized network input)."""
        if a.dt == dt:
            return a
        assert a.ld == a.Cp
        y = Act(self, self.empty(a.N, a.H, a.W, a.Cp, dt), a.C, a.gw, a.gwp, dt, requires_grad=False)
        call.pn2_copy(a.dt, a.ptr, a.ld, dt, y.ptr, y.ld, a.M, a.Cp, 0, _stream())
        return y

    def to_nchw(self, a):
        """Module output: (N,C,H,W) fp32 tensor.  K=1 maps are returned as zero-copy views."""
        t = a.t[..., :a.C] if a.gw == a.C or a.Cp == a.C else self._gather_logical(a)
        if t.dtype != torch.float32:
            t = t.float()
        return t.permute(0, 3, 1, 2)

    def _gather_logical(self, a):
        idx = torch.tensor([(c // a.gw) * a.gwp + c % a.gw for c in range(a.C)], device=self.dev)
        return a.t.index_select(3, idx)

    def record(self, fn):
        if self.need_grad:
            self.tape.append(fn)

    def backward(self):
        for fn in reversed(self.tape):
            fn()
        self.tape = []
        if self._pending_pool:
            raise RuntimeError("a deferred AvgPool2d backward (avgpool(fold_bwd=True)) was never applied: its activation has no conv consumer that runs later in the backward pass")
        self.flush_colsum()
        self._keep = []

    # ------------------------------------------------------------------ lock step: independent chains share table-driven launches
    def lockstep(self, key, fns):
        """[f() for f in fns] for chains that do not depend on each other: their launches are issued position by position, the launches of equal
        kind at a position as ONE table-driven launch (pn2/lockstep.py) - forward and backward.  `key` names the region (stable across steps).
        Needs a persistent table cache (the trainer's); without one, or inside another lock-step region, the chains simply run one after the other."""
        from . import lockstep as LS
        if not core.LOCKSTEP or self.lock_cache is None or LS._ACTIVE or len(fns) < 2:
            self._in_region = getattr(self, "_in_region", 0) + 1
            try:
                return [f() for f in fns]
            finally:
                self._in_region -= 1
        self._in_region = getattr(self, "_in_region", 0) + 1
        try:
            return self._lockstep_run(LS, key, fns)
        finally:
            self._in_region -= 1

    def _lockstep_run(self, LS, key, fns):
        self._nregion = getattr(self, "_nregion", 0) + 1          # regions are entered in the same order every step: a stable cache key
        key = f"{key}#{self._nregion}"
        rec = LS.Lockstep(key + ":f", self.lock_cache)
        outs, segs = [], []
        for f in fns:
            t0 = len(self.tape)
            with rec.lane():
                outs.append(f())
            segs.append(self.tape[t0:])
            del self.tape[t0:]
        rec.emit()
        if self.need_grad and any(segs):
            def bwd():
                if LS._ACTIVE:                     # inside an outer region's backward: plain order
                    for seg in reversed(segs):
                        for fn in reversed(seg):
                            fn()
                    return
                r = LS.Lockstep(key + ":b", self.lock_cache)
                for seg in reversed(segs):
                    with r.lane():
                        for fn in reversed(seg):
                            fn()
                r.emit()
            self.record(bwd)
        return outs

    # ------------------------------------------------------------------ bookkeeping
    def finish_forward(self):
        """BatchNorm's num_batches_tracked += 1 (nn.BatchNorm2d train-mode side effect)."""
        seen, ctrs = set(), []
        for bn in self.bn_modules:
            if id(bn) not in seen and bn.num_batches_tracked is not None:
                ctrs.append(bn.num_batches_tracked)
                seen.add(id(bn))
        if ctrs:
            torch._foreach_add_(ctrs, 1)       # one multi-tensor launch for all ~157 counters
        self.bn_modules = []
