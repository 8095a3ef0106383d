"""Engine ops, part 3: pooling, bilinear resampling, element-wise ops, concat copies and the DSRA fusion / reverse-attention gate.  Mixed into
pn2.engine.Engine."""
import math
import os

import torch

from . import core
from .capi import call, F32, BF16
from .core import (Act, _p, _stream)


class SpatialOps:
    # ------------------------------------------------------------------ pooling
    def maxpool3x3s2(self, x):
        N, H, W = x.N, x.H, x.W
        OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = Act(self, self.empty(N, OH, OW, x.Cp), x.C, x.gw, x.gwp, x.dt)
        idx = torch.empty((N, OH, OW, x.Cp), dtype=torch.uint8, device=self.dev)
        call.pn2_maxpool3x3s2_fwd(x.dt, x.ptr, x.ld, y.ptr, y.ld, _p(idx), N, H, W, x.Cp, OH, OW, _stream())

        def bwd():
            if not x.requires_grad:
                return
            gx, acc = x.grad_sink()
            assert not acc
            call.pn2_maxpool3x3s2_bwd(x.dt, _p(y.grad_buf()), y.grad_buf().stride(2), _p(idx), _p(gx), gx.stride(2), N, H, W, x.Cp, OH, OW, _stream())
        self.record(bwd)
        return y

    def avgpool(self, x, k, stride, pad, ceil_mode=False, count_include_pad=True, out=None, fold_bwd=False):
        """fold_bwd: (2, 2, 0) pools of an activation whose gradient a later-running conv dgrad completes (x_last): the backward launches nothing and leaves the pooled
        gradient on x (Act.pool_prior) - that dgrad's epilogue adds 1/4 of it (ConvOps.conv_bn_act; flush_pool_prior() is the fall-back)."""
        N, H, W = x.N, x.H, x.W

        def osz(i):
            o = (i + 2 * pad - k + (stride - 1 if ceil_mode else 0)) // stride + 1
            if ceil_mode and (o - 1) * stride >= i + pad:
                o -= 1
            return o
        OH, OW = osz(H), osz(W)
        y = out if out is not None else Act(self, self.empty(N, OH, OW, x.Cp), x.C, x.gw, x.gwp, x.dt)
        assert (y.H, y.W, y.Cp) == (OH, OW, x.Cp)
        inc = 1 if count_include_pad else 0
        call.pn2_avgpool_fwd(x.dt, x.ptr, x.ld, y.ptr, y.ld, N, H, W, x.Cp, OH, OW, k, stride, pad, inc, _stream())

        def bwd():
            if not x.requires_grad:
                return
            gy = y.grad_buf()
            if (fold_bwd and core.POOL_FOLD and (k, stride, pad) == (2, 2, 0) and H % 2 == 0 and W % 2 == 0 and not x.grad_written and x.pool_prior is None
                    and x.dt == self.dt and gy.stride(2) % (4 if x.dt == F32 else 8) == 0):
                x.pool_prior = (gy, (N, H, W, x.Cp, OH, OW, k, stride, pad, inc))
                self._pending_pool.append(x)
                return
            gx, acc = x.grad_sink()
            call.pn2_avgpool_bwd(x.dt, _p(gy), gy.stride(2), _p(gx), gx.stride(2), N, H, W, x.Cp, OH, OW, k, stride, pad, inc, acc, _stream())
        self.record(bwd)
        return y

    def flush_pool_prior(self, x):
        """A deferred AvgPool2d(2, 2) backward that no dgrad epilogue took: the plain launch, now."""
        if x.pool_prior is None:
            return
        gy, (N, H, W, Cp, OH, OW, k, stride, pad, inc) = x.pool_prior
        x.pool_prior = None
        self._pending_pool.remove(x)
        gx, acc = x.grad_sink()
        call.pn2_avgpool_bwd(x.dt, _p(gy), gy.stride(2), _p(gx), gx.stride(2), N, H, W, Cp, OH, OW, k, stride, pad, inc, acc, _stream())

    # ------------------------------------------------------------------ bilinear
    def bilinear(self, x, scale=None, align_corners=False, out=None):
        """F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=...) — the given scale is used
        for the source-index map when align_corners is False (PyTorch default recompute_scale_factor=None)."""
        N, H, W = x.N, x.H, x.W
        OH, OW = int(math.floor(H * scale)), int(math.floor(W * scale))
        if align_corners:
            rh = (H - 1) / (OH - 1) if OH > 1 else 0.0
            rw = (W - 1) / (OW - 1) if OW > 1 else 0.0
        else:
            rh = rw = 1.0 / scale
        return self._resize(x, OH, OW, align_corners, rh, rw, out)

    def resize_to(self, x, OH, OW, align_corners=False):
        """F.interpolate(x, size=(OH,OW), mode='bilinear')"""
        if align_corners:
            rh = (x.H - 1) / (OH - 1) if OH > 1 else 0.0
            rw = (x.W - 1) / (OW - 1) if OW > 1 else 0.0
        else:
            rh, rw = x.H / OH, x.W / OW
        return self._resize(x, OH, OW, align_corners, rh, rw, None)

    def _resize(self, x, OH, OW, ac, rh, rw, out):
        N, H, W = x.N, x.H, x.W
        y = out if out is not None else Act(self, self.empty(N, OH, OW, x.Cp, x.dt), x.C, x.gw, x.gwp, x.dt)
        ac = 1 if ac else 0
        if self.fuse_tail and out is not None and out.lat is not None and x.dt == F32 and x.Cp == 1 and x.ld == 1 and OW % 4 == 0 and OW <= 1024:
            self.tail[out.lat] = (x, ac, rh, rw)      # produced (and differentiated) by pn2_dsra_tail_fwd / _bwd
            return y
        call.pn2_bilinear_fwd(x.dt, x.ptr, x.ld, y.ptr, y.ld, N, H, W, x.Cp, OH, OW, ac, rh, rw, _stream())

        def bwd():
            if not x.requires_grad:
                return
            if not (y.grad_written or y.child_written):
                return              # nothing ever contributed to this output's gradient (e.g. the K = 1 DSRA crop maps): its adjoint is exactly zero
            gy = y.grad_buf()
            gx, acc = x.grad_sink()
            st = _stream()
            # fp32 K-channel class maps (EMCAD's K = 9 laterals): the one-launch row kernel (vertical sums with 16-byte loads, then the row's columns in LDS) instead of the
            # two generic gathers - config 5: 1.29 -> 1.04 ms per step for the 11 adjoints (PN2_BL_ROWS=0: the separable pair)
            rows_ok = x.dt == F32 and x.Cp <= 16 and gy.stride(2) == x.Cp and OW * x.Cp <= 8192 and os.environ.get('PN2_BL_ROWS', '1') == '1'
            if OH >= 4 * H and OW >= 4 * W and x.Cp >= (4 if x.dt == F32 else 8) and not rows_ok:
                # separable adjoint: reduce along x first, then along y (keeps per-thread loops short)
                tmp = self.empty(N, OH, W, x.Cp, x.dt)
                call.pn2_bilinear_bwd(x.dt, _p(gy), gy.stride(2), _p(tmp), x.Cp, N, OH, W, x.Cp, OH, OW, ac, 1.0, rw, 0, st)
                call.pn2_bilinear_bwd(x.dt, _p(tmp), x.Cp, _p(gx), gx.stride(2), N, H, W, x.Cp, OH, W, ac, rh, 1.0, acc, st)
            else:
                call.pn2_bilinear_bwd(x.dt, _p(gy), gy.stride(2), _p(gx), gx.stride(2), N, H, W, x.Cp, OH, OW, ac, rh, rw, acc, st)
        self.record(bwd)
        return y

    # ------------------------------------------------------------------ element-wise
    def binary(self, op, a, b, out=None, grad_alias=False):
        """op 0: a+b ; op 1: a*b (same geometry).  Gradients flow to both operands.
        grad_alias (op 0 only): the caller guarantees that `b` has no other consumer - the sum then keeps its gradient IN b's gradient
        storage (d(a+b)/db = 1), so the backward pass is one accumulate into a's gradient instead of two copies."""
        assert (a.N, a.H, a.W, a.Cp) == (b.N, b.H, b.W, b.Cp) and a.dt == b.dt
        y = out if out is not None else Act(self, self.empty(a.N, a.H, a.W, a.Cp, a.dt), a.C, a.gw, a.gwp, a.dt)
        call.pn2_binary(a.dt, op, a.ptr, a.ld, b.ptr, b.ld, y.ptr, y.ld, a.M, a.Cp, 0, _stream())
        alias = core.GRAD_ALIAS and bool(grad_alias) and op == 0 and out is None and b.requires_grad and self.need_grad
        if alias:
            y.galias = b

        def bwd():
            gy = y.grad_buf()
            st = _stream()
            if alias:
                assert not b._written, "grad_alias: the aliased operand received another gradient"
                b.grad_written = True
            if core.MUL_BWD and op == 1 and a.requires_grad and b.requires_grad and a.grad_buf().data_ptr() != b.grad_buf().data_ptr():
                ga, acc_a = a.grad_sink()          # both gradients of the product from ONE pass over gy
                gb, acc_b = b.grad_sink()
                call.pn2_mul_bwd(a.dt, _p(gy), gy.stride(2), a.ptr, a.ld, b.ptr, b.ld, _p(ga), ga.stride(2), acc_a, _p(gb), gb.stride(2), acc_b, a.M, a.Cp, st)
                return
            for u, v in ((a, b), (b, a)):
                if not u.requires_grad or (alias and u is b):
                    continue
                gu, acc = u.grad_sink()
                if op == 0:
                    call.pn2_copy(a.dt, _p(gy), gy.stride(2), a.dt, _p(gu), gu.stride(2), a.M, a.Cp, acc, st)
                else:
                    call.pn2_binary(a.dt, 1, _p(gy), gy.stride(2), v.ptr, v.ld, _p(gu), gu.stride(2), a.M, a.Cp, acc, st)
        self.record(bwd)
        return y

    def add(self, a, b, out=None, grad_alias=False):
        return self.binary(0, a, b, out, grad_alias)

    def mul(self, a, b, out=None):
        return self.binary(1, a, b, out)

    def copy_into(self, src, dst, forward=True):
        """dst = src (a channel slice of a concat buffer); forward=False: the producer of src has already written dst as well (conv_bn_act(tee=)), only the
        backward of the copy is recorded."""
        if forward:
            call.pn2_copy(src.dt, src.ptr, src.ld, dst.dt, dst.ptr, dst.ld, src.M, src.Cp, 0, _stream())

        def bwd():
            if not src.requires_grad:
                return
            gd = dst.grad_buf()
            gs, acc = src.grad_sink()
            if gd.data_ptr() == gs.data_ptr() and gd.stride() == gs.stride() and dst.dt == src.dt:
                assert not acc, "aliased gradient buffers: the copy must be the only contribution"
                return              # the two gradients share storage (Bottle2neck: d(cat) lives in d(out1)): nothing to move
            call.pn2_copy(dst.dt, _p(gd), gd.stride(2), src.dt, _p(gs), gs.stride(2), src.M, src.Cp, acc, _stream())
        self.record(bwd)
        return dst

    # ------------------------------------------------------------------ DSRA / RA
    def dsra_fuse(self, fg, crop_fg, crop_bg, use_softmax=True):
        """fg + fg * softmax(crop_fg - crop_bg, dim=C)   (fp32 K-channel maps)"""
        for a in (fg, crop_fg, crop_bg):
            assert a.dt == F32 and a.ld == a.C
        K, M = fg.C, fg.M
        y = Act(self, self.empty(fg.N, fg.H, fg.W, K, F32), K, K, K, F32)
        sm = 1 if use_softmax else 0
        call.pn2_dsra_fuse_fwd(fg.ptr, crop_fg.ptr, crop_bg.ptr, y.ptr, M, K, sm, _stream())

        # softmax over ONE channel is identically 1 (num_class = 1, the only value the binary scripts use): y = 2 * fg and the gradient into both crop
        # maps is exactly zero - they receive no contribution at all, so the resamples that produced them skip their adjoints
        zero_crop = bool(K == 1 and sm and core.ZERO_CROP_SKIP)

        def bwd():
            gy = y.grad_buf()
            if zero_crop:
                g, acc = fg.grad_sink()
                dfg = self.fbuf(M, K) if acc else g
                scratch = self.fbuf(2, M, K)
                call.pn2_dsra_fuse_bwd(fg.ptr, crop_fg.ptr, crop_bg.ptr, _p(gy), _p(dfg), _p(scratch[0]), _p(scratch[1]), M, K, sm, _stream())
                if acc:
                    call.pn2_copy(F32, _p(dfg), K, F32, _p(g), g.stride(2), M, K, 1, _stream())
                return
            gs = [a.grad_sink() for a in (fg, crop_fg, crop_bg)]
            tmp = [self.fbuf(M, K) if acc else None for (_, acc) in gs]
            dst = [t if t is not None else g for t, (g, _) in zip(tmp, gs)]
            call.pn2_dsra_fuse_bwd(fg.ptr, crop_fg.ptr, crop_bg.ptr, _p(gy), _p(dst[0]), _p(dst[1]), _p(dst[2]), M, K, sm, _stream())
            for t, (g, acc) in zip(tmp, gs):
                if t is not None:
                    call.pn2_copy(F32, _p(t), K, F32, _p(g), g.stride(2), M, K, 1, _stream())
        self.record(bwd)
        return y

    def ra_gate(self, x, crop):
        """(1 - sigmoid(crop)).expand(C) * x      (PraNet V1 reverse attention)"""
        assert crop.dt == F32 and crop.C == 1
        y = Act(self, self.empty(x.N, x.H, x.W, x.Cp, x.dt), x.C, x.gw, x.gwp, x.dt)
        call.pn2_ra_gate_fwd(x.dt, x.ptr, x.ld, crop.ptr, y.ptr, y.ld, x.M, x.Cp, _stream())

        def bwd():
            gy = y.grad_buf()
            gx, acc = x.grad_sink()
            gc, cacc = crop.grad_sink()
            dc = self.fbuf(x.M) if cacc else gc
            call.pn2_ra_gate_bwd(x.dt, x.ptr, x.ld, crop.ptr, _p(gy), gy.stride(2), _p(gx), gx.stride(2), acc, _p(dc), x.M, x.Cp, _stream())
            if cacc:
                call.pn2_copy(F32, _p(dc), 1, F32, _p(gc), 1, x.M, 1, 1, _stream())
        self.record(bwd)
        return y
