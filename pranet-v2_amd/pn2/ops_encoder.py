"""Engine ops, part 2: what the PVTv2 encoder and the EMCAD decoder need beyond convs - LayerNorm, attention, depth-wise convs, DropPath, channel
shuffles, gates, global pools.  Mixed into pn2.engine.Engine."""
import ctypes as C

import torch

from . import capi
from . import core
from .capi import call, F32, BF16
from .core import (Act, _PERMS, _p, _stream, rup)


_DP_KEEP = {}          # (drop probabilities, batch, device) -> [K][N] table of keep probabilities (drop_path_plan)


class EncoderOps:
    def layernorm(self, x, ln):
        """nn.LayerNorm over the channel axis (tokens = pixels)."""
        assert x.Cp == x.C and x.ld == x.Cp and tuple(ln.normalized_shape) == (x.C,)
        M, Cc, st = x.M, x.C, _stream()
        y = Act(self, self.empty(x.N, x.H, x.W, Cc), Cc, Cc, Cc, self.dt)
        mean, rstd = self.fbuf(M), self.fbuf(M)
        call.pn2_layernorm_fwd(self.dt, x.ptr, x.ld, y.ptr, y.ld, M, Cc, _p(ln.weight), _p(ln.bias), float(ln.eps), _p(mean), _p(rstd), st)

        def bwd():
            st = _stream()
            dy = y.grad_buf()
            assert y.grad_written
            nb = call.pn2_rows_blocks(M, call.pn2_ln_slots(self.dt, Cc))
            pg, pb = self.fbuf(nb, Cc), self.fbuf(nb, Cc)
            gx, acc = x.grad_sink() if x.requires_grad else (self.empty(x.N, x.H, x.W, Cc), 0)
            call.pn2_layernorm_bwd(self.dt, _p(dy), dy.stride(2), x.ptr, x.ld, M, Cc, _p(ln.weight), _p(mean), _p(rstd), _p(gx), gx.stride(2), acc,
                                   _p(pg), _p(pb), nb, st)
            gg, ga = self.pgrads.sink(ln.weight)
            gb, gba = self.pgrads.sink(ln.bias)
            self.colsum_finalize(pg, nb, Cc, Cc, gg, ga)
            self.colsum_finalize(pb, nb, Cc, Cc, gb, gba)
        self.record(bwd)
        return y

    def dwconv_gelu(self, x, conv):
        """gelu(DWConv(x)) of Mlp.forward (pvtv2.py:44-45): depth-wise 3x3, pad 1, bias, exact GELU."""
        Cc = x.C
        assert x.Cp == Cc and x.ld == Cc and conv.groups == Cc and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.stride == (1, 1)
        N, H, W, st = x.N, x.H, x.W, _stream()
        z = self.empty(N, H, W, Cc)
        y = Act(self, self.empty(N, H, W, Cc), Cc, Cc, Cc, self.dt)
        call.pn2_dwconv3x3(self.dt, x.ptr, _p(conv.weight), _p(conv.bias), _p(z), y.ptr, N, H, W, Cc, 0, 0, st)

        def bwd():
            st = _stream()
            dy = y.grad_buf()
            assert y.grad_written and dy.stride(2) == Cc
            dz = self.empty(N, H, W, Cc)
            nb = call.pn2_dwconv3x3_wgrad_blocks(self.dt, N, H, W, Cc)
            part = self.fbuf(nb, Cc * 10)
            # dz = dy * gelu'(z) is formed inside the weight-gradient walk (one pass over dy, z, x) and kept for the data gradient
            call.pn2_dwconv3x3_wgrad(self.dt, _p(dy), x.ptr, _p(part), nb, N, H, W, Cc, _p(z), _p(dz), st)
            gw, gwa = self.pgrads.sink(conv.weight)
            gb, gba = self.pgrads.sink(conv.bias)
            self.colsum_finalize(part, nb, Cc * 9, Cc * 10, gw, gwa)
            self.colsum_finalize(part[:, Cc * 9:], nb, Cc, Cc * 10, gb, gba)
            if x.requires_grad:
                gx, acc = x.grad_sink()
                assert gx.stride(2) == Cc
                nbc = int(call.pn2_dwconv3x3_colsum_blocks(self.dt, N, H, W, Cc)) if (core.DW_COLSUM and not acc and gx.is_contiguous()) else -1
                if nbc >= 1:
                    # this data gradient IS the output gradient of the Linear in front (Mlp.fc1): its column sums - fc1's bias gradient - are left by the
                    # same walk (per-workgroup partial rows), so the deferred column-sum pass does not read the block's largest gradient tensor again
                    cpart = self.fbuf(nbc, Cc)
                    call.pn2_dwconv3x3_colsum(self.dt, _p(dz), _p(conv.weight), C.c_void_p(0), _p(gx), N, H, W, Cc, 1, _p(cpart), nbc, st)
                    x.colparts = (cpart, nbc, gx)
                else:
                    call.pn2_dwconv3x3(self.dt, _p(dz), _p(conv.weight), C.c_void_p(0), _p(gx), C.c_void_p(0), N, H, W, Cc, 1, acc, st)
        self.record(bwd)
        return y

    def drop_path_plan(self, drop_probs, N):
        """The DropPath draws of a whole forward in TWO launches (one bernoulli over a [K][N] table of keep probabilities, one divide) instead of two per
        residual branch (60 tiny launches per PVTv2-B2 step): `drop_probs` = the non-zero drop probabilities in the order drop_path / drop_path_add will ask
        for them.  A request that does not match the plan (other probability, other batch) falls back to its own draw."""
        ps = [float(p) for p in drop_probs if p > 0.0]
        self._dp_rows, self._dp_next = None, 0
        if not ps or not self.training:
            return
        key = (tuple(ps), N, str(self.dev))
        keep = _DP_KEEP.get(key)
        if keep is None:
            keep = _DP_KEEP[key] = (1.0 - torch.tensor(ps, dtype=torch.float32, device=self.dev)).reshape(-1, 1).expand(len(ps), N).contiguous()
        self._dp_rows = (torch.bernoulli(keep).div_(keep), ps)

    def _dp_scale(self, N, drop_prob):
        """[N] fp32: keep mask / keep probability of the next DropPath (from the plan's table when it matches)."""
        rows = getattr(self, "_dp_rows", None)
        if rows is not None and self._dp_next < len(rows[1]) and rows[1][self._dp_next] == float(drop_prob) and rows[0].shape[1] == N:
            sc = rows[0][self._dp_next]
            self._dp_next += 1
            return sc
        keep = 1.0 - drop_prob
        return torch.empty(N, dtype=torch.float32, device=self.dev).bernoulli_(keep).div_(keep)

    def drop_path(self, x, drop_prob):
        """timm DropPath in train mode: every sample is kept with probability 1 - drop_prob and rescaled by 1 / keep."""
        if drop_prob == 0.0 or not self.training:
            return x
        sc = self._dp_scale(x.N, drop_prob)
        assert x.ld == x.Cp
        y = Act(self, self.empty(x.N, x.H, x.W, x.Cp), x.C, x.gw, x.gwp, self.dt)
        per = x.H * x.W * x.Cp
        call.pn2_scale_samples(self.dt, x.ptr, y.ptr, _p(sc), C.c_void_p(0), x.N, per, _stream())

        def bwd():
            dy = y.grad_buf()
            assert y.grad_written and not x.grad_written
            gx, _ = x.grad_sink()
            call.pn2_scale_samples(self.dt, _p(dy), _p(gx), _p(sc), C.c_void_p(0), x.N, per, _stream())
        self.record(bwd)
        return y

    def drop_path_add(self, res, x, drop_prob):
        """res + DropPath(x) in one pass (Block.forward pvtv2.py:148-149 in train mode); plain add when nothing is dropped."""
        if drop_prob == 0.0 or not self.training:
            return self.add(res, x)
        sc = self._dp_scale(x.N, drop_prob)
        assert x.ld == x.Cp and res.ld == res.Cp and (res.N, res.H, res.W, res.Cp) == (x.N, x.H, x.W, x.Cp) and res.dt == x.dt == self.dt
        y = Act(self, self.empty(x.N, x.H, x.W, x.Cp), x.C, x.gw, x.gwp, self.dt)
        per = x.H * x.W * x.Cp
        call.pn2_scale_samples(self.dt, x.ptr, y.ptr, _p(sc), res.ptr, x.N, per, _stream())

        def bwd():
            st = _stream()
            dy = y.grad_buf()
            assert y.grad_written and not x.grad_written and dy.stride(2) == x.Cp
            gx, _ = x.grad_sink()
            call.pn2_scale_samples(self.dt, _p(dy), _p(gx), _p(sc), C.c_void_p(0), x.N, per, st)
            if res.requires_grad:
                if (res.grad is None and res.galias is None and res.parent is None and not res.grad_written and tuple(dy.shape) == tuple(res.t.shape)
                        and dy.dtype == res.t.dtype and dy.is_contiguous()):
                    # d(y)/d(res) = 1: dy IS the residual stream's gradient - share the buffer (it is dead here: gx has been formed; the LayerNorm backward of
                    # the branch accumulates into it in place) instead of copying it per residual add
                    res.grad = dy
                    res.grad_written = True
                else:
                    gr, acc = res.grad_sink()
                    call.pn2_copy(self.dt, _p(dy), dy.stride(2), self.dt, _p(gr), gr.stride(2), x.M, x.Cp, acc, st)
        self.record(bwd)
        return y

    def attention(self, q, kv, heads):
        """softmax(q k^T / sqrt(hd)) v with the heads concatenated (Attention.forward pvtv2.py:103-107); kv holds k then v."""
        Cc = q.C
        hd = Cc // heads
        assert q.Cp == Cc and q.ld == Cc and kv.C == 2 * Cc and kv.ld == 2 * Cc and kv.N == q.N
        B, Nq, Nkv, st = q.N, q.H * q.W, kv.H * kv.W, _stream()
        scale = hd ** -0.5
        o = Act(self, self.empty(q.N, q.H, q.W, Cc), Cc, Cc, Cc, self.dt)
        lse = self.fbuf(B, heads, Nq)
        call.pn2_attn_fwd(self.dt, q.ptr, Cc, kv.ptr, 2 * Cc, o.ptr, Cc, _p(lse), B, Nq, Nkv, heads, hd, scale, st)

        def bwd():
            st = _stream()
            do = o.grad_buf()
            assert o.grad_written and do.stride(2) == Cc and not q.grad_written and not kv.grad_written
            part = self.fbuf(B, heads, call.pn2_attn_bwd_blocks(self.dt, B, heads, Nq), 2, rup(Nkv, 64), 64)
            delta = self.fbuf(B, heads, Nq)
            gq, _ = q.grad_sink()
            gkv, _ = kv.grad_sink()
            call.pn2_attn_bwd(self.dt, q.ptr, Cc, kv.ptr, 2 * Cc, o.ptr, Cc, _p(do), Cc, _p(lse), _p(gq), Cc, _p(gkv), 2 * Cc, _p(part), _p(delta),
                              B, Nq, Nkv, heads, hd, scale, st)
        self.record(bwd)
        return o

    # ------------------------------------------------------------------ EMCAD decoder ops (multiclass_seg/EMCAD/lib/decoders.py)
    def bn_after(self, N, H, W, Cc, nblk, launch, back, bn, relu=False, residual=None, bias=None):
        """y = act(BN(raw) + residual) for a producer other than the implicit-GEMM conv: `launch(raw, psum, psq)` writes raw [M][Cc] and
        nblk partial rows of sum / sum of squares; `back(draw)` receives the gradient w.r.t. raw.  relu: False / True / 2 (ReLU6)."""
        assert Cc % 8 == 0
        M, st, train = N * H * W, _stream(), self.training
        raw = self.empty(N, H, W, Cc)
        psum, psq = (self.fbuf(nblk, Cc), self.fbuf(nblk, Cc)) if train else (None, None)
        launch(raw, psum, psq)
        bd = capi.BnDesc()
        bd.M, bd.Cp, bd.C, bd.gw, bd.gwp, bd.eps, bd.momentum = M, Cc, Cc, Cc, Cc, bn.eps, (bn.momentum if bn.momentum is not None else 0.1)
        scale, shift = self.fbuf(Cc), self.fbuf(Cc)
        mean = invstd = None
        if train:
            mean, invstd = self.fbuf(Cc), self.fbuf(Cc)
            call.pn2_bn_finalize(_p(psum), _p(psq), nblk, C.byref(bd), _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var),
                                 _p(scale), _p(shift), _p(mean), _p(invstd), st)
            self.bn_modules.append(bn)
            if bias is not None:
                with torch.no_grad():
                    bn.running_mean.add_(bias.detach(), alpha=bd.momentum)
        else:
            call.pn2_bn_eval_prepare(C.byref(bd), _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var), _p(scale), _p(shift), st)
            if bias is not None:
                shift += bias.detach() * scale
        out = Act(self, self.empty(N, H, W, Cc), Cc, Cc, Cc, self.dt)
        if residual is not None:
            assert residual.Cp == Cc and residual.dt == self.dt
        call.pn2_affine_act(self.dt, _p(raw), Cc, self.dt, out.ptr, out.ld, M, Cc, _p(scale), _p(shift),
                            residual.ptr if residual is not None else C.c_void_p(0), residual.ld if residual is not None else 0, (2 if relu == 2 else 1) if relu else 0, st)
        if not self.need_grad:
            return out

        def bwd():
            st = _stream()
            if not train:
                raise RuntimeError("backward through eval-mode BatchNorm is not supported")
            dy = out.grad_buf()
            assert out.grad_written or out.child_written
            draw = self.empty(N, H, W, Cc)
            ymask = out if relu else None
            r6 = 1 if relu == 2 else 0
            nul = C.c_void_p(0)
            nb = call.pn2_bn_bwd_blocks(M, Cc, self.dt)
            p1, p2 = self.fbuf(nb, Cc), self.fbuf(nb, Cc)
            call.pn2_bn_bwd_reduce(self.dt, self.dt, _p(dy), dy.stride(2), Cc, ymask.ptr if ymask else nul, ymask.ld if ymask else 0, self.dt,
                                   _p(raw), Cc, M, Cc, _p(mean), _p(invstd), _p(p1), _p(p2), nb, nul, nul, r6, st)
            coef = self.fbuf(3 * Cc)
            gg, ga = self.pgrads.sink(bn.weight)
            gb, gba = self.pgrads.sink(bn.bias)
            call.pn2_bn_bwd_finalize(_p(p1), _p(p2), nb, C.byref(bd), _p(bn.weight), _p(invstd), _p(gg), _p(gb), ga, _p(coef), st)
            rg, racc = (None, 0)
            if residual is not None and residual.requires_grad:
                rg, racc = residual.grad_sink()
            call.pn2_bn_bwd_apply(self.dt, self.dt, _p(dy), dy.stride(2), Cc, ymask.ptr if ymask else nul, ymask.ld if ymask else 0, self.dt,
                                  _p(raw), Cc, M, Cc, _p(mean), _p(invstd), _p(coef), _p(draw), Cc,
                                  _p(rg), rg.stride(2) if rg is not None else 0, racc, nul, nul, r6, st)
            if bias is not None:
                gbi, gbia = self.pgrads.sink(bias)
                self.colsum(draw, M, Cc, Cc, gbi, gbia)
            back(draw)
        self.record(bwd)
        return out

    def dwconv_bn_act(self, x, conv, bn, relu=False):
        """act(BN(depth-wise KxK conv(x))), K in (1, 3, 5), stride 1, pad K/2, bias-free (MSDC decoders.py:90-96, EUCB :172-174)."""
        Cc, K = x.C, conv.kernel_size[0]
        assert x.Cp == Cc and x.ld == Cc and conv.groups == Cc and conv.bias is None and conv.stride == (1, 1) and conv.padding == (K // 2, K // 2)
        N, H, W = x.N, x.H, x.W
        nblk = call.pn2_dwconv_blocks(self.dt, N, H, W, Cc, K, 0)
        w = conv.weight

        def launch(raw, psum, psq):
            call.pn2_dwconv(self.dt, x.ptr, _p(w), _p(raw), N, H, W, Cc, K, 0, 0, _p(psum), _p(psq), _stream())

        def back(draw):
            st = _stream()
            nbw = call.pn2_dwconv_blocks(self.dt, N, H, W, Cc, K, 1)
            part = self.fbuf(nbw, Cc * K * K)
            call.pn2_dwconv_wgrad(self.dt, _p(draw), x.ptr, _p(part), N, H, W, Cc, K, st)
            gw, gwa = self.pgrads.sink(w)
            self.colsum_finalize(part, nbw, Cc * K * K, Cc * K * K, gw, gwa)
            if x.requires_grad:
                gx, acc = x.grad_sink()
                assert gx.stride(2) == Cc
                call.pn2_dwconv(self.dt, _p(draw), _p(w), _p(gx), N, H, W, Cc, K, 1, acc, C.c_void_p(0), C.c_void_p(0), st)
        return self.bn_after(N, H, W, Cc, nblk, launch, back, bn, relu=relu)

    def pairconv_bn(self, x, conv, bn, relu=False, residual=None):
        """act(BN(grouped 3x3 conv with two input channels per group (+bias)) + residual)   (LGAG.W_g / W_x, decoders.py:193-200)."""
        F_ = conv.out_channels
        assert x.C == 2 * F_ and x.Cp == x.C and x.ld == x.C and conv.groups == F_ and conv.kernel_size == (3, 3) and conv.padding == (1, 1)
        N, H, W = x.N, x.H, x.W
        nblk = call.pn2_pairconv_blocks(self.dt, N, H, W, F_)
        w = conv.weight

        def launch(raw, psum, psq):
            if psum is None:
                psum, psq = self.fbuf(nblk, F_), self.fbuf(nblk, F_)
            call.pn2_pairconv3x3_fwd(self.dt, x.ptr, _p(w), _p(raw), N, H, W, F_, _p(psum), _p(psq), _stream())

        def back(draw):
            st = _stream()
            part = self.fbuf(nblk, F_ * 18)
            call.pn2_pairconv3x3_wgrad(self.dt, _p(draw), x.ptr, _p(part), N, H, W, F_, st)
            gw, gwa = self.pgrads.sink(w)
            self.colsum_finalize(part, nblk, F_ * 18, F_ * 18, gw, gwa)
            if x.requires_grad:
                gx, acc = x.grad_sink()
                assert gx.stride(2) == 2 * F_
                call.pn2_pairconv3x3_dgrad(self.dt, _p(draw), _p(w), _p(gx), N, H, W, F_, acc, st)
        return self.bn_after(N, H, W, F_, nblk, launch, back, bn, relu=relu, residual=residual, bias=conv.bias)

    def upsample2x(self, x):
        """nn.Upsample(scale_factor=2), nearest (EUCB decoders.py:171)."""
        assert x.ld == x.Cp
        y = Act(self, self.empty(x.N, 2 * x.H, 2 * x.W, x.Cp), x.C, x.gw, x.gwp, x.dt)
        call.pn2_upsample_nearest2x(x.dt, x.ptr, y.ptr, x.N, x.H, x.W, x.Cp, _stream())

        def bwd():
            if x.requires_grad:
                gy = y.grad_buf()
                gx, acc = x.grad_sink()
                assert gx.stride(2) == x.Cp
                call.pn2_upsample_nearest2x_bwd(x.dt, _p(gy), _p(gx), x.N, x.H, x.W, x.Cp, acc, _stream())
        self.record(bwd)
        return y

    def shuffled_sum(self, parts, groups):
        """channel_shuffle(sum(parts), groups)  (MSCB.forward decoders.py:147-154, channel_shuffle :69-77) in one pass; the backward is one gather whose
        result is the gradient of every part."""
        a = parts[0]
        Cc, M = a.C, a.M
        assert all(p.C == Cc and p.Cp == Cc and p.ld == Cc for p in parts) and 1 <= len(parts) <= 3 and Cc % groups == 0
        cpg = Cc // groups
        key = ("perm", Cc, groups)
        if key not in _PERMS:
            fwd = torch.tensor([(j % groups) * cpg + j // groups for j in range(Cc)], dtype=torch.int32)
            inv = torch.empty_like(fwd); inv[fwd.long()] = torch.arange(Cc, dtype=torch.int32)
            _PERMS[key] = (fwd.to(self.dev), inv.to(self.dev))
        fwd, inv = _PERMS[key]
        y = Act(self, self.empty(a.N, a.H, a.W, Cc), Cc, Cc, Cc, self.dt)
        ps = [p.ptr for p in parts] + [C.c_void_p(0)] * (3 - len(parts))
        call.pn2_gather_sum(self.dt, ps[0], ps[1], ps[2], _p(fwd), y.ptr, M, Cc, _stream())

        def bwd():
            gy = y.grad_buf()
            assert y.grad_written and gy.stride(2) == Cc
            g = self.empty(a.N, a.H, a.W, Cc)
            call.pn2_gather_sum(self.dt, _p(gy), C.c_void_p(0), C.c_void_p(0), _p(inv), _p(g), M, Cc, _stream())
            for p in parts:         # single-consumer BN outputs: the shared tensor is only read
                assert not p.grad_written
                p.grad = g
                p.grad_written = True
        self.record(bwd)
        return y

    def sigmoid_gate(self, x, pre, mode):
        """x * sigmoid(pre): mode 0 channel gate, pre = [N,1,1,C] (CAB decoders.py:241,442); mode 1 pixel gate, pre = [N,H,W,1] (SAB :258,443; LGAG :213-214)."""
        N, HW, Cc = x.N, x.H * x.W, x.C
        assert x.Cp == Cc and x.ld == Cc
        n = N * Cc if mode == 0 else N * HW
        usedC = Cc if mode == 0 else 1
        assert pre.M * usedC == n and pre.C >= usedC
        st = _stream()
        g = self.fbuf(n)
        call.pn2_sigmoid(pre.dt, pre.ptr, pre.ld, usedC, _p(g), n, st)
        y = Act(self, self.empty(x.N, x.H, x.W, Cc), Cc, Cc, Cc, self.dt)
        call.pn2_gate_mul(self.dt, x.ptr, _p(g), y.ptr, N, HW, Cc, mode, 0, st)

        def bwd():
            st = _stream()
            gy = y.grad_buf()
            assert y.grad_written and gy.stride(2) == Cc
            dg = self.fbuf(n)
            if mode == 1:
                call.pn2_gate_bwd(self.dt, _p(gy), x.ptr, _p(dg), N, HW, Cc, 1, st)
            else:
                nb = call.pn2_gate_blocks(self.dt, HW, Cc)
                part = self.fbuf(nb, N * Cc)
                call.pn2_gate_bwd(self.dt, _p(gy), x.ptr, _p(part), N, HW, Cc, 0, st)
                call.pn2_colsum_finalize(_p(part), nb, N * Cc, N * Cc, _p(dg), 0, st)
            gp, pacc = pre.grad_sink()
            call.pn2_sigmoid_bwd(pre.dt, _p(dg), _p(g), _p(gp), gp.stride(2), usedC, n, pacc, st)
            if x.requires_grad:
                gx, acc = x.grad_sink()
                assert gx.stride(2) == Cc
                call.pn2_gate_mul(self.dt, _p(gy), _p(g), _p(gx), N, HW, Cc, mode, acc, st)
        self.record(bwd)
        return y

    def global_pool(self, x):
        """(AdaptiveAvgPool2d(1)(x), AdaptiveMaxPool2d(1)(x)) as two [N,1,1,C] maps (CAB decoders.py:234-237)."""
        N, HW, Cc = x.N, x.H * x.W, x.C
        assert x.Cp == Cc and x.ld == Cc
        avg = Act(self, self.empty(N, 1, 1, Cc), Cc, Cc, Cc, self.dt)
        mx = Act(self, self.empty(N, 1, 1, Cc), Cc, Cc, Cc, self.dt)
        arg = self.alloc((N, Cc), torch.int32)
        call.pn2_global_pool(self.dt, x.ptr, avg.ptr, mx.ptr, _p(arg), N, HW, Cc, _stream())

        def bwd():
            if not x.requires_grad:
                return
            ga, gm = avg.grad_buf(), mx.grad_buf()
            if not avg.grad_written:
                ga.zero_()
            if not mx.grad_written:
                gm.zero_()
            gx, acc = x.grad_sink()
            assert gx.stride(2) == Cc
            call.pn2_global_pool_bwd(self.dt, _p(ga), _p(gm), _p(arg), _p(gx), N, HW, Cc, acc, _stream())
        self.record(bwd)
        return avg, mx

    def chan_stats(self, x):
        """cat([mean over channels, max over channels]) as a 2-channel map (8 physical slots)   (SAB decoders.py:253-255)."""
        Cc = x.C
        assert x.Cp == Cc and x.ld == Cc
        y = Act(self, self.empty(x.N, x.H, x.W, 8), 2, 2, 8, self.dt)
        arg = self.alloc((x.M,), torch.int32)
        call.pn2_chan_stats(self.dt, x.ptr, y.ptr, _p(arg), x.M, Cc, _stream())

        def bwd():
            if not x.requires_grad:
                return
            gy = y.grad_buf()
            assert y.grad_written and gy.stride(2) == 8
            gx, acc = x.grad_sink()
            assert gx.stride(2) == Cc
            call.pn2_chan_stats_bwd(self.dt, _p(gy), _p(arg), _p(gx), x.M, Cc, acc, _stream())
        self.record(bwd)
        return y
