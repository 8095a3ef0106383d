"""Engine ops, part 1: weight packing, the per-shape kernel / tile tuners, conv (+ BatchNorm + activation + residual) forward and backward, the
fused multi-conv GEMM, deferred column sums, nn.Linear.  Mixed into pn2.engine.Engine; the behaviour switches are read from pn2.core at call time."""
import ctypes as C

import torch

from . import capi
from . import core
from .capi import call, F32, BF16
from .core import (Act, Bnb, KSPLIT_MINK, TUNE_REPS, WGRAD_SLAB_MB, WGRAD_WGS, _LinearAsConv, _job_table, _p, _stream, _thrash, _w4, rup)

KS_UNTUNED = 3 | (1 << 2) | (2 << 4) | 0x40          # tuning code: LDS-DMA 2-stage ring, 64 x 128 tile, two K groups (see ConvOps._untuned)


class ConvOps:
    # ------------------------------------------------------------------ weights
    def _pack_desc(self, w, x_map, out_map, transposed):
        Cout, Cin, KH, KW = _w4(w)
        gw_in, gwp_in, Cin_p = x_map
        gw_out, gwp_out, Cout_p = out_map
        d = capi.PackDesc()
        d.Cout, d.Cin, d.KH, d.KW = Cout, Cin, KH, KW
        d.Cout_p, d.gw_out, d.gwp_out = Cout_p, gw_out, gwp_out
        d.Cin_p, d.gw_in, d.gwp_in = Cin_p, gw_in, gwp_in
        d.transposed = 1 if transposed else 0
        if transposed:
            d.Rp, d.Kp = rup(Cin_p, 128), rup(KH * KW * Cout_p, 128)
        else:
            d.Rp, d.Kp = rup(Cout_p, 128), rup(KH * KW * Cin_p, 128)
        return d

    def pack(self, w, x_map, out_map, transposed):
        """K-contiguous weight panel in the compute dtype.  With a pack cache (trainer) the panel is persistent and is
        refreshed for ALL convs by one pn2_pack_weights_multi launch per step instead of one launch per conv."""
        cache = self.pack_cache
        key = (id(w), bool(transposed), x_map, out_map, self.dt)
        if cache is not None and key in cache.entries:
            return cache.entries[key]
        d = self._pack_desc(w, x_map, out_map, transposed)
        wp = torch.empty((d.Rp, d.Kp), dtype=self.tdt, device=self.dev)
        call.pn2_pack_weight(self.dt, _p(w), _p(wp), C.byref(d), _stream())
        if cache is not None:
            cache.add(key, w, wp, d)
        return wp, d

    # ------------------------------------------------------------------ per-shape kernel / tile selection
    def _tune_gemm(self, cd, in_ptr, wp, M, Cout, ep=None):
        """Pick (kernel, BM, BN) for this forward/dgrad shape by timing every candidate once (first eager step; results are cached in
        self.tuner and reused under hipGraph capture).  Returns the code for pn2_conv_desc.flags bits 8..15 (0 = library heuristic).
        ep: the launch carries a BatchNorm-backward epilogue (pn2_conv_gemm_ep): the candidates are timed WITH it (its extra operand reads and
        per-tile work favour other tiles than the plain kernel), writing to scratch destinations."""
        fast = self.dt == F32 and capi.F32_MMA == capi.F32F          # fp32fast: tiles of the register-staged kernel are tuned per shape too (64-row tiles win where
        if self.dt != BF16 and not fast:                             # the heuristic takes 128: more workgroups per CU; 716 -> 748 images/s with 64 rows everywhere)
            return 0
        t = self.tuner
        if t is None:                    # PN2_AUTOTUNE=0: the library's heuristic tile, inside the class the shape rule names
            return self._untuned(cd, M, Cout, ep)
        key = ("g", cd.N, cd.H, cd.W, cd.OH, cd.OW, cd.Cin_p, cd.ld_in, Cout, cd.KH, cd.KW, cd.stride, cd.pad_h, cd.pad_w, cd.dil_h, cd.dil_w, cd.transposed)
        if ep is not None:
            key = key + ("ep", ep.a.mode, ep.b.mode, 1 if ep.b.out else 0, cd.flags & capi.CONV_ACCUM) + (("pool",) if ep.pool else ())
        if fast:
            key = key + ("f32f",)
        if key in t:
            if core.TUNE_LOG is not None:
                core.TUNE_LOG.append((key, t[key], "hit"))
            return t[key]
        if torch.cuda.is_current_stream_capturing():
            return self._untuned(cd, M, Cout, ep)
        from . import lockstep as LS
        with LS.pause():                 # the candidates are timed with real launches even inside a lock-step region
            return self._tune_gemm_run(t, key, cd, in_ptr, wp, M, Cout, ep)

    @staticmethod
    def _ks_class(cd, M, Cout):
        """The shape rule of the intra-workgroup split-K kernels (see _tune_gemm_run): about one wave of tiles and a long K loop, no global split-K."""
        ksteps = -(-(cd.KH * cd.KW * cd.Cin_p) // 64)
        return bool(core.KS2 and Cout > 32 and ksteps >= 9 and -(-M // 64) * -(-Cout // 64) <= 484 and not (cd.flags >> 16) & 15)

    def _untuned(self, cd, M, Cout, ep):
        """Tuning code without a timing run (PN2_AUTOTUNE=0, or a shape first met under capture).  The split-K class sums in its own fp32 order, so the
        rule has to hold here too: module eval without the tuner and a tuned Predictor must return the same bits.  2-stage ring, 64 x 128 tile: fits
        the LDS for every shape of the class.  Launches with a BatchNorm-backward epilogue (training only) keep the heuristic: whether a split-K tile
        fits next to their operand tiles is only known by trying, which is the tuner's job."""
        if ep is None and self.dt == BF16 and self._ks_class(cd, M, Cout):
            return KS_UNTUNED
        return 0

    def _tune_gemm_run(self, t, key, cd, in_ptr, wp, M, Cout, ep):
        st = _stream()
        nul = C.c_void_p(0)
        scratch = torch.empty((M, Cout), dtype=self.tdt, device=self.dev)
        d2 = capi.ConvDesc()
        C.memmove(C.byref(d2), C.byref(cd), C.sizeof(capi.ConvDesc))
        d2.ld_out, d2.Cout = Cout, Cout
        if ep is not None:
            d2.flags = cd.flags & capi.CONV_ACCUM
            e2 = capi.ConvEp()
            C.memmove(C.byref(e2), C.byref(ep), C.sizeof(capi.ConvEp))
            nb64 = (M + 63) // 64
            tp = torch.empty((4, nb64, Cout), dtype=torch.float32, device=self.dev)
            e2.a.p1, e2.a.p2, e2.a.ldp = tp[0].data_ptr(), tp[1].data_ptr(), Cout
            if ep.b.out:
                scratch_b = torch.empty((M, Cout), dtype=self.tdt, device=self.dev)
                e2.b.out, e2.b.ld_out = scratch_b.data_ptr(), Cout
                e2.b.p1, e2.b.p2, e2.b.ldp = tp[2].data_ptr(), tp[3].data_ptr(), Cout
            base = d2.flags

            def launch(code):
                d2.flags = base | (code << 8)
                call.pn2_conv_gemm_ep(self.dt, in_ptr, _p(wp), _p(scratch), C.byref(d2), C.byref(e2), st)
        else:
            def launch(code):
                d2.flags = code << 8
                call.pn2_conv_gemm(self.dt, in_ptr, _p(wp), _p(scratch), nul, nul, C.byref(d2), st)
        # Intra-workgroup split-K (conv_dma_gemm_ks, tuning-code bit 6: two K groups of four waves) sums a tile's K loop in another fp32 order than the plain
        # kernels, which all agree bit for bit.  Whether a shape takes it is therefore a RULE of the shape, not of a timing: about one wave of tiles and a long
        # K loop (what the free tuning run of round 5 picked it for) - every call site that computes the same conv gets the same bits, and the tuner chooses
        # kernel / tile inside the class.
        ks2 = self.dt == BF16 and self._ks_class(cd, M, Cout)
        cands, plain = [], []
        for kern in ((1, 2, 3) if self.dt == BF16 else (1,)):             # 1 register-staged, 2 LDS-DMA with a 3-stage ring, 3 LDS-DMA with a 2-stage ring (more workgroups per CU); fp32fast: 1
            for bm in (1, 2):
                if bm == 2 and M <= 64:
                    continue
                for bn in ((1, 2, 3) if self.dt == BF16 else (1, 2)):
                    if (bn == 2 and Cout <= 32) or (bn == 3 and Cout <= 64):
                        continue
                    plain.append(kern | (bm << 2) | (bn << 4))
                    if ks2 and kern >= 2 and bn >= 2 and not (kern == 2 and bm == 2 and bn == 3):          # (128 x 128 with two 3-stage rings: 192 KB)
                        cands.append(kern | (bm << 2) | (bn << 4) | 0x40)
        evs = []
        feasible = []
        for code in (cands + plain if ks2 else plain):
            if ks2 and feasible and not code & 0x40:
                break          # (the plain kernels only stand in when no split-K tile fits the LDS next to this launch's epilogue operands)
            try:
                launch(code)
            except RuntimeError as err:       # status -4 only: the epilogue's operand tiles of this tile shape do not fit the LDS; anything else is a real failure
                if "status -4" not in str(err):
                    raise
                continue
            feasible.append(code)
            per = []
            for _ in range(TUNE_REPS):
                if core.TUNE_COLD:       # inside a step every conv runs once, on operands the caches have mostly lost: time it that way
                    _thrash()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(code)
                e1.record()
                per.append((e0, e1))
            evs.append(per)
        torch.cuda.synchronize()
        times = [min(a.elapsed_time(b) for a, b in per) for per in evs]
        if not feasible:
            raise RuntimeError(f"no conv kernel candidate could be launched for {key}")
        best = feasible[min(range(len(feasible)), key=lambda i: times[i])]
        t[key] = best
        return best

    def _ksplit(self, M, K, Cout_p):
        """Split-K factor for a conv GEMM with M output rows and contraction K (bf16 LDS-DMA kernels only).  Measured on cold operands
        (tools/splitk_micro.py): 4 pays for K >= 4096 with M <= 4096 (5x5, 256 channels, 11x11 maps: 66 -> 49 us); shorter contractions lose
        to the partial-tile traffic."""
        if not core.SPLITK or self.dt != BF16 or K < KSPLIT_MINK or M > 4096 or Cout_p % 8:
            return 1
        return 4

    def _stat_blocks(self, M, Cout, tune):
        bm = (tune >> 2) & 3
        if bm:
            b = 64 if bm == 1 else 128
            return (M + b - 1) // b
        return call.pn2_conv_stat_blocks(M, Cout, capi.F32_MMA if self.dt == F32 else self.dt)          # (fp32fast picks its own tiles)

    def _tune_wgrad(self, wd, dy_ptr, x_ptr, rd, nsplit, wshape):
        """-> (kernel code, pixel splits) for this wgrad shape.  Candidates: register-staged / LDS-DMA / LDS-DMA with 128 x 256 tiles x
        {1, 1/2, 1/4, 1/8} of the heuristic split count; each is timed together with the slab reduction its split count implies."""
        t = self.tuner
        fast = self.dt == F32 and capi.F32_MMA == capi.F32F          # fp32fast: one kernel, but the pixel-split count is worth timing
        if t is None or (self.dt != BF16 and not fast):
            return 0, nsplit
        key = ("w", wd.N, wd.H, wd.W, wd.OH, wd.OW, wd.Cin_p, wd.ld_x, wd.Cout_p, wd.ld_dy, wd.KH, wd.KW, wd.stride, wd.pad_h, wd.pad_w, wd.dil_h, wd.dil_w, nsplit)
        if fast:
            key = key + ("f32f",)
        if key in t:
            return t[key]
        if torch.cuda.is_current_stream_capturing():
            return 0, nsplit
        from . import lockstep as LS
        with LS.pause():
            return self._tune_wgrad_run(t, key, wd, dy_ptr, x_ptr, rd, nsplit, wshape)

    def _tune_wgrad_run(self, t, key, wd, dy_ptr, x_ptr, rd, nsplit, wshape):
        st = _stream()
        slab = torch.empty((nsplit, wd.Rp, wd.Kp), dtype=torch.float32, device=self.dev)
        gw = torch.empty(tuple(wshape), dtype=torch.float32, device=self.dev)
        codes = (1, 2, 3) if (call.pn2_wgrad_tile_co(wd.Cout_p) == 128 and wd.Kp >= 256) else (1, 2)      # 3: LDS-DMA kernel with 128 x 256 tiles
        if self.dt != BF16:
            codes = (1,)
        cands = [(code, ns) for ns in sorted({max(1, nsplit >> k) for k in range(4)}, reverse=True) for code in codes]
        evs = []

        def run(code, ns):
            wd.tune = code
            call.pn2_conv_wgrad(self.dt, dy_ptr, x_ptr, _p(slab), C.byref(wd), ns, st)
            call.pn2_wgrad_reduce(_p(slab), _p(gw), C.byref(rd), ns, 0, st)
        for code, ns in cands:
            run(code, ns)
            per = []
            for _ in range(TUNE_REPS):
                if core.TUNE_COLD:       # the deferred wgrads run long after dy / x were produced: cold operands
                    _thrash()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(code, ns)
                e1.record()
                per.append((e0, e1))
            evs.append(per)
        torch.cuda.synchronize()
        times = [min(a_.elapsed_time(b_) for a_, b_ in per) for per in evs]
        best = cands[min(range(len(cands)), key=lambda i: times[i])]
        t[key] = best
        return best

    # ------------------------------------------------------------------ conv (+BN +ReLU +residual)
    def conv_bn_act(self, x, conv, bn=None, relu=False, residual=None, out=None, out_map=None, y_dt=None, y_C=None, bias=None, sum_with=None,
                    raw_out=None, par_out=None, x_last=False, gate=None, tee=None, pool=False):
        """y = act(BN(conv(x)) + residual)   — BasicConv2d / Bottle2neck pieces.

        conv: nn.Conv2d (bias-free unless `bias` given), bn: nn.BatchNorm2d or None.
        out: optional destination Act view (writes y into a slice of a concat buffer).
        out_map: (gw, gwp) group-padded layout of the produced channels (default identity).
        y_dt/y_C: fp32 K-channel head outputs (physical raw output stays padded to 8).
        sum_with: an Act of the output's geometry that has no other consumer; returns (y, y + sum_with) - the second tensor (Bottle2neck's
                  sp + spx[i+1]) is written by the same pass and keeps its gradient in sum_with's gradient storage.
        raw_out / par_out: where the raw conv output ([N,OH,OW,Cout_p] view) and the BatchNorm's per-channel rows (scale, shift, mean, invstd:
                  a [4][Cout_p] view) go - channel slices of buffers shared by the convs that write one concat buffer (Engine.concat_bnb).
        gate:     a 1-channel fp32 Act of x's geometry in front of a 1x1 conv: the conv sees (1 - sigmoid(gate)) * x (V1 reverse attention,
                  PraNet_Res2Net.py:153-155).  The per-pixel factor is applied to the GEMM's accumulator rows (pn2_conv_gemm_gated) - the gated copy
                  of x is never written; backward: one pass over (raw, dz) of the conv's OUTPUT width, then plain dgrad / wgrad.
        x_last:   the caller guarantees that this conv's data gradient is the LAST contribution to x's gradient (x's first consumer in forward
                  order).  If x is the output of a train-mode BatchNorm, the dgrad GEMM then takes that BatchNorm's backward statistics in its
                  epilogue (pn2_conv_gemm_ep) and x's producer skips its pn2_bn_bwd_reduce pass.
        """
        assert tee is None or (self.training and bn is not None and bias is None and gate is None and sum_with is None), "tee: train-mode conv + BatchNorm outputs only"
        # pool: the op is conv -> BatchNorm -> ReLU -> MaxPool2d(3, 2, 1) and returns the POOLED activation; the full-resolution BatchNorm output is never written
        # (pn2_bn_relu_maxpool_fwd; backward: pn2_maxpool3x3s2_bwd, then the BatchNorm passes with the mask recomputed from raw).  Train-mode BatchNorm with a plain ReLU only.
        assert not pool or (self.training and bn is not None and relu is True and bias is None and gate is None and sum_with is None and residual is None and out is None
                            and tee is None and y_C is None and y_dt in (None, self.dt)), "pool: train-mode conv + BatchNorm + ReLU only"
        w = conv.weight
        Cout, Cin, KH, KW = _w4(w)
        sh, sw = conv.stride
        assert sh == sw and conv.groups == 1
        ph, pw = conv.padding
        dh, dw = conv.dilation
        assert x.C == Cin, (x.C, Cin)
        N, H, W = x.N, x.H, x.W
        OH = (H + 2 * ph - dh * (KH - 1) - 1) // sh + 1
        OW = (W + 2 * pw - dw * (KW - 1) - 1) // sw + 1
        gw_o, gwp_o = out_map if out_map is not None else (Cout, rup(Cout, 8))
        Cout_p = (Cout + gw_o - 1) // gw_o * gwp_o
        x_map = (x.gw, x.gwp, x.Cp)
        o_map = (gw_o, gwp_o, Cout_p)
        M = N * OH * OW
        st = _stream()
        train_bn = bn is not None and self.training
        V = 4 if self.dt == F32 else 8

        wp, pd = self.pack(w, x_map, o_map, False)
        # biased conv / nn.Linear with nothing behind it (no BN, activation, residual or re-layout): the bias goes into the GEMM epilogue and
        # the GEMM writes the output itself - no separate affine pass
        fuse_bias = (core.FUSE_BIAS and bn is None and bias is not None and not relu and residual is None and out is None and y_C is None
                     and (y_dt is None or y_dt == self.dt))
        if fuse_bias:
            out = Act(self, self.empty(N, OH, OW, Cout_p), Cout, gw_o, gwp_o, self.dt)
            raw = out.t
            bvec = self._padded_bias(bias, Cout, Cout_p)
        elif raw_out is not None:
            assert tuple(raw_out.shape) == (N, OH, OW, Cout_p) and raw_out.dtype == self.tdt and raw_out.stride(2) % V == 0
            raw = raw_out
        else:
            raw = self.empty(N, OH, OW, Cout_p)
        raw_ld = raw.stride(2)
        cd = capi.ConvDesc()
        cd.N, cd.H, cd.W, cd.OH, cd.OW = N, H, W, OH, OW
        cd.Cin_p, cd.ld_in, cd.Cout, cd.ld_out = x.Cp, x.ld, Cout_p, raw_ld
        cd.KH, cd.KW, cd.stride, cd.pad_h, cd.pad_w, cd.dil_h, cd.dil_w = KH, KW, sh, ph, pw, dh, dw
        cd.transposed, cd.Kp, cd.flags = 0, pd.Kp, (capi.CONV_STATS if train_bn else 0)
        psum = psq = None
        tune = self._tune_gemm(cd, x.ptr, wp, M, Cout_p)
        cd.flags |= tune << 8
        if (core.EVAL_FUSE and bn is not None and not self.training and not self.need_grad and gate is None and sum_with is None and y_C is None and not fuse_bias
                and (y_dt is None or y_dt == self.dt) and self._ksplit(M, KH * KW * x.Cp, Cout_p) == 1 and relu in (False, True, 2)
                and (residual is None or (residual.dt == self.dt and residual.ld % V == 0 and Cout_p % V == 0 and (out is None or out.ld % V == 0)))):
            # eval mode (MyTest_med.py:98-104, the in-training evaluation): the folded BatchNorm, the activation and the residual add ride in the GEMM epilogue -
            # no raw conv output, no separate normalise pass (half the launches and half the activation traffic of the layer)
            scale, shift = self._bn_eval_rows(bn, M, Cout_p, Cout, gw_o, gwp_o, bias)
            if out is None:
                out = Act(self, self.empty(N, OH, OW, Cout_p), Cout, gw_o, gwp_o, self.dt)
            if residual is not None:
                assert residual.Cp == Cout_p
            cd.ld_out = out.ld
            cd.flags = (tune << 8) | capi.CONV_AFFINE | (capi.CONV_RELU6 if relu == 2 else (capi.CONV_RELU if relu else 0))
            capi.WORK.update(flops=2 * M * Cout * Cin * KH * KW, tag=":fwd", shape=f"{Cin}->{Cout} k{KH}x{KW} s{sh} d{dh} {N}x{OH}x{OW}")
            call.pn2_conv_gemm_affine(self.dt, x.ptr, _p(wp), out.ptr, _p(scale), _p(shift), residual.ptr if residual is not None else C.c_void_p(0),
                                      residual.ld if residual is not None else 0, C.byref(cd), st)
            return out
        tile_rows = 0
        if train_bn:
            nblk = self._stat_blocks(M, Cout_p, tune)
            tile_rows = self._tile_m(M, Cout_p, tune)
            psum, psq = self.fbuf(nblk, Cout_p), self.fbuf(nblk, Cout_p)
        flops = 2 * M * Cout * Cin * KH * KW
        shape = f"{Cin}->{Cout} k{KH}x{KW} s{sh} d{dh} {N}x{OH}x{OW}"
        capi.WORK.update(flops=flops, tag=":fwd", shape=shape)
        ksplit = self._ksplit(M, KH * KW * x.Cp, Cout_p)
        if tune & 0xC0:
            ksplit = 1          # the tuner found the intra-workgroup split-K kernel faster than every plain tile: it also replaces the global split-K + reduce pair
        if gate is not None:
            assert (KH, KW, sh) == (1, 1, 1) and gate.dt == F32 and gate.C == 1 and gate.M == M and bias is None, "the fused gate sits in front of a bias-free 1x1 conv"
            ksplit = 1
            call.pn2_conv_gemm_gated(self.dt, x.ptr, _p(wp), _p(raw), _p(psum), _p(psq), C.byref(cd), gate.ptr, st)
        elif ksplit > 1:
            # few output rows, long contraction (the 5x5 convs of the ra4 branch on 11x11 maps): the K loop of every tile is shared by ksplit
            # workgroups that leave fp32 partial tiles; the reduce sums them and takes the BatchNorm statistics / adds the bias
            ws = self.fbuf(ksplit, M, Cout_p)
            cd.flags = ((2 | (1 << 2) | ((3 if Cout_p > 64 else 2) << 4)) << 8) | (ksplit << 16)
            if train_bn:
                nblk, tile_rows = (M + 63) // 64, 0          # the reduce leaves raw moments of 64-row blocks
                psum, psq = self.fbuf(nblk, Cout_p), self.fbuf(nblk, Cout_p)
            call.pn2_conv_gemm(self.dt, x.ptr, _p(wp), _p(raw), _p(ws), _p(None), C.byref(cd), st)
            call.pn2_conv_splitk_reduce(self.dt, _p(ws), ksplit, M, Cout_p, _p(raw), raw_ld, _p(bvec) if fuse_bias else _p(None),
                                        _p(psum) if train_bn else _p(None), _p(psq) if train_bn else _p(None), 0, st)
        elif fuse_bias:
            cd.flags |= capi.CONV_BIAS
            call.pn2_conv_gemm(self.dt, x.ptr, _p(wp), _p(raw), _p(bvec), _p(None), C.byref(cd), st)
        else:
            call.pn2_conv_gemm(self.dt, x.ptr, _p(wp), _p(raw), _p(psum), _p(psq), C.byref(cd), st)

        scale = shift = mean = invstd = par = None
        bd = None
        if bn is not None:
            bd = capi.BnDesc()
            bd.M, bd.Cp, bd.C, bd.gw, bd.gwp, bd.eps, bd.momentum = M, Cout_p, Cout, gw_o, gwp_o, bn.eps, (bn.momentum if bn.momentum is not None else 0.1)
            bd.tile_rows = tile_rows
            par = par_out if par_out is not None else self.fbuf(4, Cout_p)          # rows: scale, shift, mean, invstd
            assert tuple(par.shape) == (4, Cout_p) and par.stride(1) == 1
            scale, shift = par[0], par[1]
            if train_bn:
                mean, invstd = par[2], par[3]
                call.pn2_bn_finalize(_p(psum), _p(psq), nblk, C.byref(bd), _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var),
                                     _p(scale), _p(shift), _p(mean), _p(invstd), st)
                self.bn_modules.append(bn)
                if bias is not None:          # biased conv followed by train-mode BN: the output is unchanged, only the running mean sees the bias
                    with torch.no_grad():
                        bn.running_mean.add_(bias.detach(), alpha=bd.momentum)
            else:
                call.pn2_bn_eval_prepare(C.byref(bd), _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var), _p(scale), _p(shift), st)
                if bias is not None:
                    shift[:Cout] += bias.detach() * scale[:Cout]
        elif bias is not None:
            shift = self._padded_bias(bias, Cout, Cout_p)

        y_dt = self.dt if y_dt is None else y_dt
        pidx = None
        if pool:
            PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
            out = Act(self, self.empty(N, PH, PW, Cout_p), Cout, gw_o, gwp_o, self.dt)
            pidx = self.alloc((N, PH, PW, Cout_p), torch.uint8)
            call.pn2_bn_relu_maxpool_fwd(self.dt, _p(raw), raw_ld, _p(scale), _p(shift), out.ptr, out.ld, _p(pidx), N, OH, OW, Cout_p, PH, PW, st)
        if out is None:
            if y_C is not None:
                out = Act(self, self.empty(N, OH, OW, y_C, y_dt), y_C, y_C, y_C, y_dt)
            else:
                out = Act(self, self.empty(N, OH, OW, Cout_p, y_dt), Cout, gw_o, gwp_o, y_dt)
        ncopy = y_C if y_C is not None else Cout_p
        if residual is not None:
            assert residual.Cp == Cout_p and residual.dt == self.dt
        y2 = None
        if sum_with is not None and (fuse_bias or residual is not None or y_dt != self.dt or ncopy != Cout_p or sum_with.dt != self.dt
                                     or (sum_with.N, sum_with.H, sum_with.W, sum_with.Cp) != (N, OH, OW, Cout_p) or out.ld % 8 or sum_with.ld % 8):
            raise RuntimeError("sum_with needs a plain same-dtype BN/activation output of the same geometry")
        if sum_with is not None:
            y2 = Act(self, self.empty(N, OH, OW, Cout_p), Cout, gw_o, gwp_o, self.dt)
            call.pn2_affine_act_sum(self.dt, _p(raw), raw_ld, out.ptr, out.ld, M, Cout_p, _p(scale), _p(shift), (2 if relu == 2 else 1) if relu else 0,
                                    sum_with.ptr, sum_with.ld, y2.ptr, y2.ld, st)
            if self.need_grad and sum_with.requires_grad:
                y2.galias = sum_with            # d(y + s)/ds = 1 and s has no other consumer: the sum's gradient lives in s's gradient storage
                y2.sum_of = (out, sum_with)
        elif pool:
            pass                      # (written above, pooled)
        elif not fuse_bias:
            # tee = (Act, c_lo): the output channels >= c_lo also go to that activation (Bottle2neck: spx[3] lands in the concat buffer, no copy launch)
            if tee is not None and residual is None and y_dt == self.dt and ncopy == Cout_p and core.TEE_CONCAT and tee[0].dt == self.dt and not (tee[1] % V or tee[0].ld % V or out.ld % V or raw_ld % V or Cout_p % V):
                call.pn2_affine_act_tee(self.dt, _p(raw), raw_ld, out.ptr, out.ld, M, Cout_p, _p(scale), _p(shift), (2 if relu == 2 else 1) if relu else 0,
                                        tee[0].ptr, tee[0].ld, tee[1], st)
                tee = None
            else:
                call.pn2_affine_act(self.dt, _p(raw), raw_ld, y_dt, out.ptr, out.ld, M, ncopy, _p(scale), _p(shift),
                                    residual.ptr if residual is not None else C.c_void_p(0), residual.ld if residual is not None else 0, (2 if relu == 2 else 1) if relu else 0, st)
        if tee is not None:          # (a path without the second output: the slice is copied)
            call.pn2_copy(out.dt, _p(out.t[..., tee[1]:tee[1] + tee[0].Cp]), out.ld, tee[0].dt, tee[0].ptr, tee[0].ld, M, tee[0].Cp, 0, st)

        if not self.need_grad:
            return out if y2 is None else (out, y2)
        # the BatchNorm-backward statistics of this output's gradient can be taken by the dgrad GEMM that completes it (x_last of the consumer)
        bnb_ok = core.BNB_EPILOGUE and train_bn and not fuse_bias and y_C is None and y_dt == self.dt and relu in (False, True) and out.ld % V == 0 and Cout_p % V == 0 and not pool
        if bnb_ok:
            out.bnb = Bnb(raw, par, bool(relu), out.t if residual is not None else None)
            if (residual is not None and relu and residual.bnb is not None and not residual.bnb.relu and residual.bnb.ymask is None and residual.bnb.split == 0
                    and residual.parent is None and residual.Cp == Cout_p):
                out.bnb.res = residual          # (only as this op's residual operand: Bottle2neck's downsample branch)

        def bwd():
            st = _stream()
            if out.pool_prior is not None:
                self.flush_pool_prior(out)          # (no dgrad epilogue took the deferred pool backward of this output: apply it before the gradient is consumed)
            if y2 is not None and y2.grad_written and not y2.dual_done:          # the sum's gradient also flows into y (the other operand holds it already)
                if y2.galias is not None:
                    assert not sum_with._written, "sum_with: the aliased operand received another gradient"
                    sum_with.grad_written = True
                g2 = y2.grad_buf()
                go, oacc = out.grad_sink()
                call.pn2_copy(self.dt, _p(g2), g2.stride(2), self.dt, _p(go), go.stride(2), M, Cout_p, oacc, st)
                if y2.galias is None and sum_with.requires_grad:
                    gs, sacc = sum_with.grad_sink()
                    call.pn2_copy(self.dt, _p(g2), g2.stride(2), self.dt, _p(gs), gs.stride(2), M, Cout_p, sacc, st)
            dy = out.grad_buf()
            assert out.grad_written or out.child_written, "conv output never received a gradient"
            cvq = Cout_p // V
            quad = (pool and core.POOL_BWD_QUAD and OH % 2 == 0 and OW % 2 == 0 and cvq <= 256 and cvq & (cvq - 1) == 0 and dy.stride(2) % V == 0
                    and N * (OH // 2) * (OW // 2) < (1 << 24) - (1 << 21))
            if pool and not quad:          # the gradient of the (never stored) full-resolution BatchNorm output, as MaxPool2d's backward leaves it
                dyf = self.empty(N, OH, OW, Cout_p)
                call.pn2_maxpool3x3s2_bwd(self.dt, _p(dy), dy.stride(2), _p(pidx), _p(dyf), Cout_p, N, OH, OW, Cout_p, out.H, out.W, st)
                dy = dyf
            draw = self.empty(N, OH, OW, Cout_p)
            Cdy = ncopy
            ymask = out if relu else None
            r6 = 1 if relu == 2 else 0          # relu: False / True (ReLU) / 2 (ReLU6: the mask also drops the saturated y == 6)
            msc = msh = None
            if (bnb_ok or pool) and relu and residual is None and dy.stride(2) % V == 0:
                # ReLU mask recomputed from the raw conv output (fmaf(x, scale, shift) > 0, bit-identical to the forward):
                # the backward passes then do not read y at all
                ymask, msc, msh = None, scale, shift
            nul = C.c_void_p(0)
            if train_bn:
                coef = self.fbuf(3 * Cout_p)
                gg, ga = self.pgrads.sink(bn.weight)
                gb, gba = self.pgrads.sink(bn.bias)
                assert ga == gba
                segs = out.find_bstats() if (bnb_ok and dy.stride(2) % V == 0) else None
                if quad:
                    # both BatchNorm passes form the incoming gradient per 2 x 2 input quad from the pooled gradient + argmax bytes: no pool-backward launch, no 176 x 176 gradient tensor
                    nb = 512          # workgroups = partial rows of the reduce (flat from 512 to 4096)
                    p1, p2 = self.fbuf(nb, Cout_p), self.fbuf(nb, Cout_p)
                    call.pn2_pool_bn_bwd_reduce(self.dt, _p(dy), dy.stride(2), _p(pidx), _p(raw), raw_ld, N, OH, OW, Cout_p, out.H, out.W,
                                                _p(mean), _p(invstd), _p(msc), _p(msh), _p(p1), _p(p2), nb, st)
                    call.pn2_bn_bwd_finalize(_p(p1), _p(p2), nb, C.byref(bd), _p(bn.weight), _p(invstd), _p(gg), _p(gb), ga, _p(coef), st)
                elif segs is not None and len(segs) <= 4 and any(s_[2] is not None for s_ in segs):
                    # (part of) the statistics were left by dgrad epilogues; channel ranges nobody covered get a reduce pass of their own
                    sg = capi.BnSegs()
                    sg.nseg = len(segs)
                    for k_, (c0, nc, p1, p2, nb_, ldp) in enumerate(segs):
                        if p1 is None:
                            nb_ = call.pn2_bn_bwd_blocks(M, nc, self.dt)
                            p1, p2, ldp = self.fbuf(nb_, nc), self.fbuf(nb_, nc), nc
                            ym = ymask.t[..., c0:c0 + nc] if ymask is not None else None
                            call.pn2_bn_bwd_reduce(self.dt, out.dt, _p(dy[..., c0:c0 + nc]), dy.stride(2), nc, _p(ym), ym.stride(2) if ym is not None else 0, self.dt,
                                                   _p(raw[..., c0:c0 + nc]), raw_ld, M, nc, _p(mean[c0:]), _p(invstd[c0:]), _p(p1), _p(p2), nb_,
                                                   _p(msc[c0:]) if msc is not None else nul, _p(msh[c0:]) if msc is not None else nul, r6, st)
                        sg.c0[k_], sg.nblk[k_], sg.ldp[k_], sg.p1[k_], sg.p2[k_] = c0, nb_, ldp, p1.data_ptr(), p2.data_ptr()
                        self._keep.append((p1, p2))
                    call.pn2_bn_bwd_finalize_seg(C.byref(sg), C.byref(bd), _p(bn.weight), _p(invstd), _p(gg), _p(gb), ga, _p(coef), st)
                else:
                    nb = call.pn2_bn_bwd_blocks(M, Cout_p, self.dt)
                    p1, p2 = self.fbuf(nb, Cout_p), self.fbuf(nb, Cout_p)
                    call.pn2_bn_bwd_reduce(self.dt, out.dt, _p(dy), dy.stride(2), Cdy, ymask.ptr if ymask else nul, ymask.ld if ymask else 0, self.dt,
                                           _p(raw), raw_ld, M, Cout_p, _p(mean), _p(invstd), _p(p1), _p(p2), nb, _p(msc), _p(msh), r6, st)
                    call.pn2_bn_bwd_finalize(_p(p1), _p(p2), nb, C.byref(bd), _p(bn.weight), _p(invstd), _p(gg), _p(gb), ga, _p(coef), st)
            else:
                coef = None
                if bn is not None:
                    raise RuntimeError("backward through eval-mode BatchNorm is not supported")
                if bias is not None and out.dt == F32 and self.dt != F32 or (bias is not None and Cdy != Cout_p):
                    gb, gba = self.pgrads.sink(bias)          # fp32 K-channel head maps: sum the fp32 gradient itself
                    call.pn2_bias_grad(_p(dy), M, Cdy, _p(gb), gba, st)
                    bias_done = True
                else:
                    bias_done = bias is None
            rg, racc = (None, 0)
            if out.grad_masked:
                ymask = None                  # dy already carries the ReLU mask (PN2_BNB_STORE_MASKED)
            if residual is not None and residual.requires_grad:
                if (out.grad_masked and residual.grad is None and residual.galias is None and residual.parent is None and not residual.grad_written
                        and tuple(dy.shape) == tuple(residual.t.shape) and dy.dtype == residual.t.dtype and dy.is_contiguous()):
                    # d(out)/d(residual) = the ReLU mask: the masked dy IS the residual's gradient - share the buffer (it is dead here once dz
                    # has been formed; later contributions to the residual's gradient accumulate into it in place)
                    residual.grad = dy
                    residual.grad_written = True
                else:
                    rg, racc = residual.grad_sink()
            if quad:
                call.pn2_pool_bn_bwd_apply(self.dt, _p(dy), dy.stride(2), _p(pidx), _p(raw), raw_ld, N, OH, OW, Cout_p, out.H, out.W,
                                           _p(mean), _p(invstd), _p(coef), _p(msc), _p(msh), _p(draw), Cout_p, st)
            elif coef is None and ymask is None and rg is None and out.dt == self.dt and Cdy == Cout_p and dy.stride(2) == Cout_p and dy.is_contiguous():
                draw = dy               # no BN, no activation, no residual (nn.Linear / biased conv): dz IS dy - no copy pass
            else:
                call.pn2_bn_bwd_apply(self.dt, out.dt, _p(dy), dy.stride(2), Cdy, ymask.ptr if ymask else nul, ymask.ld if ymask else 0, self.dt,
                                      _p(raw), raw_ld, M, Cout_p, _p(mean), _p(invstd), _p(coef), _p(draw), Cout_p,
                                      _p(rg), rg.stride(2) if rg is not None else 0, racc, _p(msc), _p(msh), r6, st)
            if gate is not None:
                # dz is the gradient of the GATED GEMM result: dzg = (1 - s) * dz feeds dgrad / wgrad, d gate[m] = -s * sum_c raw[m][c] * dz[m][c]
                gc_, cacc = gate.grad_sink()
                dc = self.fbuf(M) if cacc else gc_
                dzg = draw if draw is not dy else self.alloc((N, OH, OW, Cout_p), self.tdt)
                call.pn2_ra_gate_post_bwd(self.dt, _p(raw), raw_ld, gate.ptr, _p(draw), Cout_p, _p(dzg), Cout_p, _p(dc), M, Cout_p, st)
                if cacc:
                    call.pn2_copy(F32, _p(dc), 1, F32, _p(gc_), 1, M, 1, 1, st)
                draw = dzg
            if train_bn:
                bias_done = bias is None
            if not bias_done:                                  # biased conv / nn.Linear: db = column sums of dz (~0 under a train-mode BN)
                gb, gba = self.pgrads.sink(bias)
                cp = getattr(out, "colparts", None)
                if cp is not None and draw is dy and cp[2].data_ptr() == dy.data_ptr() and tuple(cp[0].shape) == (cp[1], Cout_p):
                    self.colsum_finalize(cp[0], cp[1], Cout, Cout_p, gb, gba)          # (the consumer's data-gradient kernel left the column sums of this very buffer)
                else:
                    self.colsum(draw, M, Cout_p, Cout, gb, gba)
            # ---- weight gradient
            wd = capi.WgradDesc()
            wd.N, wd.H, wd.W, wd.OH, wd.OW = N, H, W, OH, OW
            wd.Cin_p, wd.ld_x, wd.Cout_p, wd.ld_dy = x.Cp, x.ld, Cout_p, Cout_p
            wd.KH, wd.KW, wd.stride, wd.pad_h, wd.pad_w, wd.dil_h, wd.dil_w = KH, KW, sh, ph, pw, dh, dw
            tco = call.pn2_wgrad_tile_co(Cout_p)
            wd.Rp, wd.Kp = rup(Cout_p, tco), pd.Kp
            tiles = (wd.Rp // tco) * (pd.Kp // 128)
            steps = (M + 31) // 32
            # pixel splits: enough workgroups to fill 256 CUs twice, >= 4 steps each, slabs capped at 24 MB
            nsplit = max(1, min(steps // 4 if steps >= 8 else 1, (WGRAD_WGS + tiles - 1) // tiles, (WGRAD_SLAB_MB << 20) // (wd.Rp * wd.Kp * 4) or 1))
            rd = self._pack_desc(w, x_map, o_map, False)
            rd.Rp = wd.Rp
            wd.tune, nsplit = self._tune_wgrad(wd, _p(draw), x.ptr, rd, nsplit, w.shape)
            rq = self.grad_queue
            if rq is not None and rq.defer_wgrad:
                nsplit = rq.table_splits(nsplit, M, x.Cp + Cout_p, wd, 4 if self.dt == F32 else 2)
            slab = self.fbuf(nsplit, wd.Rp, wd.Kp) if rq is None else rq.slab((id(w), x.M), (nsplit, wd.Rp, wd.Kp), self.dev)
            gwt, gwa = self.pgrads.sink(w)
            # wgrad (+ slab reduce) only feeds the parameter gradient: with a gradient queue both are deferred into the table-driven launches of its flush
            if rq is not None and rq.defer_wgrad:
                rq.add_wgrad(self.dt, draw, x.ptr, x.t, slab, wd, nsplit, flops)
                rq.add_reduce(slab, gwt, rd, nsplit, gwa)
            else:
                capi.WORK.update(flops=flops, tag="", shape=shape)
                call.pn2_conv_wgrad(self.dt, _p(draw), x.ptr, _p(slab), C.byref(wd), nsplit, st)
                if rq is None:
                    call.pn2_wgrad_reduce(_p(slab), _p(gwt), C.byref(rd), nsplit, gwa, st)
                else:
                    rq.add_reduce(slab, gwt, rd, nsplit, gwa)
            # ---- data gradient
            if x.requires_grad and core.PATCH_DGRAD and KH == sh and KW == sw and KH > 1 and ph == 0 and pw == 0 and dh == 1 and dw == 1 \
                    and x.gw == x.gwp and gw_o == gwp_o and x.ld == x.Cp:
                # patchify conv (kernel == stride): every input pixel sees exactly one tap -> GEMM over the patches + depth-to-space
                gx, gxa = x.grad_sink()
                taps, Ct = KH * KW, KH * KW * x.Cp
                Rp2, Kp2 = rup(Ct, 128), rup(Cout_p, 128)
                wp2 = self.alloc((Rp2, Kp2), self.tdt)
                call.pn2_pack_patch_weight(self.dt, _p(w), _p(wp2), Cout, Cin, KH, KW, x.Cp, Rp2, Kp2, st)
                tpatch = self.empty(N, OH, OW, Ct)
                dd = capi.ConvDesc()
                dd.N, dd.H, dd.W, dd.OH, dd.OW = N, OH, OW, OH, OW
                dd.Cin_p, dd.ld_in, dd.Cout, dd.ld_out = Cout_p, Cout_p, Ct, Ct
                dd.KH, dd.KW, dd.stride, dd.pad_h, dd.pad_w, dd.dil_h, dd.dil_w = 1, 1, 1, 0, 0, 1, 1
                dd.transposed, dd.Kp, dd.flags = 0, Kp2, 0
                dd.flags |= self._tune_gemm(dd, _p(draw), wp2, M, Ct) << 8
                capi.WORK.update(flops=flops, tag=":dgrad", shape=shape + " patch")
                call.pn2_conv_gemm(self.dt, _p(draw), _p(wp2), _p(tpatch), C.c_void_p(0), C.c_void_p(0), C.byref(dd), st)
                call.pn2_depth_to_space(self.dt, _p(tpatch), Ct, _p(gx), gx.stride(2), N, H, W, OH, OW, KH, x.Cp, gxa, st)
            elif x.requires_grad and core.SMALL_CIN_DGRAD and Cin <= 4 and sh > 1 and dh == 1 and dw == 1 and ph == pw and (x.gw == x.gwp or x.Cp == x.gwp) and gw_o == gwp_o \
                    and Cout_p == Cout and KH * KW * Cout * 16 <= 64 * 1024 and x.ld == x.Cp:
                # few-channel strided conv (EMCADNet's patch embedding behind the 1 -> 3 stem): only the taps that land on an output pixel
                gx, gxa = x.grad_sink()
                capi.WORK.update(flops=flops, tag=":dgrad", shape=shape + " small-cin")
                call.pn2_conv_dgrad_small_cin(self.dt, _p(draw), Cout_p, _p(w), _p(gx), gx.stride(2), N, H, W, OH, OW, Cout, Cin, KH, KW, sh, ph, gxa, st)
            elif x.requires_grad:
                wt, ptd = self.pack(w, x_map, o_map, True)
                # a deferred AvgPool2d(2, 2) backward on x (SpatialOps.avgpool(fold_bwd=True)): this dgrad completes x's gradient with a BatchNorm-backward epilogue and
                # nothing has been written to it yet -> the epilogue adds 1/4 of the pooled gradient itself (pn2_conv_ep.pool); anything else: the plain pool-backward launch
                pool_ok = (x.pool_prior is not None and core.BNB_EPILOGUE and x_last and not x.grad_written and x.bnb is not None and x.sum_of is None
                           and x.Cp % V == 0 and x.pool_prior[0].shape[3] == x.Cp and self.dt == x.dt and x.grad_buf().stride(2) % V == 0
                           and self._ksplit(N * H * W, KH * KW * Cout_p, x.Cp) == 1)
                if x.pool_prior is not None and not pool_ok:
                    self.flush_pool_prior(x)
                gx, gxa = x.grad_sink()
                Mx = N * H * W
                dd = capi.ConvDesc()
                dd.N, dd.H, dd.W, dd.OH, dd.OW = N, OH, OW, H, W
                dd.Cin_p, dd.ld_in, dd.Cout, dd.ld_out = Cout_p, Cout_p, x.Cp, gx.stride(2)
                dd.KH, dd.KW, dd.stride, dd.pad_h, dd.pad_w, dd.dil_h, dd.dil_w = KH, KW, sh, ph, pw, dh, dw
                dd.transposed, dd.Kp, dd.flags = 1, ptd.Kp, (capi.CONV_ACCUM if gxa else 0)
                ks = self._ksplit(Mx, KH * KW * Cout_p, x.Cp) if gx.stride(2) == x.Cp else 1
                dual = x.sum_of is not None and x.galias is x.sum_of[1] and x.sum_of[0].requires_grad
                want_ep = bool(core.BNB_EPILOGUE and x_last and x.Cp % V == 0 and gx.stride(2) % V == 0 and (x.bnb is not None or dual))
                if ks > 1:
                    # as in the forward: intra-workgroup split-K instead of partial tiles + reduce.  The probe asks the tuner with THE KEY THE LAUNCH WILL USE (a launch
                    # with a BatchNorm-backward epilogue is tuned with it: other LDS budget, other key); if that code is not of the split-K class (no tile of it fits
                    # next to the epilogue's operand tiles, or no tuner) the global split-K + reduce pair stays - never a long-K launch with neither
                    probe = None
                    if want_ep:
                        probe = capi.ConvEp()
                        for t_, a_ in ((probe.a, x.sum_of[0] if dual else x), (probe.b, x.sum_of[1] if dual else None)):
                            if a_ is not None and a_.bnb is not None:
                                self._fill_bnb(t_, a_, 0)
                        if dual:
                            probe.b.out = 1
                        if dual and x.sum_of[0].grad_written:
                            dd.flags |= capi.CONV_ACCUM
                    if self._tune_gemm(dd, _p(draw), wt, Mx, x.Cp, probe) & 0xC0:
                        ks = 1
                    dd.flags = capi.CONV_ACCUM if gxa else 0
                capi.WORK.update(flops=flops, tag=":dgrad", shape=shape)
                if ks > 1:
                    ws = self.fbuf(ks, Mx, x.Cp)
                    dd.flags = ((2 | (1 << 2) | ((3 if x.Cp > 64 else 2) << 4)) << 8) | (ks << 16)
                    call.pn2_conv_gemm(self.dt, _p(draw), _p(wt), _p(gx), _p(ws), C.c_void_p(0), C.byref(dd), st)
                    call.pn2_conv_splitk_reduce(self.dt, _p(ws), ks, Mx, x.Cp, _p(gx), x.Cp, C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), gxa, st)
                elif want_ep:
                    ep = capi.ConvEp()
                    if dual and x.sum_of[0].grad_written:
                        dd.flags |= capi.CONV_ACCUM
                    for t_, a_ in ((ep.a, x.sum_of[0] if dual else x), (ep.b, x.sum_of[1] if dual else None)):       # what the tuner needs to know
                        if a_ is not None and a_.bnb is not None:
                            self._fill_bnb(t_, a_, 0)
                    if dual:
                        ep.b.out = 1
                    if pool_ok:
                        assert not gxa and not dual
                        dd.flags |= capi.CONV_ACCUM
                        ep.pool, ep.ld_pool = x.pool_prior[0].data_ptr(), x.pool_prior[0].stride(2)
                    tcode = self._tune_gemm(dd, _p(draw), wt, Mx, x.Cp, ep)
                    dd.flags |= tcode << 8
                    nbx = self._stat_blocks(Mx, x.Cp, tcode)
                    ep = capi.ConvEp()
                    if pool_ok:
                        ep.pool, ep.ld_pool = x.pool_prior[0].data_ptr(), x.pool_prior[0].stride(2)
                    if dual:
                        # x = u + v (Bottle2neck's sp + spx[i]): the gradient goes to BOTH operands - accumulated into u's (the concat buffer slice
                        # conv3's dgrad wrote), stored as v's (aliased by x) - each with the statistics of its own BatchNorm
                        u, v = x.sum_of
                        gu, gua = u.grad_sink()
                        assert gu.stride(2) % V == 0
                        dd.ld_out, dd.flags = gu.stride(2), (dd.flags & ~capi.CONV_ACCUM) | (capi.CONV_ACCUM if gua else 0)
                        self._fill_bnb(ep.a, u, nbx)
                        ep.b.out, ep.b.ld_out = gx.data_ptr(), gx.stride(2)
                        self._fill_bnb(ep.b, v, nbx)
                        v.grad_written = True
                        x.dual_done = True
                        u._sealed = v._sealed = True
                        call.pn2_conv_gemm_ep(self.dt, _p(draw), _p(wt), _p(gu), C.byref(dd), C.byref(ep), st)
                    else:
                        self._fill_bnb(ep.a, x, nbx, ep, tcode)
                        call.pn2_conv_gemm_ep(self.dt, _p(draw), _p(wt), _p(gx), C.byref(dd), C.byref(ep), st)
                    if pool_ok and x.pool_prior is not None:
                        x.pool_prior = None
                        self._pending_pool.remove(x)
                    x._sealed = True
                else:
                    dd.flags |= self._tune_gemm(dd, _p(draw), wt, Mx, x.Cp) << 8
                    call.pn2_conv_gemm(self.dt, _p(draw), _p(wt), _p(gx), C.c_void_p(0), C.c_void_p(0), C.byref(dd), st)

        self.record(bwd)
        return out if y2 is None else (out, y2)

    def _padded_bias(self, bias, Cout, Cout_p):
        """[Cout_p] fp32 row: the bias in its first Cout slots, zeros behind.  With a persistent pack cache the row persists (its pad slots are written once) and a step
        refreshes it with ONE copy instead of a fill + a copy (the K-channel head convs of pranet.py:104-105,303-325: 4 launches per step fewer)."""
        if Cout_p == Cout:
            return bias.detach()
        cache = self.pack_cache
        if cache is None:
            row = torch.zeros(Cout_p, dtype=torch.float32, device=self.dev)
        else:
            rows = cache.__dict__.setdefault("bias_rows", {})
            row = rows.get((id(bias), Cout_p))
            if row is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("run an eager step before capturing (the padded bias rows are created then)")
                row = rows[(id(bias), Cout_p)] = torch.zeros(Cout_p, dtype=torch.float32, device=self.dev)
                cache.__dict__.setdefault("bias_keep", []).append(bias)          # (id(bias) stays unique while the row exists)
        row[:Cout].copy_(bias.detach())
        return row

    def _bn_eval_rows(self, bn, M, Cout_p, Cout, gw_o, gwp_o, bias=None):
        """-> (scale, shift) fp32 [Cout_p] rows of an eval-mode BatchNorm (bias of the conv in front folded into shift).  With a BnFoldCache the rows persist and
        were refreshed at the start of this forward; a layer seen for the first time is folded here and joins the cache."""
        bd = capi.BnDesc()
        bd.M, bd.Cp, bd.C, bd.gw, bd.gwp, bd.eps, bd.momentum = M, Cout_p, Cout, gw_o, gwp_o, bn.eps, (bn.momentum if bn.momentum is not None else 0.1)
        fold = self.bn_fold if bias is None else None
        key = (id(bn), Cout_p, gw_o, gwp_o)
        if fold is not None and key in fold.entries:
            par = fold.entries[key]
            return par[0], par[1]
        par = torch.empty((2, Cout_p), dtype=torch.float32, device=self.dev) if fold is not None else self.fbuf(2, Cout_p)
        call.pn2_bn_eval_prepare(C.byref(bd), _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var), _p(par[0]), _p(par[1]), _stream())
        if bias is not None:
            par[1][:Cout] += bias.detach() * par[0][:Cout]
        if fold is not None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("run an eager forward before capturing (the BatchNorm fold table is built then)")
            fold.add(key, bn, par, bd)
        return par[0], par[1]

    def _tile_m(self, M, Cout, tune):
        bm = (tune >> 2) & 3
        return (64 if bm == 1 else 128) if bm else call.pn2_conv_tile_m(M, Cout, capi.F32_MMA if self.dt == F32 else self.dt)

    def _fill_bnb(self, t, act, nblk, ep=None, tcode=0):
        """Describe `act`'s BatchNorm to a dgrad epilogue target and register the partial rows it will leave."""
        b = act.bnb
        if b is None:
            t.mode = 0
            return
        t.mode = capi.BNB_STATS | (capi.BNB_MASK_Y if b.ymask is not None else (capi.BNB_MASK_RAW if b.relu else 0))
        if b.ymask is not None and core.MASKED_STORE and nblk and b.split == 0 and act.parent is None:
            # BN + residual + ReLU: the masked gradient is also the residual branch's gradient - store it masked, the producer aliases it
            t.mode |= capi.BNB_STORE_MASKED
            act.grad_masked = True
        t.raw, t.ld_raw = b.raw.data_ptr(), b.raw.stride(2)
        if b.ymask is not None:
            t.y, t.ld_y = b.ymask.data_ptr(), b.ymask.stride(2)
        t.par, t.ps = b.par.data_ptr(), b.par.stride(0)
        Cp = act.Cp
        if b.split:
            t.split = b.split
            if b.par2 is not None:
                t.raw2, t.par2 = b.raw2.data_ptr(), b.par2.data_ptr()
        if nblk == 0:               # description only (the tuner supplies its own partial rows)
            return
        p1, p2 = self.fbuf(nblk, Cp), self.fbuf(nblk, Cp)
        t.p1, t.p2, t.ldp = p1.data_ptr(), p2.data_ptr(), Cp
        bm_, bn_ = (tcode >> 2) & 3, (tcode >> 4) & 3
        if (ep is not None and core.RES_STATS and b.res is not None and (t.mode & capi.BNB_STORE_MASKED) and self.dt == BF16 and bm_ and bn_
                and (64 << (bm_ - 1)) * (16 << bn_) <= 4096 and not b.res.grad_written and b.res.galias is None and b.res.grad is None):
            # the residual operand is a BatchNorm output without activation whose gradient will BE the masked dz stored here: its backward sums ride in this
            # epilogue too (LDS-DMA / matrix-core form: tiles of <= 4096 elements); its producer then finds them like any other epilogue statistics
            rb = b.res.bnb
            c = ep.c
            c.mode, c.raw, c.ld_raw, c.par, c.ps = capi.BNB_STATS, rb.raw.data_ptr(), rb.raw.stride(2), rb.par.data_ptr(), rb.par.stride(0)
            q1, q2 = self.fbuf(nblk, Cp), self.fbuf(nblk, Cp)
            c.p1, c.p2, c.ldp = q1.data_ptr(), q2.data_ptr(), Cp
            b.res.add_bstats(0, Cp, q1, q2, nblk, Cp)
        if b.split:
            t.split = b.split
            if b.par2 is not None:
                assert b.raw2.stride(2) == b.raw.stride(2) and b.par2.stride(0) == b.par.stride(0)
                t.raw2, t.par2 = b.raw2.data_ptr(), b.par2.data_ptr()
                b.tail.add_bstats(0, Cp - b.split, p1[:, b.split:], p2[:, b.split:], nblk, Cp)
            act.add_bstats(0, b.split, p1, p2, nblk, Cp)
        else:
            act.add_bstats(0, Cp, p1, p2, nblk, Cp)

    def concat_bnb(self, cat, raw, par, split=0, tail=None):
        """Declare that the channels [0, split or all) of the concat buffer `cat` were written by train-mode conv+BN(+ReLU) ops whose raw outputs /
        parameter rows sit in the matching channel slices of `raw` / `par` (conv_bn_act(raw_out=, par_out=)); channels >= split are a copy of the
        BN+ReLU output `tail` (None: they carry no BatchNorm).  The dgrad that completes cat's gradient can then take all those BatchNorms'
        backward statistics in one epilogue."""
        if not (core.BNB_EPILOGUE and self.need_grad and self.training):
            return
        tb = tail.bnb if tail is not None else None
        if tail is not None and (tb is None or not tb.relu or tb.ymask is not None):
            return
        r2 = p2 = None
        if tb is not None:
            # raw2 / par2 are indexed with cat's local column: shift the tail's views back by `split` columns
            tr, off = tail.root()
            rb = tr.bnb
            if rb is None or off != split or rb.raw.stride(2) != raw.stride(2) or rb.par.stride(0) != par.stride(0):
                return
            r2, p2 = rb.raw, rb.par
        cat.bnb = Bnb(raw, par, True, None, split, r2, p2, tail)

    # ------------------------------------------------------------------ fused 1x1 reducers sharing one input
    def conv_bn_multi(self, x, mods):
        """[BN_j(conv_j(x)) for j] for bias-free 1x1 / stride-1 BasicConv2d-style modules `mods` (each has .conv, .bn; no ReLU).

        The RFB branches, conv_res and the RA stage's conv1 all read the same encoder map (pranet.py:52,55,61,67,73,303,312,320):
        their weights are packed side by side into ONE panel so the map is read once in forward, dx is written once in dgrad
        (instead of J read-modify-write passes) and wgrad reads it once.  BatchNorm stays per module (own gamma/beta/running stats).
        Returns the channel-slice views of the fused [M][sum Cout] output."""
        convs = [m.conv for m in mods]
        for c in convs:
            assert c.kernel_size == (1, 1) and c.stride == (1, 1) and c.padding == (0, 0) and c.bias is None and c.in_channels == x.C
        couts = [c.out_channels for c in convs]
        assert all(co % 8 == 0 for co in couts)
        offs = [sum(couts[:j]) for j in range(len(couts))]
        Ct = sum(couts)
        N, H, W = x.N, x.H, x.W
        M = N * H * W
        st = _stream()
        train = self.training
        x_map = (x.gw, x.gwp, x.Cp)
        Kp = rup(x.Cp, 128)
        Rp = rup(Ct, 128)
        Rt, Kt = rup(x.Cp, 128), rup(Ct, 128)          # transposed (dgrad) panel

        def panel(transposed):
            cache = self.pack_cache
            key = ("multi", tuple(id(c.weight) for c in convs), transposed, x_map, self.dt)
            if cache is not None and key in cache.entries:
                return cache.entries[key][0]
            wp = torch.zeros((Rt, Kt) if transposed else (Rp, Kp), dtype=self.tdt, device=self.dev)
            for c, co, off in zip(convs, couts, offs):
                d = self._pack_desc(c.weight, x_map, (co, co, co), transposed)
                if transposed:
                    d.Rp, d.Kp, d.ld, d.koff = Rt, co, Kt, off          # columns [off, off+co) of every row
                    dst = wp
                else:
                    d.Rp, d.Kp = co, Kp                                  # rows [off, off+co)
                    dst = wp[off:]
                call.pn2_pack_weight(self.dt, _p(c.weight), _p(dst), C.byref(d), st)
                if cache is not None:
                    cache.add(key + (off,), c.weight, dst, d)
            if cache is not None:
                cache.entries[key] = (wp, None)
            return wp

        wp = panel(False)
        cd = capi.ConvDesc()
        cd.N, cd.H, cd.W, cd.OH, cd.OW = N, H, W, H, W
        cd.Cin_p, cd.ld_in, cd.Cout, cd.ld_out = x.Cp, x.ld, Ct, Ct
        cd.KH, cd.KW, cd.stride, cd.pad_h, cd.pad_w, cd.dil_h, cd.dil_w = 1, 1, 1, 0, 0, 1, 1
        cd.transposed, cd.Kp, cd.flags = 0, Kp, (capi.CONV_STATS if train else 0)
        psum = psq = None
        tune = self._tune_gemm(cd, x.ptr, wp, M, Ct)
        cd.flags |= tune << 8
        tile_rows = 0
        if train:
            nblk = self._stat_blocks(M, Ct, tune)
            tile_rows = self._tile_m(M, Ct, tune)
            psum, psq = self.fbuf(nblk, Ct), self.fbuf(nblk, Ct)
        flops = 2 * M * Ct * x.C
        shape = f"{x.C}->{'+'.join(map(str, couts))} k1x1 s1 d1 {N}x{H}x{W}"
        capi.WORK.update(flops=flops, tag=":fwd", shape=shape)
        if core.EVAL_FUSE and not train and not self.need_grad:
            # eval mode: the folded BatchNorms of all the reducers ride in the GEMM epilogue (pn2_conv_gemm_affine) - no raw output, no normalise pass
            fold = self.bn_fold
            key = ("multi",) + tuple(id(m.bn) for m in mods)
            par = fold.entries.get(key) if fold is not None else None
            if par is None:
                par = torch.empty((2, Ct), dtype=torch.float32, device=self.dev) if fold is not None else self.fbuf(2, Ct)
                for m, co, off in zip(mods, couts, offs):
                    bn = m.bn
                    bd = capi.BnDesc()
                    bd.M, bd.Cp, bd.C, bd.gw, bd.gwp, bd.eps, bd.momentum = M, co, co, co, co, bn.eps, (bn.momentum if bn.momentum is not None else 0.1)
                    call.pn2_bn_eval_prepare(C.byref(bd), _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var), _p(par[0][off:]), _p(par[1][off:]), st)
                    if fold is not None:
                        if torch.cuda.is_current_stream_capturing():
                            raise RuntimeError("run an eager forward before capturing (the BatchNorm fold table is built then)")
                        fold.add(key, bn, par, bd, off)
            out = Act(self, self.empty(N, H, W, Ct), Ct, Ct, Ct, self.dt)
            cd.flags = (tune << 8) | capi.CONV_AFFINE
            call.pn2_conv_gemm_affine(self.dt, x.ptr, _p(wp), out.ptr, _p(par[0]), _p(par[1]), C.c_void_p(0), 0, C.byref(cd), st)
            return [out.slice(off, off + co) for co, off in zip(couts, offs)]
        raw = self.empty(N, H, W, Ct)
        call.pn2_conv_gemm(self.dt, x.ptr, _p(wp), _p(raw), _p(psum), _p(psq), C.byref(cd), st)
        scale, shift = self.fbuf(Ct), self.fbuf(Ct)
        mean, invstd = (self.fbuf(Ct), self.fbuf(Ct)) if train else (None, None)
        bds = []
        for m, co, off in zip(mods, couts, offs):
            bn = m.bn
            bd = capi.BnDesc()
            bd.M, bd.Cp, bd.C, bd.gw, bd.gwp, bd.eps, bd.momentum, bd.ldp = M, co, co, co, co, bn.eps, (bn.momentum if bn.momentum is not None else 0.1), Ct
            bd.tile_rows = tile_rows
            bds.append(bd)
            if train:
                call.pn2_bn_finalize(_p(psum[:, off:]), _p(psq[:, off:]), nblk, C.byref(bd), _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var),
                                     _p(scale[off:]), _p(shift[off:]), _p(mean[off:]), _p(invstd[off:]), st)
                self.bn_modules.append(bn)
            else:
                call.pn2_bn_eval_prepare(C.byref(bd), _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var), _p(scale[off:]), _p(shift[off:]), st)
        out = Act(self, self.empty(N, H, W, Ct), Ct, Ct, Ct, self.dt)
        call.pn2_affine_act(self.dt, _p(raw), Ct, self.dt, out.ptr, out.ld, M, Ct, _p(scale), _p(shift), C.c_void_p(0), 0, 0, st)
        outs = [out.slice(off, off + co) for co, off in zip(couts, offs)]
        if not self.need_grad:
            return outs

        def bwd():
            st = _stream()
            if not train:
                raise RuntimeError("backward through eval-mode BatchNorm is not supported")
            dy = out.grad_buf()
            assert out.grad_written or out.child_written
            nb = call.pn2_bn_bwd_blocks(M, Ct, self.dt)
            p1, p2 = self.fbuf(nb, Ct), self.fbuf(nb, Ct)
            nul = C.c_void_p(0)
            call.pn2_bn_bwd_reduce(self.dt, self.dt, _p(dy), Ct, Ct, nul, 0, self.dt, _p(raw), Ct, M, Ct, _p(mean), _p(invstd), _p(p1), _p(p2), nb, nul, nul, 0, st)
            coef = self.fbuf(3 * Ct)
            for m, bd, off in zip(mods, bds, offs):
                gg, ga = self.pgrads.sink(m.bn.weight)
                gb, gba = self.pgrads.sink(m.bn.bias)
                call.pn2_bn_bwd_finalize(_p(p1[:, off:]), _p(p2[:, off:]), nb, C.byref(bd), _p(m.bn.weight), _p(invstd[off:]), _p(gg), _p(gb), ga, _p(coef[off:]), st)
            draw = self.empty(N, H, W, Ct)
            call.pn2_bn_bwd_apply(self.dt, self.dt, _p(dy), Ct, Ct, nul, 0, self.dt, _p(raw), Ct, M, Ct, _p(mean), _p(invstd), _p(coef), _p(draw), Ct, nul, 0, 0, nul, nul, 0, st)
            wd = capi.WgradDesc()
            wd.N, wd.H, wd.W, wd.OH, wd.OW = N, H, W, H, W
            wd.Cin_p, wd.ld_x, wd.Cout_p, wd.ld_dy = x.Cp, x.ld, Ct, Ct
            wd.KH, wd.KW, wd.stride, wd.pad_h, wd.pad_w, wd.dil_h, wd.dil_w = 1, 1, 1, 0, 0, 1, 1
            tco = call.pn2_wgrad_tile_co(Ct)
            wd.Rp, wd.Kp = rup(Ct, tco), Kp
            tiles = (wd.Rp // tco) * (Kp // 128)
            steps = (M + 31) // 32
            nsplit = max(1, min(steps // 4 if steps >= 8 else 1, (WGRAD_WGS + tiles - 1) // tiles, (WGRAD_SLAB_MB << 20) // (wd.Rp * wd.Kp * 4) or 1))
            rd0 = self._pack_desc(convs[0].weight, x_map, (couts[0], couts[0], couts[0]), False)
            rd0.Rp, rd0.Kp = wd.Rp, Kp
            wd.tune, nsplit = self._tune_wgrad(wd, _p(draw), x.ptr, rd0, nsplit, convs[0].weight.shape)
            rq = self.grad_queue
            if rq is not None and rq.defer_wgrad:
                nsplit = rq.table_splits(nsplit, M, x.Cp + Ct, wd, 4 if self.dt == F32 else 2)
            slab = self.fbuf(nsplit, wd.Rp, wd.Kp) if rq is None else rq.slab(tuple(id(c.weight) for c in convs), (nsplit, wd.Rp, wd.Kp), self.dev)
            if rq is not None and rq.defer_wgrad:
                rq.add_wgrad(self.dt, draw, x.ptr, x.t, slab, wd, nsplit, flops)
            else:
                capi.WORK.update(flops=flops, tag="", shape=shape)
                call.pn2_conv_wgrad(self.dt, _p(draw), x.ptr, _p(slab), C.byref(wd), nsplit, st)
            for c, co, off in zip(convs, couts, offs):
                gwt, gwa = self.pgrads.sink(c.weight)
                rd = self._pack_desc(c.weight, x_map, (co, co, co), False)
                rd.Rp, rd.Kp = wd.Rp, Kp
                if rq is None:
                    call.pn2_wgrad_reduce(_p(slab[:, off:]), _p(gwt), C.byref(rd), nsplit, gwa, st)
                else:
                    rq.add_reduce(slab[:, off:], gwt, rd, nsplit, gwa)
            if x.requires_grad:
                wt = panel(True)
                gx, gxa = x.grad_sink()
                dd = capi.ConvDesc()
                dd.N, dd.H, dd.W, dd.OH, dd.OW = N, H, W, H, W
                dd.Cin_p, dd.ld_in, dd.Cout, dd.ld_out = Ct, Ct, x.Cp, gx.stride(2)
                dd.KH, dd.KW, dd.stride, dd.pad_h, dd.pad_w, dd.dil_h, dd.dil_w = 1, 1, 1, 0, 0, 1, 1
                dd.transposed, dd.Kp, dd.flags = 1, Kt, (capi.CONV_ACCUM if gxa else 0)
                dd.flags |= self._tune_gemm(dd, _p(draw), wt, M, x.Cp) << 8
                capi.WORK.update(flops=flops, tag=":dgrad", shape=shape)
                call.pn2_conv_gemm(self.dt, _p(draw), _p(wt), _p(gx), nul, nul, C.byref(dd), st)
        self.record(bwd)
        return outs

    # ------------------------------------------------------------------ PVTv2 encoder ops (lib/pvtv2.py)
    def colsum_finalize(self, part, nblk, Cc, ld, out, accumulate):
        """out[:Cc] (+)= sum of the nblk partial rows.  These sums only feed parameter gradients: they are queued and run as ONE
        table-driven launch per flush (end of backward / before a gradient bucket leaves), not one launch each."""
        if not core.DEFER_COLSUM or self.grad_queue is None:      # (without a persistent queue the job table would be rebuilt and uploaded every step)
            call.pn2_colsum_finalize(_p(part), nblk, Cc, ld, _p(out), accumulate, _stream())
            return
        if accumulate:                  # a second contribution to the same gradient must see the first one finished
            self.flush_colsum()
        self.cjobs.append((part.data_ptr(), out.data_ptr(), nblk, Cc, ld, accumulate))
        self.ckeep.append((part, out))  # the partial rows must not be recycled before the launch is queued

    def flush_colsum(self):
        """Run the queued column sums (one pn2_colsum_multi per dtype) and then their finalisations (one pn2_colsum_finalize_multi)."""
        if not self.cjobs and not self.cin:
            return
        sig = (tuple(self.cin), tuple(self.cjobs))
        cache = self.grad_queue.ccache if self.grad_queue is not None else None
        hit = cache.get(self.cseg) if cache is not None else None
        if hit is None or hit[0] != sig:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("run two eager steps before capturing (the deferred-launch tables are built then)")
            launches = []
            for dt in sorted({j[0] for j in self.cin}):
                arr, blocks = [], []
                for d_, ptr, part, ld, M, Cc in self.cin:
                    if d_ != dt:
                        continue
                    j = capi.ColsumInJob()
                    j.dy, j.partial, j.ld, j.M, j.C = ptr, part, ld, M, Cc
                    nb = call.pn2_colsum_job_blocks(dt, C.byref(j))
                    if nb < 1:
                        raise RuntimeError("unsupported column-sum geometry")
                    arr.append(j); blocks.append(nb)
                launches.append(("s", dt, len(arr)) + _job_table(capi.ColsumInJob, arr, blocks))
            if self.cjobs:
                arr = []
                for part, out, nblk, Cc, ld, acc in self.cjobs:
                    j = capi.ColsumJob()
                    j.partial, j.out, j.nblk, j.C, j.ld, j.accumulate = part, out, nblk, Cc, ld, acc
                    arr.append(j)
                launches.append(("f", 0, len(arr)) + _job_table(capi.ColsumJob, arr, [call.pn2_colsum_finalize_blocks(j.C) for j in arr]))
            hit = (sig, launches)
            if cache is not None:
                cache[self.cseg] = hit
        st = _stream()
        for kind, dt, n, table, bstart, nblocks in hit[1]:
            if kind == "s":
                call.pn2_colsum_multi(dt, _p(table), _p(bstart), n, nblocks, st)
            else:
                call.pn2_colsum_finalize_multi(_p(table), _p(bstart), n, nblocks, st)
        self.ctables.append(hit)        # the tables must outlive the launches
        self.cseg += 1
        self.cjobs, self.cin, self.ckeep = [], [], []

    def colsum(self, t, M, Cp, Cc, out, accumulate):
        """out[:Cc] (+)= column sums of the [M][Cp] tensor t (bias gradients).  With a gradient queue both passes are deferred into the
        table-driven launches of flush_colsum (t stays alive in the step arena)."""
        st = _stream()
        dt = F32 if t.dtype == torch.float32 else BF16
        nb = call.pn2_rows_blocks(M, call.pn2_colsum_unit(dt, Cp))
        part = self.fbuf(nb, Cp)
        if core.DEFER_COLSUM and self.grad_queue is not None:
            self.cin.append((dt, t.data_ptr(), part.data_ptr(), Cp, M, Cp))
            self.ckeep.append(t)
        else:
            call.pn2_colsum(dt, _p(t), Cp, M, Cp, _p(part), nb, st)
        self.colsum_finalize(part, nb, Cc, Cp, out, accumulate)

    def linear(self, x, lin, residual=None):
        """nn.Linear (+ residual add) over the channels of NHWC tokens."""
        return self.conv_bn_act(x, _LinearAsConv(lin), None, bias=lin.bias, residual=residual)

    def conv_bias(self, x, conv):
        """biased nn.Conv2d without BN (patch embedding pvtv2.py:167, spatial reduction :70)."""
        return self.conv_bn_act(x, conv, None, bias=conv.bias)
