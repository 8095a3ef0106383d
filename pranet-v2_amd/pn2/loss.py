"""Dual structure loss of the reference (MyTrain_med.py:19-38) as fused gfx950 kernels.

`structure_loss(pred, pred_bg, mask_fg, mask_bg)` has the reference's signature; `structure_loss_multi` evaluates
all supervision pairs of MyTrain_med.py:78-82 in one pass (the 31x31 boundary weights are computed once instead
of four times).  The background target is 1 - mask_fg, as at every call site of the reference (:74).
"""
import ctypes as C

import torch

import os

from .capi import F32, call
from .engine import _p, _stream

TAIL_ISUM = os.environ.get("PN2_TAIL_ISUM", "1") == "1"          # 0: the one-pass tail without the fixed-point image-sum accumulators (three launches; A/B)


def loss_forward(buf, P, mask, N, HW, H, W, weit=None):
    """buf: [2P][N][HW] fp32 contiguous logits.  Returns (loss[P+1], saved) — raw kernel driver, no autograd.  weit: the boundary weights of `mask` when the
    caller has them already."""
    dev = buf.device
    if weit is None:
        weit = torch.empty((N, HW), dtype=torch.float32, device=dev)
        call.pn2_loss_weights(_p(mask), _p(weit), N, H, W, 31, _stream())
    nb = call.pn2_loss_blocks(HW)
    partial = torch.empty((P, N, nb, 5), dtype=torch.float32, device=dev)
    sums = torch.empty((P, N, 4), dtype=torch.float32, device=dev)
    wsum = torch.empty((N,), dtype=torch.float32, device=dev)
    loss = torch.empty((P + 1,), dtype=torch.float32, device=dev)
    call.pn2_structure_loss_fwd(_p(buf), N * HW, P, _p(mask), _p(weit), _p(partial), _p(sums), _p(wsum), _p(loss), N, HW, _stream())
    return loss, (weit, sums, wsum)


def loss_backward(buf, dbuf, P, mask, saved, N, HW, gscale=1.0):
    weit, sums, wsum = saved
    call.pn2_structure_loss_bwd(_p(buf), _p(dbuf), N * HW, P, _p(mask), _p(weit), _p(wsum), _p(sums), float(gscale), N, HW, _stream())


def tail_desc(eng, tail, P, N, OH, OW):
    """pn2_tail_desc for the engine's deferred lateral maps (Engine.tail: slot j -> (low-res Act, align_corners, rh, rw))."""
    from . import capi
    d = capi.TailDesc()
    d.N, d.OH, d.OW, d.P = N, OH, OW, P
    acs = {t[1] for t in tail.values()}
    if len(acs) != 1:
        raise RuntimeError("lateral maps mix align_corners modes")
    d.align_corners = acs.pop()
    for j in range(2 * P):
        x, _, rh, rw = tail[j]
        m = d.maps[j]
        m.src, m.h, m.w, m.rh, m.rw = x.t.data_ptr(), x.H, x.W, rh, rw
    return d


def tail_forward(eng, tail, lat, P, mask, N, H, W):
    """Fused lateral up-sampling + dual structure loss (pn2_dsra_tail_fwd).  Returns (loss[P+1], saved)."""
    HW = H * W
    weit = eng.alloc((N, HW), torch.float32)
    call.pn2_loss_weights(_p(mask), _p(weit), N, H, W, 31, _stream())
    nb = call.pn2_dsra_tail_blocks(H)
    partial = eng.alloc((P, N, nb, 5), torch.float32)
    sums = eng.alloc((P, N, 4), torch.float32)
    wsum = eng.alloc((N,), torch.float32)
    loss = torch.empty((P + 1,), dtype=torch.float32, device=lat.device)
    d = tail_desc(eng, tail, P, N, H, W)
    call.pn2_dsra_tail_fwd(C.byref(d), _p(lat), _p(mask), _p(weit), _p(partial), _p(sums), _p(wsum), _p(loss), _stream())
    return loss, (weit, sums, wsum, d)


def tail_forward_backward(eng, tail, lat, P, mask, N, H, W, gscale=1.0):
    """Up-sampling + dual structure loss + their backward in ONE pass over the pixels (pn2_dsra_tail_fwd_bwd: the forward walk also leaves the gradient's
    linear components, a small kernel applies the image-wide coefficients).  Returns loss[P+1], or None when the geometry is not served (the caller then
    runs tail_forward + tail_backward)."""
    d = tail_desc(eng, tail, P, N, H, W)
    if not int(call.pn2_dsra_tail_fused_ok(C.byref(d))):
        return None
    for j in range(2 * P):
        g, acc = tail[j][0].grad_sink()
        d.maps[j].dsrc, d.maps[j].accumulate = g.data_ptr(), acc
    HW = H * W
    weit = eng.alloc((N, HW), torch.float32)
    # the image sums are accumulated by the walk itself (two-word fixed-point integer atomics into isum - order-independent, so replays agree bit for bit -
    # zeroed by the weights launch in front of it): two launches instead of three
    isum = eng.alloc((P * N * 10,), torch.int64) if (TAIL_ISUM and P * N <= 1024) else None
    if isum is not None:
        call.pn2_loss_weights_clear(_p(mask), _p(weit), N, H, W, 31, _p(isum), P * N * 10, _stream())
    else:
        call.pn2_loss_weights(_p(mask), _p(weit), N, H, W, 31, _stream())
    nb = call.pn2_dsra_tail_blocks(H)
    partial = eng.alloc((P, N, nb, 5), torch.float32)
    sums = eng.alloc((P, N, 4), torch.float32)
    wsum = eng.alloc((N,), torch.float32)
    per = eng.alloc((P, N), torch.float32)
    loss = torch.empty((P + 1,), dtype=torch.float32, device=lat.device)
    need = int(call.pn2_dsra_tail_fused_scratch(C.byref(d)))
    scratch = eng.alloc((need,), torch.float32)
    call.pn2_dsra_tail_fwd_bwd(C.byref(d), _p(lat), _p(mask), _p(weit), _p(partial), _p(sums), _p(wsum), _p(per), _p(loss), float(gscale), _p(scratch), need,
                               _p(isum), _stream())
    return loss


def tail_backward(eng, tail, P, mask, saved, gscale=1.0):
    """Loss gradient + bilinear adjoint straight into the low-res maps' gradients (pn2_dsra_tail_bwd)."""
    weit, sums, wsum, d = saved
    for j in range(2 * P):
        x = tail[j][0]
        g, acc = x.grad_sink()
        d.maps[j].dsrc, d.maps[j].accumulate = g.data_ptr(), acc
    need = int(call.pn2_dsra_tail_scratch(C.byref(d)))          # band partials of the band-wise backward (0: row kernels)
    scratch = eng.alloc((need,), torch.float32) if need > 0 else None
    call.pn2_dsra_tail_bwd(C.byref(d), _p(mask), _p(weit), _p(wsum), _p(sums), float(gscale),
                           _p(scratch) if scratch is not None else None, need, _stream())


# The four structure_loss calls of a step (MyTrain_med.py:78-81) pass the SAME mask tensor: its 31 x 31 boundary weights are computed by the first call and reused
# by the other three.  The entry holds the mask object itself (identity + version counter decide a hit; holding it keeps its address from being recycled for
# another mask while the entry is alive) - one entry, replaced by the next new mask.
_WEIT = [None]


def _mask_weights(mask, N, H, W):
    e = _WEIT[0]
    if e is not None and e[0] is mask and e[1] == mask._version:
        return e[2], e[3]
    HW = H * W
    m = mask.reshape(N, HW).float().contiguous()
    weit = torch.empty((N, HW), dtype=torch.float32, device=mask.device)
    call.pn2_loss_weights(_p(m), _p(weit), N, H, W, 31, _stream())
    _WEIT[0] = (mask, mask._version, m, weit)
    return m, weit


def _in_place_maps(preds, P, N, HW):
    """-> (base tensor, map stride in elements) when the 2P logit maps are fp32, contiguous and sit at base + j * stride (stride > 0) - the maps a call site of the
    mirror model hands out are views of ONE block, slots i and i + 4 - so the loss kernels read them where they are; None: they are gathered into one buffer."""
    if any(p.dtype != torch.float32 or not p.is_contiguous() or p.numel() != N * HW for p in preds):
        return None
    a0 = preds[0].data_ptr()
    if len(preds) < 2:
        return None
    d = preds[1].data_ptr() - a0
    if d <= 0 or d % 4 or any(p.data_ptr() - a0 != j * d for j, p in enumerate(preds)):
        return None
    return preds[0], d // 4


class _StructureLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, P, *preds):
        if not mask.is_cuda:
            raise RuntimeError("pn2.structure_loss needs GPU tensors (no CPU fallback)")
        N, _, H, W = mask.shape
        HW = H * W
        m, weit = _mask_weights(mask, N, H, W)
        inp = _in_place_maps(preds, P, N, HW)
        if inp is None:
            buf = torch.stack([p.reshape(N, HW) for p in preds]).float().contiguous()
            stride = N * HW
            loss, saved = loss_forward(buf, P, m, N, HW, H, W, weit)
            ctx.save_for_backward(buf, m, *saved)
        else:
            base, stride = inp
            dev = base.device
            nb = call.pn2_loss_blocks(HW)
            partial = torch.empty((P, N, nb, 5), dtype=torch.float32, device=dev)
            sums = torch.empty((P, N, 4), dtype=torch.float32, device=dev)
            wsum = torch.empty((N,), dtype=torch.float32, device=dev)
            loss = torch.empty((P + 1,), dtype=torch.float32, device=dev)
            call.pn2_structure_loss_fwd(_p(base), stride, P, _p(m), _p(weit), _p(partial), _p(sums), _p(wsum), _p(loss), N, HW, _stream())
            ctx.save_for_backward(base, m, weit, sums, wsum, *preds)          # (the maps themselves: nothing is copied)
        ctx.dims = (P, N, HW, preds[0].shape, stride, inp is not None)
        return loss[P], loss[:P]

    @staticmethod
    def backward(ctx, gtotal, gpairs):
        buf, m, weit, sums, wsum = ctx.saved_tensors[:5]
        P, N, HW, shape, stride, in_place = ctx.dims
        dbuf = torch.empty((2 * P, N, HW), dtype=torch.float32, device=buf.device)
        scale = torch.zeros(P, device=buf.device)
        if gtotal is not None:
            scale = scale + gtotal
        if gpairs is not None:
            scale = scale + gpairs
        # the upstream gradient of every pair goes to the kernel on the device: no scaling pass over the gradient maps
        call.pn2_structure_loss_bwd_dev(_p(buf), _p(dbuf), stride, N * HW, P, _p(m), _p(weit), _p(wsum), _p(sums), _p(scale), 1.0, N, HW, _stream())
        return (None, None, *[dbuf[j].reshape(shape) for j in range(2 * P)])


def structure_loss_multi(preds_fg, preds_bg, mask, return_pairs=False):
    """sum_j structure_loss(preds_fg[j], preds_bg[j], mask, 1 - mask)  (MyTrain_med.py:78-82)."""
    P = len(preds_fg)
    total, pairs = _StructureLoss.apply(mask, P, *preds_fg, *preds_bg)
    return (total, pairs) if return_pairs else total


def structure_loss(pred, pred_bg, mask_fg, mask_bg=None):
    """Reference signature (MyTrain_med.py:19).  mask_bg is taken to be 1 - mask_fg, as at its call sites."""
    return structure_loss_multi([pred], [pred_bg], mask_fg)


class _MutationLoss(torch.autograd.Function):
    """The multi-class dual-supervision loss of EMCAD/trainer.py:106-140 as two kernels (pn2_mutation_loss_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, label, bg_mask, lc, *maps):
        if not maps[0].is_cuda:
            raise RuntimeError("pn2.mutation_loss needs GPU tensors (no CPU fallback)")
        N, K, H, W = maps[0].shape
        nhwc = [m.permute(0, 2, 3, 1).float().contiguous() for m in maps]          # no copy for the engine's K-channel output maps
        lab = label.long().contiguous()
        bgm = bg_mask.float().contiguous()
        nb, wd = call.pn2_mutation_loss_blocks(N * H * W), call.pn2_mutation_loss_width(K)
        if wd < 0:
            raise RuntimeError(f"pn2.mutation_loss is built for K = 9 classes (got {K})")
        dev = maps[0].device
        partial = torch.empty((nb, wd), dtype=torch.float32, device=dev)
        sums = torch.empty(wd, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        PA = C.c_void_p * 4
        fg, bg = PA(*[t.data_ptr() for t in nhwc[:4]]), PA(*[t.data_ptr() for t in nhwc[4:]])
        call.pn2_mutation_loss_fwd(fg, bg, _p(lab), _p(bgm), N, H * W, K, lc[0], lc[1], lc[2], _p(partial), _p(sums), _p(loss), _stream())
        ctx.save_for_backward(lab, bgm, sums, *nhwc)
        ctx.meta = (N, K, H, W, lc)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lab, bgm, sums, *nhwc = ctx.saved_tensors
        N, K, H, W, lc = ctx.meta
        grads = [torch.empty_like(t) for t in nhwc]
        PA = C.c_void_p * 4
        fg, bg = PA(*[t.data_ptr() for t in nhwc[:4]]), PA(*[t.data_ptr() for t in nhwc[4:]])
        dfg, dbg = PA(*[t.data_ptr() for t in grads[:4]]), PA(*[t.data_ptr() for t in grads[4:]])
        call.pn2_mutation_loss_bwd(fg, bg, dfg, dbg, _p(lab), _p(bgm), N, H * W, K, lc[0], lc[1], lc[2], _p(sums), 1.0, _stream())
        return (None, None, None, *[(gr * g).permute(0, 3, 1, 2) for gr in grads])


def mutation_forward_backward(eng, outs, label, bg_mask, lc=(0.5, 0.7, 0.3), gscale=1.0):
    """Trainer path of the same loss: `outs` are the engine's 8 fp32 [N][H][W][K] maps; the loss kernel reads them in place and the backward
    kernel writes their gradients straight into the activations' gradient buffers (no autograd bridge, no copies).  Returns loss[1]."""
    N, H, W, K = outs[0].N, outs[0].H, outs[0].W, outs[0].C
    for o in outs:
        assert o.dt == F32 and o.ld == K and (o.N, o.H, o.W, o.C) == (N, H, W, K)
    nb, wd = call.pn2_mutation_loss_blocks(N * H * W), call.pn2_mutation_loss_width(K)
    if wd < 0:
        raise RuntimeError(f"pn2.mutation_loss is built for K = 9 classes (got {K})")
    lab = label.long().contiguous()
    bgm = bg_mask.float().contiguous()
    partial, sums, loss = eng.fbuf(nb, wd), eng.fbuf(wd), eng.fbuf(1)
    PA = C.c_void_p * 4
    fg, bg = PA(*[o.t.data_ptr() for o in outs[:4]]), PA(*[o.t.data_ptr() for o in outs[4:]])
    st = _stream()
    call.pn2_mutation_loss_fwd(fg, bg, _p(lab), _p(bgm), N, H * W, K, lc[0], lc[1], lc[2], _p(partial), _p(sums), _p(loss), st)
    grads = [o.grad_buf() for o in outs]
    dfg, dbg = PA(*[g.data_ptr() for g in grads[:4]]), PA(*[g.data_ptr() for g in grads[4:]])
    call.pn2_mutation_loss_bwd(fg, bg, dfg, dbg, _p(lab), _p(bgm), N, H * W, K, lc[0], lc[1], lc[2], _p(sums), float(gscale), st)
    for o in outs:
        o.grad_written = True
    eng.keep_alive = (lab, bgm)
    return loss


def mutation_loss(outs, label, bg_mask, lc=(0.5, 0.7, 0.3)):
    """sum over the 15 non-empty subsets s of the 4 scales of lc1*CE(sum fg) + lc2*Dice(softmax(sum fg)) + lc3*BCEWithLogits(sum bg, bg_mask)
    (EMCAD/trainer.py:106-140, supervision='mutation', dual): outs = the 8 maps EMCADNet returns, label (N,H,W) int, bg_mask (N,K,H,W)."""
    return _MutationLoss.apply(label, bg_mask, tuple(float(v) for v in lc), *outs)
