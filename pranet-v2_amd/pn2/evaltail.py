"""Inference tail of the reference's MyTest_med.py:104-111 on the GPU: res = p2+p3+p4+p5 -> bilinear resize to the
ground-truth size (align_corners=False) -> sigmoid -> min-max normalise -> uint8."""
import ctypes as C

import torch

from .capi import call, F32
from .engine import Engine, Act, _p, _stream


def test_postprocess(outs, gt_shape):
    """outs: the model's 8-tuple (or its first four fg maps), each (1,1,h,w) fp32 on the GPU.  Returns a uint8 (H,W) GPU tensor."""
    eng = Engine(F32, training=False, need_grad=False)
    acts = []
    for o in outs[:4]:
        if not o.is_cuda:
            raise RuntimeError("pn2.evaltail needs GPU tensors (no CPU fallback)")
        n, k, h, w = o.shape
        assert n == 1 and k == 1
        acts.append(Act(eng, o.float().contiguous().reshape(1, h, w, 1), 1, 1, 1, F32, requires_grad=False))
    s = eng.add(eng.add(eng.add(acts[0], acts[1]), acts[2]), acts[3])
    r = eng.resize_to(s, int(gt_shape[0]), int(gt_shape[1]), align_corners=False)
    n = r.M
    out = torch.empty((int(gt_shape[0]), int(gt_shape[1])), dtype=torch.uint8, device=r.t.device)
    scratch = torch.empty(2 + 2 * 512, dtype=torch.float32, device=r.t.device)
    call.pn2_eval_tail(r.ptr, _p(out), _p(scratch), n, _stream())
    return out


def threshold_metrics(pred_u8, gt):
    """The reference's 256-threshold sweep (eval.py:22-50, Fmeasure_calu eval_functions.py:131-166) for one uint8 prediction map and its
    ground truth, both on the GPU.  One histogram kernel reads the maps; the 256x6 curves (precision, recall, specificity, Dice,
    F-measure, IoU) are then finished on the host in float64 with the reference's expressions, so they equal its numpy result exactly.
    Returns {"curves": ndarray[256, 6], "meanDic", "meanIoU", "meanSen", "meanSpe", "meanFm", "mae"}."""
    import numpy as np
    if not (pred_u8.is_cuda and gt.is_cuda):
        raise RuntimeError("pn2.evaltail needs GPU tensors (no CPU fallback)")
    if pred_u8.dtype != torch.uint8 or pred_u8.numel() != gt.numel():
        raise ValueError("pred must be uint8 with as many pixels as gt")
    p = pred_u8.contiguous()
    g = gt.float().contiguous()
    hist = torch.empty(512, dtype=torch.int32, device=p.device)
    call.pn2_eval_hist(_p(p), _p(g), p.numel(), _p(hist), _stream())
    h = hist.cpu().numpy().astype(np.int64)
    h_all, h_gt = h[:256], h[256:]
    vals = np.arange(256).astype(np.float64) / 255                  # pred.astype(float64) / 255   (eval.py:28)
    thr = np.minimum(np.linspace(1, 0, 256), 1)                     # eval.py:19, eval_functions.py:132-133
    ge = (vals[None, :] >= thr[:, None]).astype(np.int64)           # Label3[pred >= threshold] = 1
    num_rec, num_and = ge @ h_all, ge @ h_gt
    total, num_obj = int(h_all.sum()), np.float64(h_gt.sum())
    num_no_rec = total - num_rec
    fn = num_obj - num_and
    fp = num_rec - num_and
    tn = num_no_rec - fn
    cols = np.zeros((256, 6))
    ok = num_and != 0
    with np.errstate(divide="ignore", invalid="ignore"):
        pre = num_and / num_rec
        rec = num_and / num_obj
        cols[:, 0], cols[:, 1] = pre, rec
        cols[:, 2] = tn / (tn + fp)
        cols[:, 3] = 2 * num_and / (num_obj + num_rec.astype(np.float64))
        cols[:, 4] = (2.0 * pre * rec) / (pre + rec)
        cols[:, 5] = num_and / (fn + num_rec)
    cols[~ok] = 0
    h_bg = h_all - h_gt
    mae = float((h_gt * np.abs(1.0 - vals)).sum() + (h_bg * vals).sum()) / total
    m = cols.mean(axis=0)
    return {"curves": cols, "meanDic": float(m[3]), "meanIoU": float(m[5]), "meanSen": float(m[1]), "meanSpe": float(m[2]), "meanFm": float(m[4]),
            "mae": mae}
