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


def threshold_metrics(pred_u8, gt, full=False):
    """full=True: also meanEm, Sm and wFm - with meanDic / meanIoU / mae the complete opt["metrics"] set of the reference's eval_for_testAllInOne
    (eval.py:18-66); pred_u8 / gt are then 2-D maps (H, W).
    The reference's 256-threshold sweep (eval.py:22-50, Fmeasure_calu eval_functions.py:131-166) for one uint8 prediction map and its
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
    out = {"curves": cols, "meanDic": float(m[3]), "meanIoU": float(m[5]), "meanSen": float(m[1]), "meanSpe": float(m[2]), "meanFm": float(m[4]),
           "mae": mae}
    if full:
        out.update(_em_sm_wfm(p, g, h_all, h_gt, total, num_rec, num_and))
    return out


def eval_for_testAllInOne(opt, pred_u8, gt):
    """The reference's eval.py:18-66 with GPU tensors: the values of opt["metrics"] (any of meanDic, meanIoU, wFm, Sm, meanEm, mae) in that order."""
    r = threshold_metrics(pred_u8, gt, full=any(k in ("wFm", "Sm", "meanEm") for k in opt["metrics"]))
    return [r[k] for k in opt["metrics"]]


def _em_sm_wfm(p, g, h_all, h_gt, total, num_rec, num_and):
    """meanEm / Sm / wFm of eval_for_testAllInOne (eval.py:31-33,46-47,59-60) from the two histograms, the integer quadrant moments of pn2_eval_region_sums and
    the weighted error sums of pn2_eval_wfm; float64 on the host with the expressions of utils/eval_functions.py (the sums over pixels the reference takes with
    numpy's pairwise summation become count x value / exact integer moments here: agreement to ~1e-15, not bit for bit)."""
    import numpy as np
    eps = np.finfo(np.float64).eps
    H, W = int(p.shape[-2]), int(p.shape[-1])
    N, num_obj = int(total), int(h_gt.sum())
    vals = np.arange(256).astype(np.float64) / 255
    # ---- EnhancedMeasure per threshold (eval_functions.py:168-192): the alignment matrix of a binary map takes one value per (prediction bit, gt bit) class
    E = np.zeros(256)
    for i in range(256):
        nr, na = int(num_rec[i]), int(num_and[i])
        if num_obj == 0:
            s = N - nr
        elif num_obj == N:
            s = nr
        else:
            mp_, mg = np.float64(nr) / N, np.float64(num_obj) / N
            s = 0.0
            for b, gg, cnt in ((1, 1, na), (1, 0, nr - na), (0, 1, num_obj - na), (0, 0, N - nr - (num_obj - na))):
                if cnt:
                    ap, ag = b - mp_, gg - mg
                    al = 2 * (ag * ap) / (ag ** 2 + ap ** 2 + eps)
                    s += cnt * (((al + 1) ** 2) / 4)
        E[i] = s / (N - 1 + eps)
    out = {"E": E, "meanEm": float(E.mean())}
    # ---- StructureMeasure (eval_functions.py:5-94)
    h_bg = h_all - h_gt
    mean_pred = float((h_all * vals).sum()) / N
    y = np.float64(num_obj) / N
    if y == 0:
        sm = 1 - mean_pred
    elif y == 1:
        sm = mean_pred
    else:
        def obj(h, v):              # Object(): mean and population std of the values v (histogram h)
            n = h.sum()
            x = float((h * v).sum()) / n
            sd = np.sqrt(float((h * (v - x) ** 2).sum()) / n)
            return 2.0 * x / (x ** 2 + 1 + sd + eps)
        s_obj = y * obj(h_gt, vals) + (1 - y) * obj(h_bg, 1 - vals)
        q25 = torch.empty(25, dtype=torch.int64, device=p.device)
        call.pn2_eval_region_sums(_p(p), _p(g), H, W, _p(q25), _stream())
        q = [int(v) for v in q25.cpu()]
        s_reg = 0.0
        for k in range(4):
            n, sk, skk, sg, skg = q[3 + 5 * k: 8 + 5 * k]
            if n == 0:
                s_reg = float("nan")            # numpy's mean of an empty quadrant is nan and the reference propagates it (eval_functions.py:50-70)
                continue
            x, yy = sk / 255 / n, sg / n
            den = n - 1 + eps
            sx = (n * skk - sk * sk) / (n * 65025) / den          # sum (p - x)^2 / (N - 1 + eps), exact integer numerators
            sy = (n * sg - sg * sg) / n / den
            sxy = (n * skg - sk * sg) / (n * 255) / den
            al, be = 4 * x * yy * sxy, (x ** 2 + yy ** 2) * (sx + sy)
            qq = al / (be + eps) if al != 0 else (1 if be == 0 else 0)
            s_reg = s_reg + qq * (n / N)
        sm = 0.5 * s_obj + 0.5 * s_reg
        if sm < 0:
            sm = 0
    out["Sm"] = float(sm)
    # ---- original_WFb (eval_functions.py:96-129)
    if num_obj == 0:
        out["wFm"] = float("nan")           # scipy's feature transform has no site to return: the reference's value is undefined here
    else:
        xk, yk = np.mgrid[-7 // 2 + 1:7 // 2 + 1, -7 // 2 + 1:7 // 2 + 1]
        K = np.exp(-((xk ** 2 + yk ** 2) / (2.0 * 5 ** 2))); K = K / K.sum()          # fspecial_gauss(7, 5)
        Kd = torch.from_numpy(np.ascontiguousarray(K)).to(p.device)
        nblk = int(call.pn2_eval_wfm_blocks(H, W))
        wi = torch.empty(H * W, dtype=torch.int32, device=p.device)
        wd = torch.empty(2 * H * W, dtype=torch.float64, device=p.device)
        part = torch.empty(nblk, 2, dtype=torch.float64, device=p.device)
        call.pn2_eval_wfm(_p(p), _p(g), H, W, _p(Kd), float(np.log(1 - 0.5) / 5), _p(wi), _p(wd), _p(part), _stream())
        pt = part.cpu().numpy()
        s_fg, s_bg = float(pt[:, 0].sum()), float(pt[:, 1].sum())
        TPw, FPw = num_obj - s_fg, s_bg
        R = 1 - s_fg / num_obj
        Pq = TPw / (TPw + FPw + eps)
        out["wFm"] = float(2 * R * Pq / (R + Pq + eps))
    return out
