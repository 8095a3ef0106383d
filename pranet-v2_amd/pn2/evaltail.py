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
