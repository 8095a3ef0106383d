#!/usr/bin/env python3
"""Micro-benchmark of the spatial-reduction attention kernels on the PVTv2-B2 shapes of configs 4 (352^2) and 5 (512^2), bs=16, bf16."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch
from pn2.capi import call, BF16

P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def bench(B, Nq, Nkv, heads):
    dev = "cuda"; st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    Cc = heads * 64
    q = torch.randn(B, Nq, Cc, device=dev).bfloat16(); kv = torch.randn(B, Nkv, 2 * Cc, device=dev).bfloat16()
    o = torch.empty_like(q); do = torch.randn_like(q); dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    lse = torch.empty(B, heads, Nq, device=dev); delta = torch.empty(B, heads, Nq, device=dev)
    nb = call.pn2_attn_bwd_blocks(BF16, B, heads, Nq)
    part = torch.empty(B, heads, nb, 2, (Nkv + 63) // 64 * 64, 64, device=dev)
    sc = 0.125
    t_f = timeit(lambda: call.pn2_attn_fwd(BF16, P(q), Cc, P(kv), 2 * Cc, P(o), Cc, P(lse), B, Nq, Nkv, heads, 64, sc, st))
    t_b = timeit(lambda: call.pn2_attn_bwd(BF16, P(q), Cc, P(kv), 2 * Cc, P(o), Cc, P(do), Cc, P(lse), P(dq), Cc, P(dkv), 2 * Cc, P(part), P(delta), B, Nq, Nkv, heads, 64, sc, st))
    fl = 4.0 * B * heads * Nq * Nkv * 64
    # numerics against torch (fp32 math on the bf16-rounded inputs), first two samples
    nb_ = min(B, 2)
    qq = q[:nb_].float().reshape(nb_, Nq, heads, 64).permute(0, 2, 1, 3).requires_grad_(True)
    kk = kv[:nb_].float().reshape(nb_, Nkv, 2, heads, 64).permute(2, 0, 3, 1, 4).detach().requires_grad_(True)
    r = ((qq @ kk[0].transpose(-2, -1)) * sc).softmax(-1) @ kk[1]
    r.backward(do[:nb_].float().reshape(nb_, Nq, heads, 64).permute(0, 2, 1, 3))
    ro = r.permute(0, 2, 1, 3).reshape(nb_, Nq, Cc)
    rel = lambda a, b: float((a.float() - b).norm() / b.norm())
    e_o = rel(o[:nb_], ro)
    e_q = rel(dq[:nb_], qq.grad.permute(0, 2, 1, 3).reshape(nb_, Nq, Cc))
    e_kv = rel(dkv[:nb_], kk.grad.permute(1, 3, 0, 2, 4).reshape(nb_, Nkv, 2 * Cc))
    print(f"B{B} Nq{Nq:6d} Nkv{Nkv:4d} h{heads}: fwd {t_f:7.1f} us {fl/t_f/1e6:6.1f} TF/s | bwd(slots {nb:3d}) {t_b:7.1f} us {2.5*fl/t_b/1e6:6.1f} TF/s | rel err o {e_o:.1e} dq {e_q:.1e} dkv {e_kv:.1e}")


if __name__ == "__main__":
    for shp in [(16, 7744, 121, 1), (16, 1936, 121, 2), (16, 484, 121, 5), (16, 121, 121, 8),
                (16, 16384, 256, 1), (16, 4096, 256, 2), (16, 1024, 256, 5), (16, 256, 256, 8), (2, 256, 196, 2), (2, 256, 160, 2), (2, 200, 250, 1)]:
        bench(*shp)
