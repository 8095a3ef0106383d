#!/usr/bin/env python3
"""A few launches of the depth-wise 3x3 kernels on the stage-1 Mlp shape of PVTv2-B2 (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch
from pn2.capi import call, BF16
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
N, H, W, Cc = 16, 88, 88, 512
M = N * H * W
x = torch.randn(M, Cc, device="cuda").bfloat16(); dz = torch.randn(M, Cc, device="cuda").bfloat16()
z = torch.empty_like(x); y = torch.empty_like(x); w = torch.randn(Cc, 9, device="cuda"); b = torch.randn(Cc, device="cuda")
nb = call.pn2_dwconv3x3_wgrad_blocks(BF16, N, H, W, Cc); part = torch.empty(nb, Cc * 10, device="cuda")
for _ in range(3):
    call.pn2_dwconv3x3(BF16, P(x), P(w), P(b), P(z), P(y), N, H, W, Cc, 0, 0, st)
    call.pn2_dwconv3x3(BF16, P(dz), P(w), None, P(z), None, N, H, W, Cc, 1, 0, st)
    call.pn2_dwconv3x3_wgrad(BF16, P(dz), P(x), P(part), nb, N, H, W, Cc, None, None, st)
    call.pn2_dwconv3x3_wgrad(BF16, P(dz), P(x), P(part), nb, N, H, W, Cc, P(x), P(z), st)
torch.cuda.synchronize()
print("tensor MB", M * Cc * 2 / 1e6)
