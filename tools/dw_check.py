#!/usr/bin/env python3
"""Outputs of the depth-wise 3x3 kernels on seeded inputs: `dw_check.py save F` under PN2_DW_WIN=0 and =1, then `dw_check.py cmp F0 F1`.
The conv outputs (z, gelu(z), data gradient, dz = dy * gelu'(z)) of the window kernels must be bit-identical with the round-3 walks; the weight-gradient
sums are taken over another thread geometry (fp32 order), so they are compared after the sum over the chunks, relative to the largest term."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch

SHAPES = [(2, 88, 88, 512), (2, 44, 44, 1024), (3, 22, 22, 1280), (2, 11, 11, 2048), (1, 7, 5, 64), (2, 13, 37, 96), (1, 3, 100, 40), (2, 9, 9, 6), (1, 1, 1, 8), (1, 2, 17, 4)]


def run():
    from pn2.capi import call, BF16
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    for (N, H, W, Cc) in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(N * 1000 + H + W + Cc)
        M = N * H * W
        x = torch.randn(M, Cc, generator=g).cuda().bfloat16(); dz = torch.randn(M, Cc, generator=g).cuda().bfloat16()
        w = torch.randn(Cc, 9, generator=g).cuda(); b = torch.randn(Cc, generator=g).cuda()
        z = torch.zeros_like(x); y = torch.zeros_like(x); gx = torch.zeros_like(x); dzo = torch.zeros_like(x)
        call.pn2_dwconv3x3(BF16, P(x), P(w), P(b), P(z), P(y), N, H, W, Cc, 0, 0, st)
        call.pn2_dwconv3x3(BF16, P(dz), P(w), None, P(gx), None, N, H, W, Cc, 1, 0, st)
        nb = call.pn2_dwconv3x3_wgrad_blocks(BF16, N, H, W, Cc)
        part = torch.zeros(nb, Cc * 10, device="cuda"); part2 = torch.zeros(nb, Cc * 10, device="cuda")
        call.pn2_dwconv3x3_wgrad(BF16, P(dz), P(x), P(part), nb, N, H, W, Cc, None, None, st)
        call.pn2_dwconv3x3_wgrad(BF16, P(dz), P(x), P(part2), nb, N, H, W, Cc, P(z), P(dzo), st)
        torch.cuda.synchronize()
        import torch.nn.functional as F
        xi = x.double().view(N, H, W, Cc).permute(0, 3, 1, 2); di = dz.double().view(N, H, W, Cc).permute(0, 3, 1, 2)
        zr = F.conv2d(xi, w.double().view(Cc, 1, 3, 3), b.double(), padding=1, groups=Cc).permute(0, 2, 3, 1).reshape(M, Cc)
        gr = F.conv2d(di, w.double().view(Cc, 1, 3, 3).flip(2, 3), None, padding=1, groups=Cc).permute(0, 2, 3, 1).reshape(M, Cc)
        out[(N, H, W, Cc)] = dict(z=z.cpu(), y=y.cpu(), gx=gx.cpu(), dzo=dzo.cpu(), wg=part.double().sum(0).cpu(), wg2=part2.double().sum(0).cpu(),
                                  zerr=float((z.double() - zr).abs().max()), gxerr=float((gx.double() - gr).abs().max()), zmax=float(zr.abs().max()))
    return out


if __name__ == "__main__":
    if sys.argv[1] == "save":
        torch.save(run(), sys.argv[2])
    else:
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        bad = 0
        for k in a:
            eq = {n: bool(torch.equal(a[k][n], b[k][n])) for n in ("z", "y", "gx", "dzo")}
            rel = {n: float((a[k][n] - b[k][n]).abs().max() / a[k][n].abs().max()) for n in ("wg", "wg2")}
            ok = all(eq.values()) and all(v < 2e-6 for v in rel.values())
            bad += not ok
            dif = {n: float((a[k][n].double() - b[k][n].double()).abs().max()) for n in ("z", "y", "gx", "dzo")}
            print(k, eq, {n: f"{v:.1e}" for n, v in rel.items()}, "" if ok else "  <-- MISMATCH", "| max |a-b|", {n: f"{v:.1e}" for n, v in dif.items()},
                  "| vs f64: z %.1e / %.1e  gx %.1e / %.1e (max |z| %.1f)" % (a[k]["zerr"], b[k]["zerr"], a[k]["gxerr"], b[k]["gxerr"], a[k]["zmax"]))
        print("ALL OK" if not bad else f"{bad} MISMATCHED")
