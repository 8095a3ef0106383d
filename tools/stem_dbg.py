import ctypes as C, os, sys
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/pranet-v2_amd")
import torch, torch.nn.functional as F
from pn2 import capi
from pn2.capi import call, BF16
key=('g', 32, 352, 352, 176, 176, 8, 8, 32, 3, 3, 2, 1, 1, 1, 1, 0)
_, N, H, W, OH, OW, Cin_p, ld_in, Cout, KH, KW, s, ph, pw, dh, dw, tr = key
code = 2 | (2 << 2) | (1 << 4)
taps=9; M=N*OH*OW
g = torch.Generator(device="cpu").manual_seed(H * 7 + Cin_p + Cout + taps)
src = torch.randn(N, H, W, ld_in, generator=g).bfloat16()
w = (torch.randn(Cout, Cin_p, KH, KW, generator=g) * (2.0 / (taps * Cin_p)) ** 0.5).bfloat16()
ref = F.conv2d(src.double().permute(0,3,1,2), w.double(), None, s, (ph,pw), (dh,dw)).permute(0,2,3,1).reshape(M, Cout)
rup=lambda v,m:(v+m-1)//m*m
d=capi.ConvDesc(); d.N,d.H,d.W,d.OH,d.OW=N,H,W,OH,OW; d.Cin_p,d.ld_in,d.Cout,d.ld_out=Cin_p,ld_in,Cout,Cout
d.KH,d.KW,d.stride,d.pad_h,d.pad_w,d.dil_h,d.dil_w=KH,KW,s,ph,pw,dh,dw; d.transposed=0; d.Kp=rup(taps*Cin_p,128)
wp=torch.zeros(128,d.Kp,dtype=torch.bfloat16); wp[:Cout,:taps*Cin_p]=w.permute(0,2,3,1).reshape(Cout,-1)
P=lambda t:C.c_void_p(t.data_ptr())
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
sg,wg=src.cuda(),wp.cuda()
rvar=ref.var(0,unbiased=False); rmean=ref.mean(0)
for extra in (0, 0x20, 0x10, 0x40):
  for tm,cd in ((128, code), (64, 2 | (1<<2) | (1<<4))):
    nblk=(M+tm-1)//tm
    out=torch.empty(M,Cout,dtype=torch.bfloat16,device="cuda"); psum=torch.empty(nblk,Cout,device="cuda"); psq=torch.empty(nblk,Cout,device="cuda")
    d.flags=1|extra|(cd<<8)
    call.pn2_conv_gemm(BF16,P(sg),P(wg),P(out),P(psum),P(psq),C.byref(d),st); torch.cuda.synchronize()
    n_t=torch.full((nblk,),float(tm),dtype=torch.float64); n_t[-1]=M-(nblk-1)*tm
    mt,m2=psum.double().cpu(),psq.double().cpu()
    mean=(mt*n_t[:,None]).sum(0)/M; var=(m2.sum(0)+(n_t[:,None]*(mt-mean)**2).sum(0))/M
    print(hex(extra), tm, "mean err/sd", float(((mean-rmean).abs()/rvar.sqrt()).max()), "var rel", float(((var-rvar)/rvar).abs().max()), "out err", float((out.double().cpu()-ref).abs().max()))
