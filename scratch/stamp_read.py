"""run the stamped dgrad-mode band kernel once on a conv_up2_1-shaped problem and print per-phase cycle splits"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import _lib
from unet_amd._lib import call, lib
from unet_amd.layout import new_act, pool_for
DEV = torch.device('cuda', 0)
S = lambda: torch.cuda.current_stream().cuda_stream
B,H,W,Cin,Cout = 32,64,64,512,256
a = new_act(B,H,W,Cin,DEV); a.buf.normal_()
out = new_act(B,H,W,Cout,DEV)
w = torch.randn(Cout*9*Cin, device=DEV).bfloat16()
mask = pool_for(DEV).rowmask(B,H,W)
dbg = torch.zeros(4096, dtype=torch.int64, device=DEV)
for _ in range(3):
    call("uh_conv_gemm", B,H,W,a.ptr,a.cstride,0,Cin,w.data_ptr(),Cout,9,_lib.EPI_MASKED,out.ptr,out.cstride,0,None,None,None,dbg.data_ptr(),mask.data_ptr(),S())
torch.cuda.synchronize()
d = dbg.cpu().view(-1)[:2*8*3*5].view(2, 24, 5)
for grp in range(2):
    print("wave", grp*4)
    base = d[grp,0,0].item()
    for ph in range(24):
        t = d[grp,ph]
        nxt = d[grp,ph+1,0].item() if ph+1 < 24 else t[4].item()
        print(f"  phase {ph:2d}: start {t[0].item()-base:6d}  load {t[1]-t[0]:5d}  bar1 {t[2]-t[1]:5d}  mfma {t[3]-t[2]:5d}  bar2 {t[4]-t[3]:5d}  total {nxt-t[0].item():5d}")
