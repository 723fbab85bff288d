import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from unet_amd import _lib
from unet_amd._lib import call, lib
from unet_amd.layout import new_act, pool_for
DEV = torch.device('cuda', 0)
S = lambda: torch.cuda.current_stream().cuda_stream
def run(B,H,W,Cin,Cout,w,x):
    a = new_act(B,H,W,Cin,DEV); a.fill_from_nchw(x.to(DEV))
    out = new_act(B,H,W,Cout,DEV)
    wd = w.to(DEV)
    pk = torch.empty(Cout*9*Cin, dtype=torch.bfloat16, device=DEV)
    call("uh_pack_weights", wd.data_ptr(), pk.data_ptr(), Cout, 9, Cin, Cin*9, 1, 9, 9*Cin, Cin, 0, S())
    parts = lib.uh_conv_gemm_mtiles(B,H,W,Cout)
    stat = torch.zeros(parts,2,Cout,device=DEV)
    call("uh_conv_gemm", B,H,W,a.ptr,a.cstride,a.coff,Cin,pk.data_ptr(),Cout,9,_lib.EPI_RAW_STATS,out.ptr,out.cstride,out.coff,None,None,None,stat.data_ptr(),pool_for(DEV).rowmask(B,H,W).data_ptr(),S())
    torch.cuda.synchronize()
    return out.to_nchw().cpu()
for (B,H,W,Cin,Cout) in [(2,16,16,64,64),(1,8,24,128,128)]:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B,Cin,H,W,generator=g).bfloat16().float()
    for t in range(9):
        w = torch.zeros(Cout,Cin,3,3)
        for c in range(min(Cin,Cout)): w[c,c,t//3,t%3] = 1.0
        got = run(B,H,W,Cin,Cout,w,x)
        want = F.conv2d(x,w,None,padding=1)
        bad = (got-want).abs() > 1e-3
        print((B,H,W,Cin,Cout),'tap',t,'bad frac',bad.float().mean().item(), 'bad per chan (first 8):', bad.float().mean((0,2,3))[:8].tolist() if bad.any() else '')
        if bad.any() and t==0:
            idx = bad.nonzero()[:5]
            print(' first bad idx', idx.tolist())
            print(' bad by y:', bad.float().mean((0,1,3)).tolist())
            print(' bad by x:', bad.float().mean((0,1,2)).tolist())
