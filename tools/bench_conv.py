"""micro-benchmark of uh_conv_gemm / uh_conv_wgrad on UNet layer shapes (B=32, 256^2)"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import _lib
from unet_amd._lib import call, lib
from unet_amd.layout import new_act, pool_for
DEV = torch.device('cuda', 0)
S = lambda: torch.cuda.current_stream().cuda_stream
shapes = {  # name: (B,H,W,Cin,Cout)
  'conv1_2': (32,256,256,64,64), 'conv2_2': (32,128,128,128,128), 'conv3_2': (32,64,64,256,256),
  'conv4_2': (32,32,32,512,512), 'conv5_2': (32,16,16,1024,1024), 'conv_up1_1': (32,32,32,1024,512),
  'conv_up3_1': (32,128,128,256,128), 'conv_up4_1': (32,256,256,128,64), 'conv_up2_1': (32,64,64,512,256),
}
which = sys.argv[1].split(',') if len(sys.argv) > 1 and sys.argv[1] != 'all' else list(shapes)
mode = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for name in which:
    B,H,W,Cin,Cout = shapes[name]
    a = new_act(B,H,W,Cin,DEV); a.buf.normal_(); 
    out = new_act(B,H,W,Cout,DEV)
    w = torch.randn(Cout*9*Cin, device=DEV).bfloat16()
    mask = pool_for(DEV).rowmask(B,H,W)
    parts = lib.uh_conv_gemm_mtiles(B,H,W,Cout)
    stat = torch.zeros(parts*2*Cout, device=DEV)
    bias = torch.zeros(Cout, device=DEV)
    if mode == 'fwd':
        f = lambda: call("uh_conv_gemm", B,H,W,a.ptr,a.cstride,0,Cin,w.data_ptr(),Cout,9,_lib.EPI_RAW_STATS,out.ptr,out.cstride,0,bias.data_ptr(),None,None,stat.data_ptr(),mask.data_ptr(),S())
    elif mode == 'fwdws':
        need = lib.uh_conv_gemm_workspace(B,H,W,Cin,Cout,9)
        ws = torch.empty(need//4, device=DEV)
        f = lambda: call("uh_conv_gemm_ws", B,H,W,a.ptr,a.cstride,0,Cin,w.data_ptr(),Cout,9,_lib.EPI_RAW_STATS,out.ptr,out.cstride,0,bias.data_ptr(),None,None,stat.data_ptr(),mask.data_ptr(),ws.data_ptr(),need,S())
    else:
        dy = new_act(B,H,W,Cout,DEV); dy.buf.normal_()
        Sp = lib.uh_conv_wgrad_splits(B,H,W,Cout,Cin,9)
        part = torch.empty(Sp*Cout*9*Cin, device=DEV)
        f = lambda: call("uh_conv_wgrad", B,H,W,dy.ptr,dy.cstride,0,Cout,a.ptr,a.cstride,0,Cin,9,part.data_ptr(),S())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/reps
    fl = 2.0*B*H*W*Cin*Cout*9
    print(f"{name:12s} {mode} {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TFLOP/s", flush=True)
