"""conv1_1 forward and weight gradient alone at batch 32, 3 x 256 x 256"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import _lib
from unet_amd._lib import call, lib
from unet_amd.layout import new_act
DEV = torch.device("cuda", 0)
S = lambda: torch.cuda.current_stream().cuda_stream
B, H, W, Cin, C = 32, 256, 256, 3, 64
x = torch.rand(B, Cin, H, W, device=DEV)
w = torch.randn(C, Cin, 3, 3, device=DEV) / 5
b = torch.randn(C, device=DEV)
out = new_act(B, H, W, C, DEV)
dy = new_act(B, H, W, C, DEV); dy.buf.normal_()
parts = lib.uh_conv_first_parts(B, H, W)
stat = torch.zeros(parts, 2, C, device=DEV)
wparts = lib.uh_conv_first_wgrad_parts(B, H, W)
part = torch.empty(wparts * C * Cin * 9, device=DEV)
fwd = lambda: call("uh_conv_first_fwd", B, H, W, Cin, C, x.data_ptr(), w.data_ptr(), b.data_ptr(), _lib.EPI_RAW_STATS, None, None,
                   out.ptr, out.cstride, out.coff, stat.data_ptr(), S())
wg = lambda: call("uh_conv_first_wgrad", B, H, W, Cin, C, x.data_ptr(), dy.ptr, dy.cstride, dy.coff, part.data_ptr(), S())
for name, f in (("fwd", fwd), ("wgrad", wg)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"conv1_1 {name}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us (parts {parts if name == 'fwd' else wparts})", flush=True)
