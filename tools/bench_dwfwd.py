"""depthwise 3x3 forward (train mode: fp16 output + BatchNorm partial sums) alone at the LUNeXt level shapes of batch 32 / 256^2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import _lib
from unet_amd._lib import call, lib
from unet_amd.layout import new_act, pool_for
DEV = torch.device("cuda", 0)
S = lambda: torch.cuda.current_stream().cuda_stream
for (B, H, W, C, cr) in [(32, 256, 256, 8, 3), (32, 128, 128, 64, 64), (32, 64, 64, 128, 128), (32, 32, 32, 256, 256), (32, 16, 16, 512, 512)]:
    x, y = new_act(B, H, W, C, DEV), new_act(B, H, W, C, DEV)
    x.fill_from_nchw(torch.randn(B, C, H, W).to(DEV))
    wt, bias = torch.randn(cr, 1, 3, 3, device=DEV), torch.randn(cr, device=DEV)
    mask = pool_for(DEV).rowmask(B, H, W)
    parts = lib.uh_dwconv3x3_fwd_parts(B, H, W, C)
    stat = torch.empty(parts * 2 * C, device=DEV)
    f = lambda: call("uh_dwconv3x3_fwd", B, H, W, C, x.ptr, x.cstride, x.coff, wt.data_ptr(), cr, bias.data_ptr(), _lib.EPI_RAW_STATS,
                     None, None, mask.data_ptr(), y.ptr, y.cstride, y.coff, stat.data_ptr(), S())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = x.Mp * C * 2 / 1e6
    print(f"B{B} {H}x{W} C{C}: {us:6.1f} us, {2 * mb / us:5.2f} TB/s of in + out", flush=True)
