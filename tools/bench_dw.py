"""Depthwise-block backward, stand-alone, at the LUNeXt level shapes of batch 32 / 256^2: the three passes
(uh_bn_res_relu_bwd_apply, uh_dwconv3x3_dgrad, uh_dwconv3x3_wgrad) against uh_dwblock_bwd; us per launch and the
traffic each moves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd._lib import call, lib
from unet_amd.layout import new_act, pool_for

DEV = torch.device("cuda", 0)
S = lambda: torch.cuda.current_stream().cuda_stream


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (B, H, W, C, cr) in [(32, 256, 256, 8, 3), (32, 128, 128, 64, 64), (32, 64, 64, 128, 128), (32, 32, 32, 256, 256),
                         (32, 16, 16, 512, 512)]:
    g = torch.Generator().manual_seed(1)
    acts = [new_act(B, H, W, C, DEV) for _ in range(7)]
    da, y, x, dy, dr, dx0, dx1 = acts
    for a in (da, x):
        a.fill_from_nchw(torch.randn(B, C, H, W, generator=g).to(DEV))
    y.fill_from_nchw(torch.randn(B, C, H, W, generator=g).to(DEV), half=True)
    wt = torch.randn(cr, 1, 3, 3, device=DEV)
    st = [torch.rand(C, device=DEV) + 0.5 for _ in range(4)]
    stp = [t.data_ptr() for t in st]
    coef = torch.randn(2, C, device=DEV) * 0.05
    mask = pool_for(DEV).rowmask(B, H, W)
    p0, p1 = lib.uh_dw_parts(B, H, W, C), lib.uh_dwblock_bwd_parts(B, H, W, C)
    wp0, wp1 = torch.empty(p0, 9, C, device=DEV), torch.empty(p1, 9, C, device=DEV)
    Mp = x.Mp
    t_apply = timed(lambda: call("uh_bn_res_relu_bwd_apply", Mp, C, da.ptr, da.cstride, da.coff, y.ptr, y.cstride, y.coff,
                                 x.ptr, x.cstride, x.coff, *stp, coef.data_ptr(), mask.data_ptr(), dy.ptr, dy.cstride,
                                 dy.coff, dr.ptr, dr.cstride, dr.coff, S()))
    t_dgrad = timed(lambda: call("uh_dwconv3x3_dgrad", B, H, W, C, dy.ptr, dy.cstride, dy.coff, wt.data_ptr(), cr, dr.ptr,
                                 dr.cstride, dr.coff, mask.data_ptr(), dx0.ptr, dx0.cstride, dx0.coff, S()))
    t_wgrad = timed(lambda: call("uh_dwconv3x3_wgrad", B, H, W, C, dy.ptr, dy.cstride, dy.coff, x.ptr, x.cstride, x.coff,
                                 wp0.data_ptr(), S()))
    t_fused = timed(lambda: call("uh_dwblock_bwd", B, H, W, C, da.ptr, da.cstride, da.coff, y.ptr, y.cstride, y.coff,
                                 x.ptr, x.cstride, x.coff, wt.data_ptr(), cr, *stp, coef.data_ptr(),
                                 dx1.ptr, dx1.cstride, dx1.coff, wp1.data_ptr(), S()))
    mb = Mp * C * 2 / 1e6
    print(f"B{B} {H}x{W} C{C}: tensor {mb:6.1f} MB | apply {t_apply:6.1f}  dgrad {t_dgrad:6.1f}  wgrad {t_wgrad:6.1f}  "
          f"sum {t_apply + t_dgrad + t_wgrad:6.1f} us | fused {t_fused:6.1f} us = {4 * mb / t_fused:5.2f} TB/s of its 4 tensors",
          flush=True)
