"""Two kernels of the backward pass side by side on two streams, at the level-0..2 shapes of batch 32 / 256^2: the
weight gradient of a 3x3 layer (second stream) beside (a) the BatchNorm-backward reduce + apply pair of the next layer,
(b) the input gradient of the same layer.  Prints each alone, both back to back, and both at once.
UNET_AMD_WGRAD_WGS=256 gives the weight gradient one workgroup per CU instead of two."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import _lib
from unet_amd._lib import call, lib
from unet_amd.layout import new_act, pool_for
from unet_amd.engine import side_stream_for

DEV = torch.device("cuda", 0)
main = torch.cuda.current_stream()
side = side_stream_for(DEV)


def span(fns_main, fns_side, n=10):
    """us per round of: everything in fns_main on the main stream, everything in fns_side on the second one"""
    def once():
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        for f in fns_side:
            f(side.cuda_stream)
        for f in fns_main:
            f(main.cuda_stream)
        main.wait_stream(side)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(n):
        once()
    b.record(main)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (B, H, W, C) in [(32, 256, 256, 64), (32, 128, 128, 128), (32, 64, 64, 256)]:
    a_in, dy, da, y, dyo, dx = (new_act(B, H, W, C, DEV) for _ in range(6))
    for t in (a_in, dy, da, y):
        t.buf.normal_()
    mask = pool_for(DEV).rowmask(B, H, W)
    Sp = lib.uh_conv_wgrad_splits(B, H, W, C, C, 9)
    part = torch.empty(Sp * C * 9 * C, device=DEV)
    w = torch.randn(C * 9 * C, device=DEV).bfloat16()
    st = [torch.rand(C, device=DEV) + 0.5 for _ in range(4)]
    stp = [t.data_ptr() for t in st]
    coef = torch.randn(2 * C, device=DEV) * 0.01
    parts = lib.uh_bn_bwd_parts(da.Mp, C)
    bp = torch.empty(parts * 2 * C, device=DEV)
    wgrad = lambda s: call("uh_conv_wgrad", B, H, W, dy.ptr, dy.cstride, 0, C, a_in.ptr, a_in.cstride, 0, C, 9, part.data_ptr(), s)
    reduce_ = lambda s: call("uh_bn_relu_bwd_reduce", da.Mp, C, da.ptr, da.cstride, 0, y.ptr, y.cstride, 0, *stp, bp.data_ptr(), s)
    apply_ = lambda s: call("uh_bn_relu_bwd_apply", da.Mp, C, da.ptr, da.cstride, 0, y.ptr, y.cstride, 0, *stp, coef.data_ptr(),
                            mask.data_ptr(), dyo.ptr, dyo.cstride, 0, s)
    dgrad = lambda s: call("uh_conv_gemm", B, H, W, dy.ptr, dy.cstride, 0, C, w.data_ptr(), C, 9, _lib.EPI_MASKED, dx.ptr,
                           dx.cstride, 0, None, None, None, None, mask.data_ptr(), s)
    print(f"C{C} {H}x{W}: dgrad || chain {span([dgrad], [reduce_, apply_]):6.1f}  dgrad || apply {span([dgrad], [apply_]):6.1f}  "
          f"dgrad || reduce {span([dgrad], [reduce_]):6.1f}  apply alone {span([apply_], []):6.1f}  reduce alone {span([reduce_], []):6.1f}", flush=True)
    t_w = span([], [wgrad])
    t_c = span([reduce_, apply_], [])
    t_d = span([dgrad], [])
    print(f"C{C} {H}x{W} splits {Sp}: wgrad {t_w:6.1f}  reduce+apply {t_c:6.1f}  dgrad {t_d:6.1f} | wgrad || chain {span([reduce_, apply_], [wgrad]):6.1f} "
          f"(serial {t_w + t_c:6.1f}) | wgrad || dgrad {span([dgrad], [wgrad]):6.1f} (serial {t_w + t_d:6.1f}) | "
          f"wgrad || dgrad+chain {span([dgrad, reduce_, apply_], [wgrad]):6.1f} (serial {t_w + t_d + t_c:6.1f})", flush=True)
