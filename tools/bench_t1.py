"""micro-benchmark of the T = 1 GEMMs of the U-Net (transposed convs forward + input gradient) at B=32, 256^2"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import _lib
from unet_amd._lib import call, lib
from unet_amd.layout import new_act, pool_for
DEV = torch.device('cuda', 0)
S = lambda: torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 32
tot = 0.0
for k, (hs, w) in enumerate(((16, 512), (32, 256), (64, 128), (128, 64)), 1):
    cin = 2 * w
    a = new_act(B, hs, hs, cin, DEV); a.buf.normal_()
    cat = new_act(B, 2 * hs, 2 * hs, 2 * w, DEV)
    mask = pool_for(DEV).rowmask(B, hs, hs)
    wf = torch.randn(4 * w * cin, device=DEV).bfloat16()
    bias = torch.zeros(w, device=DEV)
    d4 = new_act(B, hs, hs, 4 * w, DEV); d4.buf.normal_()
    da = new_act(B, hs, hs, cin, DEV)
    Mp = B * (hs + 1) * (hs + 1)
    fwd = lambda: call("uh_conv_gemm", B, hs, hs, a.ptr, a.cstride, 0, cin, wf.data_ptr(), 4 * w, 1, _lib.EPI_CONVT, cat.ptr, cat.cstride, w, bias.data_ptr(), None, None, None, mask.data_ptr(), S())
    dgr = lambda: call("uh_conv_gemm", B, hs, hs, d4.ptr, d4.cstride, 0, 4 * w, wf.data_ptr(), cin, 1, _lib.EPI_MASKED, da.ptr, da.cstride, 0, None, None, None, None, mask.data_ptr(), S())
    Sp = lib.uh_conv_wgrad_splits(B, hs, hs, 4 * w, cin, 1)
    part = torch.empty(Sp * 4 * w * cin, device=DEV)
    wgr = lambda: call("uh_conv_wgrad", B, hs, hs, d4.ptr, d4.cstride, 0, 4 * w, a.ptr, a.cstride, 0, cin, 1, part.data_ptr(), S())
    own = new_act(B, 2 * hs, 2 * hs, w, DEV)          # the same forward into a buffer of its own (contiguous rows)
    fwc = lambda: call("uh_conv_gemm", B, hs, hs, a.ptr, a.cstride, 0, cin, wf.data_ptr(), 4 * w, 1, _lib.EPI_CONVT, own.ptr, own.cstride, 0, bias.data_ptr(), None, None, None, mask.data_ptr(), S())
    for nm, f in (("fwd", fwd), ("fwd_c", fwc), ("dgrad", dgr), ("wgrad", wgr)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * B * hs * hs * cin * 4 * w
        by = Mp * (cin + 4 * w) * 2
        tot += ms
        print(f"upconv{k} {nm:5s} M={Mp} K={cin if nm.startswith('fwd') else 4*w} N={4*w if nm.startswith('fwd') else cin}: {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TFLOP/s {by/ms/1e6:7.1f} GB/s", flush=True)
print(f"total {tot*1e3:.1f} us")
