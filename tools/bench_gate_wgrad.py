"""micro-benchmark of LUNeXt's T = 1 weight gradients alone at batch 32, 256 x 256: the gates' dz^T [g | x]
(N = 3C, K = 2C) at the three GEMM levels, the strided and transposed convolutions' after space-to-depth"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd._lib import call, lib
from unet_amd.layout import new_act
DEV = torch.device('cuda', 0)
S = lambda: torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 32
cases = [("gate C=64", 128, 192, 128), ("gate C=128", 64, 384, 256), ("gate C=256", 32, 768, 512),
         ("g64 dz*g", 128, 192, 64, 192, 0, 128, 0), ("g64 dz3*x", 128, 64, 64, 192, 128, 128, 64),
         ("g128 dz*g", 64, 384, 128, 384, 0, 256, 0), ("g128 dz3*x", 64, 128, 128, 384, 256, 256, 128),
         ("g256 dz*g", 32, 768, 256, 768, 0, 512, 0), ("g256 dz3*x", 32, 256, 256, 768, 512, 512, 256),
         ("down2", 64, 128, 256), ("down3", 32, 256, 512), ("down4", 16, 512, 1024),
         ("up1", 16, 1024, 512), ("up2", 32, 512, 256), ("up3", 64, 256, 128), ("up4/down1", 128, 64, 64)]
for name, hs, N, K, *view in cases:
    nt, no, kt, ko = view if view else (N, 0, K, 0)          # the operands as column ranges of wider tensors
    dy = new_act(B, hs, hs, nt, DEV); dy.buf.normal_()
    a = new_act(B, hs, hs, kt, DEV); a.buf.normal_()
    Mp = B * (hs + 1) * (hs + 1)
    Sp = lib.uh_conv_wgrad_splits(B, hs, hs, N, K, 1)
    part = torch.empty(Sp * N * K, device=DEV)
    f = lambda: call("uh_conv_wgrad", B, hs, hs, dy.ptr, dy.cstride, no, N, a.ptr, a.cstride, ko, K, 1, part.data_ptr(), S())
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    by = Mp * (N + K) * 2 + Sp * N * K * 4
    print(f"{name:10s} M={Mp} N={N} K={K} splits={Sp}: {ms*1e3:7.1f} us {2.0*Mp*N*K/ms/1e9:7.1f} TFLOP/s "
          f"{by/ms/1e6:7.1f} GB/s of the operands read once + the partial blocks", flush=True)
