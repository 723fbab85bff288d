"""is the convolution kernel power-limited?  same launch with random and with all-zero operands, and hipBLASLt
(torch.matmul, bf16) on GEMMs of the same size for comparison"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import _lib
from unet_amd._lib import call, lib
from unet_amd.layout import new_act, pool_for
DEV = torch.device('cuda', 0)
S = lambda: torch.cuda.current_stream().cuda_stream
def timeit(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, (B, H, W, Cin, Cout) in {'conv3_2': (32, 64, 64, 256, 256), 'conv_up2_1': (32, 64, 64, 512, 256), 'conv2_2': (32, 128, 128, 128, 128)}.items():
    for fill in ('randn', 'zeros', 'small_ints'):
        a = new_act(B, H, W, Cin, DEV); out = new_act(B, H, W, Cout, DEV)
        w = torch.zeros(Cout * 9 * Cin, device=DEV).bfloat16()
        if fill == 'randn':
            a.buf.normal_(); w = torch.randn(Cout * 9 * Cin, device=DEV).bfloat16()
        elif fill == 'small_ints':
            a.buf.copy_(torch.randint(0, 2, a.buf.shape, device=DEV).bfloat16()); w = torch.randint(0, 2, (Cout * 9 * Cin,), device=DEV).bfloat16()
        mask = pool_for(DEV).rowmask(B, H, W)
        parts = lib.uh_conv_gemm_mtiles(B, H, W, Cout)
        stat = torch.zeros(parts * 2 * Cout, device=DEV); bias = torch.zeros(Cout, device=DEV)
        need = lib.uh_conv_gemm_workspace(B, H, W, Cin, Cout, 9); ws = torch.empty(need // 4, device=DEV)
        f = lambda: call("uh_conv_gemm_ws", B, H, W, a.ptr, a.cstride, 0, Cin, w.data_ptr(), Cout, 9, _lib.EPI_RAW_STATS, out.ptr, out.cstride, 0, bias.data_ptr(), None, None, stat.data_ptr(), mask.data_ptr(), ws.data_ptr(), need, S())
        ms = timeit(f)
        print(f"{name:11s} {fill:10s} {ms*1e3:8.1f} us {2.0*B*H*W*Cin*Cout*9/ms/1e9:8.1f} TFLOP/s", flush=True)
for (M, N, K) in [(135200, 256, 2304), (8192, 8192, 8192), (16384, 4096, 4608)]:
    for fill in ('randn', 'zeros'):
        x = torch.zeros(M, K, device=DEV, dtype=torch.bfloat16); y = torch.zeros(N, K, device=DEV, dtype=torch.bfloat16)
        if fill == 'randn': x.normal_(); y.normal_()
        o = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        f = lambda: torch.matmul(x, y.t(), out=o)
        ms = timeit(f)
        print(f"hipBLASLt {M}x{N}x{K} {fill:6s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
