"""Per-kernel parity: every C-ABI entry point of libunet_hip.so against the same operation done
with plain PyTorch fp32 ops on the CPU (the ATen calls the reference's modules dispatch).
Inputs are rounded to bf16 first, so the only differences left are fp32 summation order and the
final bf16 rounding of stored activations; tolerances below are stated per test."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from conftest import gpu_present            # noqa: E402

if not gpu_present():
    pytest.skip("needs a GPU", allow_module_level=True)

from unet_amd import _lib                    # noqa: E402
from unet_amd._lib import call, lib          # noqa: E402
from unet_amd.layout import new_act, pool_for, mp_rows   # noqa: E402

DEV = torch.device("cuda", 0)


def S():
    return torch.cuda.current_stream().cuda_stream


def bf(t):
    return t.to(torch.bfloat16).float()


def act_from(t, cstride=None, coff=0, half=False):
    B, C, H, W = t.shape
    a = new_act(B, H, W, cstride or C, DEV, cstride=cstride or C).window(coff, C)
    a.fill_from_nchw(t.to(DEV), half=half)
    return a


def rowmask(B, H, W):
    return pool_for(DEV).rowmask(B, H, W)


def close(got, want, rel, what=""):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    assert err <= rel * ref + 1e-6, f"{what}: max err {err:.4g} vs ref max {ref:.4g} (rel tol {rel})"


def pads_are_zero(a):
    B, H, W = a.B, a.H, a.W
    rows = a.padded_rows().float().view(B, H + 1, W + 1, a.C)
    assert rows[:, H, :, :].abs().max().item() == 0
    assert rows[:, :, W, :].abs().max().item() == 0
    g = a.guard
    assert a.buf[:g].abs().max().item() == 0 and a.buf[g + a.Mp:].abs().max().item() == 0, "guard rows were written"


def pack_conv(w):          # OIHW -> [co][t][ci] bf16
    co, ci = w.shape[:2]
    dst = torch.empty(co * 9 * ci, dtype=torch.bfloat16, device=DEV)
    call("uh_pack_weights", w.data_ptr(), dst.data_ptr(), co, 9, ci, ci * 9, 1, 9, 9 * ci, ci, 0, S())
    return dst


CONV_SHAPES = [  # B, H, W, Cin, Cout
    (2, 16, 16, 64, 64),       # 256x64 tile config
    (1, 8, 24, 128, 128),      # 128x128 tile config, non-square
    (3, 4, 4, 64, 192),        # N not a multiple of 128, tiny image (tiles span several images)
    (1, 32, 32, 192, 128),     # three K chunks
    (2, 2, 2, 128, 256),       # deepest-level shape of a 32x32 input
    (2, 24, 24, 128, 256),     # 256-wide tiles, several M tiles
    (1, 16, 16, 64, 512),      # two N tiles of 256
]


def test_pack_weights_multi_matches_single():
    """one launch over a device-resident descriptor table == one uh_pack_weights call per tensor"""
    g = torch.Generator().manual_seed(3)
    specs = []   # (src, N, T, K, sn, st, sk, on, ot, reverse)
    w1 = torch.randn(128, 64, 3, 3, generator=g).to(DEV)
    w2 = torch.randn(256, 128, 2, 2, generator=g).to(DEV)          # transposed-conv layout [cin][w][2][2]
    specs.append((w1, 128, 9, 64, 64 * 9, 1, 9, 9 * 64, 64, 0))
    specs.append((w1, 64, 9, 128, 9, 1, 64 * 9, 9 * 128, 128, 1))
    specs.append((w2, 128, 4, 256, 4, 1, 128 * 4, 256, 128 * 256, 0))
    singles, rows, multis = [], [], []
    for (src, N, T, K, sn, st, sk, on, ot, rev) in specs:
        one = torch.zeros(N * T * K, dtype=torch.bfloat16, device=DEV)
        call("uh_pack_weights", src.data_ptr(), one.data_ptr(), N, T, K, sn, st, sk, on, ot, rev, S())
        many = torch.zeros(N * T * K, dtype=torch.bfloat16, device=DEV)
        rows.append([src.data_ptr(), many.data_ptr(), N, T, K, rev, sn, st, sk, on, ot])
        singles.append(one)
        multis.append(many)
    desc = torch.tensor(rows, dtype=torch.int64).to(DEV)
    call("uh_pack_weights_multi", desc.data_ptr(), len(rows), 7, S())
    torch.cuda.synchronize()
    for one, many in zip(singles, multis):
        assert torch.equal(one, many)
    with pytest.raises(_lib.HipKernelError):
        call("uh_pack_weights_multi", None, 1, 1, S())


@pytest.mark.parametrize("B,H,W,Cin,Cout", CONV_SHAPES)
def test_conv_gemm_raw_stats(B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin)
    x = bf(torch.randn(B, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    b = torch.randn(Cout, generator=g)
    a = act_from(x)
    out = new_act(B, H, W, Cout, DEV)
    wd = w.to(DEV)
    parts = lib.uh_conv_gemm_mtiles(B, H, W, Cout)
    stat = torch.zeros(parts, 2, Cout, device=DEV)
    bd = b.to(DEV)
    pk = pack_conv(wd)          # keep the packed weights alive until the kernel has run
    call("uh_conv_gemm", B, H, W, a.ptr, a.cstride, a.coff, Cin, pk.data_ptr(), Cout, 9, _lib.EPI_RAW_STATS,
         out.ptr, out.cstride, out.coff, bd.data_ptr(), None, None, stat.data_ptr(), rowmask(B, H, W).data_ptr(), S())
    torch.cuda.synchronize()
    want = F.conv2d(x, w, b, padding=1)
    close(out.to_nchw(half=True), want, 2e-3, "conv output (fp16 store)")
    pads_are_zero(out)
    s = stat.sum(0).cpu()
    close(s[0], want.sum((0, 2, 3)), 2e-3, "sum y")
    close(s[1], (want * want).sum((0, 2, 3)), 2e-3, "sum y^2")


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 32, 64, 128), (1, 16, 16, 128, 64), (2, 48, 48, 64, 256),
                                            (1, 16, 16, 64, 64), (3, 40, 40, 128, 256), (2, 16, 16, 192, 512),
                                            (2, 16, 16, 512, 512), (1, 32, 32, 1024, 256), (3, 16, 16, 512, 1024)])
def test_conv_gemm_even_cuts(B, H, W, Cin, Cout):
    """uh_conv_gemm_ws: workgroups that own pieces of tiles + the fix-up launch give the whole-tile result.  (The
    last three cases have weights of more than 3 MB: their cut launches number the tiles N tile first.)"""
    g = torch.Generator().manual_seed(B * 77 + H + Cin)
    x = bf(torch.randn(B, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    b = torch.randn(Cout, generator=g)
    a = act_from(x)
    wd, bd = w.to(DEV), b.to(DEV)
    pk = pack_conv(wd)
    parts = lib.uh_conv_gemm_mtiles(B, H, W, Cout)
    need = lib.uh_conv_gemm_workspace(B, H, W, Cin, Cout, 9)
    assert need > 0 and lib.uh_conv_gemm_workspace(B, H, W, Cin, Cout, 1) == 0
    ws = torch.full((need // 4,), float("nan"), device=DEV)        # stale contents must not matter
    outs, stats = [], []
    for use_ws in (False, True):
        out = new_act(B, H, W, Cout, DEV)
        stat = torch.zeros(parts, 2, Cout, device=DEV)
        call("uh_conv_gemm_ws", B, H, W, a.ptr, a.cstride, a.coff, Cin, pk.data_ptr(), Cout, 9, _lib.EPI_RAW_STATS,
             out.ptr, out.cstride, out.coff, bd.data_ptr(), None, None, stat.data_ptr(), rowmask(B, H, W).data_ptr(),
             ws.data_ptr() if use_ws else None, need if use_ws else 0, S())
        torch.cuda.synchronize()
        pads_are_zero(out)
        outs.append(out.to_nchw(half=True))
        stats.append(stat.cpu())
    want = F.conv2d(x, w, b, padding=1)
    close(outs[1], want, 2e-3, "conv output, even cuts")
    close(outs[1], outs[0], 1e-3, "even cuts vs whole tiles")
    close(stats[1], stats[0], 1e-4, "BatchNorm partial sums, even cuts vs whole tiles")
    # input gradient mode through the same path
    out = new_act(B, H, W, Cout, DEV)
    call("uh_conv_gemm_ws", B, H, W, a.ptr, a.cstride, a.coff, Cin, pk.data_ptr(), Cout, 9, _lib.EPI_MASKED,
         out.ptr, out.cstride, out.coff, None, None, None, None, rowmask(B, H, W).data_ptr(), ws.data_ptr(), need, S())
    torch.cuda.synchronize()
    close(out.to_nchw(), F.conv2d(x, w, None, padding=1), 1e-2, "masked mode, even cuts")
    pads_are_zero(out)


def test_conv_gemm_affine_relu_and_views():
    """eval-mode epilogue, reading from and writing into channel windows of wider buffers."""
    B, H, W, Cin, Cout = 2, 8, 8, 64, 128
    g = torch.Generator().manual_seed(5)
    x = bf(torch.randn(B, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / 24)
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    a = act_from(x, cstride=192, coff=64)
    out = new_act(B, H, W, 256, DEV, cstride=256).window(128, Cout)
    wd, sc, sh = w.to(DEV), scale.to(DEV), shift.to(DEV)
    pk = pack_conv(wd)
    call("uh_conv_gemm", B, H, W, a.ptr, a.cstride, a.coff, Cin, pk.data_ptr(), Cout, 9,
         _lib.EPI_AFFINE_RELU, out.ptr, out.cstride, out.coff, None, sc.data_ptr(), sh.data_ptr(), None,
         rowmask(B, H, W).data_ptr(), S())
    torch.cuda.synchronize()
    want = F.relu(F.conv2d(x, w, None, padding=1) * scale[None, :, None, None] + shift[None, :, None, None])
    close(out.to_nchw(), want, 1e-2, "affine+relu")
    pads_are_zero(out)
    assert out.buf[:, :128].abs().max().item() == 0, "wrote outside its channel window"


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 16, 64, 64), (1, 8, 8, 128, 64), (2, 4, 8, 64, 128)])
def test_conv_dgrad(B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(17)
    dy = bf(torch.randn(B, Cout, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / 24)
    wd = w.to(DEV)
    pk = torch.empty(Cin * 9 * Cout, dtype=torch.bfloat16, device=DEV)
    call("uh_pack_weights", wd.data_ptr(), pk.data_ptr(), Cin, 9, Cout, 9, 1, Cin * 9, 9 * Cout, Cout, 1, S())
    a = act_from(dy)
    out = new_act(B, H, W, Cin, DEV)
    call("uh_conv_gemm", B, H, W, a.ptr, a.cstride, a.coff, Cout, pk.data_ptr(), Cin, 9, _lib.EPI_MASKED,
         out.ptr, out.cstride, out.coff, None, None, None, None, rowmask(B, H, W).data_ptr(), S())
    torch.cuda.synchronize()
    want = F.conv_transpose2d(dy, w, padding=1)          # = input gradient of conv2d(pad=1)
    close(out.to_nchw(), want, 1e-2, "dgrad")
    pads_are_zero(out)


@pytest.mark.parametrize("B,h,w,Cin,Cout", [(2, 4, 4, 128, 64), (1, 2, 6, 64, 64), (2, 8, 8, 256, 128),
                                            # at least 512 tiles of 256 x 256: the sixteen-wave kernel
                                            (8, 128, 128, 128, 64), (6, 150, 120, 64, 128)])
def test_conv_transpose_fwd(B, h, w, Cin, Cout):
    g = torch.Generator().manual_seed(23)
    x = bf(torch.randn(B, Cin, h, w, generator=g))
    wt = bf(torch.randn(Cin, Cout, 2, 2, generator=g) / Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    wd, bd = wt.to(DEV), b.to(DEV)
    pk = torch.empty(4 * Cout * Cin, dtype=torch.bfloat16, device=DEV)
    call("uh_pack_weights", wd.data_ptr(), pk.data_ptr(), Cout, 4, Cin, 4, 1, Cout * 4, Cin, Cout * Cin, 0, S())
    a = act_from(x)
    cat = new_act(B, 2 * h, 2 * w, 2 * Cout, DEV, cstride=2 * Cout)
    cat.buf.fill_(7.0)                                  # poison: every row of the window must be written
    out = cat.window(Cout, Cout)
    call("uh_conv_gemm", B, h, w, a.ptr, a.cstride, a.coff, Cin, pk.data_ptr(), 4 * Cout, 1, _lib.EPI_CONVT,
         out.ptr, out.cstride, out.coff, bd.data_ptr(), None, None, None, rowmask(B, h, w).data_ptr(), S())
    torch.cuda.synchronize()
    want = F.conv_transpose2d(x, wt, b, stride=2)
    close(out.to_nchw(), want, 1e-2, "convT")
    rows = out.padded_rows().float().view(B, 2 * h + 1, 2 * w + 1, Cout)
    assert rows[:, 2 * h].abs().max().item() == 0 and rows[:, :, 2 * w].abs().max().item() == 0
    assert (cat.window(0, Cout).padded_rows().float() == 7.0).all(), "wrote outside its channel window"


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 16, 64, 64), (1, 8, 24, 128, 64), (3, 4, 4, 64, 128),
                                            (2, 2, 2, 128, 128), (1, 40, 40, 64, 64),
                                            # widths that are multiples of 64 walk down column strips
                                            (1, 6, 128, 64, 64), (2, 3, 256, 128, 64), (3, 5, 128, 64, 128),
                                            (2, 40, 256, 64, 64), (2, 10, 64, 64, 64), (1, 1, 64, 64, 64),
                                            (5, 64, 64, 128, 128),
                                            # wider than 138 and no multiple of 64: three separate bands per stage
                                            (1, 4, 150, 64, 64), (2, 3, 139, 64, 128), (1, 5, 138, 64, 64)])
def test_conv_wgrad(B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(31)
    x = bf(torch.randn(B, Cin, H, W, generator=g))
    dy = bf(torch.randn(B, Cout, H, W, generator=g))
    a, d = act_from(x), act_from(dy)
    Sp = lib.uh_conv_wgrad_splits(B, H, W, Cout, Cin, 9)
    assert lib.uh_conv_wgrad_workspace(B, H, W, Cout, Cin, 9) == Sp * Cout * 9 * Cin * 4
    part = torch.empty(Sp * Cout * 9 * Cin, device=DEV)
    call("uh_conv_wgrad", B, H, W, d.ptr, d.cstride, d.coff, Cout, a.ptr, a.cstride, a.coff, Cin, 9, part.data_ptr(), S())
    dw = torch.empty(Cout, Cin, 3, 3, device=DEV)
    call("uh_wgrad_reduce", part.data_ptr(), Sp, Cout, 9, Cin, 9 * Cin, Cin, Cin * 9, 1, 9, dw.data_ptr(), 0, S())
    torch.cuda.synchronize()
    want = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy, padding=1)
    close(dw, want, 2e-3, "wgrad (fp32 accumulate of bf16 products)")


@pytest.mark.parametrize("B,h,w,Cin,Cout", [(2, 4, 6, 128, 64), (2, 4, 6, 64, 64), (3, 10, 9, 256, 128),
                                            (1, 16, 16, 128, 32), (8, 100, 100, 256, 128), (8, 130, 130, 256, 64)])
def test_conv_wgrad_1x1_and_transposed(B, h, w, Cin, Cout):
    """T=1 weight gradient as used for the transposed conv: space-to-depth then GEMM.  (The last case is large enough,
    21 GFLOP with N = 4 Cout and K = Cin multiples of 128, for the 128 x 128-block kernel; the others take 64 x 64
    blocks.  The last case's input gradient has 537 tiles of 256 x 256: the sixteen-wave GEMM kernel.)"""
    g = torch.Generator().manual_seed(37)
    x = bf(torch.randn(B, Cin, h, w, generator=g))
    dy = bf(torch.randn(B, Cout, 2 * h, 2 * w, generator=g))
    wt = bf(torch.randn(Cin, Cout, 2, 2, generator=g) / 8)
    a, d = act_from(x), act_from(dy, cstride=2 * Cout, coff=Cout)
    d4 = new_act(B, h, w, 4 * Cout, DEV)
    call("uh_space_to_depth2", B, h, w, Cout, d.ptr, d.cstride, d.coff, d4.ptr, d4.cstride, d4.coff, S())
    Sp = lib.uh_conv_wgrad_splits(B, h, w, 4 * Cout, Cin, 1)
    part = torch.empty(Sp * 4 * Cout * Cin, device=DEV)
    call("uh_conv_wgrad", B, h, w, d4.ptr, d4.cstride, d4.coff, 4 * Cout, a.ptr, a.cstride, a.coff, Cin, 1,
         part.data_ptr(), S())
    dw = torch.empty(Cin, Cout, 2, 2, device=DEV)
    call("uh_wgrad_reduce", part.data_ptr(), Sp, Cout, 4, Cin, Cin, Cout * Cin, 4, 1, Cout * 4, dw.data_ptr(), 0, S())
    # input gradient through the same d4
    wd = wt.to(DEV)
    pk = torch.empty(Cin * 4 * Cout, dtype=torch.bfloat16, device=DEV)
    call("uh_pack_weights", wd.data_ptr(), pk.data_ptr(), Cin, 4, Cout, Cout * 4, 1, 4, 4 * Cout, Cout, 0, S())
    dx = new_act(B, h, w, Cin, DEV)
    call("uh_conv_gemm", B, h, w, d4.ptr, d4.cstride, d4.coff, 4 * Cout, pk.data_ptr(), Cin, 1, _lib.EPI_MASKED,
         dx.ptr, dx.cstride, dx.coff, None, None, None, None, rowmask(B, h, w).data_ptr(), S())
    torch.cuda.synchronize()
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    F.conv_transpose2d(xr, wr, None, stride=2).backward(dy)
    close(dw, wr.grad, 2e-3, "convT wgrad")
    close(dx.to_nchw(), xr.grad, 1e-2, "convT dgrad")
    pads_are_zero(dx)
    pads_are_zero(d4)
    # depth_to_space is the inverse scatter
    back = new_act(B, 2 * h, 2 * w, Cout, DEV)
    back.buf.fill_(3.0)
    back.buf[:back.guard] = 0
    back.buf[back.guard + back.Mp:] = 0
    call("uh_depth_to_space2", B, h, w, Cout, d4.ptr, d4.cstride, d4.coff, back.ptr, back.cstride, back.coff, S())
    torch.cuda.synchronize()
    close(back.to_nchw(), dy, 0.0, "depth_to_space(space_to_depth(x)) == x")
    pads_are_zero(back)


@pytest.mark.parametrize("Cin", [1, 2, 3, 4])
def test_conv_first(Cin):
    B, H, W, Cout = 2, 16, 24, 64
    g = torch.Generator().manual_seed(41 + Cin)
    x = torch.rand(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 5
    b = torch.randn(Cout, generator=g)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    out = new_act(B, H, W, Cout, DEV)
    parts = lib.uh_conv_first_parts(B, H, W)
    stat = torch.zeros(parts, 2, Cout, device=DEV)
    call("uh_conv_first_fwd", B, H, W, Cin, Cout, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), _lib.EPI_RAW_STATS,
         None, None, out.ptr, out.cstride, out.coff, stat.data_ptr(), S())
    torch.cuda.synchronize()
    want = F.conv2d(x, w, b, padding=1)
    close(out.to_nchw(half=True), want, 1e-3, "first conv (fp16 store)")
    pads_are_zero(out)
    s = stat.sum(0).cpu()
    close(s[0], want.sum((0, 2, 3)), 1e-4, "sum y (fp32 path)")
    close(s[1], (want * want).sum((0, 2, 3)), 1e-4, "sum y^2")
    # eval epilogue
    scale, shift = (torch.rand(Cout, generator=g) + 0.5), torch.randn(Cout, generator=g)
    sc, sh = scale.to(DEV), shift.to(DEV)
    call("uh_conv_first_fwd", B, H, W, Cin, Cout, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), _lib.EPI_AFFINE_RELU,
         sc.data_ptr(), sh.data_ptr(), out.ptr, out.cstride, out.coff, None, S())
    torch.cuda.synchronize()
    want2 = F.relu(F.conv2d(x, w, None, padding=1) * scale[None, :, None, None] + shift[None, :, None, None])
    close(out.to_nchw(), want2, 6e-3, "first conv eval")
    # weight gradient
    dy = bf(torch.randn(B, Cout, H, W, generator=g))
    d = act_from(dy)
    wparts = lib.uh_conv_first_wgrad_parts(B, H, W)
    part = torch.empty(wparts * Cout * Cin * 9, device=DEV)
    call("uh_conv_first_wgrad", B, H, W, Cin, Cout, xd.data_ptr(), d.ptr, d.cstride, d.coff, part.data_ptr(), S())
    dw = torch.empty(Cout, Cin, 3, 3, device=DEV)
    call("uh_wgrad_reduce", part.data_ptr(), wparts, Cout, 1, Cin * 9, Cin * 9, 0, Cin * 9, 0, 1, dw.data_ptr(), 0, S())
    torch.cuda.synchronize()
    close(dw, torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy, padding=1), 1e-4, "first conv wgrad")


@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 64), (3, 4, 6, 256), (1, 8, 8, 1024)])
def test_batchnorm_relu_forward_backward(B, H, W, C):
    g = torch.Generator().manual_seed(43)
    y = (torch.randn(B, C, H, W, generator=g) * 2 + 0.5).half().float()   # BatchNorm input is stored as fp16
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    rm0, rv0 = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    # statistics come from the conv epilogue; emulate a two-row partial here
    ys = y.double()
    part = torch.zeros(2, 2, C, dtype=torch.float32)
    part[0, 0] = ys[:1].sum((0, 2, 3)).float() if B > 1 else ys.sum((0, 2, 3)).float()
    part[0, 1] = (ys[:1] ** 2).sum((0, 2, 3)).float() if B > 1 else (ys ** 2).sum((0, 2, 3)).float()
    if B > 1:
        part[1, 0] = ys[1:].sum((0, 2, 3)).float()
        part[1, 1] = (ys[1:] ** 2).sum((0, 2, 3)).float()
    pd = part.to(DEV)
    gd, bd, rm, rv = gamma.to(DEV), beta.to(DEV), rm0.to(DEV), rv0.to(DEV)
    nbt = torch.tensor(5, dtype=torch.int64, device=DEV)
    st = torch.empty(4, C, device=DEV)
    call("uh_bn_finalize", pd.data_ptr(), 2, C, C, B * H * W, gd.data_ptr(), bd.data_ptr(), 1e-5, 0.1, st[0].data_ptr(),
         st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), S())
    ya = act_from(y, half=True)
    a = new_act(B, H, W, C, DEV)
    mask = rowmask(B, H, W)
    call("uh_bn_relu_apply", ya.Mp, C, ya.ptr, ya.cstride, ya.coff, st[2].data_ptr(), st[3].data_ptr(), mask.data_ptr(),
         a.ptr, a.cstride, a.coff, S())
    torch.cuda.synchronize()
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm0); bn.running_var.copy_(rv0)
        bn.num_batches_tracked.fill_(5)
    yr = y.clone().requires_grad_(True)
    ar = F.relu(bn(yr))
    close(a.to_nchw(), ar, 1e-2, "bn+relu")
    pads_are_zero(a)
    close(rm, bn.running_mean, 1e-5, "running_mean")
    close(rv, bn.running_var, 1e-5, "running_var")
    assert nbt.item() == 6
    # backward
    da = bf(torch.randn(B, C, H, W, generator=g))
    ar.backward(da)
    dact = act_from(da)
    parts = lib.uh_bn_bwd_parts(ya.Mp, C)
    bp = torch.empty(parts, 2, C, device=DEV)
    call("uh_bn_relu_bwd_reduce", ya.Mp, C, dact.ptr, dact.cstride, dact.coff, ya.ptr, ya.cstride, ya.coff,
         st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), bp.data_ptr(), S())
    dg, db, coef = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(2, C, device=DEV)
    call("uh_bn_bwd_finalize", bp.data_ptr(), parts, C, C, B * H * W, dg.data_ptr(), db.data_ptr(), coef.data_ptr(), 0, S())
    dy = new_act(B, H, W, C, DEV)
    call("uh_bn_relu_bwd_apply", ya.Mp, C, dact.ptr, dact.cstride, dact.coff, ya.ptr, ya.cstride, ya.coff,
         st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), coef.data_ptr(), mask.data_ptr(),
         dy.ptr, dy.cstride, dy.coff, S())
    torch.cuda.synchronize()
    close(dg, bn.weight.grad, 1e-4, "dgamma")
    close(db, bn.bias.grad, 1e-4, "dbeta")
    close(dy.to_nchw(), yr.grad, 1e-2, "dy")
    pads_are_zero(dy)
    # eval fold
    cb = torch.randn(C, generator=g).to(DEV)
    fs = torch.empty(2, C, device=DEV)
    call("uh_bn_fold_eval", C, gd.data_ptr(), bd.data_ptr(), rm.data_ptr(), rv.data_ptr(), cb.data_ptr(), 1e-5,
         fs[0].data_ptr(), fs[1].data_ptr(), S())
    torch.cuda.synchronize()
    sc = gamma / torch.sqrt(rv.cpu() + 1e-5)
    close(fs[0], sc, 1e-6, "fold scale")
    close(fs[1], beta + (cb.cpu() - rm.cpu()) * sc, 1e-5, "fold shift")


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 64), (1, 4, 12, 128), (3, 2, 2, 512)])
def test_maxpool(B, H, W, C):
    g = torch.Generator().manual_seed(47)
    # coarse values so that windows contain exact ties: the FIRST maximum (row-major) takes the gradient
    x = (torch.randint(0, 4, (B, C, H, W), generator=g).float() / 2)
    a = act_from(x, cstride=2 * C, coff=0)
    p = new_act(B, H // 2, W // 2, C, DEV)
    call("uh_maxpool2_fwd", B, H, W, C, a.ptr, a.cstride, a.coff, p.ptr, p.cstride, p.coff, S())
    xr = x.clone().requires_grad_(True)
    pr = F.max_pool2d(xr, 2, 2)
    torch.cuda.synchronize()
    close(p.to_nchw(), pr, 0.0, "maxpool")
    pads_are_zero(p)
    dp = bf(torch.randn(B, C, H // 2, W // 2, generator=g))
    add = bf(torch.randn(B, C, H, W, generator=g))
    pr.backward(dp)
    dpa, adda = act_from(dp), act_from(add, cstride=2 * C, coff=0)
    da = new_act(B, H, W, C, DEV)
    da.buf[da.guard:da.guard + da.Mp] = 9.0
    call("uh_maxpool2_bwd", B, H, W, C, a.ptr, a.cstride, a.coff, dpa.ptr, dpa.cstride, dpa.coff, adda.ptr,
         adda.cstride, adda.coff, da.ptr, da.cstride, da.coff, S())
    torch.cuda.synchronize()
    close(da.to_nchw(), bf(xr.grad + add), 4e-3, "maxpool bwd + skip add")
    pads_are_zero(da)


def test_colsum_and_head():
    B, H, W, C = 2, 8, 12, 64
    g = torch.Generator().manual_seed(53)
    x = bf(torch.randn(B, C, H, W, generator=g))
    a = act_from(x, cstride=128, coff=64)
    parts = lib.uh_colsum_parts(a.Mp, C)
    cp = torch.empty(parts, C, device=DEV)
    call("uh_colsum", a.Mp, C, a.ptr, a.cstride, a.coff, cp.data_ptr(), S())
    out = torch.empty(C, device=DEV)
    call("uh_colsum_finalize", cp.data_ptr(), parts, C, 1, out.data_ptr(), 0, S())
    torch.cuda.synchronize()
    close(out, x.sum((0, 2, 3)), 1e-5, "colsum")
    for OC in (1, 2):
        w = torch.randn(OC, C, generator=g) / 8
        b = torch.randn(OC, generator=g)
        wd, bd = w.to(DEV), b.to(DEV)
        logits = torch.empty(B, OC, H, W, device=DEV)
        call("uh_head_fwd", B, H, W, C, OC, a.ptr, a.cstride, a.coff, wd.data_ptr(), bd.data_ptr(), logits.data_ptr(), S())
        torch.cuda.synchronize()
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        br = b.clone().requires_grad_(True)
        lr = F.conv2d(xr, wr.view(OC, C, 1, 1), br)
        close(logits, lr, 1e-5, "head fwd (fp32 out)")
        dl = torch.randn(B, OC, H, W, generator=g)
        lr.backward(dl)
        dld = dl.to(DEV)
        da = new_act(B, H, W, C, DEV)
        hp_parts = lib.uh_head_bwd_parts(B, H, W, C)
        hp = torch.empty(hp_parts, OC, C + 1, device=DEV)
        call("uh_head_bwd", B, H, W, C, OC, a.ptr, a.cstride, a.coff, wd.data_ptr(), dld.data_ptr(), da.ptr, da.cstride,
             da.coff, hp.data_ptr(), S())
        hw = torch.empty(OC * (C + 1), device=DEV)
        call("uh_colsum_finalize", hp.data_ptr(), hp_parts, OC * (C + 1), 1, hw.data_ptr(), 0, S())
        torch.cuda.synchronize()
        hw = hw.view(OC, C + 1).cpu()
        close(hw[:, :C], wr.grad, 1e-4, "head dW")
        close(hw[:, C], br.grad, 1e-4, "head db")
        close(da.to_nchw(), xr.grad, 1e-2, "head dX")
        pads_are_zero(da)


def test_bad_arguments_are_reported():
    a = new_act(1, 4, 4, 64, DEV)
    with pytest.raises(_lib.HipKernelError, match="multiple of 64"):
        call("uh_conv_gemm", 1, 4, 4, a.ptr, 64, 0, 48, a.ptr, 64, 9, 0, a.ptr, 64, 0, None, None, None, a.ptr,
             a.ptr, S())
    with pytest.raises(_lib.HipKernelError, match="even"):
        call("uh_maxpool2_fwd", 1, 3, 4, 64, a.ptr, 64, 0, a.ptr, 64, 0, S())


def test_adam_step_matches_torch():
    """uh_adam_step over a flat buffer == torch.optim.Adam (defaults of training_utils.py:107), three steps"""
    g = torch.Generator().manual_seed(21)
    n = 4096 * 3 + 8
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * s for s in (1.0, 0.1, 3.0)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3)
    pd, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step, gr in enumerate(grads, start=1):
        ref.grad = gr.clone()
        opt.step()
        gd = gr.to(DEV)
        call("uh_adam_step", n, pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), 1e-3, 0.9, 0.999, 1e-8, 0.0,
             step, S())
        torch.cuda.synchronize()
        torch.testing.assert_close(pd.cpu(), ref.detach(), rtol=1e-6, atol=1e-7)
    with pytest.raises(_lib.HipKernelError):
        call("uh_adam_step", n + 1, pd.data_ptr(), pd.data_ptr(), m.data_ptr(), v.data_ptr(), 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, S())


@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 64), (1, 20, 12, 8), (3, 5, 7, 24), (2, 33, 31, 128), (1, 48, 48, 512)])
def test_depthwise_conv_forward_backward(B, H, W, C, monkeypatch):
    """uh_dwconv3x3_fwd (both epilogues) / _dgrad / _wgrad against F.conv2d(groups=C) on the same bf16-rounded
    operands; shapes whose rows do not fill the 16-row strips and whose width does not fill a workgroup."""
    g = torch.Generator().manual_seed(B * 100 + H + C)
    x = bf(torch.randn(B, C, H, W, generator=g))
    w = torch.randn(C, 1, 3, 3, generator=g) / 3
    bias = torch.randn(C, generator=g)
    xa = act_from(x)
    wd = w.to(DEV)                       # the kernels read the module's own [C][1][3][3] layout
    mask = rowmask(B, H, W)
    parts = lib.uh_dw_parts(B, H, W, C)
    assert parts > 0 and lib.uh_dw_parts(B, H, W, C + 1) == 0
    fparts = lib.uh_dwconv3x3_fwd_parts(B, H, W, C)
    assert fparts > 0 and lib.uh_dwconv3x3_fwd_parts(B, H, W, C + 1) == 0
    stat = torch.full((fparts, 2, C), float("nan"), device=DEV)
    y = new_act(B, H, W, C, DEV)
    bd = bias.to(DEV)
    call("uh_dwconv3x3_fwd", B, H, W, C, xa.ptr, xa.cstride, xa.coff, wd.data_ptr(), C, bd.data_ptr(), _lib.EPI_RAW_STATS,
         None, None, mask.data_ptr(), y.ptr, y.cstride, y.coff, stat.data_ptr(), S())
    # the one-thread-per-column kernel (UNET_AMD_DWFWD_TILES=1; the only one where C is not a power of two): same bits
    monkeypatch.setenv("UNET_AMD_DWFWD_TILES", "1")
    assert lib.uh_dwconv3x3_fwd_parts(B, H, W, C) == parts
    stat_t = torch.full((parts, 2, C), float("nan"), device=DEV)
    y_t = new_act(B, H, W, C, DEV)
    call("uh_dwconv3x3_fwd", B, H, W, C, xa.ptr, xa.cstride, xa.coff, wd.data_ptr(), C, bd.data_ptr(), _lib.EPI_RAW_STATS,
         None, None, mask.data_ptr(), y_t.ptr, y_t.cstride, y_t.coff, stat_t.data_ptr(), S())
    monkeypatch.delenv("UNET_AMD_DWFWD_TILES")
    torch.cuda.synchronize()
    assert torch.equal(y.buf, y_t.buf), "strip walk and tile kernel disagree on y"
    torch.testing.assert_close(stat.sum(0), stat_t.sum(0), rtol=2e-5, atol=1e-3)
    want = F.conv2d(x, w, bias, padding=1, groups=C)
    close(y.to_nchw(half=True), want, 1e-3, "depthwise output (fp16 store)")
    pads_are_zero(y)
    s = stat.sum(0).cpu()
    close(s[0], want.sum((0, 2, 3)), 1e-3, "sum v")
    close(s[1], (want * want).sum((0, 2, 3)), 1e-3, "sum v^2")
    # eval epilogue: relu(v*scale + shift + x)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    a = new_act(B, H, W, C, DEV)
    scd, shd = sc.to(DEV), sh.to(DEV)       # named: a temporary's memory could be handed out again before the launch
    call("uh_dwconv3x3_fwd", B, H, W, C, xa.ptr, xa.cstride, xa.coff, wd.data_ptr(), C, bd.data_ptr(), _lib.EPI_AFFINE_RELU,
         scd.data_ptr(), shd.data_ptr(), mask.data_ptr(), a.ptr, a.cstride, a.coff, None, S())
    torch.cuda.synchronize()
    close(a.to_nchw(), torch.relu(want * sc.view(1, C, 1, 1) + sh.view(1, C, 1, 1) + x), 1e-2, "eval block output (bf16 store)")
    pads_are_zero(a)
    # input gradient with the residual branch added, and weight gradient
    dy = bf(torch.randn(B, C, H, W, generator=g))
    dr = bf(torch.randn(B, C, H, W, generator=g))
    dya, dra = act_from(dy), act_from(dr)
    dx = new_act(B, H, W, C, DEV)
    call("uh_dwconv3x3_dgrad", B, H, W, C, dya.ptr, dya.cstride, dya.coff, wd.data_ptr(), C, dra.ptr, dra.cstride, dra.coff,
         mask.data_ptr(), dx.ptr, dx.cstride, dx.coff, S())
    wp = torch.full((parts, 9, C), float("nan"), device=DEV)
    call("uh_dwconv3x3_wgrad", B, H, W, C, dya.ptr, dya.cstride, dya.coff, xa.ptr, xa.cstride, xa.coff, wp.data_ptr(), S())
    torch.cuda.synchronize()
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, padding=1, groups=C).backward(dy)
    close(dx.to_nchw(), xr.grad + dr, 1e-2, "depthwise input gradient + residual (bf16 store)")
    pads_are_zero(dx)
    close(wp.sum(0).t().reshape(C, 1, 3, 3), wr.grad, 1e-3, "depthwise weight gradient")
    # without a residual
    call("uh_dwconv3x3_dgrad", B, H, W, C, dya.ptr, dya.cstride, dya.coff, wd.data_ptr(), C, None, 0, 0,
         mask.data_ptr(), dx.ptr, dx.cstride, dx.coff, S())
    torch.cuda.synchronize()
    close(dx.to_nchw(), xr.grad, 1e-2, "depthwise input gradient")


def test_depthwise_conv_padded_channels():
    """The 3-channel level: activations carry 8 channels (5 of them zero), weight and bias have 3."""
    B, H, W, C, Cp = 2, 20, 28, 3, 8
    g = torch.Generator().manual_seed(5)
    x = bf(torch.randn(B, C, H, W, generator=g))
    w = torch.randn(C, 1, 3, 3, generator=g) / 3
    bias = torch.randn(C, generator=g)
    xp = torch.zeros(B, Cp, H, W)
    xp[:, :C] = x
    xa = act_from(xp)
    wd, bd = w.to(DEV), bias.to(DEV)
    mask = rowmask(B, H, W)
    parts = lib.uh_dw_parts(B, H, W, Cp)
    stat = torch.zeros(lib.uh_dwconv3x3_fwd_parts(B, H, W, Cp), 2, Cp, device=DEV)
    y = new_act(B, H, W, Cp, DEV)
    call("uh_dwconv3x3_fwd", B, H, W, Cp, xa.ptr, xa.cstride, xa.coff, wd.data_ptr(), C, bd.data_ptr(), _lib.EPI_RAW_STATS,
         None, None, mask.data_ptr(), y.ptr, y.cstride, y.coff, stat.data_ptr(), S())
    dy = bf(torch.randn(B, Cp, H, W, generator=g))
    dya = act_from(dy)
    dx = new_act(B, H, W, Cp, DEV)
    call("uh_dwconv3x3_dgrad", B, H, W, Cp, dya.ptr, dya.cstride, dya.coff, wd.data_ptr(), C, None, 0, 0,
         mask.data_ptr(), dx.ptr, dx.cstride, dx.coff, S())
    torch.cuda.synchronize()
    want = F.conv2d(x, w, bias, padding=1, groups=C)
    got = y.to_nchw(half=True)
    close(got[:, :C], want, 1e-3, "real channels")
    assert got[:, C:].abs().max().item() == 0, "padding channels stay zero"
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, padding=1, groups=C).backward(dy[:, :C])
    gx = dx.to_nchw()
    close(gx[:, :C], xr.grad, 1e-2, "input gradient, real channels")
    assert gx[:, C:].abs().max().item() == 0
    with pytest.raises(_lib.HipKernelError):
        call("uh_dwconv3x3_fwd", B, H, W, Cp, xa.ptr, xa.cstride, xa.coff, wd.data_ptr(), Cp + 1, bd.data_ptr(),
             _lib.EPI_RAW_STATS, None, None, mask.data_ptr(), y.ptr, y.cstride, y.coff, stat.data_ptr(), S())


@pytest.mark.parametrize("shape", [(32, 1, 256, 256), (3, 1, 17, 33), (1, 1, 1, 1)])
def test_bce_with_logits_value_and_gradient(shape):
    """uh_bce_logits against torch's BCEWithLogitsLoss (the reference's criterion, training_utils.py:105) and
    its autograd gradient, fp32: saturated logits included."""
    from unet_amd.dp import bce_with_logits
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g) * 4
    x.view(-1)[:1] = 90.0
    if x.numel() > 4:
        x.view(-1)[1:4] = torch.tensor([-90.0, 0.0, -0.0])
    t = (torch.rand(shape, generator=g) > 0.5).float()
    xr = x.clone().double().requires_grad_(True)
    ref = torch.nn.BCEWithLogitsLoss()(xr, t.double())
    ref.backward()
    loss, dx = bce_with_logits(x.to(DEV), t.to(DEV), 0.5)
    assert loss.dim() == 0
    assert abs(float(loss) - float(ref.detach())) <= 2e-6 * abs(float(ref.detach())) + 1e-30    # denormal losses may flush
    # sigmoid(x) - t cancels near saturation: fp32 rounding of the sigmoid (6e-8) is the absolute floor
    torch.testing.assert_close(dx.cpu(), 0.5 * xr.grad.float(), rtol=1e-5, atol=2e-7 * 0.5 / x.numel())


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 16, 64, 64), (1, 8, 8, 128, 64), (2, 4, 8, 64, 128),
                                            (3, 20, 20, 128, 128), (2, 40, 24, 64, 192)])
def test_conv_dgrad_with_batchnorm_backward_sums(B, H, W, Cin, Cout):
    """uh_conv_gemm_bnbwd: the input gradient equals uh_conv_gemm(UH_EPI_MASKED) bit for bit, and the per-tile sums
    it leaves add up to what uh_bn_relu_bwd_reduce gets from that gradient in a pass of its own (which reads the
    bf16-rounded gradient; the fused sums come from the fp32 accumulators: agreement to bf16 rounding) and to a
    double-precision evaluation of the same sums."""
    g = torch.Generator().manual_seed(23)
    dy = bf(torch.randn(B, Cout, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / 24)
    y = (torch.randn(B, Cin, H, W, generator=g) * 2 + 0.3).half().float()     # conv output of the preceding layer (fp16)
    mean, invstd = torch.randn(Cin, generator=g) * 0.2, torch.rand(Cin, generator=g) + 0.5
    gamma, beta = torch.randn(Cin, generator=g), torch.randn(Cin, generator=g) * 0.3
    scale = gamma * invstd
    shift = beta - mean * scale
    pk = torch.empty(Cin * 9 * Cout, dtype=torch.bfloat16, device=DEV)
    call("uh_pack_weights", w.to(DEV).data_ptr(), pk.data_ptr(), Cin, 9, Cout, 9, 1, Cin * 9, 9 * Cout, Cout, 1, S())
    a = act_from(dy)
    ya = act_from(y, cstride=Cin + 64, coff=64, half=True)                    # y inside a wider buffer
    mask = rowmask(B, H, W)
    st = [t.to(DEV) for t in (mean, invstd, scale, shift)]
    plain, fused = new_act(B, H, W, Cin, DEV), new_act(B, H, W, Cin, DEV)
    call("uh_conv_gemm", B, H, W, a.ptr, a.cstride, a.coff, Cout, pk.data_ptr(), Cin, 9, _lib.EPI_MASKED,
         plain.ptr, plain.cstride, plain.coff, None, None, None, None, mask.data_ptr(), S())
    parts = lib.uh_conv_gemm_mtiles(B, H, W, Cin)
    fp = torch.full((parts, 2, Cin), float("nan"), device=DEV)
    call("uh_conv_gemm_bnbwd", B, H, W, a.ptr, a.cstride, a.coff, Cout, pk.data_ptr(), Cin, fused.ptr, fused.cstride,
         fused.coff, mask.data_ptr(), ya.ptr, ya.cstride, ya.coff, st[0].data_ptr(), st[1].data_ptr(),
         st[2].data_ptr(), st[3].data_ptr(), fp.data_ptr(), S())
    rparts = lib.uh_bn_bwd_parts(plain.Mp, Cin)
    rp = torch.empty(rparts, 2, Cin, device=DEV)
    call("uh_bn_relu_bwd_reduce", plain.Mp, Cin, plain.ptr, plain.cstride, plain.coff, ya.ptr, ya.cstride, ya.coff,
         st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), rp.data_ptr(), S())
    torch.cuda.synchronize()
    assert torch.equal(plain.buf, fused.buf)
    pads_are_zero(fused)
    got, sep = fp.double().sum(0).cpu(), rp.double().sum(0).cpu()
    assert torch.isfinite(got).all()
    # double-precision sums from the exact gradient
    da = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    c = lambda v: v.double().view(1, -1, 1, 1)                                   # noqa: E731
    d = da * ((y.double() * c(scale) + c(shift)) > 0)
    want = torch.stack([d.sum((0, 2, 3)), (d * (y.double() - c(mean)) * c(invstd)).sum((0, 2, 3))])
    xh = (y.double() - c(mean)) * c(invstd)
    ref = torch.stack([d.abs().sum((0, 2, 3)), (d * xh).abs().sum((0, 2, 3))]).max(dim=1, keepdim=True).values
    # an element whose pre-activation y*scale + shift is within fp32 rounding of zero may fall on either side of the
    # ReLU on the device: its contribution is allowed to be missing or extra
    z = y.double() * c(scale) + c(shift)
    edge = (z.abs() < 1e-5 * (1 + (y.double() * c(scale)).abs())).double()
    slack = torch.stack([(da.abs() * edge).sum((0, 2, 3)), ((da * xh).abs() * edge).sum((0, 2, 3))])
    assert ((got - want).abs() <= 2e-4 * ref + slack).all(), "fused sums against double precision"
    assert ((got - sep).abs() <= 4e-3 * ref).all(), "fused sums against the separate pass"


@pytest.mark.parametrize("B,H,W,K,N", [(2, 16, 16, 256, 128), (1, 8, 8, 128, 64), (3, 20, 20, 512, 256), (2, 5, 7, 64, 192)])
def test_t1_gemm_with_batchnorm_backward_sums(B, H, W, K, N):
    """uh_conv_gemm1_bnbwd: the output equals uh_conv_gemm(T = 1, UH_EPI_MASKED) bit for bit and the per-tile sums add up
    to what uh_bn_relu_bwd_reduce gets from that output (fp32 accumulators against the bf16-rounded tensor)."""
    g = torch.Generator().manual_seed(29)
    a = act_from(bf(torch.randn(B, K, H, W, generator=g)))
    wk = (torch.randn(N, K, generator=g) / (K ** 0.5)).to(torch.bfloat16).to(DEV)
    y = (torch.randn(B, N, H, W, generator=g) * 2 + 0.3).half().float()
    mean, invstd = torch.randn(N, generator=g) * 0.2, torch.rand(N, generator=g) + 0.5
    gamma, beta = torch.randn(N, generator=g), torch.randn(N, generator=g) * 0.3
    scale = gamma * invstd
    st = [t.to(DEV) for t in (mean, invstd, scale, beta - mean * scale)]
    ya = act_from(y, cstride=N + 64, coff=64, half=True)
    mask = rowmask(B, H, W)
    plain, fused = new_act(B, H, W, N, DEV), new_act(B, H, W, N, DEV)
    call("uh_conv_gemm", B, H, W, a.ptr, a.cstride, a.coff, K, wk.data_ptr(), N, 1, _lib.EPI_MASKED, plain.ptr,
         plain.cstride, plain.coff, None, None, None, None, mask.data_ptr(), S())
    parts = lib.uh_conv_gemm_mtiles(B, H, W, N)
    fp = torch.full((parts, 2, N), float("nan"), device=DEV)
    call("uh_conv_gemm1_bnbwd", B, H, W, a.ptr, a.cstride, a.coff, K, wk.data_ptr(), N, fused.ptr, fused.cstride,
         fused.coff, mask.data_ptr(), ya.ptr, ya.cstride, ya.coff, st[0].data_ptr(), st[1].data_ptr(),
         st[2].data_ptr(), st[3].data_ptr(), fp.data_ptr(), S())
    torch.cuda.synchronize()
    assert torch.equal(plain.buf, fused.buf)
    pads_are_zero(fused)
    want, mag = _separate_sums(plain, ya, st)
    got = fp.double().sum(0).cpu()
    assert torch.isfinite(got).all()
    xh_max = ((y - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)).abs().max().item()
    assert ((got - want).abs() <= 6e-3 * mag.max() * max(1.0, xh_max)).all()


def _bn_bwd_setup(g, B, C, H, W):
    y = (torch.randn(B, C, H, W, generator=g) * 2 + 0.3).half().float()       # the layer's fp16 conv output
    mean, invstd = torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.3
    scale = gamma * invstd
    st = [t.to(DEV) for t in (mean, invstd, scale, beta - mean * scale)]
    return act_from(y, cstride=C + 64, coff=64, half=True), st


def _separate_sums(da, ya, st):
    C = da.C
    rparts = lib.uh_bn_bwd_parts(da.Mp, C)
    rp = torch.empty(rparts, 2, C, device=DEV)
    call("uh_bn_relu_bwd_reduce", da.Mp, C, da.ptr, da.cstride, da.coff, ya.ptr, ya.cstride, ya.coff,
         st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), rp.data_ptr(), S())
    torch.cuda.synchronize()
    mag = da.to_nchw().abs().sum((0, 2, 3)).cpu().double()
    return rp.double().sum(0).cpu(), mag


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 64), (1, 4, 12, 128), (3, 2, 2, 512), (2, 64, 48, 64), (1, 2, 2, 1024)])
def test_maxpool_backward_with_batchnorm_backward_sums(B, H, W, C):
    """uh_maxpool2_bwd_bnstat writes the same gradient as uh_maxpool2_bwd, bit for bit, and its sums are those of
    uh_bn_relu_bwd_reduce over that gradient (same inputs, fp32 additions in another order)."""
    g = torch.Generator().manual_seed(61)
    x = (torch.randint(0, 4, (B, C, H, W), generator=g).float() / 2)
    a = act_from(x, cstride=2 * C, coff=C)
    dpa = act_from(bf(torch.randn(B, C, H // 2, W // 2, generator=g)))
    adda = act_from(bf(torch.randn(B, C, H, W, generator=g)), cstride=2 * C, coff=0)
    ya, st = _bn_bwd_setup(g, B, C, H, W)
    plain, fused = new_act(B, H, W, C, DEV), new_act(B, H, W, C, DEV)
    fused.buf[fused.guard:fused.guard + fused.Mp] = 9.0
    call("uh_maxpool2_bwd", B, H, W, C, a.ptr, a.cstride, a.coff, dpa.ptr, dpa.cstride, dpa.coff, adda.ptr,
         adda.cstride, adda.coff, plain.ptr, plain.cstride, plain.coff, S())
    parts = lib.uh_maxpool2_bwd_bnstat_parts(B, H, W, C)
    assert parts >= 1
    fp = torch.full((parts, 2, C), float("nan"), device=DEV)
    call("uh_maxpool2_bwd_bnstat", B, H, W, C, a.ptr, a.cstride, a.coff, dpa.ptr, dpa.cstride, dpa.coff, adda.ptr,
         adda.cstride, adda.coff, fused.ptr, fused.cstride, fused.coff, ya.ptr, ya.cstride, ya.coff,
         st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), fp.data_ptr(), S())
    torch.cuda.synchronize()
    assert torch.equal(plain.buf, fused.buf)
    pads_are_zero(fused)
    want, mag = _separate_sums(plain, ya, st)
    got = fp.double().sum(0).cpu()
    assert torch.isfinite(got).all()
    assert ((got - want).abs() <= 1e-5 * mag.max() * 8 + 1e-6).all()


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 64), (1, 4, 12, 128), (3, 2, 2, 512), (2, 64, 48, 64), (1, 6, 2, 256)])
def test_batchnorm_relu_apply_with_pooling(B, H, W, C):
    """uh_bn_relu_apply_pool2 writes what uh_bn_relu_apply followed by uh_maxpool2_fwd write, bit for bit, pads included."""
    g = torch.Generator().manual_seed(71)
    ya, st = _bn_bwd_setup(g, B, C, H, W)
    mask = rowmask(B, H, W)
    a0, a1 = new_act(B, H, W, 2 * C, DEV).window(0, C), new_act(B, H, W, 2 * C, DEV).window(0, C)
    p0, p1 = new_act(B, H // 2, W // 2, C, DEV), new_act(B, H // 2, W // 2, C, DEV)
    for t in (a1, p1):                                       # stale contents that have to be overwritten
        t.padded_rows()[:, :] = 7.0
    call("uh_bn_relu_apply", ya.Mp, C, ya.ptr, ya.cstride, ya.coff, st[2].data_ptr(), st[3].data_ptr(), mask.data_ptr(),
         a0.ptr, a0.cstride, a0.coff, S())
    call("uh_maxpool2_fwd", B, H, W, C, a0.ptr, a0.cstride, a0.coff, p0.ptr, p0.cstride, p0.coff, S())
    call("uh_bn_relu_apply_pool2", B, H, W, C, ya.ptr, ya.cstride, ya.coff, st[2].data_ptr(), st[3].data_ptr(),
         a1.ptr, a1.cstride, a1.coff, p1.ptr, p1.cstride, p1.coff, S())
    torch.cuda.synchronize()
    assert torch.equal(a0.padded_rows(), a1.padded_rows())
    assert torch.equal(p0.buf, p1.buf)
    pads_are_zero(p1)


@pytest.mark.parametrize("B,H,W,C,c_real", [(2, 16, 16, 64, 64), (1, 20, 12, 8, 3), (2, 5, 7, 128, 128), (1, 33, 4, 256, 256),
                                            (1, 3, 3, 512, 512), (2, 40, 30, 64, 64), (3, 37, 45, 16, 16)])
def test_depthwise_block_backward_in_one_pass(B, H, W, C, c_real, monkeypatch):
    """uh_dwblock_bwd against the three passes it replaces (uh_bn_res_relu_bwd_apply, uh_dwconv3x3_dgrad,
    uh_dwconv3x3_wgrad): the same gradients up to the bf16 rounding of dy that the fused pass does not make."""
    g = torch.Generator().manual_seed(73)
    live = (torch.arange(C) < c_real).float().view(1, C, 1, 1)
    x = bf(torch.randn(B, C, H, W, generator=g)) * live
    da = bf(torch.randn(B, C, H, W, generator=g)) * live
    y = (torch.randn(B, C, H, W, generator=g) * 1.5 + 0.2).half().float() * live
    wt = (torch.randn(c_real, 1, 3, 3, generator=g) / 3).to(DEV)
    mean, invstd = torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.3
    scale = gamma * invstd
    st = [(t * live.view(-1) if i >= 2 else t).to(DEV) for i, t in enumerate((mean, invstd, scale, beta - mean * scale))]
    coef = (torch.randn(2, C, generator=g) * 0.05).to(DEV)
    xa, daa, ya = act_from(x, cstride=2 * C, coff=C), act_from(da), act_from(y, half=True)
    mask = rowmask(B, H, W)
    stp = [t.data_ptr() for t in st]
    dy, dr, dx0, dx1 = (new_act(B, H, W, C, DEV) for _ in range(4))
    call("uh_bn_res_relu_bwd_apply", xa.Mp, C, daa.ptr, daa.cstride, daa.coff, ya.ptr, ya.cstride, ya.coff, xa.ptr,
         xa.cstride, xa.coff, *stp, coef.data_ptr(), mask.data_ptr(), dy.ptr, dy.cstride, dy.coff, dr.ptr, dr.cstride,
         dr.coff, S())
    call("uh_dwconv3x3_dgrad", B, H, W, C, dy.ptr, dy.cstride, dy.coff, wt.data_ptr(), c_real, dr.ptr, dr.cstride,
         dr.coff, mask.data_ptr(), dx0.ptr, dx0.cstride, dx0.coff, S())
    p0 = lib.uh_dw_parts(B, H, W, C)
    wp0 = torch.empty(p0, 9, C, device=DEV)
    call("uh_dwconv3x3_wgrad", B, H, W, C, dy.ptr, dy.cstride, dy.coff, xa.ptr, xa.cstride, xa.coff, wp0.data_ptr(), S())
    p1 = lib.uh_dwblock_bwd_parts(B, H, W, C)
    assert p1 >= 1
    wp1 = torch.full((p1, 9, C), float("nan"), device=DEV)
    dx1.padded_rows()[:, :] = 5.0
    call("uh_dwblock_bwd", B, H, W, C, daa.ptr, daa.cstride, daa.coff, ya.ptr, ya.cstride, ya.coff, xa.ptr, xa.cstride,
         xa.coff, wt.data_ptr(), c_real, *stp, coef.data_ptr(), dx1.ptr, dx1.cstride, dx1.coff,
         wp1.data_ptr(), S())
    dw0, dw1 = torch.empty(c_real, 1, 3, 3, device=DEV), torch.empty(c_real, 1, 3, 3, device=DEV)
    call("uh_dw_wgrad_finalize", wp0.data_ptr(), p0, C, c_real, dw0.data_ptr(), S())
    call("uh_dw_wgrad_finalize", wp1.data_ptr(), p1, C, c_real, dw1.data_ptr(), S())
    # the workgroup kernel with the LDS exchange (UNET_AMD_DWBLOCK_LDS=1): dx bit for bit, dw up to summation order
    monkeypatch.setenv("UNET_AMD_DWBLOCK_LDS", "1")
    p2 = lib.uh_dwblock_bwd_parts(B, H, W, C)
    wp2 = torch.full((p2, 9, C), float("nan"), device=DEV)
    dx2 = new_act(B, H, W, C, DEV)
    call("uh_dwblock_bwd", B, H, W, C, daa.ptr, daa.cstride, daa.coff, ya.ptr, ya.cstride, ya.coff, xa.ptr, xa.cstride,
         xa.coff, wt.data_ptr(), c_real, *stp, coef.data_ptr(), dx2.ptr, dx2.cstride, dx2.coff,
         wp2.data_ptr(), S())
    monkeypatch.delenv("UNET_AMD_DWBLOCK_LDS")
    dw2 = torch.empty(c_real, 1, 3, 3, device=DEV)
    call("uh_dw_wgrad_finalize", wp2.data_ptr(), p2, C, c_real, dw2.data_ptr(), S())
    torch.cuda.synchronize()
    assert torch.equal(dx1.to_nchw(), dx2.to_nchw()), "strip walk and workgroup kernel disagree on dx"
    close(dw1, dw2, 1e-4, "dw, strip walk against workgroup kernel")
    pads_are_zero(dx1)
    close(dx1.to_nchw(), dx0.to_nchw(), 1.2e-2, "dx")           # nine products of values rounded at 2^-9, then summed
    assert torch.isfinite(dw1).all()
    scale_dw = (dy.to_nchw().abs().flatten(2).sum(2).sum(0)[:c_real] * x.abs().max()).cpu().view(-1, 1, 1, 1)
    assert ((dw1 - dw0).abs().cpu() <= 4e-3 * scale_dw + 1e-5).all(), "dw"
    # and against fp32 autograd of the same block
    xr, wr = x.clone().requires_grad_(True), wt.cpu().clone().requires_grad_(True)
    c = lambda v: v.view(1, -1, 1, 1)                                            # noqa: E731
    d = da * ((y * c(st[2].cpu()) + c(st[3].cpu()) + x) > 0)
    dyr = c(st[2].cpu()) * (d - c(coef[0].cpu()) - (y - c(mean)) * c(invstd) * c(coef[1].cpu())) * live
    out = F.conv2d(xr, torch.cat([wr, torch.zeros(C - c_real, 1, 3, 3)]), padding=1, groups=C)
    out.backward(dyr)
    close(dx1.to_nchw(), xr.grad + d, 1e-2, "dx against autograd")
    close(dw1, wr.grad, 2e-3, "dw against autograd")


@pytest.mark.parametrize("B,H,W,C,c_real", [(2, 16, 16, 64, 64), (1, 20, 12, 8, 3), (2, 40, 30, 64, 64), (3, 37, 45, 16, 16),
                                            (1, 3, 3, 512, 512), (2, 33, 31, 128, 128)])
def test_depthwise_forward_applies_the_previous_block(B, H, W, C, c_real):
    """uh_dwconv3x3_fwd_applied against uh_bn_res_relu_apply followed by uh_dwconv3x3_fwd: the activation it writes,
    the convolution output and the partial sums, bit for bit."""
    assert lib.uh_dwconv3x3_fwd_takes_apply(C) == 1 and lib.uh_dwconv3x3_fwd_takes_apply(24) == 0
    g = torch.Generator().manual_seed(83)
    live = (torch.arange(C) < c_real).float().view(1, C, 1, 1)
    xq = act_from(bf(torch.randn(B, C, H, W, generator=g)) * live, cstride=2 * C, coff=C)
    yq = act_from((torch.randn(B, C, H, W, generator=g) * 1.5).half().float() * live, half=True)
    sc = ((torch.rand(C, generator=g) + 0.5) * live.view(-1)).to(DEV)
    sh = ((torch.randn(C, generator=g) * 0.3) * live.view(-1)).to(DEV)
    wt = (torch.randn(c_real, 1, 3, 3, generator=g) / 3).to(DEV)
    bias = torch.randn(c_real, generator=g).to(DEV)
    mask = rowmask(B, H, W)
    parts = lib.uh_dwconv3x3_fwd_parts(B, H, W, C)
    a0, a1, y0, y1 = (new_act(B, H, W, C, DEV) for _ in range(4))
    st0 = torch.full((parts, 2, C), float("nan"), device=DEV)
    st1 = torch.full((parts, 2, C), float("nan"), device=DEV)
    a1.padded_rows()[:, :] = 7.0                                  # every position must be written, padding included
    y1.padded_rows()[:, :] = 7.0
    call("uh_bn_res_relu_apply", xq.Mp, C, yq.ptr, yq.cstride, yq.coff, xq.ptr, xq.cstride, xq.coff, sc.data_ptr(),
         sh.data_ptr(), mask.data_ptr(), a0.ptr, a0.cstride, a0.coff, S())
    call("uh_dwconv3x3_fwd", B, H, W, C, a0.ptr, a0.cstride, a0.coff, wt.data_ptr(), c_real, bias.data_ptr(),
         _lib.EPI_RAW_STATS, None, None, mask.data_ptr(), y0.ptr, y0.cstride, y0.coff, st0.data_ptr(), S())
    call("uh_dwconv3x3_fwd_applied", B, H, W, C, yq.ptr, yq.cstride, yq.coff, xq.ptr, xq.cstride, xq.coff, sc.data_ptr(),
         sh.data_ptr(), a1.ptr, a1.cstride, a1.coff, wt.data_ptr(), c_real, bias.data_ptr(), y1.ptr, y1.cstride, y1.coff,
         st1.data_ptr(), S())
    torch.cuda.synchronize()
    assert torch.equal(a0.padded_rows(), a1.padded_rows()), "activation"
    assert torch.equal(y0.padded_rows(), y1.padded_rows()), "convolution output"
    assert torch.equal(st0, st1), "partial sums"
    pads_are_zero(a1)
    pads_are_zero(y1)


@pytest.mark.parametrize("B,H,W,C,c_real", [(2, 16, 16, 64, 64), (1, 20, 12, 8, 3), (2, 40, 30, 64, 64), (3, 37, 45, 16, 16),
                                            (1, 3, 3, 512, 512)])
def test_depthwise_block_backward_leaves_the_previous_block_its_sums(B, H, W, C, c_real):
    """uh_dwblock_bwd_prev: same dx and weight-gradient partials as uh_dwblock_bwd, bit for bit, plus the
    BatchNorm-backward sums of the block that produced x -- against uh_bn_res_relu_bwd_reduce on the stored dx."""
    assert lib.uh_dwblock_bwd_takes_prev() == 1
    g = torch.Generator().manual_seed(79)
    live = (torch.arange(C) < c_real).float().view(1, C, 1, 1)
    rnd = lambda: bf(torch.randn(B, C, H, W, generator=g)) * live                # noqa: E731
    stats = lambda: [((t * live.view(-1)) if i >= 2 else t).to(DEV) for i, t in enumerate(        # noqa: E731
        (torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g),
         torch.randn(C, generator=g) * 0.3))]
    # the previous block: x = relu(scale_q * yq + shift_q + xin), made by the forward kernel itself
    xin, yq = act_from(rnd()), act_from((torch.randn(B, C, H, W, generator=g) * 1.5).half().float() * live, half=True)
    stq = stats()
    mask = rowmask(B, H, W)
    xa = new_act(B, H, W, C, DEV)
    call("uh_bn_res_relu_apply", xa.Mp, C, yq.ptr, yq.cstride, yq.coff, xin.ptr, xin.cstride, xin.coff, stq[2].data_ptr(),
         stq[3].data_ptr(), mask.data_ptr(), xa.ptr, xa.cstride, xa.coff, S())
    # this block
    daa = act_from(rnd())
    ya = act_from((torch.randn(B, C, H, W, generator=g) * 1.5 + 0.2).half().float() * live, half=True)
    wt = (torch.randn(c_real, 1, 3, 3, generator=g) / 3).to(DEV)
    st = stats()
    stp = [t.data_ptr() for t in st]
    coef = (torch.randn(2, C, generator=g) * 0.05).to(DEV)
    parts = lib.uh_dwblock_bwd_parts(B, H, W, C)
    dx0, dx1 = new_act(B, H, W, C, DEV), new_act(B, H, W, C, DEV)
    wp0, wp1 = torch.empty(parts, 9, C, device=DEV), torch.empty(parts, 9, C, device=DEV)
    qp = torch.full((parts, 2, C), float("nan"), device=DEV)
    call("uh_dwblock_bwd", B, H, W, C, daa.ptr, daa.cstride, daa.coff, ya.ptr, ya.cstride, ya.coff, xa.ptr, xa.cstride,
         xa.coff, wt.data_ptr(), c_real, *stp, coef.data_ptr(), dx0.ptr, dx0.cstride, dx0.coff, wp0.data_ptr(), S())
    call("uh_dwblock_bwd_prev", B, H, W, C, daa.ptr, daa.cstride, daa.coff, ya.ptr, ya.cstride, ya.coff, xa.ptr,
         xa.cstride, xa.coff, wt.data_ptr(), c_real, *stp, coef.data_ptr(), dx1.ptr, dx1.cstride, dx1.coff,
         wp1.data_ptr(), yq.ptr, yq.cstride, yq.coff, stq[0].data_ptr(), stq[1].data_ptr(), qp.data_ptr(), S())
    # the separate pass over (dx, yq, xin)
    rparts = lib.uh_bn_bwd_parts(xa.Mp, C)
    rp = torch.empty(rparts, 2, C, device=DEV)
    call("uh_bn_res_relu_bwd_reduce", xa.Mp, C, dx0.ptr, dx0.cstride, dx0.coff, yq.ptr, yq.cstride, yq.coff, xin.ptr,
         xin.cstride, xin.coff, *[t.data_ptr() for t in stq], rp.data_ptr(), S())
    outs = []
    for part, n in ((rp, rparts), (qp, parts)):
        dgam, dbet = torch.empty(c_real, device=DEV), torch.empty(c_real, device=DEV)
        cf = torch.empty(2, C, device=DEV)
        call("uh_bn_bwd_finalize", part.data_ptr(), n, C, c_real, B * H * W, dgam.data_ptr(), dbet.data_ptr(),
             cf.data_ptr(), 0, S())
        outs.append((dgam, dbet, cf))
    torch.cuda.synchronize()
    assert torch.equal(dx0.buf, dx1.buf) and torch.equal(wp0, wp1)
    assert torch.isfinite(qp).all()
    # scale of the sums: sum |d| and sum |d * xhat| per channel
    d = dx0.to_nchw().float() * (xa.to_nchw() > 0)
    xhat = (yq.to_nchw(half=True).float() - stq[0].view(1, -1, 1, 1)) * stq[1].view(1, -1, 1, 1)
    mag0 = d.abs().sum((0, 2, 3))[:c_real] + 1e-6
    mag1 = (d * xhat).abs().sum((0, 2, 3))[:c_real] + 1e-6
    (g0, b0, c0), (g1, b1, c1) = outs
    assert ((b1 - b0).abs() <= 2e-5 * mag0).all(), "dbeta"
    assert ((g1 - g0).abs() <= 2e-5 * mag1).all(), "dgamma"
    assert ((c1 - c0)[:, :c_real].abs() <= 2e-5 * torch.stack([mag0, mag1]) / (B * H * W) + 1e-9).all(), "coef"
    # and the variant that leaves the column sums of dx (the bias gradient of a convolution in front of the block)
    dx2 = new_act(B, H, W, C, DEV)
    wp2 = torch.empty(parts, 9, C, device=DEV)
    cpart = torch.full((parts, C), float("nan"), device=DEV)
    call("uh_dwblock_bwd_colsum", B, H, W, C, daa.ptr, daa.cstride, daa.coff, ya.ptr, ya.cstride, ya.coff, xa.ptr,
         xa.cstride, xa.coff, wt.data_ptr(), c_real, *stp, coef.data_ptr(), dx2.ptr, dx2.cstride, dx2.coff,
         wp2.data_ptr(), cpart.data_ptr(), S())
    db = torch.empty(C, device=DEV)
    call("uh_colsum_finalize", cpart.data_ptr(), parts, C, 1, db.data_ptr(), 0, S())
    torch.cuda.synchronize()
    assert torch.equal(dx0.buf, dx2.buf) and torch.equal(wp0, wp2)
    stored = dx0.to_nchw().double()
    torch.testing.assert_close(db.double().cpu(), stored.sum((0, 2, 3)).cpu(), rtol=1e-5,
                               atol=2e-6 * stored.abs().sum((0, 2, 3)).max().item() + 1e-6)


@pytest.mark.parametrize("B,H,W,C,OC", [(2, 8, 12, 64, 1), (1, 5, 7, 64, 2), (2, 16, 16, 8, 1)])
def test_head_backward_with_batchnorm_backward_sums(B, H, W, C, OC):
    """uh_head_bwd_bnstat: gradient and weight partials as uh_head_bwd, bit for bit; sums as the separate pass."""
    g = torch.Generator().manual_seed(67)
    a = act_from(bf(torch.randn(B, C, H, W, generator=g)), cstride=2 * C, coff=C)
    w = (torch.randn(OC, C, generator=g) / 8).to(DEV)
    dl = torch.randn(B, OC, H, W, generator=g).to(DEV)
    ya, st = _bn_bwd_setup(g, B, C, H, W)
    plain, fused = new_act(B, H, W, C, DEV), new_act(B, H, W, C, DEV)
    parts = lib.uh_head_bwd_parts(B, H, W, C)
    hp0, hp1 = torch.empty(parts, OC, C + 1, device=DEV), torch.empty(parts, OC, C + 1, device=DEV)
    fp = torch.full((parts, 2, C), float("nan"), device=DEV)
    call("uh_head_bwd", B, H, W, C, OC, a.ptr, a.cstride, a.coff, w.data_ptr(), dl.data_ptr(), plain.ptr,
         plain.cstride, plain.coff, hp0.data_ptr(), S())
    call("uh_head_bwd_bnstat", B, H, W, C, OC, a.ptr, a.cstride, a.coff, w.data_ptr(), dl.data_ptr(), fused.ptr,
         fused.cstride, fused.coff, hp1.data_ptr(), ya.ptr, ya.cstride, ya.coff, st[0].data_ptr(), st[1].data_ptr(),
         st[2].data_ptr(), st[3].data_ptr(), fp.data_ptr(), S())
    torch.cuda.synchronize()
    assert torch.equal(plain.buf, fused.buf) and torch.equal(hp0, hp1)
    pads_are_zero(fused)
    want, mag = _separate_sums(plain, ya, st)
    got = fp.double().sum(0).cpu()
    assert torch.isfinite(got).all()
    assert ((got - want).abs() <= 1e-5 * mag.max() * 8 + 1e-6).all()


@pytest.mark.parametrize("B,Hi,Wi,Ho,Wo,C", [(2, 4, 6, 5, 6, 64), (1, 10, 12, 11, 13, 8), (2, 8, 8, 8, 9, 128),
                                             (1, 6, 6, 12, 12, 64), (1, 9, 7, 4, 5, 8)])
def test_bilinear_resize_forward_backward(B, Hi, Wi, Ho, Wo, C):
    """uh_resize_bilinear_fwd/bwd against F.interpolate(mode='bilinear', align_corners=False) and its autograd"""
    g = torch.Generator().manual_seed(79)
    x = bf(torch.randn(B, C, Hi, Wi, generator=g))
    dout = bf(torch.randn(B, C, Ho, Wo, generator=g))
    xa, da = act_from(x, cstride=2 * C, coff=C), act_from(dout)
    out, dx = new_act(B, Ho, Wo, C, DEV), new_act(B, Hi, Wi, C, DEV)
    out.padded_rows()[:, :] = 3.0
    dx.padded_rows()[:, :] = 3.0
    call("uh_resize_bilinear_fwd", B, Hi, Wi, Ho, Wo, C, xa.ptr, xa.cstride, xa.coff, out.ptr, out.cstride, out.coff, S())
    call("uh_resize_bilinear_bwd", B, Hi, Wi, Ho, Wo, C, da.ptr, da.cstride, da.coff, dx.ptr, dx.cstride, dx.coff, S())
    torch.cuda.synchronize()
    xr = x.clone().requires_grad_(True)
    want = F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=False)
    want.backward(dout)
    close(out.to_nchw(), want, 5e-3, "resize fwd")
    close(dx.to_nchw(), xr.grad, 5e-3, "resize bwd")
    pads_are_zero(out)
    pads_are_zero(dx)


@pytest.mark.parametrize("B,h,w,bigH,bigW,C", [(2, 3, 4, 7, 8, 64), (1, 5, 6, 11, 13, 8), (2, 2, 2, 4, 5, 16)])
def test_space_to_depth_of_odd_sizes(B, h, w, bigH, bigW, C):
    g = torch.Generator().manual_seed(83)
    x = bf(torch.randn(B, C, bigH, bigW, generator=g))
    xa = act_from(x)
    x4 = new_act(B, h, w, 4 * C, DEV)
    call("uh_space_to_depth2_of", B, h, w, C, bigH, bigW, xa.ptr, xa.cstride, xa.coff, x4.ptr, x4.cstride, 0, S())
    back = new_act(B, bigH, bigW, C, DEV)
    back.padded_rows()[:, :] = 3.0
    call("uh_depth_to_space2_to", B, h, w, C, bigH, bigW, x4.ptr, x4.cstride, 0, back.ptr, back.cstride, back.coff, S())
    torch.cuda.synchronize()
    got = x4.to_nchw().view(B, 4, C, h, w)                    # channel index t*C + c, t = 2*dy + dx
    for t in range(4):
        assert torch.equal(got[:, t].cpu(), x[:, :, t // 2:2 * h:2, t % 2:2 * w:2])
    pads_are_zero(x4)
    want = torch.zeros_like(x)
    want[:, :, :2 * h, :2 * w] = x[:, :, :2 * h, :2 * w]      # the dropped row / column gets zeros
    assert torch.equal(back.to_nchw().cpu(), want)
    pads_are_zero(back)
    extra = bf(torch.randn(B, C, bigH, bigW, generator=g))
    ea = act_from(extra, cstride=2 * C, coff=C)
    back.padded_rows()[:, :] = 3.0
    call("uh_depth_to_space2_add", B, h, w, C, bigH, bigW, x4.ptr, x4.cstride, 0, ea.ptr, ea.cstride, ea.coff, back.ptr,
         back.cstride, back.coff, S())
    torch.cuda.synchronize()
    assert torch.equal(back.to_nchw().cpu(), bf(want + extra))
    pads_are_zero(back)


@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 64), (1, 9, 13, 8), (3, 6, 20, 128), (1, 2, 2, 512)])
def test_block_output_stored_in_space_to_depth_layout_too(B, H, W, C):
    """uh_bn_res_relu_apply_s2d against uh_bn_res_relu_apply followed by uh_space_to_depth2_of, odd sizes included."""
    g = torch.Generator().manual_seed(89)
    xq = act_from(bf(torch.randn(B, C, H, W, generator=g)))
    yq = act_from((torch.randn(B, C, H, W, generator=g) * 1.5).half().float(), half=True)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.3).to(DEV)
    Hs, Ws, K = H // 2, W // 2, max(64, 4 * C)
    a0, a1 = new_act(B, H, W, C, DEV), new_act(B, H, W, C, DEV)
    d0, d1 = new_act(B, Hs, Ws, K, DEV), new_act(B, Hs, Ws, K, DEV)             # zero-initialised
    a1.padded_rows()[:, :] = 3.0
    call("uh_bn_res_relu_apply", xq.Mp, C, yq.ptr, yq.cstride, yq.coff, xq.ptr, xq.cstride, xq.coff, sc.data_ptr(),
         sh.data_ptr(), rowmask(B, H, W).data_ptr(), a0.ptr, a0.cstride, a0.coff, S())
    call("uh_space_to_depth2_of", B, Hs, Ws, C, H, W, a0.ptr, a0.cstride, a0.coff, d0.ptr, d0.cstride, 0, S())
    call("uh_bn_res_relu_apply_s2d", B, H, W, C, yq.ptr, yq.cstride, yq.coff, xq.ptr, xq.cstride, xq.coff, sc.data_ptr(),
         sh.data_ptr(), a1.ptr, a1.cstride, a1.coff, d1.ptr, d1.cstride, 0, S())
    torch.cuda.synchronize()
    assert torch.equal(a0.padded_rows(), a1.padded_rows())
    assert torch.equal(d0.buf, d1.buf)
    assert d1.to_nchw().abs().sum().item() > 0
    pads_are_zero(d1)


def test_depthwise_weight_gradient_tables_added_up_in_one_launch():
    """uh_dw_wgrad_finalize_multi against one uh_dw_wgrad_finalize per table: the same bits."""
    g = torch.Generator().manual_seed(97)
    shapes = [(37, 8, 3), (2300, 64, 64), (1, 16, 16), (513, 128, 128), (64, 512, 512)]
    tabs = [torch.randn(p, 9, c, generator=g).to(DEV) for p, c, _ in shapes]
    one = [torch.empty(cr, 1, 3, 3, device=DEV) for _, _, cr in shapes]
    many = [torch.full((cr, 1, 3, 3), float("nan"), device=DEV) for _, _, cr in shapes]
    for t, o, (p, c, cr) in zip(tabs, one, shapes):
        call("uh_dw_wgrad_finalize", t.data_ptr(), p, c, cr, o.data_ptr(), S())
    descs = (_lib.DwFinDesc * len(shapes))(*[_lib.DwFinDesc(t.data_ptr(), o.data_ptr(), p, c, cr)
                                             for t, o, (p, c, cr) in zip(tabs, many, shapes)])
    call("uh_dw_wgrad_finalize_multi", descs, len(shapes), S())
    torch.cuda.synchronize()
    for a, b in zip(one, many):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        call("uh_dw_wgrad_finalize_multi", descs, _lib.DWFIN_MAX + 1, S())


def test_small_reductions_in_one_launch():
    """uh_reduce_multi against the single launches it stands for: weight-gradient reductions of every route (transposed
    convolution, few and many splits) bit for bit, column sums (plain, gathered, accumulating) to double rounding."""
    g = torch.Generator().manual_seed(101)
    R = _lib.ReduceDesc
    descs, checks, keep = [], [], []

    def wgrad(ns, N, T, K, on, ot, sn, st, sk, total_out, split_stride=None, accumulate=0):
        split_stride = split_stride or N * T * K
        part = torch.randn(ns, split_stride, generator=g).to(DEV)
        base = torch.randn(total_out, generator=g).to(DEV)
        one, many = base.clone(), base.clone()
        call("uh_wgrad_reduce_strided", part.data_ptr(), ns, split_stride, N, T, K, on, ot, sn, st, sk, one.data_ptr(),
             accumulate, S())
        descs.append(R.wgrad(part.data_ptr(), ns, split_stride, N, T, K, on, ot, sn, st, sk, many.data_ptr(), accumulate))
        checks.append((one, many, True))
        keep.append(part)

    cin, cout = 64, 128
    wgrad(7, cout, 4, cin, cin, cout * cin, 4, 1, 4 * cout, cin * cout * 4)                    # transposed convolution
    wgrad(96, 64, 1, 64, 64, 0, 64, 0, 1, 64 * 64)                                           # many splits, small tensor
    wgrad(5, 128, 4, 24, 32, 1, 4, 1, 128 * 4 + 7, 128 * 24 * 4 + 4 * 128 * 8, 128 * 4 * 32)  # few splits, odd strides
    wgrad(3, 64, 1, 64, 64, 0, 64, 0, 1, 64 * 64, accumulate=1)

    def colsum(P, C, n_out=None, cols=None, accumulate=0):
        part = torch.randn(P, C, generator=g).to(DEV)
        n = n_out or C
        base = torch.randn(n, generator=g).to(DEV)
        one, many = base.clone(), base.clone()
        if cols is None:
            call("uh_colsum_finalize", part.data_ptr(), P, C, 1, one.data_ptr(), accumulate, S())
            descs.append(R.colsum(part.data_ptr(), P, C, many.data_ptr(), accumulate=accumulate))
        else:
            ct = torch.tensor(cols, dtype=torch.int32, device=DEV)
            call("uh_colsum_finalize_gather", part.data_ptr(), P, C, n, ct.data_ptr(), one.data_ptr(), S())
            descs.append(R.colsum(part.data_ptr(), P, C, many.data_ptr(), n_out=n, cols=ct.data_ptr()))
            keep.append(ct)
        checks.append((one, many, False))
        keep.append(part)

    colsum(300, 192)
    colsum(2300, 64, accumulate=1)
    colsum(17, 216, n_out=5, cols=[3, 215, 0, 64, 64])
    colsum(1, 8)
    table = (R * len(descs))(*descs)
    call("uh_reduce_multi", table, len(descs), S())
    torch.cuda.synchronize()
    for one, many, exact in checks:
        if exact:
            assert torch.equal(one, many)
        else:
            torch.testing.assert_close(many, one, rtol=1e-6, atol=1e-6)
    with pytest.raises(RuntimeError):
        call("uh_reduce_multi", table, _lib.REDUCE_MAX + 1, S())
    # a 3x3 weight gradient is not taken
    with pytest.raises(RuntimeError):
        bad = (R * 1)(R.wgrad(keep[0].data_ptr(), 2, 64 * 9 * 64, 64, 9, 64, 9 * 64, 64, 9 * 64, 1, 9, checks[0][1].data_ptr()))
        call("uh_reduce_multi", bad, 1, S())


# ---- AttentionGate kernels (lunext.py:76-122) ---------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,C", [(2, 8, 8, 64), (1, 6, 10, 128), (2, 4, 4, 256)])
def test_gate_pointwise_forward_backward(B, H, W, C):
    """uh_gate_fwd / uh_gate_bwd: out = sigmoid(z1) * tanh(z2) + sigmoid(z3) * x on z = [z1 | z2 | z3]"""
    g = torch.Generator().manual_seed(31)
    z = bf(torch.randn(B, 3 * C, H, W, generator=g))
    x = bf(torch.randn(B, C, H, W, generator=g))
    dout = bf(torch.randn(B, C, H, W, generator=g))
    za, xa, da = act_from(z), act_from(x), act_from(dout)
    out = new_act(B, H, W, C, DEV)
    call("uh_gate_fwd", za.Mp, C, za.ptr, za.cstride, 0, xa.ptr, xa.cstride, xa.coff, out.ptr, out.cstride, out.coff, S())
    zt, xt = z.clone().requires_grad_(True), x.clone().requires_grad_(True)
    z1, z2, z3 = zt[:, :C], zt[:, C:2 * C], zt[:, 2 * C:]
    want = torch.sigmoid(z1) * torch.tanh(z2) + torch.sigmoid(z3) * xt
    want.backward(dout)
    torch.cuda.synchronize()
    close(out.to_nchw(), want, 1e-2, "gate forward")
    pads_are_zero(out)
    dz, dxd = new_act(B, H, W, 3 * C, DEV), new_act(B, H, W, C, DEV)
    call("uh_gate_bwd", za.Mp, C, da.ptr, da.cstride, da.coff, za.ptr, za.cstride, 0, xa.ptr, xa.cstride, xa.coff,
         rowmask(B, H, W).data_ptr(), dz.ptr, dz.cstride, 0, dxd.ptr, dxd.cstride, dxd.coff, S())
    torch.cuda.synchronize()
    close(dz.to_nchw(), zt.grad, 1e-2, "gate backward: dz")
    close(dxd.to_nchw(), xt.grad, 1e-2, "gate backward: the direct path to x")
    pads_are_zero(dz)
    pads_are_zero(dxd)
    # the same launch with the column sums of dz (the three bias gradients): same tensors, sums of the stored values
    dz2, dxd2 = new_act(B, H, W, 3 * C, DEV), new_act(B, H, W, C, DEV)
    parts = lib.uh_gate_bwd_parts(za.Mp, C)
    assert parts >= 1
    part = torch.full((parts, 3 * C), float("nan"), device=DEV)
    call("uh_gate_bwd_colsum", za.Mp, C, da.ptr, da.cstride, da.coff, za.ptr, za.cstride, 0, xa.ptr, xa.cstride, xa.coff,
         rowmask(B, H, W).data_ptr(), dz2.ptr, dz2.cstride, 0, dxd2.ptr, dxd2.cstride, dxd2.coff, part.data_ptr(), S())
    db = torch.empty(3 * C, device=DEV)
    call("uh_colsum_finalize", part.data_ptr(), parts, 3 * C, 1, db.data_ptr(), 0, S())
    torch.cuda.synchronize()
    assert torch.equal(dz.buf, dz2.buf) and torch.equal(dxd.buf, dxd2.buf)
    stored = dz.to_nchw().double()
    torch.testing.assert_close(db.double().cpu(), stored.sum((0, 2, 3)).cpu(), rtol=1e-5,
                               atol=1e-5 * stored.abs().sum((0, 2, 3)).max().item())


@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 3), (1, 8, 12, 2), (3, 4, 4, 8)])
def test_gate8_forward_backward(B, H, W, C):
    """uh_gate8_fwd / uh_gate8_bwd: the whole gate of the 8-channel level in one kernel (lunext.py:100-117):
    out = sigmoid(W1 g + b1) * tanh(W2 g + b2) + sigmoid(W3 (g + x) + b3) * x, and its gradients"""
    gen = torch.Generator().manual_seed(37)
    gt = bf(torch.randn(B, C, H, W, generator=gen)).requires_grad_(True)
    xt = bf(torch.randn(B, C, H, W, generator=gen)).requires_grad_(True)
    dout = bf(torch.randn(B, C, H, W, generator=gen))
    Ws = [torch.randn(C, C, generator=gen).requires_grad_(True) for _ in range(3)]
    bs = [torch.randn(C, generator=gen).requires_grad_(True) for _ in range(3)]

    def lin(Wm, b, t):
        return torch.einsum("oc,bchw->bohw", Wm, t) + b.view(1, -1, 1, 1)
    want = (torch.sigmoid(lin(Ws[0], bs[0], gt)) * torch.tanh(lin(Ws[1], bs[1], gt))
            + torch.sigmoid(lin(Ws[2], bs[2], gt + xt)) * xt)
    want.backward(dout)
    wts = torch.zeros(216)
    for j in range(3):
        wts[64 * j:64 * j + 64].view(8, 8)[:C, :C] = Ws[j].detach()
        wts[192 + 8 * j:192 + 8 * j + C] = bs[j].detach()
    wts = wts.to(DEV)

    def act8(t):
        a = new_act(B, H, W, 8, DEV)
        full = torch.zeros(B, 8, H, W)
        full[:, :C] = t.detach()
        a.fill_from_nchw(full.to(DEV))
        return a
    ga, xa, da = act8(gt), act8(xt), act8(dout)
    out = new_act(B, H, W, 8, DEV)
    mask = rowmask(B, H, W).data_ptr()
    call("uh_gate8_fwd", ga.Mp, C, ga.ptr, ga.cstride, ga.coff, xa.ptr, xa.cstride, xa.coff, wts.data_ptr(), mask,
         out.ptr, out.cstride, out.coff, S())
    torch.cuda.synchronize()
    close(out.to_nchw()[:, :C], want, 1e-2, "gate8 forward")
    pads_are_zero(out)
    dg, dx = new_act(B, H, W, 8, DEV), new_act(B, H, W, 8, DEV)
    parts = lib.uh_gate8_parts(ga.Mp)
    part = torch.zeros(parts * 216, device=DEV)
    call("uh_gate8_bwd", ga.Mp, C, da.ptr, da.cstride, da.coff, ga.ptr, ga.cstride, ga.coff, xa.ptr, xa.cstride, xa.coff,
         wts.data_ptr(), dg.ptr, dg.cstride, dg.coff, dx.ptr, dx.cstride, dx.coff, part.data_ptr(), S())
    torch.cuda.synchronize()
    close(dg.to_nchw()[:, :C], gt.grad, 1e-2, "gate8 backward: dg")
    close(dx.to_nchw()[:, :C], xt.grad, 1e-2, "gate8 backward: dx")
    gw = part.view(parts, 216).double().sum(0).float().cpu()
    for j in range(3):
        close(gw[64 * j:64 * j + 64].view(8, 8)[:C, :C], Ws[j].grad, 2e-3, f"gate8 backward: dW{j + 1}")
        close(gw[192 + 8 * j:192 + 8 * j + C], bs[j].grad, 2e-3, f"gate8 backward: db{j + 1}")


def test_flat_adam_under_grad_scaler_matches_torch():
    """scaler.step(optimizer); scaler.update() (training_utils.py:153-155) with FlatAdam, which unscales inside its Adam
    kernel and skips the step on an inf (uh_adam_step_amp), against torch.amp.GradScaler + torch.optim.Adam on the same
    scaled gradients: steps 1-2 clean, step 3 carries an inf (skipped by both, scale halved), step 4 clean again, so the
    bias corrections must count three steps, not four."""
    import torch.nn as nn
    from unet_amd.dp import FlatAdam

    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Parameter(torch.randn(37, 5))
            self.b = nn.Parameter(torch.randn(1001))

    class Flat:                      # the two attributes of GradBuckets that FlatAdam uses
        def __init__(self, model):
            self.order = [n for n, _ in model.named_parameters()]
            total = sum(p.numel() for p in model.parameters())
            self.flat = torch.zeros((total + 3) // 4 * 4, device=DEV)
            off = 0
            for p in model.parameters():
                p.grad = self.flat[off:off + p.numel()].view(p.shape)
                off += p.numel()

    torch.manual_seed(5)
    mine = Tiny().to(DEV)
    ref = Tiny().to(DEV)
    ref.load_state_dict(mine.state_dict())
    opt = FlatAdam(mine, Flat(mine), lr=1e-2)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    sc, rsc = torch.amp.GradScaler("cuda", init_scale=1024.0), torch.amp.GradScaler("cuda", init_scale=1024.0)
    g = torch.Generator(device=DEV).manual_seed(9)
    for step in range(4):
        scale = rsc.get_scale()
        assert sc.get_scale() == scale
        before = [p.detach().clone() for p in mine.parameters()]
        for p, q in zip(mine.parameters(), ref.parameters()):
            grad = torch.randn(p.shape, device=DEV, generator=g) * scale           # what scaled backward leaves
            if step == 2 and p.dim() == 1:
                grad[17] = float("inf")
            p.grad.copy_(grad)
            q.grad = grad.clone()
        sc.scale(torch.ones((), device=DEV))            # initialises the scaler's device state, as scale(loss) does
        rsc.scale(torch.ones((), device=DEV))
        sc.step(opt)
        sc.update()
        rsc.step(ropt)
        rsc.update()
        for p, q, b in zip(mine.parameters(), ref.parameters(), before):
            if step == 2:
                assert torch.equal(p, b)                 # skipped as a whole
            torch.testing.assert_close(p, q, rtol=2e-6, atol=2e-7)
    assert sc.get_scale() == rsc.get_scale() == 512.0
    assert opt._steps_dev.item() == 3.0
    # the flat gradient the scaler walked is the one buffer: its inf check saw every parameter's gradient
    assert opt.param_groups[0]["params"][0].grad.data_ptr() == opt.g.data_ptr()


def _fin_ws(parts, C):
    nbytes = lib.uh_fin_ws_bytes(parts, C, 2)
    return torch.full(((nbytes + 7) // 8,), float("nan"), dtype=torch.float64, device=DEV), nbytes


@pytest.mark.parametrize("B,H,W,C,c_real", [(2, 8, 8, 64, 64), (1, 4, 12, 128, 128), (2, 96, 80, 64, 64), (3, 2, 2, 512, 512),
                                            (8, 64, 64, 8, 3), (1, 2, 2, 1024, 1024)])
def test_sums_finalised_inside_the_launch_equal_the_two_launch_form(B, H, W, C, c_real):
    """uh_bn_relu_bwd_reduce_fin / uh_bn_res_relu_bwd_reduce_fin (the last workgroups of the launch add the partial rows
    up and write dgamma, dbeta and the two coefficients) against reduce + uh_bn_bwd_finalize: both add the same rows in
    fp64, in different groupings, so results agree to fp32 rounding; three runs in a row give identical bits (who adds
    depends on timing, what is added does not), leave the arrival counters zero, and write nothing past c_real."""
    g = torch.Generator().manual_seed(B * 7 + C)
    ya, st = _bn_bwd_setup(g, B, C, H, W)
    da = act_from(bf(torch.randn(B, C, H, W, generator=g)))
    xa = act_from(bf(torch.randn(B, C, H, W, generator=g)))
    count = B * H * W
    counters = torch.zeros(lib.uh_fin_counters(), dtype=torch.int32, device=DEV)
    parts = lib.uh_bn_bwd_parts(da.Mp, C)
    for residual in (False, True):
        rp = torch.empty(parts, 2, C, device=DEV)
        if residual:
            call("uh_bn_res_relu_bwd_reduce", da.Mp, C, da.ptr, da.cstride, da.coff, ya.ptr, ya.cstride, ya.coff, xa.ptr,
                 xa.cstride, xa.coff, *[t.data_ptr() for t in st], rp.data_ptr(), S())
        else:
            call("uh_bn_relu_bwd_reduce", da.Mp, C, da.ptr, da.cstride, da.coff, ya.ptr, ya.cstride, ya.coff,
                 *[t.data_ptr() for t in st], rp.data_ptr(), S())
        want = [torch.full((c_real + 2,), 9.0, device=DEV), torch.full((c_real + 2,), 9.0, device=DEV),
                torch.empty(2 * C, device=DEV)]
        call("uh_bn_bwd_finalize", rp.data_ptr(), parts, C, c_real, count, want[0].data_ptr(), want[1].data_ptr(),
             want[2].data_ptr(), 0, S())
        runs = []
        for _ in range(3):
            ws, nbytes = _fin_ws(parts, C)
            got = [torch.full((c_real + 2,), 9.0, device=DEV), torch.full((c_real + 2,), 9.0, device=DEV),
                   torch.full((2 * C,), float("nan"), device=DEV)]
            tail = (c_real, count, got[0].data_ptr(), got[1].data_ptr(), got[2].data_ptr(), ws.data_ptr(), nbytes,
                    counters.data_ptr(), S())
            if residual:
                call("uh_bn_res_relu_bwd_reduce_fin", da.Mp, C, da.ptr, da.cstride, da.coff, ya.ptr, ya.cstride, ya.coff,
                     xa.ptr, xa.cstride, xa.coff, *[t.data_ptr() for t in st], *tail)
            else:
                call("uh_bn_relu_bwd_reduce_fin", da.Mp, C, da.ptr, da.cstride, da.coff, ya.ptr, ya.cstride, ya.coff,
                     *[t.data_ptr() for t in st], *tail)
            torch.cuda.synchronize()
            assert counters.abs().sum().item() == 0
            runs.append(got)
        for a, b in zip(runs[0], runs[1]):
            assert torch.equal(a, b)
        for a, b in zip(runs[0], runs[2]):
            assert torch.equal(a, b)
        scale = max(w.abs().max().item() for w in want[:2])
        for a, w, what in zip(runs[0], want, ("dgamma", "dbeta", "coef")):
            assert torch.isfinite(a).all(), what
            assert (a - w).abs().max().item() <= 2e-6 * max(scale, w.abs().max().item()) + 1e-9, what
        assert runs[0][0][c_real:].tolist() == [9.0, 9.0] and runs[0][1][c_real:].tolist() == [9.0, 9.0]
    with pytest.raises(_lib.HipKernelError):
        ws, nbytes = _fin_ws(parts, C)
        call("uh_bn_relu_bwd_reduce_fin", da.Mp, C, da.ptr, da.cstride, da.coff, ya.ptr, ya.cstride, ya.coff,
             *[t.data_ptr() for t in st], c_real, count, want[0].data_ptr(), want[1].data_ptr(), want[2].data_ptr(),
             ws.data_ptr(), nbytes - 8, counters.data_ptr(), S())


@pytest.mark.parametrize("B,H,W,C,c_real", [(2, 16, 16, 64, 64), (4, 64, 64, 8, 3), (3, 5, 7, 24, 24), (2, 128, 128, 64, 64),
                                            (1, 48, 48, 512, 512)])
def test_depthwise_forward_with_batchnorm_finalised_inside(B, H, W, C, c_real):
    """uh_dwconv3x3_fwd_bn against uh_dwconv3x3_fwd (train mode) + uh_bn_finalize: the same fp16 output bit for bit;
    mean, invstd, scale, shift and the running statistics to fp32 rounding; num_batches_tracked counted once."""
    g = torch.Generator().manual_seed(B + H + C)
    x = bf(torch.randn(B, C, H, W, generator=g))
    if c_real < C:
        x[:, c_real:] = 0
    xa = act_from(x)
    w = (torch.randn(c_real, 1, 3, 3, generator=g) / 3).to(DEV)
    bias = torch.randn(c_real, generator=g).to(DEV)
    gamma, beta = (torch.rand(c_real, generator=g) + 0.5).to(DEV), torch.randn(c_real, generator=g).to(DEV)
    mask = rowmask(B, H, W)
    parts = lib.uh_dw_parts(B, H, W, C)
    counters = torch.zeros(lib.uh_fin_counters(), dtype=torch.int32, device=DEV)

    def fresh():
        return dict(y=new_act(B, H, W, C, DEV), f=torch.full((4, C), float("nan"), device=DEV),
                    rm=torch.zeros(c_real, device=DEV), rv=torch.ones(c_real, device=DEV),
                    nbt=torch.zeros((), dtype=torch.int64, device=DEV))
    a = fresh()
    fparts = lib.uh_dwconv3x3_fwd_parts(B, H, W, C)
    stat = torch.empty(fparts, 2, C, device=DEV)
    call("uh_dwconv3x3_fwd", B, H, W, C, xa.ptr, xa.cstride, xa.coff, w.data_ptr(), c_real, bias.data_ptr(),
         _lib.EPI_RAW_STATS, None, None, mask.data_ptr(), a["y"].ptr, a["y"].cstride, a["y"].coff, stat.data_ptr(), S())
    call("uh_bn_finalize", stat.data_ptr(), fparts, C, c_real, B * H * W, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1,
         a["f"][0].data_ptr(), a["f"][1].data_ptr(), a["f"][2].data_ptr(), a["f"][3].data_ptr(), a["rm"].data_ptr(),
         a["rv"].data_ptr(), a["nbt"].data_ptr(), S())
    outs = []
    for _ in range(2):
        b = fresh()
        ws, nbytes = _fin_ws(parts, C)
        call("uh_dwconv3x3_fwd_bn", B, H, W, C, xa.ptr, xa.cstride, xa.coff, w.data_ptr(), c_real, bias.data_ptr(),
             mask.data_ptr(), b["y"].ptr, b["y"].cstride, b["y"].coff, gamma.data_ptr(), beta.data_ptr(), 1e-5, 0.1,
             b["f"][0].data_ptr(), b["f"][1].data_ptr(), b["f"][2].data_ptr(), b["f"][3].data_ptr(), b["rm"].data_ptr(),
             b["rv"].data_ptr(), b["nbt"].data_ptr(), ws.data_ptr(), nbytes, counters.data_ptr(), S())
        torch.cuda.synchronize()
        assert counters.abs().sum().item() == 0
        outs.append(b)
    b = outs[0]
    assert torch.equal(b["y"].buf, a["y"].buf)
    assert b["nbt"].item() == 1 and a["nbt"].item() == 1
    for k in ("f", "rm", "rv"):
        assert torch.equal(outs[0][k], outs[1][k]), k                       # run-independent
        torch.testing.assert_close(b[k], a[k], rtol=3e-6, atol=1e-6)
    assert torch.isfinite(b["f"]).all()
    if c_real < C:                                                           # padding channels: scale = shift = 0
        assert b["f"][2, c_real:].abs().max().item() == 0 and b["f"][3, c_real:].abs().max().item() == 0


def test_repack_launch_writes_the_fp32_tables_of_the_8_channel_level():
    """descriptors with bit 1 of `reverse` set write fp32: the repack launch of a LUNeXt leaves the tables that
    uh_gate8_pack and uh_pad_rows_f32 make (gate weights and biases, the head's weight, the last transposed
    convolution's bias, zero-padded to 8), and both directions of a training step go in one launch"""
    import unet_amd
    from unet_amd.packing import lunext_packer_for
    torch.manual_seed(12)
    m = unet_amd.LUNeXt(3, 2).to(DEV)
    pk = lunext_packer_for(m, DEV)
    wf, wb = pk.get_both(S())
    att, up, fc = m.att4, m.upconv4, m.final_conv
    want = torch.full((216,), 7.0, device=DEV)
    call("uh_gate8_pack", 3, att.sigmoid_conv.weight.data_ptr(), att.sigmoid_conv.bias.data_ptr(),
         att.tanh_conv.weight.data_ptr(), att.tanh_conv.bias.data_ptr(), att.combine_conv.weight.data_ptr(),
         att.combine_conv.bias.data_ptr(), want.data_ptr(), S())
    assert torch.equal(wf["att4.gate8.f32"], want)
    want = torch.full((8,), 7.0, device=DEV)
    call("uh_pad_rows_f32", 1, 3, 8, up.bias.data_ptr(), want.data_ptr(), S())
    assert torch.equal(wf["up4.bias.f32"], want)
    want = torch.full((2, 8), 7.0, device=DEV)
    call("uh_pad_rows_f32", 2, 3, 8, fc.weight.data_ptr(), want.data_ptr(), S())
    assert torch.equal(wf["head.w.f32"].view(2, 8), want)
    # one launch for both directions == one launch per direction
    both = {d: {k: v.clone() for k, v in views.items()} for d, views in (("fwd", wf), ("bwd", wb))}
    pk.mark_dirty()
    for d in ("fwd", "bwd"):
        for v in pk.groups[d]["views"].values():
            v.fill_(3.0)
    f2, b2 = pk.get("fwd", S()), pk.get("bwd", S())
    torch.cuda.synchronize()
    for d, views in (("fwd", f2), ("bwd", b2)):
        for k, v in views.items():
            written = both[d][k] != 0                            # zero padding is never written by either
            assert torch.equal(v[written], both[d][k][written]), (d, k)
