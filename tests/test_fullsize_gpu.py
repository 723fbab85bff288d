"""Checks at the sizes of BASELINE.json's GPU configurations (batch 32, 3x256x256), where the fp32 CPU oracle
would take minutes per case: size-independent properties instead of element-wise comparison.

  * determinism: the HIP path has no atomics and no run-dependent summation order, so the same step twice gives
    bit-identical logits and gradients (U-Net and LUNeXt + clDice);
  * batch independence in eval mode: logits of a batch of 32 == logits of its two halves (to rounding: how a
    convolution's tiles are cut over the CUs depends on the batch size, which changes fp32 summation order);
  * linearity of the convolution kernel: conv(2x) == 2 conv(x) exactly (power-of-two scaling commutes with every
    bf16/fp32 rounding), forward and input-gradient mode, at a full-size layer shape;
  * train-mode BatchNorm identity: the gradient of a conv bias in front of BatchNorm is exactly 0, and every other
    parameter receives a finite, non-zero gradient;
  * the depthwise block's strip-walk kernels (forward and backward) give the bits of the kernels they replace at the
    LUNeXt level shapes of batch 32 (thousands of strips, every column block);
  * soft-skeleton order relations: erode(x) <= x <= dilate(x), 0 <= skel <= 1, skel(mask) supported on the mask,
    loss of a prediction equal to the (binary) target < loss of its complement.
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from conftest import gpu_present            # noqa: E402

if not gpu_present():
    pytest.skip("needs a GPU", allow_module_level=True)

import unet_amd                                        # noqa: E402
from unet_amd import _lib                              # noqa: E402
from unet_amd._lib import call, lib                    # noqa: E402
from unet_amd.layout import new_act, pool_for          # noqa: E402

DEV = torch.device("cuda", 0)
B, HW = 32, 256


def S():
    return torch.cuda.current_stream().cuda_stream


def batch(seed=7):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, HW, HW, generator=g)
    m = (torch.rand(B, 1, HW, HW, generator=g) > 0.5).float()
    return x.to(DEV), m.to(DEV)


def one_step(model, x, m, crit):
    model.zero_grad(set_to_none=True)
    logits = model(x)
    loss = crit(logits, m)
    loss.backward()
    return logits.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def test_unet_step_is_deterministic_and_bn_identities_hold():
    torch.manual_seed(0)
    model = unet_amd.UNet(3, 1).to(DEV).train()
    x, m = batch()
    bce = nn.BCEWithLogitsLoss()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    l1, g1 = one_step(model, x, m, bce)
    model.load_state_dict(state)                       # running statistics back to their initial values
    l2, g2 = one_step(model, x, m, bce)
    assert torch.equal(l1, l2)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    assert torch.isfinite(l1).all() and l1.abs().max().item() < 50
    # conv biases in front of train-mode BatchNorm: the gradient is identically zero
    for n, g in g1.items():
        if n.endswith(".bias") and n.startswith("conv"):
            assert g.abs().max().item() == 0.0, n
    # every parameter got a finite, non-trivial gradient
    for n, g in g1.items():
        assert torch.isfinite(g).all(), n
        if not (n.endswith(".bias") and n.startswith("conv")):
            assert g.abs().max().item() > 0, n


def test_unet_eval_is_batch_independent():
    torch.manual_seed(1)
    model = unet_amd.UNet(3, 1).to(DEV)
    x, _ = batch(11)
    model.train()
    with torch.no_grad():
        model(x[:8])                                   # move the running statistics off their initial values
    model.eval()
    with torch.no_grad():
        full = model(x)
        halves = torch.cat([model(x[:16]), model(x[16:])])
        single = model(x[5:6])
    scale = full.abs().max().item()
    assert (full - halves).abs().max().item() <= 1e-2 * scale
    assert (full[5:6] - single).abs().max().item() <= 1e-2 * scale
    with torch.no_grad():
        assert torch.equal(model(x), full)             # same batch size: bit-identical
    assert len(model.intermediate_features) == 15


@pytest.mark.parametrize("mode", ["forward", "input_gradient"])
def test_conv_kernel_is_linear_at_full_size(mode):
    """conv1_2 / conv_up4_2 shape: 32 x 256 x 256, 64 -> 64 channels (2.1 M rows)."""
    Bq, H, W, K, N = 32, 256, 256, 64, 64
    g = torch.Generator(device=DEV).manual_seed(3)
    a = new_act(Bq, H, W, K, DEV)
    rows = a.buf[a.guard:a.guard + a.Mp].view(Bq, H + 1, W + 1, K)
    rows[:, :H, :W, :] = torch.randn(Bq, H, W, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N * 9 * K, device=DEV, generator=g) / 24).bfloat16()
    mask = pool_for(DEV).rowmask(Bq, H, W)
    need = lib.uh_conv_gemm_workspace(Bq, H, W, K, N, 9)
    ws = torch.empty(max(need, 16) // 4, device=DEV)
    epi = _lib.EPI_RAW_STATS if mode == "forward" else _lib.EPI_MASKED
    parts = lib.uh_conv_gemm_mtiles(Bq, H, W, N)

    def run(inp):
        out = new_act(Bq, H, W, N, DEV)
        stat = torch.zeros(parts, 2, N, device=DEV)
        call("uh_conv_gemm_ws", Bq, H, W, inp.ptr, inp.cstride, 0, K, w.data_ptr(), N, 9, epi, out.ptr, out.cstride, 0,
             None, None, None, stat.data_ptr() if mode == "forward" else None, mask.data_ptr(), ws.data_ptr(), need, S())
        return out, stat

    o1, s1 = run(a)
    a2 = new_act(Bq, H, W, K, DEV)
    a2.buf.copy_(a.buf * 2)
    o2, s2 = run(a2)
    torch.cuda.synchronize()
    v1 = o1.buf.view(torch.float16) if mode == "forward" else o1.buf
    v2 = o2.buf.view(torch.float16) if mode == "forward" else o2.buf
    assert torch.isfinite(v1.float()).all()
    if mode == "forward":
        # fp16 storage: doubling commutes with the rounding except in the subnormal range (|v| < 2^-14), where
        # the spacing is the absolute 2^-24
        d = (v2.float() - v1.float() * 2).abs()
        assert d.max().item() <= 2.0 ** -24
        assert (d[v1.float().abs() >= 2.0 ** -13] == 0).all()
    else:
        assert torch.equal(v2.float(), v1.float() * 2)
    if mode == "forward":
        assert torch.equal(s2[:, 0], s1[:, 0] * 2) and torch.equal(s2[:, 1], s1[:, 1] * 4)
        # pad rows and guard rows stay zero
        r = o1.buf[o1.guard:o1.guard + o1.Mp].view(Bq, H + 1, W + 1, N)
        assert r[:, H].abs().max().item() == 0 and r[:, :, W].abs().max().item() == 0
        assert o1.buf[:o1.guard].abs().max().item() == 0 and o1.buf[o1.guard + o1.Mp:].abs().max().item() == 0


def test_lunext_cldice_step_is_deterministic():
    torch.manual_seed(2)
    model = unet_amd.LUNeXt(3, 1).to(DEV).train()
    x, m = batch(5)
    crit = unet_amd.soft_dice_cldice()

    def loss_fn(logits, masks):
        return crit(masks, torch.sigmoid(logits).clamp(1e-6, 1 - 1e-6))       # training_utils.py:146-148

    state = {k: v.clone() for k, v in model.state_dict().items()}
    l1, g1 = one_step(model, x, m, loss_fn)
    model.load_state_dict(state)
    l2, g2 = one_step(model, x, m, loss_fn)
    assert torch.equal(l1, l2)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    for k in range(1, 5):                              # AttentionGate.output_conv is never used (lunext.py:76-122)
        assert model.get_parameter(f"att{k}.output_conv.weight").grad is None


def test_soft_skeleton_order_relations_at_full_size():
    g = torch.Generator(device=DEV).manual_seed(9)
    p = torch.rand(B, 1, HW, HW, device=DEV, generator=g)
    n = B * HW * HW
    er, di, sk = torch.empty_like(p), torch.empty_like(p), torch.empty_like(p)
    call("uh_soft_erode", B, HW, HW, p.data_ptr(), er.data_ptr(), 0, S())
    call("uh_soft_dilate", B, HW, HW, p.data_ptr(), di.data_ptr(), S())
    ws = torch.empty(lib.uh_soft_skel_workspace(B, HW, HW) // 4, device=DEV)
    call("uh_soft_skel", B, HW, HW, p.data_ptr(), sk.data_ptr(), ws.data_ptr(), S())
    torch.cuda.synchronize()
    assert (er <= p).all() and (p <= di).all()
    assert sk.min().item() >= 0 and sk.max().item() <= 1
    # spot check against torch on the device: same pooling definitions as cl_dice.py:12-27
    er_t = torch.min(-F.max_pool2d(-p, (3, 1), (1, 1), (1, 0)), -F.max_pool2d(-p, (1, 3), (1, 1), (0, 1)))
    assert torch.equal(er, er_t) and torch.equal(di, F.max_pool2d(p, (3, 3), (1, 1), (1, 1)))
    # a binary mask: its skeleton lives on the mask
    mask = torch.zeros(B, 1, HW, HW, device=DEV)
    for y0 in range(8, HW - 8, 32):                    # bars seven pixels thick, and a few blobs
        mask[:, :, y0:y0 + 7, 16:HW - 16] = 1
    mask[:, :, 100:140, 60:100] = 1
    call("uh_soft_skel", B, HW, HW, mask.data_ptr(), sk.data_ptr(), ws.data_ptr(), S())
    torch.cuda.synchronize()
    assert (sk <= mask).all() and sk.sum().item() > 0
    crit = unet_amd.soft_dice_cldice()
    good = crit(mask, mask.clamp(1e-6, 1 - 1e-6)).item()
    bad = crit(mask, (1 - mask).clamp(1e-6, 1 - 1e-6)).item()
    assert 0 <= good < 0.05 and bad > 0.9, (good, bad)
    assert n == p.numel()


@pytest.mark.parametrize("b,hw", [(2, 512), (1, 1024), (16, 512), (4, 1024)])
def test_unet_large_images(b, hw):
    """BASELINE configs[3] and [4] image sizes (512^2 training, 1024^2 training + batch-1 inference), on small batches
    and on the batches those configurations train with per GPU (16 at 512^2, 4 at 1024^2):
    a training step is deterministic and finite, and eval-mode inference of one image does not depend on the batch
    it arrives in (to rounding)."""
    torch.manual_seed(3)
    model = unet_amd.UNet(3, 1).to(DEV).train()
    g = torch.Generator().manual_seed(hw)
    x = torch.rand(b, 3, hw, hw, generator=g).to(DEV)
    m = (torch.rand(b, 1, hw, hw, generator=g) > 0.5).float().to(DEV)
    bce = nn.BCEWithLogitsLoss()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    l1, g1 = one_step(model, x, m, bce)
    model.load_state_dict(state)
    l2, g2 = one_step(model, x, m, bce)
    assert l1.shape == (b, 1, hw, hw) and torch.isfinite(l1).all()
    assert torch.equal(l1, l2)
    for n in g1:
        assert torch.isfinite(g1[n]).all(), n
        assert torch.equal(g1[n], g2[n]), n
        if n.endswith(".bias") and n.startswith("conv"):       # in front of train-mode BatchNorm: exactly zero
            assert g1[n].abs().max().item() == 0.0, n
        else:
            assert g1[n].abs().max().item() > 0, n
    model.eval()
    with torch.no_grad():
        full = model(x)
        single = model(x[:1])
    scale = full.abs().max().item()
    assert (full[:1] - single).abs().max().item() <= 1e-2 * scale
    # the evaluation driver at this size: counts add up to the number of pixels
    from unet_amd import trainer
    counts = torch.zeros(4, dtype=torch.int64, device=DEV)
    trainer.confusion_counts(full, m, counts)
    assert counts.sum().item() == b * hw * hw


@pytest.mark.parametrize("H,C,c_real", [(256, 8, 3), (128, 64, 64), (32, 256, 256)])
def test_depthwise_strip_walks_give_the_column_kernels_bits_at_full_size(H, C, c_real, monkeypatch):
    """LUNeXt level shapes at batch 32: uh_dwconv3x3_fwd (training) and uh_dwblock_bwd as per-wave strip walks against
    the one-thread-per-column kernels (UNET_AMD_DWFWD_TILES=1, UNET_AMD_DWBLOCK_LDS=1): y and dx bit for bit, the
    reduced sums to summation order."""
    g = torch.Generator(device=DEV).manual_seed(H + C)
    live = (torch.arange(C, device=DEV) < c_real).float().view(1, C, 1, 1)
    rnd = lambda: torch.randn(B, C, H, H, generator=g, device=DEV) * live          # noqa: E731
    x, da, y = new_act(B, H, H, C, DEV), new_act(B, H, H, C, DEV), new_act(B, H, H, C, DEV)
    x.fill_from_nchw(rnd())
    da.fill_from_nchw(rnd())
    y.fill_from_nchw(rnd() * 1.5, half=True)
    wt = torch.randn(c_real, 1, 3, 3, generator=g, device=DEV) / 3
    bias = torch.randn(c_real, generator=g, device=DEV)
    st = [torch.randn(C, generator=g, device=DEV) * 0.2, torch.rand(C, generator=g, device=DEV) + 0.5,
          torch.randn(C, generator=g, device=DEV) * live.view(-1), torch.randn(C, generator=g, device=DEV) * 0.3 * live.view(-1)]
    coef = torch.randn(2, C, generator=g, device=DEV) * 0.05
    mask = pool_for(DEV).rowmask(B, H, H)
    outs = {}
    for mode in ("strip", "column"):
        if mode == "column":
            monkeypatch.setenv("UNET_AMD_DWFWD_TILES", "1")
            monkeypatch.setenv("UNET_AMD_DWBLOCK_LDS", "1")
        fp, bp = lib.uh_dwconv3x3_fwd_parts(B, H, H, C), lib.uh_dwblock_bwd_parts(B, H, H, C)
        yo, dx = new_act(B, H, H, C, DEV), new_act(B, H, H, C, DEV)
        stat, wp = torch.empty(fp, 2, C, device=DEV), torch.empty(bp, 9, C, device=DEV)
        call("uh_dwconv3x3_fwd", B, H, H, C, x.ptr, x.cstride, x.coff, wt.data_ptr(), c_real, bias.data_ptr(),
             _lib.EPI_RAW_STATS, None, None, mask.data_ptr(), yo.ptr, yo.cstride, yo.coff, stat.data_ptr(), S())
        call("uh_dwblock_bwd", B, H, H, C, da.ptr, da.cstride, da.coff, y.ptr, y.cstride, y.coff, x.ptr, x.cstride,
             x.coff, wt.data_ptr(), c_real, *[t.data_ptr() for t in st], coef.data_ptr(), dx.ptr, dx.cstride, dx.coff,
             wp.data_ptr(), S())
        dw = torch.empty(c_real, 1, 3, 3, device=DEV)
        call("uh_dw_wgrad_finalize", wp.data_ptr(), bp, C, c_real, dw.data_ptr(), S())
        torch.cuda.synchronize()
        outs[mode] = (yo, dx, stat.double().sum(0), dw)
    (y0, dx0, s0, dw0), (y1, dx1, s1, dw1) = outs["strip"], outs["column"]
    assert torch.equal(y0.buf, y1.buf), "forward"
    assert torch.equal(dx0.buf, dx1.buf), "input gradient"
    torch.testing.assert_close(s0, s1, rtol=1e-6, atol=1e-3)
    torch.testing.assert_close(dw0, dw1, rtol=1e-4, atol=1e-4 * dw1.abs().max().item())
