"""End-to-end parity of the HIP LUNeXt (unet_amd.LUNeXt) with the reference: the golden fixture captured
from the reference import (B=2, 3x32x32, soft_dice_cldice loss) and the CPU oracle on seeded inputs.
Storage is bf16 (fp16 for BatchNorm inputs) with fp32 arithmetic, as for the U-Net; LUNeXt is 46 layers deep
but each block is residual, which keeps the error lower than the U-Net's."""
import copy
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from conftest import gpu_present            # noqa: E402

if not gpu_present():
    pytest.skip("needs a GPU", allow_module_level=True)

import unet_amd                              # noqa: E402
from oracle import ref_cpu                   # noqa: E402

DEV = torch.device("cuda", 0)
LOGIT_REL = 3e-2
LOSS_REL = 5e-3
GRAD_COS = 0.97
GRAD_NORM = 0.08


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def check_grads(model, ref_grads, cos_min=GRAD_COS, norm_tol=GRAD_NORM, min_numel=1):
    total = max(g.norm().item() for g in ref_grads.values() if g is not None)
    worst_c, worst_n = 1.0, 0.0
    for n, p in model.named_parameters():
        ref = ref_grads.get(n, "missing")
        if isinstance(ref, str):
            continue
        if ref is None:
            assert p.grad is None, f"{n}: reference has no gradient (unused parameter)"
            continue
        assert p.grad is not None, n
        got = p.grad.detach().float().cpu()
        assert got.shape == ref.shape and torch.isfinite(got).all(), n
        if n.endswith("depthwise.bias"):
            assert got.abs().max().item() <= 1e-3 * total, n      # conv bias in front of train-mode BN
            continue
        if ref.norm().item() < 1e-3 * total or ref.numel() < min_numel:
            continue
        c, r = cosine(got, ref), abs(got.norm().item() / ref.norm().item() - 1)
        worst_c, worst_n = min(worst_c, c), max(worst_n, r)
        assert c >= cos_min, f"{n}: cosine {c:.4f}"
        assert r <= norm_tol, f"{n}: norm error {r:.4f}"
    return worst_c, worst_n


def test_lunext_keys(golden_dir):
    with open(os.path.join(golden_dir, "lunext_keys.json")) as f:
        want = json.load(f)
    got = [[k, list(v.shape), str(v.dtype)] for k, v in unet_amd.LUNeXt(3, 1).state_dict().items()]
    assert got == want and len(got) == 176


def test_lunext_golden_small(golden_dir):
    fx = np.load(os.path.join(golden_dir, "lunext_s0_b2_32.npz"), allow_pickle=False)
    torch.manual_seed(0)
    model = unet_amd.create_lunext(3, 1)
    cs = np.array([[p.detach().double().sum().item(), p.detach().double().abs().sum().item()]
                   for p in model.parameters()])
    np.testing.assert_allclose(cs, fx["param_checksums"], rtol=1e-12)
    model = model.to(DEV).train()
    x, m = torch.from_numpy(fx["x"]).to(DEV), torch.from_numpy(fx["mask"]).to(DEV)
    logits = model(x)
    probs = torch.sigmoid(logits).clamp(1e-6, 1 - 1e-6)          # training_utils.py:146
    loss = unet_amd.soft_dice_cldice()(m, probs)
    loss.backward()
    want = torch.from_numpy(fx["train_logits"])
    err = (logits.detach().cpu() - want).abs().max().item()
    print(f"logits max err {err:.4g} (max |ref| {want.abs().max().item():.4g}); loss {loss.item():.6f} vs {float(fx['train_loss']):.6f}")
    assert err <= LOGIT_REL * want.abs().max().item()
    assert abs(loss.item() - float(fx["train_loss"])) <= LOSS_REL * float(fx["train_loss"])
    # gradients.  With B=2 at 32x32 the bridge BatchNorm sees 8 samples per channel and the gradients around it are
    # ill-conditioned: the fp32 oracle with 16-bit storage roundings inserted reaches cosines of only 0.03-0.67
    # against the fixture for downconv4 / conv5_1.  So absolute thresholds are asserted only away from the bridge,
    # the fp32 oracle is pinned to the fixture, and the HIP path must be no further from the fixture than twice
    # the emulation is (the yardstick).
    ref_grads = {k[5:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("grad/")}
    o = three_way(2, 3, 32, 32, "cldice", seed=0, x=x.cpu(), m=m.cpu())
    for n, g in ref_grads.items():
        got = o["ref"]["grads"][n]
        if g.numel() >= 64 and not n.endswith("depthwise.bias"):
            assert cosine(got, g) >= 0.999, f"oracle vs fixture {n}"
    away = {n: g for n, g in ref_grads.items() if not n.startswith(("downconv4", "conv5_1"))}
    wc, wn = check_grads(model, away, cos_min=0.75, norm_tol=0.50, min_numel=64)   # a 3-element cosine says little
    big = [n for n, g in away.items() if g is not None and g.numel() >= 64 and not n.endswith("depthwise.bias")]
    got_all = torch.cat([model.get_parameter(n).grad.detach().float().cpu().flatten() for n in big])
    ref_all = torch.cat([away[n].flatten() for n in big])
    call_cos = cosine(got_all, ref_all)
    emu_all = torch.cat([o["emu"]["grads"][n].flatten() for n in big])
    emu_cos = cosine(emu_all, ref_all)
    print(f"grads vs fixture (away from the bridge): worst cosine {wc:.5f}, worst norm error {wn:.4f}, "
          f"all tensors as one vector: cosine {call_cos:.5f} (16-bit-storage emulation of the oracle: {emu_cos:.5f})")
    assert 1.0 - call_cos <= 2.0 * (1.0 - emu_cos) + 5e-3
    yardstick(o, "golden inputs", min_numel=64, skip=("downconv4", "conv5_1"))
    model.eval()
    with torch.no_grad():
        ev = model(x).cpu()
    want_ev = torch.from_numpy(fx["eval_logits"])
    assert (ev - want_ev).abs().max().item() <= LOGIT_REL * want_ev.abs().max().item()


def test_lunext_size_not_divisible_by_16(golden_dir):
    """44x52 -> 22x26 -> 11x13 -> 5x6 -> 2x3: the k2 s2 convolutions drop a row / column twice, and the two deepest
    decoder stages resize what the transposed convolution returns (lunext.py:226-235).  Against the fixture captured
    from the reference and, for the gradients, against the oracle with the emulation as yardstick."""
    fx = np.load(os.path.join(golden_dir, "lunext_s1_b2_44x52.npz"), allow_pickle=False)
    x, m = torch.from_numpy(fx["x"]), torch.from_numpy(fx["mask"])
    o = three_way(2, 3, 44, 52, "cldice", seed=1, x=x, m=m)
    # the same weights as the fixture (bit-equal on the machine the fixture was made on: tests/test_oracle_golden.py)
    np.testing.assert_allclose(o["ref"]["logits"].numpy(), fx["train_logits"], rtol=1e-4, atol=1e-5)
    want = torch.from_numpy(fx["train_logits"])
    err = (o["hip"]["logits"] - want).abs().max().item()
    print(f"logits max err {err:.4g} (max |ref| {want.abs().max().item():.4g})")
    assert err <= LOGIT_REL * want.abs().max().item()
    assert abs(o["hip"]["loss"] - float(fx["train_loss"])) <= LOSS_REL * float(fx["train_loss"])
    # B = 2 at 2x3: the bridge BatchNorm sees 12 samples per channel (see test_lunext_golden_small)
    yardstick(o, "44x52", min_numel=64, skip=("downconv4", "conv5_1", "conv5_2"))
    model = o["hip"]["net"].eval()
    with torch.no_grad():
        ev = model(x.to(DEV)).cpu()
    want_ev = torch.from_numpy(fx["eval_logits"])
    assert (ev - want_ev).abs().max().item() <= LOGIT_REL * want_ev.abs().max().item()


@pytest.mark.parametrize("H,W", [(17, 16), (30, 50), (100, 76)])
def test_lunext_odd_sizes_against_the_oracle(H, W):
    o = three_way(2, 3, H, W, "bce", seed=5)
    err_hip, scale = yardstick(o, f"{H}x{W}", min_numel=64, skip=("downconv4", "conv5_1", "conv5_2", "upconv1"))
    assert err_hip <= LOGIT_REL * scale


def three_way(B, C, H, W, loss, seed=4, x=None, m=None):
    """one training step of the fp32 oracle, the oracle with 16-bit storage roundings, and the HIP module"""
    from oracle import emu_bf16
    torch.manual_seed(seed)
    ref = ref_cpu.RefLUNeXt(C, 1)
    emu = emu_bf16.EmuLUNeXt(C, 1)
    emu.load_state_dict(ref.state_dict())
    model = copy.deepcopy(unet_amd.LUNeXt(C, 1))
    model.load_state_dict(ref.state_dict())
    model = model.to(DEV)
    if x is None:
        x, m = ref_cpu.synth_batch(B, C, H, W)
    outs = {}
    for name, net, dev in (("ref", ref, "cpu"), ("emu", emu, "cpu"), ("hip", model, DEV)):
        net.train()
        lo = net(x.to(dev))
        if loss == "cldice":
            crit = unet_amd.soft_dice_cldice() if name == "hip" else ref_cpu.RefSoftDiceClDice()
            lv = crit(m.to(dev), torch.sigmoid(lo).clamp(1e-6, 1 - 1e-6))
        else:
            lv = nn.BCEWithLogitsLoss()(lo, m.to(dev))
        lv.backward()
        outs[name] = dict(logits=lo.detach().cpu(), loss=lv.item(), net=net,
                          grads={n: (None if p.grad is None else p.grad.detach().cpu()) for n, p in net.named_parameters()})
    return outs


def yardstick(o, what, min_numel=1, skip=()):
    """HIP may be at most twice as far from the fp32 oracle as the 16-bit-storage emulation of the oracle is
    (summed over the gradient tensors; three times for a single tensor).
    (min_numel / skip: tensors left out -- one noise sample of a 3-element gradient is no yardstick, and neither
    is a gradient that the emulation itself gets uncorrelated with the oracle.)"""
    ref, emu, hip = o["ref"], o["emu"], o["hip"]
    scale = ref["logits"].abs().max().item()
    err_hip = (hip["logits"] - ref["logits"]).abs().max().item()
    err_emu = (emu["logits"] - ref["logits"]).abs().max().item()
    assert err_hip <= 2.0 * err_emu + 1e-3 * scale, (err_hip, err_emu)
    assert abs(hip["loss"] - ref["loss"]) <= LOSS_REL * abs(ref["loss"])
    total = max(g.norm().item() for g in ref["grads"].values() if g is not None)
    worst, sum_hip, sum_emu = 0.0, 0.0, 0.0
    for n, g_ref in ref["grads"].items():
        g_hip, g_emu = hip["grads"][n], emu["grads"][n]
        if g_ref is None:
            assert g_hip is None, f"{n}: the reference gives this parameter no gradient"
            continue
        assert g_hip is not None and torch.isfinite(g_hip).all(), n
        if n.endswith("depthwise.bias"):
            assert g_hip.abs().max().item() <= 1e-3 * total, n     # analytically zero under train-mode BN
            continue
        if g_ref.numel() < min_numel or n.startswith(skip):
            continue
        d_hip, d_emu = 1.0 - cosine(g_hip, g_ref), 1.0 - cosine(g_emu, g_ref)
        n_hip = abs(g_hip.norm().item() / g_ref.norm().item() - 1.0)
        n_emu = abs(g_emu.norm().item() / g_ref.norm().item() - 1.0)
        worst = max(worst, d_hip / (d_emu + 2e-3))
        sum_hip, sum_emu = sum_hip + d_hip, sum_emu + d_emu
        # per tensor both numbers are ONE sample of rounding noise each (a different summation order inside a
        # kernel moves them by tens of per cent in the small cases), so the per-tensor bound is a factor 3 and the
        # factor 2 is asked of the sum over all tensors, which is stable.  A gradient of two or three elements (the
        # BatchNorm parameters of the full-resolution level) is the noisiest sample of all: factor 4 there (the
        # two-channel case moved from 2.1 to 3.4 times the emulation's distance when the depthwise forward changed the
        # order in which it adds up its BatchNorm sums, and nothing else)
        f1 = 3.0 if g_ref.numel() >= 8 else 4.0
        assert d_hip <= f1 * d_emu + 2e-3, f"{n}: 1-cos hip {d_hip:.4g} vs emulation {d_emu:.4g}"
        assert n_hip <= 3.0 * n_emu + 5e-2, f"{n}: norm error hip {n_hip:.4g} vs emulation {n_emu:.4g}"
    assert sum_hip <= 2.0 * sum_emu + 2e-3, f"sum of 1-cos over tensors: hip {sum_hip:.4g} vs emulation {sum_emu:.4g}"
    print(f"[{what}] loss {hip['loss']:.6f} vs {ref['loss']:.6f}; logit err hip {err_hip:.4g} emu {err_emu:.4g} "
          f"(max {scale:.3g}); worst (1-cos hip)/(1-cos emu + 2e-3) = {worst:.3f}")
    return err_hip, scale


def test_lunext_single_input_channel():
    """The other end of the 8-channel first level: one real channel.  Its one-element gradients (the first level's
    BatchNorm parameters, the last transposed convolution's bias) are single noise samples of the 16-bit storage and
    are left out of the yardstick; everything else is held to it."""
    o = three_way(2, 1, 48, 48, "bce")
    yardstick(o, "2x1x48x48 bce", min_numel=2)


@pytest.mark.parametrize("B,C,H,W,loss", [(4, 3, 64, 64, "cldice"), (2, 3, 48, 80, "bce"), (2, 2, 32, 32, "bce"),
                                          (2, 3, 128, 128, "cldice"),
                                          # 8 input channels fill the first level (no padded bias anywhere)
                                          (2, 8, 64, 64, "bce")])
def test_lunext_train_step_matches_oracle(B, C, H, W, loss):
    o = three_way(B, C, H, W, loss)
    err, scale = yardstick(o, f"{B}x{C}x{H}x{W} {loss}")
    if B * (H // 16) * (W // 16) >= 32:          # BatchNorm at the bridge sees >= 32 samples: absolute numbers too
        assert err <= LOGIT_REL * scale
        wc, wn = check_grads(o["hip"]["net"], o["ref"]["grads"], cos_min=0.93, norm_tol=0.10)
        print(f"   absolute: worst cosine {wc:.5f}, worst norm err {wn:.4f}")
    bufs = dict(o["hip"]["net"].named_buffers())
    for n, buf in o["ref"]["net"].named_buffers():
        got = bufs[n].cpu()
        if n.endswith("num_batches_tracked"):
            assert got.item() == buf.item()
        elif n.endswith("running_mean"):
            assert (got - buf).abs().max().item() <= 2e-2, n
        else:
            assert ((got - buf).abs() <= 5e-2 * buf.abs() + 1e-3).all(), n


def test_lunext_eval_and_surface():
    torch.manual_seed(5)
    ref = ref_cpu.RefLUNeXt(3, 1)
    x, _ = ref_cpu.synth_batch(2, 3, 64, 48)
    ref.train()
    with torch.no_grad():
        for _ in range(3):
            ref(x + 0.1 * torch.randn_like(x))
    model = unet_amd.LUNeXt(3, 1)
    model.load_state_dict(ref.state_dict())
    model = model.to(DEV).eval()
    ref.eval()
    with torch.no_grad():
        want = ref(x)
        got = model(x.to(DEV)).cpu()
        assert (got - want).abs().max().item() <= LOGIT_REL * want.abs().max().item()
        one = model(x[:1].to(DEV)).cpu()
        assert (one - want[:1]).abs().max().item() <= LOGIT_REL * want.abs().max().item()
    with pytest.raises(ValueError, match="at least 16"):
        model(torch.rand(1, 3, 20, 12, device=DEV))
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        model(torch.rand(1, 3, 16, 16))
