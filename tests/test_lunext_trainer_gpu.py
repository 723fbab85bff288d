"""unet_amd.dp.LUNeXtTrainer (forward + loss + backward + Adam for LUNeXt without autograd: gradients written into one
flat buffer by the kernels that make them, one Adam launch, two launches for sigmoid / clamp and their backward)
against the module's autograd path with torch.optim.Adam, which the other LUNeXt tests hold to the oracle and to the
reference's fixtures.  Counterpart of the loop body of trainer/training_utils.py:131-170."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from conftest import gpu_present            # noqa: E402

if not gpu_present():
    pytest.skip("needs a GPU", allow_module_level=True)

import unet_amd                              # noqa: E402
from unet_amd._lib import call, lib          # noqa: E402
from unet_amd.dp import LUNeXtTrainer, FlatGrads     # noqa: E402
from unet_amd.lunext import flat_param_order         # noqa: E402

DEV = torch.device("cuda", 0)


def S():
    return torch.cuda.current_stream().cuda_stream


def batch(b, c, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(b, c, h, w, generator=g).to(DEV), (torch.rand(b, 1, h, w, generator=g) > 0.5).float().to(DEV)


def test_sigmoid_clamp_forward_and_backward_match_torch():
    g = torch.Generator().manual_seed(1)
    for n in (4 * 1000, 4 * 1000 + 3, 7):
        x = (torch.randn(n + 4, generator=g) * 9).to(DEV)[:n]            # saturates both ways (|x| > 13.8)
        x[:3] = torch.tensor([0.0, 40.0, -40.0], device=DEV)[:min(3, n)]
        dp = torch.randn(n, generator=g).to(DEV)
        p = torch.empty(n, device=DEV)
        dx = torch.empty(n, device=DEV)
        call("uh_sigmoid_clamp", n, x.data_ptr(), 1e-6, 1 - 1e-6, p.data_ptr(), S())
        call("uh_sigmoid_clamp_bwd", n, x.data_ptr(), 1e-6, 1 - 1e-6, dp.data_ptr(), dx.data_ptr(), S())
        xr = x.clone().requires_grad_(True)
        want = torch.sigmoid(xr).clamp(1e-6, 1 - 1e-6)
        want.backward(dp)
        torch.testing.assert_close(p, want.detach(), rtol=2e-6, atol=0)
        assert p.min().item() == pytest.approx(1e-6, rel=1e-6) and p.max().item() == pytest.approx(1 - 1e-6, rel=1e-6)
        # near saturation (1 - s) turns a one-ulp difference of s (expf against torch's exp) into ~1e-5 relative
        torch.testing.assert_close(dx, xr.grad, rtol=2e-4, atol=1e-12)
        assert dx[1].item() == 0.0 and dx[2].item() == 0.0           # clamped: no gradient


def test_colsum_finalize_gather_lays_sums_out_by_table():
    P, C = 37, 216
    g = torch.Generator().manual_seed(2)
    part = torch.randn(P, C, generator=g).to(DEV)
    cols = torch.tensor([5, 0, 215, 64, 64, 191, 8, 9, 10, 11, 200], dtype=torch.int32, device=DEV)
    out = torch.full((cols.numel() + 2,), 7.0, device=DEV)
    call("uh_colsum_finalize_gather", part.data_ptr(), P, C, cols.numel(), cols.data_ptr(), out.data_ptr(), S())
    want = part.double().sum(0)[cols.long()].float()
    torch.testing.assert_close(out[:-2], want, rtol=1e-6, atol=1e-6)
    assert out[-2:].tolist() == [7.0, 7.0]                           # nothing written past n_out


def test_flat_order_keeps_joint_gradients_side_by_side():
    model = unet_amd.LUNeXt(3, 1).to(DEV)
    order = flat_param_order(model)
    assert sorted(order) == sorted(n for n, _ in model.named_parameters())
    fg = FlatGrads(model, order, aligned=lambda n: n.endswith("depthwise.weight"))
    for k in (1, 2, 3, 4):
        C = getattr(model, f"att{k}").sigmoid_conv.weight.shape[0]
        o = fg.offset[f"att{k}.sigmoid_conv.weight"]
        assert [fg.offset[f"att{k}.{c}.weight"] for c in ("sigmoid_conv", "tanh_conv", "combine_conv")] == \
            [o, o + C * C, o + 2 * C * C]
        assert [fg.offset[f"att{k}.{c}.bias"] for c in ("sigmoid_conv", "tanh_conv", "combine_conv")] == \
            [o + 3 * C * C, o + 3 * C * C + C, o + 3 * C * C + 2 * C]
    assert fg.offset["final_conv.bias"] == fg.offset["final_conv.weight"] + 3
    for n, p in model.named_parameters():
        assert p.grad.data_ptr() == fg.views[n].data_ptr()
        if n.endswith("depthwise.weight"):
            assert fg.views[n].data_ptr() % 16 == 0


@pytest.mark.parametrize("b,c,h,w", [(2, 3, 32, 32), (2, 3, 44, 52), (3, 2, 48, 80)])
def test_trainer_gradients_equal_the_autograd_path(b, c, h, w):
    """With a criterion both paths differentiate by autograd at the logits, they run the same kernels on the same
    numbers: every parameter gradient must be bit-identical, the unused att*.output_conv ones exactly zero."""
    torch.manual_seed(3)
    a = unet_amd.LUNeXt(c, 1).to(DEV).train()
    tm = unet_amd.LUNeXt(c, 1).to(DEV).train()
    tm.load_state_dict(a.state_dict())
    x, m = batch(b, c, h, w, seed=h)
    crit = lambda lg, mk: ((torch.sigmoid(lg) - mk) ** 2).mean() + 0.1 * nn.BCEWithLogitsLoss()(lg, mk)   # noqa: E731
    crit(a(x), m).backward()
    trainer = LUNeXtTrainer(tm, lr=1e-3, criterion=crit, standalone=True)
    before = trainer.optimizer.p.clone()
    trainer.train_step(x, m)
    for n, p in a.named_parameters():
        got = trainer.grads.views[n]
        if p.grad is None:
            assert ".output_conv." in n and got.abs().max().item() == 0.0, n
            continue
        assert torch.equal(got, p.grad), (n, (got - p.grad).abs().max().item())
    # and the optimizer moved every parameter that has a gradient, none of the unused ones
    after = trainer.optimizer.p
    for n in trainer.grads.order:
        o = trainer.grads.offset[n]
        moved = not torch.equal(before[o:o + trainer.grads.views[n].numel()], after[o:o + trainer.grads.views[n].numel()])
        zero_grad = trainer.grads.views[n].abs().max().item() == 0.0
        assert moved == (not zero_grad), n


@pytest.mark.parametrize("criterion", ["cl_dice", "bce"])
def test_trainer_steps_follow_torch_adam_on_the_autograd_path(criterion):
    """Three steps.  The first step's gradients are the same numbers in both runs (previous test; for "cl_dice" up to
    expf against torch's exp), so the losses agree to ~1e-5 at step 0 and stay within 3e-3 per step afterwards.
    Parameters are only held to Adam's own bound (2 * lr apart per step): at random initialisation a good part of
    LUNeXt's 1.7 M weights have gradients at the level of the bf16 rounding noise of the forward pass, Adam's first
    steps move every weight by about lr * sign(gradient) whatever its size, and one-ulp differences after step 1
    re-draw that noise (measured: 10 % of the elements end more than 2e-4 apart with "bce", 26 % with "cl_dice",
    while the losses stay together)."""
    torch.manual_seed(4)
    a = unet_amd.LUNeXt(3, 1).to(DEV).train()
    tm = unet_amd.LUNeXt(3, 1).to(DEV).train()
    tm.load_state_dict(a.state_dict())
    x, m = batch(4, 3, 64, 64, seed=11)
    opt = torch.optim.Adam(a.parameters(), lr=1e-3)
    cl = unet_amd.soft_dice_cldice()
    trainer = LUNeXtTrainer(tm, lr=1e-3, criterion=criterion, standalone=True)
    for step in range(3):
        opt.zero_grad(set_to_none=True)
        logits = a(x)
        if criterion == "cl_dice":
            loss = cl(m, torch.sigmoid(logits).clamp(1e-6, 1 - 1e-6))          # training_utils.py:146-148
        else:
            loss = nn.BCEWithLogitsLoss()(logits, m)
        loss.backward()
        opt.step()
        got = trainer.train_step(x, m)
        assert abs(got.item() - loss.item()) <= 2e-5 * abs(loss.item()) + (3e-3 * step), (step, got.item(), loss.item())
    sd_a, sd_t = a.state_dict(), tm.state_dict()
    assert list(sd_a) == list(sd_t)
    params = {n for n, _ in a.named_parameters()}
    worst = 0.0
    for k in sd_a:
        if k in params:
            worst = max(worst, (sd_a[k] - sd_t[k]).abs().max().item())
        elif sd_a[k].dtype.is_floating_point:
            # running statistics follow the activations, which Adam's bound does not cover: same scale, that is all
            ref = sd_a[k].abs().max().item()
            assert (sd_a[k] - sd_t[k]).abs().max().item() <= 0.05 * ref + 0.02, k
    assert worst <= 3 * 2e-3 * 1.05, worst


def test_cldice_with_logits_matches_the_autograd_glue():
    from unet_amd.cl_dice import cldice_with_logits
    g = torch.Generator().manual_seed(6)
    logits = (torch.randn(3, 1, 40, 56, generator=g) * 4).to(DEV)
    mask = (torch.rand(3, 1, 40, 56, generator=g) > 0.6).float().to(DEV)
    lr = logits.clone().requires_grad_(True)
    want = unet_amd.soft_dice_cldice()(mask, torch.sigmoid(lr).clamp(1e-6, 1 - 1e-6))
    (want * 0.25).backward()
    loss, dl = cldice_with_logits(mask, logits, grad_scale=0.25)
    assert abs(loss.item() - want.item()) <= 1e-6 * abs(want.item())
    err = (dl - lr.grad).abs().max().item()
    assert err <= 2e-5 * lr.grad.abs().max().item(), (err, lr.grad.abs().max().item())


@pytest.mark.parametrize("own_stream", [False, True])
def test_cldice_labels_prepared_ahead_give_the_same_bits(own_stream):
    """cldice_prepare (the labels' skeleton, started before the prediction exists, on a second stream or not) followed by
    cldice_with_logits(prepared=...) == cldice_with_logits alone, twice in a row with different labels (the second call
    finds the first one's workspace in the allocator)."""
    from unet_amd.cl_dice import cldice_prepare, cldice_with_logits
    from unet_amd.engine import side_stream_for
    g = torch.Generator().manual_seed(16)
    side = side_stream_for(DEV) if own_stream else None
    for shape in ((3, 1, 40, 56), (3, 1, 40, 56), (2, 1, 33, 65)):
        logits = (torch.randn(*shape, generator=g) * 4).to(DEV)
        mask = (torch.rand(*shape, generator=g) > 0.6).float().to(DEV)
        prep = cldice_prepare(mask, stream=side)
        assert (prep.event is not None) == own_stream
        got = cldice_with_logits(mask, logits, grad_scale=0.5, prepared=prep)
        want = cldice_with_logits(mask, logits, grad_scale=0.5)
        torch.cuda.synchronize()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        del prep, got, want
    with pytest.raises(ValueError):
        cldice_with_logits(mask, logits, prepared=cldice_prepare(mask[:1]))


def test_trainer_with_and_without_prepared_labels(monkeypatch):
    """LUNeXtTrainer's clDice step with the labels' skeleton made beside the forward pass (default) and inside the loss
    call (UNET_AMD_CLDICE_PREPARE=0): same losses, same parameters, bit for bit."""
    x, m = batch(2, 3, 32, 48, seed=21)
    out = []
    for flag in ("1", "0"):
        monkeypatch.setenv("UNET_AMD_CLDICE_PREPARE", flag)
        torch.manual_seed(8)
        model = unet_amd.LUNeXt(3, 1).to(DEV).train()
        tr = LUNeXtTrainer(model, lr=1e-3, criterion="cl_dice", standalone=True)
        losses = [tr.train_step(x, m).item() for _ in range(3)]
        torch.cuda.synchronize()
        out.append((losses, {k: v.clone() for k, v in model.state_dict().items()}))
    assert out[0][0] == out[1][0]
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k
