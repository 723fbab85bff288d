"""clDice kernels against the fixtures captured from the reference's losses/cl_dice.py
(tests/golden/cldice_cases.npz: random, non-square, tie-heavy constant / quantised / saturated inputs,
a vessel-like blob, inputs outside [0,1]) and against the CPU oracle at a larger size.
Everything runs in fp32: morphology and skeleton planes must be bit-exact; the loss and its gradient
differ only by summation order of the seven global sums (tolerance 1e-5 relative)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import gpu_present            # noqa: E402

if not gpu_present():
    pytest.skip("needs a GPU", allow_module_level=True)

import unet_amd                              # noqa: E402
from oracle import ref_cpu                   # noqa: E402

DEV = torch.device("cuda", 0)
REL = 1e-5


def test_cldice_golden_cases(golden_dir):
    fx = np.load(os.path.join(golden_dir, "cldice_cases.npz"), allow_pickle=False)
    crit = unet_amd.soft_dice_cldice()
    sk = unet_amd.SoftSkeletonize(num_iter=10)
    for name in fx["case_names"]:
        probs = torch.from_numpy(fx[f"{name}/probs"]).to(DEV).requires_grad_(True)
        mask = torch.from_numpy(fx[f"{name}/mask"]).to(DEV)
        loss = crit(mask, probs)
        loss.backward()
        want = float(fx[f"{name}/loss"])
        assert abs(loss.item() - want) <= REL * abs(want), (name, loss.item(), want)
        g, gw = probs.grad.cpu().numpy(), fx[f"{name}/grad"]
        scale = np.abs(gw).max()
        bad = np.abs(g - gw) > REL * scale + 1e-12
        assert not bad.any(), f"{name}: {bad.sum()} of {bad.size} gradient entries differ, max err " \
                              f"{np.abs(g - gw).max():.3g} (scale {scale:.3g})"
        pc = probs.detach().clamp(0, 1)
        np.testing.assert_array_equal(sk.soft_erode(pc).cpu().numpy(), fx[f"{name}/erode"], err_msg=name)
        np.testing.assert_array_equal(sk.soft_dilate(pc).cpu().numpy(), fx[f"{name}/dilate"], err_msg=name)
        np.testing.assert_array_equal(sk.soft_skel(pc).cpu().numpy(), fx[f"{name}/skel_pred"], err_msg=name)
        np.testing.assert_array_equal(sk(mask).cpu().numpy(), fx[f"{name}/skel_true"], err_msg=name)
        d = unet_amd.soft_dice(mask, probs.detach())
        assert abs(d.item() - float(fx[f"{name}/dice"])) <= REL, name


@pytest.mark.parametrize("B,H,W", [(4, 128, 128), (2, 100, 76), (1, 33, 65)])
def test_cldice_matches_oracle(B, H, W):
    """sizes that are not multiples of the 32x32 backward tile, through sigmoid().clamp() as the trainer does."""
    g = torch.Generator().manual_seed(B * 100 + H)
    logits = torch.randn(B, 1, H, W, generator=g) * 3
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.6).float()
    lr = logits.clone().requires_grad_(True)
    loss_ref = ref_cpu.RefSoftDiceClDice()(mask, ref_cpu.probs_from_logits(lr))
    loss_ref.backward()
    lg = logits.to(DEV).requires_grad_(True)
    probs = torch.sigmoid(lg).clamp(1e-6, 1 - 1e-6)             # training_utils.py:146
    loss = unet_amd.soft_dice_cldice()(mask.to(DEV), probs)
    (loss * 3.0).backward()                                       # upstream gradient != 1 (GradScaler-like)
    assert abs(loss.item() - loss_ref.item()) <= REL * loss_ref.item()
    gw = 3.0 * lr.grad
    err = (lg.grad.cpu() - gw).abs().max().item()
    assert err <= 2 * REL * gw.abs().max().item(), (err, gw.abs().max().item())


def test_cldice_multichannel_and_errors():
    g = torch.Generator().manual_seed(9)
    p = torch.rand(2, 3, 24, 24, generator=g)
    t = (torch.rand(2, 3, 24, 24, generator=g) > 0.5).float()
    pr = p.clone().requires_grad_(True)
    lr = ref_cpu.RefSoftDiceClDice(alpha=0.3, smooth=2.0)(t, pr)
    lr.backward()
    pg = p.to(DEV).requires_grad_(True)
    lg = unet_amd.soft_dice_cldice(alpha=0.3, smooth=2.0)(t.to(DEV), pg)
    lg.backward()
    assert abs(lg.item() - lr.item()) <= REL * lr.item()
    assert (pg.grad.cpu() - pr.grad).abs().max().item() <= 2 * REL * pr.grad.abs().max().item()
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        unet_amd.soft_dice_cldice()(t, p)
    with pytest.raises(NotImplementedError):
        unet_amd.soft_dice_cldice()(t.to(DEV)[None], p.to(DEV)[None])


@pytest.mark.parametrize("B,H,W", [(2, 7, 5), (1, 64, 58), (2, 33, 59), (1, 41, 117), (3, 100, 76), (2, 256, 256)])
def test_backward_strip_walk_equals_tile_kernel(B, H, W, monkeypatch):
    """The backward pass has two kernels for a level: the register strip walk (default; a wave per 58-column strip,
    R core rows) and the LDS-tile kernel (UNET_AMD_SKEL_TILES=1).  They evaluate the same terms in the same order, so
    the gradients agree bit for bit: on widths around the strip width, heights that no strip length divides, and
    tie-heavy inputs (probabilities quantised to 1/8, saturated plateaus), for several strip lengths."""
    g = torch.Generator().manual_seed(H * 1000 + W)
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.55).float().to(DEV)
    probs = torch.rand(B, 1, H, W, generator=g)
    probs[0] = (probs[0] * 8).round() / 8                    # ties everywhere
    if B > 1:
        probs[1, :, : H // 2] = 1.0                          # a saturated plateau
    probs = probs.clamp(1e-6, 1 - 1e-6).to(DEV).requires_grad_(True)
    crit = unet_amd.soft_dice_cldice()
    monkeypatch.setenv("UNET_AMD_SKEL_TILES", "1")
    (want,) = torch.autograd.grad(crit(mask, probs) * 2.5, probs)
    monkeypatch.delenv("UNET_AMD_SKEL_TILES")
    for R in (None, "1", "5", "8", "16", "64"):
        if R is None:
            monkeypatch.delenv("UNET_AMD_SKEL_R", raising=False)
        else:
            monkeypatch.setenv("UNET_AMD_SKEL_R", R)
        (got,) = torch.autograd.grad(crit(mask, probs) * 2.5, probs)
        assert torch.equal(got, want), (R, (got - want).abs().max().item())
    assert want.abs().max().item() > 0


@pytest.mark.parametrize("B,H,W", [(2, 7, 5), (1, 64, 40), (2, 33, 41), (1, 50, 117), (3, 100, 76), (2, 256, 256)])
def test_forward_strip_walk_equals_level_kernels(B, H, W, monkeypatch):
    """The soft skeleton has two forward paths: all eleven levels in one strip walk (default; a wave per 40-column strip
    with a 12-column halo) and one launch per level (UNET_AMD_SKEL_FWD=0).  Same operations, so: the same skeleton,
    the same loss, and (through the saved E / skel planes the backward pass reads) the same gradient, bit for bit, on
    widths around the strip width, heights no strip length divides, tie-heavy inputs and several strip lengths."""
    g = torch.Generator().manual_seed(H * 1000 + W + 1)
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.55).float().to(DEV)
    probs = torch.rand(B, 1, H, W, generator=g)
    probs[0] = (probs[0] * 8).round() / 8
    if B > 1:
        probs[1, :, :, : W // 2] = 1.0
    probs = probs.clamp(1e-6, 1 - 1e-6).to(DEV).requires_grad_(True)
    crit = unet_amd.soft_dice_cldice()
    sk = unet_amd.SoftSkeletonize(num_iter=10, check_range=False)

    def everything():
        loss = crit(mask, probs)
        (grad,) = torch.autograd.grad(loss, probs)
        return loss.detach().clone(), grad, sk(probs.detach()), sk(mask)

    monkeypatch.setenv("UNET_AMD_SKEL_FWD", "0")
    want = everything()
    monkeypatch.delenv("UNET_AMD_SKEL_FWD")
    for R in (None, "1", "7", "16", "100"):
        if R is None:
            monkeypatch.delenv("UNET_AMD_SKEL_FWD_R", raising=False)
        else:
            monkeypatch.setenv("UNET_AMD_SKEL_FWD_R", R)
        got = everything()
        for a, b, what in zip(got, want, ("loss", "gradient", "skeleton of the prediction", "skeleton of the mask")):
            assert torch.equal(a, b), (R, what, (a - b).abs().max().item())
    assert want[2].abs().max().item() > 0
