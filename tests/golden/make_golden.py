#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE.

Run only in the build container, where the reference checkout is mounted at
/root/reference:

    python tests/golden/make_golden.py

The reference never travels to the GPU box; only the small .npz/.json files
written here do.  Each fixture stores inputs and the reference's outputs
(logits, losses, gradients or gradient summaries).  Weights are not stored:
they are regenerated from the seed, because ``oracle/ref_cpu.py`` constructs
its layers in the reference's order; a checksum of every parameter is stored so
the replay can prove that.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from models.unet import UNet                      # noqa: E402  (reference)
from models.lunext import LUNeXt                  # noqa: E402  (reference)
from losses.cl_dice import soft_dice_cldice, SoftSkeletonize, soft_dice   # noqa: E402  (reference)

torch.set_num_threads(8)
torch.use_deterministic_algorithms(False)


def rand_inputs(b, c, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(b, c, h, w, generator=g)
    m = (torch.rand(b, 1, h, w, generator=g) > 0.5).float()
    return x, m


def param_checksums(model):
    return np.array([[p.detach().double().sum().item(), p.detach().double().abs().sum().item()]
                     for p in model.parameters()], dtype=np.float64)


def grad_summary(model):
    out = []
    for p in model.parameters():
        if p.grad is None:
            out.append([np.nan, np.nan])
        else:
            out.append([p.grad.double().norm().item(), p.grad.double().sum().item()])
    return np.array(out, dtype=np.float64)


def keys_json(model, path):
    sd = model.state_dict()
    with open(path, "w") as f:
        json.dump([[k, list(v.shape), str(v.dtype)] for k, v in sd.items()], f, indent=0)


def unet_small():
    torch.manual_seed(0)
    model = UNet(3, 1)
    keys_json(model, os.path.join(HERE, "unet_keys.json"))
    x, m = rand_inputs(2, 3, 32, 32, seed=11)
    fx = {"x": x.numpy(), "mask": m.numpy(), "param_checksums": param_checksums(model)}
    model.train()
    logits = model(x)
    loss = nn.BCEWithLogitsLoss()(logits, m)
    loss.backward()
    fx["train_logits"] = logits.detach().numpy()
    fx["train_loss"] = np.float64(loss.item())
    fx["grad_summary"] = grad_summary(model)
    names = [n for n, _ in model.named_parameters()]
    fx["param_names"] = np.array(names)
    for n, p in model.named_parameters():
        if p.numel() <= 2048 or n in ("conv1_1.weight",):
            fx["grad/" + n] = p.grad.numpy().copy()
    # a few large gradients, for element-wise comparison of the GEMM paths
    for n in ("conv1_2.weight", "conv_up4_1.weight", "upconv4.weight", "conv5_2.bias", "upconv1.bias"):
        fx["grad/" + n] = dict(model.named_parameters())[n].grad.numpy().copy()
    for bn in ("bn1_1", "bn5_2", "bn_up4_2"):
        mod = getattr(model, bn)
        fx[f"{bn}.running_mean"] = mod.running_mean.numpy().copy()
        fx[f"{bn}.running_var"] = mod.running_var.numpy().copy()
        fx[f"{bn}.num_batches_tracked"] = np.int64(mod.num_batches_tracked.item())
    fx["feature_shapes"] = np.array([list(t.shape) for t in model.intermediate_features])
    fx["feature_sums"] = np.array([t.detach().double().sum().item() for t in model.intermediate_features])
    fx["feature_abs_sums"] = np.array([t.detach().double().abs().sum().item() for t in model.intermediate_features])
    model.eval()
    with torch.no_grad():
        fx["eval_logits"] = model(x).numpy()
    np.savez_compressed(os.path.join(HERE, "unet_s0_b2_32.npz"), **fx)
    print("unet_s0_b2_32: train loss", fx["train_loss"])


def unet_in2():
    """in_channels=2 (the my_proj2 dataset shape, training_utils.py:87-94) at 16x48."""
    torch.manual_seed(3)
    model = UNet(2, 1)
    x, m = rand_inputs(1, 2, 16, 48, seed=5)
    model.train()
    logits = model(x)
    loss = nn.BCEWithLogitsLoss()(logits, m)
    loss.backward()
    fx = {"x": x.numpy(), "mask": m.numpy(), "train_logits": logits.detach().numpy(),
          "train_loss": np.float64(loss.item()), "grad_summary": grad_summary(model),
          "param_checksums": param_checksums(model)}
    np.savez_compressed(os.path.join(HERE, "unet_s3_in2_b1_16x48.npz"), **fx)
    print("unet_in2: loss", fx["train_loss"])


def unet_adam_128():
    """config 1: B=4, 3x128x128, BCE, three Adam steps (lr 1e-3)."""
    torch.manual_seed(0)
    model = UNet(3, 1)
    x, m = rand_inputs(4, 3, 128, 128, seed=7)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = nn.BCEWithLogitsLoss()
    losses, dices = [], []
    model.train()
    first_logits = None
    for step in range(3):
        opt.zero_grad()
        logits = model(x)
        if first_logits is None:
            first_logits = logits.detach().clone()
        loss = crit(logits, m)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        with torch.no_grad():
            dices.append(soft_dice(m, torch.sigmoid(logits)).item())
    fx = {"seed_inputs": np.int64(7), "losses": np.array(losses), "soft_dice": np.array(dices),
          "logits0_sum": np.float64(first_logits.double().sum().item()),
          "logits0_abs_sum": np.float64(first_logits.double().abs().sum().item()),
          "logits0_slice": first_logits[0, 0, :32, :32].numpy().copy()}
    np.savez_compressed(os.path.join(HERE, "unet_s0_b4_128_adam.npz"), **fx)
    print("unet_adam_128 losses", losses)


def lunext_small():
    torch.manual_seed(0)
    model = LUNeXt(3, 1)
    keys_json(model, os.path.join(HERE, "lunext_keys.json"))
    x, m = rand_inputs(2, 3, 32, 32, seed=13)
    crit = soft_dice_cldice()
    model.train()
    logits = model(x)
    probs = torch.sigmoid(logits).clamp(1e-6, 1 - 1e-6)
    loss = crit(m, probs)
    loss.backward()
    fx = {"x": x.numpy(), "mask": m.numpy(), "param_checksums": param_checksums(model),
          "train_logits": logits.detach().numpy(), "train_loss": np.float64(loss.item()),
          "grad_summary": grad_summary(model),
          "param_names": np.array([n for n, _ in model.named_parameters()])}
    for n, p in model.named_parameters():
        if p.grad is not None and (p.numel() <= 4096 or n in ("downconv2.weight", "upconv3.weight")):
            fx["grad/" + n] = p.grad.numpy().copy()
    bce = nn.BCEWithLogitsLoss()(logits.detach(), m)
    fx["bce_loss"] = np.float64(bce.item())
    model.eval()
    with torch.no_grad():
        fx["eval_logits"] = model(x).numpy()
    np.savez_compressed(os.path.join(HERE, "lunext_s0_b2_32.npz"), **fx)
    print("lunext_s0_b2_32: loss", fx["train_loss"])


def lunext_odd():
    """A size that is not divisible by 16: 44x52 -> 22x26 -> 11x13 -> 5x6 -> 2x3, so the transposed convolutions of the
    two deepest decoder stages return 4x6 and 10x12 where the skip connections are 5x6 and 11x13 (lunext.py:226-235)."""
    from oracle import ref_cpu
    torch.manual_seed(1)
    model = LUNeXt(3, 1)
    x, m = rand_inputs(2, 3, 44, 52, seed=17)
    import copy
    oracle = ref_cpu.RefLUNeXt(3, 1)
    oracle.load_state_dict(model.state_dict())
    assert torch.equal(copy.deepcopy(model)(x), oracle(x)), "oracle != reference on the size-fix path"
    crit = soft_dice_cldice()
    model.train()
    logits = model(x)
    loss = crit(m, torch.sigmoid(logits).clamp(1e-6, 1 - 1e-6))
    loss.backward()
    fx = {"x": x.numpy(), "mask": m.numpy(), "param_checksums": param_checksums(model),
          "train_logits": logits.detach().numpy(), "train_loss": np.float64(loss.item()),
          "grad_summary": grad_summary(model),
          "param_names": np.array([n for n, _ in model.named_parameters()])}
    for n, p in model.named_parameters():
        if p.grad is not None and (p.numel() <= 4096 or n == "upconv3.weight"):
            fx["grad/" + n] = p.grad.numpy().copy()
    model.eval()
    with torch.no_grad():
        fx["eval_logits"] = model(x).numpy()
    np.savez_compressed(os.path.join(HERE, "lunext_s1_b2_44x52.npz"), **fx)
    print("lunext_s1_b2_44x52: loss", fx["train_loss"])


def cldice_cases():
    sk = SoftSkeletonize(num_iter=10)
    crit = soft_dice_cldice()
    g = torch.Generator().manual_seed(21)
    cases = {}

    def add(name, probs, mask):
        p = probs.clone().requires_grad_(True)
        loss = crit(mask, p)
        loss.backward()
        with torch.no_grad():
            pc = p.detach().clamp(0, 1)
            cases[name] = {
                "probs": probs.numpy(), "mask": mask.numpy(),
                "loss": np.float64(loss.item()), "grad": p.grad.numpy().copy(),
                "erode": sk.soft_erode(pc).numpy(), "dilate": sk.soft_dilate(pc).numpy(),
                "skel_pred": sk.soft_skel(pc).numpy(), "skel_true": sk.soft_skel(mask).numpy(),
                "dice": np.float64(soft_dice(mask, pc).item()),
            }
        print("cldice", name, "loss", loss.item())

    for hw in (16, 64):
        probs = torch.rand(2, 1, hw, hw, generator=g).clamp(1e-6, 1 - 1e-6)
        mask = (torch.rand(2, 1, hw, hw, generator=g) > 0.5).float()
        add(f"rand{hw}", probs, mask)
    # non-square, batch 3
    probs = torch.rand(3, 1, 24, 40, generator=g).clamp(1e-6, 1 - 1e-6)
    mask = (torch.rand(3, 1, 24, 40, generator=g) > 0.6).float()
    add("rand24x40", probs, mask)
    # tie-heavy: constant image
    add("const", torch.full((2, 1, 16, 16), 0.5), (torch.rand(2, 1, 16, 16, generator=g) > 0.5).float())
    # tie-heavy: probabilities quantised to 1/8
    q = (torch.rand(2, 1, 32, 32, generator=g) * 8).floor() / 8
    add("quant8", q.clamp(1e-6, 1 - 1e-6), (torch.rand(2, 1, 32, 32, generator=g) > 0.5).float())
    # saturated plateaus as produced by sigmoid().clamp(1e-6, 1-1e-6)
    logits = torch.randn(2, 1, 32, 32, generator=g) * 20
    add("saturated", torch.sigmoid(logits).clamp(1e-6, 1 - 1e-6), (logits > 0).float())
    # blob-shaped mask with smooth probabilities (vessel-like structure)
    yy, xx = torch.meshgrid(torch.arange(48.), torch.arange(48.), indexing="ij")
    blob = ((yy - 24).abs() < 3) | ((xx - 20).abs() < 2)
    mask = blob.float()[None, None].repeat(2, 1, 1, 1)
    probs = (0.8 * mask + 0.1 + 0.1 * torch.rand(2, 1, 48, 48, generator=g)).clamp(1e-6, 1 - 1e-6)
    add("blob48", probs, mask)
    # values outside [0,1]: the loss clamps to [0,1] first (cl_dice.py:116-117)
    add("unclamped", torch.randn(1, 1, 16, 16, generator=g) * 0.7 + 0.5,
        (torch.rand(1, 1, 16, 16, generator=g) > 0.5).float())
    flat = {}
    for name, d in cases.items():
        for k, v in d.items():
            flat[f"{name}/{k}"] = v
    flat["case_names"] = np.array(list(cases.keys()))
    np.savez_compressed(os.path.join(HERE, "cldice_cases.npz"), **flat)


def tie_rules():
    """The autograd tie-breaking rules the clDice backward kernels must reproduce."""
    import torch.nn.functional as F
    x = torch.ones(1, 1, 3, 3, requires_grad=True)
    F.max_pool2d(x, 3, 1, 1).sum().backward()
    a = torch.tensor([1.0, 2.0, 3.0], requires_grad=True)
    b = torch.tensor([1.0, 3.0, 2.0], requires_grad=True)
    torch.min(a, b).sum().backward()
    r = torch.tensor([-1.0, 0.0, 1.0], requires_grad=True)
    F.relu(r).sum().backward()
    c = torch.tensor([-0.5, 0.0, 0.5, 1.0, 1.5], requires_grad=True)
    c.clamp(0.0, 1.0).sum().backward()
    np.savez(os.path.join(HERE, "tie_rules.npz"), maxpool_ones=x.grad.numpy(), min_a=a.grad.numpy(),
             min_b=b.grad.numpy(), relu=r.grad.numpy(), clamp=c.grad.numpy())
    print("tie rules", x.grad.flatten().tolist(), a.grad.tolist(), r.grad.tolist(), c.grad.tolist())


def check_oracle_matches_reference():
    """Bit-level check of oracle/ref_cpu.py against the reference import (container only)."""
    from oracle import ref_cpu
    torch.manual_seed(0); a = UNet(3, 1)
    torch.manual_seed(0); b = ref_cpu.RefUNet(3, 1)
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(va, vb), ka
    x, m = rand_inputs(2, 3, 32, 32, seed=11)
    ya, yb = a(x), b(x)
    assert torch.equal(ya, yb), (ya - yb).abs().max()
    torch.manual_seed(0); a = LUNeXt(3, 1)
    torch.manual_seed(0); b = ref_cpu.RefLUNeXt(3, 1)
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    ya, yb = a(x), b(x)
    assert torch.equal(ya, yb)
    p = torch.sigmoid(ya).clamp(1e-6, 1 - 1e-6)
    la = soft_dice_cldice()(m, p); lb = ref_cpu.RefSoftDiceClDice()(m, p)
    assert torch.equal(la, lb), (la, lb)
    print("oracle == reference (bit-exact) on UNet, LUNeXt, soft_dice_cldice")


if __name__ == "__main__":
    check_oracle_matches_reference()
    tie_rules()
    cldice_cases()
    unet_small()
    unet_in2()
    lunext_small()
    lunext_odd()
    unet_adam_128()
