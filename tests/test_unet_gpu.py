"""End-to-end parity of the HIP U-Net (unet_amd.UNet) with the reference.

Two anchors: (1) the golden fixtures captured from the reference import
(tests/golden/make_golden.py) and (2) the CPU oracle (oracle/ref_cpu.py, itself pinned to those
fixtures) on fresh seeded inputs.  The HIP path stores activations and activation gradients in
bf16 (8 significant bits) through 23 stacked convolutions, with fp32 accumulation, fp32
BatchNorm statistics and fp32 parameter gradients; the stated tolerances are for that."""
import copy
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

# Tolerances against the fp32 reference (16-bit storage, fp32 accumulate).  Measured on MI355X at
# 64x64 / 256x256 inputs: logits 1.4 % of max, loss 3e-5 relative, gradient-norm error < 1 %,
# gradient cosine 0.998 at the last decoder layer falling to 0.93 at the bridge (the gradient has
# then crossed 13 BatchNorm layers in bf16).  A CPU emulation of the same storage roundings
# (oracle/emu_bf16.py) sits at the same distance from the fp32 reference, which is what
# test_error_is_the_bf16_noise_floor checks.
LOGIT_REL = 3e-2        # max |logit error| relative to max |logit|
LOSS_REL = 5e-3         # BCE loss
GRAD_COS = 0.90         # cosine similarity per parameter tensor (tensors with a non-trivial gradient)
GRAD_COS_LAST = 0.99    # same, for conv_up4_*, bn_up4_* and final_conv (first layers of the backward pass)
GRAD_NORM = 0.08        # relative error of the gradient norm per tensor


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def compare_grads(model, ref_grads, min_norm_frac=1e-3, cos_min=None, cos_last=None, norm_tol=None):
    """ref_grads: name -> tensor.  Tensors whose reference gradient is pure rounding noise
    (conv biases under train-mode BatchNorm: analytically zero) are checked for smallness only."""
    cos_min = GRAD_COS if cos_min is None else cos_min
    cos_last = GRAD_COS_LAST if cos_last is None else cos_last
    norm_tol = GRAD_NORM if norm_tol is None else norm_tol
    total = max(g.norm().item() for g in ref_grads.values())
    worst_cos, worst_norm = 1.0, 0.0
    for n, p in model.named_parameters():
        if n not in ref_grads:
            continue
        ref = ref_grads[n]
        got = p.grad.detach().float().cpu()
        assert got.shape == ref.shape, n
        assert torch.isfinite(got).all(), n
        if n.endswith(".bias") and n.startswith("conv") and not n.startswith("conv_final"):
            # bias of a conv in front of BatchNorm: exactly zero in exact arithmetic
            assert got.abs().max().item() <= 1e-3 * total, n
            continue
        if ref.norm().item() < min_norm_frac * total:
            continue
        c = cosine(got, ref)
        r = abs(got.norm().item() / ref.norm().item() - 1.0)
        worst_cos, worst_norm = min(worst_cos, c), max(worst_norm, r)
        need = cos_last if (n.startswith(("conv_up4", "bn_up4", "final_conv"))) else cos_min
        assert c >= need, f"{n}: cosine {c:.4f} < {need}"
        assert r <= norm_tol, f"{n}: norm ratio error {r:.4f}"
    return worst_cos, worst_norm


def test_unet_golden_small(golden_dir):
    fx = np.load(os.path.join(golden_dir, "unet_s0_b2_32.npz"), allow_pickle=False)
    torch.manual_seed(0)
    model = unet_amd.create_unet(3, 1)
    cs = np.array([[p.detach().double().sum().item(), p.detach().double().abs().sum().item()]
                   for p in model.parameters()])
    np.testing.assert_allclose(cs, fx["param_checksums"], rtol=1e-12)   # same init as the reference (sums: order-free to 1e-12)
    model = model.to(DEV)
    x, m = torch.from_numpy(fx["x"]).to(DEV), torch.from_numpy(fx["mask"]).to(DEV)
    model.train()
    logits = model(x)
    assert logits.shape == (2, 1, 32, 32) and logits.dtype == torch.float32
    loss = nn.BCEWithLogitsLoss()(logits, m)
    loss.backward()
    want = torch.from_numpy(fx["train_logits"])
    err = (logits.detach().cpu() - want).abs().max().item()
    print(f"train logits: max err {err:.4g} (max |ref| {want.abs().max().item():.4g}); "
          f"loss {loss.item():.6f} vs {float(fx['train_loss']):.6f}")
    assert err <= LOGIT_REL * want.abs().max().item()
    assert abs(loss.item() - float(fx["train_loss"])) <= LOSS_REL * float(fx["train_loss"])
    ref_grads = {k[5:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("grad/")}
    # B=2 at 32x32 leaves 8 samples per channel for the bridge BatchNorm layers: ill conditioned, so the
    # absolute numbers are looser here than at the sizes of test_unet_train_step_matches_oracle
    wc, wn = compare_grads(model, ref_grads, cos_min=0.80, cos_last=0.98, norm_tol=0.15)
    print(f"grads: worst cosine {wc:.5f}, worst norm error {wn:.4f}")
    # gradient norms of every parameter against the reference summary
    names = list(fx["param_names"])
    gs = fx["grad_summary"]
    tot = gs[:, 0].max()
    for n, p in model.named_parameters():
        i = names.index(n)
        if gs[i, 0] > 1e-3 * tot:
            assert abs(p.grad.norm().item() / gs[i, 0] - 1) <= 0.15, n
    for bn in ("bn1_1", "bn5_2", "bn_up4_2"):
        mod = getattr(model, bn)
        np.testing.assert_allclose(mod.running_mean.cpu().numpy(), fx[f"{bn}.running_mean"], rtol=0, atol=2e-2)
        np.testing.assert_allclose(mod.running_var.cpu().numpy(), fx[f"{bn}.running_var"], rtol=5e-2, atol=1e-3)
        assert mod.num_batches_tracked.item() == 1
    feats = model.intermediate_features
    assert len(feats) == 15
    assert [list(t.shape) for t in feats] == fx["feature_shapes"].tolist()
    np.testing.assert_allclose([t.double().abs().sum().item() for t in feats], fx["feature_abs_sums"], rtol=2e-2)


def test_unet_eval_matches_oracle():
    """inference path (folded BatchNorm): fresh model with non-trivial running statistics."""
    torch.manual_seed(1)
    ref = ref_cpu.RefUNet(3, 1)
    x, _ = ref_cpu.synth_batch(2, 3, 64, 48)
    ref.train()
    with torch.no_grad():
        for _ in range(3):
            ref(x + 0.1 * torch.randn_like(x))          # move the running statistics off their defaults
    model = unet_amd.UNet(3, 1)
    model.load_state_dict(ref.state_dict())               # checkpoints interchange with the reference layout
    model = model.to(DEV).eval()
    ref.eval()
    with torch.no_grad():
        want = ref(x)
        got = model(x.to(DEV)).cpu()
    err = (got - want).abs().max().item()
    print(f"eval logits: max err {err:.4g} (max |ref| {want.abs().max().item():.4g})")
    assert err <= LOGIT_REL * want.abs().max().item()
    # batch-1 fp32 inference as in test_helper/utils.py:116-125
    with torch.no_grad():
        one = model(x[:1].to(DEV)).cpu()
    assert (one - want[:1]).abs().max().item() <= LOGIT_REL * want.abs().max().item()


def run_three(B, C, H, W, seed=2):
    """One BCE training step of: the fp32 oracle, the oracle with 16-bit storage roundings, the HIP module."""
    from oracle import emu_bf16
    torch.manual_seed(seed)
    ref = ref_cpu.RefUNet(C, 1)
    emu = emu_bf16.EmuUNet(C, 1)
    emu.load_state_dict(ref.state_dict())
    model = copy.deepcopy(unet_amd.UNet(C, 1))
    model.load_state_dict(ref.state_dict())
    model = model.to(DEV)
    x, m = ref_cpu.synth_batch(B, C, H, W)
    outs = {}
    for name, net, dev in (("ref", ref, "cpu"), ("emu", emu, "cpu"), ("hip", model, DEV)):
        net.train()
        logits = net(x.to(dev))
        loss = nn.BCEWithLogitsLoss()(logits, m.to(dev))
        loss.backward()
        outs[name] = dict(logits=logits.detach().cpu(), loss=loss.item(), net=net,
                          grads={n: p.grad.detach().cpu() for n, p in net.named_parameters()})
    return outs


@pytest.mark.parametrize("B,C,H,W", [(4, 3, 64, 64), (1, 2, 16, 48), (3, 3, 32, 80), (2, 3, 128, 128)])
def test_unet_train_step_matches_oracle(B, C, H, W):
    """HIP vs the fp32 oracle, with the oracle-with-16-bit-storage as the yardstick: per tensor, the HIP
    gradient may be at most twice as far from the fp32 gradient (1 - cosine, norm error) as the emulation is.
    Absolute thresholds are applied where BatchNorm is well conditioned (>= 32 samples per channel at the
    deepest level); a 16x48 single image normalises over 3 samples there and amplifies any rounding."""
    o = run_three(B, C, H, W)
    ref, emu, hip = o["ref"], o["emu"], o["hip"]
    scale = ref["logits"].abs().max().item()
    err_hip = (hip["logits"] - ref["logits"]).abs().max().item()
    err_emu = (emu["logits"] - ref["logits"]).abs().max().item()
    assert err_hip <= 2.0 * err_emu + 1e-3 * scale
    assert abs(hip["loss"] - ref["loss"]) <= LOSS_REL * ref["loss"]
    total = max(g.norm().item() for g in ref["grads"].values())
    worst = 0.0
    for n, g_ref in ref["grads"].items():
        g_hip, g_emu = hip["grads"][n], emu["grads"][n]
        assert torch.isfinite(g_hip).all(), n
        if n.endswith(".bias") and n.startswith("conv"):
            assert g_hip.abs().max().item() <= 1e-3 * total, n     # analytically zero under train-mode BN
            continue
        d_hip, d_emu = 1.0 - cosine(g_hip, g_ref), 1.0 - cosine(g_emu, g_ref)
        n_hip = abs(g_hip.norm().item() / g_ref.norm().item() - 1.0)
        n_emu = abs(g_emu.norm().item() / g_ref.norm().item() - 1.0)
        worst = max(worst, d_hip / (d_emu + 2e-3))
        assert d_hip <= 2.0 * d_emu + 2e-3, f"{n}: 1-cos hip {d_hip:.4g} vs emulation {d_emu:.4g}"
        assert n_hip <= 2.0 * n_emu + 5e-2, f"{n}: norm error hip {n_hip:.4g} vs emulation {n_emu:.4g}"
    print(f"[{B}x{C}x{H}x{W}] loss {hip['loss']:.6f} vs {ref['loss']:.6f}; logit err hip {err_hip:.4g} emu {err_emu:.4g} "
          f"(max {scale:.3g}); worst (1-cos hip)/(1-cos emu + 2e-3) = {worst:.3f}")
    if B * (H // 16) * (W // 16) >= 32:
        assert err_hip <= LOGIT_REL * scale
        wc, wn = compare_grads(hip["net"], ref["grads"])
        print(f"   absolute: worst cosine {wc:.5f}, worst norm err {wn:.4f}")
    bufs = dict(hip["net"].named_buffers())
    for n, buf in ref["net"].named_buffers():
        got = bufs[n].cpu()
        if n.endswith("num_batches_tracked"):
            assert got.item() == buf.item()
        elif n.endswith("running_mean"):
            assert (got - buf).abs().max().item() <= 2e-2, n
        else:
            assert ((got - buf).abs() <= 5e-2 * buf.abs() + 1e-3).all(), n


def test_unet_under_autocast_and_gradscaler():
    """The trainer's AMP contract (training_utils.py:117,139-155): fp16 autocast + GradScaler."""
    torch.manual_seed(3)
    model = unet_amd.UNet(3, 1).to(DEV)
    x, m = ref_cpu.synth_batch(2, 3, 32, 32)
    x, m = x.to(DEV), m.to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    scaler = torch.amp.GradScaler("cuda")
    losses = []
    for _ in range(3):
        opt.zero_grad()
        with torch.amp.autocast("cuda"):
            out = model(x)
            loss = nn.BCEWithLogitsLoss()(out, m)
        assert out.dtype == torch.float32
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert all(p.grad is not None and p.grad.dtype == torch.float32 for p in model.parameters())


def test_unet_adam_trajectory_golden(golden_dir):
    """config 1: B=4, 3x128x128, BCE, three Adam steps (lr 1e-3) against the reference's losses."""
    fx = np.load(os.path.join(golden_dir, "unet_s0_b4_128_adam.npz"), allow_pickle=False)
    torch.manual_seed(0)
    model = unet_amd.UNet(3, 1).to(DEV)
    g = torch.Generator().manual_seed(int(fx["seed_inputs"]))
    x = torch.rand(4, 3, 128, 128, generator=g).to(DEV)
    m = (torch.rand(4, 1, 128, 128, generator=g) > 0.5).float().to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for step in range(3):
        opt.zero_grad()
        logits = model(x)
        if step == 0:
            sl = logits[0, 0, :32, :32].detach().cpu().numpy()
            assert np.abs(sl - fx["logits0_slice"]).max() <= LOGIT_REL * np.abs(fx["logits0_slice"]).max()
        loss = nn.BCEWithLogitsLoss()(logits, m)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    print("adam losses", losses, "reference", fx["losses"].tolist())
    np.testing.assert_allclose(losses, fx["losses"], rtol=2e-2)


def test_module_surface():
    model = unet_amd.create_unet(3, 1).to(DEV)
    x = torch.rand(1, 3, 16, 16, device=DEV)
    model.eval()
    with torch.no_grad():
        y1 = model(x)
    clone = copy.deepcopy(model)                           # k_fold.py:139
    with torch.no_grad():
        y2 = clone(x)
    assert torch.equal(y1, y2)
    sd = model.state_dict()
    assert len(sd) == 136
    with pytest.raises(ValueError, match="divisible by 16"):
        model(torch.rand(1, 3, 20, 16, device=DEV))
    with pytest.raises(RuntimeError, match="HIP kernels only"):
        model(torch.rand(1, 3, 16, 16))
    # two forwards before one backward must not share buffers
    model.train()
    a = model(x)
    b = model(x * 0.5)
    (a.sum() + b.sum()).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_two_stream_step_equals_one_stream_step(monkeypatch):
    """The data-parallel trainer's step with the second stream (weight gradients, operand repacks, bucket-wise Adam
    beside backward) against the same step with everything on one stream and one Adam launch at the end: only the
    order of independent launches differs, so losses and parameters are bit-identical after several steps."""
    from unet_amd.dp import DataParallelTrainer
    x, m = ref_cpu.synth_batch(4, 3, 64, 64)
    x, m = x.to(DEV), m.to(DEV)
    results = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("UNET_AMD_WGRAD_STREAM", mode)
        torch.manual_seed(11)
        model = unet_amd.UNet(3, 1).to(DEV)
        tr = DataParallelTrainer(model, lr=1e-3, bucket_bytes=8 << 20)      # several buckets
        assert len(tr.buckets.buckets) >= 4
        losses = [tr.train_step(x, m).item() for _ in range(4)]
        torch.cuda.synchronize()
        results[mode] = (losses, tr.optimizer.p.clone(), tr.optimizer.m.clone(), tr.optimizer.v.clone(), tr.optimizer.steps)
    a, b = results["0"], results["1"]
    assert a[0] == b[0], (a[0], b[0])
    assert a[4] == b[4] == 4
    for u, v in zip(a[1:4], b[1:4]):
        assert torch.equal(u, v)
    assert a[0][-1] < a[0][0]                      # and it trains


def test_fused_batchnorm_backward_sums_give_the_same_step(monkeypatch):
    """UNET_AMD_FUSE_BN_BWD=1 (BatchNorm-backward sums taken in the input-gradient epilogue, from the fp32
    accumulators instead of the bf16-rounded gradient): same loss, gradients equal to rounding."""
    from unet_amd.dp import DataParallelTrainer
    x, m = ref_cpu.synth_batch(4, 3, 64, 64)
    x, m = x.to(DEV), m.to(DEV)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("UNET_AMD_FUSE_BN_BWD", mode)
        torch.manual_seed(5)
        model = unet_amd.UNet(3, 1).to(DEV)
        tr = DataParallelTrainer(model, lr=1e-3)
        loss = tr.train_step(x, m, step_optimizer=False).item()
        torch.cuda.synchronize()
        got[mode] = (loss, {n: tr.buckets.views[n].clone() for n, _ in model.named_parameters()})
    assert got["0"][0] == got["1"][0]
    for n, g0 in got["0"][1].items():
        g1 = got["1"][1][n]
        if g0.abs().max().item() == 0:
            assert g1.abs().max().item() == 0, n
            continue
        cos = torch.nn.functional.cosine_similarity(g0.flatten().double(), g1.flatten().double(), dim=0).item()
        assert cos > 0.999, (n, cos)            # bias gradients are sums with heavy cancellation: the most sensitive
        assert abs(g1.norm().item() / g0.norm().item() - 1) < 2e-2, n
