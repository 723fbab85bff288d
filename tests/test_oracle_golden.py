"""The CPU oracle (oracle/ref_cpu.py) replayed against the fixtures that were
captured from the reference import (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import torch
import torch.nn as nn

from oracle import ref_cpu

torch.set_num_threads(8)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_state_dict_keys_match_reference(golden_dir):
    for cls, fn in ((ref_cpu.RefUNet, "unet_keys.json"), (ref_cpu.RefLUNeXt, "lunext_keys.json")):
        with open(os.path.join(golden_dir, fn)) as f:
            want = json.load(f)
        sd = cls(3, 1).state_dict()
        got = [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()]
        assert got == want
    assert len(ref_cpu.RefUNet().state_dict()) == 136
    assert len(ref_cpu.RefLUNeXt().state_dict()) == 176


def test_unet_small_matches_reference(golden_dir):
    fx = _load(golden_dir, "unet_s0_b2_32.npz")
    torch.manual_seed(0)
    model = ref_cpu.RefUNet(3, 1)
    cs = np.array([[p.detach().double().sum().item(), p.detach().double().abs().sum().item()]
                   for p in model.parameters()])
    np.testing.assert_allclose(cs, fx["param_checksums"], rtol=1e-12)   # same seed -> same init
    x, m = torch.from_numpy(fx["x"]), torch.from_numpy(fx["mask"])
    model.train()
    logits = model(x)
    loss = nn.BCEWithLogitsLoss()(logits, m)
    loss.backward()
    np.testing.assert_array_equal(logits.detach().numpy(), fx["train_logits"])
    assert loss.item() == float(fx["train_loss"])
    names = [n for n, _ in model.named_parameters()]
    assert names == list(fx["param_names"])
    gs = fx["grad_summary"]
    for i, (n, p) in enumerate(model.named_parameters()):
        np.testing.assert_allclose(p.grad.double().norm().item(), gs[i, 0], rtol=1e-5, atol=1e-9, err_msg=n)
        key = "grad/" + n
        if key in fx.files:
            np.testing.assert_allclose(p.grad.numpy(), fx[key], rtol=1e-4, atol=1e-7, err_msg=n)
    for bn in ("bn1_1", "bn5_2", "bn_up4_2"):
        mod = getattr(model, bn)
        np.testing.assert_allclose(mod.running_mean.numpy(), fx[f"{bn}.running_mean"], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(mod.running_var.numpy(), fx[f"{bn}.running_var"], rtol=1e-6, atol=1e-8)
        assert mod.num_batches_tracked.item() == int(fx[f"{bn}.num_batches_tracked"])
    assert len(model.intermediate_features) == 15
    assert [list(t.shape) for t in model.intermediate_features] == fx["feature_shapes"].tolist()
    np.testing.assert_allclose([t.double().sum().item() for t in model.intermediate_features],
                               fx["feature_sums"], rtol=1e-6)
    model.eval()
    with torch.no_grad():
        np.testing.assert_allclose(model(x).numpy(), fx["eval_logits"], rtol=1e-6, atol=1e-7)


def test_unet_two_channel_input(golden_dir):
    fx = _load(golden_dir, "unet_s3_in2_b1_16x48.npz")
    torch.manual_seed(3)
    model = ref_cpu.RefUNet(2, 1)
    x, m = torch.from_numpy(fx["x"]), torch.from_numpy(fx["mask"])
    logits = model(x)
    loss = nn.BCEWithLogitsLoss()(logits, m)
    np.testing.assert_array_equal(logits.detach().numpy(), fx["train_logits"])
    assert loss.item() == float(fx["train_loss"])


def test_unet_adam_trajectory(golden_dir):
    fx = _load(golden_dir, "unet_s0_b4_128_adam.npz")
    torch.manual_seed(0)
    model = ref_cpu.RefUNet(3, 1)
    g = torch.Generator().manual_seed(int(fx["seed_inputs"]))
    x = torch.rand(4, 3, 128, 128, generator=g)
    m = (torch.rand(4, 1, 128, 128, generator=g) > 0.5).float()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for step in range(3):
        opt.zero_grad()
        logits = model(x)
        if step == 0:
            np.testing.assert_allclose(logits[0, 0, :32, :32].detach().numpy(), fx["logits0_slice"],
                                       rtol=1e-5, atol=1e-6)
        loss = nn.BCEWithLogitsLoss()(logits, m)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, fx["losses"], rtol=2e-4)


def test_lunext_small_matches_reference(golden_dir):
    fx = _load(golden_dir, "lunext_s0_b2_32.npz")
    torch.manual_seed(0)
    model = ref_cpu.RefLUNeXt(3, 1)
    x, m = torch.from_numpy(fx["x"]), torch.from_numpy(fx["mask"])
    model.train()
    logits = model(x)
    loss = ref_cpu.RefSoftDiceClDice()(m, ref_cpu.probs_from_logits(logits))
    loss.backward()
    np.testing.assert_array_equal(logits.detach().numpy(), fx["train_logits"])
    assert loss.item() == float(fx["train_loss"])
    gs = fx["grad_summary"]
    n_none = 0
    for i, (n, p) in enumerate(model.named_parameters()):
        if np.isnan(gs[i, 0]):
            assert p.grad is None, n
            assert ".output_conv." in n       # declared, never called (lunext.py:74,120)
            n_none += 1
        else:
            np.testing.assert_allclose(p.grad.double().norm().item(), gs[i, 0], rtol=1e-5, atol=1e-10, err_msg=n)
    assert n_none == 8
    model.eval()
    with torch.no_grad():
        np.testing.assert_allclose(model(x).numpy(), fx["eval_logits"], rtol=1e-6, atol=1e-7)


def test_lunext_size_not_divisible_by_16_matches_reference(golden_dir):
    """44x52: the two deepest decoder stages go through the bilinear size fix (lunext.py:226-235)"""
    fx = _load(golden_dir, "lunext_s1_b2_44x52.npz")
    torch.manual_seed(1)
    model = ref_cpu.RefLUNeXt(3, 1)
    x, m = torch.from_numpy(fx["x"]), torch.from_numpy(fx["mask"])
    model.train()
    logits = model(x)
    loss = ref_cpu.RefSoftDiceClDice()(m, ref_cpu.probs_from_logits(logits))
    loss.backward()
    np.testing.assert_array_equal(logits.detach().numpy(), fx["train_logits"])
    assert loss.item() == float(fx["train_loss"])
    gs = fx["grad_summary"]
    for i, (n, p) in enumerate(model.named_parameters()):
        if not np.isnan(gs[i, 0]):
            np.testing.assert_allclose(p.grad.double().norm().item(), gs[i, 0], rtol=1e-5, atol=1e-10, err_msg=n)
    model.eval()
    with torch.no_grad():
        np.testing.assert_allclose(model(x).numpy(), fx["eval_logits"], rtol=1e-6, atol=1e-7)


def test_cldice_cases_match_reference(golden_dir):
    fx = _load(golden_dir, "cldice_cases.npz")
    crit = ref_cpu.RefSoftDiceClDice()
    for name in fx["case_names"]:
        probs = torch.from_numpy(fx[f"{name}/probs"]).requires_grad_(True)
        mask = torch.from_numpy(fx[f"{name}/mask"])
        loss = crit(mask, probs)
        loss.backward()
        assert loss.item() == float(fx[f"{name}/loss"]), name
        np.testing.assert_array_equal(probs.grad.numpy(), fx[f"{name}/grad"], err_msg=name)
        pc = probs.detach().clamp(0, 1)
        np.testing.assert_array_equal(ref_cpu.soft_erode(pc).numpy(), fx[f"{name}/erode"])
        np.testing.assert_array_equal(ref_cpu.soft_dilate(pc).numpy(), fx[f"{name}/dilate"])
        np.testing.assert_array_equal(ref_cpu.soft_skel(pc).numpy(), fx[f"{name}/skel_pred"])
        np.testing.assert_array_equal(ref_cpu.soft_skel(mask).numpy(), fx[f"{name}/skel_true"])


def test_autograd_tie_rules(golden_dir):
    fx = _load(golden_dir, "tie_rules.npz")
    # max-pool backward: whole gradient to the FIRST maximal element in row-major window order
    np.testing.assert_array_equal(fx["maxpool_ones"].reshape(3, 3), [[4, 2, 0], [2, 1, 0], [0, 0, 0]])
    np.testing.assert_array_equal(fx["min_a"], [0.5, 1.0, 0.0])     # min: ties split 1/2, 1/2
    np.testing.assert_array_equal(fx["relu"], [0.0, 0.0, 1.0])      # relu'(0) = 0
    np.testing.assert_array_equal(fx["clamp"], [0.0, 1.0, 1.0, 1.0, 0.0])  # clamp passes grad at the bounds


def test_metrics_oracle_matches_reference(golden_dir):
    """oracle/metrics.py against the outputs of the reference's evaluation helpers (metrics_cases.npz)."""
    from oracle import metrics as om
    fx = _load(golden_dir, "metrics_cases.npz")
    fold_keys, sample_keys = list(fx["fold_keys"]), list(fx["sample_keys"])
    cases = sorted(k[:-len(".logits")] for k in fx.files if k.endswith(".logits"))
    assert len(cases) == 6
    for c in cases:
        logits, mask = fx[c + ".logits"], fx[c + ".mask"]
        pred = om.binarise(logits)
        np.testing.assert_array_equal(pred, fx[c + ".pred"], err_msg=c)
        y_true = np.transpose(mask, (0, 2, 3, 1)).squeeze()            # masks arrive as [N,H,W,1] arrays
        got = om.fold_metrics(y_true, pred.squeeze())
        np.testing.assert_array_equal([got[k] for k in fold_keys], fx[c + ".fold"], err_msg=c)
        for j in range(len(logits)):
            sm = om.sample_metrics(pred[j], mask[j])
            np.testing.assert_array_equal([sm[k] for k in sample_keys], fx[c + ".sample"][j], err_msg=f"{c}[{j}]")


def selfreg_case(fx, name):
    """(features, logits, targets, seeds, [lsrc, lifd, dice, ce, loss], dlogits) of one stored case."""
    n = int(fx[f"{name}.n"])
    feats = [torch.from_numpy(fx[f"{name}.f{i}"].astype(np.float32)) for i in range(n)]
    return (feats, torch.from_numpy(fx[f"{name}.logits"]), torch.from_numpy(fx[f"{name}.targets"]),
            [int(s) for s in fx[f"{name}.seeds"]], fx[f"{name}.scalars"], fx[f"{name}.dlogits"])


def test_self_reg_oracle_matches_reference(golden_dir):
    """oracle/self_reg.py against the reference's SelfRegLoss outputs (selfreg_cases.npz), the seeds of the
    channel subsets taken from the fixture."""
    from oracle import self_reg as osr
    fx = _load(golden_dir, "selfreg_cases.npz")
    names = list(fx["names"])
    assert names == ["unet_like", "dead_channels", "final_wider", "single_feature", "two_channels"]
    for name in names:
        feats, logits, targets, seeds, want, dwant = selfreg_case(fx, name)
        logits = logits.clone().requires_grad_(True)
        np.testing.assert_allclose(float(osr.lsrc(feats, seeds)), want[0], rtol=1e-6, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(float(osr.lifd(feats)), want[1], rtol=1e-6, atol=1e-9, err_msg=name)
        np.testing.assert_allclose(float(osr.dice_loss(logits.detach(), targets)), want[2], rtol=1e-6, err_msg=name)
        assert float(osr.cross_entropy(logits.detach(), targets)) == 0.0 == want[3]    # one class: identically zero
        loss = osr.self_reg_loss(logits, targets, feats, seeds)
        np.testing.assert_allclose(float(loss.detach()), want[4], rtol=1e-6, err_msg=name)
        loss.backward()
        np.testing.assert_allclose(logits.grad.numpy(), dwant, rtol=1e-5, atol=1e-9, err_msg=name)
