"""Evaluation and training drivers on the device (unet_amd/trainer.py) against the oracle.

  * `uh_confusion_counts` through the C ABI: integer counts, bit-exact against oracle/metrics.py on the cases
    captured from the reference (logit 0, saturated logits, empty prediction, non-binary mask values), in
    whole-set and per-sample mode, with accumulation over calls, odd sizes and unaligned pointers, and at
    32 x 256^2 / 1 x 1024^2 against a torch expression on the device;
  * `evaluate_model` / `save_results`: the HIP model's logits taken to the host and pushed through the oracle's
    restatement of the reference's metric code give exactly the numbers the device path reports;
  * `train_model`: a short run on a learnable toy problem lowers the validation loss, keeps the best checkpoint
    (loadable into the fp32 oracle model: same keys), returns the model in eval mode.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import gpu_present            # noqa: E402

if not gpu_present():
    pytest.skip("needs a GPU", allow_module_level=True)

import unet_amd                                        # noqa: E402
from unet_amd import trainer                           # noqa: E402
from unet_amd._lib import call                         # noqa: E402
from oracle import metrics as om                       # noqa: E402
from oracle import ref_cpu                             # noqa: E402

DEV = torch.device("cuda", 0)


def S():
    return torch.cuda.current_stream().cuda_stream


def _cases(golden_dir):
    fx = np.load(os.path.join(golden_dir, "metrics_cases.npz"), allow_pickle=False)
    return fx, sorted(k[:-len(".logits")] for k in fx.files if k.endswith(".logits"))


def test_confusion_counts_match_oracle_on_reference_cases(golden_dir):
    fx, cases = _cases(golden_dir)
    for c in cases:
        logits = torch.from_numpy(fx[c + ".logits"]).to(DEV)
        mask = torch.from_numpy(fx[c + ".mask"]).to(DEV)
        want = om.confusion_counts(fx[c + ".mask"], fx[c + ".pred"])           # the reference's own predictions
        counts = torch.zeros(4, dtype=torch.int64, device=DEV)
        trainer.confusion_counts(logits, mask, counts)
        assert tuple(counts.tolist()) == want, c
        trainer.confusion_counts(logits, mask, counts)                          # accumulates
        assert tuple(counts.tolist()) == tuple(2 * v for v in want), c
        per = torch.zeros(len(logits), 4, dtype=torch.int64, device=DEV)
        trainer.confusion_counts(logits, mask, per, per_sample=True)
        for j in range(len(logits)):
            assert tuple(per[j].tolist()) == om.confusion_counts(fx[c + ".mask"][j], fx[c + ".pred"][j]), (c, j)
        got = trainer.metrics_from_counts(*want)
        np.testing.assert_array_equal([got[k] for k in fx["fold_keys"]], fx[c + ".fold"], err_msg=c)


@pytest.mark.parametrize("n,offset", [(1, 0), (3, 0), (4, 1), (1001, 0), (4099, 3), (65536, 2)])
def test_confusion_counts_odd_sizes_and_alignment(n, offset):
    g = torch.Generator().manual_seed(n)
    planes = 3
    logits = torch.randn(planes * n + offset, generator=g) * 2
    mask = (torch.rand(planes * n + offset, generator=g) > 0.5).float()
    logits[::7] = 0.0
    ld, md = logits.to(DEV), mask.to(DEV)
    counts = torch.zeros(planes, 4, dtype=torch.int64, device=DEV)
    call("uh_confusion_counts", planes, n, ld[offset:].data_ptr(), md[offset:].data_ptr(), counts.data_ptr(), S())
    for p in range(planes):
        sl = slice(offset + p * n, offset + (p + 1) * n)
        want = om.confusion_counts(mask[sl].numpy(), om.binarise(logits[sl].numpy()))
        assert tuple(counts[p].tolist()) == want, p


@pytest.mark.parametrize("shape", [(32, 1, 256, 256), (1, 1, 1024, 1024)])
def test_confusion_counts_full_size(shape):
    g = torch.Generator(device=DEV).manual_seed(5)
    logits = torch.randn(shape, device=DEV, generator=g) * 4
    mask = (torch.rand(shape, device=DEV, generator=g) > 0.7).float()
    counts = torch.zeros(4, dtype=torch.int64, device=DEV)
    trainer.confusion_counts(logits, mask, counts)
    pred = torch.sigmoid(logits) > 0.5
    t = mask == 1
    want = [(pred & t).sum().item(), (~pred & ~t).sum().item(), (pred & ~t).sum().item(), (~pred & t).sum().item()]
    assert counts.tolist() == want
    assert sum(want) == logits.numel()
    per = torch.zeros(shape[0], 4, dtype=torch.int64, device=DEV)
    trainer.confusion_counts(logits, mask, per, per_sample=True)
    assert per.sum(0).tolist() == want


def test_confusion_counts_rejects_bad_arguments():
    x = torch.zeros(8, device=DEV)
    with pytest.raises(ValueError):
        trainer.confusion_counts(x, torch.zeros(4, device=DEV), torch.zeros(4, dtype=torch.int64, device=DEV))
    with pytest.raises(ValueError):
        trainer.confusion_counts(x, x, torch.zeros(4, dtype=torch.int32, device=DEV))
    with pytest.raises(unet_amd.HipKernelError):
        call("uh_confusion_counts", 0, 8, x.data_ptr(), x.data_ptr(), x.data_ptr(), S())


def _toy_data(n, hw, seed, channels=3):
    """Images whose mask is a threshold of a blurred channel: learnable in a few dozen steps."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(n, channels, hw, hw, generator=g)
    blur = torch.nn.functional.avg_pool2d(img[:, :1], 5, 1, 2)
    mask = (blur > 0.5).float()
    return img.permute(0, 2, 3, 1).numpy().copy(), mask.permute(0, 2, 3, 1).numpy().copy()


@pytest.mark.parametrize("kind", ["UNet", "LUNeXt"])
def test_evaluate_model_matches_oracle_metrics(kind):
    torch.manual_seed(4)
    model = (unet_amd.UNet(3, 1) if kind == "UNet" else unet_amd.LUNeXt(3, 1)).to(DEV)
    images, masks = _toy_data(11, 32, seed=1)
    x = torch.from_numpy(images).permute(0, 3, 1, 2).contiguous().to(DEV)
    model.train()
    with torch.no_grad():
        model(x[:8])                                   # running statistics off their initial values
        model.eval()
        model.final_conv.bias -= model(x).median()     # an untrained head: centre the logits so both classes occur
    got = trainer.evaluate_model(model, (images, masks), DEV, batch_size=4)
    model.eval()
    with torch.no_grad():
        logits = torch.cat([model(x[i:i + 4]) for i in range(0, 11, 4)]).cpu().numpy()   # same batches
    pred = om.binarise(logits)
    want = om.fold_metrics(masks.squeeze(), pred.squeeze())
    assert got == want
    assert 0 < pred.mean() < 1                         # both classes occur: the comparison is not vacuous


def test_save_results_matches_oracle_metrics(tmp_path):
    torch.manual_seed(6)
    model = unet_amd.UNet(3, 1).to(DEV).eval()
    images, masks = _toy_data(5, 32, seed=2)
    x = torch.from_numpy(images).permute(0, 3, 1, 2).contiguous().to(DEV)
    with torch.no_grad():
        model.final_conv.bias -= model(x).median()
    m = torch.from_numpy(masks).permute(0, 3, 1, 2).contiguous()
    path = tmp_path / "report.txt"
    avg = trainer.save_results(model, (images, masks), 12.5, str(path), log=lambda *_: None)
    with torch.no_grad():
        logits = torch.cat([model(x[i:i + 1]) for i in range(5)]).cpu()
    pred = om.binarise(logits.numpy())
    per = [om.sample_metrics(pred[j], m[j]) for j in range(5)]
    assert 0 < pred.mean() < 1
    for k in trainer.SAMPLE_KEYS:
        assert avg[k] == sum(p[k] for p in per) / 5, k
    losses = [torch.nn.functional.binary_cross_entropy_with_logits(logits[j:j + 1], m[j:j + 1]).item() for j in range(5)]
    assert abs(avg["loss"] - sum(losses) / 5) <= 1e-6
    text = path.read_text()
    assert "test samples: 5" in text and f"Dice Coefficient (DC): {avg['dice']:.6f}" in text


def test_train_model_unet_learns_and_keeps_best(tmp_path):
    tr, va = _toy_data(48, 32, seed=3), _toy_data(16, 32, seed=4)
    ckpt = tmp_path / "best.pth"
    lines = []
    torch.manual_seed(0)
    model = trainer.train_model(tr, va, model_type="UNet", epochs=8, batch_size=8, learning_rate=1e-3, patience=10,
                                device=DEV, model_save_path=str(ckpt), log=lines.append)
    assert not model.training and ckpt.exists()
    val = [float(s.split(":")[1]) for s in lines if s.strip().startswith("Val Loss")]
    # the fp32 oracle model reaches a validation loss of 0.22 / accuracy 0.91 on this problem in the same 48 steps
    assert len(val) == 8 and min(val[1:]) < val[0] and min(val) < 0.5
    sd = torch.load(str(ckpt), map_location="cpu", weights_only=True)
    ref = ref_cpu.RefUNet(3, 1)
    ref.load_state_dict(sd)                            # same 136 keys, shapes and dtypes
    metrics = trainer.evaluate_model(model, va, DEV)
    assert metrics["accuracy"] > 0.75
    # the returned model carries the checkpoint's weights
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k


def test_train_model_lunext_cldice_runs_and_early_stops(tmp_path):
    tr, va = _toy_data(16, 32, seed=5), _toy_data(8, 32, seed=6)
    ckpt = tmp_path / "best_l.pth"
    lines = []
    torch.manual_seed(0)
    model = trainer.train_model(tr, va, model_type="LUNeXt", epochs=4, batch_size=8, learning_rate=1e-3, patience=2,
                                device=DEV, custom_loss="cl_dice", model_save_path=str(ckpt), log=lines.append)
    assert not model.training and ckpt.exists()
    val = [float(s.split(":")[1]) for s in lines if s.strip().startswith("Val Loss")]
    assert 1 <= len(val) <= 4 and all(np.isfinite(val)) and all(0 <= v <= 1 for v in val)
    ref = ref_cpu.RefLUNeXt(3, 1)
    ref.load_state_dict(torch.load(str(ckpt), map_location="cpu", weights_only=True))
    with pytest.raises(ValueError):                     # the reference's LUNeXt keeps no intermediate_features either
        trainer.train_model(tr, va, model_type="LUNeXt", custom_loss="self_reg", device=DEV)


def test_train_kfold_single_process(tmp_path):
    """Every fold on one GPU (world 1): 2 x 2 short trainings; table, best checkpoint and test metrics come back."""
    from unet_amd import kfold
    images, masks = _toy_data(24, 32, seed=8)
    test = _toy_data(6, 32, seed=9)
    torch.manual_seed(0)
    rec, path, best = kfold.train_kfold(images, masks, test_data=test, model_type="UNet", k_fold=2, times=2, epochs=3,
                                        batch_size=6, eval_batch_size=4, results_file=str(tmp_path / "r.csv"),
                                        best_model_path=str(tmp_path / "best.pth"), device=DEV, log=lambda *_: None)
    assert [(r["Time"], r["Fold"]) for r in rec] == [(1, 1), (1, 2), (2, 1), (2, 2)]
    assert all(0.0 <= r["Dice"] <= 1.0 and r["Training Time"] > 0 for r in rec)
    assert os.path.exists(path) and (tmp_path / "r.csv").exists()
    assert sorted(os.listdir(tmp_path)) == ["best.pth", "r.csv"]              # per-fold checkpoints are removed
    if max(r["Dice"] for r in rec) > 0:
        assert set(best) == {"dice", "iou", "precision", "recall", "accuracy", "f1"}
        ref_cpu.RefUNet(3, 1).load_state_dict(torch.load(path, map_location="cpu", weights_only=True))


@pytest.mark.parametrize("b,hw,per_sample", [(5, 48, False), (5, 48, True), (1, 1024, True), (8, 256, False)])
def test_head_counts_equal_confusion_counts_of_the_same_logits(b, hw, per_sample):
    """predict_counts (uh_head_fwd_counts: binarise + tally inside the head launch, k_fold.py:289-290) against the
    two-pass form it replaces (uh_head_fwd, then uh_confusion_counts): logits and integer counts bit-exact."""
    torch.manual_seed(8)
    model = unet_amd.UNet(3, 1).to(DEV).eval()
    g = torch.Generator().manual_seed(hw + b)
    x = torch.rand(b, 3, hw, hw, generator=g).to(DEV)
    m = (torch.rand(b, 1, hw, hw, generator=g) > 0.4).float()
    m[0, 0, :2, :] = 0.5                                    # mask values outside {0, 1} fall into no class
    m = m.to(DEV)
    with torch.no_grad():
        model.final_conv.bias -= model(x).median()          # both predicted classes occur
        logits = model(x)
    planes = b if per_sample else 1
    want = torch.zeros(planes, 4, dtype=torch.int64, device=DEV)
    trainer.confusion_counts(logits, m, want.view(-1) if not per_sample else want, per_sample=per_sample)
    got = torch.full((planes, 4), 3, dtype=torch.int64, device=DEV)           # the launch ADDS to what is there
    out = model.predict_counts(x, m, got.view(-1) if not per_sample else got, per_sample=per_sample, return_logits=True)
    assert torch.equal(out, logits)
    assert torch.equal(got - 3, want)
    assert want[:, 0].sum().item() > 0 and want[:, 1].sum().item() > 0
    assert int(want.sum()) == b * hw * hw - 2 * hw          # the two rows of 0.5 are in no class
    again = torch.zeros(planes, 4, dtype=torch.int64, device=DEV)
    assert model.predict_counts(x, m, again.view(-1) if not per_sample else again, per_sample=per_sample) is None
    assert torch.equal(again, want)
    with pytest.raises(ValueError):
        model.predict_counts(x, m, torch.zeros(3, dtype=torch.int64, device=DEV))
