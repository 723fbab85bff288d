"""Self-regularisation loss on the device (unet_amd/self_reg.py, csrc/selfreg.hip) against the oracle.

  * the cases captured from the reference (tests/golden/selfreg_cases.npz, every value exact in bf16): LSRC, LIFD,
    loss and d(loss)/d(logits) to fp32 rounding, channel-subset seeds taken from the fixture;
  * feature maps inside wider buffers (channel windows), odd sizes and tile tails against oracle/self_reg.py;
  * a U-Net forward pass: the in-place path (bf16 activations read where they lie) equals the path over the
    materialised NCHW list and the oracle on that list; backward reaches the parameters; a stale list raises;
  * `train_model(custom_loss="self_reg")` runs the reference's loop with it.
Tolerances: the kernels sum in fp32 in a different order than the oracle: 2e-5 relative on the scalars.  LIFD hangs
on an ordering of channels by a mean of normalised values; where two channels agree to fp32 rounding (small maps of
quantised values make exact ties common) the reference's own order is an accident of its summation order.  So the
ordering is checked as "a permutation that sorts the oracle's importance up to rounding", LIFD exactly against the
oracle evaluated with the device's ordering, and against the reference's value within 2e-3.
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
from unet_amd.layout import Act, mp_rows               # noqa: E402
from unet_amd._lib import lib as L                     # noqa: E402
from unet_amd.self_reg import SelfRegLoss              # noqa: E402
from oracle import self_reg as osr                     # noqa: E402
from test_oracle_golden import selfreg_case            # noqa: E402

DEV = torch.device("cuda", 0)
RTOL = 2e-5


def _fixture(golden_dir):
    return np.load(os.path.join(golden_dir, "selfreg_cases.npz"), allow_pickle=False)


class _ActList(list):
    def acts(self):
        return list(self)


def _device_orders(acts, feats):
    """Channel order of every feature but the last from the two ordering kernels, each checked against the
    oracle's importance."""
    orders = []
    s = torch.cuda.current_stream().cuda_stream
    for f, a in zip(feats[:-1], acts[:-1]):
        B, C, H, W = a.B, a.C, a.H, a.W
        sp = L.uh_selfreg_stat_parts(B, H, W, C)
        stats = torch.empty(B * sp * 2 * C, dtype=torch.float32, device=DEV)
        order = torch.empty(C, dtype=torch.int32, device=DEV)
        assert L.uh_selfreg_plane_stats(B, H, W, C, a.ptr, a.cstride, a.coff, stats.data_ptr(), s) == 0
        assert L.uh_selfreg_rank(B, H, W, C, stats.data_ptr(), order.data_ptr(), s) == 0
        got = order.cpu().numpy()
        assert sorted(got.tolist()) == list(range(C))
        want_order, imp = osr.channel_order(f)
        imp = imp.numpy()
        np.testing.assert_allclose(imp[got], imp[want_order.numpy()], rtol=1e-5, atol=1e-9)
        orders.append(got)
    return orders


def _check_aux(crit, acts, feats, seeds, reference=None):
    aux = crit.auxiliaries(_ActList(acts)).cpu().numpy()
    orders = _device_orders(acts, feats)
    np.testing.assert_allclose(aux, [float(osr.lsrc(feats, seeds)), float(osr.lifd(feats, orders))], rtol=RTOL)
    if reference is not None:
        np.testing.assert_allclose(aux, reference, rtol=2e-3)
        np.testing.assert_allclose(aux[0], reference[0], rtol=RTOL)
    return aux


@pytest.mark.parametrize("name", ["unet_like", "dead_channels", "final_wider", "single_feature"])
def test_reference_cases(golden_dir, name):
    feats, logits, targets, seeds, want, dwant = selfreg_case(_fixture(golden_dir), name)
    crit = SelfRegLoss(seed_of=lambda idx: seeds[idx])
    dfeats = [f.to(DEV) for f in feats]
    if len(feats) <= 1:
        assert crit.auxiliaries(dfeats) is None
        aux = np.zeros(2)
    else:
        aux = _check_aux(crit, crit._acts_of(dfeats), feats, seeds, reference=want[:2])
    x = logits.to(DEV).requires_grad_(True)
    loss = crit(x, targets.to(DEV), dfeats)
    np.testing.assert_allclose(float(loss.detach()), 0.6 * want[2] + 0.015 * float(aux[0] + aux[1]), rtol=RTOL)
    np.testing.assert_allclose(float(loss.detach()), want[4], rtol=2e-4)
    (3.0 * loss).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), 3.0 * dwant, rtol=1e-4, atol=1e-9)


def test_unsupported_inputs_raise(golden_dir):
    feats, logits, targets, seeds, _, _ = selfreg_case(_fixture(golden_dir), "two_channels")
    crit = SelfRegLoss()
    with pytest.raises(ValueError):                     # 2 and 1 channels: not multiples of 8
        crit(logits.to(DEV), targets.to(DEV), [f.to(DEV) for f in feats])
    with pytest.raises(NotImplementedError):            # more than one logit channel
        crit(torch.zeros(2, 2, 8, 8, device=DEV), torch.zeros(2, 2, 8, 8, device=DEV), [])
    with pytest.raises(RuntimeError):
        crit(torch.zeros(2, 1, 8, 8), torch.zeros(2, 1, 8, 8), [])
    f = [torch.zeros(1, 8, 12, 12, device=DEV), torch.zeros(1, 8, 16, 16, device=DEV)]
    with pytest.raises(NotImplementedError):            # 16 -> 12 is no power-of-two reduction
        crit(torch.zeros(1, 1, 16, 16, device=DEV), torch.zeros(1, 1, 16, 16, device=DEV), f)


def _relu_like(g, *shape):
    return torch.clamp(torch.randint(-128, 256, shape, generator=g), min=0).float() / 64.0


def _windowed_act(t, cstride, coff):
    """Padded-linear bf16 buffer of `cstride` channels holding t [B,C,H,W] at channel offset coff, junk elsewhere."""
    B, C, H, W = t.shape
    guard = L.uh_guard_rows(W)
    buf = torch.full((guard * 2 + mp_rows(B, H, W), cstride), 7.0, dtype=torch.bfloat16, device=DEV)
    a = Act(buf, B, H, W, C, cstride, coff, guard)
    a.fill_from_nchw(t.to(DEV))
    return a


@pytest.mark.parametrize("B,size,spec", [
    (3, 48, [(128, 1, 256, 128), (64, 2, 64, 0), (520, 4, 520, 0), (72, 8, 80, 8), (64, 1, 192, 64)]),
    (1, 16, [(8, 1, 8, 0), (1024, 16, 1024, 0), (40, 2, 48, 0), (24, 1, 24, 0)]),
    (2, 272, [(64, 1, 64, 0), (64, 2, 128, 64), (64, 1, 64, 0)]),
])
def test_channel_windows_and_tails(B, size, spec):
    """(C, reduction, cstride, coff) per feature, the last one being the pooled side."""
    g = torch.Generator().manual_seed(B * 1000 + size)
    feats = [_relu_like(g, B, c, size // r, size // r) for c, r, _, _ in spec]
    feats[0][:, 3] = 0.0
    acts = [_windowed_act(f, cs, co) for f, (_, _, cs, co) in zip(feats, spec)]
    seeds = [11 * i + 5 for i in range(len(spec))]
    _check_aux(SelfRegLoss(seed_of=lambda idx: seeds[idx]), acts, feats, seeds)


def _toy_batch(B, size, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, size, size, generator=g)
    m = (x.mean(1, keepdim=True) > 0.5).float()
    return x.to(DEV), m.to(DEV)


def test_unet_features_in_place_equal_materialised_and_oracle():
    torch.manual_seed(0)
    model = unet_amd.create_unet(3, 1).to(DEV).train()
    x, m = _toy_batch(4, 64, 1)
    seeds = list(range(100, 115))
    crit = SelfRegLoss(seed_of=lambda idx: seeds[idx])
    logits = model(x)
    feats = model.intermediate_features
    assert feats.acts() is not None
    aux_in_place = crit.auxiliaries(feats).cpu().numpy()
    loss = crit(logits, m, feats)
    tensors = [t for t in feats]                           # materialises the 15 NCHW fp32 tensors
    assert feats.acts() is None and len(tensors) == 15
    aux_list = crit.auxiliaries(tensors).cpu().numpy()
    np.testing.assert_array_equal(aux_in_place, aux_list)   # same kernels on the same bf16 values
    cpu = [t.cpu() for t in tensors]
    _check_aux(crit, crit._acts_of(tensors), cpu, seeds)
    ref = osr.self_reg_loss(logits.detach().cpu(), m.cpu(), cpu, seeds)
    np.testing.assert_allclose(float(loss.detach()), float(ref), rtol=2e-4)
    loss.backward()
    g = model.final_conv.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    # eval / no_grad: the list is readable until the next pass takes its buffers
    model.eval()
    with torch.no_grad():
        model(x)
        old = model.intermediate_features
        crit.auxiliaries(old)
        model(x)
    with pytest.raises(RuntimeError):
        crit.auxiliaries(old)
    with pytest.raises(RuntimeError):
        len(old)


def test_train_model_with_self_reg(tmp_path):
    g = np.random.default_rng(5)
    images = g.random((16, 32, 32, 3), dtype=np.float32)
    masks = (images.mean(-1, keepdims=True) > 0.5).astype(np.float32)
    lines = []
    torch.manual_seed(0)
    model = trainer.train_model((images, masks), (images[:8], masks[:8]), model_type="UNet", epochs=3, batch_size=8,
                                learning_rate=1e-3, patience=5, device=DEV, custom_loss="self_reg",
                                model_save_path=str(tmp_path / "sr.pth"), log=lines.append)
    assert not model.training and (tmp_path / "sr.pth").exists()
    val = [float(s.split(":")[1]) for s in lines if s.strip().startswith("Val Loss")]
    assert len(val) == 3 and all(np.isfinite(val)) and all(v > 0 for v in val)
