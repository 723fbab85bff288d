"""Host-side operand repack plans (unet_amd/packing.py): every entry writes inside its buffer, buffers of one
direction do not overlap, and the plans cover exactly the layers that run as MFMA GEMMs.  CPU only."""
import torch

import unet_amd
from unet_amd import packing


def _extent(N, T, K, on, ot):
    """last element index + 1 that pack(dst[n*on + t*ot + k]) touches"""
    return (N - 1) * on + (T - 1) * ot + (K - 1) + 1


def test_lunext_plan_entries_fit_their_buffers():
    torch.manual_seed(0)
    m = unet_amd.LUNeXt(3, 1)
    sizes, entries = packing.lunext_pack_plan(m)
    assert set(sizes) == {"fwd", "bwd"}
    keys = {d: set() for d in sizes}
    for (d, key, off, p, N, T, K, sn, st, sk, on, ot) in entries:
        assert key in sizes[d], (d, key)
        keys[d].add(key)
        assert off + _extent(N, T, K, on, ot) <= sizes[d][key], (d, key)
        # the source side stays inside the parameter
        assert (N - 1) * sn + (T - 1) * st + (K - 1) * sk < p.numel(), (d, key)
    assert keys["fwd"] == set(sizes["fwd"]) and keys["bwd"] == set(sizes["bwd"])
    # four strided convolutions, four transposed convolutions, three gates at 64 channels and more
    assert sum(k.startswith("down") for k in sizes["fwd"]) == 4
    assert sum(k.startswith("up") and not k.endswith(".f32") for k in sizes["fwd"]) == 4
    assert sorted(k for k in sizes["fwd"] if k.startswith("att") and not k.endswith(".f32")) == \
        [f"att{k}.{w}" for k in (1, 2, 3) for w in ("w12", "w3", "wcat")]
    # operands narrower than 64 are padded to 64 rows / columns
    assert sizes["fwd"]["down1"] == 64 * 64 and sizes["fwd"]["up4"] == 64 * 64
    # the fp32 tables of the 8-channel level: the last transposed convolution's bias, its gate's table, the head's weight
    assert {k: n for k, n in sizes["fwd"].items() if k.endswith(".f32")} == \
        {"up4.bias.f32": 8, "att4.gate8.f32": 216, "head.w.f32": 8}
    assert not any(k.endswith(".f32") for k in sizes["bwd"])


def test_unet_plan_covers_every_gemm_layer_once_per_direction():
    torch.manual_seed(0)
    m = unet_amd.UNet(3, 1)
    plan = packing.unet_pack_plan(m)
    for d in ("fwd", "bwd"):
        names = [e[1] for e in plan if e[0] == d]
        assert len(names) == len(set(names)) == 17 + 4            # 3x3 layers but the first, transposed convolutions
        assert "conv1_1" not in names
    for (d, name, p, N, T, K, sn, st, sk, on, ot, rev) in plan:
        assert N * T * K == p.numel(), name
        assert _extent(N, T, K, on, ot) == N * T * K, name        # dense [N][T][K] operand
        assert (N - 1) * sn + (T - 1) * st + (K - 1) * sk == p.numel() - 1, name


def test_lunext_plan_refuses_more_than_eight_input_channels():
    import pytest
    with pytest.raises(ValueError):
        packing.lunext_pack_plan(unet_amd.LUNeXt(9, 1))
    for cin in (1, 3, 8):                                         # what the 8-channel first level carries
        sizes, entries = packing.lunext_pack_plan(unet_amd.LUNeXt(cin, 2))
        assert sizes["fwd"]["head.w.f32"] == 16 and sizes["fwd"]["att4.gate8.f32"] == 216
        assert ("up4.bias.f32" in sizes["fwd"]) == (cin != 8)
        for (d, key, off, p, N, T, K, sn, st, sk, on, ot) in entries:
            assert off + (N - 1) * on + (T - 1) * ot + K <= sizes[d][key], (cin, key)
