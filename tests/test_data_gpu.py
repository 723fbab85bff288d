"""The device-resident data set (unet_amd.data) against the reference's host pipeline: a gathered batch must hold
exactly the values of torch.FloatTensor(images).permute(0, 3, 1, 2)[idx] (trainer/training_utils.py:52-55), bit for
bit, whether the set is stored as bytes or as fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import gpu_present            # noqa: E402

if not gpu_present():
    pytest.skip("needs a GPU", allow_module_level=True)

from unet_amd.data import DeviceDataset, DeviceTensorSet      # noqa: E402

DEV = torch.device("cuda", 0)


@pytest.mark.parametrize("N,H,W,C", [(9, 32, 48, 3), (5, 16, 16, 1), (4, 7, 9, 3), (3, 12, 20, 2), (3, 8, 8, 4),
                                     (2, 6, 6, 5)])
@pytest.mark.parametrize("kind", ["bytes", "float"])
def test_gathered_batch_equals_the_reference_tensor(N, H, W, C, kind):
    rng = np.random.default_rng(N * 100 + C)
    if kind == "bytes":
        images = np.array(rng.integers(0, 256, size=(N, H, W, C), dtype=np.uint8), dtype=np.float32) / 255.0
    else:
        images = rng.standard_normal((N, H, W, C)).astype(np.float32)
    masks = (rng.random((N, H, W, 1)) > 0.5).astype(np.float32)
    ds = DeviceDataset(images, masks, DEV)
    assert ds.images.is_bytes == (kind == "bytes") and ds.masks.is_bytes
    want_x = torch.FloatTensor(images).permute(0, 3, 1, 2)            # the reference's data set tensors
    want_m = torch.FloatTensor(masks).permute(0, 3, 1, 2)
    idx = torch.tensor([N - 1, 0, 0, N // 2, 1][:max(1, min(5, N + 1))])
    x, m = ds.batch(idx)
    assert x.shape == (len(idx), C, H, W) and x.dtype == torch.float32 and x.is_contiguous()
    assert torch.equal(x.cpu().view(torch.int32), want_x[idx].contiguous().view(torch.int32))
    assert torch.equal(m.cpu().view(torch.int32), want_m[idx].contiguous().view(torch.int32))
    x, m = ds.slice(1, N)
    assert torch.equal(x.cpu(), want_x[1:N]) and torch.equal(m.cpu(), want_m[1:N])
    with pytest.raises(IndexError):
        ds.batch(torch.tensor([N]))


def test_three_dimensional_masks_and_empty_batches():
    rng = np.random.default_rng(5)
    masks = (rng.random((4, 8, 8)) > 0.5).astype(np.float32)            # [N,H,W]: a channel axis is added
    t = DeviceTensorSet(masks, DEV)
    got = t.gather(torch.tensor([3, 1], device=DEV))
    assert torch.equal(got.cpu(), torch.from_numpy(masks)[[3, 1]].unsqueeze(1))
    assert t.gather(torch.empty(0, dtype=torch.int64, device=DEV)).shape == (0, 1, 8, 8)


def test_uint8_arrays_keep_their_values():
    """torch.FloatTensor(uint8 array) holds 0..255, not k/255 (training_utils.py:52-55): a uint8 {0,1} mask must come
    out as {0.0, 1.0} and a uint8 image as its byte values."""
    rng = np.random.default_rng(11)
    images = rng.integers(0, 256, size=(5, 10, 12, 3), dtype=np.uint8)
    masks = rng.integers(0, 2, size=(5, 10, 12, 1), dtype=np.uint8)
    ds = DeviceDataset(images, masks, DEV)
    assert ds.images.is_bytes and ds.masks.is_bytes
    idx = torch.tensor([4, 0, 2])
    x, m = ds.batch(idx)
    want_x = torch.FloatTensor(images.astype(np.float32)).permute(0, 3, 1, 2)[idx]
    want_m = torch.FloatTensor(masks.astype(np.float32)).permute(0, 3, 1, 2)[idx]
    assert torch.equal(x.cpu(), want_x) and torch.equal(m.cpu(), want_m)
    assert set(m.cpu().unique().tolist()) <= {0.0, 1.0} and float(x.max()) > 1.0
