"""uh_bce_logits_noalloc (the entry point that allocates nothing: its partial sums live in an array of the library's
code object) against uh_bce_logits with a caller workspace, through the C-ABI: identical loss and gradient, bit for
bit, on two calls in a row, the second with a smaller input (the shared array keeps the first call's contents)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import gpu_present            # noqa: E402

if not gpu_present():
    pytest.skip("needs a GPU", allow_module_level=True)

from unet_amd import _lib                    # noqa: E402
from unet_amd._lib import lib                # noqa: E402

DEV = torch.device("cuda", 0)


def _both(n, seed, scale):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, generator=g) * 3).to(DEV)
    t = (torch.rand(n, generator=g) > 0.5).float().to(DEV)
    s = torch.cuda.current_stream().cuda_stream
    ws_bytes = lib.uh_bce_logits_workspace(n)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=DEV)
    loss_a, dx_a = torch.full((1,), -1.0, device=DEV), torch.empty_like(x)
    loss_b, dx_b = torch.full((1,), -2.0, device=DEV), torch.empty_like(x)
    _lib.call("uh_bce_logits", n, x.data_ptr(), t.data_ptr(), scale, loss_a.data_ptr(), dx_a.data_ptr(), ws.data_ptr(),
              ws_bytes, s)
    before = torch.cuda.memory_stats(DEV)["allocation.all.allocated"]
    _lib.call("uh_bce_logits_noalloc", n, x.data_ptr(), t.data_ptr(), scale, loss_b.data_ptr(), dx_b.data_ptr(), s)
    assert torch.cuda.memory_stats(DEV)["allocation.all.allocated"] == before
    # torch's counters only see torch's allocator: a second, warmed-up call must also leave the DRIVER's free-byte count
    # where it was (a hipMalloc / hipMallocAsync inside the entry point would move it)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(DEV)[0]
    _lib.call("uh_bce_logits_noalloc", n, x.data_ptr(), t.data_ptr(), scale, loss_b.data_ptr(), dx_b.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(DEV)[0] == free_before
    torch.cuda.synchronize()
    ref = torch.nn.functional.binary_cross_entropy_with_logits(x, t)
    return loss_a, dx_a, loss_b, dx_b, ref


def test_no_alloc_entry_point_matches_workspace_entry_point_twice():
    for n, seed, scale in ((32 * 256 * 256, 1, 0.5), (3 * 100 * 76 + 1, 2, 1.0)):   # large, then smaller and ragged
        loss_a, dx_a, loss_b, dx_b, ref = _both(n, seed, scale)
        assert torch.equal(loss_a, loss_b), (loss_a.item(), loss_b.item())
        assert torch.equal(dx_a, dx_b)
        assert abs(loss_b.item() - ref.item()) <= 1e-5 * abs(ref.item())


def test_no_alloc_entry_point_refuses_sizes_beyond_its_maximum():
    x = torch.zeros(8, device=DEV)
    st = lib.uh_bce_logits_noalloc((1 << 32) + 1, x.data_ptr(), x.data_ptr(), 1.0, x.data_ptr(), None, 0)
    assert st != 0
    assert b"exceeds the maximum" in lib.uh_last_error()
