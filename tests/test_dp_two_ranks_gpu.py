"""The data-parallel step with the HIP kernels and two ranks (tests/dp_worker.py; the ranks share the GPU and the
all-reduces are staged through host memory over gloo): after a few steps every rank holds bit-identical parameters, and
they equal what one process gets by running the two shards itself, averaging the gradients and stepping the same
optimizer.

The ranks run before this process has opened a GPU context (tests/conftest.py::pytest_collection_modifyitems).  A rank
whose device work stops completing reports the launches it is behind and exits; that, like any other failure of the
ranks, fails the test."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import gpu_present            # noqa: E402

if not gpu_present():
    pytest.skip("needs a GPU", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device("cuda", 0)


def test_two_ranks_agree_and_match_manual_averaging(two_rank_run):
    r = two_rank_run
    steps, world, out = r["steps"], r["world"], r["out_dir"]
    stall = [ln for ln in r["log"].splitlines() if ln.startswith("STALL")]
    assert not stall, "a rank's device work stopped completing:\n" + "\n".join(stall)
    assert r["returncode"] == 0, r["log"]
    ranks = [torch.load(os.path.join(out, f"rank{i}.pt"), weights_only=True) for i in range(world)]
    assert torch.equal(ranks[0]["params"], ranks[1]["params"])
    assert all(torch.isfinite(torch.tensor(rk["losses"])).all() for rk in ranks)

    # the same training in one process: shard by shard, gradients averaged, the same flat Adam
    import unet_amd
    from unet_amd.dp import DataParallelTrainer
    x, m = ranks[0]["x"], ranks[0]["m"]
    torch.manual_seed(100)                                          # rank 0's initial weights
    model = unet_amd.UNet(3, 1).to(DEV)
    tr = DataParallelTrainer(model, lr=1e-3, standalone=True)
    names = [n for n, _ in model.named_parameters()]
    losses = []
    for _ in range(steps):
        acc, per = None, []
        for rk in range(world):
            per.append(float(tr.train_step(x[rk::world].to(DEV), m[rk::world].to(DEV), step_optimizer=False)))
            flat = tr.buckets.flat.clone()                           # every gradient, in backward order
            acc = flat if acc is None else acc + flat
        tr.buckets.flat.copy_(acc / world)
        tr.optimizer.step()
        losses.append(per)
    flat = torch.cat([p.detach().flatten() for p in model.parameters()]).cpu()
    for rk in range(world):
        assert ranks[rk]["losses"] == pytest.approx([l[rk] for l in losses], rel=1e-5, abs=1e-7)
    # a + b is the same sum in either order; Adam on identical gradients gives identical parameters up to the
    # order in which the shards' BatchNorm running statistics were updated (not compared)
    torch.testing.assert_close(ranks[0]["params"], flat, rtol=1e-4, atol=1e-6)
    assert len(names) == 82
