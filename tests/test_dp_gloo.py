"""Data-parallel path on CPU, world_size 2, gloo backend: the bucket layout, the order in which gradients become
ready, the asynchronous bucketed all-reduce and the 1/world scaling of unet_amd.dp, driven by gradients that the
CPU oracle computes per shard (the HIP kernels need a GPU; everything around them does not).
Checked property (DDP semantics): after the step every rank holds the MEAN of the per-shard gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import ref_cpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import unet_amd
    from unet_amd.dp import DataParallelTrainer, backward_param_order
    torch.manual_seed(100 + rank)                 # different initial weights per rank: the trainer must broadcast rank 0's
    model = unet_amd.UNet(3, 1)                   # parameter container only; no forward on CPU
    trainer = DataParallelTrainer(model, lr=1e-3, bucket_bytes=8 << 20)
    assert trainer.world == world and len(trainer.buckets.buckets) >= 4
    ref = ref_cpu.RefUNet(3, 1)
    ref.load_state_dict(model.state_dict())       # weights after the broadcast
    x, m = ref_cpu.synth_batch(2 * world, 3, 16, 16)
    idx = DataParallelTrainer.shard(2 * world, epoch=0, rank=rank, world=world)
    loss = nn.BCEWithLogitsLoss()(ref.train()(x[idx]), m[idx])
    (loss / world).backward()                     # the trainer seeds backward with d(loss/world)
    # play the backward pass: gradients appear last layer first, into the bucket views
    trainer.buckets.reset()
    for name in backward_param_order(model):
        trainer.buckets.views[name].copy_(dict(ref.named_parameters())[name].grad)
        trainer._on_ready((name,))
    for b in trainer.buckets.buckets:
        w = b.pop("work", None)
        assert w is not None, "every bucket must have been reduced"
        w.wait()
    flat = trainer.buckets.flat[:trainer.buckets.total].clone()     # the buffer is padded to a multiple of 4 floats
    assert trainer.buckets.flat[trainer.buckets.total:].abs().sum().item() == 0
    assert all(p.grad.data_ptr() == trainer.buckets.views[n].data_ptr() for n, p in model.named_parameters())
    torch.save({"flat": flat, "idx": idx, "w0": model.conv1_1.weight.detach().clone()},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_allreduce_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rank{i}.pt"), weights_only=True) for i in range(world)]
    assert torch.equal(r[0]["flat"], r[1]["flat"]), "ranks disagree after all-reduce"
    assert torch.equal(r[0]["w0"], r[1]["w0"]), "initial weights were not broadcast"
    assert sorted(torch.cat([r[0]["idx"], r[1]["idx"]]).tolist()) == list(range(2 * world)), "shards must partition"
    # single-process answer: mean over shards of the per-shard gradients
    import unet_amd
    from unet_amd.dp import backward_param_order
    torch.manual_seed(100)
    model = unet_amd.UNet(3, 1)                   # rank 0's weights
    ref = ref_cpu.RefUNet(3, 1)
    ref.load_state_dict(model.state_dict())
    assert torch.equal(ref.conv1_1.weight, r[0]["w0"])
    x, m = ref_cpu.synth_batch(2 * world, 3, 16, 16)
    want = None
    for i in range(world):
        ref.zero_grad()
        nn.BCEWithLogitsLoss()(ref.train()(x[r[i]["idx"]]), m[r[i]["idx"]]).backward()
        g = torch.cat([dict(ref.named_parameters())[n].grad.flatten() for n in backward_param_order(model)])
        want = g / world if want is None else want + g / world
    torch.testing.assert_close(r[0]["flat"], want, rtol=1e-5, atol=1e-7)


def test_bucket_layout():
    import unet_amd
    from unet_amd.dp import GradBuckets, backward_param_order
    model = unet_amd.UNet(3, 1)
    order = backward_param_order(model)
    assert order[0] == "final_conv.weight" and order[-1] == "conv1_1.bias" and len(order) == 82
    b = GradBuckets(model, bucket_bytes=25 << 20)
    assert sum(x["tensor"].numel() for x in b.buckets) == 31043521      # every parameter exactly once
    assert all(x["tensor"].numel() * 4 >= (25 << 20) for x in b.buckets[:-1])
    # buckets are contiguous slices in backward order
    off = 0
    for x in b.buckets:
        assert x["tensor"].data_ptr() == b.flat.data_ptr() + off * 4
        off += x["tensor"].numel()


class _Toy(nn.Module):
    """three used layers and one that never takes part in forward (like LUNeXt's att*.output_conv)"""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 5)
        self.unused = nn.Linear(5, 5)
        self.b = nn.Linear(5, 3)
        self.c = nn.Linear(3, 1)

    def forward(self, x):
        return self.c(torch.tanh(self.b(torch.tanh(self.a(x)))))


def _avg_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd.dp import GradAverager
    torch.manual_seed(7 + rank)                   # different weights per rank until the broadcast
    model = _Toy()
    avg = GradAverager(model.parameters(), bucket_bytes=64)      # tiny buckets: several of them, one holding `unused`
    assert len(avg.buckets) >= 3
    avg.broadcast_from_rank0(model)
    g = torch.Generator().manual_seed(100 + rank)
    x, y = torch.randn(4, 6, generator=g), torch.randn(4, 1, generator=g)
    for step in range(2):                          # twice: stale bucket contents must not leak into the second mean
        model.zero_grad(set_to_none=True)
        nn.functional.mse_loss(model(x * (step + 1)), y).backward()
        local = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        avg.average()
        torch.save({"local": local, "mean": {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None},
                    "w": model.a.weight.detach().clone(), "none": [n for n, p in model.named_parameters() if p.grad is None]},
                   os.path.join(out_dir, f"avg{rank}_{step}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_grad_averager_world2_with_unused_parameters(tmp_path):
    world = 2
    mp.spawn(_avg_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for step in range(2):
        r = [torch.load(os.path.join(str(tmp_path), f"avg{k}_{step}.pt")) for k in range(world)]
        assert torch.equal(r[0]["w"], r[1]["w"])                       # same weights after the broadcast
        assert r[0]["none"] == r[1]["none"] == ["unused.weight", "unused.bias"]
        for n in r[0]["local"]:
            want = (r[0]["local"][n] + r[1]["local"][n]) / 2
            assert torch.allclose(r[0]["mean"][n], want, rtol=1e-6, atol=1e-8), n
            assert torch.equal(r[0]["mean"][n], r[1]["mean"][n]), n
            assert not torch.equal(r[0]["local"][n], r[1]["local"][n]), n   # the shards really differed


def test_shards_have_equal_length_when_items_do_not_divide():
    """97 items over 2 ranks, batch 16: 49 + 48 samples would be 4 steps on one rank and 3 on the other."""
    from unet_amd.dp import DataParallelTrainer
    for n, world in ((97, 2), (5, 4), (8, 8), (3, 8), (1, 2)):
        shards = [DataParallelTrainer.shard(n, epoch=3, rank=r, world=world, seed=1) for r in range(world)]
        assert len({len(s) for s in shards}) == 1, (n, world, [len(s) for s in shards])
        assert set(torch.cat(shards).tolist()) == set(range(n))          # every item is still seen
        assert len(torch.cat(shards)) == -(-n // world) * world
    assert len(DataParallelTrainer.shard(96, 0, 0, 2)) == 48 and len(DataParallelTrainer.shard(0, 0, 0, 2)) == 0


def _fit_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_amd import trainer
    lr = [1e-3]
    sched = trainer.PlateauLR(lambda: lr[0], lambda v: lr.__setitem__(0, v), patience=1, factor=0.5)
    # per-replica validation losses that would stop rank 1 two epochs before rank 0
    local = {0: [1.0, 0.9, 0.8, 0.7, 0.71, 0.72, 0.73, 0.74], 1: [1.0, 0.9, 0.91, 0.92, 0.93, 0.6, 0.5, 0.4]}[rank]
    steps = []

    def train_epoch(epoch):
        t = torch.ones(1)
        dist.all_reduce(t)                         # stands for the step's gradient all-reduce: needs every rank
        steps.append(epoch)
        return 0.0
    saved = []
    tr, va, best = trainer.fit_loop(len(local), 3, train_epoch, lambda e: local[e], sched, lambda: saved.append(len(steps)),
                                    log=lambda *_: None, agree=trainer.rank0_value)
    torch.save({"epochs": len(steps), "lr": lr[0], "val": va, "best": best, "saved": saved},
               os.path.join(out_dir, f"fit{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_fit_loop_decisions_agree_across_ranks(tmp_path):
    """learning-rate schedule, best checkpoint and early stop follow rank 0's validation loss on every rank"""
    world = 2
    mp.spawn(_fit_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"fit{i}.pt"), weights_only=True) for i in range(world)]
    assert r[0] == r[1]
    assert r[0]["epochs"] == 7 and r[0]["best"] == 0.7 and r[0]["lr"] == 5e-4
