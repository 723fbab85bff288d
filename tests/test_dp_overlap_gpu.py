"""The data-parallel step must not put a collective in front of the weight gradients: each bucket's all-reduce is waited
for on the trainer's third stream (with that bucket's Adam update), never on the stream the weight-gradient kernels
run on.  Shown on one GPU by replacing dist.all_reduce with a handle whose wait() makes the waiting stream sleep
for about 2 ms, as a slow ring all-reduce would: weight-gradient launches issued after bucket 0 became ready must start
before that sleep has ended.  (Counterpart of the loop at trainer/training_utils.py:125-183, which has no collective.)"""
import os

import pytest
import torch

from conftest import gpu_present

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not gpu_present(), reason="needs a GPU")]


class _SlowWork:
    """What dist.all_reduce(async_op=True) returns, with a wait() that costs the waiting stream `cycles` of sleep."""

    def __init__(self, log, cycles):
        self.log, self.cycles = log, cycles

    def wait(self):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(self.cycles)
        e1.record()
        self.log.append((torch.cuda.current_stream().cuda_stream, e0, e1))
        return True


def test_weight_gradients_do_not_wait_for_collectives(monkeypatch, tmp_path):
    import torch.distributed as dist
    import unet_amd
    from unet_amd import engine
    from unet_amd.dp import DataParallelTrainer, KernelTimer
    dev = torch.device("cuda", 0)
    # cycles of torch.cuda._sleep per millisecond on this device
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    a.record()
    torch.cuda._sleep(2_000_000)
    b.record()
    torch.cuda.synchronize()
    cycles = int(2_000_000 * 2.0 / a.elapsed_time(b))          # about 2 ms

    monkeypatch.setenv("UNET_AMD_FORCE_COLLECTIVES", "1")
    store = dist.FileStore(str(tmp_path / "store"), 1)
    dist.init_process_group("gloo", store=store, rank=0, world_size=1)
    try:
        log = []
        monkeypatch.setattr(dist, "broadcast", lambda *a, **k: None)
        monkeypatch.setattr(dist, "all_reduce", lambda t, **k: _SlowWork(log, cycles))
        torch.manual_seed(0)
        model = unet_amd.UNet(3, 1).to(dev)
        trainer = DataParallelTrainer(model, lr=1e-3, bucket_bytes=25 << 20)
        assert trainer.collectives and len(trainer.buckets.buckets) >= 4
        x = torch.rand(4, 3, 64, 64, device=dev)
        m = (torch.rand(4, 1, 64, 64, device=dev) > 0.5).float()
        trainer.train_step(x, m)                               # warm-up: operand packing, buffer pool
        torch.cuda.synchronize()
        p_before = trainer.optimizer.p.clone()
        del log[:]
        trainer.timer = timer = KernelTimer()
        origin = torch.cuda.Event(enable_timing=True)
        origin.record()
        trainer.train_step(x, m)
        torch.cuda.synchronize()
        side = engine.side_stream_for(dev).cuda_stream
        opt = engine.side_stream_for(dev, "opt").cuda_stream
        main = torch.cuda.current_stream().cuda_stream
        assert len({side, opt, main}) == 3
        # every bucket's collective is waited for on the third stream during backward, and once more on the main stream
        # at the end of the step; never on the weight-gradient stream
        streams = [s for s, _, _ in log]
        assert side not in streams
        nb = len(trainer.buckets.buckets)
        assert streams.count(opt) == nb and streams.count(main) == nb
        _, s0, e0 = next(r for r in log if r[0] == opt)        # bucket 0's wait
        t_sleep0, t_sleep1 = origin.elapsed_time(s0), origin.elapsed_time(e0)
        assert t_sleep1 - t_sleep0 > 1.0
        starts = sorted(origin.elapsed_time(r[1]) for r in timer.records["conv_wgrad"])
        # weight gradients that were issued after bucket 0's wait (they start after it starts) begin before it is over
        during = [t for t in starts if t_sleep0 < t < t_sleep1]
        assert during, (t_sleep0, t_sleep1, starts)
        # and the optimizer did step every bucket behind its collective
        assert not torch.equal(p_before, trainer.optimizer.p)
        assert trainer.optimizer.steps == 2
    finally:
        dist.destroy_process_group()


def test_train_step_under_grad_scaler_equals_the_unscaled_step():
    """train_step(scaler=GradScaler) (training_utils.py:153-155): the loss gradient is scaled by 2^16, the optimizer
    unscales inside its kernel.  Scaling by a power of two commutes with every bf16 / fp32 rounding of the backward
    pass, so two steps end at the parameters of the unscaled run (up to the reciprocal's rounding: none here)."""
    import unet_amd
    from unet_amd.dp import DataParallelTrainer
    dev = torch.device("cuda", 0)
    x = torch.rand(4, 3, 32, 32, device=dev)
    m = (torch.rand(4, 1, 32, 32, device=dev) > 0.5).float()
    out = []
    for amp in (False, True):
        torch.manual_seed(0)
        model = unet_amd.UNet(3, 1).to(dev)
        trainer = DataParallelTrainer(model, lr=1e-3, standalone=True)
        scaler = torch.amp.GradScaler("cuda") if amp else None
        losses = [trainer.train_step(x, m, scaler=scaler).item() for _ in range(2)]
        out.append((losses, trainer.optimizer.p.clone()))
        if amp:
            assert scaler.get_scale() == 65536.0 and trainer.optimizer._steps_dev.item() == 2.0
    assert out[0][0] == out[1][0]
    torch.testing.assert_close(out[0][1], out[1][1], rtol=1e-6, atol=1e-8)
    assert not torch.equal(out[0][1], torch.zeros_like(out[0][1]))
