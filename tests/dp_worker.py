"""Worker of tests/test_dp_two_ranks_gpu.py: one rank of a data-parallel U-Net run whose ranks share one GPU.  RCCL refuses two
ranks on one device and gloo's own device-tensor path does not complete here, so the collectives are staged through
host memory: copy to the host on the CURRENT stream (which therefore has to be ordered after every producer of the
tensor, exactly what RCCL's stream would wait for), gloo on the host copy, copy back on the current stream.
Writes rank<r>.pt with the losses and the parameters."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import unet_amd                                    # noqa: E402
from unet_amd.dp import DataParallelTrainer        # noqa: E402


class _Done:
    def wait(self):
        return True


_all_reduce, _broadcast = dist.all_reduce, dist.broadcast


def staged_all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
    h = t.detach().cpu()
    _all_reduce(h, op=op, group=group)
    t.copy_(h)
    return _Done() if async_op else None


def staged_broadcast(t, src=0, group=None, async_op=False):
    h = t.detach().cpu()
    _broadcast(h, src=src, group=group)
    t.copy_(h)
    return _Done() if async_op else None


def main():
    import faulthandler
    faulthandler.dump_traceback_later(40, exit=True)      # a stalled rank ends itself and says where it was
    dist.all_reduce, dist.broadcast = staged_all_reduce, staged_broadcast
    out_dir, steps = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    torch.manual_seed(100 + rank)                  # different initial weights per rank: the trainer must broadcast rank 0's
    model = unet_amd.UNet(3, 1).to(dev)
    trainer = DataParallelTrainer(model, lr=1e-3)
    g = torch.Generator().manual_seed(7)           # the same global batch on every rank, sharded by rank
    x = torch.rand(2 * world, 3, 32, 48, generator=g)
    m = (torch.rand(2 * world, 1, 32, 48, generator=g) > 0.5).float()
    xs, ms = x[rank::world].to(dev), m[rank::world].to(dev)
    losses = [float(trainer.train_step(xs, ms)) for _ in range(steps)]
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    torch.save({"losses": losses, "params": flat.cpu(), "x": x, "m": m}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
