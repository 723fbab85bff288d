"""Worker of tests/test_dp_two_ranks_gpu.py: one rank of a data-parallel U-Net run whose ranks share one GPU.  RCCL refuses two
ranks on one device and gloo's own device-tensor path does not complete here, so the collectives are staged through
host memory: wait for the CURRENT stream (which therefore has to be ordered after every producer of the tensor,
exactly what RCCL's stream would wait for), copy to the host, gloo on the host copy, copy back on the current stream.
Writes rank<r>.pt with the losses and the parameters.

A rank whose device work does not complete within STALL_SECONDS does not sit in the copy: it prints, per stream, the
first kernel launch that has not completed ("STALL ..." lines) and exits with status 3, so a stall names the launch
it is behind.

Environment switches (diagnosis): UNET_AMD_SIDE_PRIORITY=-1 gives the second stream high priority (the setting under
which the ranks stalled); DP_WORKER_PAGEABLE=1 stages through pageable memory; DP_WORKER_TRACE=0 turns the per-launch
events off."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import unet_amd                                    # noqa: E402
from unet_amd import _lib, engine                  # noqa: E402
from unet_amd.dp import DataParallelTrainer        # noqa: E402

STALL_SECONDS = 25.0
RANK = int(os.environ.get("RANK", "0"))


class _Done:
    def wait(self):
        return True


_all_reduce, _broadcast = dist.all_reduce, dist.broadcast
_TRACE = []                                        # (entry point, stream handle, event recorded behind its launches)
_STREAMS = {}                                      # stream handle -> torch stream
_SEQ = [0]
_COLLECTIVES = [0]


def _traced_call(name, *args):
    _lib_call(name, *args)
    handle = args[-1] or 0                         # every launching entry point takes the stream last
    st = _STREAMS.get(handle)
    if st is None:
        st = _STREAMS[handle] = torch.cuda.ExternalStream(handle) if handle else torch.cuda.default_stream()
    ev = torch.cuda.Event()
    ev.record(st)
    _SEQ[0] += 1
    _TRACE.append((f"#{_SEQ[0]} {name}{tuple(a for a in args[:3] if isinstance(a, int))}", handle, ev))
    if len(_TRACE) > 4000:
        del _TRACE[:2000]


_lib_call = _lib.call


def _report_stall(where):
    print(f"STALL rank {RANK} in {where}: device work not complete after {STALL_SECONDS:.0f} s "
          f"({_COLLECTIVES[0]} collectives done)", flush=True)
    for again in (0, 1):
        per_stream = {}
        for name, handle, ev in _TRACE:
            s = per_stream.setdefault(handle, dict(pending=[], total=0, last_done=None))
            s["total"] += 1
            if ev.query():
                s["last_done"] = name
            else:
                s["pending"].append(name)
        for handle, s in per_stream.items():
            print(f"STALL rank {RANK} stream {handle:#x}: {len(s['pending'])} of the last {s['total']} launches "
                  f"incomplete; last complete {s['last_done']}; incomplete: {s['pending'][:5]}", flush=True)
        if not again:
            time.sleep(5.0)                        # slow or stopped: the same lists again
    sys.stdout.flush()
    os._exit(3)


def _wait_current_stream(where):
    ev = torch.cuda.Event()
    ev.record()
    t0 = time.time()
    while not ev.query():
        if time.time() - t0 > STALL_SECONDS:
            _report_stall(where)
        time.sleep(0.0005)


_PINNED = {}                                       # numel -> page-locked fp32 staging buffer, kept for the whole run


def _to_host(t, where):
    """t on the host, ordered after everything enqueued on the current stream.  DP_WORKER_PAGEABLE=1 stages through a
    fresh pageable tensor per call (t.cpu()), the default through page-locked buffers that live as long as the process."""
    _wait_current_stream(where)
    if os.environ.get("DP_WORKER_PAGEABLE") == "1":
        return t.detach().cpu()
    flat = t.detach().reshape(-1)
    key = (flat.numel(), flat.dtype)
    h = _PINNED.get(key)
    if h is None:
        h = _PINNED[key] = torch.empty(flat.numel(), dtype=flat.dtype, pin_memory=True)
    h.copy_(flat, non_blocking=True)
    _wait_current_stream(where + " (copy to the host)")
    return h.view(t.shape)


def _to_device(t, h, where):
    t.copy_(h, non_blocking=True)
    _wait_current_stream(where + " (copy to the device)")    # h is reused by the next call


def staged_all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
    h = _to_host(t, "all_reduce")
    _all_reduce(h, op=op, group=group)
    _to_device(t, h, "all_reduce")
    _COLLECTIVES[0] += 1
    return _Done() if async_op else None


def staged_broadcast(t, src=0, group=None, async_op=False):
    h = _to_host(t, "broadcast")
    _broadcast(h, src=src, group=group)
    _to_device(t, h, "broadcast")
    _COLLECTIVES[0] += 1
    return _Done() if async_op else None


def main():
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)      # a rank stuck anywhere else ends itself and says where it was
    dist.all_reduce, dist.broadcast = staged_all_reduce, staged_broadcast
    out_dir, steps = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if os.environ.get("DP_WORKER_TRACE", "1") != "0":
        _lib.call = engine.call = _traced_call
    dist.init_process_group("gloo")
    torch.manual_seed(100 + rank)                  # different initial weights per rank: the trainer must broadcast rank 0's
    model = unet_amd.UNet(3, 1).to(dev)
    trainer = DataParallelTrainer(model, lr=1e-3)
    g = torch.Generator().manual_seed(7)           # the same global batch on every rank, sharded by rank
    x = torch.rand(2 * world, 3, 32, 48, generator=g)
    m = (torch.rand(2 * world, 1, 32, 48, generator=g) > 0.5).float()
    xs, ms = x[rank::world].to(dev), m[rank::world].to(dev)
    losses = []
    for _ in range(steps):
        losses.append(float(trainer.train_step(xs, ms)))
    _wait_current_stream("end of training")
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    torch.save({"losses": losses, "params": flat.cpu(), "x": x, "m": m}, os.path.join(out_dir, f"rank{rank}.pt"))
    print(f"rank {rank} done: losses {losses}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
