"""What a family of launches costs a training step: the step is timed with the named entry points turned into no-ops
(results are wrong, only the time is of interest) -- an upper bound of what removing or folding them could gain, with
every knock-on effect (stream dependencies, launch boundaries, contention with the other stream) included.

    python tools/skip_probe.py unet|lunext  name[,name...] [name[,name...] ...]

Each argument is one experiment (on a fresh model); "none" is the unchanged step.  Names are uh_ entry points (exact).
Caveat: a skipped launch leaves its outputs stale, and what the matrix cores draw depends on the data (zeros and NaNs
run at higher clocks): on the U-Net step, which is bound by the matrix cores, the differences overstate the gain (the
36 BatchNorm finalisations "cost" 0.9 ms this way, their kernels add up to 0.39 ms); on the bandwidth-bound LUNeXt step
the numbers held up when the launches were really folded (0.135 ms bound, 0.07 ms gained for the 18 depthwise
weight-gradient finalisations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
from unet_amd import _lib
from unet_amd.dp import DataParallelTrainer, LUNeXtTrainer

DEV = torch.device("cuda", 0)
SKIP = set()
_real = _lib.call


def _call(name, *args):
    if name in SKIP:
        return None
    return _real(name, *args)


for mod in list(sys.modules.values()):
    if getattr(mod, "__name__", "").startswith("unet_amd") and getattr(mod, "call", None) is _real:
        mod.call = _call

which = sys.argv[1]
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
x = torch.rand(32, 3, 256, 256, generator=g).to(DEV)
m = (torch.rand(32, 1, 256, 256, generator=g) > 0.5).float().to(DEV)
for exp in sys.argv[2:] or ["none"]:
    SKIP.clear()
    torch.manual_seed(0)
    if which == "unet":
        tr = DataParallelTrainer(unet_amd.UNet(3, 1).to(DEV), lr=1e-3)
    else:
        tr = LUNeXtTrainer(unet_amd.LUNeXt(3, 1).to(DEV).train(), lr=1e-3, criterion="cl_dice", standalone=True)
    SKIP.update(n for n in exp.split(",") if n != "none")
    for _ in range(4):
        tr.train_step(x, m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.train_step(x, m)
    torch.cuda.synchronize()
    print(f"{exp:60s} {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms per step", flush=True)
