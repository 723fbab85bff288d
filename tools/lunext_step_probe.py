"""LUNeXt + clDice training step (unet_amd.dp.LUNeXtTrainer, batch 32, 256 x 256): time the host needs to enqueue a
step against the time the GPU needs to run it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
from unet_amd.dp import LUNeXtTrainer
DEV = torch.device("cuda", 0)
torch.manual_seed(0)
model = unet_amd.LUNeXt(3, 1).to(DEV).train()
tr = LUNeXtTrainer(model, lr=1e-3, criterion=sys.argv[1] if len(sys.argv) > 1 else "cl_dice", standalone=True)
g = torch.Generator().manual_seed(1)
x = torch.rand(32, 3, 256, 256, generator=g).to(DEV)
m = (torch.rand(32, 1, 256, 256, generator=g) > 0.5).float().to(DEV)
for _ in range(5):
    tr.train_step(x, m)
torch.cuda.synchronize()
# host only: every step starts with an idle GPU, so the enqueue time is not hidden behind a full queue
host = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(x, m)
    host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    tr.train_step(x, m)
t_host = (time.perf_counter() - t0) / 30                      # the loop's own time: how far the host runs ahead
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
print(f"host enqueue {sorted(host)[len(host) // 2] * 1e3:.3f} ms per step; back-to-back steps {dt * 1e3:.3f} ms per step "
      f"(the host left the loop after {t_host * 1e3:.3f} ms per step)")
