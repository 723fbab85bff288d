"""Is a training step bound by the host enqueueing launches or by the device?  Prints, per step, the host time to
enqueue and the wall time including the device.  python scratch/host_probe.py [lunext|unet]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
from oracle import ref_cpu
which = sys.argv[1] if len(sys.argv) > 1 else "lunext"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x, m = ref_cpu.synth_batch(32, 3, 256, 256)
x, m = x.to(dev), m.to(dev)
if which == "lunext":
    model = unet_amd.LUNeXt(3, 1).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    crit = unet_amd.soft_dice_cldice()
    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(m, torch.sigmoid(model(x)).clamp(1e-6, 1 - 1e-6))
        loss.backward()
        opt.step()
else:
    from unet_amd.dp import DataParallelTrainer
    tr = DataParallelTrainer(unet_amd.UNet(3, 1).to(dev), lr=1e-3, standalone=True)
    def step():
        tr.train_step(x, m)
for _ in range(5):
    step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"workload": which, "host_enqueue_ms_per_step": (t1 - t0) / n * 1e3, "wall_ms_per_step": (t2 - t0) / n * 1e3}))
