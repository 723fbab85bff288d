import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd.dp import DataParallelTrainer
from oracle import ref_cpu
dev = torch.device("cuda", 0)
x, m = ref_cpu.synth_batch(8, 3, 64, 64); x, m = x.to(dev), m.to(dev)
res = {}
for kind in ("flat", "torch"):
    torch.manual_seed(0)
    model = unet_amd.UNet(3, 1).to(dev)
    tr = DataParallelTrainer(model, lr=1e-3)
    if kind == "torch":
        tr.optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    res[kind] = [tr.train_step(x, m).item() for _ in range(12)]
for a, b in zip(res["flat"], res["torch"]):
    print(f"{a:.6f} {b:.6f}")
