"""LUNeXt + clDice training step captured into a HIP graph: does it capture, does it replay to the same losses, how
fast is it.  python scratch/graph_probe.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
from oracle import ref_cpu
dev = torch.device("cuda", 0)
x, m = ref_cpu.synth_batch(32, 3, 256, 256)
x, m = x.to(dev), m.to(dev)


def build():
    torch.manual_seed(0)
    model = unet_amd.LUNeXt(3, 1).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
    crit = unet_amd.soft_dice_cldice()

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(m, torch.sigmoid(model(x)).clamp(1e-6, 1 - 1e-6))
        loss.backward()
        opt.step()
        return loss
    return model, step


# eager reference: losses of steps 1..8
model, step = build()
eager = [float(step()) for _ in range(8)]
# graphed: 3 eager warm-up steps on a side stream, then capture step 4 and replay
model, step = build()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
losses = []
with torch.cuda.stream(s):
    for _ in range(3):
        losses.append(float(step()))
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
# capture does not execute: replay now runs step 4
for _ in range(5):
    g.replay()
    losses.append(float(static_loss))
print("eager  ", [round(v, 6) for v in eager])
print("graphed", [round(v, 6) for v in losses])
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(json.dumps({"graph_ms_per_step": (t1 - t0) / n * 1e3}))
