"""LUNeXt + clDice training step: GPU time of its phases (forward, loss, backward, optimizer) from events recorded on the
main stream between them, in back-to-back steps; and the same with the second stream switched off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
from unet_amd.dp import LUNeXtTrainer
from unet_amd.lunext import LUNeXtRun
DEV = torch.device("cuda", 0)
torch.manual_seed(0)
model = unet_amd.LUNeXt(3, 1).to(DEV).train()
tr = LUNeXtTrainer(model, lr=1e-3, criterion=sys.argv[1] if len(sys.argv) > 1 else "cl_dice", standalone=True)
g = torch.Generator().manual_seed(1)
x = torch.rand(32, 3, 256, 256, generator=g).to(DEV)
m = (torch.rand(32, 1, 256, 256, generator=g) > 0.5).float().to(DEV)


def step(marks=None):
    ev = lambda: marks.append(torch.cuda.Event(enable_timing=True)) or marks[-1].record() if marks is not None else None
    ev()
    run = LUNeXtRun(model, x, training=True)
    logits = run.forward()
    ev()
    loss, dlogits = tr._loss(logits, m)
    ev()
    run.backward(dlogits, gflat=tr.grads)
    run.release()
    ev()
    tr.optimizer.step()
    ev()
    return loss


for _ in range(5):
    step()
torch.cuda.synchronize()
N = 20
all_marks = []
for _ in range(N):
    mk = []
    step(mk)
    all_marks.append(mk)
torch.cuda.synchronize()
names = ["forward", "loss", "backward", "optimizer"]
tot = [0.0] * 4
for mk in all_marks:
    for i in range(4):
        tot[i] += mk[i].elapsed_time(mk[i + 1])
span = all_marks[0][0].elapsed_time(all_marks[-1][-1]) / N
print("  ".join(f"{n} {t / N:.3f} ms" for n, t in zip(names, tot)), f"| sum {sum(tot) / N:.3f} ms, step-to-step {span:.3f} ms")
