"""N calls of the clDice loss (forward + backward with respect to the probabilities) at batch 32, 256 x 256: the program
the profiler passes of tools/prof_cldice.sh run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
DEV = torch.device("cuda", 0)
g = torch.Generator().manual_seed(3)
m = (torch.rand(32, 1, 256, 256, generator=g) > 0.5).float().to(DEV)
p = torch.rand(32, 1, 256, 256, generator=g).clamp(1e-6, 1 - 1e-6).to(DEV).requires_grad_(True)
crit = unet_amd.soft_dice_cldice()
for _ in range(N):
    torch.autograd.grad(crit(m, p), p)
torch.cuda.synchronize()
print("calls", N)
