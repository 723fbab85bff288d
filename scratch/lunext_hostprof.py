import sys, os, time, cProfile, pstats, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from oracle import ref_cpu
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = unet_amd.LUNeXt(3, 1).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
crit = unet_amd.soft_dice_cldice()
x, m = ref_cpu.synth_batch(32, 3, 256, 256); x, m = x.to(dev), m.to(dev)
def step():
    opt.zero_grad(set_to_none=True)
    probs = torch.sigmoid(model(x)).clamp(1e-6, 1 - 1e-6)
    loss = crit(m, probs); loss.backward(); opt.step(); return loss
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {(t1-t0)/20*1e3:.2f} ms/step, total {(t2-t0)/20*1e3:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
